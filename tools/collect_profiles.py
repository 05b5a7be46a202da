#!/usr/bin/env python3
"""Turn the rocprofv3 output that a gpurun call merged into gpurun_out/ into the summaries kept under profiles/.

  python tools/collect_profiles.py TAG STATS_DIR FETCH_DIR WRITE_DIR [BENCH_JSON] [WORKLOAD e.g. cfg4_100k]

STATS_DIR : `rocprofv3 --kernel-trace --stats -d STATS_DIR -o s -- python3 bench.py --steps 1 --warmup 0 --no-cpu`
FETCH_DIR : `rocprofv3 --kernel-trace --pmc FETCH_SIZE -d FETCH_DIR -o b -- python3 bench.py ...` (own pass)
WRITE_DIR : same with WRITE_SIZE (own pass; counters are never combined with other trace domains)
HBM bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024: on this device FETCH_SIZE counts half of the bytes of dword loads
(tools/pmc_calibrate.py, 1 GiB each way), WRITE_SIZE is exact.
"""
import csv, glob, json, os, shutil, sys
from collections import defaultdict


def kname(raw):
    """kernel name without return type / arguments; k_window is a template (<false> = the launch that does the work, <true> =
    the full-size second launch for windows that overflowed the first launch's DP scratch)"""
    k = raw.split("(")[0].replace("void ", "")
    import re
    k = k.replace("k_window<false>", "k_window").replace("k_window<true>", "k_window_second_launch")
    # k_poa<W32, DEF, WIDE>: the first-pass instances are "k_poa", the 32-bit pass its own line
    m = re.match(r"k_poa<(true|false), (true|false), (true|false)>", k)
    if m:
        k = "k_poa_32bit_pass" if m.group(1) == "true" else "k_poa"
    return k


def kernel_src_sha(root):
    """same hash as bench.py's kernel_src_sha(): the kernel sources these counters were measured on"""
    import hashlib
    hsh = hashlib.sha1()
    for f in sorted(glob.glob(os.path.join(root, "c3poa_amd", "csrc", "*.hip")) + glob.glob(os.path.join(root, "c3poa_amd", "csrc", "*.h"))):
        hsh.update(os.path.basename(f).encode() + b"\0" + open(f, "rb").read())
    return hsh.hexdigest()[:16]


def counter_sum(d, name):
    acc, calls = defaultdict(float), defaultdict(int)
    for f in glob.glob(os.path.join(d, "*results.db")):          # rocprofv3 default output (rocpd sqlite)
        import sqlite3
        for kn, v in sqlite3.connect(f).execute("select kernel_name, value from counters_collection where counter_name = ?", (name,)):
            k = kname(kn)
            acc[k] += float(v); calls[k] += 1
    for f in glob.glob(os.path.join(d, "*counter_collection.csv")):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == name:
                k = kname(row["Kernel_Name"])
                acc[k] += float(row["Counter_Value"]); calls[k] += 1
    return {k: acc[k] / calls[k] for k in acc}


def main():
    tag, stats_dir, fdir, wdir = sys.argv[1:5]
    wl = sys.argv[6] if len(sys.argv) > 6 else "cfg2_100k"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prof = os.path.join(root, "profiles")
    dst = os.path.join(prof, "%s_kernel_stats_bench_%s.csv" % (tag, wl))
    csvs = glob.glob(os.path.join(stats_dir, "*kernel_stats.csv"))
    if csvs:
        shutil.copy(csvs[0], dst)
    else:                                                           # same table from the rocpd database
        import sqlite3
        db = sqlite3.connect(glob.glob(os.path.join(stats_dir, "*results.db"))[0])
        rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name order by 3 desc").fetchall()
        tot = sum(r[2] for r in rows)
        with open(dst, "w") as fh:
            fh.write('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"\n')
            for r in rows:
                fh.write('"%s",%d,%d,%.1f,%.2f,%d,%d\n' % (r[0], r[1], r[2], r[3], 100.0 * r[2] / tot, r[4], r[5]))
    st = dst
    fetch, write = counter_sum(fdir, "FETCH_SIZE"), counter_sum(wdir, "WRITE_SIZE")
    out = {"note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on `python3 bench.py --steps 1 "
                   "--warmup 0 --no-cpu` (%s); KB per launch as reported, averaged over launches. " % wl +
                   "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (FETCH_SIZE reports 1/2 of dword loads on this device, "
                   "calibrated with tools/pmc_calibrate.py; WRITE_SIZE exact).",
           "calibration": {"source": "tools/pmc_calibrate.py under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE on this pool (round 1)",
                           "bytes_read": 1073741824, "bytes_written": 1073741824, "FETCH_SIZE_KB": 524307.25, "WRITE_SIZE_KB": 1048576.0},
           "workload": wl, "kernel_src_sha": kernel_src_sha(root), "kernels": {}}
    for k in sorted(set(fetch) | set(write), key=lambda k: -(2 * fetch.get(k, 0) + write.get(k, 0))):
        if not k.startswith("k_"):
            continue
        out["kernels"][k] = {"FETCH_SIZE_KB": fetch.get(k, 0.0), "WRITE_SIZE_KB": write.get(k, 0.0),
                             "hbm_bytes_per_launch": (2 * fetch.get(k, 0.0) + write.get(k, 0.0)) * 1024}
    json.dump(out, open(os.path.join(prof, "%s_pmc_traffic_%s.json" % (tag, wl)), "w"), indent=1)
    if len(sys.argv) > 5:
        shutil.copy(sys.argv[5], os.path.join(prof, "%s_bench_under_rocprof_%s.json" % (tag, wl)))
    print(open(st).read())
    print(json.dumps(out["kernels"], indent=1))


if __name__ == "__main__":
    main()
