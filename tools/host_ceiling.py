#!/usr/bin/env python3
"""What ONE host can feed: the streamed CLI pipeline of c3poa_amd/stream.py -- native byte-range readers -> splint / strand assignment
-> [GPU] -> native formatter -> pwrite -- with the GPU replaced by a stand-in that answers at once (every read gets three
subreads, two dangling pieces and a 1.5 kb consensus; no kernels, no device).  Everything else is the product code, so the
rate it reaches with N workers is the ceiling a node's host side puts on N GPUs (cfg5: 8 GPUs x ~0.4 M reads/s = 3.2 M reads/s
= 32 GB/s of FASTQ in, 37 GB/s out).

    python tools/host_ceiling.py [N reads] [--dir /dev/shm] [--workers 1,2,4,8] [--gz]

Prints one JSON line per worker count; profiles/rNN_host_ceiling.json keeps them."""
import argparse
import json
import multiprocessing as mp
import os
import shutil
import sys
import tempfile
import time
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from c3poa_amd import _lib, stream, synth  # noqa: E402
from c3poa_amd.seqio import revcomp  # noqa: E402


class InstantHandle:
    """stand-in for _lib.Handle: the methods stream.run calls, no device; results are fabricated with numpy in ~1 ms per batch"""

    def __init__(self, **cfg):
        self.cfg = types.SimpleNamespace(conk_match=5)
        self.cur = self.staged = self.snap = None
        self.last_timing = {"ms_pack": 0.0, "ms_total": 0.0, "ms_wall": 0.0, "ms_alloc": 0.0}

    def set_splints(self, splints):
        pass

    def upload_host(self, hb, strands, splint_ids):
        self.cur = (hb.n, hb.off)

    def stage_host(self, hb, strands, splint_ids):
        self.staged = (hb.n, hb.off)

    def commit(self):
        self.cur, self.staged = self.staged, None

    def run(self):
        pass

    def results_snapshot(self):
        assert self.snap is None
        self.snap = self.cur
        return self.cur[0], int(self.cur[1][-1]) + 16

    def results_fetch(self, into, shape):
        n, off = self.snap
        res, buf, coff = into.fit(n, int(off[-1]) + 16)
        L = (off[1:] - off[:-1]).astype(np.int32)
        res[:] = np.zeros(1, dtype=_lib.RESULT_DTYPE)
        res["n_sub"] = 3; res["has_front"] = 1; res["has_tail"] = 1; res["n_peaks"] = 4
        res["front_end"] = 250; res["tail_beg"] = L - 250
        step = (L - 500) // 3
        for k in range(3):
            res["sub_beg"][:, k] = 250 + k * step
            res["sub_end"][:, k] = 250 + (k + 1) * step
        clen = np.minimum(1500, np.maximum(L - 16, 1)).astype(np.int64)
        res["cons_len"] = clen
        coff[0] = 0
        np.cumsum(clen, out=coff[1:n + 1])
        buf[:int(coff[n])] = 65                      # 'A'
        self.snap = None
        return res, buf, coff

    def close(self):
        pass


def _bgzf_chunk(args):
    import struct
    import zlib
    path, beg, end = args
    out = []
    with open(path, "rb") as f:
        f.seek(beg)
        data = f.read(end - beg)
    for i in range(0, len(data), 0xff00):
        chunk = data[i:i + 0xff00]
        co = zlib.compressobj(1, zlib.DEFLATED, -15)
        comp = co.compress(chunk) + co.flush()
        out.append(struct.pack("<BBBBIBBH", 0x1f, 0x8b, 8, 4, 0, 0, 0xff, 6) + b"BC" + struct.pack("<HH", 2, len(comp) + 25) + comp
                   + struct.pack("<II", zlib.crc32(chunk), len(chunk)))
    return b"".join(out)


def bgzf_compress(src, dst, procs=16):
    """bgzip layout (independent members of <= 64 KiB with the 'BC' size subfield + the empty end member), written with zlib"""
    size = os.path.getsize(src)
    step = 0xff00 * 256
    jobs = [(src, b, min(b + step, size)) for b in range(0, size, step)]
    with mp.Pool(procs) as pool, open(dst, "wb") as f:
        for part in pool.imap(_bgzf_chunk, jobs):
            f.write(part)
        f.write(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))


def _gen(job):
    s0, cnt, path = job
    st = []
    with open(path, "w") as fh:
        for r in synth.generate("cfg5", n_reads=cnt, start=s0):
            fh.write("@%s\n%s\n+\n%s\n" % (r[0], r[1], r[2]))
            st.append(r[3])
    return "".join(st)


def _cpu_stat():
    """cgroup v2 CPU accounting of this container (usage, throttling by the CPU quota)"""
    try:
        return {l.split()[0]: int(l.split()[1]) for l in open("/sys/fs/cgroup/cpu.stat").read().splitlines() if len(l.split()) == 2}
    except Exception:
        return {}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("n", type=int, nargs="?", default=1000000)
    ap.add_argument("--dir", default="/dev/shm")
    ap.add_argument("--workers", default="1,2,4,8")
    ap.add_argument("--gz", action="store_true")
    ap.add_argument("--bgzf", action="store_true", help="BGZF-compressed input (bgzip layout): members inflated in parallel by the native reader")
    ap.add_argument("--unique", type=int, default=100000, help="distinct reads generated; the file repeats them (names stay distinct)")
    a = ap.parse_args()
    d = tempfile.mkdtemp(prefix="c3host_", dir=a.dir)
    try:
        t0 = time.time()
        nu = min(a.unique, a.n)
        chunk = 5000
        jobs = [(s0, min(chunk, nu - s0), "%s/part%06d.fastq" % (d, s0)) for s0 in range(0, nu, chunk)]
        with mp.Pool(min(16, os.cpu_count() or 1)) as pool:
            strands = "".join(pool.map(_gen, jobs))
        fq = d + "/reads.fastq"
        block = b"".join(open(j[2], "rb").read() for j in jobs)
        for j in jobs:
            os.remove(j[2])
        # repeat the block with distinct names: r%08d -> the copy index goes into the first digits
        with open(fq, "wb") as fh:
            for c in range((a.n + nu - 1) // nu):
                fh.write(block.replace(b"@r0", b"@r%d" % (c % 10)) if c else block)
        n_total = ((a.n + nu - 1) // nu) * nu
        if a.gz:
            os.system("gzip -1 %s" % fq); fq += ".gz"
        elif a.bgzf:
            bgzf_compress(fq, fq + ".gz"); os.remove(fq); fq += ".gz"
        size = os.path.getsize(fq)
        print("generated %d reads (%d distinct), %.2f GB in %.1f s" % (n_total, nu, size / 1e9, time.time() - t0), file=sys.stderr)
        sd = {"Splint1": [synth.SPLINT1, revcomp(synth.SPLINT1)]}
        # splint / strand of every read through the native PSL table, as the CLI does (bin/preprocess.py:22-45)
        psl = d + "/splint_to_read_alignments.psl"
        with open(psl, "w") as fh:
            for c in range(n_total // nu):
                for k in range(nu):
                    name = "r%d%07d" % (c % 10, k) if c else "r%08d" % k
                    fh.write("\t".join(["280", "4", "0", "0", "0", "0", "0", "0", strands[k], name, "5000", "0", "284", "Splint1", "284", "0", "284", "1", "284,", "0,", "0,"]) + "\n")
        _lib.Handle = InstantHandle
        _lib.device_count = lambda: 64
        for w in [int(x) for x in a.workers.split(",")]:
            out = "%s/out%d/" % (d, w)
            os.makedirs(out + "Splint1")
            args = types.SimpleNamespace(out_path=out, reads=fq, groupSize=1000, lencutoff=1000, mdistcutoff=500, zero=True, compress_output=False)
            st = {}
            cg0, tm0 = _cpu_stat(), os.times()
            t0 = time.time()
            assigner = _lib.Assigner(psl, ["Splint1"])
            t_psl = time.time() - t0
            n = stream.run(args, sd, assigner, {"Splint1"}, w, stats=st)
            dt = time.time() - t0
            assigner.close()
            cg1, tm1 = _cpu_stat(), os.times()
            cpu = {"user_s": round(tm1.user - tm0.user, 2), "sys_s": round(tm1.system - tm0.system, 2),
                   "cgroup": {k: cg1[k] - cg0.get(k, 0) for k in cg1 if k in ("usage_usec", "nr_periods", "nr_throttled", "throttled_usec")}}
            osz = sum(os.path.getsize(out + "Splint1/" + f) for f in os.listdir(out + "Splint1"))
            print(json.dumps({"workers": w, "reads": n, "seconds": round(dt, 2), "reads_per_s": round(n / dt, 1), "input_GB": round(size / 1e9, 2),
                              "output_GB": round(osz / 1e9, 2), "gz": "bgzf" if a.bgzf else a.gz, "gz_threads": os.environ.get("C3_GZ_THREADS"), "ranges": st.get("ranges"), "psl_table_s": round(t_psl, 2), "host_threads": os.environ.get("C3_HOST_THREADS"), "cpu": cpu,
                              "stage_s": {k: round(st[k], 2) for k in ("parse", "assign", "upload", "fetch", "write", "wait_in", "wait_out") if k in st}}))
            shutil.rmtree(out, ignore_errors=True)
    finally:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
