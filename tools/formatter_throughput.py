#!/usr/bin/env python3
"""Formatter and writer of the CLI, separated from the GPU and from each other (C3POa.py:141-173 writes the
two record streams of every read):  c3_write_group of one real GPU batch (cfg5 shape)
  * formatter alone (C3_WRITER_NO_IO=1), by thread count,
  * formatter + pwrite onto tmpfs (/dev/shm) and onto the box's disk (--disk DIR), by thread count.
python tools/formatter_throughput.py [reads per batch] [--disk DIR]"""
import argparse, json, os, shutil, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from c3poa_amd import _lib, synth

ap = argparse.ArgumentParser()
ap.add_argument("n", type=int, nargs="?", default=131072)
ap.add_argument("--disk", default=None)
a = ap.parse_args()
shm = tempfile.mkdtemp(prefix="c3fmt_", dir="/dev/shm")
disk = tempfile.mkdtemp(prefix="c3fmt_", dir=a.disk or tempfile.gettempdir())
try:
    uniq = list(synth.generate("cfg5", n_reads=min(a.n, 8192)))
    fq = shm + "/reads.fastq"
    with open(fq, "w") as fh:
        for k in range(a.n):
            r = uniq[k % len(uniq)]
            fh.write("@r%08d\n%s\n+\n%s\n" % (k, r[1], r[2]))
    rd = _lib.Reader(fq, n_sets=1)
    hb = rd.next(a.n)
    strands = "".join(uniq[k % len(uniq)][3] for k in range(hb.n))
    h = _lib.Handle()
    h.set_splints([synth.SPLINT1])
    sid = np.zeros(hb.n, dtype=np.int16)
    h.upload_host(hb, strands.encode(), sid)
    h.run()
    res, buf, coff = h.results_raw()
    out = {"reads_per_batch": int(hb.n), "consensus_ok": int((res["status"] == 0).sum()), "host_cores": os.cpu_count(), "rows": []}
    for label, d, noio in (("format_only", shm, True), ("tmpfs", shm, False), ("disk", disk, False)):
        for T in (1, 2, 4, 8, 16, 32):
            os.environ["C3_WRITER_THREADS"] = str(T)
            if noio:
                os.environ["C3_WRITER_NO_IO"] = "1"
            else:
                os.environ.pop("C3_WRITER_NO_IO", None)
            cp, sp = [d + "/c.fa"], [d + "/s.fq"]
            best = None
            for rep in range(3):
                for p in cp + sp:
                    open(p, "w").close()
                _lib.load().c3_writer_reset()
                t0 = time.perf_counter()
                _lib.write_group(hb, res, buf, coff, sid, cp, sp, True)
                if not noio and label == "disk":
                    os.sync()
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            nbytes = 0 if noio else os.path.getsize(cp[0]) + os.path.getsize(sp[0])
            out["rows"].append({"what": label, "threads": T, "seconds": round(best, 4), "reads_per_s": round(hb.n / best, 1), "output_GB": round(nbytes / 1e9, 3)})
            print(out["rows"][-1], file=sys.stderr)
    print(json.dumps(out))
finally:
    shutil.rmtree(shm, ignore_errors=True); shutil.rmtree(disk, ignore_errors=True)
