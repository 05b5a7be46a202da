#!/usr/bin/env python3
"""Host-only measurement: the native reader over a plain .gz FASTQ file (and the same file in BGZF layout) with the own DEFLATE decoder
against zlib (C3_GZ_ZLIB=1), reads and MB of FASTQ per second.  CPU only.
    python tools/inflate_bench.py [n_reads] [--dir /dev/shm] [--level 6]"""
import argparse
import os
import subprocess
import sys
import tempfile
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CHILD = r'''
import sys, time
sys.path.insert(0, %r)
from c3poa_amd import _lib
rd = _lib.Reader(sys.argv[1], n_sets=2)
t0 = time.perf_counter(); n = 0; nb = 0
while True:
    hb = rd.next(65536, 0, 1 << 30)
    if hb.n == 0: break
    n += hb.n; nb += int(hb.off[hb.n])
dt = time.perf_counter() - t0
print("%%d %%d %%.3f" %% (n, nb, dt))
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("n", type=int, nargs="?", default=100000)
    ap.add_argument("--dir", default="/dev/shm")
    ap.add_argument("--level", type=int, default=6)
    a = ap.parse_args()
    from c3poa_amd import synth, _lib
    d = tempfile.mkdtemp(prefix="c3inf_", dir=a.dir)
    try:
        fq = os.path.join(d, "reads.fastq")
        with open(fq, "w") as fh:
            for r in synth.generate("cfg5", n_reads=a.n):
                fh.write("@%s\n%s\n+\n%s\n" % (r[0], r[1], r[2]))
        raw = os.path.getsize(fq)
        subprocess.check_call("gzip -%d -k %s" % (a.level, fq), shell=True)
        gz = fq + ".gz"
        bg = os.path.join(d, "reads_bgzf.fastq.gz")
        _lib.compress_file(fq, bg, level=a.level, remove=False)
        print("%d reads, %.2f GB of FASTQ; gzip -%d %.2f GB, BGZF %.2f GB" % (a.n, raw / 1e9, a.level, os.path.getsize(gz) / 1e9, os.path.getsize(bg) / 1e9))
        for path, what in ((gz, "plain gzip"), (bg, "BGZF")):
            # (plain gzip since round 6: "own decoder" = several inflating threads behind the parser, c3_gzpar.hpp; C3_GZ_SERIAL=1 = round 5's single thread)
            variants = [({"C3_GZ_ZLIB": "1"}, "zlib"), ({}, "own decoder")]
            if what == "plain gzip":
                variants.insert(1, ({"C3_GZ_SERIAL": "1"}, "own, 1 thread"))
            for env, name in variants + variants:
                e = dict(os.environ); e.pop("C3_GZ_ZLIB", None); e.pop("C3_GZ_SERIAL", None); e.update(env)
                out = subprocess.run([sys.executable, "-c", CHILD, path], env=e, capture_output=True, text=True)
                if out.returncode:
                    print(what, name, "FAILED", out.stderr[-300:]); continue
                n, nb, dt = out.stdout.split()
                print("%-10s %-14s %7.1f k reads/s  %6.0f MB of FASTQ per s" % (what, name, int(n) / float(dt) / 1e3, raw / float(dt) / 1e6), flush=True)
    finally:
        subprocess.call(["rm", "-rf", d])


if __name__ == "__main__":
    main()
