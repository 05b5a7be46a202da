#!/usr/bin/env python3
"""Reader-only throughput of the native FASTQ parser with T byte-range readers in T threads (what the streamed CLI runs
per GPU worker):  python tools/reader_throughput.py [N_READS] [T ...]
Generates a cfg5-shaped FASTQ (or takes --file), parses it fully (names, bases, qualities into SoA batches) and reports
reads/s and GB/s for every T."""
import argparse, os, shutil, sys, tempfile, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from c3poa_amd import _lib, synth


def _gen(job):
    s0, cnt, path = job
    with open(path, "w") as fh:
        for r in synth.generate("cfg5", n_reads=cnt, start=s0):
            fh.write("@%s\n%s\n+\n%s\n" % (r[0], r[1], r[2]))


ap = argparse.ArgumentParser()
ap.add_argument("n", type=int, nargs="?", default=300000)
ap.add_argument("threads", type=int, nargs="*", default=[1, 2, 4, 8, 16])
ap.add_argument("--file", default=None)
a = ap.parse_args()
d = tempfile.mkdtemp(prefix="c3rd_")
try:
    fq = a.file
    if fq is None:
        import multiprocessing as mp
        fq = d + "/reads.fastq"
        jobs = [(s0, min(5000, a.n - s0), "%s/p%06d" % (d, s0)) for s0 in range(0, a.n, 5000)]
        with mp.Pool(min(16, os.cpu_count() or 1)) as pool:
            pool.map(_gen, jobs)
        with open(fq, "wb") as fh:
            for j in jobs:
                with open(j[2], "rb") as src:
                    shutil.copyfileobj(src, fh, 1 << 24)
                os.remove(j[2])
    size = os.path.getsize(fq)
    for T in a.threads:
        counts = [0] * T
        cuts = [size * k // T for k in range(T)] + [-1]

        def work(k):
            rd = _lib.Reader(fq, n_sets=2, byte_range=(cuts[k], cuts[k + 1])) if T > 1 else _lib.Reader(fq, n_sets=2)
            while True:
                hb = rd.next(16384, 1000, 1 << 30)           # small groups: the page-locked buffers stay small, the parse rate is measured
                if hb.n == 0:
                    break
                counts[k] += hb.n
            rd.close()
        th = [threading.Thread(target=work, args=(k,)) for k in range(T)]
        t0 = time.perf_counter()
        for x in th:
            x.start()
        for x in th:
            x.join()
        dt = time.perf_counter() - t0
        print('{"reader_threads": %d, "reads": %d, "reads_per_s": %.0f, "GB_per_s": %.2f, "seconds": %.2f}' % (T, sum(counts), sum(counts) / dt, size / dt / 1e9, dt))
finally:
    shutil.rmtree(d, ignore_errors=True)
