#!/usr/bin/env python3
"""Throughput of the post-processing step on N synthetic consensus reads (1.5 kb, 5'/3' adapters):
adapter finder alone (c3_scan_adapters) and the whole CLI (FASTA in -> trimmed FASTA out)."""
import os, shutil, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import C3POa_postprocessing as P
from c3poa_amd import _lib
from c3poa_amd.seqio import revcomp

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
rng = np.random.default_rng(1)
acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
rnd = lambda L: acgt[rng.integers(0, 4, L)].tobytes().decode()   # noqa: E731
a5, a3 = rnd(33), rnd(36)
d = tempfile.mkdtemp(prefix="c3post_")
try:
    recs = []
    for i in range(n):
        cdna = rnd(1400)
        recs.append(("c%07d_12.0_5000_3_1500" % i, rnd(20) + a5 + cdna + revcomp(a3) + rnd(20)))
    with open(d + "/cons.fasta", "w") as fh:
        for nm, s in recs:
            fh.write(">%s\n%s\n" % (nm, s))
    open(d + "/ad.fasta", "w").write(">3Prime_adapter\n%s\n>5Prime_adapter\n%s\n" % (a3, a5))
    h = _lib.Handle(); h.set_splints([a3, a5])
    h.upload([r[1] for r in recs], ["!" * len(r[1]) for r in recs], "?" * n)
    h.scan_adapters()
    t = time.time(); tab = h.scan_adapters(); dt = time.time() - t
    cells = sum(len(r[1]) for r in recs) * (33 + 36) * 2
    print('{"adapter_finder_reads_per_s": %.0f, "gcups": %.1f, "ms": %.1f}' % (n / dt, cells / dt / 1e9, dt * 1e3))
    h.close()
    t = time.time()
    k = P.main(P.parse_args(["-i", d + "/cons.fasta", "-a", d + "/ad.fasta", "-o", d + "/out", "-t"]))
    dt = time.time() - t
    print('{"post_cli_reads_per_s": %.0f, "reads": %d, "written": %d, "seconds": %.2f}' % (n / dt, n, k, dt))
finally:
    shutil.rmtree(d, ignore_errors=True)
