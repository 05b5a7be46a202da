"""Probe: P handles (pipelines) on ONE GPU, each driven by its own host thread over its own streams, against one handle.

Each pipeline runs the bench step (stage || run -> results -> commit) on its own copy of a cfg batch.  Kernels of different
pipelines share the device: the work-queue tails, the host gaps between kernels and the results fetch of one pipeline are
filled by the other's kernels.  Usage: python tools/dual_pipeline_probe.py [cfg2] [reads_total] [P ...]
"""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, ROOT + "/tools")


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
    total = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
    plist = [int(x) for x in sys.argv[3:]] or [1, 2]
    import bench
    from c3poa_amd import synth
    recs = bench.make_reads(cfg, min(total, 20000), 0, 8)
    from c3poa_amd import _lib
    md = synth.CONFIGS[cfg]["mdist"]
    steps = 4
    for P in plist:
        n = total // P
        reps = (n + len(recs) - 1) // len(recs)
        seqs = ([r[0] for r in recs] * reps)[:n]; quals = ([r[1] for r in recs] * reps)[:n]; strands = ([r[2] for r in recs] * reps)[:n]
        lens = np.array([len(s) for s in seqs], dtype=np.int64)
        off = np.zeros(n + 1, dtype=np.int64); np.cumsum(lens, out=off[1:])
        sq, ql, st = "".join(seqs).encode(), "".join(quals).encode(), "".join(strands)
        pipes = []
        for p in range(P):
            h = _lib.Handle(device=0, mdistcutoff=md); h.set_splints([synth.SPLINT1])
            host = _lib.PinnedBatch(sq, ql, off, st)
            h.upload_pinned(host)
            pipes.append((h, host))
        digests = [None] * P

        def loop(p, k):
            h, host = pipes[p]
            for _ in range(k):
                h.stage_pinned(host); h.run(); out = h.results_raw(); h.commit()
            digests[p] = (int((out[0]["status"] == 0).sum()), int(out[2][-1]))

        def run_all(k):
            th = [threading.Thread(target=loop, args=(p, k)) for p in range(P)]
            t0 = time.perf_counter()
            for t in th: t.start()
            for t in th: t.join()
            return time.perf_counter() - t0
        run_all(1)
        dt = run_all(steps)
        print("P=%d pipelines x %d reads: %d steps each in %.1f ms -> %.1f k reads/s (%.2f ms per %d reads); ok/bytes %s" % (
            P, n, steps, dt * 1e3, n * P * steps / dt / 1e3, dt / steps * 1e3, n * P, digests), flush=True)
        for h, host in pipes:
            h.close(); host.close()


if __name__ == "__main__":
    main()
