"""Runs bench.py's worker path (bench.main) on a box WITHOUT GPUs, one process per rank under torch.distributed.run: the HIP library
binding (c3poa_amd._lib) is replaced by a stand-in that records which device every rank asked for and answers through the oracle
(test infrastructure), torch.cuda's device calls are recorded instead of executed, and the "nccl" process group the worker asks
for is recorded and created over gloo.  Used by tests/test_two_gpu_readiness_cpu.py; never imported by the product."""
import json
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

REC = {"local_rank": int(os.environ.get("LOCAL_RANK", "-1")), "rank": int(os.environ.get("RANK", "-1")),
       "HIP_VISIBLE_DEVICES": os.environ.get("HIP_VISIBLE_DEVICES"), "set_device": [], "handles": [], "pg": None, "events": []}

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

torch.cuda.is_available = lambda: True
def _set_device(d):
    REC["set_device"].append(int(d.index if hasattr(d, "index") else d)); REC["events"].append("set_device")


torch.cuda.set_device = _set_device
torch.cuda.synchronize = lambda *a, **k: None
_real_init = dist.init_process_group


def _init(backend=None, **kw):
    REC["pg"] = {"backend": backend, "device_id": str(kw.get("device_id"))}; REC["events"].append("init_process_group")
    return _real_init("gloo")          # (no RCCL without GPUs; the collectives of bench.py are the barrier, the MAX and two gathers)


dist.init_process_group = _init

from c3poa_amd import _lib as real_lib  # noqa: E402  (ctypes declarations only: the shared library is not loaded)
from c3poa_amd import synth  # noqa: E402

stub = types.ModuleType("c3poa_amd._lib")
stub.RESULT_DTYPE = real_lib.RESULT_DTYPE


class PinnedBatch:
    def __init__(self, seq_cat, qual_cat, off, strands):
        self.seq, self.qual, self.off, self.strands = seq_cat, qual_cat, np.asarray(off), strands

    def close(self):
        pass


class Handle:
    def __init__(self, **cfg):
        REC["handles"].append(cfg.get("device")); REC["events"].append("c3_create")
        self.md = cfg.get("mdistcutoff", 500)
        self.cur = self.staged = None
        self.last_timing = None

    def set_splints(self, s):
        self.splint = s[0]

    def upload_pinned(self, pb, splint_ids=None):
        self.cur = pb

    def stage_pinned(self, pb, splint_ids=None):
        self.staged = pb

    def commit(self):
        self.cur, self.staged = self.staged, None

    def run(self):
        z = {k: 1.0 for k in ("ms_conk", "ms_peaks", "ms_poa", "ms_prep", "ms_window", "ms_stitch", "ms_wall", "ms_host_worklist", "ms_alloc", "ms_host_gap")}
        z.update({k: 1 for k in ("cells_conk", "cells_poa", "cells_polish", "cells_polish_computed", "n_band_layers", "n_band_fallback", "n_windows", "n_win_redo", "n_poa_redo", "n_poa_redo16")})
        self.last_timing = z

    def close(self):
        pass


class ResultFetcher:
    def __init__(self, handle, n_buffers=2, pinned=False):
        self.h = handle

    def after_run(self):
        self.batch = self.h.cur
        return None

    def drain(self):
        from oracle import oracle_py as O
        pb = self.batch
        n = len(pb.off) - 1
        reads = [(pb.seq[pb.off[i]:pb.off[i + 1]].decode(), pb.qual[pb.off[i]:pb.off[i + 1]].decode()) for i in range(n)]
        ores, ocons = O.process_batch(self.h.splint, reads, list(pb.strands), params=O.default_params(mdistcutoff=self.h.md), threads=2)
        res = np.zeros(n, dtype=real_lib.RESULT_DTYPE)
        coff = np.zeros(n + 1, dtype=np.int64)
        for i, (r, c) in enumerate(zip(ores, ocons)):
            res[i]["status"] = r.status; res[i]["n_peaks"] = r.n_peaks; res[i]["cons_len"] = len(c)
            coff[i + 1] = coff[i] + len(c)
        return res, np.frombuffer("".join(ocons).encode() + b"\0" * 16, dtype=np.uint8), coff

    def close(self):
        pass


stub.PinnedBatch, stub.Handle, stub.ResultFetcher = PinnedBatch, Handle, ResultFetcher
sys.modules["c3poa_amd._lib"] = stub
import c3poa_amd  # noqa: E402
c3poa_amd._lib = stub

import bench  # noqa: E402

sys.argv = ["bench.py"] + sys.argv[1:]
try:
    bench.main()
finally:
    json.dump(REC, open(os.path.join(os.environ["C3_STUB_OUT"], "rank%d.json" % REC["rank"]), "w"))
