"""The product's rank / shard / device plumbing executed for real on a one-GPU box (C3POa.py:236-256 shards
groups of reads over workers).  RCCL refuses two ranks on one device, so a test-only device map puts every rank / worker on
GPU 0 and the barrier + MAX of bench.py go over gloo; everything else is the path an 8-GPU node runs:
  * bench.py's worker path under torch.distributed.run with world_size 2: every rank generates, uploads and processes ITS
    shard on its handle; the union of the shard digests must equal single-process results of the same shards;
  * stream.run with -n 2: two worker threads, each with its own handle, streams, pinned buffers and byte-range readers."""
import hashlib
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from c3poa_amd import synth  # noqa: E402

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_bench_worker_path_two_ranks_on_one_gpu():
    from c3poa_amd import _lib
    n = 768
    env = dict(os.environ, C3_BENCH_DEVICE_MAP="0,0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--reads", str(n), "--steps", "2",
           "--warmup", "1", "--no-cpu", "--other-configs", "none"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-1000:]                                  # ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0
    assert out["value"] == pytest.approx(2 * n * out["steps"] / (out["ms_per_step"] * out["steps"] * 1e-3), rel=1e-3)   # whole-job rate
    got = out["config"]["shard_digests"]
    assert len(got) == 2 and got[0] != got[1]                                 # two different shards were processed
    # the same shards through one handle in this process
    exp = []
    for rank in range(2):
        recs = list(synth.generate("cfg2", n_reads=n, start=rank * n))
        h = _lib.Handle()
        h.set_splints([synth.SPLINT1])
        h.upload([x[1] for x in recs], [x[2] for x in recs], [x[3] for x in recs])
        h.run()
        res, cbuf, coff = h.results_raw()
        exp.append(hashlib.sha1(res["status"].tobytes() + res["cons_len"].tobytes() + cbuf.tobytes()[:int(coff[-1])]).hexdigest())
        assert (res["status"] == 0).sum() >= n - 2
        h.close()
    assert got == exp


def test_cli_two_workers_on_one_gpu(tmp_path):
    """`-n 2` with the device map "0,0": two worker threads (own handle, streams, pinned buffers, byte-range readers each)
    append to the same output files; the records equal the one-worker run"""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_gpu_configs import _cli, _sorted_records, _write_inputs
    n = 900
    recs = list(synth.generate("cfg5", n_reads=n))
    fq, fa = _write_inputs(tmp_path, recs, True)
    one = _cli(tmp_path, "one", fq, fa, recs, True, {"C3_GPU_BATCH_READS": "100000"})
    two = _cli(tmp_path, "two", fq, fa, recs, True, {"C3_GPU_BATCH_READS": "100", "C3_DEVICE_MAP": "0,0", "C3_MIN_RANGE_BYTES": "1"}, extra=("-n", "2"))
    assert _sorted_records(one) == _sorted_records(two)
    assert open(one + "/c3poa.log").read() == open(two + "/c3poa.log").read()
