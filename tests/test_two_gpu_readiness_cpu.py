"""Two-GPU readiness without two GPUs (the reference shards groups of reads over `-n` workers, C3POa.py:236-256).

bench.py's worker path runs under `torch.distributed.run --nproc-per-node 2` exactly as the driver starts it -- HIP_VISIBLE_DEVICES
unset, RANK / LOCAL_RANK / WORLD_SIZE from the launcher -- with the HIP binding replaced by a recording stand-in
(tests/helpers/bench_two_rank_stub.py).  Checked: every rank opens its handle on device LOCAL_RANK, selects that device before the
process group exists, asks for the "nccl" (= RCCL) backend bound to that same device, the two ranks use DIFFERENT devices, process
different shards, and rank 0 prints ONE line carrying both ranks' step times."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(600)
def test_two_ranks_pick_distinct_devices_and_an_rccl_group_each(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "C3_BENCH_DEVICE_MAP",
                                                            "WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["C3_STUB_OUT"] = str(tmp_path)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "helpers", "bench_two_rank_stub.py"), "--gpus", "2", "--reads", "12", "--steps", "2", "--warmup", "1", "--no-cpu",
           "--other-configs", "none"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=560, cwd=ROOT)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    recs = [json.load(open(tmp_path / ("rank%d.json" % k))) for k in range(2)]
    for k, rec in enumerate(recs):
        assert rec["rank"] == k and rec["local_rank"] == k
        assert rec["HIP_VISIBLE_DEVICES"] is None                              # the worker does not rely on a per-rank device mask
        assert rec["handles"] == [k], rec                                      # c3_create(device = LOCAL_RANK), once
        assert rec["set_device"] and set(rec["set_device"]) == {k}, rec        # hipSetDevice(LOCAL_RANK) only
        assert rec["pg"] == {"backend": "nccl", "device_id": "cuda:%d" % k}, rec
        ev = rec["events"]
        assert ev.index("set_device") < ev.index("init_process_group") < ev.index("c3_create"), ev
    assert recs[0]["handles"] != recs[1]["handles"]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                                   # ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak"
    assert len(out["per_rank_ms_per_step"]) == 2 and all(t > 0 for t in out["per_rank_ms_per_step"])
    assert out["ms_per_step"] >= max(out["per_rank_ms_per_step"]) - 1e-6       # the line's step time is the MAX over ranks
    assert out["value"] == pytest.approx(2 * 12 * out["steps"] / (out["ms_per_step"] * out["steps"] * 1e-3), rel=1e-3)      # whole-job rate
    d = out["config"]["shard_digests"]
    assert len(d) == 2 and d[0] != d[1]                                        # two different shards
