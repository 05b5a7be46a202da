"""bench.py launcher logic (CPU): `--gpus N` without WORLD_SIZE starts N child ranks, and fails loudly without N GPUs."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_worker_command_is_a_child_torchrun():
    cmd = bench.worker_command(4, ["--gpus", "4", "--steps", "2"], port=29511)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    assert cmd[-5] == os.path.join(ROOT, "bench.py") and cmd[-4:] == ["--gpus", "4", "--steps", "2"]
    # a free port is picked when none is given
    assert int(bench.worker_command(2, [])[bench.worker_command(2, []).index("--master-port") + 1]) > 0


def test_launcher_refuses_more_gpus_than_visible():
    with pytest.raises(SystemExit) as e:
        bench.launch_workers(2, ["--gpus", "2"], have=1)
    assert "only 1 GPU" in str(e.value)


def test_per_config_defaults():
    assert bench.parse_args([]).reads == 100000 and bench.parse_args([]).cfg == "cfg2"
    assert bench.parse_args(["--cfg", "cfg3"]).reads == 125000
    assert bench.parse_args(["--cfg", "cfg4"]).reads == 100000
    assert bench.parse_args(["--cfg", "cfg4", "--reads", "5"]).reads == 5


@pytest.mark.timeout(300)
def test_gpus_2_on_a_box_without_two_gpus_fails_loudly():
    """the way the driver runs it: `python3 bench.py --gpus 2` (no WORLD_SIZE).  Here no GPU is visible at all."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs visible: covered by the driver's scaling run")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=280)
    assert p.returncode != 0
    assert "GPU(s) visible" in (p.stderr + p.stdout)


def test_world_size_must_match_gpus():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--reads", "8"],
                       capture_output=True, text=True, env=env, timeout=280)
    assert p.returncode != 0 and "WORLD_SIZE=2" in (p.stderr + p.stdout)


def test_pmc_traffic_is_tied_to_the_kernel_sources():
    """roofline.traffic comes from a committed rocprofv3 --pmc summary; bench.py only uses it when that summary was measured on the
    kernels that are running (hash of c3poa_amd/csrc/*.hip + *.h, the same in tools/collect_profiles.py)"""
    import glob
    import json
    import bench
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import collect_profiles
    sha = bench.kernel_src_sha()
    assert sha == collect_profiles.kernel_src_sha(ROOT) and len(sha) == 16
    newest = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic_cfg2_100k.json")))[-1]
    pm = json.load(open(newest))
    # the newest committed summary either matches the sources (traffic reported) or is declared stale by bench.py (traffic null)
    assert "kernels" in pm and ("kernel_src_sha" not in pm or isinstance(pm["kernel_src_sha"], str))
    assert "k_poa" in pm["kernels"] or any(k.startswith("k_poa") for k in pm["kernels"])


def test_traffic_and_valu_lookups_take_only_this_builds_counters(tmp_path):
    """both branches of the lookup behind roofline.traffic / roofline.valu: a committed summary measured on THESE kernel sources is
    reported, one measured on other sources (or without a hash) is named but never reported as this build's, none at all is None"""
    import json
    sha = "0123456789abcdef"
    pm = {"kernel_src_sha": sha, "kernels": {"k_poa": {"hbm_bytes_per_launch": 123.0}, "k_window": {"hbm_bytes_per_launch": 7.0}}}
    json.dump(pm, open(tmp_path / "r05_pmc_traffic_cfg2_100k.json", "w"))
    t, src, tot = bench.pmc_traffic_lookup(str(tmp_path), "cfg2", 100000, "ms_poa", sha)
    assert t == 123.0 and "r05_pmc_traffic_cfg2_100k.json" in src and "this build" in src
    assert tot == 130.0                                                        # roofline.traffic_total: every kernel of the step (round 6)
    t, src, tot = bench.pmc_traffic_lookup(str(tmp_path), "cfg2", 100000, "ms_poa", "f" * 16)
    assert t is None and tot is None and src.startswith("none for this build") and sha in src
    pm.pop("kernel_src_sha")                                                   # a summary from before the hash existed
    json.dump(pm, open(tmp_path / "r05_pmc_traffic_cfg2_100k.json", "w"))
    t, src, tot = bench.pmc_traffic_lookup(str(tmp_path), "cfg2", 100000, "ms_poa", sha)
    assert t is None and src.startswith("none for this build")
    assert bench.pmc_traffic_lookup(str(tmp_path), "cfg4", 100000, "ms_poa", sha) == (None, None, None)      # no file for the workload
    # the newest tag wins: an r04 file of this build is not consulted when an r05 file (stale) exists
    json.dump({"kernel_src_sha": sha, "kernels": {"k_poa": {"hbm_bytes_per_launch": 5.0}}}, open(tmp_path / "r04_pmc_traffic_cfg2_100k.json", "w"))
    assert bench.pmc_traffic_lookup(str(tmp_path), "cfg2", 100000, "ms_poa", sha)[0] is None
    sq = {"kernel_src_sha": sha, "cycles_per_inst_assumed": 2.9, "reads": 32768, "kernels": {"k_poa": {"insts_per_cell": 3.5, "busy_frac": 0.65, "insts_per_simd_cycle": 0.22}}}
    json.dump(sq, open(tmp_path / "r05_sq_counters_cfg2.json", "w"))
    v = bench.valu_lookup(str(tmp_path), "cfg2", "ms_poa", sha)
    assert v["insts_per_cell"] == 3.5 and v["busy_frac"] == 0.65 and "this build" in v["source"]
    assert v["cycles_per_inst"] == 2.9                                         # a summary without the per-kernel figure: the file's constant
    sq["kernels"]["k_poa"]["cycles_per_inst"] = 3.31                           # round 6: priced from the kernel's own listing (tools/isa_cpi.py)
    json.dump(sq, open(tmp_path / "r06_sq_counters_cfg2.json", "w"))
    assert bench.valu_lookup(str(tmp_path), "cfg2", "ms_poa", sha)["cycles_per_inst"] == 3.31
    os.remove(tmp_path / "r06_sq_counters_cfg2.json")
    v = bench.valu_lookup(str(tmp_path), "cfg2", "ms_poa", "e" * 16)
    assert v["insts_per_cell"] is None and v["busy_frac"] is None and v["source"].startswith("none for this build")
    assert bench.valu_lookup(str(tmp_path), "cfg3", "ms_poa", sha) == {"insts_per_cell": None, "busy_frac": None, "source": None}
