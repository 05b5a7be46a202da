#!/usr/bin/env python3
"""bench.py -- R2C2 reads->consensus/sec on MI355X (BASELINE.json metric, config cfg2).

A "step" is one pass of the whole hot path (conk -> peaks/split -> POA draft -> polish) over one
batch of synthetic reads that is ALREADY RESIDENT in HBM (packed 2-bit bases + quality bytes);
the consensus sequences are left in HBM.  One process per GPU; reads are sharded, there is no
data-path collective -> weak scaling: every rank processes --reads reads/step.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0.  roofline = algorithmic HBM bytes of the dominant kernel per launch
/ its HIP-event duration (events recorded by the library on its own stream);
cpu_baseline = the oracle (own CPU restatement, "port") on a bounded sample, rank 0, N=1 only.
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec (/opt/skills/guides/MI355X_MICROARCH.md)


def _gen_shard(args):
    cfg, start, count = args
    from c3poa_amd import synth
    return [(r[1], r[2], r[3], r[4]) for r in synth.generate(cfg, n_reads=count, start=start)]


def make_reads(cfg, n, start, procs):
    """n synthetic reads of config `cfg` starting at stream index `start`."""
    procs = max(1, min(procs, 64, n // 64 + 1))
    if procs == 1:
        return _gen_shard((cfg, start, n))
    per = (n + procs - 1) // procs
    jobs = [(cfg, start + i * per, min(per, n - i * per)) for i in range(procs) if n - i * per > 0]
    with mp.get_context("fork").Pool(len(jobs)) as pool:
        parts = pool.map(_gen_shard, jobs)
    return [r for p in parts for r in p]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=100000, help="reads per step per GPU (cfg2: 100k)")
    ap.add_argument("--cfg", default="cfg2")
    ap.add_argument("--unique", type=int, default=0, help="distinct synthetic reads generated per rank (0 = all)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    # generate inputs BEFORE anything touches the GPU (fork-based pool)
    from c3poa_amd import synth
    n_unique = a.unique if a.unique > 0 else a.reads
    n_unique = min(n_unique, a.reads)
    t_gen = time.time()
    recs = make_reads(a.cfg, n_unique, rank * a.reads, max(1, effective_cores() // max(world, 1)))
    t_gen = time.time() - t_gen

    import torch
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the c3poa HIP backend has no CPU fallback")
    torch.cuda.set_device(local_rank)
    from c3poa_amd import _lib
    md = synth.CONFIGS[a.cfg]["mdist"]
    h = _lib.Handle(device=local_rank, mdistcutoff=md)
    h.set_splints([synth.SPLINT1])

    # tile the distinct reads up to --reads (identical work per copy; stated in `data`)
    reps = (a.reads + n_unique - 1) // n_unique
    seqs = [r[0] for r in recs] * reps
    quals = [r[1] for r in recs] * reps
    strands = [r[2] for r in recs] * reps
    seqs, quals, strands = seqs[:a.reads], quals[:a.reads], strands[:a.reads]
    lens = np.array([len(s) for s in seqs], dtype=np.int64)
    off = np.zeros(len(lens) + 1, dtype=np.int64)
    np.cumsum(lens, out=off[1:])
    seq_cat, qual_cat = "".join(seqs).encode(), "".join(quals).encode()   # the boundary hands over flat host buffers
    del seqs, quals
    h.upload_flat(seq_cat, qual_cat, off, "".join(strands))               # first call also allocates the device buffers
    t_up = time.perf_counter()
    h.upload_flat(seq_cat, qual_cat, off, "".join(strands))               # H2D + 2-bit pack: outside the timed region
    t_up = time.perf_counter() - t_up
    del seq_cat, qual_cat

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        h.run()
    stage_ms = {}
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        h.run()                                  # synchronous on return (library stream drained)
        tm = h.timing()
        for k, v in tm.items():
            if k.startswith("ms_") and k not in ("ms_pack", "ms_total"):
                stage_ms.setdefault(k, []).append(v)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    res, cons = h.results(with_consensus=(rank == 0))
    tm = h.timing()
    ok = res["status"] == 0

    out = None
    if rank == 0:
        total_reads = a.reads * a.steps * world
        value = total_reads / dt
        # dominant kernel + roofline (algorithmic bytes: SURVEY.md 8(d))
        avg = {k: float(np.mean(v)) for k, v in stage_ms.items()}
        dom = max(avg, key=avg.get)
        alg_bytes = float(np.sum((lens + 3) // 4 + lens) + np.sum(res["cons_len"][ok]) + 4 * np.sum(res["n_peaks"]))
        achieved = alg_bytes / (avg[dom] * 1e-3) / 1e9
        cells = tm["cells_conk"] + tm["cells_poa"] + tm["cells_polish"]
        # consensus % identity vs the synthetic truth (second half of BASELINE.json's metric), on a sample
        n_id = min(200, n_unique)
        idents = np.array([synth.identity(cons[i], recs[i][3]) if cons[i] else 0.0 for i in range(n_id)])
        # HBM traffic of the dominant kernel: PMC counters cannot be collected from inside this process;
        # the number comes from the committed rocprofv3 pass of THIS command when the workload matches
        traffic = None
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic_cfg2_100k.json")))
            if a.cfg == "cfg2" and a.reads == 100000:
                traffic = pm["kernels"][dom.replace("ms_", "k_")]["hbm_bytes_per_launch"]
        except Exception:
            traffic = None
        out = {
            "metric": "R2C2 reads->consensus/sec", "value": round(value, 1), "unit": "reads/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32",
            "data": "synthetic (c3poa_amd.synth %s: %d distinct reads%s)" % (
                a.cfg, n_unique, "" if reps == 1 else ", tiled x%d" % reps),
            "config": {"workload": "%s: %d reads/GPU/step, 5 kb, 3x1.5 kb repeats, Splint1, 10%% error" % (a.cfg, a.reads)
                       if a.cfg == "cfg2" else "%s: %d reads/GPU/step" % (a.cfg, a.reads),
                       "stages": "conk+peaks/split+POA+polish", "reads_per_gpu_step": a.reads,
                       "consensus_ok": int(ok.sum()), "mean_read_len": float(lens.mean()),
                       "identity_vs_truth": {"mean": round(float(idents.mean()), 5), "median": round(float(np.median(idents)), 5), "reads": int(n_id)},
                       "upload_ms": round(t_up * 1e3, 1),
                       "pcie_inclusive_reads_per_s": round(a.reads * world / (dt / a.steps + t_up), 1)},
            "roofline": {"bound": "hbm", "kernel": dom.replace("ms_", "k_"), "achieved": round(achieved, 3),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6),
                         "traffic": traffic, "traffic_source": "profiles/r01_pmc_traffic_cfg2_100k.json (rocprofv3 --pmc, (2*FETCH_SIZE+WRITE_SIZE)*1024)" if traffic else None,
                         "alg_bytes_per_launch": alg_bytes,
                         "kernel_ms": {k: round(v, 3) for k, v in avg.items()},
                         "gcups": round(cells / (sum(avg.values()) * 1e-3) / 1e9, 2)},
            "gen_s": round(t_gen, 1),
        }
    h.close()
    if rank == 0 and world == 1 and not a.no_cpu:
        out["cpu_baseline"] = cpu_baseline(recs, md, a.cpu_seconds)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


def effective_cores():
    """usable host cores: min(cpu_count, affinity mask, cgroup CPU quota)"""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return n


def cpu_baseline(recs, mdist, seconds):
    """the oracle (own CPU restatement) on all host cores, bounded sample of the same workload"""
    from c3poa_amd import synth
    from oracle import oracle_py as O
    cores = effective_cores()
    P = O.default_params(mdistcutoff=mdist)
    probe = recs[:min(len(recs), 4 * cores)]
    t = time.perf_counter()
    O.process_batch(synth.SPLINT1, [(r[0], r[1]) for r in probe], [r[2] for r in probe], params=P, threads=cores)
    rate = len(probe) / (time.perf_counter() - t)
    n = int(min(len(recs), max(len(probe), rate * seconds)))
    sample = recs[:n]
    t = time.perf_counter()
    res, cons = O.process_batch(synth.SPLINT1, [(r[0], r[1]) for r in sample], [r[2] for r in sample], params=P, threads=cores)
    dt = time.perf_counter() - t
    # per-core rate ("-n 1"): one thread on a small slice of the same sample
    one = sample[:min(len(sample), max(16, int(3.0 * rate / max(cores, 1))))]
    t1 = time.perf_counter()
    O.process_batch(synth.SPLINT1, [(r[0], r[1]) for r in one], [r[2] for r in one], params=P, threads=1)
    per_core = len(one) / (time.perf_counter() - t1)
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except Exception:
        pass
    return {"value": round(n / dt, 1), "unit": "reads/s", "cores": cores, "kind": "port",
            "per_core_reads_per_s": round(per_core, 1), "cpu_model": model, "host_cpus": os.cpu_count(),
            "sample": "first %d reads of the same synthetic batch, oracle/libc3oracle.so (own CPU restatement, "
                      "-O3, OpenMP %d threads = usable host cores of %d), %.1f s" % (n, cores, os.cpu_count(), dt)}


if __name__ == "__main__":
    main()
