#!/usr/bin/env python3
"""bench.py -- R2C2 reads->consensus/sec on MI355X (BASELINE.json metric; default workload cfg2).

A "step" is one pass of the whole hot path (conk -> peaks/split -> POA draft -> polish) over one batch of synthetic
reads, run as the steady state of the streaming pipeline the CLI uses ("first batch submit -> last
result fetched"):

    c3_batch_stage(next batch)      H2D + 2-bit pack on the library's second stream, overlapped with ...
    c3_batch_run(resident batch)    ... the kernels of the resident batch
    c3_batch_results(...)           per-read records + consensus bytes fetched to the host
    c3_batch_commit()               the staged batch becomes resident

so the input of every timed step is already resident in HBM when the step starts (it was staged during the step before),
and the timed region still pays for the copy engine traffic, the result fetch and every host gap.  `value` is that rate;
`config.resident_only_reads_per_s` is the kernels-only rate of the same steps (sum of c3_batch_run times).

One process per GPU; reads are sharded by rank, there is no data-path collective -> weak scaling: every
rank processes --reads reads per step.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python bench.py --gpus N ...            # WORLD_SIZE unset: starts N workers itself (child torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0.  roofline = algorithmic HBM bytes of the dominant kernel per launch / its HIP-event
duration (events recorded by the library on its own stream); cpu_baseline = the oracle (own CPU restatement, "port") on
a bounded sample, rank 0, N=1 only.
"""
import argparse
import json
import multiprocessing as mp
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec (/opt/skills/guides/MI355X_MICROARCH.md)
# per-GPU reads per step: BASELINE.json configs (cfg3: 1M reads over 8 GPUs)
DEFAULT_READS = {"cfg1": 1000, "cfg2": 100000, "cfg3": 125000, "cfg4": 100000, "cfg5": 131072, "cfgL": 50000, "cfg2e15": 100000, "cfg2e20": 100000}
WORKLOAD_TEXT = {
    "cfg1": "5 kb, 3x1.5 kb repeats (plumbing case)",
    "cfg2": "5 kb, 3x1.5 kb repeats, Splint1, 10% error",
    "cfg3": "mixed 2-10 subreads, 1 kb insert (1M reads read-sharded over 8 GPUs -> 125k per GPU)",
    "cfg4": "20 kb, 12 subreads, -d 1500 (wide adaptive band)",
    "cfg5": "cfg2 shape, one GPU batch of the streamed CLI",
    "cfgL": "not a BASELINE config: long inserts (3 / 6 kb, 3-5 repeats, reads of 10-32 kb) -- the subread-length axis",
    "cfg2e15": "not a BASELINE config: the cfg2 shape at 15 % errors (every rate x 1.5) -- the error-rate axis, where raw ONT R2C2 reads live",
    "cfg2e20": "not a BASELINE config: the cfg2 shape at 20 % errors (every rate x 2.0)",
}


def kernel_src_sha():
    """content hash of the kernel / library sources (the GPU box has no .git): ties a committed PMC summary to the kernels
    it was measured on -- tools/collect_profiles.py stores the same hash"""
    import glob
    import hashlib
    hsh = hashlib.sha1()
    for f in sorted(glob.glob(os.path.join(ROOT, "c3poa_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "c3poa_amd", "csrc", "*.h"))):
        hsh.update(os.path.basename(f).encode() + b"\0" + open(f, "rb").read())
    return hsh.hexdigest()[:16]


PROFILE_TAGS = ("r06", "r05", "r04", "r03", "r02", "r01")
SIMDS = 1024              # 256 CUs x 4 SIMDs


def pmc_traffic_lookup(profiles_dir, cfg, reads, dom, sha):
    """HBM bytes per launch of kernel `dom` from the newest committed rocprofv3 PMC summary of this workload -- ONLY when that summary
    was measured on the kernel sources running here (same hash); otherwise traffic is None and the source says what exists.
    PMC counters cannot be collected from inside this process (tools/profile_round.sh + tools/collect_profiles.py make the file)."""
    for tag in PROFILE_TAGS:
        name = "%s_pmc_traffic_%s_%dk.json" % (tag, cfg, reads // 1000)
        try:
            pm = json.load(open(os.path.join(profiles_dir, name)))
            t_ = pm["kernels"][dom.replace("ms_", "k_")]["hbm_bytes_per_launch"]
        except Exception:
            continue
        if pm.get("kernel_src_sha") == sha:
            # every kernel of the step (one launch each per step, but for the two launches of k_window that the file already sums)
            tot = float(sum(v.get("hbm_bytes_per_launch", 0.0) for v in pm["kernels"].values()))
            return t_, "profiles/%s (rocprofv3 --pmc, (2*FETCH_SIZE+WRITE_SIZE)*1024; kernel sources %s = this build)" % (name, sha), tot
        # measured on other kernels than the ones running here: not this run's traffic
        return None, "none for this build (kernel sources %s; profiles/%s was measured on %s: %.4g bytes per launch)" % (
            sha, name, pm.get("kernel_src_sha", "an earlier round's kernels"), t_), None
    return None, None, None


def valu_lookup(profiles_dir, cfg, dom, sha):
    """what actually binds the DP kernels -- vector instruction issue -- from the newest committed SQ-counter summary of this config
    (tools/pmc_sq.sh -> profiles/rNN_sq_counters_<cfg>.json), under the same kernel-source-hash rule as the traffic"""
    for tag in PROFILE_TAGS:
        name = "%s_sq_counters_%s.json" % (tag, cfg)
        try:
            sq = json.load(open(os.path.join(profiles_dir, name)))
            k = sq["kernels"][dom.replace("ms_", "k_")]
        except Exception:
            continue
        if sq.get("kernel_src_sha") != sha:
            return {"insts_per_cell": None, "busy_frac": None,
                    "source": "none for this build (kernel sources %s; profiles/%s was measured on %s: %.3f vector wave-instructions per cell)" % (
                        sha, name, sq.get("kernel_src_sha", "an earlier round's kernels"), k.get("insts_per_cell") or 0.0)}
        cpi = k.get("cycles_per_inst") or sq.get("cycles_per_inst_assumed") or 0.0
        return {"insts_per_cell": k.get("insts_per_cell"), "busy_frac": k.get("busy_frac"),
                "insts_per_simd_cycle": k.get("insts_per_simd_cycle"), "cycles_per_inst": cpi,
                "source": "profiles/%s (rocprofv3 --pmc SQ_INSTS_VALU / SQ_BUSY_CYCLES on tools/phase_prof.py %s %s; busy_frac = vector wave-instructions x "
                          "%.2f issue cycles (this kernel's own instruction mix priced with the two classes of tools/ubench/valu_cost.hip: tools/isa_cpi.py) "
                          "/ SIMD cycles of the kernel; kernel sources %s = this build)" % (name, sq.get("reads"), cfg, cpi, sha)}
    return {"insts_per_cell": None, "busy_frac": None, "source": None}


def parity_sample(recs, n_unique, res, cbuf, coff, mdist, k=1024, seed=1234):
    """CHECKER (outside every timed region): the oracle on k random reads of the batch the last step processed; counts reads
    whose status or consensus bytes differ from what the GPU delivered for them"""
    from c3poa_amd import synth
    from oracle import oracle_py as O
    n = len(res)
    idx = sorted(np.random.default_rng(seed).choice(n, size=min(k, n), replace=False).tolist())
    rs = [recs[i % n_unique] for i in idx]
    ores, ocons = O.process_batch(synth.SPLINT1, [(r[0], r[1]) for r in rs], [r[2] for r in rs],
                                  params=O.default_params(mdistcutoff=mdist), threads=effective_cores())
    raw = cbuf.tobytes()
    bad = 0
    for i, orr, oc in zip(idx, ores, ocons):
        got = raw[coff[i]:coff[i + 1]].decode() if res["status"][i] == 0 else ""
        bad += int(res["status"][i] != orr.status or got != oc)
    return {"reads": len(idx), "mismatches": bad, "checker": "oracle/libc3oracle.so (status + consensus bytes), after the timed region"}


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


def visible_gpus():
    """GPUs visible to this process WITHOUT initialising the HIP runtime (the launcher must stay GPU-free: its children
    are separate processes, and a process that has touched the GPU must never exec).  Counted from the KFD topology (nodes
    with SIMDs), honouring HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES; torch.cuda.device_count() can fall back to
    hipGetDeviceCount, which does initialise the runtime."""
    n = 0
    try:
        base = "/sys/class/kfd/kfd/topology/nodes"
        for d in os.listdir(base):
            try:
                props = dict(l.split(None, 1) for l in open(os.path.join(base, d, "properties")).read().splitlines() if " " in l)
                n += int(props.get("simd_count", "0")) > 0
            except OSError:
                continue
    except OSError:
        n = 0
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def launch_workers(n_gpus, argv, have=None):
    """`python bench.py --gpus N` with WORLD_SIZE unset: start N ranks as CHILD processes (torch.distributed.run), forward
    their output and return their exit code.  Fails loudly when fewer than N GPUs are visible."""
    have = visible_gpus() if have is None else have
    if have < n_gpus:
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible on this node" % (n_gpus, have))
    cmd = worker_command(n_gpus, argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def worker_command(n_gpus, argv, port=None):
    if port is None:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reads", type=int, default=0, help="reads per step per GPU (0 = the config's size: cfg2 100k)")
    ap.add_argument("--cfg", default="cfg2")
    ap.add_argument("--unique", type=int, default=0, help="distinct synthetic reads generated per rank (0 = all)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--parity-reads", type=int, default=1024, help="random reads of the last batch (of every config run) checked against the oracle after the timed region")
    ap.add_argument("--other-configs", default="auto",
                    help="comma list of further configs run for 3 steps after the headline (rank 0, N=1): 'auto' = cfg3,cfg4,cfgL,cfg2e15 "
                         "when the headline is cfg2 at full size, 'none' = skip")
    ap.add_argument("--other-unique", type=int, default=0, help="distinct reads generated for each of the other configs (0 = all distinct, no tiling)")
    a = ap.parse_args(argv)
    if a.reads <= 0:
        a.reads = DEFAULT_READS.get(a.cfg, 100000)
    return a


def main():
    a = parse_args()
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        sys.exit(launch_workers(a.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (a.gpus, world))
    dist = None
    # generate inputs BEFORE anything touches the GPU (fork-based pool)
    from c3poa_amd import synth
    n_unique = a.unique if a.unique > 0 else a.reads
    n_unique = min(n_unique, a.reads)
    t_gen = time.time()
    recs = make_reads(a.cfg, n_unique, rank * a.reads, max(1, effective_cores() // max(world, 1)))
    t_gen = time.time() - t_gen
    # inputs of the other configs too (rank 0, N=1): the fork pool must run before anything touches the GPU
    others = a.other_configs
    if others == "auto":
        others = "cfg3,cfg4,cfgL,cfg2e15" if (a.cfg == "cfg2" and a.reads >= 100000) else "none"
    other_recs = {}
    if rank == 0 and world == 1 and others != "none":
        for c in [x for x in others.split(",") if x]:
            t_ = time.time()
            other_recs[c] = (make_reads(c, DEFAULT_READS[c] if a.other_unique <= 0 else min(a.other_unique, DEFAULT_READS[c]), 0, effective_cores()), time.time() - t_)

    import torch
    # test hook (tests/test_gpu_two_ranks.py): C3_BENCH_DEVICE_MAP="0,0" puts every rank on GPU 0 and the barrier / MAX go over
    # gloo (RCCL refuses two ranks on one device) -- the rank / shard / device plumbing of the real worker path on a one-GPU box
    dmap = os.environ.get("C3_BENCH_DEVICE_MAP")
    device = int(dmap.split(",")[local_rank % len(dmap.split(","))]) if dmap else local_rank
    backend = "gloo" if dmap else "nccl"
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(device)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group("gloo")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the c3poa HIP backend has no CPU fallback")
    torch.cuda.set_device(device)
    local_rank = device
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
    # the boundary hands over flat host buffers; page-locked, as the native reader's are (c3_reader_*)
    host = _lib.PinnedBatch("".join(seqs).encode(), "".join(quals).encode(), off, "".join(strands))
    del seqs, quals
    h.upload_pinned(host)                                                  # first call also allocates the device buffers
    t_up = time.perf_counter()
    h.upload_pinned(host)                                                  # un-overlapped H2D + 2-bit pack, for reference
    t_up = time.perf_counter() - t_up

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    stage_ms, run_s, fetch_s = {}, [], []

    KERNEL_MS = ("ms_conk", "ms_peaks", "ms_poa", "ms_prep", "ms_window", "ms_stitch")     # hipEvent times of the kernels
    host_ms = {}                                                                             # host wall figures of c3_batch_run

    # results: c3_batch_results_snapshot freezes them on the device right after the run; a helper thread copies the snapshot to the
    # host (c3_batch_results_fetch) while this thread commits, stages and runs the NEXT batch -- the same pattern as the CLI's
    # fetch thread (c3poa_amd/stream.py).  Every timed step still delivers one batch of results: the copy of step k lands during
    # step k+1 and the last one is waited for inside the timed region.
    fetcher = _lib.ResultFetcher(h, pinned=bool(int(os.environ.get('C3_BENCH_PINNED_RESULTS', '0'))))

    def step(timed, k):
        h.stage_pinned(host)                     # next batch: copy engine + pack kernel on the second stream
        t1 = time.perf_counter()
        h.run()                                  # resident batch: conk -> peaks -> POA -> polish
        t2 = time.perf_counter()
        out = fetcher.after_run()                # snapshot of this batch; returns the PREVIOUS batch's records + consensus bytes
        t3 = time.perf_counter()
        h.commit()
        if timed:
            run_s.append(t2 - t1); fetch_s.append(t3 - t2)
            for k_, v in h.last_timing.items():
                if k_ in KERNEL_MS:
                    stage_ms.setdefault(k_, []).append(v)
                elif k_ in ("ms_wall", "ms_host_worklist", "ms_alloc", "ms_host_gap"):
                    host_ms.setdefault(k_, []).append(v)
        return out

    for k in range(a.warmup):
        step(False, k)
    barrier()
    t0 = time.perf_counter()
    for k in range(a.steps):
        h.last_timing = None
        step(True, a.warmup + k)
    t_last = time.perf_counter()
    res_raw = fetcher.drain()                    # the last batch's results
    fetch_s.append(time.perf_counter() - t_last)
    barrier()
    dt = time.perf_counter() - t0
    fetcher.close()
    per_rank_ms = [round(dt / a.steps * 1e3, 3)]
    if dist is not None:
        # every rank's own step time (the line's ms_per_step is their MAX): an imbalance between GPUs shows in the driver's SCALE line
        per_rank_ms = [None] * world
        dist.all_gather_object(per_rank_ms, round(dt / a.steps * 1e3, 3))
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    tm = h.last_timing
    res, cbuf, coff = res_raw
    ok = res["status"] == 0
    # digest of this rank's shard (status + consensus bytes of the last step): lets a test compare the union of the shards of an
    # N-rank run with single-process results
    import hashlib
    digest = hashlib.sha1(res["status"].tobytes() + res["cons_len"].tobytes() + cbuf.tobytes()[:int(coff[-1])]).hexdigest()
    digests = [digest]
    if dist is not None:
        digests = [None] * world
        dist.all_gather_object(digests, digest)

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
        raw = cbuf.tobytes()
        idents = np.array([synth.identity(raw[coff[i]:coff[i + 1]].decode(), recs[i][3]) if coff[i + 1] > coff[i] else 0.0
                           for i in range(n_id)])
        # HBM traffic of the dominant kernel: PMC counters cannot be collected from inside this process; the number
        # comes from the committed rocprofv3 passes of THIS command when the workload matches
        sha = kernel_src_sha()
        traffic, tsrc, traffic_total = pmc_traffic_lookup(os.path.join(ROOT, "profiles"), a.cfg, a.reads, dom, sha)
        valu = valu_lookup(os.path.join(ROOT, "profiles"), a.cfg, dom, sha)
        out = {
            "metric": "R2C2 reads->consensus/sec", "value": round(value, 1), "unit": "reads/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3),
            "per_rank_ms_per_step": per_rank_ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32",
            "data": "synthetic (c3poa_amd.synth %s: %d distinct reads%s)" % (
                a.cfg, n_unique, "" if reps == 1 else ", tiled x%d" % reps),
            "config": {"workload": "%s: %d reads/GPU/step, %s" % (a.cfg, a.reads, WORKLOAD_TEXT.get(a.cfg, "")),
                       "stages": "stage(H2D+pack, overlapped) | conk+peaks/split+POA+polish | results snapshot (D2H by a helper thread beside the next run) | commit",
                       "reads_per_gpu_step": a.reads, "shard_digests": digests,
                       "consensus_ok": int(ok.sum()), "mean_read_len": float(lens.mean()),
                       "identity_vs_truth": {"mean": round(float(idents.mean()), 5), "median": round(float(np.median(idents)), 5), "reads": int(n_id)},
                       "resident_only_reads_per_s": round(a.reads * world / float(np.mean(run_s)), 1),
                       "run_ms": round(float(np.mean(run_s)) * 1e3, 2), "fetch_ms_exposed": round(float(np.sum(fetch_s)) / a.steps * 1e3, 2),
                       "upload_ms_unoverlapped": round(t_up * 1e3, 1)},
            "roofline": {"bound": "hbm", "kernel": dom.replace("ms_", "k_"), "achieved": round(achieved, 3),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6),
                         "traffic": traffic, "traffic_source": tsrc,
                         # all kernels of one step together (PMC), and how many times the algorithmic bytes that is: the state of the DP never has to
                         # leave a CU, so everything above ~1x is scratch the kernels chose to keep in memory (direction cells, graph arrays)
                         "traffic_total": traffic_total,
                         "traffic_over_algorithmic": (round(traffic_total / alg_bytes, 1) if traffic_total else None),
                         "alg_bytes_per_launch": alg_bytes,
                         "kernel_ms": {k: round(v, 3) for k, v in avg.items()},
                         "run_host_ms": {k: round(float(np.mean(v)), 3) for k, v in host_ms.items()},
                         "cells_per_step": int(cells),
                         "cells_polish_full_matrix": int(tm["cells_polish"]), "cells_polish_computed": int(tm["cells_polish_computed"]),
                         "band_layers": int(tm["n_band_layers"]), "band_fallback_layers": int(tm["n_band_fallback"]),
                         "windows": int(tm["n_windows"]), "windows_second_launch": int(tm["n_win_redo"]),
                         "cells_poa": int(tm["cells_poa"]), "poa_gcells_per_s": round(tm["cells_poa"] / ((avg["ms_poa"] + float(tm.get("ms_poa_tail", 0.0))) * 1e-3) / 1e9, 1),
                         "poa_second_pass_reads": int(tm["n_poa_redo"]), "poa_reads_beyond_16bit": int(tm["n_poa_redo16"]),
                         # cell updates per second of kernel time.  gcups_computed counts the cells the kernels COMPUTE (conk + POA + the banded
                         # polish rows); gcups_full_matrices counts what the oracle's full polish matrices hold (the figure printed as "gcups"
                         # until round 4) -- the band's certificate makes the two results identical, not the two amounts of work
                         "gcups_computed": round((tm["cells_conk"] + tm["cells_poa"] + tm["cells_polish_computed"]) / ((sum(avg.values()) + float(tm.get("ms_poa_tail", 0.0))) * 1e-3) / 1e9, 2),
                         "gcups_full_matrices": round(cells / (sum(avg.values()) * 1e-3) / 1e9, 2),
                         # what binds the dominant kernel is vector instruction issue, not HBM: instructions per counted cell and the
                         # share of the SIMDs' issue cycles they fill, from the committed SQ-counter pass of this build (None otherwise)
                         "valu": valu},
            "gen_s": round(t_gen, 1),
        }
    h.close()
    host.close()
    if rank == 0 and not a.no_cpu:
        out["parity_sample"] = parity_sample(recs, n_unique, res, cbuf, coff, md, k=a.parity_reads)
    if rank == 0 and world == 1 and other_recs:
        out["other_configs"] = {c: run_other_config(c, local_rank, *other_recs[c], check=not a.no_cpu, parity_reads=a.parity_reads) for c in other_recs}
    if rank == 0 and world == 1 and not a.no_cpu:
        out["cpu_baseline"] = cpu_baseline(recs, md, a.cpu_seconds)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


def run_other_config(cfg, device, recs, gen_s, steps=3, check=True, parity_reads=1024):
    """the same pipelined step (stage || run -> results snapshot -> commit, fetch beside the next run) on another BASELINE config at its per-GPU size, every
    read distinct (unless --other-unique tiles them): driver-clocked rates for cfg3 / cfg4 next to the cfg2 headline, each with an oracle check of --parity-reads (1 024)
    random reads of the batch after its timed region (full parity of these shapes: tests/test_gpu_configs.py)"""
    import torch
    from c3poa_amd import _lib, synth
    n = DEFAULT_READS[cfg]
    nu = len(recs)
    reps = (n + nu - 1) // nu
    seqs = ([r[0] for r in recs] * reps)[:n]; quals = ([r[1] for r in recs] * reps)[:n]; strands = ([r[2] for r in recs] * reps)[:n]
    lens = np.array([len(s) for s in seqs], dtype=np.int64)
    off = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(lens, out=off[1:])
    host = _lib.PinnedBatch("".join(seqs).encode(), "".join(quals).encode(), off, "".join(strands))
    del seqs, quals
    h = _lib.Handle(device=device, mdistcutoff=synth.CONFIGS[cfg]["mdist"])
    h.set_splints([synth.SPLINT1])
    h.upload_pinned(host)
    kms = {}

    fetcher = _lib.ResultFetcher(h, pinned=bool(int(os.environ.get('C3_BENCH_PINNED_RESULTS', '0'))))

    def step(timed):
        h.stage_pinned(host); h.run(); fetcher.after_run(); h.commit()
        if timed:
            for k, v in h.last_timing.items():
                if k.startswith("ms_") and k not in ("ms_pack", "ms_total", "ms_wall", "ms_alloc", "ms_host_worklist", "ms_host_gap", "ms_poa_tail"):
                    kms.setdefault(k, []).append(v)
    step(False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step(True)
    res_raw = fetcher.drain()                    # (the copy of every step but the last ran beside the next step's kernels)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    fetcher.close()
    tm = h.last_timing
    res, cbuf, coff = res_raw
    raw = cbuf.tobytes()
    n_id = min(100, nu)
    idents = [synth.identity(raw[coff[i]:coff[i + 1]].decode(), recs[i][3]) if coff[i + 1] > coff[i] else 0.0 for i in range(n_id)]
    o = {"value": round(n * steps / dt, 1), "unit": "reads/s", "reads_per_step": n, "steps": steps, "ms_per_step": round(dt / steps * 1e3, 2),
         "kernel_ms": {k: round(float(np.mean(v)), 2) for k, v in kms.items()},
         "cells": int(tm["cells_conk"] + tm["cells_poa"] + tm["cells_polish"]), "cells_polish_computed": int(tm["cells_polish_computed"]),
         # (the last POA pass runs beside the polish on a stream of its own: its cells are in cells_poa, so its time joins the denominator -- kernel
         # time, not wall time: an overlapped kernel counts twice, which under-states the rate rather than inflating it)
         "gcups_computed": round((tm["cells_conk"] + tm["cells_poa"] + tm["cells_polish_computed"]) / ((sum(float(np.mean(v)) for v in kms.values()) + float(tm.get("ms_poa_tail", 0.0))) * 1e-3) / 1e9, 2),
         "band_fallback_layers": int(tm["n_band_fallback"]), "band_layers": int(tm["n_band_layers"]), "windows": int(tm["n_windows"]), "windows_second_launch": int(tm["n_win_redo"]),
         "cells_poa": int(tm["cells_poa"]), "poa_gcells_per_s": round(tm["cells_poa"] / ((float(np.mean(kms["ms_poa"])) + float(tm.get("ms_poa_tail", 0.0))) * 1e-3) / 1e9, 1),
         "poa_second_pass_reads": int(tm["n_poa_redo"]), "poa_reads_beyond_16bit": int(tm["n_poa_redo16"]),
         "ms_poa_last_pass_beside_polish": round(float(tm.get("ms_poa_tail", 0.0)), 2),       # (its own stream, overlapped: not in kernel_ms)
         "consensus_ok": int((res["status"] == 0).sum()), "identity_vs_truth_mean": round(float(np.mean(idents)), 5),
         "data": "synthetic %s, %d distinct reads%s" % (cfg, nu, "" if reps == 1 else " tiled x%d" % reps), "gen_s": round(gen_s, 1)}
    h.close(); host.close()
    if check:
        o["parity_sample"] = parity_sample(recs, nu, res, cbuf, coff, synth.CONFIGS[cfg]["mdist"], k=parity_reads)
    return o


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
