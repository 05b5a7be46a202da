#!/usr/bin/env python3
"""Writes the section of profiles/README.md for one round FROM THE FILES under profiles/ (kernel-source hash, headline, kernel
durations, HBM traffic, vector-issue figures): numbers in that section are never typed by hand.

    python tools/profiles_readme.py r05

Replaces the block between `<!-- TAG:begin -->` and `<!-- TAG:end -->` (inserted in front of the previous round when missing)."""
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")

NOTES = {   # what each further file of the round is (facts about HOW it was made; its numbers are read from it where a table row quotes them)
    "ab_occupancy_curves.txt": "`tools/experiments/ab_r05_occupancy.sh`: `k_poa` / `k_window` time against resident wave slots (3-6 waves per SIMD), the data behind `T(w) = a + b / w` (DESIGN 5.2)",
    "ab_window_6_waves.txt": "first A/B of the 80-VGPR `k_window` at 6 656 bytes of LDS (six granules, 21 resident waves): flat",
    "ab_window_6_waves_lds_granule.txt": "the same inside five LDS granules (6 400 bytes): -8 % on cfg2 / cfg4 / cfg3; 5 632 slots in between",
    "ab_poa_register_allocation.txt": "`k_poa` with / without two edits outside its row loops: +3.5 % with both, back to round 4's time with both removed",
    "ab_compiler_flags.txt": "`tools/experiments/ab_r05_flags.sh`: `k_poa.hip` / `k_polish.hip` rebuilt with ten backend options each (scheduling strategies, metric bias, no post-RA scheduler, SGPR-spill pre-allocation, -O2 ...): none beats the default build beyond run-to-run noise, `max-ilp` / `max-memory-clause` lose 9-13 %",
    "ab_window_mask_stamps.txt": "`tools/experiments/ab_r05_mask_stamps.sh`: `k_window`'s sub-graph fixpoint with per-turn stamps instead of a clearing sweep (one sweep and one barrier less per turn) against the shipped clearing sweep: no faster (cfg2 +0.5 %, cfg3 -0.4 %, cfg4 +0.6 %); not shipped",
    "ab_window_row_trim.txt": "`tools/experiments/ab_r05_rowtrim.sh`: `k_window` before the hand-trimmed row tail (base), with it in the fast rows (rowtrim), in all banded rows (rowtrim2 = shipped)",
    "ab_window_run_loop.txt": "the same with runs of fast rows as a loop of their own (runloop2): fewer instructions per fast row, more register copies everywhere else, slower; not shipped",
    "ab_tb_uniform.txt": "`tools/experiments/ab_r05_tb_uniform.sh`: `k_window`'s traceback bookkeeping (row, column, ballot shift) kept in scalar registers (tbuni: 69 vector instructions fewer, 103 scalar more) against the shipped vector form (tbvec): cfg2 -0.9 %, cfg3 +0.3 %, cfg4 -0.2 %, inside run-to-run noise; not shipped",
    "ab_poa_near_rows_two_columns_per_lane.txt": "`tools/experiments/ab_r05_near_pair.sh` with `tools/experiments/k_poa_near_pair.patch` applied: `k_poa`'s near rows of 65-128 columns with two columns per lane, in both instances / in the WIDE instance only, against the shipped two chunks of 64 (_nopair): cfg2 +1.5 %, cfg4 -1.5 % / cfg4 -0.7 %, cfgL +1.1 %; not shipped",
    "phase_occupancy_cfg2.txt": "`tools/phase_occupancy.py 16384 cfg2` (phase-profiling build): rows and graph phases of `k_poa` / `k_window` at 3-6 resident waves per SIMD, each fitted as a + b / w and extrapolated to the hardware's eight",
    "two_handles_side_by_side.txt": "`tools/experiments/two_handles.py`: two handles fed from two host threads against one handle running the same batches in series: cfg2 +2.0 %, cfg4 +0.6 %",
    "ab_conk_saturating_subtract.txt": "`tools/experiments/ab_r05_conk.sh` (first form): `k_conk` with the gap candidate's subtraction saturating at zero (six vector instructions per cell) against round 4's seven (_oldconk): 8.2 -> 7.1 ms per 32 768 cfg2 reads",
    "ab_conk_scores_from_lds.txt": "`tools/experiments/ab_r05_conk.sh`: `k_conk` with the substitution scores as LDS byte reads (five vector instructions per cell) against the v_bfe_i32 form (_conk6): 7.1 -> 6.0 ms",
    "ab_peaks_blocked_smoothing.txt": "`tools/experiments/ab_r05_peaks_sg.sh`: `k_peaks` with the smoothing passes run once / twice (old_1 / old_2: their marginal cost) and with four outputs per thread as real calls (new_*), inlined (inl), inlined within 96 registers (inl5) and with tiles of 944 outputs (inl5b): no variant beats the shipped kernel",
    "ab_poa_consensus_prepass.txt": "`tools/experiments/ab_r05_poa_consensus.sh` with `tools/experiments/k_poa_consensus_prepass.patch`: `k_poa`'s consensus sweeps with a pre-pass and register hand-over, inlined (consinl) / as a real call (consfn), against the shipped kernel (base): cfg2 +3.5 % / +1.7 %, cfg4 +1.5 % / -0.8 %; not shipped",
    "ab_window_graph_code_trims.txt": "`tools/experiments/ab_r05_pred_row.sh`, four runs: `k_window` with `win_pred_row` branch-free (-1.6 / -1.0 / -0.3 % at cfg2 / cfg3 / cfg4), plus 24-bit multiplies in the traceback (nothing, not kept), plus the band centre by a 32-bit division (shipped with the first: -1.6 / -0.4 / -1.2 %), plus the traceback's divisions as multiply-shift (nothing, not kept)",
    "ab_window_graph_code_batch2.txt": "`tools/experiments/ab_r05_window_graph2.sh` with `tools/experiments/k_window_graph_batch2.patch`: `k_window`'s consensus pointer doubling with selects instead of nested branches (-443 instructions in the listing) and the graph loops' conditional loads made unconditional at clamped indices: nothing on the clock (cfg2 0.0 %, cfg3 +0.9 %, cfg4 0.0 %); not kept",
    "ab_poa_align_call.txt": "`tools/experiments/ab_r05_poa_align_call.sh`: `k_poa` with `poa_align` as a real call in every instance (cfg2 +10 %, cfg4 -2.7 %, cfgL -11 %), then in the WIDE instance only (shipped: cfg2 / cfg4 unchanged, cfgL -10 %)",
    "cli_3m_tmpfs_final_kernels.log": "`tools/cli_throughput.py 3000000 --dir /dev/shm` at the round's final kernels: 283.6 k reads/s end to end (10.6 s: 7.2 s in the device loop = 415 k reads/s, the rest is the process's fixed cost)",
    "inflate_bench_gpu_box_host.txt": "`tools/inflate_bench.py 200000` on the GPU box's host (CPU only): the native reader over a plain `.gz` FASTQ and the same file in BGZF layout, zlib (`C3_GZ_ZLIB=1`) against the own decoder, alternating: plain gzip 30.8 -> 49.8 k reads/s, BGZF ~150 k either way",
    "cli_500k_gz_own_decoder_ab.log": "`tools/cli_throughput.py 500000 --dir /dev/shm --gz --ab C3_GZ_ZLIB=1`: the command line end to end on a plain `.gz` input, four alternating runs: 27.1 k reads/s with zlib, 36.6 k with the own decoder (13.4 s of which ~2.5 s are the process's fixed cost)",
    "host_ceiling_bgzf_ranges.txt": "`tools/host_ceiling.py 1000000 --dir /dev/shm --workers 1,2,4 --bgzf` with the BGZF input cut into ranges of its inflated bytes (2 / 4 / 8 ranges): 122 / 110 / 75 k reads/s on the 16-core quota -- no better than one reader with eight inflating threads, so the ranges stay opt-in (`C3_BGZF_RANGES=1`)",
    "pmc_mem_cfg2.txt": "`tools/pmc_mem.sh 32768 cfg2`: TA / TCP / UTCL1 / TCC counters per kernel (one group per pass)",
    "vmem_rates_gfx950.txt": "`tools/ubench/vmem_rates.hip`: CU-cycles per vector memory instruction by shape, 24 / 12 / 4 waves per CU",
    "tmpfs_write_one_file_pwrite_mmap.txt": "`tools/experiments/tmpfs_write_bench.cpp`: one tmpfs file by `pwrite` / `mmap` from 1-16 threads against one file per thread",
    "host_ceiling_tmpfs_16core_quota_before_pin_policy.txt": "`tools/host_ceiling.py 2000000` before the page-lock policy, with the cgroup's CPU accounting (no throttling)",
    "host_ceiling_tmpfs_16core_quota_pin_policy.txt": "the same after it",
    "cli_1m_tmpfs_pin_policy_ab.log": "`tools/cli_throughput.py 1000000 --dir /dev/shm --ab C3_PIN_MIN_REFILLS=0`: four alternating runs, default policy / always page-locked",
    "cli_3m_tmpfs_pin_policy_ab.log": "the same with 3 M reads (buffers refilled 3.4 times: page-locked either way)",
    "bench_cfgL_50k_last_pass_beside_polish.json": "`bench.py --cfg cfgL`: the last POA pass on its own stream beside the polish",
    "bench_cfgL_50k_last_pass_in_series.json": "the same with `C3_NO_TAIL_OVERLAP=1`",
    "noisy_reads_window_time.txt": "`tools/noisy_window_time.py 8192 F` with the round-4 library beside this one: kernel times on cfg2-shaped reads with the error rates scaled by F (the constant ~2 ms tail of the full-size launch against the first launch's 7-8 %)",
    "band_verify_configs.txt": "`tools/band_verify_configs.py` (`C3_DEBUG_BAND=verify`): accepted band layers re-aligned unbanded on the device",
    "fuzz_parity.txt": "one line per `tools/fuzz_gate.sh` run, appended by the script: date, kernel-source hash, reads, mismatches",
    "phase_prof_cfg2.txt": "`tools/phase_prof.py` on the `-DC3_PHASE_PROF` build (shares of the wave time, not times)", "phase_prof_cfg4.txt": "the same, cfg4",
    # ---- round 6
    "ab_poa_steady_rows.txt": "`k_poa`'s steady rows (DESIGN 5.3): A/B on DISTINCT reads (`tools/ab_poa_distinct.py`) against `-DC3_STEADY=0` -- cfg2 -5.3 %, cfg2 at 15 % errors -3.8 %, cfg3 -1.9 %, cfg4 -0.7 % --; rows that a row further down reads (nothing); `poa_align` as a real call per instance on distinct reads (+1.3 to +3.4 %: not shipped); the tiled A/B incl. EIGHT waves per SIMD (+12 to +35 %); the phase-profiling build's cycles per row kind; one-predecessor near rows without a predecessor byte (-0.8 to -1.5 %, shipped); a steady run that starts from the LDS ring (+2.5 to +6.4 %, not shipped)",
    "poa_fast_row_run_lengths.txt": "`tools/poa_run_lengths.py` (the oracle counts): rows in runs of k consecutive fast rows, and in runs whose band moved exactly one column at both ends -- the prediction a steady row checks (DESIGN 5.3, first bullet)",
    "ubench_poa_rows_floor.txt": "`tools/ubench/poa_rows.hip`: the arithmetic of `k_poa`'s rows and nothing else at six waves per SIMD -- steady row 2 005 wave cycles, fast row 2 856, without the scans 1 476, without the direction byte 1 675: the floor of the row formulation",
    "pmc_icache_cfg2.txt": "`tools/pmc_icache.sh 16384 cfg2`: SQC_ICACHE_* per kernel for the shipped library, the ring-entry variant and `-DC3_STEADY=0`: 0.00 % of the instruction-cache requests miss in every build",
    "noisy_reads_window_time.txt": "`tools/noisy_window_time.py 8192 1.0 1.5 2.0 [3.0]`: `k_window` on cfg2-shaped reads with the error rates scaled, at the start of the round, with the band width remembered across the layers of a window (four margins), and with the full-size launch as a consumer beside the first launch (on / off)",
    "ab_window_consumer_beside_first_launch.txt": "`tools/ab_env.py CFG N C3_NO_WIN_CONSUMER`: one resident batch of distinct reads at bench size, the consumer on / off, alternating in one process: `k_window` -1.9 % (cfg2), -4.1 % (cfg4)",
    "gzip_parallel_decoder_throughput.txt": "`tools/experiments/host_r06_gz.sh` on the GPU box's 16-core quota: zlib, the single-thread decoder and `csrc/c3_gzpar.hpp` with 1-16 threads and three chunk sizes on 1 GB of FASTQ (`gzip -6`)",
    "cli_500k_gz_parallel_decoder_ab.log": "`tools/cli_throughput.py 500000 --gz --ab C3_GZ_SERIAL`: the command line on a `.gz` file, parallel decoder against the single-thread one, alternating cold processes",
    "cli_2m_gz_parallel_decoder.log": "`tools/cli_throughput.py 2000000 --gz` on two boxes",
    "gzip_parallel_decoder_throughput_index_plane.txt": "the same script with the decoder's final form (bytes + bitmap + 16-bit index plane): `gzip -6` 353 / 1 074 / 1 890 / 3 404 MB/s with 1 / 4 / 8 / 16 threads at 1 MiB chunks, 4 296 at 2 MiB (the default since), 3 730 at 4 MiB, 1 904 at 8 MiB; the same reads through `gzip -1`: 1 913 (1 MiB) - 2 214 (4 MiB) MB/s with 16 threads",
    "cli_2m_gz_index_plane_chunk_ab.log": "`tools/cli_throughput.py 2000000 --gz --ab C3_GZ_CHUNK=4194304` (`gzip -1` input): 125-145 k reads/s with 1 MiB chunks, 128-129 k with 4 MiB -- the parse stage (waiting for the inflating threads) is 12.3 s in all four runs",
    "cli_2m_gz_sorted_marks_rejected.log": "the intermediate decoder (marks in a position-sorted list) on `gzip -1` input: 61 k reads/s, half of what it replaced -- why the index plane exists",
    "gpu_tests_and_smoke_final_tree.txt": "`pytest -m gpu` (94 passed, 1 skipped: the two-GPU CLI test on a one-GPU box) and `__graft_entry__.smoke()` on the round's final tree, in the same call as `r06_bench_default.json`",
    "ab_peaks_registers_ballots_occupancy.txt": "`k_peaks` with the suppression loop in registers, ballot-compacted candidates and 7-8 workgroups per CU: 3.74 -> 3.60 ms per 32 768 cfg2 reads at 7 workgroups (3.85 at 6) -- first kept as a patch, shipped at 7 workgroups in the round's last kernel commit with `k_prep`'s mask reuse (4.65 -> 4.32 ms)",
    "cli_2m_gz6_chunks_per_round_ab.log": "`tools/cli_throughput.py 2000000 --gz --ab C3_GZ_ROUND=...` on one `gzip -6` member: 16 against 32 chunks per round with recycled buffers 175-181 k -> 197-200 k reads/s; the same A/B before recycling (147-152 -> 168-173 k, another box); 32 against 48: no change",
    "gzip_parallel_decoder_round_phases.txt": "`tools/gzpar_prof.cpp` (phase times per round) with 16 threads: find 2.7 ms, decode 12-15, convert 1.1 per round of 16 x 2 MiB; a round of 32 chunks takes twice its phases' sum standalone -- the harness frees 32 touched 8.5 MB buffers before the next round, which is what the reader's recycling removes",
    "cli_2m_gz1_final_reader.log": "`tools/cli_throughput.py 2000000 --gz --gz-level 1 --ab C3_GZ_ROUND=16`: the decoder's slow case (dense marks) with the final reader, 159-176 k reads/s; 130-132 k with 16 chunks per round",
    "inflate_bench_gpu_box_host.txt": "`tools/inflate_bench.py 200000`: the reader alone -- one `gzip -6` member 31 k (zlib) / 49.6 k (own decoder, one thread) / **207-219 k reads/s** (several inflating threads, 2.1 GB of FASTQ per second: the one parser thread behind them); BGZF 172-191 k (150-158 k in round 5: its members' CRC-32 by carry-less multiplication)",
    "cli_2m_bgzf_inflating_threads_ab.log": "`tools/cli_throughput.py 2000000 --bgzf --ab C3_GZ_THREADS=16`: BGZF input with 8 inflating threads (the default until then) 167 k reads/s, with 16 (the default since) **223-244 k**",
    "host_ceiling_gz_bgzf.txt": "`tools/host_ceiling.py 1000000 --gz / --bgzf` with the reader's wait time (`C3_STREAM_STATS`): 135 k reads/s on both (27.5 k on gzip in round 4); the one parser thread waits 3.2-3.4 s of its 5.5 s for inflated bytes; BGZF with 8 / 12 / 16 inflating threads 119-124 / 139 / 153-160 k",
    "cli_2m_bgzf_parser_prefetch_ab.log": "`tools/cli_throughput.py 2000000 --bgzf --ab C3_PARSE_PREFETCH=...`: the parser's prefetch cursor (16 KiB default) against 64 KiB and against none -- 192-252 k reads/s whatever the setting: inside the command line's run-to-run spread",
    "cli_2m_gz6.log": "`tools/cli_throughput.py 2000000 --gz --gz-level 6 --ab C3_GZ_SERIAL`: ONE gzip member at level 6 (deflated in pieces by the process pool, each ended by a sync flush), parallel decoder against the single-thread one, alternating",
    "cli_3m_tmpfs.log": "`tools/cli_throughput.py 3000000 --dir /dev/shm` with the round's library",
    "cli_1m_tmpfs_window_consumer_ab.log": "`tools/cli_throughput.py 1000000 --dir /dev/shm --ab C3_NO_WIN_CONSUMER`: four cold processes",
}


def kernel_ms(path):
    out = {}
    for row in csv.DictReader(open(path)):
        name = row.get("Name") or row.get("KernelName") or ""
        avg = float(row.get("AverageNs") or row.get("AverageNs ") or 0) / 1e6
        k = name.split("(")[0].replace("void ", "")
        if k.startswith(("k_", "c3mw::k_")) and avg >= 0.3:
            out[k] = (avg, int(row.get("Calls") or 0))
    return out


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
    L = []
    sha = None
    for wl in ("cfg2_100k", "cfg4_100k", "cfgL_50k"):
        f = os.path.join(P, "%s_pmc_traffic_%s.json" % (tag, wl))
        if os.path.exists(f):
            sha = sha or json.load(open(f)).get("kernel_src_sha")
    L.append("## round %s" % tag[1:].lstrip("0"))
    L.append("")
    L.append("Generated by `python tools/profiles_readme.py %s` from the files below; kernel sources hash `%s` (the value `bench.py` compares before it reports "
             "`roofline.traffic` / `roofline.valu`)." % (tag, sha))
    L.append("")
    L.append("| file | what (numbers read from the file) |")
    L.append("|---|---|")
    bd = os.path.join(P, "%s_bench_default.json" % tag)
    if os.path.exists(bd):
        o = json.loads(open(bd).read().strip().splitlines()[-1])
        oc = o.get("other_configs", {})
        L.append("| `%s_bench_default.json` | `python bench.py` (defaults): cfg2 headline **%.1f k reads/s** (%.1f ms per step, %d steps), kernels %s; `gcups_computed` %.0f; "
                 "parity sample %d reads / %d mismatches; other configs in the same process: %s; CPU baseline %.0f reads/s on %d cores |" % (
                     tag, o["value"] / 1e3, o["ms_per_step"], o["steps"], ", ".join("%s %.1f" % (k.replace("ms_", "k_"), v) for k, v in o["roofline"]["kernel_ms"].items()),
                     o["roofline"].get("gcups_computed", 0), o["parity_sample"]["reads"], o["parity_sample"]["mismatches"],
                     ", ".join("%s **%.1f k** (%d-read parity sample: %d mismatches)" % (c, v["value"] / 1e3, v["parity_sample"]["reads"], v["parity_sample"]["mismatches"]) for c, v in oc.items()),
                     o.get("cpu_baseline", {}).get("value", 0), o.get("cpu_baseline", {}).get("cores", 0)))
    for wl in ("cfg2_100k", "cfg4_100k", "cfgL_50k"):
        st = os.path.join(P, "%s_kernel_stats_bench_%s.csv" % (tag, wl))
        tr = os.path.join(P, "%s_pmc_traffic_%s.json" % (tag, wl))
        if os.path.exists(st):
            km = kernel_ms(st)
            L.append("| `%s` | one step of `bench.py --cfg %s --steps 1 --warmup 0 --no-cpu` under `rocprofv3 --kernel-trace --stats`: %s |" % (
                os.path.basename(st), wl.split("_")[0], ", ".join("`%s` %.2f ms" % (k, v[0]) for k, v in sorted(km.items(), key=lambda t: -t[1][0])[:7])))
        if os.path.exists(tr):
            t = json.load(open(tr))
            ks = sorted(t["kernels"].items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"])[:5]
            tot = sum(v["hbm_bytes_per_launch"] for v in t["kernels"].values())
            L.append("| `%s` | HBM bytes per launch, `(2*FETCH_SIZE + WRITE_SIZE)*1024` from two PMC passes of the same command: %s; all kernels of the step %.0f GB; kernel sources `%s` |" % (
                os.path.basename(tr), ", ".join("`%s` %.1f GB" % (k, v["hbm_bytes_per_launch"] / 1e9) for k, v in ks), tot / 1e9, t.get("kernel_src_sha")))
        for extra in ("%s_bench_%s.json" % (tag, wl), "%s_bench_under_rocprof_%s.json" % (tag, wl)):
            fe = os.path.join(P, extra)
            if os.path.exists(fe):
                try:
                    o = json.loads(open(fe).read().strip().splitlines()[-1])
                    L.append("| `%s` | bench line of `tools/profile_round.sh %s %s`%s: %.1f k reads/s |" % (extra, tag, wl.split("_")[0], " (the WRITE_SIZE pass)" if "rocprof" in extra else " (3 steps, unprofiled)", o["value"] / 1e3))
                except Exception:
                    pass
    for cfg in ("cfg2", "cfg4"):
        fj = os.path.join(P, "%s_sq_counters_%s.json" % (tag, cfg))
        if os.path.exists(fj):
            q = json.load(open(fj))
            L.append("| `%s`, `.txt` | `tools/pmc_sq.sh %s %s` (two passes of 8 SQ counters): %s; kernel sources `%s` |" % (
                os.path.basename(fj), q.get("reads"), cfg,
                "; ".join("`%s` %s vector wave-instructions per counted cell, %.3f per SIMD cycle = %.0f %% of the issue cycles at %.2f cycles each" % (
                    k, ("%.2f" % v["insts_per_cell"]) if v.get("insts_per_cell") else "-", v["insts_per_simd_cycle"], 100 * v["busy_frac"], v.get("cycles_per_inst") or q["cycles_per_inst_assumed"])
                    for k, v in q["kernels"].items() if k in ("k_poa", "k_window", "k_conk")), q.get("kernel_src_sha")))
    for name, note in NOTES.items():
        f = os.path.join(P, "%s_%s" % (tag, name))
        if os.path.exists(f):
            extra = ""
            if name == "fuzz_parity.txt":
                lines = [l for l in open(f).read().splitlines() if "fuzz_gate" in l]
                extra = " -- %d runs, last: %s" % (len(lines), lines[-1][:200] if lines else "")
            if name == "band_verify_configs.txt":
                extra = " -- " + "; ".join(l.strip() for l in open(f).read().splitlines() if l.startswith("cfg"))
            L.append("| `%s_%s` | %s%s |" % (tag, name, note, extra))
    block = "<!-- %s:begin -->\n%s\n<!-- %s:end -->\n" % (tag, "\n".join(L), tag)
    rd = os.path.join(P, "README.md")
    s = open(rd).read()
    if "<!-- %s:begin -->" % tag in s:
        s = re.sub(r"<!-- %s:begin -->.*?<!-- %s:end -->\n" % (tag, tag), lambda m: block, s, flags=re.S)
    else:
        i = s.index("## round ")
        s = s[:i] + block + "\n" + s[i:]
    open(rd, "w").write(s)
    print(block)


if __name__ == "__main__":
    main()
