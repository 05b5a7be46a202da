#!/bin/bash
# Copy what one gpurun call of the round's profile commands merged into gpurun_out/ into profiles/ (run locally, sources = the ones profiled):
#   tools/collect_round.sh r05 [fuzz gate line from the call's log]
TAG=${1:-r06}
for c in cfg2:cfg2_100k cfg4:cfg4_100k cfgL:cfgL_50k; do
  cfg=${c%%:*}; wl=${c##*:}; R=gpurun_out/${TAG}_$cfg
  python tools/collect_profiles.py $TAG $R/stats $R/fetch $R/write $R/bench.json $wl > /dev/null 2>&1
  cp $R/bench.json profiles/${TAG}_bench_$wl.json; cp $R/bench_write.json profiles/${TAG}_bench_under_rocprof_$wl.json
done
for c in cfg2 cfg4; do cp gpurun_out/pmcsq_$c/summary.json profiles/${TAG}_sq_counters_$c.json; cp gpurun_out/pmcsq_$c/summary.txt profiles/${TAG}_sq_counters_$c.txt; done
cp gpurun_out/pmcmem_cfg2/summary.txt profiles/${TAG}_pmc_mem_cfg2.txt
cp gpurun_out/${TAG}_band_verify_configs.txt gpurun_out/${TAG}_phase_prof_cfg2.txt gpurun_out/${TAG}_phase_prof_cfg4.txt profiles/
cp gpurun_out/bench_default_final.json profiles/${TAG}_bench_default.json
python tools/profiles_readme.py $TAG > /dev/null
