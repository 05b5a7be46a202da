# The round's profile commands in ONE gpurun call (round 6): kernel stats + PMC traffic (cfg2, cfg4, cfgL), SQ counters (cfg2, cfg4), memory-pipeline
# counters (cfg2), band verification incl. the noisy configs, phase profiles, the fuzz gate and the default bench -- tools/collect_round.sh r06 copies
# the summaries into profiles/ afterwards.    gpurun --timeout 3000 -- 'bash tools/experiments/final_r06.sh'
cd "$GRAFT_REPO_ROOT"
TAG=r06
bash tools/profile_round.sh $TAG cfg2 > gpurun_out/${TAG}_profile_cfg2.log 2>&1
bash tools/profile_round.sh $TAG cfg4 > gpurun_out/${TAG}_profile_cfg4.log 2>&1
bash tools/profile_round.sh $TAG cfgL > gpurun_out/${TAG}_profile_cfgL.log 2>&1
bash tools/pmc_sq.sh 32768 cfg2 > /dev/null 2>&1
bash tools/pmc_sq.sh 8192 cfg4 > /dev/null 2>&1
bash tools/pmc_mem.sh 32768 cfg2 > /dev/null 2>&1
python tools/band_verify_configs.py > gpurun_out/${TAG}_band_verify_configs.txt 2>&1
for c in cfg2 cfg4; do C3POA_LIB=c3poa_amd/lib/libc3poa_hip_prof.so python tools/phase_prof.py 16384 $c > gpurun_out/${TAG}_phase_prof_$c.txt 2>&1; done
bash tools/fuzz_gate.sh 6100 $TAG > gpurun_out/${TAG}_fuzz_gate.log 2>&1
python bench.py > gpurun_out/bench_default_final.json 2> gpurun_out/bench_default_final.err
tail -c 600 gpurun_out/bench_default_final.json; tail -2 gpurun_out/${TAG}_fuzz_gate.log; cat gpurun_out/${TAG}_band_verify_configs.txt
