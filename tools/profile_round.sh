#!/bin/bash
# One round of profiles for a bench workload (run on the GPU box through gpurun):
#   tools/profile_round.sh TAG CFG [bench args...]     e.g.  tools/profile_round.sh r02 cfg2
# writes gpurun_out/TAG_CFG/{stats,fetch,write}/ (rocprofv3 databases) + the bench lines of each pass.
# Counters are collected in their own passes (kernel trace only), as the pool requires.
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
TAG=$1; CFG=$2; shift 2
# profiled runs must be the single-process `python3 bench.py` right after `--`: with --gpus N > 1 bench.py becomes a launcher
# (python -> torch.distributed.run -> workers), i.e. an exec hop behind a profiler that has already initialised the GPU
case " $* " in *" --gpus "[2-9]*|*" --gpus="[2-9]*) echo "profile_round.sh: --gpus > 1 cannot be profiled (launcher hop after --)" >&2; exit 2;; esac
R=gpurun_out/${TAG}_${CFG}; rm -rf $R; mkdir -p $R
ARGS="--cfg $CFG --steps 1 --warmup 0 --no-cpu --other-configs none $*"
echo "python3 bench.py $ARGS" > $R/command.txt
timeout 900 rocprofv3 --kernel-trace --stats -d $R/stats -o s -- python3 bench.py $ARGS > $R/bench_stats.json 2> $R/stats.err
# (counter passes serialise kernel dispatch: k_window's consumer beside the first launch (round 6) would only wait out its bounded spin)
export C3_NO_WIN_CONSUMER=1
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/fetch -o b -- python3 bench.py $ARGS > $R/bench_fetch.json 2> $R/fetch.err
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/write -o b -- python3 bench.py $ARGS > $R/bench_write.json 2> $R/write.err
unset C3_NO_WIN_CONSUMER
python3 bench.py --cfg $CFG --steps 3 --warmup 2 $* > $R/bench.json 2> $R/bench.err
ls -la $R $R/stats | head -30
