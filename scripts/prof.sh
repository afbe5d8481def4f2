#!/bin/bash
# rocprofv3 kernel trace of the bench's training region (the command the roofline object is measured on):
#   python3 bench.py --steps 10 --warmup 2 --no-cpu [extra args]
# A first, unprofiled run of the same command writes the kernel-variant choices (MADIFF_TUNE_CACHE); the profiled run reads
# them, so its --stats summary holds no autotuning launches: warm-up steps, the capture and graph replays only.
# usage: scripts/prof.sh <tag> [extra bench args]; the caller copies the stats CSV into profiles/.
cd /tmp && export TMPDIR=/tmp
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export MADIFF_TUNE_CACHE=$OUT/tune.json
timeout -k 10 400 python3 bench.py --steps 4 --warmup 2 --no-cpu "$@" > $OUT/tune.log 2>&1 || { echo "tuning run failed"; tail -5 $OUT/tune.log; exit 1; }
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --steps 10 --warmup 2 --no-cpu "$@" > $OUT/bench.log 2>&1
echo rc=$?
grep '^{' $OUT/bench.log | tail -1 > $OUT/bench.json
find $OUT -name "*kernel_stats*" | head
