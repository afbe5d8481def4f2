#!/bin/bash
# rocprofv3 kernel trace of the sampling leg alone (scripts/sample_prof.py: spread, 1024 episodes, K=200, guided), steady
# state: a first unprofiled run writes the kernel-variant choices (MADIFF_TUNE_CACHE), the profiled run reads them, so the
# --stats summary holds no autotuning launches.  usage: scripts/sample_prof.sh <tag> [episodes] [workload]
cd /tmp && export TMPDIR=/tmp
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export MADIFF_TUNE_CACHE=$OUT/tune.json
timeout -k 10 400 python3 scripts/sample_prof.py "$@" > $OUT/tune.log 2>&1 || { echo "tuning run failed"; tail -5 $OUT/tune.log; exit 1; }
tail -1 $OUT/tune.log
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 scripts/sample_prof.py "$@" > $OUT/run.log 2>&1
echo rc=$?
tail -1 $OUT/run.log
python3 scripts/stats.py $OUT 40
