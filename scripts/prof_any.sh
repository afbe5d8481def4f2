#!/bin/bash
# rocprofv3 kernel trace of an arbitrary python script: scripts/prof_any.sh <tag> <script> [args]; prints per-kernel stats
cd /tmp && export TMPDIR=/tmp
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 "$@" > $OUT/run.log 2>&1
echo rc=$?
python3 scripts/stats.py $OUT 30
