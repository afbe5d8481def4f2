#!/bin/bash
# usage: scripts/pmc_probe.sh <tag> <counters (space separated, one pass)> -- <python script + args>
# one rocprofv3 --pmc pass (kernel-trace only) of a short program; prints per-kernel counter sums
cd /tmp && export TMPDIR=/tmp
TAG=$1; shift
CTRS=""
while [ "$1" != "--" ]; do CTRS="$CTRS $1"; shift; done
shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -s KILL 120 rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $OUT -- python3 "$@" > $OUT/run.log 2>&1
echo "rc=$?"; tail -2 $OUT/run.log
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/**/*counter_collection.csv", recursive=True)
if not f: print("no counter file"); raise SystemExit
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
seen = set()
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"][:60]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (k, r.get("Dispatch_Id"))
    if key not in seen: seen.add(key); cnt[k] += 1
for k, d in agg.items():
    if "wgrad" in k or "conv_taps" in k or "gemm" in k:
        print(k, "dispatches", cnt[k], {c: round(v / cnt[k]) for c, v in d.items()})
PY
