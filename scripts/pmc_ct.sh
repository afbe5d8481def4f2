#!/bin/bash
# PMC passes over one layer launch: scripts/pmc_ct.sh <tile> <fwd|dgrad> <tag>
cd /tmp && export TMPDIR=/tmp
TILE=$1; MODE=$2; TAG=$3
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
i=0
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVES" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VALU" \
         "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $C --output-format csv -d $OUT/p$i -- python3 scripts/ct_one.py $TILE $MODE 3 > $OUT/p$i.log 2>&1 || { echo pass $i failed; tail -5 $OUT/p$i.log; exit 1; }
done
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "conv_taps" in r["Kernel_Name"] or "gemm_fwd_bf16" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print(f"{k:28s} per launch {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
