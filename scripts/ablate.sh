#!/bin/bash
# Marginal wall-clock cost of each kernel family inside the real (graph-replayed) step: the bench's training region with
# the family's launches skipped (results are garbage, only the step time means anything).  usage: scripts/ablate.sh
cd $GRAFT_REPO_ROOT
for fam in none gn_mish_bwd_rb_kernel gemm_wgrad_bf16_kernel adam_ema_kernel attn_bwd_v2_kernel gn_mish_bwd_rb_kernel,gemm_wgrad_bf16_kernel,adam_ema_kernel; do
  MADIFF_AUTOTUNE_CACHE=gpurun_out/tune_ablate.json MADIFF_ABLATE=$([ $fam = none ] && echo "" || echo $fam) timeout -k 10 200 python bench.py --no-sample --no-cpu --steps 30 > gpurun_out/abl.json 2> gpurun_out/abl.err || { tail -5 gpurun_out/abl.err; exit 1; }
  python - "$fam" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/abl.json").read().strip().splitlines()[-1])
print(f"ablate {sys.argv[1]:70s} {d['ms_per_step']:.3f} ms/step", flush=True)
PY
done
