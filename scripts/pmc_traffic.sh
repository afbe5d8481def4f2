#!/bin/bash
# HBM-side traffic of the GEMM kernels of one train step: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; they do not
# fit one pass) over a short eager run of the bench's training step; writes gpurun_out/pmc_traffic.json (copy it to
# profiles/pmc_traffic.json).  Eager on purpose: one row per dispatch with the kernel's name.
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_traffic
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
CMD="bench.py --steps 2 --warmup 2 --no-sample --no-cpu --no-dp1 --no-graph"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 300 rocprofv3 --pmc $C --output-format csv -d $OUT/$C -- python3 $CMD > $OUT/$C.log 2>&1 || { echo "$C pass failed"; tail -5 $OUT/$C.log; exit 1; }
done
python3 - <<PY
import csv, glob, json, collections, os, re
def collect(counter):
    f = max(glob.glob("$OUT/%s/**/*counter_collection.csv" % counter, recursive=True), key=os.path.getsize)
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return rows
out = {}
fam = lambda n: re.sub(r"^void ", "", n).split("<")[0].split("(")[0]
FAMS = ("conv_taps_bf16_kernel", "gemm_fwd_bf16_kernel", "gemm_wgrad_taps8_bf16_kernel", "gemm_wgrad_taps_bf16_kernel",
        "gemm_wgrad_bf16_kernel", "gn_mish_bwd_rb_kernel", "gn_mish_fwd_rb_kernel", "adam_ema_kernel", "attn_fwd_v2_kernel",
        "attn_bwd_v2_kernel", "attn_fwd_mfma_kernel", "attn_bwd_mfma_kernel")
def family(name):
    for f in FAMS:
        if f in name:
            return f
    return None
res = {}
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = collect(counter)
    adam = [i for i, r in enumerate(rows) if "loss_finish_kernel" in r["Kernel_Name"]]   # one per optimizer step
    lo, hi = adam[-3] + 1, adam[-1] + 1            # the last two optimizer steps (steady state, after autotuning)
    for r in rows[lo:hi]:
        f = family(r["Kernel_Name"])
        if f:
            d = res.setdefault(f, {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "n": {"FETCH_SIZE": 0, "WRITE_SIZE": 0}})
            d[counter] += float(r["Counter_Value"])
            d["n"][counter] += 1
kernels = {}
for f, d in res.items():
    n = d["n"]["FETCH_SIZE"]
    if n == 0 or d["n"]["WRITE_SIZE"] == 0:     # launched in one pass only (a tuning choice that differed between the runs)
        continue
    fetch = 2.0 * 1024.0 * d["FETCH_SIZE"] / n      # KB -> bytes, x2: gfx950 tallies 128-byte read requests at 64 bytes
    write = 1024.0 * d["WRITE_SIZE"] / d["n"]["WRITE_SIZE"]
    kernels[f] = {"fetch_bytes": fetch, "write_bytes": write, "bytes_per_launch": fetch + write, "launches_per_step": n / 2.0,
                  "command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE -- python3 $CMD"}
    print("%-30s launches/step %5.1f  fetch %8.2f MB  write %8.2f MB per launch" % (f, n / 2.0, fetch / 1e6, write / 1e6))
json.dump({"kernels": kernels, "note": "FETCH_SIZE doubled (gfx950 correction, MI355X_MICROARCH.md HBM section); per-launch averages over the last 2 steps"},
          open("$GRAFT_REPO_ROOT/gpurun_out/pmc_traffic.json", "w"), indent=1)
PY
