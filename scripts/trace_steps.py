"""Per-step breakdown from a rocprofv3 kernel trace: takes the LAST `nsteps` optimizer steps (delimited by the
loss_finish_kernel dispatches, one per step), and prints, per (kernel, grid) bucket, launches per step, average duration and share, plus
GPU busy time vs wall time per step (the difference is launch gaps / host-bound time).
usage: python scripts/trace_steps.py <dir with *_kernel_trace.csv> [nsteps=5] [topN=40] [skip_last_steps=0]"""
import csv, glob, os, sys, collections, re

d = sys.argv[1]
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
skip = int(sys.argv[4]) if len(sys.argv) > 4 else 0      # ignore the last `skip` steps (bench.py's instrumented eager pass)
f = max(glob.glob(d + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# one loss_finish_kernel (fused loss; mse_fwd_kernel before it existed) per optimizer step (the Adam/EMA pass is issued in
# per-bucket slices all over backward)
mark = "loss_finish_kernel" if any("loss_finish_kernel" in r["Kernel_Name"] for r in rows) else "mse_fwd_kernel"
adam = [i for i, r in enumerate(rows) if mark in r["Kernel_Name"]]
assert len(adam) > nsteps + skip, "not enough optimizer steps in the trace"
lo, hi = adam[-nsteps - 1 - skip] + 1, adam[-1 - skip] + 1
sel = rows[lo:hi]
wall = (int(sel[-1]["End_Timestamp"]) - int(rows[lo - 1]["End_Timestamp"])) / 1e3 / nsteps
agg = collections.OrderedDict()
busy = 0.0
for r in sel:
    name = re.sub(r"\(.*", "", r["Kernel_Name"])
    name = name.replace("_Z20gemm_fwd_bf16_kernelIDF16b", "gemm_fwd_bf16<bf16,").replace("EEv13GemmFwdParams", ">")
    grid = "x".join(str(int(r[f"Grid_Size_{a}"]) // max(1, int(r[f"Workgroup_Size_{a}"]))) for a in "XYZ")
    key = (name[:60], grid)
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = agg.setdefault(key, [0, 0.0])
    a[0] += 1
    a[1] += dur
    busy += dur
busy /= nsteps
print(f"steps analysed: {nsteps}; wall/step {wall:.1f} us; GPU busy/step {busy:.1f} us; idle {wall - busy:.1f} us; "
      f"launches/step {len(sel) / nsteps:.0f}")
fam = collections.Counter()
for (name, grid), (n, t) in agg.items():
    fam[name.split("<")[0].split("IL")[0]] += t / nsteps
print("-- by kernel family (us/step):")
for k, v in fam.most_common(25):
    print(f"   {k[:70]:70s} {v:9.1f} {100 * v / busy:5.1f}%")
print("-- by (kernel, grid in workgroups):")
for (name, grid), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"   {name:60s} grid {grid:12s} x{n / nsteps:5.1f}/step avg {t / n:7.1f} us  {t / nsteps:8.1f} us/step")
