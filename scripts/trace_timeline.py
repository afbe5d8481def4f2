"""Ordered timeline of ONE optimizer step from a rocprofv3 kernel trace (the step before the last `skip` ones): start offset,
duration, queue, grid and kernel name of every dispatch.  usage: python scripts/trace_timeline.py <dir> [skip=0]"""
import csv, glob, os, re, sys
d = sys.argv[1]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
f = max(glob.glob(d + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
mark = "loss_finish_kernel" if any("loss_finish_kernel" in r["Kernel_Name"] for r in rows) else "mse_fwd_kernel"
marks = [i for i, r in enumerate(rows) if mark in r["Kernel_Name"]]
lo, hi = marks[-2 - skip], marks[-1 - skip]
t0 = int(rows[lo]["Start_Timestamp"])
qs = {}
for r in rows[lo:hi]:
    q = qs.setdefault(r.get("Queue_Id", "?"), len(qs))
    name = re.sub(r"\(.*", "", r["Kernel_Name"])
    name = re.sub(r"^void ", "", name)
    grid = "x".join(str(int(r[f"Grid_Size_{a}"]) // max(1, int(r[f"Workgroup_Size_{a}"]))) for a in "XYZ")
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:7.1f} q{q} {grid:>12s} {name[:110]}")
