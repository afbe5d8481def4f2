"""debug: where the adam_ema_kernel slices run inside a step (from a rocprofv3 kernel trace)"""
import csv, glob, os, sys
d = sys.argv[1]
f = max(glob.glob(d + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
pk = [i for i, r in enumerate(rows) if "mse_fwd_kernel" in r["Kernel_Name"]]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 12
lo, hi = pk[-skip - 1], pk[-skip]
t0 = int(rows[lo]["Start_Timestamp"])
print("step wall us", (int(rows[hi]["Start_Timestamp"]) - t0) / 1e3, "kernels", hi - lo)
for r in rows[lo:hi]:
    if "adam" in r["Kernel_Name"]:
        s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
        print(f"adam  start {s:8.1f} us  end {e:8.1f} us  dur {e - s:7.1f} grid {int(r['Grid_Size_X']) // 256}")
