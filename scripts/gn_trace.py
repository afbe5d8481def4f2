"""Median GPU duration of the GN kernels per gn_bench configuration, from a rocprofv3 kernel trace directory."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "gn_mish" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
i, out = 0, []
while i < len(rows):
    j = i
    while j < len(rows) and rows[j]["Kernel_Name"] == rows[i]["Kernel_Name"] and j - i < 33:
        j += 1
    d = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows[i:j])
    if j - i > 1:
        out.append(("f" if "fwd" in rows[i]["Kernel_Name"] else "b") + "%.1f" % d[len(d) // 2])
    i = j
print(" ".join(out))
