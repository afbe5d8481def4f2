import csv, os, sys, glob
f = max(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
for r in rows[:n]:
    print("%-80s calls %6s avg %9.1f us  total %8.2f ms  %5.1f%%" % (r['Name'][:80], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6, 100*float(r['TotalDurationNs'])/tot))
print("total kernel ms", tot/1e6)
