"""Print the top rows of a rocprofv3 kernel_stats.csv: name, calls, average microseconds.  python tools/kstats.py file.csv [rows] [filter]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
flt = sys.argv[3] if len(sys.argv) > 3 else ""
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in [r for r in rows if flt in r["Name"]][:n]:
    print("%-98s %5s %9.1f us  %5.1f %%" % (r["Name"][:98], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
