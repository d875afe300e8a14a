"""Print avg duration of kernels whose name contains one of the given substrings from a rocprofv3 kernel_stats.csv."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
keys = sys.argv[2:]
for r in rows:
    nm = r["Name"].split("(")[0][:60]
    if not keys or any(k in nm for k in keys):
        print("%-62s calls %5s avg_us %10.1f total_ms %9.2f" % (nm, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
