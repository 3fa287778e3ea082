"""Prints the per-kernel summary of a rocprofv3 --kernel-trace --stats output directory."""
import csv
import glob
import sys

for d in sys.argv[1:]:
    fs = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)
    if not fs:
        print(d, ": no kernel_stats.csv")
        continue
    print("==", d)
    for r in list(csv.DictReader(open(fs[0])))[:12]:
        print("  %-64s n=%5s avg %8.2f min %8.2f max %8.2f us  %5s%%" % (
            r["Name"][:64], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3,
            float(r["MaxNs"]) / 1e3, r["Percentage"][:5]))
