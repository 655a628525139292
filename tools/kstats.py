#!/usr/bin/env python3
"""Print rocprofv3 --kernel-trace --stats (csv) kernel statistics: name, calls, average / total duration."""
import csv
import glob
import sys

d = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 12
for f in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:top]:
        print("%-110s calls %5s  avg %10.1f us  total %10.3f ms" % (r["Name"][:110], r["Calls"], float(r["AverageNs"]) / 1e3,
                                                                       float(r["TotalDurationNs"]) / 1e6))
