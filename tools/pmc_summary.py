#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output per kernel: median counter values + duration (ms)."""
import collections
import csv
import glob
import sys

d = sys.argv[1]
cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(cc)):
    agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for r in csv.DictReader(open(kt)):
    dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
filt = sys.argv[2] if len(sys.argv) > 2 else ""
for k, v in agg.items():
    if filt not in k:
        continue
    ds = sorted(dur[k])
    print("%s\n   launches %d  duration ms min %.4f med %.4f" % (k[:110], len(ds), ds[0], ds[len(ds) // 2]))
    for c, vals in sorted(v.items()):
        vals = sorted(vals)
        print("   %-28s med %.5g" % (c, vals[len(vals) // 2]))
