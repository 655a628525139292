#!/usr/bin/env python3
"""Per-launch averages IN LAUNCH ORDER over the last transitions of a rocprofv3 kernel trace (a transition starts at its
reconstruct kernel).  usage: per_layer.py <trace dir> <kernels per transition> <transitions to average>"""
import collections
import csv
import glob
import sys

d, per, last = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [(r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows]
starts = [i for i, n in enumerate(names) if "reconstruct_kernel" in n[0]]
acc = collections.defaultdict(list)
for t in starts[-last:]:
    for k, n in enumerate(names[t:t + per]):
        acc[(k, n[0])].append(n[1])
print("per launch, in launch order, averaged over the last %d transitions:" % last)
for (k, nm), v in sorted(acc.items()):
    print("  %2d %-100s %9.1f us" % (k, nm[:100], sum(v) / len(v)))
