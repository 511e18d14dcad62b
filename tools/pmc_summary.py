"""Summarise a rocprofv3 --pmc counter_collection.csv: per kernel (optionally per grid size) average counters."""
import collections
import csv
import glob
import sys

path = glob.glob(sys.argv[1])[0]
filt = sys.argv[2] if len(sys.argv) > 2 else "wmd"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(path)):
    if filt in r["Kernel_Name"]:
        agg[(r["Kernel_Name"][:64], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(k)
    for c, xs in sorted(v.items()):
        print("    %-32s n=%d avg=%.4g" % (c, len(xs), sum(xs) / len(xs)))
