"""Reduce one rocprofv3 --pmc output directory to per-(kernel, grid, counter) averages and delete the raw files
(a counter_collection.csv of a training step is tens of MB; gpurun merges at most 64 MiB back).
    python tools/pmc_reduce.py <dir>  ->  <dir>.csv   (Kernel_Name, Grid_Size, Counter_Name, launches, mean)"""
import collections
import csv
import glob
import os
import shutil
import sys

d = sys.argv[1].rstrip("/")
acc = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "wmd" not in r["Kernel_Name"]:
            continue
        a = acc[(r["Kernel_Name"], r["Grid_Size"], r["Counter_Name"])]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
with open(d + ".csv", "w", newline="") as out:
    w = csv.writer(out)
    w.writerow(["Kernel_Name", "Grid_Size", "Counter_Name", "launches", "mean"])
    for (k, g, c), (n, s) in sorted(acc.items()):
        w.writerow([k, g, c, n, "%.3f" % (s / n)])
shutil.rmtree(d, ignore_errors=True)
print("%s: %d rows" % (d + ".csv", len(acc)))
