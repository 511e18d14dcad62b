"""Turn the raw rocprofv3 outputs of one profiling session (gpurun_out/) into the committed profiles/ artefacts:
    python tools/make_profile_summary.py <tag> <bench.json> <stats_dir> <pmc_fetch_dir> <pmc_write_dir> <pmc_mfma_dir> [tune_cache.json]
writes profiles/<tag>_bench_fwd.json, <tag>_kernel_stats.csv, <tag>_pmc_fwd.json, pmc_traffic.json."""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

tag, bench, stats, fetch, write, mfma = sys.argv[1:7]      # mfma may be "-" (no SQ pass in this session)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")


def per_kernel(d, counter):
    a = collections.defaultdict(list)
    for r in csv.DictReader(open(glob.glob(os.path.join(d, "*", "*counter_collection.csv"))[0])):
        if r["Counter_Name"] == counter:
            a[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return a


shutil.copy(bench, os.path.join(P, tag + "_bench_fwd.json"))
shutil.copy(glob.glob(os.path.join(stats, "*", "*kernel_stats.csv"))[0], os.path.join(P, tag + "_kernel_stats.csv"))
if len(sys.argv) > 7:
    shutil.copy(sys.argv[7], os.path.join(P, tag.split("_")[0] + "_tune_cache.json"))
f, w = per_kernel(fetch, "FETCH_SIZE"), per_kernel(write, "WRITE_SIZE")
mb, ga = (per_kernel(mfma, "SQ_VALU_MFMA_BUSY_CYCLES"), per_kernel(mfma, "GRBM_GUI_ACTIVE")) if mfma != "-" else ({}, {})
out, traffic = {}, {}
for k in f:
    if "wmd" not in k:
        continue
    row = {"launches": len(f[k]), "FETCH_SIZE_KiB_avg": sum(f[k]) / len(f[k]),
           "WRITE_SIZE_KiB_avg": sum(w.get(k, [0])) / max(1, len(w.get(k, [1])))}
    if k in mb and sum(ga.get(k, [0])):
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs; MFMA busy cycles over the 1024 SIMDs
        row["mfma_busy_frac"] = (sum(mb[k]) / len(mb[k])) / ((sum(ga[k]) / len(ga[k])) / 8 * 1024)
    out[k] = row
    m = re.search(r"(conv_fwd_kernel|conv_wino_kernel)<([0-9a-z, ]+)>", k)
    if m:
        name = m.group(1) + "<" + m.group(2).replace(" ", "").replace(",false,2", "") + ">"
        name = name.replace(",false>", ">")      # dense instantiation of the Winograd kernel (MASKED = false)
        if "true" in name:
            continue
        traffic[name] = {"traffic_bytes_per_launch": int((2 * row["FETCH_SIZE_KiB_avg"] + row["WRITE_SIZE_KiB_avg"]) * 1024),
                         "fetch_kib_raw": round(row["FETCH_SIZE_KiB_avg"], 1), "write_kib": round(row["WRITE_SIZE_KiB_avg"], 1),
                         "mfma_busy_frac": round(row.get("mfma_busy_frac", 0.0), 3)}
json.dump(out, open(os.path.join(P, tag + "_pmc_fwd.json"), "w"), indent=1)
json.dump({"_method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / SQ counters in separate passes over `WMD_BENCH_GRAPH=0 python "
                      "bench.py --steps 3 --warmup 2 --no-cpu-baseline` with the committed autotune cache; per-launch averages; bytes = "
                      "(2*FETCH_SIZE + WRITE_SIZE)*1024 (MI355X_MICROARCH.md: FETCH_SIZE reports half of the fetched bytes on gfx950; "
                      "WRITE_SIZE calibrated exact on the IDWT kernel: 11520 KiB reported = 11520 KiB written)",
           "session": tag, "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py, session " + tag, "kernels": traffic}, open(os.path.join(P, "pmc_traffic.json"), "w"), indent=1)
for k, v in sorted(traffic.items()):
    print(k, v)
