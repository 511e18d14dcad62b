"""Turn the raw outputs of one profiling session (tools/profile_session.sh -> gpurun_out/<dir>/) into the committed profiles/ artefacts:
    python tools/make_profile_summary.py <tag> <gpurun_out/dir>
writes profiles/<tag>_bench_fwd.json, <tag>_kernel_stats.csv, <tag>_pmc_{fwd,bwd,sparse}.json, pmc_traffic.json and copies the
text reports (training profiles, sparse workloads, GPU test log) under the same tag.

PMC conventions (MI355X_MICROARCH.md): FETCH_SIZE / WRITE_SIZE / SQ counters come from SEPARATE rocprofv3 --pmc passes of the same
command (never combined with a trace domain); FETCH_SIZE reports half of the fetched bytes on gfx950, so HBM-side bytes per
launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024; GRBM_GUI_ACTIVE is summed over the 8 XCDs and SQ_VALU_MFMA_BUSY_CYCLES over the
1024 SIMDs, so MFMA-busy = busy / (gui_active / 8 * 1024); SQ_WAIT_* / SQ_ACTIVE_* are fractions of SQ_WAVE_CYCLES."""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

tag, src = sys.argv[1:3]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")


def short(name):
    """'void wmd::conv_wino32_kernel<8, 32, 2, 8, false>(wmd::ConvKArgs)' -> 'conv_wino32_kernel<8,32,2,8>' (bench.py's names)."""
    m = re.search(r"(?:wmd::)?([A-Za-z0-9_]+)(<[^>]*>)?\(", name)
    if not m:
        return name[:80]
    args = (m.group(2) or "").replace(" ", "")
    args = args.replace(",false,2>", ">")                  # dense instantiation, NBUF = 2 default of conv_fwd_kernel
    args = re.sub(r"(,false)+>$", ">", args)               # dense / pure-and-dense instantiations
    return m.group(1) + args


def counters(kind, which):
    """reads the reduced pmc_<kind>_<which>.csv of tools/pmc_reduce.py: (kernel, grid) -> counter -> (launches, mean)"""
    f = os.path.join(src, "pmc_%s_%s.csv" % (kind, which))
    rows = collections.defaultdict(dict)
    if os.path.exists(f):
        for r in csv.DictReader(open(f)):
            rows[(short(r["Kernel_Name"]), r["Grid_Size"])][r["Counter_Name"]] = (int(r["launches"]), float(r["mean"]))
    return rows


def summarise(kind):
    f, w, s = counters(kind, "fetch"), counters(kind, "write"), counters(kind, "mfma")
    out = {}
    avg = lambda x: x[1] if x else 0.0
    for key in sorted(set(f) | set(w) | set(s)):
        name, grid = key
        row = {"launches_seen": (f[key].get("FETCH_SIZE") or s[key].get("GRBM_GUI_ACTIVE") or (0, 0))[0]}
        fk, wk = avg(f[key].get("FETCH_SIZE")), avg(w[key].get("WRITE_SIZE"))
        row["fetch_kib_raw"], row["write_kib"] = round(fk, 1), round(wk, 1)
        row["hbm_bytes_per_launch"] = int((2 * fk + wk) * 1024)
        c = s[key]
        if avg(c.get("GRBM_GUI_ACTIVE")) > 0:
            simd_cycles = avg(c["GRBM_GUI_ACTIVE"]) / 8 * 1024
            row["gpu_cycles"] = int(avg(c["GRBM_GUI_ACTIVE"]) / 8)
            row["mfma_busy_frac"] = round(avg(c.get("SQ_VALU_MFMA_BUSY_CYCLES")) / simd_cycles, 3)
            wc = avg(c.get("SQ_WAVE_CYCLES"))
            if wc:
                row["waves_per_simd"] = round(4 * wc / simd_cycles, 2)
                for k, label in (("SQ_WAIT_ANY", "wait_frac"), ("SQ_WAIT_INST_ANY", "issue_stall_frac"), ("SQ_ACTIVE_INST_ANY", "issuing_frac")):
                    row[label] = round(avg(c.get(k)) / wc, 3)
            row["mfma_insts"] = int(avg(c.get("SQ_INSTS_MFMA")))
        out.setdefault(name, {})[grid] = row
    return out


def merged(per_grid):
    """launch-weighted average over the grids of one kernel name (what bench.py's per-name records average over)."""
    n = sum(max(r["launches_seen"], 1) for r in per_grid.values())
    tot = lambda k: sum(r.get(k, 0) * max(r["launches_seen"], 1) for r in per_grid.values()) / n
    return {"traffic_bytes_per_launch": int(tot("hbm_bytes_per_launch")), "fetch_kib_raw": round(tot("fetch_kib_raw"), 1),
            "write_kib": round(tot("write_kib"), 1), "mfma_busy_frac": round(tot("mfma_busy_frac"), 3),
            "by_grid_size": per_grid}


def conv_plan(name="conv_plan.txt"):
    """conv_plan.txt = stderr of one eager bench step under WMD_CONV_VERBOSE=1: the problem signature of every trunk layer with
    the kernel and grid it ran on -> {signature: (kernel, rocprof Grid_Size)} (Grid_Size = total work-items of the launch)."""
    f = os.path.join(src, name)
    plan = {}
    if os.path.exists(f):
        for ln in open(f):
            m = re.search(r"cfg (\S+) grid (\d+),(\d+),(\d+) block (\d+) sig (\S+)", ln)
            if m:
                plan[m.group(6)] = (m.group(1), str(int(m.group(2)) * int(m.group(3)) * int(m.group(4)) * int(m.group(5))))
    return plan


def algorithmic_bytes(sig):
    """inputs + weights + outputs once, fp32 (what bench.py's per-launch `bytes` is): sig = conv|B|H|W|C1|up1|C2|Cout|k"""
    B, H, W, C1, up1, C2, Cout, k = (int(v) for v in sig.split("|")[1:9])
    pix = B * H * W
    return int(4 * (pix * C1 / (up1 * up1) + pix * C2 + (C1 + C2) * k * k * Cout + pix * Cout))


def copy(rel, dst):
    files = glob.glob(os.path.join(src, rel))
    if files:
        shutil.copy(files[0], os.path.join(P, dst))
        return True
    return False


copy("bench.json", tag + "_bench_fwd.json")
copy("kernel_stats.csv", tag + "_kernel_stats.csv")
for f in glob.glob(os.path.join(src, "train_profile_*.txt")):
    shutil.copy(f, os.path.join(P, tag + "_" + os.path.basename(f)))
copy("sparse_workloads.txt", tag + "_sparse_workloads.txt")
copy("sparse_profile_thr0.15.txt", tag + "_sparse_profile_thr0.15.txt")
copy("sparse_timelines.txt", tag + "_sparse_timelines.txt")
copy("sparse_workloads_r03form.txt", tag + "_sparse_workloads_r03form.txt")
copy("tune_cache.json", tag + "_tune_cache.json")
copy("gpu_tests.txt", tag + "_gpu_tests.txt")
method = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc <SQ counters> GRBM_GUI_ACTIVE in separate passes (tools/profile_session.sh); "
          "per-launch averages; HBM-side bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (FETCH_SIZE reports half of the fetched bytes on gfx950)")
layers, fwd_kernels = {}, {}
for kind, what in (("fwd", "WMD_BENCH_GRAPH=0 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-train"),
                   ("fwd1024", "WMD_BENCH_GRAPH=0 python bench.py --workload fwd-1024 --steps 3 (KITTI ResNet50 1024x320, batch 8, forward)"),
                   ("bwd", "python tools/train_profile.py (decoder forward + backward, config 2 shapes)"),
                   ("sparse", "python tools/sparse_profile.py 0.15 (sparse KITTI decoder, batch 1)")):
    summ = summarise(kind)
    if not summ:
        continue
    json.dump({"_method": method, "command": what, "session": tag, "kernels": summ}, open(os.path.join(P, "%s_pmc_%s.json" % (tag, kind)), "w"), indent=1)
    if kind in ("fwd", "fwd1024"):
        # trunk layers by PROBLEM SIGNATURE (what bench.py's roofline.traffic reads): joined on (kernel, grid) with the plan dump;
        # two layers that share a kernel AND a grid cannot be told apart by the counters and are flagged instead of guessed
        # (both forward workloads land in ONE `layers` object: their signatures differ in batch and size)
        plan = conv_plan("conv_plan.txt" if kind == "fwd" else "conv_plan_1024.txt")
        fwd_kernels = {k: merged(v) for k, v in summ.items()} if kind == "fwd" else fwd_kernels
        for sig, (kern, grid) in sorted(plan.items()):
            twins = [s2 for s2, kg in plan.items() if kg == (kern, grid)]
            row = summ.get(kern, {}).get(grid)
            if row is None:
                continue
            layers[sig] = {"kernel": kern, "grid_size": grid, "hbm_bytes_per_launch": row["hbm_bytes_per_launch"],
                           "fetch_kib_raw": row["fetch_kib_raw"], "write_kib": row["write_kib"], "algorithmic_bytes": algorithmic_bytes(sig),
                           "mfma_busy_frac": row.get("mfma_busy_frac"), "ambiguous_with": [t for t in twins if t != sig] or None}
        json.dump({"_method": method, "session": tag, "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py, session " + tag,
                   "layers": layers, "kernels": fwd_kernels}, open(os.path.join(P, "pmc_traffic.json"), "w"), indent=1)
        for sig, v in layers.items():
            if sig not in plan:
                continue
            print("  layer %-38s %-34s %7.1f MB HBM vs %6.1f MB algorithmic (%.2fx)" % (sig, v["kernel"], v["hbm_bytes_per_launch"] / 1e6,
                                                                                      v["algorithmic_bytes"] / 1e6, v["hbm_bytes_per_launch"] / v["algorithmic_bytes"]))
    print("==", kind)
    for name, grids in sorted(summ.items()):
        for g, r in grids.items():
            print("  %-44s grid %-9s %7.1f MB  mfma %.3f  wait %.2f  stall %.2f" % (name, g, r["hbm_bytes_per_launch"] / 1e6, r.get("mfma_busy_frac", 0),
                                                                                r.get("wait_frac", 0), r.get("issue_stall_frac", 0)))


# ---- the ten largest kernels of the training step with their counters ------------------------------
tp = os.path.join(P, tag + "_train_profile_r18_640x192_bs12.txt")
pb = os.path.join(P, tag + "_pmc_bwd.json")
if os.path.exists(tp) and os.path.exists(pb):
    pm = json.load(open(pb))["kernels"]
    rows = []
    for ln in open(tp):
        m = re.match(r"\s+(\S+)\s+calls/step\s+(\d+)\s+([0-9.]+) ms/step\s+([0-9.]+) TFLOP/s", ln)
        if m:
            rows.append((m.group(1), int(m.group(2)), float(m.group(3)), float(m.group(4))))
    out = ["# Decoder forward + backward at BASELINE config 2 (R18 640x192, batch 12): the ten largest kernels with their counters",
           "", "hipEvent time per step from `tools/train_profile.py` (`%s`); counters per launch from the separate `rocprofv3 --pmc` passes of the"
           % os.path.basename(tp), "same script (`%s`: HBM-side bytes = (2 FETCH_SIZE + WRITE_SIZE) KiB, MFMA-busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024),"
           % os.path.basename(pb), "wait = SQ_WAIT_ANY / SQ_WAVE_CYCLES), one line per grid size the kernel is launched with.", "",
           "| kernel | calls / step | ms / step | TFLOP/s (algorithmic) | grid: HBM MB, MFMA-busy, wait |", "|---|---|---|---|---|"]
    for name, calls, ms, tf in rows[:10]:
        grids = {}
        for k, v in pm.items():       # a profile name without template arguments covers every instantiation
            if k == name or ("<" not in name and k.split("<")[0] == name):
                grids.update(v)
        det = "; ".join("%s: %.0f MB, %.2f, %.2f" % (g, r["hbm_bytes_per_launch"] / 1e6, r.get("mfma_busy_frac", 0.0), r.get("wait_frac", 0.0))
                        for g, r in sorted(grids.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"])[:4]) if grids else "n/a"
        out.append("| `%s` | %d | %.3f | %.1f | %s |" % (name, calls, ms, tf, det))
    open(os.path.join(P, tag + "_backward_top10.md"), "w").write("\n".join(out) + "\n")
