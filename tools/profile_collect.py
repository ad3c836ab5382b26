#!/usr/bin/env python3
"""Copies what tools/profile_round.sh left under gpurun_out/round_<tag>/ into profiles/ (tracked): bench lines, rocprofv3 kernel stats,
the per-dispatch PMC CSVs, the derived summaries.   usage: tools/profile_collect.py r03"""
import glob, os, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O, P = os.path.join(ROOT, "gpurun_out", f"round_{tag}"), os.path.join(ROOT, "profiles")


def first(pattern):
    """the newest match: gpurun merges a run's files into gpurun_out/ without removing those of earlier runs (rocprofv3 names its
    files after the process id)"""
    g = glob.glob(pattern, recursive=True)
    return max(g, key=os.path.getmtime) if g else None


def cp(src, name):
    if src and os.path.exists(src):
        shutil.copy(src, os.path.join(P, f"{tag}_{name}"))
        print("copied", name)


for n in ("bench_default.json", "bench_one_process_virtual8.json", "bench_config5_world1.json", "lane_util.txt", "valu_issue_cycles.txt",
          "shard_times_4k.txt", "roofline_frac.txt", "lane_hist.txt", "build_probe.txt"):
    cp(os.path.join(O, n), n)
for wl in ("headline", "mesh1m_d4", "cornell_d1"):
    cp(first(os.path.join(O, f"stats_{wl}", "**", "*kernel_stats.csv")), f"{wl}_kernel_stats.csv")
    cp(os.path.join(O, f"stats_{wl}.json"), f"{wl}_bench.json")
for wl in ("mesh1m_d1", "mesh1m_d4", "cornell_d1", "mesh520_d1", "mesh520_d4"):
    for kind in ("fetch", "write", "sq"):
        cp(first(os.path.join(O, f"pmc_{kind}_{wl}", "**", "*counter_collection.csv")), f"{wl}_pmc_{kind}.csv")
if os.path.exists(os.path.join(O, "pmc_traffic.json")):
    shutil.copy(os.path.join(O, "pmc_traffic.json"), os.path.join(P, "pmc_traffic.json"))
    print("copied pmc_traffic.json")
