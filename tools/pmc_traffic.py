#!/usr/bin/env python3
"""profiles/pmc_traffic.json from the --pmc passes of tools/profile_round.sh.

usage: tools/pmc_traffic.py <round dir> <out json>
Per workload (cornell_d1, mesh1m_d1, mesh1m_d4), over the non-counting k_segment launches of the timed region:
  l2_fabric_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE counts half of the fetched bytes
      (MI355X_MICROARCH.md, HBM section); both counters are in KiB.  These are bytes crossing L2 <-> Infinity Fabric; with the
      scene resident in the 256 MiB Infinity Cache the bytes that reach HBM are fewer still.
  valu_issue: busy = 4 * SQ_ACTIVE_INST_VALU / (SIMDs * GRBM_GUI_ACTIVE / 8)   (SQ_* count quad-cycles; a wave64 VALU instruction
      holds its SIMD's issue port ~4 cycles — measured, scratch/ubench in DESIGN.md §5; GRBM_GUI_ACTIVE is summed over the 8 XCDs);
      lane_util = SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU); frac = busy * lane_util = the share of the chip's
      lane-cycles doing enabled VALU work.  This, not HBM, is what bounds the kernel.
"""
import collections, csv, glob, json, os, sys

N_SIMD = 256 * 4


def per_kernel(dirname):
    agg = collections.defaultdict(list)
    for f in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            agg[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
    return agg


def entry_from_dirs(dirs, wl):
    """dirs: {"fetch" | "write" | "sq" | "grbm": directory of that --pmc pass}; wl = "<workload>_d<depth>".  None if no pass
    saw the segment kernels."""
    agg = {}
    for d in dirs.values():
        agg.update(per_kernel(d))
    names = sorted({k for k, _ in agg if k.startswith("void crt::k_segment<") and k.split(",")[1].strip() == "false"})
    if not names:
        return None

    def total(c):       # summed over the segment kernels (first + bounce), per launch
        vals = [v for k in names for v in agg.get((k, c), [])]
        return sum(vals), len(vals)
    fetch, n_l = total("FETCH_SIZE")
    write, n_w = total("WRITE_SIZE")
    hit, _ = total("TCC_HIT_sum")
    miss, _ = total("TCC_MISS_sum")
    e = {"kernels": names, "dispatches": n_l}
    if n_l and n_w:
        e.update({"FETCH_SIZE_KB_per_launch": round(fetch / n_l, 1), "WRITE_SIZE_KB_per_launch": round(write / n_w, 1),
                  "l2_fabric_bytes_per_launch": int((2 * fetch / n_l + write / n_w) * 1024),
                  "correction": "(2*FETCH_SIZE + WRITE_SIZE)*1024: gfx950 FETCH_SIZE reports half of the fetched bytes (MI355X_MICROARCH.md, HBM); L2<->fabric traffic, not HBM",
                  "l2_hit_rate": round(hit / max(1.0, hit + miss), 3)})
    act, n_sq = total("SQ_ACTIVE_INST_VALU")
    thr, _ = total("SQ_THREAD_CYCLES_VALU")
    insts, _ = total("SQ_INSTS_VALU")
    waves, _ = total("SQ_WAVES")
    gui, n_g = total("GRBM_GUI_ACTIVE")
    if n_sq and n_g and act > 0:
        busy_raw = 4.0 * (act / n_sq) / (N_SIMD * (gui / n_g) / 8.0)
        # 4 cycles per wave64 VALU instruction is the nominal figure; the microbenchmark measures 3.6 - 4.3 (profiles/r02_valu_rate_ubench.txt),
        # so a launch that keeps every SIMD issuing can come out a little above 1: reported as 1, with the raw value next to it
        busy = min(1.0, busy_raw)
        lane = thr / (64.0 * act)
        e["valu_issue"] = {"busy": round(busy, 3), "busy_raw": round(busy_raw, 3), "lane_util": round(lane, 3), "frac": round(busy * lane, 3),
                           "valu_instructions_per_wave": round(insts / max(1.0, waves), 1),
                           "source": "rocprofv3 --pmc passes of `bench.py --workload %s --depth %s --spp 1`; formulae in tools/pmc_traffic.py"
                                     % tuple(wl.split("_d"))}
    return e


# the four counter groups, one rocprofv3 pass each (MI355X_MICROARCH.md: separate --pmc passes)
PASSES = {"fetch": ["FETCH_SIZE", "TCC_HIT_sum"], "write": ["WRITE_SIZE", "TCC_MISS_sum"],
          "sq": ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY"],
          "grbm": ["GRBM_GUI_ACTIVE", "GRBM_TA_BUSY"]}


def main(round_dir, out):
    res = {}
    for wl in ("cornell_d1", "mesh1m_d1", "mesh1m_d4"):
        e = entry_from_dirs({kind: os.path.join(round_dir, f"pmc_{kind}_{wl}") for kind in PASSES}, wl)
        if e:
            res[wl] = e
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
