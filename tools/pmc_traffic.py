#!/usr/bin/env python3
"""profiles/pmc_traffic.json from the --pmc passes of tools/profile_round.sh.

usage: tools/pmc_traffic.py <round dir> <out json>
Per workload (cornell_d1, mesh1m_d1, mesh1m_d4), over the non-counting k_segment launches of the timed region:
  l2_fabric_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE counts half of the fetched bytes
      (MI355X_MICROARCH.md, HBM section); both counters are in KiB.  These are bytes crossing L2 <-> Infinity Fabric; with the
      scene resident in the 256 MiB Infinity Cache the bytes that reach HBM are fewer still.
  valu_issue: busy = 2 * SQ_INSTS_VALU / (SIMDs * GRBM_GUI_ACTIVE / 8): the share of the vector issue slots used (a SIMD-32 issues one
      wave64 instruction per 2 cycles; GRBM_GUI_ACTIVE is summed over the 8 XCDs); busy_vs_mix_floor = the same against the 3.3 cycles
      this kernel's instruction mix needs at best (tools/ubench/valu_issue_cycles.hip); lane_util = SQ_THREAD_CYCLES_VALU /
      (64 * SQ_ACTIVE_INST_VALU); frac = busy * lane_util = the share of the chip's lane-cycles doing enabled VALU work.
"""
import collections, csv, glob, json, os, sys

N_SIMD = 256 * 4


def per_kernel(dirname, tail=0):
    """{(kernel, counter): [value per dispatch]} of one pass.  `tail` = how many of the LAST dispatches of the non-counting segment
    kernels to leave out: bench.py ends every run with min(10, steps x spp) single-sample frames rendered with full event timing, and
    those launches — a quarter of the work of a 4-sample launch, some of them by the same kernel — must not be averaged in."""
    rows = []
    for f in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))

    def is_seg(r):
        n = r["Kernel_Name"].split("(")[0]
        return n.startswith("void crt::k_segment<") and n.split(",")[1].strip() == "false"

    def is_frame_end(r):      # one dispatch per multi-segment frame: the deferred shadow rays' launch, the fold
        n = r["Kernel_Name"].split("(")[0]
        return n.startswith("void crt::k_shadow_deferred<false>") or n.startswith("crt::k_fold_paths")
    if tail:
        ids = sorted({int(r["Dispatch_Id"]) for r in rows if is_seg(r)})
        drop = set(ids[-tail:])
        if drop:
            # ... and what those tail frames launched behind their segments
            drop |= {int(r["Dispatch_Id"]) for r in rows if is_frame_end(r) and int(r["Dispatch_Id"]) > min(drop)}
        rows = [r for r in rows if not ((is_seg(r) or is_frame_end(r)) and int(r["Dispatch_Id"]) in drop)]
    agg = collections.defaultdict(list)
    for r in rows:
        agg[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
    return agg


def entry_from_dirs(dirs, wl, tail=0):
    """dirs: {"fetch" | "write" | "sq": directory of that --pmc pass}; wl = "<workload>_d<depth>"; tail: see per_kernel.  None if no pass
    saw the segment kernels."""
    agg = {}
    for d in dirs.values():
        agg.update(per_kernel(d, tail))
    names = sorted({k for k, _ in agg if k.startswith("void crt::k_segment<") and k.split(",")[1].strip() == "false"})
    if not names:
        return None

    extra = sorted({k for k, _ in agg if k.startswith("void crt::k_shadow_deferred<false>") or k.startswith("crt::k_fold_paths")})

    def total(c):       # summed over the segment kernels (first + bounce) and what a multi-segment frame launches behind them (the deferred
        # shadow rays' walk, the fold), per SEGMENT launch — the unit bench.py's launch time of such a step is in (step time / segments)
        vals = [v for k in names for v in agg.get((k, c), [])]
        more = [v for k in extra for v in agg.get((k, c), [])]
        return sum(vals) + sum(more), len(vals)
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
    if n_sq and n_g and act > 0 and insts > 0:
        # Issue slots: a SIMD-32 needs 2 cycles for a wave64 VALU instruction (MI355X_MICROARCH.md), so the share of the chip's issue
        # slots a launch used is 2 x wave-instructions / (SIMDs x shader cycles).  (Round 2 divided 4 x SQ_ACTIVE_INST_VALU by the SIMD
        # cycles and got 1.19: that counter adds up the quad-cycles during which EACH wave has a vector instruction in flight, and two
        # waves of a SIMD overlap theirs, so it is not bounded by 1.)  GRBM_GUI_ACTIVE is summed over the 8 XCDs.
        cycles = (gui / n_g) / 8.0
        per_launch = insts / n_sq
        busy = 2.0 * per_launch / (N_SIMD * cycles)
        lane = thr / (64.0 * act)
        e["valu_issue"] = {"busy": round(busy, 3), "busy_vs_mix_floor": round(busy * 3.3 / 2.0, 3), "lane_util": round(lane, 3), "frac": round(busy * lane, 3),
                           "cycles_per_instr_per_simd": round(N_SIMD * cycles / per_launch, 2),
                           "valu_instructions_per_launch": int(per_launch), "valu_instructions_per_wave": round(insts / max(1.0, waves), 1),
                           "shader_cycles_per_launch": int(cycles),
                           "source": "rocprofv3 --pmc passes of `bench.py --workload %s --depth %s`; formulae in tools/pmc_traffic.py"
                                     % tuple(wl.split("_d"))}
    return e


# three counter groups, one rocprofv3 pass each (MI355X_MICROARCH.md, "rocprofv3 PMC slots": FETCH_SIZE takes 3 of the 4 TCC slots and
# WRITE_SIZE 2, so they cannot share a pass; the two GRBM counters are independent of SQ / TCC and ride with the fetch pass)
PASSES = {"fetch": ["FETCH_SIZE", "TCC_HIT_sum", "GRBM_GUI_ACTIVE", "GRBM_TA_BUSY"], "write": ["WRITE_SIZE", "TCC_MISS_sum"],
          "sq": ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY"]}


def main(round_dir, out, steps=5):
    res = {}
    for wl, spp in (("cornell_d1", 1), ("mesh1m_d1", 4), ("mesh1m_d4", 4), ("mesh520_d1", 4), ("mesh520_d4", 4)):
        depth = int(wl.split("_d")[1])
        e = entry_from_dirs({kind: os.path.join(round_dir, f"pmc_{kind}_{wl}") for kind in PASSES}, wl, tail=min(10, steps * spp) * depth)
        if e:
            e["samples_per_launch"] = spp
            res[wl] = e
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
