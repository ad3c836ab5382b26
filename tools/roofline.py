#!/usr/bin/env python3
"""The binding roofline of the segment kernels: VALU issue, and everything needed to recompute it from committed files.

  tools/roofline.py isa     [asm] [remarks] [tag]   -> profiles/isa_counts.json, profiles/<tag>_kernel_resources.txt
  tools/roofline.py frac    bench.json [kernel_stats.csv]   -> every block's roofline recomputed from the line's own counters
                                                               (and, with a CSV, from rocprofv3's average duration of the kernel)

Why VALU issue and not HBM.  SURVEY 8d prices a ray at 80 B per node + 52 B per triangle test and divides by 8 TB/s.  Every BASELINE
scene (<= 110 MB) lives in L1 / L2 / the 256 MiB Infinity Cache, so those bytes never reach HBM and that quotient exceeds 1: it is
kept as `algorithmic_gbps`, never as a fraction; the bytes that do cross L2 <-> fabric are `traffic`, and `hbm_frac` = traffic / time /
8 TB/s.  The kernel is bound by vector-instruction issue, so the roof is

    peak      = 1024 SIMDs x 2.4 GHz / 2 cycles            = 1228.8 G wave-instructions/s
                (MI355X_MICROARCH.md: SIMD-32, a wave64 VALU instruction occupies the ALU for 2 cycles; the same figure as the
                157.3 TFLOP/s fp32 vector peak)
    achieved  = TRAVERSAL wave-instructions of a launch / its duration
              = (N_node x I_node + N_tri x I_tri) / 64 / t
                N_*  : node visits and triangle tests of the launch (the counting kernels; equal to the CPU oracle's counters on the
                       same frame, asserted by the GPU tests and by bench.py's cpu_baseline leg)
                I_*  : vector instructions ONE lane executes for one node visit (the 8-wide box test and the mask assembly) / one
                       Moller-Trumbore test: straight-line blocks, counted in the kernel's ISA between the CRT_MARK lines of a
                       `make asm` build (profiles/isa_counts.json)
    frac      = achieved / peak

i.e. the share of the chip's vector lane-cycles spent inside node and triangle tests — the work SURVEY 8d's algorithmic bytes stand
for — as 64-lane equivalents.  Every one of these instructions really was issued with those lanes enabled, so the hardware counters of
the same run bound it from above:

    counter_frac = issue_busy x lane_util      issue_busy = 2 x SQ_INSTS_VALU / (1024 SIMDs x shader cycles)
                                               lane_util  = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU)
    frac <= counter_frac                       (asserted on the committed line by tests/test_tools.py)
    non_traversal_share = 1 - frac / counter_frac

`non_traversal_share` is everything the lanes executed that is not a node or triangle test: the ray's shell (ray generation or queue
fetch, shading, NEE set-up, bounce sampling, queue emission) and the traversal loop's own bookkeeping (votes, stack, pops).  Round 3
added a STATIC count of the shell to `achieved` (both sides of every branch): on the Cornell box that claimed more instructions than
the launch issued altogether (frac 0.82 against a counter bound of 0.44) — the shell's executed cost is not a static count, so it is
no longer part of `frac`; its static size stays on the line as `shell_static_wave_instr_per_launch` for reference.  `attainable` is the
roof with the measured issue cost of this kernel's instruction mix (tools/ubench/valu_issue_cycles.hip: 3.3 cycles per wave-instruction
per SIMD with 4 waves resident, against 2.5-2.7 for plain v_fma_f32 and the 2.0 of the datasheet) instead of 2 cycles.
"""
import collections
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_SIMD = 1024
CLOCK_GHZ = 2.4
SPEC_CYCLES_PER_INSTR = 2.0
PEAK_GINSTR = N_SIMD * CLOCK_GHZ / SPEC_CYCLES_PER_INSTR          # 1228.8 G wave-instructions/s
MIX_CYCLES_PER_INSTR = 3.3                                         # profiles/r03_valu_issue_cycles.txt, node-test mix, 4 waves per SIMD
ATTAINABLE_GINSTR = N_SIMD * CLOCK_GHZ / MIX_CYCLES_PER_INSTR


def demangle_args(name):
    """k_segment<...> template arguments of a mangled name: _ZN3crt9k_segmentILb1ELb0E...EEvNS_11SegmentArgsE -> [1,0,...]"""
    m = re.search(r"k_segmentI((?:Lb[01]E)+)E", name)
    return [int(x) for x in re.findall(r"Lb([01])E", m.group(1))] if m else None


SEG_PARAMS = ["FIRST", "STATS", "TEX", "PRETRACED", "INPLACE", "BVH2", "MAT", "BATCH", "WIDE", "ONE"]


def label_of(name):
    a = demangle_args(name)
    if a is None:
        m = re.match(r"_ZN3crt\d+(k_[a-z0-9_]+)", name)
        return m.group(1) if m else name
    on = [p for p, v in zip(SEG_PARAMS, a) if v]
    return "k_segment<" + ",".join(on) + ">"


def parse_asm(path):
    """{kernel: {"regions": [(kind, tag, n_valu)], "valu_total": n, "valu_outside_loops": n, ...}} from an assembly with CRT_MARK lines.

    A `<kind>_begin` marker sits at the top of the source block `if (lane takes part) { ... }`; the compiler guards that block with
    `s_cbranch_execz <join>` just before the marker, and everything up to `<join>:` runs under the block's lane mask.  The region
    counted for a node / triangle step is therefore [that branch, its join label) — the `_end` markers are not used for it: an
    inline-asm comment has no operands, and the scheduler is free to hoist it over the arithmetic that precedes it (it does).
    Loops are the text between `loop_begin` and the matching `loop_end`, which do stay in place (they sit on control-flow edges)."""
    kernels, cur_name, lines = {}, None, []
    for line in open(path, errors="replace"):
        s = line.strip()
        m = re.match(r"^(_ZN3crt[A-Za-z0-9_]+):", s)
        if m:
            cur_name, lines = m.group(1), []
            kernels[cur_name] = lines
            continue
        if cur_name is None:
            continue
        if s.startswith(".end_amdhsa_kernel") or s.startswith(".Lfunc_end"):
            cur_name = None
            continue
        lines.append(s)

    def is_instr(s):
        return bool(s) and not s.startswith((";", ".", "//")) and not s.endswith(":") and not re.match(r"^\.?L?BB\d+_\d+:", s)

    out = {}
    for name, lines in kernels.items():
        e = {"regions": [], "valu_total": 0, "valu_outside_loops": 0, "salu_total": 0, "vmem_total": 0, "lds_total": 0}
        label_at = {}
        for i, s in enumerate(lines):
            m = re.match(r"^(\.LBB\d+_\d+):", s)
            if m:
                label_at[m.group(1)] = i
        in_loop = [False] * len(lines)
        open_loops = []
        for i, s in enumerate(lines):
            m = re.match(r"^; CRT_MARK loop_(begin|end)\s*(\w*)", s)
            if m and m.group(1) == "begin":
                open_loops.append((i, m.group(2)))
            elif m and open_loops:
                # an any-hit loop has two exits (found a hit / exhausted): the first end closes it, a second one finds nothing open
                b, tag = open_loops.pop()
                n = sum(1 for t in lines[b:i] if is_instr(t) and t.split()[0].startswith("v_"))
                e["regions"].append(("loop", tag, n))
                for j in range(b, i):
                    in_loop[j] = True
        # uniform node steps (the node through the scalar cache): the text between its two markers — both volatile, with the volatile
        # s_load block right behind the first — is the variant's own arithmetic; it sits INSIDE a node region and is counted apart
        uni = []
        for i, s in enumerate(lines):
            if re.match(r"^; CRT_MARK uninode_begin", s):
                for q in range(i, len(lines)):
                    if re.match(r"^; CRT_MARK uninode_end", lines[q]):
                        uni.append((i, q, sum(1 for t in lines[i:q] if is_instr(t) and t.split()[0].startswith("v_"))))
                        break
        for i, s in enumerate(lines):
            m = re.match(r"^; CRT_MARK (node|tri|share|shade)_begin", s)
            if m:
                for j in range(i - 1, max(-1, i - 40), -1):
                    mb = re.match(r"^s_cbranch_execz (\.LBB\d+_\d+)", lines[j])
                    if mb and label_at.get(mb.group(1), -1) > i:
                        # the shading block contains the in-place shadow walk: its loops are counted as loops, not as shading
                        n = sum(1 for q in range(j, label_at[mb.group(1)]) if is_instr(lines[q]) and lines[q].split()[0].startswith("v_")
                                and not (m.group(1) == "shade" and in_loop[q]))
                        if m.group(1) == "node":
                            for (ub, ue, un) in uni:
                                if j <= ub and ue <= label_at[mb.group(1)]:
                                    n -= un
                                    e["regions"].append(("uninode", "", un))
                        e["regions"].append((m.group(1), "", n))
                        break
            if not is_instr(s):
                continue
            op = s.split()[0]
            if op.startswith("v_"):
                e["valu_total"] += 1
                if not in_loop[i]:
                    e["valu_outside_loops"] += 1
            elif op.startswith("s_"):
                e["salu_total"] += 1
            elif op.startswith(("global_", "buffer_", "scratch_", "flat_")):
                e["vmem_total"] += 1
            elif op.startswith("ds_"):
                e["lds_total"] += 1
        # regions in order of appearance: sort by nothing (appended loop regions first): keep kinds separately ordered
        out[name] = e
    return out


def parse_remarks(path):
    """{kernel: {VGPRs, ScratchSize, Occupancy, VGPRs Spill, SGPRs Spill, LDS Size, TotalSGPRs}} from -Rpass-analysis=kernel-resource-usage"""
    res, cur = collections.OrderedDict(), None
    for line in open(path, errors="replace"):
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = res.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+) \[-Rpass", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = m.group(2)
    return res


def first_region(regs, kind, nth=0):
    xs = [r for r in regs if r[0] == kind]
    return xs[nth][2] if len(xs) > nth else None


def cmd_isa(asm=None, remarks=None, tag="r03"):
    csrc = os.path.join(ROOT, "caitlynrenderer_amd", "csrc")
    asm = asm or os.path.join(csrc, "rt_kernels.s")
    remarks = remarks or os.path.join(csrc, "rt_kernels.remarks")
    k = parse_asm(asm)
    # the kernels the bench line is made of
    want = {"first": [1, 0, 0, 0, 1, 0, 0, 1, 1, 1],     # <FIRST, INPLACE, BATCH, WIDE, ONE>: 4 samples per launch in the lanes of a wave, Lambert
            "first_single": [1, 0, 0, 0, 1, 0, 0, 0, 0, 0],
            "bounce_plain": [0, 0, 0, 0, 1, 0, 0, 0, 0, 0],   # <INPLACE>: closest hit + in-place shadow walk
            "bounce_deferred": [0, 0, 0, 0, 0, 0, 0, 0, 0, 0]}    # <>: closest hit only, shadow rays left to k_shadow_deferred
    counts = {"source": "VALU instructions between the CRT_MARK lines of `make -C caitlynrenderer_amd/csrc asm` (rt_kernels.s, -DCRT_ISA_MARKS); "
                        "regions in order of appearance: closest-hit loop first, in-place any-hit loop second",
              "peak_gwave_instr_per_s": PEAK_GINSTR, "attainable_gwave_instr_per_s": round(ATTAINABLE_GINSTR, 1),
              "spec_cycles_per_instr": SPEC_CYCLES_PER_INSTR, "mix_cycles_per_instr": MIX_CYCLES_PER_INSTR, "kernels": {}}
    for key, args in want.items():
        name = next((n for n in k if demangle_args(n) == args), None)
        if not name:
            continue
        e = k[name]
        regs = e["regions"]
        loops = [r for r in regs if r[0] == "loop"]
        d = {"kernel": label_of(name), "valu_total": e["valu_total"], "valu_outside_loops": e["valu_outside_loops"],
             "salu_total": e["salu_total"], "vmem_total": e["vmem_total"], "lds_total": e["lds_total"],
             "node_steps": [r[2] for r in regs if r[0] == "node"], "tri_steps": [r[2] for r in regs if r[0] == "tri"],
             "uniform_node_steps": [r[2] for r in regs if r[0] == "uninode"],
             "shared_tri_steps": [r[2] for r in regs if r[0] == "share"], "loops": [(r[1], r[2]) for r in loops],
             "shade": [r[2] for r in regs if r[0] == "shade"]}
        counts["kernels"][key] = d
    # the batched first-segment kernel is the one the bench times (and the one that carries the uniform node steps)
    f = counts["kernels"].get("first") or counts["kernels"].get("first_single")
    b = counts["kernels"].get("bounce_plain") or counts["kernels"].get("bounce")
    if f:
        # one node visit / one triangle test: the closest-hit voting loop of the first-segment kernel (the plain loop of the in-place
        # shadow walk is within a few instructions of it); the shell of a ray = everything outside the traversal loops
        counts["I_node"] = f["node_steps"][0]
        counts["I_tri"] = f["tri_steps"][0]
        # a uniform node step (every enabled lane fetches the same node: scalar loads, scalar decode): what such a visit executes instead
        if f.get("uniform_node_steps"):
            counts["I_node_uniform"] = f["uniform_node_steps"][0]
        # a ray's shell: what every ray runs (ray generation or queue fetch, loop set-up, queue emission) and what only a ray that
        # hit something runs (shading, NEE set-up, bounce sampling)
        counts["I_shade"] = f["shade"][0] if f["shade"] else 0
        counts["I_ray_first"] = f["valu_outside_loops"] - counts["I_shade"]
    if b:
        counts["I_shade_bounce"] = b["shade"][0] if b["shade"] else counts.get("I_shade", 0)
        counts["I_ray_bounce"] = b["valu_outside_loops"] - counts["I_shade_bounce"]
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    json.dump(counts, open(os.path.join(ROOT, "profiles", "isa_counts.json"), "w"), indent=1)
    print(json.dumps({x: counts.get(x) for x in ("I_node", "I_node_uniform", "I_tri", "I_ray_first", "I_shade", "I_ray_bounce", "I_shade_bounce")}))
    if os.path.exists(remarks):
        r = parse_remarks(remarks)
        path = os.path.join(ROOT, "profiles", f"{tag}_kernel_resources.txt")
        with open(path, "w") as fo:
            fo.write("# per-kernel resources of the product build (hipcc -Rpass-analysis=kernel-resource-usage, `make asm`), regenerated by tools/roofline.py isa\n")
            fo.write(f"{'kernel':62s} {'VGPR':>5s} {'SGPR':>5s} {'scratch B/lane':>15s} {'VGPR spills':>12s} {'SGPR spills':>12s} {'occupancy':>10s} {'static LDS':>11s}\n")
            for name, e in r.items():
                fo.write(f"{label_of(name):62s} {e.get('VGPRs', '?'):>5s} {e.get('TotalSGPRs', '?'):>5s} {e.get('ScratchSize', '?'):>15s} "
                         f"{e.get('VGPRs Spill', '?'):>12s} {e.get('SGPRs Spill', '?'):>12s} {e.get('Occupancy', '?'):>10s} {e.get('LDS Size', '?'):>11s}\n")
        print("wrote", path)


def traversal_wave_instr(cs, isa, depth=1, samples=1):
    """Traversal wave-instructions of ONE launch (node visits and triangle tests only, 64-lane equivalents): the frame's total over its
    `depth` segment launches / depth, x samples per launch.  cs: totals of one counting frame."""
    # node visits of uniform node steps (the node through the scalar cache, its decode on the scalar unit) execute I_node_uniform
    # vector instructions, the others I_node: executed counts, so that the figure cannot exceed what the counters saw
    n_uni = cs.get("nodes_closest_uniform", 0) + cs.get("nodes_any_uniform", 0)
    n_nodes = cs["nodes_closest"] + cs["nodes_any"]
    lane_instr = (n_nodes - n_uni) * isa["I_node"] + n_uni * isa.get("I_node_uniform", isa["I_node"]) + (cs["tris_closest"] + cs["tris_any"]) * isa["I_tri"]
    return lane_instr / 64.0 / max(1, depth) * samples


def shell_static_wave_instr(cs, isa, depth=1, samples=1):
    """The per-ray shell as a STATIC instruction count (both sides of every branch): an upper bound of what it executes, for reference only."""
    n_first = cs["primary_rays"]
    n_bounce = cs["closest_rays"] - n_first
    lane_instr = n_first * isa["I_ray_first"] + n_bounce * isa.get("I_ray_bounce", isa["I_ray_first"]) + cs["closest_hits"] * isa["I_shade"]
    return lane_instr / 64.0 / max(1, depth) * samples


def counter_figures(pmc, samples_per_launch):
    """issue_busy, lane_util, their product and the executed lane-full wave-instructions per launch, from an entry of pmc_traffic.json /
    a live pass (tools/pmc_traffic.py `valu_issue`), scaled to this block's samples per launch.  {} without counters."""
    vi = (pmc or {}).get("valu_issue") or {}
    if not vi.get("busy") or not vi.get("lane_util"):
        return {}
    scale = samples_per_launch / max(1, (pmc or {}).get("samples_per_launch", 1))
    return {"issue_busy": vi["busy"], "lane_util": vi["lane_util"], "counter_frac": round(vi["busy"] * vi["lane_util"], 4),
            "valu_instr_per_launch": int(vi.get("valu_instructions_per_launch", 0) * scale)}


def roofline_block(cs, isa, launch_ms, depth, samples, pmc=None):
    w = traversal_wave_instr(cs, isa, depth, samples)
    achieved = w / (launch_ms * 1e-3) / 1e9 if launch_ms > 0 else 0.0
    # the same visits priced at the GENERAL node step (what the walk would execute without the uniform steps): the uniform steps lower
    # the executed instructions per visit, so `frac` (executed) falls while the rays per second rise; this figure moves with the speed
    cs_general = dict(cs, nodes_closest_uniform=0, nodes_any_uniform=0)
    w_general = traversal_wave_instr(cs_general, isa, depth, samples)
    r = {"achieved": round(achieved, 1), "peak": PEAK_GINSTR, "frac": round(achieved / PEAK_GINSTR, 4),
         "frac_at_general_step": round(w_general / (launch_ms * 1e-3) / 1e9 / PEAK_GINSTR, 4) if launch_ms > 0 else 0.0,
         "attainable": round(ATTAINABLE_GINSTR, 1), "frac_of_attainable": round(achieved / ATTAINABLE_GINSTR, 4),
         "traversal_wave_instr_per_launch": int(w), "shell_static_wave_instr_per_launch": int(shell_static_wave_instr(cs, isa, depth, samples))}
    c = counter_figures(pmc, samples)
    if c:
        r.update({"issue_busy": c["issue_busy"], "lane_util": c["lane_util"], "counter_frac": c["counter_frac"]})
        if c["valu_instr_per_launch"]:
            # executed lane-full wave-instructions of the launch (everything the lanes did) against the traversal's share of them
            r["non_traversal_share"] = round(1.0 - w / (c["valu_instr_per_launch"] * c["lane_util"]), 4)
    return r


def cmd_frac(bench_json, stats_csv=None):
    isa = json.load(open(os.path.join(ROOT, "profiles", "isa_counts.json")))
    line = [l for l in open(bench_json) if l.startswith("{")][-1]
    d = json.loads(line)
    blocks = {"top": d}
    blocks.update({k: v for k, v in d.items() if isinstance(v, dict) and "roofline" in v})
    avg_us = {}
    if stats_csv:
        for r in csv.DictReader(open(stats_csv)):
            avg_us[r["Name"]] = float(r["AverageNs"]) / 1e3
    for name, b in blocks.items():
        r = b.get("roofline") or {}
        cs = r.get("counters")
        if not cs:
            continue
        pmc = {"valu_issue": {"busy": r.get("issue_busy"), "lane_util": r.get("lane_util")}} if r.get("issue_busy") else None
        got = roofline_block(cs, isa, r["launch_ms"], r.get("path_segments", 1), r.get("samples_per_launch", 1), pmc)
        bound = got.get("counter_frac")
        print(f"{name:22s} launch {r['launch_ms']:.4f} ms  frac {got['frac']:.4f} (line says {r.get('frac')})"
              + (f"  <= issue_busy x lane_util {bound:.4f}: {'ok' if got['frac'] <= bound else 'VIOLATED'}" if bound else "  (no counters on this block)")
              + f"  achieved {got['achieved']} G wave-instr/s  algorithmic GB/s {r.get('algorithmic_gbps')}")
    if avg_us:
        for n, us in sorted(avg_us.items(), key=lambda kv: -kv[1])[:6]:
            print(f"  rocprofv3 average {us:10.1f} us  {n[:110]}")


if __name__ == "__main__":
    if len(sys.argv) < 2 or sys.argv[1] not in ("isa", "frac"):
        print(__doc__)
        sys.exit(1)
    if sys.argv[1] == "isa":
        cmd_isa(*sys.argv[2:])
    else:
        cmd_frac(*sys.argv[2:])
