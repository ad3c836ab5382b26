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
            counts["I_node_uniform_any"] = f["uniform_node_steps"][-1]        # the in-place shadow walk's (the plain per-lane loop)
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


NODE_BYTES, TRI_BYTES, FB_BYTES = 80, 52, 24        # bytes per node fetch, per triangle test, per pixel-sample (sum read + write)


def traversal_wave_instr(cs, isa, executed=False):
    """Traversal wave-instructions (node visits and triangle tests only, 64-lane equivalents) of the work the counters `cs` describe
    (bench.py: ONE STEP — every sample and segment of it — counted in the form the timed launches have).
    executed = False: the ALGORITHMIC count — every node visit priced at the general 8-wide test, I_node (what SURVEY 8d's bytes stand for);
    executed = True: uniform node steps at what they really execute (I_node_uniform in closest-hit walks, I_node_uniform_any in shadow
    walks): the figure the hardware counters bound from above."""
    n_c, n_a = cs["nodes_closest"], cs["nodes_any"]
    u_c, u_a = (cs.get("nodes_closest_uniform", 0), cs.get("nodes_any_uniform", 0)) if executed else (0, 0)
    i_n = isa["I_node"]
    lane_instr = (n_c - u_c + n_a - u_a) * i_n + u_c * isa.get("I_node_uniform", i_n) + u_a * isa.get("I_node_uniform_any", isa.get("I_node_uniform", i_n)) \
        + (cs["tris_closest"] + cs["tris_any"]) * isa["I_tri"]
    return lane_instr / 64.0


def algorithmic_bytes(cs, node_bytes=NODE_BYTES):
    """80 B per node fetch + 52 B per triangle test + 24 B of sum-buffer traffic per pixel-sample, over the work `cs` describes"""
    return node_bytes * (cs["nodes_closest"] + cs["nodes_any"]) + TRI_BYTES * (cs["tris_closest"] + cs["tris_any"]) + FB_BYTES * cs["primary_rays"]


def shell_static_wave_instr(cs, isa):
    """The per-ray shell as a STATIC instruction count (both sides of every branch): an upper bound of what it executes, for reference only."""
    n_first = cs["primary_rays"]
    n_bounce = cs["closest_rays"] - n_first
    lane_instr = n_first * isa["I_ray_first"] + n_bounce * isa.get("I_ray_bounce", isa["I_ray_first"]) + cs["closest_hits"] * isa["I_shade"]
    return lane_instr / 64.0


def counter_figures(pmc, samples_per_launch):
    """issue_busy, lane_util, their product and the executed lane-full wave-instructions per launch, from an entry of pmc_traffic.json /
    a live pass (tools/pmc_traffic.py `valu_issue`), scaled to this block's samples per launch.  {} without counters."""
    vi = (pmc or {}).get("valu_issue") or {}
    if not vi.get("busy") or not vi.get("lane_util"):
        return {}
    scale = samples_per_launch / max(1, (pmc or {}).get("samples_per_launch", 1))
    return {"issue_busy": vi["busy"], "lane_util": vi["lane_util"], "counter_frac": round(vi["busy"] * vi["lane_util"], 4),
            "valu_instr_per_launch": int(vi.get("valu_instructions_per_launch", 0) * scale)}


def roofline_block(cs, isa, launch_ms, launches, pmc=None, samples_per_launch=1):
    """The VALU-issue roofline of a block.  cs: counters of one STEP; launches: segment launches the step is made of (its path segments);
    launch_ms: mean duration of one of them.  frac = algorithmic traversal wave-instructions per launch / launch time / peak — moves with
    the speed; frac_executed = the same with uniform node steps at their executed cost — bounded by the counters' issue_busy x lane_util."""
    launches = max(1, launches)
    t = launch_ms * 1e-3
    w_alg = traversal_wave_instr(cs, isa) / launches
    w_exe = traversal_wave_instr(cs, isa, executed=True) / launches
    achieved = w_alg / t / 1e9 if t > 0 else 0.0
    r = {"achieved": round(achieved, 1), "peak": PEAK_GINSTR, "frac": round(achieved / PEAK_GINSTR, 4),
         "frac_executed": round(w_exe / t / 1e9 / PEAK_GINSTR, 4) if t > 0 else 0.0,
         "attainable": round(ATTAINABLE_GINSTR, 1),
         "traversal_wave_instr_per_launch": int(w_alg), "executed_traversal_wave_instr_per_launch": int(w_exe),
         "shell_static_wave_instr_per_launch": int(shell_static_wave_instr(cs, isa) / launches)}
    c = counter_figures(pmc, samples_per_launch)
    if c:
        r.update({"issue_busy": c["issue_busy"], "lane_util": c["lane_util"], "counter_frac": c["counter_frac"]})
        if c["valu_instr_per_launch"]:
            # executed lane-full wave-instructions of the launch (everything the lanes did) against the traversal's share of them
            r["non_traversal_share"] = round(1.0 - w_exe / (c["valu_instr_per_launch"] * c["lane_util"]), 4)
    return r


def recompute_line_block(r, isa):
    """Every derived figure of a bench-line block's `roofline` object from its own counters, launch time and isa_counts.json:
    {"frac", "frac_executed", "achieved", "algorithmic_bytes_per_launch", "algorithmic_gbps", "algorithmic_over_peak", "hbm_frac", "counter_frac"}"""
    cs, launches, t = r["counters"], max(1, r.get("path_segments", 1)), r["launch_ms"] * 1e-3
    got = roofline_block(cs, isa, r["launch_ms"], launches)
    out = {"frac": got["frac"], "frac_executed": got["frac_executed"], "achieved": got["achieved"]}
    nb = 96 if r.get("accel") == "bvh2" else NODE_BYTES
    out["algorithmic_bytes_per_launch"] = int(algorithmic_bytes(cs, nb) / launches)
    out["algorithmic_gbps"] = round(out["algorithmic_bytes_per_launch"] / t / 1e9, 1)
    out["algorithmic_over_peak"] = round(out["algorithmic_gbps"] / 8000.0, 4)
    if r.get("traffic"):
        out["hbm_frac"] = round(r["traffic"] / t / 1e9 / 8000.0, 4)
    if r.get("issue_busy") and r.get("lane_util"):
        out["counter_frac"] = round(r["issue_busy"] * r["lane_util"], 4)
    return out


def cmd_frac(bench_json, stats_csv=None):
    isa = json.load(open(os.path.join(ROOT, "profiles", "isa_counts.json")))
    line = [l for l in open(bench_json) if l.startswith("{")][-1]
    d = json.loads(line)
    avg_us = {}
    if stats_csv:
        for r in csv.DictReader(open(stats_csv)):
            avg_us[r["Name"]] = float(r["AverageNs"]) / 1e3
    r = d.get("roofline") or {}
    if r.get("counters"):
        got = recompute_line_block(r, isa)
        for k, v in got.items():
            print(f"  {k:32s} recomputed {v}   line says {r.get(k)}")
        if "counter_frac" in got:
            print(f"  frac_executed <= issue_busy x lane_util: {'ok' if got['frac_executed'] <= got['counter_frac'] + 1e-3 else 'VIOLATED'}")
    for name, e in (d.get("extras") or {}).items():
        print(f"  {name:22s} {e.get('value')} Mray/s  launch {e.get('launch_ms')} ms  frac {e.get('frac')}  frac_executed {e.get('frac_executed')}  counter_frac {e.get('counter_frac')}")
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
