"""The measurement tooling behind the bench line's `roofline` object, without a GPU: the ISA counter of tools/roofline.py on a small
hand-written assembly, the instruction model against the committed bench line, and the counter summary's rule that the single-sample
tail frames of a profiled bench run are not averaged into the per-launch figures (tools/pmc_traffic.py)."""
import csv
import importlib.util
import json
import os

from conftest import ROOT


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tools", name + ".py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


ASM = """
_ZN3crt9k_segmentILb1ELb0ELb0ELb0ELb1ELb0ELb0ELb0ELb0ELb0EEEvNS_11SegmentArgsE:
	s_load_dwordx2 s[0:1], s[4:5], 0x0
	v_mov_b32_e32 v0, 0
	v_add_f32_e32 v1, v0, v0
	; CRT_MARK loop_begin voting
.LBB0_1:
	v_cmp_eq_u32_e32 vcc, 0, v0
	s_and_saveexec_b64 s[2:3], vcc
	s_cbranch_execz .LBB0_3
	; CRT_MARK node_begin
	global_load_dwordx4 v[4:7], v8, s[0:1]
	v_cvt_f32_ubyte0_e32 v2, v4
	v_fma_f32 v2, v2, v1, v1
	v_max3_f32 v2, v2, v1, v0
.LBB0_3:
	s_or_b64 exec, exec, s[2:3]
	v_cmp_eq_u32_e32 vcc, 1, v0
	s_and_saveexec_b64 s[2:3], vcc
	s_cbranch_execz .LBB0_5
	; CRT_MARK tri_begin
	v_mul_f32_e32 v3, v1, v1
	v_sub_f32_e32 v3, v3, v1
.LBB0_5:
	s_or_b64 exec, exec, s[2:3]
	s_cbranch_scc1 .LBB0_1
	; CRT_MARK loop_end
	v_cmp_lt_f32_e32 vcc, 0, v3
	s_and_saveexec_b64 s[2:3], vcc
	s_cbranch_execz .LBB0_7
	; CRT_MARK shade_begin
	v_mul_f32_e32 v9, v3, v3
	v_add_f32_e32 v9, v9, v3
	v_add_f32_e32 v9, v9, v3
.LBB0_7:
	s_endpgm
	.end_amdhsa_kernel
"""


def test_isa_regions_are_counted_between_the_guarding_branch_and_its_join_label(tmp_path):
    r = _load("roofline")
    p = tmp_path / "k.s"
    p.write_text(ASM)
    k = r.parse_asm(str(p))
    (name, e), = k.items()
    assert r.demangle_args(name) == [1, 0, 0, 0, 1, 0, 0, 0, 0, 0] and r.label_of(name) == "k_segment<FIRST,INPLACE>"
    regs = {kind: n for kind, _, n in e["regions"]}
    assert regs["node"] == 3 and regs["tri"] == 2 and regs["shade"] == 3            # vector instructions under each block's lane mask
    assert regs["loop"] == 7                                                          # 2 compares + 3 + 2 inside the loop
    assert e["valu_total"] == 13 and e["valu_outside_loops"] == 6 and e["vmem_total"] == 1


def test_the_committed_bench_line_is_recomputable_and_below_its_counter_bound():
    """profiles/r05_bench_default.json (the line bench.py printed on the MI355X box): EVERY derived figure of the headline's roofline object
    follows from the line's own counters, launch time and profiles/isa_counts.json — frac (algorithmic), frac_executed, the algorithmic
    bytes with SURVEY 8d's 24 B per pixel-sample, their rate and its ratio to the HBM peak, hbm_frac, counter_frac — and on every block
    that carries hardware counters frac_executed <= issue_busy x lane_util: useful work cannot exceed executed work."""
    import pytest
    r = _load("roofline")
    isa = json.load(open(os.path.join(ROOT, "profiles", "isa_counts.json")))
    path = os.path.join(ROOT, "profiles", "r05_bench_default.json")
    if not os.path.exists(path):
        pytest.skip("no round-5 bench line committed yet")
    line = [l for l in open(path) if l.startswith("{")][-1]
    assert len(line) < 9000
    d = json.loads(line)
    ro = d["roofline"]
    got = r.recompute_line_block(ro, isa)
    for k, v in got.items():
        assert ro.get(k) is not None and abs(v - ro[k]) <= 2e-3 * max(1.0, abs(v)), (k, v, ro.get(k))
    cs = ro["counters"]
    assert ro["algorithmic_bytes_per_launch"] == 80 * (cs["nodes_closest"] + cs["nodes_any"]) + 52 * (cs["tris_closest"] + cs["tris_any"]) + 24 * cs["primary_rays"]
    assert cs["primary_rays"] == 4 * 1920 * 1080 and ro["bound"] == "valu_issue" and ro["peak"] == r.PEAK_GINSTR and "HBM roof does not bind" in ro["note"]
    assert ro["algorithmic_over_peak"] > 1.0 > ro["hbm_frac"] > 0                      # the bytes come from the caches
    assert ro["frac_executed"] <= ro["frac"] and ro["frac_executed"] <= ro["counter_frac"] + 1e-3
    assert "1004672 tris" in d["config"]["workload"] and d["config"]["spp_per_step"] == 4
    assert d["config"]["launch"] == {"form": 2, "wide": True, "one_pass": True, "samples": 4, "shards": 1} and d["sum_rows_match_oracle"] is True
    sp = d["step_ms_spread"]
    assert sp["n"] == d["steps"] and sp["min"] <= sp["median"] <= sp["max"] and abs(sp["median"] - d["ms_per_step"]) < 0.1 * d["ms_per_step"]
    with_counters = 0
    for name, e in [("headline", ro)] + [(k, v) for k, v in d["extras"].items() if "frac" in v]:
        if e["frac"] is None:
            continue                                                                  # the BVH2 walk has no instruction model
        assert 0 < e["frac"] <= 1.0, name                                             # no block above its roof
        if e.get("issue_busy") and e.get("lane_util"):
            with_counters += 1
            assert abs(e["counter_frac"] - e["issue_busy"] * e["lane_util"]) < 2e-3, name
            assert e["frac_executed"] <= e["counter_frac"] + 1e-3, (name, e["frac_executed"], e["counter_frac"])
            assert 0 <= e["non_traversal_share"] < 1, name
        if e.get("traffic_gbps"):
            assert abs(e["hbm_frac"] - e["traffic_gbps"] / 8000.0) < 1e-3, name
    assert with_counters >= 3
    for name, e in d["extras"].items():
        assert e.get("sum_rows_match_oracle", e.get("rgba_rows_match_oracle")) in (True, None), name
    assert d["extras"]["hbm_resident"]["scene_mb"] > 256 * 1.048576 and 0 < d["extras"]["hbm_resident_d4"]["hbm_frac"] < 1
    cl = d["reference_claims"]
    assert cl["cwbvh_over_bvh2_mesh1m"] > 1.0 and cl["sbvh_over_sah_bvh2_walk"] > 0.9 and "README.md:21-22" in cl["readme"]
    fl = d["extras"]["frame_loop_mesh1m"]
    assert fl["ms_per_frame_image_in_hbm"] <= fl["ms_per_frame_image_in_host_memory"] and fl["segment_launch_ms"] > 0


def test_roofline_model_is_traversal_only_and_monotone():
    r = _load("roofline")
    isa = {"I_node": 230, "I_node_uniform": 130, "I_node_uniform_any": 120, "I_tri": 80, "I_ray_first": 800, "I_ray_bounce": 700, "I_shade": 480}
    cs = {"primary_rays": 1000, "closest_rays": 1500, "closest_hits": 900, "nodes_closest": 30000, "tris_closest": 4000, "nodes_any": 9000, "tris_any": 1000,
          "nodes_closest_uniform": 10000, "nodes_any_uniform": 2000}
    w = r.traversal_wave_instr(cs, isa)
    assert w == (39000 * 230 + 5000 * 80) / 64.0                                       # algorithmic: every visit at the general step
    we = r.traversal_wave_instr(cs, isa, executed=True)
    assert we == ((39000 - 12000) * 230 + 10000 * 130 + 2000 * 120 + 5000 * 80) / 64.0 and we < w
    assert r.algorithmic_bytes(cs) == 80 * 39000 + 52 * 5000 + 24 * 1000
    pmc = {"valu_issue": {"busy": 0.6, "lane_util": 0.5, "valu_instructions_per_launch": 2 * we}, "samples_per_launch": 4}
    b = r.roofline_block(cs, isa, 0.001, 2, pmc, 4)
    assert b["counter_frac"] == 0.3 and abs(b["non_traversal_share"] - 0.5) < 1e-3       # half of the executed lane-work is node / triangle tests
    assert b["traversal_wave_instr_per_launch"] == int(w / 2) and b["shell_static_wave_instr_per_launch"] > 0 and b["frac_executed"] < b["frac"]
    faster = r.roofline_block(cs, isa, 0.0005, 2, pmc, 4)
    assert faster["frac"] > b["frac"]                                                  # a faster launch never prints a lower fraction
    assert "counter_frac" not in r.roofline_block(cs, isa, 0.001, 2, None)


def test_counter_summary_leaves_the_single_sample_tail_out(tmp_path):
    """bench.py ends a run with single-sample frames (full event timing); under rocprofv3 --pmc those dispatches — a quarter of the
    work, partly the same kernels — must not be averaged with the steps' 4-sample launches."""
    pt = _load("pmc_traffic")
    seg = "void crt::k_segment<true, false, false, false, true, false, false, true, true, true>(crt::SegmentArgs)"
    stats = "void crt::k_segment<true, true, true, false, true, false, true, false, false, false>(crt::SegmentArgs)"
    cols = ["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value"]

    def write(kind, counters):
        d = tmp_path / f"pmc_{kind}"
        (d / "run").mkdir(parents=True)
        with open(d / "run" / "1_counter_collection.csv", "w", newline="") as f:
            w = csv.DictWriter(f, cols)
            w.writeheader()
            disp = 0
            for n, val in [(stats, 999.0)] + [(seg, 400.0)] * 7 + [(seg, 100.0)] * 10:     # counting frame, 7 step launches, 10 tail frames
                disp += 1
                for c, scale in counters.items():
                    w.writerow({"Dispatch_Id": disp, "Kernel_Name": n, "Counter_Name": c, "Counter_Value": val * scale})
        return str(d)
    dirs = {"fetch": write("fetch", {"FETCH_SIZE": 1.0, "TCC_HIT_sum": 3.0, "GRBM_GUI_ACTIVE": 8000.0, "GRBM_TA_BUSY": 1.0}),
            "write": write("write", {"WRITE_SIZE": 0.5, "TCC_MISS_sum": 1.0}),
            "sq": write("sq", {"SQ_WAVES": 1.0, "SQ_INSTS_VALU": 1000.0, "SQ_ACTIVE_INST_VALU": 10.0, "SQ_THREAD_CYCLES_VALU": 320.0})}
    mixed = pt.entry_from_dirs(dirs, "mesh1m_d1")
    steps = pt.entry_from_dirs(dirs, "mesh1m_d1", tail=10)
    assert steps["dispatches"] == 7 and mixed["dispatches"] == 17
    assert steps["l2_fabric_bytes_per_launch"] == int((2 * 400.0 + 200.0) * 1024)       # (2 FETCH_SIZE + WRITE_SIZE) KiB of a step launch
    assert mixed["l2_fabric_bytes_per_launch"] < steps["l2_fabric_bytes_per_launch"]
    assert steps["l2_hit_rate"] == 0.75 and abs(steps["valu_issue"]["lane_util"] - 0.5) < 1e-9
    # issue_busy = 2 x wave-instructions / (SIMDs x shader cycles), bounded by 1 by construction
    assert abs(steps["valu_issue"]["busy"] - 2 * 400e3 / (1024 * 400 * 8000.0 / 8)) < 1e-3


def test_counter_anatomy_reports_unit_shares_without_the_single_sample_tail(tmp_path, capsys):
    """tools/pmc_anatomy.py: cycle counters of the per-CU units as a share of 256 x the dispatch's shader cycles (GRBM_GUI_ACTIVE / 8 XCDs),
    SQ wave-cycle counters as a share of SQ_WAVE_CYCLES, the run's closing single-sample dispatches left out."""
    pa = _load("pmc_anatomy")
    seg = "void crt::k_segment<false, false, false, false, false, false, false, false, false, false>(crt::SegmentArgs)"
    cols = ["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value"]
    passes = [{"GRBM_GUI_ACTIVE": 8e6, "TA_TA_BUSY_sum": 128e6, "TA_TOTAL_WAVEFRONTS_sum": 8e6},
              {"GRBM_GUI_ACTIVE": 8e6, "SQ_WAVE_CYCLES": 1000.0, "SQ_WAIT_ANY": 400.0, "SQ_ACTIVE_INST_ANY": 200.0, "SQ_WAIT_INST_ANY": 400.0},
              {"GRBM_GUI_ACTIVE": 8e6, "TCP_TCC_READ_REQ_sum": 1e6, "TCP_TCC_READ_REQ_LATENCY_sum": 424e6}]
    for i, counters in enumerate(passes, 1):
        d = tmp_path / f"pass{i}" / "run"
        d.mkdir(parents=True)
        with open(d / "1_counter_collection.csv", "w", newline="") as f:
            w = csv.DictWriter(f, cols)
            w.writeheader()
            for disp in range(1, 8):                    # 4 step launches at full size, 3 tail launches at a quarter
                scale = 1.0 if disp <= 4 else 0.25
                for c, v in counters.items():
                    w.writerow({"Dispatch_Id": disp, "Kernel_Name": seg, "Counter_Name": c, "Counter_Value": v * scale})
    pa.main(str(tmp_path), tail=3)
    out = capsys.readouterr().out
    assert "dispatches per pass 4, shader cycles per dispatch 1,000,000" in out           # 8e6 / 8 XCDs, the tail left out
    assert "TA_TA_BUSY_sum" in out and "0.500 of 256 units x cycles" in out               # 128e6 / (256 x 1e6)
    assert "SQ_WAIT_ANY / SQ_WAVE_CYCLES = 0.400" in out and "L1->L2 read latency 424 cycles per request" in out
    pa.main(str(tmp_path))
    assert "dispatches per pass 7" in capsys.readouterr().out                             # without the tail argument everything is averaged


def test_isa_counts_file_matches_the_kernels_as_compiled(tmp_path):
    """profiles/isa_counts.json prices the roofline's node visits and triangle tests: its I_node / I_node_uniform / I_node_uniform_any / I_tri must be
    what the CURRENT rt_kernels.hip compiles to (the marker build of `make asm`), or frac_executed <= counter_frac would hold by accident."""
    import shutil
    import subprocess
    import pytest
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    r = _load("roofline")
    csrc = os.path.join(ROOT, "caitlynrenderer_amd", "csrc")
    out = tmp_path / "k.s"
    subprocess.run([hipcc, "-std=c++17", "-O3", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize", "--offload-arch=gfx950", "-DCRT_ISA_MARKS",
                    "-S", "--cuda-device-only", "-o", str(out), os.path.join(csrc, "rt_kernels.hip")], check=True, capture_output=True, timeout=600)
    k = r.parse_asm(str(out))
    name = next(n for n in k if r.demangle_args(n) == [1, 0, 0, 0, 1, 0, 0, 1, 1, 1])          # <FIRST, INPLACE, BATCH, WIDE, ONE>: the headline's launch
    regs = k[name]["regions"]
    isa = json.load(open(os.path.join(ROOT, "profiles", "isa_counts.json")))
    uni = [x[2] for x in regs if x[0] == "uninode"]
    assert isa["I_node"] == next(x[2] for x in regs if x[0] == "node") and isa["I_tri"] == next(x[2] for x in regs if x[0] == "tri")
    assert isa["I_node_uniform"] == uni[0] and isa["I_node_uniform_any"] == uni[-1] and len(uni) == 2
