#!/usr/bin/env python3
"""Per kernel: every counter of the passes tools/pmc_anatomy.sh made, as the mean per dispatch and — for cycle counters — as a share of the
dispatch's shader cycles (GRBM_GUI_ACTIVE / 8 XCDs) times the number of units that count (256 TA / TCP / TD, 1024 SIMDs for SQ wave-cycle
counters is NOT applied: SQ cycle counters are reported per SQ_WAVE_CYCLES or per SQ_BUSY where the pass has it).
usage: tools/pmc_anatomy.py DIR [TAIL]"""
import collections, csv, glob, os, sys

UNITS = {"TA_": 256, "TCP_": 256, "TD_": 256}


def main(d, tail=0):
    """tail: how many of the LAST segment-kernel dispatches of each pass to leave out (bench.py ends a run with min(10, steps x spp)
    single-sample frames, `depth` segment launches each: a quarter of the work of a 4-sample launch, some by the same kernel)."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import pmc_traffic
    per = collections.defaultdict(lambda: collections.defaultdict(list))     # kernel -> counter -> values
    for p in sorted(glob.glob(os.path.join(d, "pass*"))):
        if not os.path.isdir(p):
            continue
        for (k, c), v in pmc_traffic.per_kernel(p, tail).items():
            if k.startswith(("void crt::", "crt::")):
                per[k][c] += v
    for k, cs in sorted(per.items()):
        gui = cs.get("GRBM_GUI_ACTIVE")
        if not gui or sum(gui) / len(gui) < 8e4:
            continue
        cyc = sum(gui) / len(gui) / 8.0
        n = min(len(v) for v in cs.values())
        print(f"== {k[:110]}\n   dispatches per pass {n}, shader cycles per dispatch {cyc:,.0f}")
        for c, v in sorted(cs.items()):
            if c == "GRBM_GUI_ACTIVE":
                continue
            m = sum(v) / len(v)
            line = f"   {c:38s} {m:18,.0f}"
            for pre, units in UNITS.items():
                if c.startswith(pre) and ("CYCLES" in c or "BUSY" in c or "STALL" in c or "GATE" in c):
                    line += f"   {m / (units * cyc):6.3f} of {units} units x cycles"
            print(line)
        g = lambda c: (sum(cs[c]) / len(cs[c])) if c in cs else None
        if g("TCP_TCC_READ_REQ_LATENCY_sum") and g("TCP_TCC_READ_REQ_sum"):
            print(f"   -> L1->L2 read latency {g('TCP_TCC_READ_REQ_LATENCY_sum') / g('TCP_TCC_READ_REQ_sum'):.0f} cycles per request")
        if g("TCP_TCP_LATENCY_sum") and g("TCP_TOTAL_ACCESSES_sum"):
            print(f"   -> L1 latency {g('TCP_TCP_LATENCY_sum') / g('TCP_TOTAL_ACCESSES_sum'):.0f} cycles per access")
        if g("SQ_WAVE_CYCLES"):
            w = g("SQ_WAVE_CYCLES")
            for c in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_INST_CYCLES_VMEM_RD"):
                if g(c) is not None:
                    print(f"   -> {c} / SQ_WAVE_CYCLES = {g(c) / w:.3f}")
        if g("SQC_DCACHE_REQ"):
            print(f"   -> scalar cache hit rate {g('SQC_DCACHE_HITS') / max(1.0, g('SQC_DCACHE_REQ')):.3f}; instruction cache miss rate "
                  f"{g('SQC_ICACHE_MISSES') / max(1.0, g('SQC_ICACHE_REQ')):.4f}")
        if g("SQ_LDS_BANK_CONFLICT") is not None and g("SQ_LDS_IDX_ACTIVE"):
            print(f"   -> LDS bank-conflict cycles / LDS active cycles = {g('SQ_LDS_BANK_CONFLICT') / g('SQ_LDS_IDX_ACTIVE'):.3f}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 0)
