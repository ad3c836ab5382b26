#!/usr/bin/env python3
"""profiles/pmc_traffic.json from the two --pmc passes of tools/profile_round.sh.

usage: tools/pmc_traffic.py <round dir> <out json>
HBM bytes per launch of the dominant kernel = (2*FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE counts half of
the fetched bytes (MI355X_MICROARCH.md, HBM section); both counters are in KiB.  Only the non-counting variant of
k_segment<FIRST> (second template argument false) is aggregated, i.e. the launches of the timed region.
"""
import collections, csv, glob, json, os, sys

def per_kernel(dirname):
    agg = collections.defaultdict(list)
    for f in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            agg[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
    return agg

def main(round_dir, out):
    res = {}
    for wl in ("cornell", "mesh1m"):
        agg = {}
        for kind in ("fetch", "write"):
            agg.update(per_kernel(os.path.join(round_dir, f"pmc_{kind}_{wl}")))
        names = sorted({k for k, _ in agg if "k_segment<true, false" in k})
        if not names:
            continue
        k = names[0]
        mean = lambda c: sum(agg[(k, c)]) / len(agg[(k, c)])
        fetch, write, hit, miss = mean("FETCH_SIZE"), mean("WRITE_SIZE"), mean("TCC_HIT_sum"), mean("TCC_MISS_sum")
        res[f"{wl}_d1"] = {
            "kernel": k, "FETCH_SIZE_KB": round(fetch, 1), "WRITE_SIZE_KB": round(write, 1),
            "hbm_bytes_per_launch": int((2 * fetch + write) * 1024),
            "correction": "(2*FETCH_SIZE + WRITE_SIZE)*1024: gfx950 FETCH_SIZE reports half of the fetched bytes (MI355X_MICROARCH.md, HBM)",
            "l2_hit_rate": round(hit / (hit + miss), 3), "dispatches": len(agg[(k, "FETCH_SIZE")]),
        }
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))

if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
