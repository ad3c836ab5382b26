#!/usr/bin/env python3
"""Mean counter value per dispatch and kernel of one or more `rocprofv3 --pmc` output directories.

usage: tools/pmc_counters.py [--kernel SUBSTR] [--skip-last N] DIR [DIR ...]
--kernel     only kernels whose name contains SUBSTR (default "k_segment")
--skip-last  leave out the last N dispatches of every kernel (bench.py ends a run with up to 10 single-sample frames under full event
             timing; with --spp 4 they are launches of another kernel or another size)
Prints one line per (directory, kernel, counter): dispatches, mean, min, max.
"""
import collections, csv, glob, os, re, sys


def short(name):
    n = name.split("(")[0].replace("void crt::", "")
    m = re.match(r"k_segment<(.*)>", n)
    if not m:
        return n
    flags = [x.strip() == "true" for x in m.group(1).split(",")]
    names = ["FIRST", "STATS", "TEX", "PRETRACED", "INPLACE", "BVH2", "MAT", "COMPACT", "SHARE", "BATCH", "WIDE", "ONE"]
    return "k_segment<" + ",".join(n for n, f in zip(names, flags) if f) + ">"


def main(argv):
    kern, skip, dirs = "k_segment", 0, []
    it = iter(argv)
    for a in it:
        if a == "--kernel":
            kern = next(it)
        elif a == "--skip-last":
            skip = int(next(it))
        else:
            dirs.append(a)
    for d in dirs:
        rows = []
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            rows += list(csv.DictReader(open(f)))
        per = collections.defaultdict(lambda: collections.defaultdict(dict))     # kernel -> counter -> dispatch id -> value
        for r in rows:
            if kern in r["Kernel_Name"]:
                per[short(r["Kernel_Name"])][r["Counter_Name"]][int(r["Dispatch_Id"])] = float(r["Counter_Value"])
        for k in sorted(per):
            for c in sorted(per[k]):
                ids = sorted(per[k][c])
                if skip:
                    ids = ids[:-skip]
                vals = [per[k][c][i] for i in ids]
                if vals:
                    print(f"{os.path.basename(d.rstrip('/')):28s} {k:44s} {c:28s} n={len(vals)} mean={sum(vals) / len(vals):.1f} min={min(vals):14.2f} max={max(vals):14.2f}")


if __name__ == "__main__":
    main(sys.argv[1:])
