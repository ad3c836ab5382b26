import csv, glob, os, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(os.path.join(out, "pass*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row.get("Kernel_Name", "?").split("(")[0]
            c = row.get("Counter_Name"); v = float(row.get("Counter_Value", 0) or 0)
            agg[k][c] += v; calls[k][c] += 1
for k in sorted(agg):
    print("==", k)
    for c in sorted(agg[k]):
        n = calls[k][c]
        print(f"   {c:40s} per-dispatch {agg[k][c]/max(n,1):16.1f}   (dispatches {n})")
