import csv, glob, sys, collections, os
# usage: pmcsum.py dir  -> per kernel (short name) mean counters
for d in sys.argv[1:]:
    for f in sorted(glob.glob(os.path.join(d, "*", "*_counter_collection.csv")) + glob.glob(os.path.join(d, "*_counter_collection.csv")), key=os.path.getmtime)[-1:]:
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            k = "fixup" if "fixup" in k else ("hop" if "hop_kernel" in k else None)
            if k: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k in acc:
            print(d, k, {c: (sum(v)/len(v), len(v)) for c, v in acc[k].items()})
