import csv, glob, sys, collections, os
for d in sys.argv[2:]:
    for f in (glob.glob(os.path.join(d, "*_counter_collection.csv")) + glob.glob(os.path.join(d, "*", "*_counter_collection.csv")))[-1:]:
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if sys.argv[1] in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        print(d, {c: round(sum(v) / len(v)) for c, v in acc.items()})
