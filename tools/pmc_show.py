"""development aid: per-kernel means of the counters collected by tools/pmc_kernel_counters.sh"""
import collections, csv, glob, sys
tag, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
for d in (f"pmc_{tag}1", f"pmc_{tag}2"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"gpurun_out/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            if pat in k:
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(agg.items()):
        print(k, {c: round(sum(x) / len(x)) for c, x in v.items()})
