"""development aid: print the per-kernel table of a rocprofv3 --stats run (name shortened, avg/min us, calls)"""
import csv, glob, sys
for path in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(path)))
    for r in rows:
        import re; m = re.search(r"(\w+_kernel(<[^>]*>)?|__amd\w+|\w+elementwise\w*)", r["Name"]); name = (m.group(1) if m else r["Name"][:44])
        print(f"{name:46s} calls {int(r['Calls']):4d}  avg {float(r['AverageNs'])/1e3:9.1f} us  min {float(r['MinNs'])/1e3:9.1f}  total {float(r['TotalDurationNs'])/1e6:8.2f} ms")
