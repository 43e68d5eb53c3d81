"""development aid: tables out of rocprofv3 output directories (the summaries that are judged live in profiles/ and are
written by tools/profile_summary.py; this is for looking at one experiment's raw output under gpurun_out/).

  python tools/prof_show.py stats <dir>            per-kernel table of a --kernel-trace --stats run (avg / min us, calls)
  python tools/prof_show.py counters <tag> [sub]   per-kernel means of the two SQ passes of tools/pmc_kernel_counters.sh
                                                   (gpurun_out/pmc_<tag>1, pmc_<tag>2); sub filters kernel names
  python tools/prof_show.py valu <csv...>          per-kernel counter means of --pmc csv files plus the derived VALU issue
                                                   utilisation: SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)
"""
import collections
import csv
import glob
import re
import sys


def short(name):
    m = re.search(r"(\w+_kernel(<[^>]*>)?|__amd\w+|\w+elementwise\w*)", name)
    return m.group(1) if m else name[:44]


def stats(args):
    for path in glob.glob(args[0] + "/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            print(f"{short(r['Name']):46s} calls {int(r['Calls']):4d}  avg {float(r['AverageNs']) / 1e3:9.1f} us  "
                  f"min {float(r['MinNs']) / 1e3:9.1f}  total {float(r['TotalDurationNs']) / 1e6:8.2f} ms")


def counters(args):
    tag, pat = args[0], (args[1] if len(args) > 1 else "")
    for d in (f"pmc_{tag}1", f"pmc_{tag}2"):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for f in glob.glob(f"gpurun_out/{d}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
                if pat in k:
                    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in sorted(agg.items()):
            print(k, {c: round(sum(x) / len(x)) for c, x in v.items()})


def valu(args):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in args:
        for r in csv.DictReader(open(f)):
            name = short(r["Kernel_Name"])
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
            acc[name]["dur_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for name, c in sorted(acc.items()):
        m = {k: sum(v) / len(v) for k, v in c.items()}
        if m["dur_us"] < 8:
            continue
        line = f"{name:44s} {m['dur_us']:8.1f} us"
        if m.get("GRBM_GUI_ACTIVE") and "SQ_INSTS_VALU" in m:
            line += f"  VALU util {m['SQ_INSTS_VALU'] * 4 / (1024 * m['GRBM_GUI_ACTIVE'] / 8) * 100:5.1f} %"
        for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"):
            if k in m:
                line += f"  {k[9:]} {m[k] / 1e6:7.2f}M"
        print(line)


if __name__ == "__main__":
    modes = {"stats": stats, "counters": counters, "valu": valu}
    if len(sys.argv) < 3 or sys.argv[1] not in modes:
        raise SystemExit(__doc__)
    modes[sys.argv[1]](sys.argv[2:])
