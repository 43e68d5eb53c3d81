"""development aid: per-kernel means of a rocprofv3 --pmc run (counter values are summed over the 8 XCDs) and the
derived VALU issue utilisation: SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)"""
import csv, re, collections, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sys.argv[1:]:
    for r in csv.DictReader(open(f)):
        m = re.search(r"(\w+_kernel)(<[^>]*>)?", r["Kernel_Name"]); name = (m.group(0) if m else r["Kernel_Name"][:40])
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
        acc[name]["dur_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for name, c in sorted(acc.items()):
    m = {k: sum(v) / len(v) for k, v in c.items()}
    if m["dur_us"] < 8: continue
    line = f"{name:44s} {m['dur_us']:8.1f} us"
    if "SQ_INSTS_VALU" in m and "GRBM_GUI_ACTIVE" in m and m["GRBM_GUI_ACTIVE"]:
        line += f"  VALU util {m['SQ_INSTS_VALU'] * 4 / (1024 * m['GRBM_GUI_ACTIVE'] / 8) * 100:5.1f} %"
    for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"):
        if k in m: line += f"  {k[9:]} {m[k] / 1e6:7.2f}M"
    print(line)
