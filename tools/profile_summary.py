#!/usr/bin/env python3
"""Condense the raw rocprofv3 output of tools/profile_r01.sh into the small files kept under profiles/:

  profiles/<tag>_bench_kernel_stats.csv   the --kernel-trace --stats table of `bench.py` (verbatim)
  profiles/<tag>_pmc.json                 per kernel: dispatches, mean/max FETCH_SIZE and WRITE_SIZE (KiB as reported)
  profiles/hbm_traffic.json               HBM bytes per call for each dwarf, which bench.py reads for `roofline.traffic`

HBM bytes follow MI355X_MICROARCH.md's rocprofv3 section: the counters are in KiB; on gfx950 FETCH_SIZE counts
wide (16 B/lane) coalesced reads at half their size, so it is doubled for the streaming kernels flagged below;
WRITE_SIZE is exact.  Per call = sum over the dwarf's kernels / number of calls (dispatches of its anchor kernel).
"""
import csv
import json
import re
import shutil
import sys
from collections import defaultdict
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]

# dwarf -> (regex of its kernels, regex of the anchor kernel launched exactly once per call, double FETCH_SIZE?)
DWARFS = {
    "scan": (r"scan_chunk_kernel|scan_move_kernel", r"scan_chunk_kernel", True),
    "sort_8bit": (r"rs_\w+<8", r"rs_histogram", True),
    "sort_4bit": (r"rs_\w+<4", r"rs_histogram", True),
    "groupby": (r"gb_aggregate_kernel|gb_reduce_kernel", r"gb_aggregate_kernel", True),
    # join: 4-B/lane reads and random 16-B gathers — widths the guide calls uncalibrated: raw counter, not doubled
    "join_build": (r"jl_(hist|offsets|scatter)\w*|jl_build_kernel", r"jl_build_kernel", False),
    "join_probe": (r"jl_probe_kernel", r"jl_probe_kernel", False),
}


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0]


def read_pmc(d: Path, counter: str):
    rows = defaultdict(list)
    for f in d.rglob("*counter_collection.csv"):
        with f.open(newline="") as fh:
            for r in csv.DictReader(fh):
                if r["Counter_Name"] == counter:
                    rows[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return rows


def main():
    raw, tag = Path(sys.argv[1]), sys.argv[2]
    prof = ROOT / "profiles"
    prof.mkdir(exist_ok=True)
    stats = sorted((raw / "kt").rglob("*kernel_stats.csv"))
    fetch = read_pmc(raw / "pmc_fetch", "FETCH_SIZE")
    write = read_pmc(raw / "pmc_write", "WRITE_SIZE")
    if not stats or not fetch or not write:
        sys.exit(f"{raw}: incomplete rocprofv3 output (kernel stats: {bool(stats)}, FETCH_SIZE: {bool(fetch)}, "
                 f"WRITE_SIZE: {bool(write)}) — profiles/ left untouched")
    shutil.copy(stats[0], prof / f"{tag}_bench_kernel_stats.csv")
    per_kernel = {}
    for k in sorted(set(fetch) | set(write)):
        if not k.startswith("dbhip::"):
            continue
        f, w = fetch.get(k, []), write.get(k, [])
        per_kernel[k] = {"dispatches": len(f) or len(w),
                         "FETCH_SIZE_KiB_mean": sum(f) / len(f) if f else None, "FETCH_SIZE_KiB_max": max(f) if f else None,
                         "WRITE_SIZE_KiB_mean": sum(w) / len(w) if w else None, "WRITE_SIZE_KiB_max": max(w) if w else None}
    (prof / f"{tag}_pmc.json").write_text(json.dumps({
        "command": "rocprofv3 --pmc FETCH_SIZE -- python3 bench.py --steps 10 --warmup 2 --no-cpu --no-pjoin --no-sweep ; same with "
                   "--pmc WRITE_SIZE (separate passes, tools/profile_r01.sh)", "kernels": per_kernel}, indent=1))

    traffic = {"_comment": "HBM bytes per call from rocprofv3 --pmc passes (profiles/%s_pmc.json); counters are KiB; "
                           "FETCH_SIZE doubled for the 16-B/lane streaming reads on gfx950 (MI355X_MICROARCH.md, HBM "
                           "section); WRITE_SIZE exact; per call = sum over the dwarf's kernels / calls" % tag}
    for dwarf, (pat, anchor, dbl) in DWARFS.items():
        def total(rows, full_only):
            tot, calls = 0.0, 0
            for k, vals in rows.items():
                if not re.search(pat, k):
                    continue
                if full_only:  # scan: the bench also launches a 2^22 result check; keep the 2^28 launches only
                    vals = [v for v in vals if v >= 0.5 * max(vals)]
                tot += sum(vals)
                if re.search(anchor, k):
                    calls += len(vals)
            return tot, calls
        rd, calls_r = total(fetch, dwarf == "scan")
        wr, calls_w = total(write, dwarf == "scan")
        if not calls_r or not calls_w:
            continue
        read_b = rd * 1024 * (2 if dbl else 1) / calls_r
        write_b = wr * 1024 / calls_w
        traffic[dwarf] = {"hbm_bytes_per_launch": read_b + write_b, "read_bytes": read_b, "write_bytes": write_b,
                          "calls_seen": calls_r, "fetch_size_doubled": dbl}
    if "join_build" in traffic and "join_probe" in traffic:
        traffic["join"] = {k: traffic["join_build"][k] + traffic["join_probe"][k]
                           for k in ("hbm_bytes_per_launch", "read_bytes", "write_bytes")}
    (prof / "hbm_traffic.json").write_text(json.dumps(traffic, indent=1))
    print(json.dumps(traffic, indent=1))
    print("kernels seen:", *sorted(per_kernel), sep="\n  ")


if __name__ == "__main__":
    main()
