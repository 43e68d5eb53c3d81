#!/usr/bin/env python3
"""Condense the raw rocprofv3 output of tools/profile_round.sh into the small files kept under profiles/:

  profiles/<tag>_bench_kernel_stats.csv   the --kernel-trace --stats table of `bench.py --no-sweep --no-cpu` (verbatim)
  profiles/<tag>_headline.json            the same trace split per BASELINE configuration: for every kernel of a
                                          configuration its dispatches, mean and min duration, and per call the span
                                          from the first kernel's start to the last kernel's end — with the algorithmic
                                          bytes beside it, so every roofline fraction can be recomputed from this file
  profiles/<tag>_pmc.json                 per kernel: dispatches, mean/max FETCH_SIZE and WRITE_SIZE (KiB as reported)
  profiles/hbm_traffic.json               HBM bytes per call for each dwarf, which bench.py reads for `roofline.traffic`
                                          (with a `_source` block: tool, command, git head of the collection)

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
sys.path.insert(0, str(ROOT))
from dwarf_bench_amd.build import kernel_tree_sha256  # noqa: E402  (the device code the counters were collected on)

# dwarf -> (regex of its kernels, regex of the anchor kernel launched exactly once per call, double FETCH_SIZE?)
DWARFS = {
    "scan": (r"scan_chunk_kernel|scan_move_kernel", r"scan_chunk_kernel", True),
    "sort_8bit": (r"rs_\w+<8", r"rs_histogram", True),
    "sort_4bit": (r"rs_\w+<4", r"rs_histogram", True),
    "groupby": (r"gb_aggregate\w*kernel|gb_reduce\w*kernel", r"gb_aggregate\w*kernel", True),
    # join: 4-B/lane reads and random 16-B gathers — widths the guide calls uncalibrated: raw counter, not doubled
    "join_build": (r"jl_(hist|offsets|scatter)\w*|jl_build_kernel<false|jl_giant_(count|ids)_kernel", r"jl_build_kernel<false", False),
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


# ---- the kernel trace split per BASELINE configuration --------------------------------------------------------------
# configuration -> (regex of its kernels, regex of the kernel that opens a call, regex of the one that closes it,
#                   algorithmic bytes per call as f(n), rows per call)
N_SCAN, N_SORT, N_GB, GROUPS, N_JOIN = 1 << 28, 1 << 24, 1 << 26, 1 << 16, 1 << 26
CONFIGS = {
    "scan_2p28": (r"scan_chunk_kernel|scan_move_kernel", r"scan_chunk_kernel", r"scan_move_kernel"),
    "sort_2p24_8bit": (r"rs_\w+<8|rs_finalize", r"rs_histogram_kernel<8", r"rs_finalize_kernel"),
    "sort_2p24_4bit": (r"rs_\w+<4|rs_finalize", r"rs_histogram_kernel<4", r"rs_finalize_kernel"),
    "groupby_2p26_2p16": (r"gb_aggregate\w*kernel|gb_reduce\w*kernel", r"gb_aggregate\w*kernel", r"gb_reduce\w*kernel"),
    # (the two launches for giant partitions close a build; they find none on the bench's uniform keys)
    "join_build": (r"jl_(hist|offsets|scatter)\w*|jl_build_kernel<false|jl_giant_(count|ids)_kernel", r"jl_hist0_kernel|jl_hist_fused\w*_kernel", r"jl_giant_ids_kernel"),
    "join_probe": (r"jl_probe_kernel", r"jl_probe_kernel", r"jl_probe_kernel"),
    # the radix join: both sides through the partitioner, then the fused build + probe launch (+ two for giants)
    "join_radix_2p26": (r"jl_(hist|offsets|scatter)\w*|jl_build_kernel<true|jl_giant_\w+_kernel", r"jl_hist0_kernel|jl_hist_fused\w*_kernel", r"jl_giant_ids_kernel"),
}
# a kernel of ANOTHER configuration that shares kernels with this one: a call that meets it is not this configuration's
FOREIGN = {"join_build": r"jl_build_kernel<true", "join_radix_2p26": r"jl_build_kernel<false|jl_probe_kernel"}


def read_trace(raw: Path):
    rows = []
    for f in (raw / "kt").rglob("*kernel_trace.csv"):
        with f.open(newline="") as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), int(r["Grid_Size_X"])))
    rows.sort()
    return rows


def headline(raw: Path, matches_scan: float = 4.034e-4):
    """per configuration: kernels {dispatches, avg_us, min_us} and the per-call span, from the chronological trace"""
    rows = read_trace(raw)
    alg = {"scan_2p28": 4 * N_SCAN * (1 + matches_scan), "sort_2p24_8bit": 8 * N_SORT, "sort_2p24_4bit": 8 * N_SORT,
           "groupby_2p26_2p16": 8 * N_GB + 4 * GROUPS, "join_build": None, "join_probe": None,
           "join_radix_2p26": 24 * N_JOIN}
    out = {}
    for name, (pat, first, last) in CONFIGS.items():
        foreign = FOREIGN.get(name)
        mine = [r for r in rows if re.search(pat, r[2]) or (foreign and re.search(foreign, r[2]))]
        if not mine:
            continue
        if name == "scan_2p28":  # the bench also scans 2^22 rows once for its result check: keep the 2^28 launches
            big = max(r[3] for r in mine if re.search(first, r[2]))
            keep, on = [], False
            for r in mine:
                if re.search(first, r[2]):
                    on = r[3] == big
                if on:
                    keep.append(r)
            mine = keep
        # cut the chronological list into calls: [opening kernel .. closing kernel]
        calls, cur = [], None
        for r in mine:
            if foreign and re.search(foreign, r[2]):
                cur = None
                continue
            if re.search(first, r[2]) and cur is None:
                cur = []
            if cur is not None:
                cur.append(r)
                if re.search(last, r[2]):
                    calls.append(cur)
                    cur = None
        # the same kernels also run at other sizes in the bench (the 2^30 join of pjoin_p1, the 2^30-key sort inside
        # its device-side check): keep the calls of this configuration's size = those within 4x of the shortest one
        dur = [c[-1][1] - c[0][0] for c in calls]
        calls = [c for c, d in zip(calls, dur) if d < 4 * min(dur)]
        calls = calls[len(calls) // 3:]  # drop the warm-up calls at the head of every leg
        mine = [r for c in calls for r in c]
        kernels = defaultdict(list)
        for s0, e0, k, g in mine:
            kernels[k].append((e0 - s0) / 1e3)
        spans = [(c[-1][1] - c[0][0]) / 1e3 for c in calls]
        entry = {"calls": len(spans), "span_us_avg": sum(spans) / len(spans) if spans else None,
                 "span_us_min": min(spans) if spans else None,
                 "kernels": {k: {"dispatches": len(v), "avg_us": sum(v) / len(v), "min_us": min(v)} for k, v in sorted(kernels.items())}}
        if alg.get(name) and spans:
            entry["algorithmic_bytes"] = alg[name]
            entry["frac_of_8TBps_from_span"] = alg[name] / (entry["span_us_avg"] * 1e-6) / 8e12
        out[name] = entry
    if "join_build" in out and "join_probe" in out and out["join_build"]["span_us_avg"] and out["join_probe"]["span_us_avg"]:
        tot = out["join_build"]["span_us_avg"] + out["join_probe"]["span_us_avg"]
        out["join_2p26"] = {"span_us_avg": tot, "algorithmic_bytes": 20 * N_JOIN,
                            "frac_of_8TBps_from_span": 20 * N_JOIN / (tot * 1e-6) / 8e12,
                            "probe_gathers_per_s": N_JOIN / (out["join_probe"]["span_us_avg"] * 1e-6)}
    return out


def main():
    raw, tag = Path(sys.argv[1]), sys.argv[2]
    git_head = sys.argv[3] if len(sys.argv) > 3 else "unknown"
    prof = ROOT / "profiles"
    prof.mkdir(exist_ok=True)
    stamp = raw / "kernel_tree_sha256.txt"  # written on the GPU box at collection time (tools/profile_round.sh)
    if stamp.exists() and stamp.read_text().strip() != kernel_tree_sha256():
        sys.exit(f"{raw}: collected on device code {stamp.read_text().strip()[:12]}, this tree is {kernel_tree_sha256()[:12]} — "
                 "profiles/ left untouched")
    stats = sorted((raw / "kt").rglob("*kernel_stats.csv"))
    fetch = read_pmc(raw / "pmc_fetch", "FETCH_SIZE")
    write = read_pmc(raw / "pmc_write", "WRITE_SIZE")
    if not stats or not fetch or not write:
        sys.exit(f"{raw}: incomplete rocprofv3 output (kernel stats: {bool(stats)}, FETCH_SIZE: {bool(fetch)}, "
                 f"WRITE_SIZE: {bool(write)}) — profiles/ left untouched")
    shutil.copy(stats[0], prof / f"{tag}_bench_kernel_stats.csv")
    (prof / f"{tag}_headline.json").write_text(json.dumps({
        "_comment": "rocprofv3 --kernel-trace of `python3 bench.py --steps 20 --warmup 3 --no-cpu --no-sweep` split per "
                    "BASELINE configuration (tools/profile_summary.py); durations in us; span = first kernel start -> "
                    "last kernel end of one call; fractions against 8.0 TB/s",
        "git_head": git_head, "kernel_tree_sha256": kernel_tree_sha256(), "configurations": headline(raw)}, indent=1))
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
                   "--pmc WRITE_SIZE (separate passes, tools/profile_round.sh)", "git_head": git_head, "kernels": per_kernel}, indent=1))

    traffic = {"_source": {"tool": "tools/profile_round.sh -> tools/profile_summary.py", "round_tag": tag, "git_head": git_head,
                           "kernel_tree_sha256": kernel_tree_sha256(),
                           "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE -- python3 bench.py --steps 10 --warmup 2 "
                                      "--no-cpu --no-pjoin --no-sweep (one counter per pass)"},
               "_comment": "HBM bytes per call from rocprofv3 --pmc passes (profiles/%s_pmc.json); counters are KiB; "
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
