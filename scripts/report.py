#!/usr/bin/env python3
"""Reduce dwarf_bench CSV reports the way the reference's notebook does (scripts/report-sample.ipynb cells 6-7):
per (device_type, buffer size) drop every iteration whose host time equals the group's maximum (the warm-up),
then average the rest — extended with min / median, the kernel-time column and throughput columns
(Mrows/s, algorithmic GB/s and its fraction of the 8 TB/s HBM peak).

The input schema is the one MeasureResults::write_csv emits (common/result.cpp:59-91 in the reference):
    device_type,buf_size_bytes,<timing columns in ms...>
so CSVs written by the reference binary reduce with the same command and can sit in the same table.

    scripts/report.py reports/report_scan_hip.csv --bytes-per-row 4
    scripts/report.py a.csv b.csv --time-column kernel_time_ms --format csv
"""
from __future__ import annotations

import argparse
import csv
import statistics
import sys
from collections import OrderedDict

HBM_PEAK_GBPS = 8000.0
ELEM_BYTES = 4  # buf_size_bytes = elements * sizeof(int) (common/result.cpp:67-68)

# algorithmic bytes per input row, by report name fragment; --bytes-per-row overrides
DEFAULT_BYTES_PER_ROW = OrderedDict([
    ("groupby", 8.0),  # key + value read once (+ 4 B per group, negligible)
    ("join_omnisci", 20.0),  # n build + n probe rows: 4+4 keys in, 4 ids out, 8 (pos,count) out, per n
    ("join", 28.0),  # payload join: 2x(key,val) in, (key, a_val, b_val) out, per n
    ("hash_build", 4.0),
    ("radix", 8.0),  # keys in + keys out (compulsory traffic)
    ("sort", 8.0),
    ("scan", 4.0),  # + 4*selectivity, 4e-4 on reference data
])


def guess_bytes_per_row(path: str) -> float | None:
    low = path.lower()
    for frag, b in DEFAULT_BYTES_PER_ROW.items():
        if frag in low:
            return b
    return None


def read_rows(path: str):
    """-> (timing column names, [(device_type, buf_size_bytes, [timings ms])]); tolerates repeated headers"""
    with open(path, newline="") as f:
        rows = [r for r in csv.reader(f) if r]
    if not rows:
        return [], []
    header = rows[0]
    if header[:2] != ["device_type", "buf_size_bytes"]:
        raise ValueError(f"{path}: not a dwarf_bench report (header {header})")
    out = []
    for r in rows[1:]:
        if r == header:
            continue
        out.append((r[0], int(r[1]), [float(x) for x in r[2:]]))
    return header[2:], out


def reduce_rows(rows, col: int):
    """the notebook's reducer: group by (size, device); drop rows equal to the group max; mean of the rest"""
    groups: "OrderedDict[tuple, list]" = OrderedDict()
    for dev, size, timings in rows:
        groups.setdefault((size, dev), []).append(timings)
    out = []
    for (size, dev), ts in sorted(groups.items()):
        vals = [t[col] for t in ts]
        worst = max(vals)
        kept = [v for v in vals if v != worst] or vals  # a single iteration (or all equal) keeps itself
        out.append({
            "device_type": dev,
            "buf_size_bytes": size,
            "rows": size // ELEM_BYTES,
            "iterations": len(vals),
            "kept": len(kept),
            "mean_ms": statistics.fmean(kept),
            "min_ms": min(vals),
            "median_ms": statistics.median(vals),
            "max_ms": worst,
        })
    return out


def add_throughput(rec: dict, bytes_per_row: float | None) -> dict:
    ms = rec["mean_ms"]
    rec["mrows_per_s"] = rec["rows"] / (ms * 1e-3) / 1e6 if ms > 0 else float("inf")
    if bytes_per_row is not None and ms > 0:
        rec["gb_per_s"] = rec["rows"] * bytes_per_row / (ms * 1e-3) / 1e9
        rec["hbm_frac"] = rec["gb_per_s"] / HBM_PEAK_GBPS
    else:
        rec["gb_per_s"] = rec["hbm_frac"] = None
    return rec


COLUMNS = ["report", "device_type", "buf_size_bytes", "rows", "iterations", "kept", "mean_ms", "min_ms", "median_ms",
           "max_ms", "mrows_per_s", "gb_per_s", "hbm_frac"]


def fmt(v):
    if v is None:
        return ""
    if isinstance(v, float):
        return f"{v:.6g}"
    return str(v)


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("reports", nargs="+")
    ap.add_argument("--time-column", default=None,
                    help="timing column to reduce (default: the first one, host_time_ms / total_time)")
    ap.add_argument("--bytes-per-row", type=float, default=None,
                    help="algorithmic bytes per input row (default: guessed from the report name)")
    ap.add_argument("--format", choices=["table", "csv"], default="table")
    args = ap.parse_args(argv)

    records = []
    for path in args.reports:
        names, rows = read_rows(path)
        if not rows:
            continue
        col = 0
        if args.time_column is not None:
            if args.time_column not in names:
                print(f"{path}: no column {args.time_column} (has {names})", file=sys.stderr)
                return 2
            col = names.index(args.time_column)
        bpr = args.bytes_per_row if args.bytes_per_row is not None else guess_bytes_per_row(path)
        for rec in reduce_rows(rows, col):
            rec["report"] = path.rsplit("/", 1)[-1]
            records.append(add_throughput(rec, bpr))

    if args.format == "csv":
        w = csv.writer(sys.stdout)
        w.writerow(COLUMNS)
        for r in records:
            w.writerow([fmt(r[c]) for c in COLUMNS])
    else:
        cells = [COLUMNS] + [[fmt(r[c]) for c in COLUMNS] for r in records]
        widths = [max(len(row[i]) for row in cells) for i in range(len(COLUMNS))]
        for row in cells:
            print("  ".join(c.rjust(w) for c, w in zip(row, widths)))
    return 0


if __name__ == "__main__":
    sys.exit(main())
