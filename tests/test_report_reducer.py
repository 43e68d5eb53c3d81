"""scripts/report.py: the notebook-equivalent reducer (reference scripts/report-sample.ipynb cells 6-7:
drop rows equal to the per-(size, device) maximum, mean of the rest) plus throughput columns."""
import importlib.util
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


@pytest.fixture(scope="module")
def report():
    spec = importlib.util.spec_from_file_location("dbreport", ROOT / "scripts" / "report.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _write(path, header, rows):
    path.write_text("\n".join([",".join(header)] + [",".join(map(str, r)) for r in rows]) + "\n")


def test_drop_max_then_mean(tmp_path, report):
    p = tmp_path / "report_scan_hip.csv"
    hdr = ["device_type", "buf_size_bytes", "host_time_ms", "kernel_time_ms"]
    rows = [("HIP", 4096, t, t / 2) for t in (9.0, 1.0, 2.0, 3.0)] + [("CPU", 4096, t, 0) for t in (5.0, 5.0, 7.0)]
    _write(p, hdr, rows)
    names, parsed = report.read_rows(str(p))
    assert names == ["host_time_ms", "kernel_time_ms"]
    red = {r["device_type"]: r for r in report.reduce_rows(parsed, 0)}
    assert red["HIP"]["mean_ms"] == pytest.approx(2.0) and red["HIP"]["kept"] == 3 and red["HIP"]["max_ms"] == 9.0
    assert red["HIP"]["min_ms"] == 1.0 and red["HIP"]["median_ms"] == pytest.approx(2.5)
    assert red["CPU"]["mean_ms"] == pytest.approx(5.0)
    assert red["HIP"]["rows"] == 1024
    kern = {r["device_type"]: r for r in report.reduce_rows(parsed, 1)}
    assert kern["HIP"]["mean_ms"] == pytest.approx(1.0)


def test_all_equal_and_single_iteration_keep_their_rows(tmp_path, report):
    p = tmp_path / "r.csv"
    _write(p, ["device_type", "buf_size_bytes", "host_time_ms", "kernel_time_ms"],
           [("HIP", 1024, 2.0, 1.0), ("HIP", 2048, 3.0, 1.0), ("HIP", 2048, 3.0, 1.0)])
    _, parsed = report.read_rows(str(p))
    red = report.reduce_rows(parsed, 0)
    assert [r["mean_ms"] for r in red] == [2.0, 3.0]


def test_throughput_columns(report):
    rec = report.add_throughput({"rows": 1 << 28, "mean_ms": 0.2}, 4.0)
    assert rec["mrows_per_s"] == pytest.approx((1 << 28) / 0.2e-3 / 1e6)
    assert rec["gb_per_s"] == pytest.approx((1 << 30) / 0.2e-3 / 1e9)
    assert rec["hbm_frac"] == pytest.approx(rec["gb_per_s"] / 8000.0)
    assert report.guess_bytes_per_row("reports/report_join_omnisci_hip.csv") == 20.0
    assert report.guess_bytes_per_row("reports/report_radix_hip.csv") == 8.0
    assert report.guess_bytes_per_row("x.csv") is None


def test_appended_reports_with_repeated_header_and_custom_columns(tmp_path, report, capsys):
    p = tmp_path / "report_groupby_local_hip.csv"
    hdr = ["device_type", "buf_size_bytes", "total_time", "group_by_time", "reduction_time"]
    _write(p, hdr, [("HIP", 4096, 3, 2, 1), tuple(hdr), ("HIP", 4096, 5, 4, 1), ("HIP", 4096, 4, 3, 1)])
    assert report.main([str(p), "--time-column", "group_by_time", "--format", "csv"]) == 0
    out = capsys.readouterr().out.strip().splitlines()
    assert out[0].split(",")[:3] == ["report", "device_type", "buf_size_bytes"]
    cells = dict(zip(out[0].split(","), out[1].split(",")))
    assert float(cells["mean_ms"]) == pytest.approx(2.5) and cells["kept"] == "2" and cells["iterations"] == "3"
    assert report.main([str(p), "--time-column", "nope"]) == 2
