"""The bench line's schema (the driver's contract + roofline + cpu_baseline), checked on the newest committed N=1 line
(profiles/r*_bench_n1.json) and on bench.py's argument defaults.  No GPU needed."""
import json
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def _newest(pattern):
    files = sorted((ROOT / "profiles").glob(pattern))
    assert files, pattern
    return files[-1]


def test_committed_bench_line_has_the_contract_fields():
    line = json.loads(_newest("r*_bench_n1.json").read_text())
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["metric"] == json.loads((ROOT / "BASELINE.json").read_text())["metric"]
    assert line["unit"] == "Mrows/s" and line["higher_is_better"] is True and line["vs_baseline"] is None
    assert line["n_gpus"] == 1 and line["data"] == "synthetic" and line["dtype"] == "int32"
    assert "workload" in line["config"] and "model" not in line["config"]
    roof = line["roofline"]
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9
    assert abs(roof["achieved"] - roof["algorithmic_bytes"] / roof["kernel_us_avg"] / 1e3) < 1e-6 * roof["achieved"]
    assert roof["traffic"] is None or 0.99 * roof["algorithmic_bytes"] <= roof["traffic"] <= 1.05 * roof["algorithmic_bytes"]
    committed = json.loads((ROOT / "profiles" / "hbm_traffic.json").read_text())["scan"]["hbm_bytes_per_launch"]
    assert 0.99 * roof["algorithmic_bytes"] <= committed <= 1.05 * roof["algorithmic_bytes"]  # no re-reads
    # value = rows / wall time of a step
    assert abs(line["value"] - line["config"]["rows"] / (line["ms_per_step"] * 1e3)) < 1e-6 * line["value"]
    cpu = line["cpu_baseline"]
    assert cpu["kind"] in ("port", "reference") and cpu["cores"] >= 1 and cpu["unit"] == "Mrows/s" and cpu["sample"]
    assert line["parity_check"] is True
    for name in ("sort_8bit", "sort_4bit", "groupby", "join", "join_radix", "pjoin_p1"):
        assert name in line["dwarfs"], name
    # round 2: provenance of the traffic figure, cold-source control, the second roofline of the join, engine checks
    src = roof["traffic_source"]
    assert src["file"] == "profiles/hbm_traffic.json" and src["measured_in_this_run"] is False and src["provenance"]["git_head"]
    assert 0.97 <= line["cold_over_warm_time"] <= 1.03 and line["value_cold"] > 0
    gather = line["dwarfs"]["join"]["roofline_gather"]
    assert gather["bound"] == "random_gather" and 0 < gather["frac"] <= 1.0
    checks = line["dwarfs"]["pjoin_p1"]["checks"]
    assert checks["wrong_probe_rows"] == 0 and checks["exchange_conserved"] and checks["all_rows_delivered"]


def test_headline_profile_recomputes_the_roofline_fractions():
    """profiles/r*_headline.json (kernel trace split per BASELINE configuration): every fraction follows from the
    spans and algorithmic bytes in the file itself, and the scan's agrees with the bench line within 3 %"""
    head = json.loads(_newest("r*_headline.json").read_text())["configurations"]
    for name in ("scan_2p28", "sort_2p24_8bit", "sort_2p24_4bit", "groupby_2p26_2p16", "join_build", "join_probe", "join_2p26"):
        assert name in head, name
    for name, c in head.items():
        if "frac_of_8TBps_from_span" in c:
            assert abs(c["frac_of_8TBps_from_span"] - c["algorithmic_bytes"] / (c["span_us_avg"] * 1e-6) / 8e12) < 1e-9
    line = json.loads(_newest("r*_bench_n1.json").read_text())
    assert abs(head["scan_2p28"]["frac_of_8TBps_from_span"] / line["roofline"]["frac"] - 1) < 0.03
    k = head["scan_2p28"]["kernels"]
    assert any("scan_chunk_kernel" in n for n in k) and any("scan_move_kernel" in n for n in k)


def test_bench_defaults_and_help():
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup"):
        assert flag in r.stdout
    src = (ROOT / "bench.py").read_text()
    assert 'add_argument("--gpus", type=int, default=1)' in src


def test_bench_refuses_to_run_without_a_gpu():
    """the product has no CPU path: bench.py must say so instead of measuring something else"""
    import torch
    if torch.cuda.is_available():
        return
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "1", "--warmup", "0"], capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0 and "needs a GPU" in (r.stderr + r.stdout)


def test_gpus_n_must_match_the_world_it_runs_in():
    """`--gpus 4` inside a 2-rank job would print an n_gpus = 2 line under a command that says 4: refused, non-zero,
    before torch is imported"""
    import os
    env = {**os.environ, "WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"}
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "4"], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode != 0 and "--gpus 4 but WORLD_SIZE = 2" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_gpus_n_without_a_launcher_starts_its_own_ranks():
    """no WORLD_SIZE around `bench.py --gpus 2`: the process starts two rank processes itself and leaves with THEIR
    failure when they fail (here: no GPU) — never a one-GPU line with exit code 0"""
    import os
    import torch
    if torch.cuda.is_available():
        return  # the GPU suite runs the real thing (tests/test_gpu_bench_launcher.py)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode != 0
    assert "started ranks 0..1 as child processes" in r.stderr and r.stderr.count("needs a GPU") == 2
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_the_traffic_figures_belong_to_this_device_code():
    """`roofline.traffic` and every `pmc_traffic_bytes` of the bench line come from committed profiler passes
    (profiles/hbm_traffic.json): they are only worth reporting for the kernels they were collected on.  The file and the
    newest headline profile carry the content hash of csrc/ + include/dbhip.h at collection time; a kernel edited after that
    fails here until tools/profile_round.sh has run again."""
    sys.path.insert(0, str(ROOT))
    from dwarf_bench_amd.build import kernel_tree_sha256
    now = kernel_tree_sha256()
    traffic = json.loads((ROOT / "profiles" / "hbm_traffic.json").read_text())
    assert traffic["_source"]["kernel_tree_sha256"] == now, "device code changed since the PMC passes: run tools/profile_round.sh"
    assert json.loads(_newest("r*_headline.json").read_text())["kernel_tree_sha256"] == now
    line = json.loads(_newest("r*_bench_n1.json").read_text())
    src = line["roofline"]["traffic_source"]
    assert src["collected_on_this_kernel_tree"] is True and src["kernel_tree_sha256_now"] == now
