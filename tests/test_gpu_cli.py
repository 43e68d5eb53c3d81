"""End-to-end through the C++ host layer on the GPU: the `dwarf_bench` CLI and the `dbench` library example.
Mirrors the reference's dwarf tests (tests/dwarf_tests/dwarf_tests.cpp:12-88): every dwarf x sizes
{128..4096} x 10 iterations, every result valid (an invalid result prints "ncorrect results" on stderr)."""
import csv
import subprocess
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
LIB = Path(__file__).resolve().parents[1] / "dwarf_bench_amd" / "_lib"
SIZES = ["128", "256", "512", "1024", "2048", "4096"]


def _run(args, **kw):
    return subprocess.run([str(LIB / "dwarf_bench")] + args, capture_output=True, text=True, timeout=300, **kw)


@pytest.mark.parametrize("dwarf", ["TwoPassScanHip", "DPLScanHip", "RadixHip", "JoinOmnisciHip", "JoinHip",
                                   "HashBuildHip", "HashBuildNonBitmaskHip", "ProbeHip", "ReduceHip", "NestedLoopJoinHip"])
def test_dwarf_suite(dwarf):
    r = _run([dwarf, "--device=hip", "--iterations", "10", "--input_size"] + SIZES)
    assert r.returncode == 0, r.stderr
    assert "ncorrect results" not in r.stderr and "Caught exception" not in r.stderr, r.stderr
    assert r.stdout.count("Host duration:") == 10 * len(SIZES)
    if dwarf in ("JoinOmnisciHip", "JoinHip"):
        assert r.stdout.count("Build time:") == 10 * len(SIZES)


def test_groupby_suite_with_reference_test_options():
    """tests/dwarf_tests/utils.cpp:39-47: groups_count=64, executors=1024"""
    r = _run(["GroupByHip", "--device=hip", "--iterations", "10", "--groups_count", "64", "--executors", "1024",
              "--input_size"] + SIZES)
    assert r.returncode == 0 and "ncorrect results" not in r.stderr and "Caught exception" not in r.stderr, r.stderr
    assert r.stdout.count("Host duration:") == 10 * len(SIZES)


@pytest.mark.parametrize("executors", ["1", "16", "1024"])
def test_groupby_local_suite(tmp_path, executors):
    """GroupByLocal (groupby/groupby_local.cpp:24-142): two-phase timings and its own CSV header (:138)"""
    rep = tmp_path / "gbl.csv"
    r = _run(["GroupByLocalHip", "--device=hip", "--iterations", "10", "--groups_count", "64", "--executors", executors,
              f"--report_path={rep}", "--input_size"] + SIZES + ["1048576"])
    assert r.returncode == 0 and "ncorrect results" not in r.stderr and "Caught exception" not in r.stderr, r.stderr
    rows = list(csv.reader(rep.open()))
    assert rows[0] == ["device_type", "buf_size_bytes", "total_time", "group_by_time", "reduction_time"]
    assert len(rows) == 1 + 10 * (len(SIZES) + 1)
    for row in rows[1:]:
        total, gb, red = (float(x) for x in row[2:5])
        assert row[0] == "HIP" and gb > 0 and red >= 0 and abs(total - (gb + red)) <= 0.0025  # whole-us truncation


@pytest.mark.parametrize("gpus,env", [("1", {}), ("1", {"DWARF_BENCH_PJOIN_EXCHANGE": "copy"}), ("2", {}), ("3", {}),
                                      ("8", {})])
def test_partitioned_join_dwarf(gpus, env):
    """PartitionedJoinHip through the CLI.  One GPU on the test box: --gpus 1 runs the RCCL send/recv
    group against itself (or the peer-copy exchange when forced), more ranks than GPUs share the device and
    exchange by hipMemcpyPeerAsync.  Sizes include ragged shards (n % P != 0) and n < P."""
    import os
    r = _run(["PartitionedJoinHip", "--device=hip", "--gpus", gpus, "--iterations", "3", "--input_size", "5", "1000",
              "65536", "300007", "2097152"], env={**os.environ, **env})
    assert r.returncode == 0, r.stderr
    assert "ncorrect results" not in r.stderr and "Caught exception" not in r.stderr, r.stderr
    assert r.stdout.count("Build time:") == 3 * 5 and r.stdout.count("Exchange time:") == 3 * 5
    assert r.stdout.count("Local probe time:") == 3 * 5
    assert f"{gpus} rank(s) on 1 GPU(s)" in r.stdout
    want = "RCCL" if (gpus == "1" and not env) else "hipMemcpyPeerAsync"
    assert f"exchange by {want}" in r.stdout


@pytest.mark.parametrize("sub_joins", ["1", "2", "4"])
@pytest.mark.parametrize("gpus", ["1", "3", "8"])
def test_partitioned_join_pipelined_sub_joins(gpus, sub_joins):
    """The engine cuts a step into 1, 2 (default) or 4 independent sub-joins by more bits of the rank hash — their
    exchanges follow each other on the links while the sub-join before is joined locally (host/pjoin_engine.hpp
    Options::sub_joins).  Every variant must deliver every probe row once with the right count and ids that carry its key
    (host-side check at these sizes, on top of conservation / generator / routing-to-rank-AND-sub-join on the device),
    ragged shards and n < ranks * sub-joins included."""
    import os
    r = _run(["PartitionedJoinHip", "--device=hip", "--gpus", gpus, "--iterations", "2", "--input_size", "7", "1000", "300007",
              "2097152"], env={**os.environ, "DWARF_BENCH_PJOIN_SUBJOINS": sub_joins})
    assert r.returncode == 0, r.stderr
    assert "ncorrect results" not in r.stderr and "Caught exception" not in r.stderr, r.stderr
    assert r.stdout.count("Build time:") == 2 * 4
    assert (f"{sub_joins} pipelined sub-joins" in r.stdout) == (sub_joins != "1")


def test_baseline_plumbing_config_csv(tmp_path):
    """BASELINE configs[0] on the HIP device: TwoPassScan --input_size=1024 --iterations=9 -> 9 valid rows,
    reference CSV schema (common/result.cpp:59-91), appended on a second run."""
    rep = tmp_path / "report.csv"
    for _ in range(2):
        r = _run(["TwoPassScanHip", "--device=hip", "--input_size=1024", "--iterations=9", f"--report_path={rep}"])
        assert r.returncode == 0 and "ncorrect results" not in r.stderr, r.stderr
    rows = list(csv.reader(rep.open()))
    assert rows[0] == ["device_type", "buf_size_bytes", "host_time_ms", "kernel_time_ms"]
    assert len(rows) == 1 + 18
    for row in rows[1:]:
        assert row[0] == "HIP" and row[1] == "4096" and float(row[2]) >= 0 and float(row[3]) >= 0


def test_large_sizes_through_cli():
    r = _run(["TwoPassScanHip", "--device=hip", "--iterations", "3", "--input_size", "16777216"])
    assert r.returncode == 0 and "ncorrect results" not in r.stderr and "Caught exception" not in r.stderr, r.stderr
    r = _run(["RadixHip", "--device=hip", "--iterations", "3", "--input_size", "1000003"])
    assert r.returncode == 0 and "ncorrect results" not in r.stderr and "Caught exception" not in r.stderr, r.stderr
    r = _run(["GroupByHip", "--device=hip", "--iterations", "3", "--groups_count", "65536", "--input_size", "4194304"])
    assert r.returncode == 0 and "ncorrect results" not in r.stderr and "Caught exception" not in r.stderr, r.stderr


def test_library_usage_example():
    """the library API (DwarfBench::makeMeasurements with DeviceType::HIP) through examples/hip_library_usage.cpp"""
    r = subprocess.run([str(LIB / "hip_library_usage")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.splitlines() if "RESULT:" in l]
    assert len(lines) == 4 * 10  # 4 dwarfs x 10 iterations
    assert all(l.split()[3] == "1024" for l in lines)
    assert "ncorrect results" not in r.stderr


# ---- Result::valid at BASELINE sizes: device-side validators, and proof that they can fail ------------------------
BIG = [("TwoPassScanHip", "268435456", []), ("RadixHip", "16777216", []),
       ("GroupByHip", "67108864", ["--groups_count", "65536"]), ("JoinOmnisciHip", "67108864", []),
       ("JoinHip", "67108864", []), ("ProbeHip", "67108864", []), ("ReduceHip", "268435456", [])]


@pytest.mark.parametrize("dwarf,size,extra", BIG)
def test_valid_is_computed_at_baseline_sizes(dwarf, size, extra):
    """scan/scan.cpp:157-164 validates every iteration at every size; above DWARF_BENCH_VALIDATE_MAX the HIP dwarfs
    do it with the device-side validators (forced here for every size above 2^20 so the sort is covered as well)"""
    import os
    env = {**os.environ, "DWARF_BENCH_VALIDATE_MAX": "1048576"}
    r = _run([dwarf, "--device=hip", "--iterations", "2", "--input_size", size] + extra, env=env)
    assert r.returncode == 0, r.stderr
    assert "ncorrect results" not in r.stderr and "Caught exception" not in r.stderr, r.stderr
    assert r.stdout.count("Host duration:") == 2


@pytest.mark.parametrize("limit", ["1048576", "1"])  # host-side checks / device-side validators
@pytest.mark.parametrize("dwarf,extra", [("TwoPassScanHip", []), ("DPLScanHip", []), ("RadixHip", []),
                                         ("GroupByHip", ["--groups_count", "64"]),
                                         ("GroupByLocalHip", ["--groups_count", "64", "--executors", "16"]),
                                         ("JoinOmnisciHip", []), ("JoinHip", []), ("ProbeHip", []), ("ReduceHip", [])])
def test_fault_injection_flips_valid(dwarf, extra, limit):
    """DWARF_BENCH_INJECT_FAULT=1 corrupts one word of every result before it is checked: every iteration must be
    reported invalid, by the host checks and by the device-side validators alike"""
    import os
    env = {**os.environ, "DWARF_BENCH_INJECT_FAULT": "1", "DWARF_BENCH_VALIDATE_MAX": limit}
    r = _run([dwarf, "--device=hip", "--iterations", "3", "--input_size", "262144"] + extra, env=env)
    assert r.returncode == 0, r.stderr
    assert r.stderr.count("ncorrect results") == 3, r.stderr
    env.pop("DWARF_BENCH_INJECT_FAULT")
    r = _run([dwarf, "--device=hip", "--iterations", "3", "--input_size", "262144"] + extra, env=env)
    assert r.returncode == 0 and "ncorrect results" not in r.stderr, r.stderr


@pytest.mark.parametrize("filt", ["1001", "5001", "10001"])
def test_scan_dwarf_dense_predicates(filt):
    """DWARF_BENCH_SCAN_FILTER moves the scan dwarfs off the reference's filter value 5: from the second iteration on a
    dense predicate runs dbhip_copy_if_lt_dense_i32; every iteration is validated (host check at 2^20, device-side
    fingerprint at 2^28)"""
    import os
    env = {**os.environ, "DWARF_BENCH_SCAN_FILTER": filt}
    r = _run(["TwoPassScanHip", "--device=hip", "--iterations", "4", "--input_size", "1048576", "268435456"], env=env)
    assert r.returncode == 0, r.stderr
    assert "ncorrect results" not in r.stderr and "Caught exception" not in r.stderr, r.stderr
    assert r.stdout.count("Host duration:") == 8
    env["DWARF_BENCH_INJECT_FAULT"] = "1"
    r = _run(["DPLScanHip", "--device=hip", "--iterations", "3", "--input_size", "268435456"], env=env)
    assert r.returncode == 0 and r.stderr.count("ncorrect results") == 3, r.stderr


def test_probe_dwarf_over_the_bitmask_table():
    """ProbeHip (probe/slab_probe.cpp:9-107) over the SimpleNonOwningHashTable counterpart"""
    import os
    r = _run(["ProbeHip", "--device=hip", "--iterations", "3", "--input_size"] + SIZES + ["1048576"],
             env={**os.environ, "DWARF_BENCH_PROBE_TABLE": "bitmask"})
    assert r.returncode == 0 and "ncorrect results" not in r.stderr and "Caught exception" not in r.stderr, r.stderr
    assert r.stdout.count("Host duration:") == 3 * (len(SIZES) + 1)


def test_scan_host_time_with_transfers(tmp_path):
    """DWARF_BENCH_TIME_TRANSFERS=1: host_time spans H2D of src + kernel + D2H of n ints (scan/scan.cpp:107-128), so
    it exceeds the resident-data figure by at least the PCIe time of 2 x 64 MiB"""
    import os

    def host_ms(env):
        rep = tmp_path / f"r{len(env)}.csv"
        r = _run(["TwoPassScanHip", "--device=hip", "--iterations", "5", "--input_size", "16777216", f"--report_path={rep}"],
                 env={**os.environ, **env})
        assert r.returncode == 0 and "ncorrect results" not in r.stderr, r.stderr
        rows = list(csv.reader(rep.open()))[1:]
        return sorted(float(x[2]) for x in rows)[len(rows) // 2], sorted(float(x[3]) for x in rows)[len(rows) // 2]

    resident, k0 = host_ms({})
    with_copies, k1 = host_ms({"DWARF_BENCH_TIME_TRANSFERS": "1"})
    assert with_copies > resident + 1.5  # 128 MiB over PCIe gen5 x16 >= 2 ms; pageable copies take longer
    assert k1 < 1.0 and k0 < 1.0  # kernel_time is the same small figure either way


def test_partitioned_join_rccl_messages_above_one_gib():
    """n = 2^28 + 12345 rows through --gpus 1: the rank's RCCL self send/recv carries every column (> 1 GiB) as three
    pieces of at most 2^27 elements.  A single ncclSend/ncclRecv above 1 GiB delivers garbage without an error (RCCL
    2.27.7; pinned in round 4, host/pjoin_engine.cpp kPiece): the
    always-on conservation check and the device-side pair / routing / count checks must stay silent.  The log is
    kept under gpurun_out/ (a copy is committed under profiles/)."""
    r = _run(["PartitionedJoinHip", "--device=hip", "--gpus", "1", "--iterations", "1", "--input_size", "268447801"])
    log = Path(__file__).resolve().parents[1] / "gpurun_out"
    log.mkdir(exist_ok=True)
    (log / "pjoin_rccl_two_pieces.txt").write_text(r.stdout + "\n--- stderr ---\n" + r.stderr)
    assert r.returncode == 0, r.stderr
    assert "exchange by RCCL" in r.stdout
    assert "ncorrect results" not in r.stderr and "Caught exception" not in r.stderr, r.stderr
    assert r.stdout.count("Host duration:") == 1


@pytest.mark.parametrize("gpus,env", [("1", {}), ("4", {}), ("1", {"DWARF_BENCH_PJOIN_DIRECT": "1"})])
def test_partitioned_join_fault_injection(gpus, env):
    import os
    r = _run(["PartitionedJoinHip", "--device=hip", "--gpus", gpus, "--iterations", "2", "--input_size", "300007"],
             env={**os.environ, **env, "DWARF_BENCH_INJECT_FAULT": "1"})
    assert r.returncode == 0, r.stderr
    assert r.stderr.count("ncorrect results") >= 2, r.stderr
    r = _run(["PartitionedJoinHip", "--device=hip", "--gpus", gpus, "--iterations", "2", "--input_size", "300007"],
             env={**os.environ, **env})
    assert r.returncode == 0 and "ncorrect results" not in r.stderr, r.stderr


@pytest.mark.parametrize("gpus", ["1", "3"])
def test_partitioned_join_refuses_a_damaged_count_matrix(gpus):
    """The gathered P x P count matrix decides every address and length of the exchange.  Before anything is queued the
    engine checks it against what the host knows without it (row q sums to rank q's shard size): with one cell off by
    one (DWARF_BENCH_PJOIN_CORRUPT_MATRIX=1, from the first timed step on) the step throws, the CLI reports the
    exception like the reference's main does (main.cpp:97-100: message on stderr, exit code 0) and nothing is written
    through a wrong offset — the next, clean run of the same shape is valid."""
    import os
    args = ["PartitionedJoinHip", "--device=hip", "--gpus", gpus, "--iterations", "2", "--input_size", "300007"]
    r = _run(args, env={**os.environ, "DWARF_BENCH_PJOIN_CORRUPT_MATRIX": "1"})
    assert r.returncode == 0, r.stderr
    assert "Caught exception" in r.stderr and "gathered a count matrix whose row" in r.stderr, r.stderr
    assert r.stdout.count("Host duration:") == 0
    r = _run(args)
    assert r.returncode == 0 and "Caught exception" not in r.stderr and "ncorrect results" not in r.stderr, r.stderr


def test_partitioned_join_above_the_host_check_limit():
    """2^24 rows over 8 ranks sharing the GPU with the host-side check switched off: Result::valid rests on the
    device-side checks alone (what a 2^30-row run relies on)"""
    import os
    r = _run(["PartitionedJoinHip", "--device=hip", "--gpus", "8", "--iterations", "2", "--input_size", "16777216"],
             env={**os.environ, "DWARF_BENCH_VALIDATE_MAX": "1"})
    assert r.returncode == 0 and "ncorrect results" not in r.stderr and "Caught exception" not in r.stderr, r.stderr


@pytest.mark.parametrize("gpus,tag,extra_env", [("8", "8_virtual_ranks", {}), ("1", "direct_one_gpu", {"DWARF_BENCH_PJOIN_DIRECT": "1"}),
                                                ("1", "rccl_self_exchange", {})])
def test_partitioned_join_baseline_config_5(gpus, tag, extra_env):
    """BASELINE configs[4]: radix-partitioned hash join 2^30 x 2^30 (no reference counterpart).  The test box
    has one GPU: --gpus 8 runs the eight ranks of the target configuration as virtual ranks sharing the device (every
    rank partitions, exchanges and joins its 2^27-row shards — the per-rank work of the 8-GPU node, peer copies in
    place of xGMI), --gpus 1 with DWARF_BENCH_PJOIN_DIRECT=1 is the one-GPU point of the scaling curve (plain local radix
    join), --gpus 1 without it sends all 2^30 pairs of both relations through the RCCL send/recv group to the rank
    itself (eight pieces of 2^27 elements per column).  Host-side checks are off
    (DWARF_BENCH_VALIDATE_MAX=1): Result::valid rests on the device-side checks (conservation, generator, routing,
    per-row counts against the sorted build column).  The log is kept under gpurun_out/ (copies under profiles/)."""
    import os
    r = subprocess.run([str(LIB / "dwarf_bench"), "PartitionedJoinHip", "--device=hip", "--gpus", gpus, "--iterations", "1",
                        "--input_size", "1073741824"], capture_output=True, text=True, timeout=900,
                       env={**os.environ, "DWARF_BENCH_VALIDATE_MAX": "1", **extra_env})
    log = Path(__file__).resolve().parents[1] / "gpurun_out"
    log.mkdir(exist_ok=True)
    (log / f"pjoin_2p30_{tag}.txt").write_text(r.stdout + "\n--- stderr ---\n" + r.stderr)
    assert r.returncode == 0, r.stderr
    assert "ncorrect results" not in r.stderr and "Caught exception" not in r.stderr, r.stderr
    assert f"{gpus} rank(s) on 1 GPU(s)" in r.stdout
    for phase in ("Partition time:", "Exchange time:", "Local probe time:"):
        assert r.stdout.count(phase) >= 1, r.stdout
    assert r.stdout.count("Host duration:") == 1 and r.stdout.count("Build time:") == 1
