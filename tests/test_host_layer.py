"""C++ host layer (dwarf_bench_amd/host): the reference's plugin frame re-created.  CPU only.
Compared against outputs of the REFERENCE's own common/{result,options}.cpp (tests/golden/ref_vectors.json)."""
import json
import os
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
HOST = ROOT / "dwarf_bench_amd" / "host"
LIB = ROOT / "dwarf_bench_amd" / "_lib"


@pytest.fixture(scope="module")
def selftest(tmp_path_factory):
    exe = tmp_path_factory.mktemp("host") / "host_selftest"
    subprocess.run(["g++", "-std=c++17", "-O1", "-I", str(HOST), str(ROOT / "tests/cpp/host_selftest.cpp"),
                    str(HOST / "dwarf_api.cpp"), "-DSELFTEST_NO_REGISTRY", "-o", str(exe),
                    "-Wl,--unresolved-symbols=ignore-all"], check=True)
    return exe


@pytest.fixture(scope="module")
def refv(golden_dir):
    return json.loads((golden_dir / "ref_vectors.json").read_text())


def test_csv_and_printing_match_reference_code(selftest, refv, tmp_path):
    for i, case in enumerate(refv["csv"]):
        path = tmp_path / f"r{i}.csv"
        args = [str(selftest), "csv", str(path), case["dwarf"], case["device_type"].lower(), case["header"] or "-",
                str(len(case["kinds"]))]
        for kind, size, t in zip(case["kinds"], case["buf_sizes"], case["times_us"]):
            args += [str(kind), str(size)] + [repr(float(x)) for x in t]
        printed = ""
        for _ in range(2):  # second run appends without a header (common/result.cpp:60-66)
            printed = subprocess.run(args, check=True, capture_output=True, text=True).stdout
        assert path.read_text() == case["csv_after_two_writes"]
        assert printed == case["printed"]


def test_device_type_parse_print(selftest, refv):
    for case in refv["device_type"]:
        out = subprocess.run([str(selftest), "devtype", case["in"]], check=True, capture_output=True, text=True).stdout.split()
        if case["in"].lower() == "hip":  # the one addition
            assert out == ["4", "HIP"]
        else:
            assert out == [str(case["enum"]), case["to_string"]]


def test_registry_semantics(selftest):
    out = subprocess.run([str(selftest), "registry"], check=True, capture_output=True, text=True).stdout.split()
    assert out == ["111", "2"]


def _cli():
    exe = LIB / "dwarf_bench"
    if not exe.exists():
        from dwarf_bench_amd import build
        build.build_hip()
        build.build_host()
    return exe


def test_cli_list_and_exit_codes():
    exe = _cli()
    r = subprocess.run([str(exe), "list"], capture_output=True, text=True)
    assert r.returncode == 0
    names = [l.strip() for l in r.stdout.splitlines() if l.startswith("\t")]
    assert names == sorted(["DPLScanHip", "GroupByHip", "GroupByLocalHip", "HashBuildHip", "HashBuildNonBitmaskHip", "JoinHip",
                            "JoinOmnisciHip", "NestedLoopJoinHip", "PartitionedJoinHip", "ProbeHip", "RadixHip", "ReduceHip",
                            "TwoPassScanHip"])
    assert "DWARF_BENCH_ROOT is set to" in r.stdout
    r = subprocess.run([str(exe), "NoSuchDwarf"], capture_output=True, text=True)
    assert r.returncode == 1 and "List supported dwarfs" in r.stderr  # main.cpp:75-79
    r = subprocess.run([str(exe), "--help", "RadixHip"], capture_output=True, text=True)
    assert r.returncode == 0 and "--input_size" in r.stdout and "--groups_count" in r.stdout
    env = dict(os.environ, DWARF_BENCH_ROOT="/some/where")
    r = subprocess.run([str(exe), "list"], capture_output=True, text=True, env=env)
    assert "DWARF_BENCH_ROOT is set to /some/where" in r.stdout


@pytest.mark.skipif(not Path("/root/reference/example/bench_usage/main.cpp").exists(), reason="reference not mounted")
def test_reference_usage_example_compiles_against_our_header():
    """example/bench_usage/main.cpp:4-33 must compile unchanged against this build's <bench.hpp>."""
    subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-I", str(HOST), "/root/reference/example/bench_usage/main.cpp"],
                   check=True)
