"""The C-ABI library loads on a CPU-only box and exports every entry point include/dbhip.h declares
(no compute calls here: there is no GPU)."""
import ctypes
import re
from pathlib import Path

import pytest

from dwarf_bench_amd import _capi

ROOT = Path(__file__).resolve().parents[1]


def _declared():
    text = (ROOT / "include" / "dbhip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dbhip_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert _declared() == sorted(_capi.SIGNATURES)


def test_library_exports_every_declared_symbol():
    if not _capi.lib_path().exists():
        from dwarf_bench_amd import build
        build.build_hip()
    lib = _capi.lib()
    for name in _declared():
        assert hasattr(lib, name), name
    assert lib.dbhip_version() == 1


def test_workspace_queries_need_no_gpu():
    lib = _capi.lib()
    assert lib.dbhip_copy_if_lt_i32_workspace_bytes(1 << 28) >= 256 + (1 << 28) // 8192 * 8
    assert lib.dbhip_radix_sort_workspace_bytes(1 << 24, 8) % 256 == 0
    assert lib.dbhip_radix_sort_workspace_bytes(1 << 24, 5) == 0
    assert lib.dbhip_groupby_sum_u32_workspace_bytes(1 << 26, 1 << 16) >= 128 * (1 << 16) * 4
    assert lib.dbhip_join_workspace_bytes(1 << 20) >= 20 * (1 << 20)  # 12-byte table slots per row + 8-byte (key, row id) pairs
    assert lib.dbhip_ujoin_workspace_bytes(1000) >= 2 * 2048 * 4


def test_argument_errors_are_reported_without_a_device():
    lib = _capi.lib()
    # null pointers / bad sizes are rejected on the host before any HIP call
    assert lib.dbhip_copy_if_lt_i32(None, 16, 5, None, None, None, 0, None) == -1
    assert lib.dbhip_radix_sort_u32(None, None, 16, 7, None, 0, None) == -1
    assert lib.dbhip_groupby_sum_u32(None, None, 16, 4, None, None, 0, None) == -1


def test_product_has_no_oracle_import():
    """The product package must never import oracle/ (no CPU fallback)."""
    for py in (ROOT / "dwarf_bench_amd").rglob("*.py"):
        src = py.read_text()
        assert "pyoracle" not in src and "import oracle" not in src and "from oracle" not in src, py


def test_join_partition_geometry_for_every_row_count(tmp_path):
    """join_common.hpp jl_layout over row counts up to 2^31, for the build's and the radix join's rows per partition:
    at most 1024 level-0 buckets, a power-of-two level-1 fan-out, partitions that hold their rows (tests/cpp/
    join_layout_check.cpp; a geometry with 1171 level-0 buckets at 2^30 rows once made that join take a minute)"""
    import subprocess
    exe = tmp_path / "join_layout_check"
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O1", "-std=c++17", "-w", "-I", str(ROOT / "dwarf_bench_amd" / "csrc"),
                    str(ROOT / "tests" / "cpp" / "join_layout_check.cpp"), "-o", str(exe)], check=True, timeout=600)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "join layout ok" in r.stdout, r.stdout[-2000:]
