"""In-tree build of the native pieces (no cmake, no network):

  dwarf_bench_amd/_lib/libdbhip.so     hand-written gfx950 HIP kernels + the C ABI of include/dbhip.h
  dwarf_bench_amd/_lib/libdbench.so    C++ host layer mirroring the reference's Dwarf/Meter/Registry/bench API
  dwarf_bench_amd/_lib/dwarf_bench     CLI (same flags as the reference's main.cpp)
  oracle/_build/liboracle.so           CPU restatement of the reference algorithms (test infrastructure only)
  oracle/_ref/*                        the few reference sources that compile with plain g++ (only when
                                       /root/reference is present; never shipped as source)

`python -m dwarf_bench_amd.build` builds everything that is out of date.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
PKG = ROOT / "dwarf_bench_amd"
CSRC = PKG / "csrc"
HOST = PKG / "host"
LIB = PKG / "_lib"
ORACLE = ROOT / "oracle"
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (set HIPCC=...)")


def _stale(target: Path, sources: list[Path]) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(s.stat().st_mtime > t for s in sources if s.exists())


def _run(cmd: list[str], cwd: Path | None = None) -> None:
    print("[build]", " ".join(str(c) for c in cmd), flush=True)
    subprocess.run([str(c) for c in cmd], check=True, cwd=cwd)


def kernel_tree_sha256() -> str:
    """Content hash of the device CODE: every file under csrc/ plus include/dbhip.h, by name and by its text with comments
    and blank space removed (a reworded comment is not a different kernel).  The profiler passes that bench.py's
    `roofline.traffic` comes from record it (tools/profile_round.sh -> profiles/hbm_traffic.json), bench.py reports whether
    it still matches, and tests/test_bench_contract.py fails when a kernel changed after the counters were collected.
    (A content hash rather than `git log -1 -- csrc`: the GPU box has no .git.)"""
    import hashlib
    import re
    h = hashlib.sha256()
    for f in sorted(CSRC.glob("*.hip")) + sorted(CSRC.glob("*.hpp")) + [ROOT / "include" / "dbhip.h"]:
        text = f.read_text()
        text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)   # block comments
        text = re.sub(r"//[^\n]*", " ", text)                  # line comments (no string literal here holds "//")
        text = re.sub(r"\s+", " ", text).strip()
        h.update(f.name.encode())
        h.update(b"\0")
        h.update(text.encode())
        h.update(b"\0")
    return h.hexdigest()


def build_hip(force: bool = False) -> Path:
    """hipcc --offload-arch=gfx950: every .hip under csrc/ into one shared library."""
    LIB.mkdir(exist_ok=True)
    out = LIB / "libdbhip.so"
    srcs = sorted(CSRC.glob("*.hip"))
    deps = srcs + sorted(CSRC.glob("*.hpp")) + [ROOT / "include" / "dbhip.h"]
    if force or _stale(out, deps):
        objs = []
        odir = LIB / "obj"
        odir.mkdir(exist_ok=True)
        for s in srcs:
            o = odir / (s.stem + ".o")
            if force or _stale(o, [s] + sorted(CSRC.glob("*.hpp")) + [ROOT / "include" / "dbhip.h"]):
                _run([_hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc",
                      "-Wall", "-Wno-unused-function"] + os.environ.get("DBHIP_EXTRA_FLAGS", "").split()
                     + ["-c", s, "-o", o])
            objs.append(o)
        _run([_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", out] + objs)
    return out


def build_oracle(force: bool = False) -> Path:
    """The CPU restatement (plain C, gcc).  Test infrastructure: never linked into the product."""
    bdir = ORACLE / "_build"
    bdir.mkdir(exist_ok=True)
    out = bdir / "liboracle.so"
    srcs = sorted(ORACLE.glob("*.c"))
    deps = srcs + sorted(ORACLE.glob("*.h"))
    if force or _stale(out, deps):
        _run(["gcc", "-O3", "-std=c11", "-fPIC", "-shared", "-pthread", "-Wall",
              "-o", out] + srcs + ["-lm"])
    return out


def build_ref(force: bool = False) -> Path | None:
    """Compile the reference pieces that build with plain g++ from where they lie (oracle/Makefile)."""
    if not Path("/root/reference").exists() or not (ORACLE / "Makefile").exists():
        return None
    _run(["make", "-s", "-C", ORACLE, "ref"] + (["-B"] if force else []))
    return ORACLE / "_ref"


def build_host(force: bool = False) -> Path | None:
    """C++ host layer + CLI (g++ for host-only files, hipcc to link against libdbhip/libamdhip64)."""
    if not (HOST / "Makefile").exists():
        return None
    _run(["make", "-s", "-C", HOST] + (["-B"] if force else []))
    return LIB / "libdbench.so"


def build_all(force: bool = False) -> None:
    build_hip(force)
    build_oracle(force)
    build_ref(force)
    build_host(force)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
