"""numpy front door to the CPU oracle (oracle/dbo.c) — TEST INFRASTRUCTURE ONLY.

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  The product package
(dwarf_bench_amd) never imports this module: it has no CPU path.
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

_DIR = Path(__file__).resolve().parent
_LIB = _DIR / "_build" / "liboracle.so"
_REF = _DIR / "_ref" / "libdbref.so"

_sz, _u32, _i32, _u64, _int, _vp = C.c_size_t, C.c_uint32, C.c_int32, C.c_uint64, C.c_int, C.c_void_p


def _build() -> None:
    subprocess.run(["make", "-s", "-C", str(_DIR), "_build/liboracle.so"], check=True)


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not _LIB.exists():
            _build()
        L = C.CDLL(str(_LIB))
        L.dbo_mix64.restype = _u64
        L.dbo_mix64.argtypes = [_u64, _u64]
        L.dbo_copy_if_lt_i32.restype = _sz
        L.dbo_chunked_scan_i32.restype = _sz
        L.dbo_count_distinct_u32.restype = _sz
        L.dbo_seq_join_u32.restype = _sz
        L.dbo_polynomial_hash.restype = _u32
        L.dbo_polynomial_hash.argtypes = [_u32, _int, _sz]
        L.dbo_simple_hash.restype = _u32
        L.dbo_simple_hash.argtypes = [_u32, _sz]
        L.dbo_murmur3_x86_32.restype = _u32
        L.dbo_murmur3_x86_32.argtypes = [_u32, _u32]
        L.dbo_bitmask_table_insert.restype = _u32
        _lib = L
    return _lib


def ref_lib():
    """oracle/_ref/libdbref.so: the reference sources that compile with plain g++ (None if not built)."""
    if not _REF.exists():
        return None
    R = C.CDLL(str(_REF))
    R.ref_murmur3_x86_32.restype = _u32
    R.ref_murmur3_x86_32.argtypes = [_u32, _u32, _u64]
    R.ref_simple_hash.restype = _u32
    R.ref_simple_hash.argtypes = [_u32, _u64]
    R.ref_polynomial_hash.restype = _u32
    R.ref_polynomial_hash.argtypes = [_u32, _int, _u64]
    R.ref_seq_join.restype = _sz
    return R


def _p(a: np.ndarray):
    return a.ctypes.data_as(_vp)


def _u32a(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.uint32)


def _i32a(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.int32)


# ---- data ---------------------------------------------------------------------------------------
def mix64(seed: int, i: int) -> int:
    return lib().dbo_mix64(seed, i)


def gen_uniform_u32(n: int, seed: int, lo: int, hi: int, first_index: int = 0) -> np.ndarray:
    out = np.empty(n, dtype=np.uint32)
    lib().dbo_gen_uniform_u32(_p(out), _sz(n), _u64(seed), _u64(first_index), _u32(lo), _u32(hi))
    return out


def gen_unique_sorted_u32(n: int, seed: int, first_index: int = 0) -> np.ndarray:
    out = np.empty(n, dtype=np.uint32)
    lib().dbo_gen_unique_sorted_u32(_p(out), _sz(n), _u64(seed), _u64(first_index))
    return out


# ---- scan ---------------------------------------------------------------------------------------
def copy_if_lt(src, filter_value: int) -> np.ndarray:
    src = _i32a(src)
    out = np.empty(max(src.size, 1), dtype=np.int32)
    k = lib().dbo_copy_if_lt_i32(_p(src), _sz(src.size), _i32(filter_value), _p(out))
    return out[:k].copy()


def two_pass_scan(src, filter_value: int, tnum: int = 8, threads: int = 1):
    """scan.cl restated: returns (out[:out_size], out_size, prefix); drops the n % tnum tail like the reference."""
    src = _i32a(src)
    out = np.full(max(src.size, 1), -1, dtype=np.int32)
    prefix = np.zeros(tnum + 1, dtype=np.int32)
    osz = C.c_int32(-1)
    lib().dbo_two_pass_scan_i32(_p(src), _sz(src.size), _i32(filter_value), _p(out), C.byref(osz), _p(prefix),
                                _int(tnum), _int(threads))
    return out[: osz.value].copy(), osz.value, prefix


def chunked_scan(src, filter_value: int, threads: int, out: np.ndarray | None = None):
    src = _i32a(src)
    if out is None:
        out = np.empty(max(src.size, 1), dtype=np.int32)
    k = lib().dbo_chunked_scan_i32(_p(src), _sz(src.size), _i32(filter_value), _p(out), _int(threads))
    return out[:k], k


def prefix_sum_exclusive(v) -> np.ndarray:
    v = _i32a(v)
    out = np.empty_like(v)
    lib().dbo_prefix_sum_exclusive_i32(_p(v), _sz(v.size), _p(out))
    return out


# ---- sort ---------------------------------------------------------------------------------------
def sort_i32(keys) -> np.ndarray:
    k = _i32a(keys).copy()
    lib().dbo_sort_i32(_p(k), _sz(k.size))
    return k


def sort_u32(keys) -> np.ndarray:
    k = _u32a(keys).copy()
    lib().dbo_sort_u32(_p(k), _sz(k.size))
    return k


def radix_sort_u32_mt(keys: np.ndarray, tmp: np.ndarray, threads: int) -> None:
    """in place (CPU baseline)."""
    lib().dbo_radix_sort_u32_mt(_p(keys), _p(tmp), _sz(keys.size), _int(threads))


# ---- hashers ------------------------------------------------------------------------------------
def polynomial_hash(v: int, p: int, sz: int) -> int:
    return lib().dbo_polynomial_hash(v, p, sz)


def simple_hash(v: int, sz: int) -> int:
    return lib().dbo_simple_hash(v, sz)


def murmur3_x86_32(key: int, seed: int) -> int:
    return lib().dbo_murmur3_x86_32(key, seed)


# ---- group-by -------------------------------------------------------------------------------------
def groupby_sum(keys, vals, groups: int) -> np.ndarray:
    keys, vals = _u32a(keys), _u32a(vals)
    out = np.zeros(max(groups, 1), dtype=np.uint32)
    lib().dbo_groupby_sum_u32(_p(keys), _p(vals), _sz(keys.size), _u32(groups), _p(out))
    return out[:groups]


def groupby_hash(keys, vals, groups: int, table_size: int | None = None, p: int = 31, threads: int = 1) -> np.ndarray:
    keys, vals = _u32a(keys), _u32a(vals)
    out = np.zeros(max(groups, 1), dtype=np.uint32)
    ts = keys.size if table_size is None else table_size  # groupby.cpp:48-49: capacity = buf_size
    rc = lib().dbo_groupby_hash_u32(_p(keys), _p(vals), _sz(keys.size), _u32(groups), _sz(max(ts, 1)), _int(p),
                                    _p(out), _int(threads))
    if rc != 0:
        raise RuntimeError("oracle group-by table full")
    return out[:groups]


def groupby_local(keys, vals, groups: int, executors: int, threads: int = 1) -> np.ndarray:
    keys, vals = _u32a(keys), _u32a(vals)
    out = np.zeros(max(groups, 1), dtype=np.uint32)
    lib().dbo_groupby_local_u32(_p(keys), _p(vals), _sz(keys.size), _u32(groups), _sz(executors), _p(out),
                                _int(threads))
    return out[:groups]


# ---- one-to-many join ---------------------------------------------------------------------------
class _JoinTable(C.Structure):
    _fields_ = [("ht_size", _sz), ("ht", _vp), ("cnt", _vp), ("pos", _vp), ("ids", _vp), ("n_build", _sz)]


def count_distinct(v) -> int:
    v = _u32a(v)
    return lib().dbo_count_distinct_u32(_p(v), _sz(v.size))


def join_omnisci(build, probe, threads: int = 1):
    """OmniSci table restated: returns (pos, cnt, ids) with ht_size = 2*distinct(build) (join_omnisci.cpp:69)."""
    build, probe = _u32a(build), _u32a(probe)
    ht_size = max(2 * count_distinct(build), 1)
    t = _JoinTable()
    if lib().dbo_join_build(C.byref(t), _p(build), _sz(build.size), _sz(ht_size), _int(threads)) != 0:
        raise MemoryError
    pos = np.zeros(max(probe.size, 1), dtype=np.uint64)
    cnt = np.zeros(max(probe.size, 1), dtype=np.uint64)
    lib().dbo_join_probe(C.byref(t), _p(probe), _sz(probe.size), _p(pos), _p(cnt), _int(threads))
    ids = np.ctypeslib.as_array(C.cast(t.ids, C.POINTER(C.c_uint64)), shape=(max(build.size, 1),)).copy()
    lib().dbo_join_free(C.byref(t))
    return pos[: probe.size], cnt[: probe.size], ids[: build.size]


def join_omnisci_timings(build, probe, threads: int = 1):
    """(build seconds, probe seconds, matches) of the restated OmniSci table — what join_omnisci.cpp:78-95 times;
    ht_size = 2*distinct(build) is computed outside the timed part, as in the reference (:69)."""
    import time
    build, probe = _u32a(build), _u32a(probe)
    ht_size = max(2 * count_distinct(build), 1)
    t = _JoinTable()
    t0 = time.perf_counter()
    if lib().dbo_join_build(C.byref(t), _p(build), _sz(build.size), _sz(ht_size), _int(threads)) != 0:
        raise MemoryError
    t1 = time.perf_counter()
    pos = np.zeros(max(probe.size, 1), dtype=np.uint64)
    cnt = np.zeros(max(probe.size, 1), dtype=np.uint64)
    t2 = time.perf_counter()
    lib().dbo_join_probe(C.byref(t), _p(probe), _sz(probe.size), _p(pos), _p(cnt), _int(threads))
    t3 = time.perf_counter()
    lib().dbo_join_free(C.byref(t))
    return t1 - t0, t3 - t2, int(cnt[: probe.size].sum())


def join_bruteforce(build, probe, want_ids: bool = True):
    """join_omnisci.cpp:15-29: per probe row the count and ascending build ids."""
    a, b = _u32a(build), _u32a(probe)
    cnt = np.zeros(max(b.size, 1), dtype=np.uint64)
    off = np.zeros(b.size + 1, dtype=np.uint64)
    lib().dbo_join_bruteforce(_p(a), _sz(a.size), _p(b), _sz(b.size), _p(cnt), _p(off), None)
    ids = None
    if want_ids:
        ids = np.zeros(max(int(off[-1]), 1), dtype=np.uint64)
        lib().dbo_join_bruteforce(_p(a), _sz(a.size), _p(b), _sz(b.size), _p(cnt), _p(off), _p(ids))
    return cnt[: b.size], off, ids


def join_counts_fast(build, probe) -> np.ndarray:
    """Same per-probe-row counts as join_bruteforce, via numpy (for sizes where O(n*m) is infeasible)."""
    a, b = _u32a(build), _u32a(probe)
    uk, uc = np.unique(a, return_counts=True)
    idx = np.searchsorted(uk, b)
    idx[idx >= uk.size] = 0
    hit = uk[idx] == b if uk.size else np.zeros(b.size, dtype=bool)
    return np.where(hit, uc[idx] if uk.size else 0, 0).astype(np.uint64)


# ---- unique-key payload join ----------------------------------------------------------------------
def seq_join(a_keys, a_vals, b_keys, b_vals):
    ak, av, bk, bv = map(_u32a, (a_keys, a_vals, b_keys, b_vals))
    n = lib().dbo_seq_join_u32(_p(ak), _p(av), _sz(ak.size), _p(bk), _p(bv), _sz(bk.size), None, None, None)
    ok, o1, o2 = (np.empty(max(n, 1), dtype=np.uint32) for _ in range(3))
    lib().dbo_seq_join_u32(_p(ak), _p(av), _sz(ak.size), _p(bk), _p(bv), _sz(bk.size), _p(ok), _p(o1), _p(o2))
    return ok[:n], o1[:n], o2[:n]


def reduce_sum(src) -> int:
    a = _i32a(src)
    f = lib().dbo_reduce_sum_i32
    f.restype = C.c_int32
    return int(f(_p(a), _sz(a.size)))


def nested_join(a_keys, a_vals, b_keys, b_vals):
    """dense (na x nb) cell matrices, reference markers in the empty cells"""
    ak, av, bk, bv = map(_u32a, (a_keys, a_vals, b_keys, b_vals))
    outs = [np.empty(max(ak.size * bk.size, 1), dtype=np.uint32) for _ in range(3)]
    f = lib().dbo_nested_join_u32
    f.restype = None
    f(_p(ak), _p(av), _sz(ak.size), _p(bk), _p(bv), _sz(bk.size), *map(_p, outs))
    return tuple(o[: ak.size * bk.size].reshape(ak.size, bk.size) for o in outs)


def ujoin(a_keys, a_vals, b_keys, b_vals, seed: int = 7):
    """join.cpp:60-131 through the bitmask table: per-probe-row outputs with 0xFFFFFFFF sentinels."""
    ak, av, bk, bv = map(_u32a, (a_keys, a_vals, b_keys, b_vals))
    ok, o1, o2 = (np.empty(max(bk.size, 1), dtype=np.uint32) for _ in range(3))
    rc = lib().dbo_ujoin_u32(_p(ak), _p(av), _sz(ak.size), _p(bk), _p(bv), _sz(bk.size), _u32(seed), _p(ok), _p(o1),
                             _p(o2))
    if rc != 0:
        raise MemoryError
    return ok[: bk.size], o1[: bk.size], o2[: bk.size]


class _BitmaskTable(C.Structure):
    _fields_ = [("size", _sz), ("bitmask_sz", _sz), ("keys", _vp), ("vals", _vp), ("bitmask", _vp),
                ("hash_kind", _int), ("seed", _u32)]


class BitmaskTable:
    """SimpleNonOwningHashTable restated (hashtable.hpp:5-93)."""

    def __init__(self, size: int, hash_kind: int = 0, seed: int = 0):
        self.t = _BitmaskTable()
        self.size = size
        if lib().dbo_bitmask_table_init(C.byref(self.t), _sz(size), _int(hash_kind), _u32(seed)) != 0:
            raise MemoryError

    def insert(self, key: int, val: int) -> int:
        return lib().dbo_bitmask_table_insert(C.byref(self.t), _u32(key), _u32(val))

    def at(self, key: int):
        v = _u32(0)
        ok = lib().dbo_bitmask_table_at(C.byref(self.t), _u32(key), C.byref(v))
        return (v.value, True) if ok else (0, False)

    def data(self) -> np.ndarray:
        return np.ctypeslib.as_array(C.cast(self.t.vals, C.POINTER(C.c_uint32)), shape=(self.size,)).copy()

    def close(self):
        if self.t.keys:
            lib().dbo_bitmask_table_free(C.byref(self.t))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
