"""ctypes binding of libdbhip.so (include/dbhip.h).  Plumbing only: every compute call lands in the
hand-written gfx950 kernels.  There is NO fallback: if the library is missing or a symbol is absent the
import of the product path fails loudly."""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_LIB_PATH = Path(__file__).resolve().parent / "_lib" / "libdbhip.so"
if os.environ.get("DBHIP_LIB"):  # kernel experiments: an alternative build of the same library
    _LIB_PATH = Path(os.environ["DBHIP_LIB"])

c_u32p = C.POINTER(C.c_uint32)
_vp, _sz, _u64, _u32, _i32, _int = C.c_void_p, C.c_size_t, C.c_uint64, C.c_uint32, C.c_int32, C.c_int

# name -> (restype, argtypes); mirrors include/dbhip.h one to one (tests check the two stay in sync)
SIGNATURES = {
    "dbhip_version": (_int, []),
    "dbhip_device_info": (_int, [_int, C.c_char_p, _sz, C.POINTER(_int), C.POINTER(_int)]),
    "dbhip_workspace_status": (_int, [_vp, C.POINTER(_u32), _vp]),
    "dbhip_gen_uniform_u32": (_int, [_vp, _sz, _u64, _u64, _u32, _u32, _vp]),
    "dbhip_gen_uniform_at_u32": (_int, [_vp, _vp, _sz, _u64, _u32, _u32, _vp]),
    "dbhip_gen_unique_sorted_u32": (_int, [_vp, _sz, _u64, _u64, _vp]),
    "dbhip_copy_if_lt_i32_workspace_bytes": (_sz, [_sz]),
    "dbhip_copy_if_lt_i32": (_int, [_vp, _sz, _i32, _vp, _vp, _vp, _sz, _vp]),
    "dbhip_copy_if_lt_dense_i32": (_int, [_vp, _sz, _i32, _vp, _vp, _vp, _sz, _vp]),
    "dbhip_radix_sort_workspace_bytes": (_sz, [_sz, _int]),
    "dbhip_radix_sort_u32": (_int, [_vp, _vp, _sz, _int, _vp, _sz, _vp]),
    "dbhip_radix_sort_i32": (_int, [_vp, _vp, _sz, _int, _vp, _sz, _vp]),
    "dbhip_radix_sort_rank_mode": (_int, []),
    "dbhip_radix_sort_prepare": (_int, [_vp]),
    "dbhip_groupby_sum_u32_workspace_bytes": (_sz, [_sz, _u32]),
    "dbhip_groupby_sum_u32": (_int, [_vp, _vp, _sz, _u32, _vp, _vp, _sz, _vp]),
    "dbhip_groupby_partial_u32": (_int, [_vp, _vp, _sz, _u32, _u32, _vp, _sz, _vp]),
    "dbhip_groupby_merge_u32": (_int, [_u32, _u32, _vp, _vp, _vp]),
    "dbhip_join_workspace_bytes": (_sz, [_sz]),
    "dbhip_join_build_u32": (_int, [_vp, _sz, _vp, _vp, _sz, _vp]),
    "dbhip_join_build_pairs_u32": (_int, [_vp, _vp, _sz, _vp, _vp, _sz, _vp]),
    "dbhip_join_probe_u32": (_int, [_vp, _sz, _vp, _sz, _vp, _vp, _vp]),
    "dbhip_join_radix_workspace_bytes": (_sz, [_sz, _sz]),
    "dbhip_join_radix_partition_u32": (_int, [_int, _vp, _vp, _sz, _sz, _sz, _vp, _sz, _vp]),
    "dbhip_join_radix_match_u32": (_int, [_sz, _sz, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "dbhip_join_radix_u32": (_int, [_vp, _vp, _sz, _vp, _vp, _sz, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "dbhip_join_answers_u32": (_int, [_vp, _vp, _vp, _sz, _vp, _vp]),
    "dbhip_ujoin_workspace_bytes": (_sz, [_sz]),
    "dbhip_ujoin_build_u32": (_int, [_vp, _vp, _sz, _vp, _sz, _vp]),
    "dbhip_ujoin_probe_u32": (_int, [_vp, _vp, _sz, _vp, _sz, _vp, _vp, _vp, _vp]),
    "dbhip_bitmask_table_workspace_bytes": (_sz, [_sz]),
    "dbhip_bitmask_table_reset": (_int, [_vp, _sz, _sz, _vp]),
    "dbhip_bitmask_table_insert_u32": (_int, [_vp, _vp, _sz, _vp, _sz, _sz, _int, _u32, _int, _vp]),
    "dbhip_bitmask_table_lookup_u32": (_int, [_vp, _sz, _vp, _sz, _int, _u32, _vp, _vp, _vp]),
    "dbhip_pjoin_partition_workspace_bytes": (_sz, [_sz, _u32]),
    "dbhip_pjoin_partition_u32": (_int, [_vp, _sz, _u64, _u32, _vp, _vp, _vp, _vp, _sz, _vp]),
    "dbhip_gather_u32": (_int, [_vp, _vp, _sz, _vp, _vp]),
    "dbhip_check_pjoin_route_u32": (_int, [_vp, _sz, _u32, _u32, _vp, _vp]),
    "dbhip_reduce_sum_i32": (_int, [_vp, _sz, _vp, _vp]),
    "dbhip_nested_join_u32": (_int, [_vp, _vp, _vp, _vp, _sz, _sz, _vp, _vp, _vp, _vp]),
    "dbhip_exclusive_scan_u32_workspace_bytes": (_sz, [_sz]),
    "dbhip_exclusive_scan_u32": (_int, [_vp, _sz, _u32, _vp, _vp, _sz, _vp]),
    "dbhip_check_fingerprint_workspace_bytes": (_sz, [_sz]),
    "dbhip_check_fingerprint_lt_i32": (_int, [_vp, _sz, _i32, _vp, _vp, _sz, _vp]),
    "dbhip_check_sorted_u32": (_int, [_vp, _sz, _int, _vp, _vp]),
    "dbhip_check_weighted_sum_u32": (_int, [_vp, _vp, _sz, _vp, _vp]),
    "dbhip_check_permutation_workspace_bytes": (_sz, [_sz]),
    "dbhip_check_permutation_u32": (_int, [_vp, _sz, _vp, _vp, _sz, _vp]),
    "dbhip_check_join_u32": (_int, [_vp, _sz, _vp, _sz, _vp, _vp, _vp, _vp, _u64, _u32, _u32, _vp, _vp]),
    "dbhip_check_ujoin_u32": (_int, [_vp, _vp, _sz, _vp, _vp, _sz, _vp, _vp, _vp, _vp, _vp]),
    "dbhip_check_gen_uniform_u32": (_int, [_vp, _vp, _sz, _u64, _u64, _u32, _u32, _vp, _vp]),
}

_lib = None


class DbhipError(RuntimeError):
    pass


def lib_path() -> Path:
    return _LIB_PATH


def lib() -> C.CDLL:
    """Load libdbhip.so (once).  Raises if it has not been built: the product has no CPU fallback."""
    global _lib
    if _lib is None:
        if not _LIB_PATH.exists():
            raise DbhipError(
                f"{_LIB_PATH} is missing: run `python -m dwarf_bench_amd.build` (hipcc, gfx950). "
                "dwarf_bench_amd has no CPU fallback.")
        try:
            import torch  # noqa: F401  — makes torch's libamdhip64 (same SONAME) the one runtime in the process
        except Exception:  # pragma: no cover - torch is plumbing, the C++ host layer runs without it
            pass
        handle = C.CDLL(str(_LIB_PATH), mode=C.RTLD_GLOBAL)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the .so does not export the symbol
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


_ERR = {-1: "DBHIP_EINVAL", -2: "DBHIP_EWORKSPACE", -3: "DBHIP_ENODEVICE"}


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise DbhipError(f"{what} failed: {_ERR.get(rc, 'hipError_t ' + str(rc))}")
