"""ctypes binding of the C++ partitioned-join engine (dwarf_bench_amd/host/pjoin_engine.{hpp,cpp} in libdbench.so):
the host the `PartitionedJoinHip` dwarf runs on, driven here one process per GPU.  The launcher (bench.py under
torch.distributed.run) only hands the ncclUniqueId round; streams, events, RCCL calls and the pipeline are C++.
No fallback: a missing library raises."""
from __future__ import annotations

import ctypes as C
from pathlib import Path

from . import _capi

_LIB = Path(__file__).resolve().parent / "_lib" / "libdbench.so"
_lib = None

STEP_FIELDS = ("total_us", "partition_us", "exchange_us", "build_us", "probe_us", "until_build_done_us", "exchange_r_us",
               "exchange_s_us")
CHECK_FIELDS = ("bad_pairs", "bad_route", "bad_rows", "matches", "recv_build", "recv_probe", "sent_rows", "conserved")


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not _LIB.exists():
            raise _capi.DbhipError(f"{_LIB} is missing: run `python -m dwarf_bench_amd.build`")
        _capi.lib()  # libdbhip.so first (RTLD_GLOBAL), same HIP runtime as torch
        h = C.CDLL(str(_LIB), mode=C.RTLD_GLOBAL)
        h.dbench_pjoin_unique_id.restype = C.c_int
        h.dbench_pjoin_unique_id.argtypes = [C.c_char_p]
        h.dbench_pjoin_create.restype = C.c_void_p
        h.dbench_pjoin_create.argtypes = [C.c_uint64, C.c_uint, C.c_uint, C.c_int, C.c_char_p, C.c_int]
        h.dbench_pjoin_step_n.restype = C.c_int
        h.dbench_pjoin_step_n.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_uint]
        h.dbench_pjoin_info.restype = C.c_int
        h.dbench_pjoin_info.argtypes = [C.c_void_p, C.POINTER(C.c_uint), C.c_char_p, C.c_ulong]
        h.dbench_pjoin_check.restype = C.c_int
        h.dbench_pjoin_check.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        h.dbench_pjoin_destroy.restype = None
        h.dbench_pjoin_destroy.argtypes = [C.c_void_p]
        _lib = h
    return _lib


def unique_id() -> bytes:
    """rank 0: a fresh ncclUniqueId (128 bytes) for every rank of the job"""
    buf = C.create_string_buffer(128)
    if lib().dbench_pjoin_unique_id(buf) != 0:
        raise _capi.DbhipError("ncclGetUniqueId failed")
    return buf.raw


class NativePartitionedJoin:
    """one rank of the join: n_total x n_total rows over `world` ranks, this process = `rank` on GPU `device`"""

    def __init__(self, n_total: int, rank: int = 0, world: int = 1, device: int = 0, nccl_id: bytes | None = None,
                 direct_single: bool = False):
        if (world > 1 and nccl_id is None) or (nccl_id is not None and len(nccl_id) != 128):
            raise ValueError("a multi-rank join needs the 128-byte id from unique_id() of rank 0")
        # world == 1 with an id: the one-rank rehearsal of the multi-process RCCL path (ncclCommInitRank, self send/recv)
        self._h = lib().dbench_pjoin_create(n_total, rank, world, device, nccl_id, int(direct_single))
        if not self._h:
            raise _capi.DbhipError("dbench_pjoin_create failed (see stderr)")

    def step(self) -> dict:
        t = (C.c_double * len(STEP_FIELDS))()
        got = lib().dbench_pjoin_step_n(self._h, t, len(STEP_FIELDS))
        if got <= 0:
            raise _capi.DbhipError("dbench_pjoin_step_n failed (see stderr)")
        return dict(zip(STEP_FIELDS[:got], (float(x) for x in t[:got])))

    def info(self) -> dict:
        """what RCCL itself says about this rank's communicator (ncclCommCount) and the GPU the rank runs on"""
        w = (C.c_uint * 4)()
        name = C.create_string_buffer(256)
        if lib().dbench_pjoin_info(self._h, w, name, 256) != 0:
            raise _capi.DbhipError("dbench_pjoin_info failed (see stderr)")
        return {"rccl_ranks_seen": int(w[0]), "world": int(w[1]), "device": int(w[2]), "local_ranks": int(w[3]),
                "device_name": name.value.decode(errors="replace")}

    def check(self) -> dict:
        """device-side checks of the last step (collective over all ranks of the join)"""
        w = (C.c_uint64 * 16)()
        if lib().dbench_pjoin_check(self._h, w) != 0:
            raise _capi.DbhipError("dbench_pjoin_check failed (see stderr)")
        out = dict(zip(CHECK_FIELDS, (int(x) for x in w[:8])))
        out["conserved"] = bool(out["conserved"])
        return out

    def close(self) -> None:
        if self._h:
            lib().dbench_pjoin_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()
