"""Tensor-level front door to the gfx950 dwarf kernels.

PyTorch is plumbing here (device memory, the current HIP stream, torch.distributed); every op is one or
more calls through the C ABI of include/dbhip.h into libdbhip.so.  Nothing in this module computes on
the host or falls back to torch ops: without the built library the first call raises.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _capi

DEV_OK, DEV_SPIN_TIMEOUT, DEV_KEY_RANGE, DEV_TABLE_FULL, DEV_RANK_ORDER = 0, 1, 2, 4, 8


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _need(t: torch.Tensor, dtype: torch.dtype, name: str) -> None:
    if not t.is_cuda or t.dtype != dtype or not t.is_contiguous():
        raise ValueError(f"{name}: expected a contiguous {dtype} tensor on the GPU, got {t.dtype} on {t.device}")


def _need16(t: torch.Tensor, name: str) -> None:
    """radix sort and group-by read their columns with 16-byte loads (include/dbhip.h): a slice like t[1:] is refused"""
    if t.numel() and t.data_ptr() % 16:
        raise ValueError(f"{name}: the column must start on a 16-byte boundary (got offset {t.data_ptr() % 16}); "
                         "copy the slice with .clone() first")


def _ws(nbytes: int, device) -> torch.Tensor:
    # torch's caching allocator hands out >= 512-byte aligned blocks; the C ABI asks for 256.  Zeroed: an entry
    # point that returns early (n == 0) must not leave an uninitialised status word behind.
    return torch.zeros(max(int(nbytes), 256), dtype=torch.uint8, device=device)


def workspace_status(ws: torch.Tensor) -> int:
    """Device-side status word of a workspace (synchronises the current stream)."""
    st = C.c_uint32(0xFFFFFFFF)
    _capi.check(_capi.lib().dbhip_workspace_status(ws.data_ptr(), C.byref(st), _stream()), "workspace_status")
    return st.value


def _check_status(ws: torch.Tensor, what: str) -> None:
    st = workspace_status(ws)
    if st != DEV_OK:
        raise _capi.DbhipError(f"{what}: device status {st:#x}")


def device_info(device: int = 0):
    name = C.create_string_buffer(64)
    cus, wave = C.c_int(0), C.c_int(0)
    _capi.check(_capi.lib().dbhip_device_info(device, name, 64, C.byref(cus), C.byref(wave)), "device_info")
    return name.value.decode(), cus.value, wave.value


# ---------------------------------------------------------------------------------------------
# deterministic synthetic columns
# ---------------------------------------------------------------------------------------------
def gen_uniform_u32(n: int, seed: int, lo: int, hi: int, first_index: int = 0, device="cuda",
                    dtype: torch.dtype = torch.int32) -> torch.Tensor:
    """lo + mix64(seed, first_index + i) % (hi - lo + 1); dtype int32 reinterprets the uint32 bits."""
    out = torch.empty(n, dtype=dtype, device=device)
    _capi.check(_capi.lib().dbhip_gen_uniform_u32(out.data_ptr(), n, seed, first_index, lo, hi, _stream()),
                "gen_uniform_u32")
    return out


def gen_unique_sorted_u32(n: int, seed: int, first_index: int = 0, device="cuda") -> torch.Tensor:
    out = torch.empty(n, dtype=torch.int32, device=device)
    _capi.check(_capi.lib().dbhip_gen_unique_sorted_u32(out.data_ptr(), n, seed, first_index, _stream()),
                "gen_unique_sorted_u32")
    return out


# ---------------------------------------------------------------------------------------------
# dwarf 1: scan / compaction
# ---------------------------------------------------------------------------------------------
class CopyIfLt:
    """Reusable plan for out = [x in src : x < filter]: owns workspace + output buffers for a size."""

    def __init__(self, n: int, device="cuda"):
        self.n = n
        lib = _capi.lib()
        self.ws_bytes = lib.dbhip_copy_if_lt_i32_workspace_bytes(n)
        self.ws = _ws(self.ws_bytes, device)
        self.out = torch.empty(max(n, 1), dtype=torch.int32, device=device)
        self.out_size = torch.zeros(1, dtype=torch.int64, device=device)
        self._seen: dict[int, float] = {}
        self._last_filter = None

    # selectivity from which the single-launch dense variant is the faster one.  2^28 rows (tools/ab.py scan), two runs:
    # s = 2.5 % two-launch 220 / dense 240 us, 5 % 242 / 240 and 218 / 242, 7.5 % 255 / 241, 10 % 265 / 242 and 241 / 246,
    # 25 % 328 / 258 and 312 / 260 — the dense variant is the same from run to run, the two-launch one is not (its staging
    # traffic), and they cross somewhere between 5 % and 10 %
    DENSE_ABOVE = 0.075

    def launch(self, src: torch.Tensor, filter_value: int, dense: bool | None = None) -> None:
        """Asynchronous on the current stream; nothing is read back.  dense: True / False pick the variant
        (dbhip_copy_if_lt_dense_i32 / dbhip_copy_if_lt_i32); None = by the selectivity the last result() saw for this
        filter value (the first call of a plan takes the two-launch path)."""
        _need(src, torch.int32, "src")
        if src.numel() != self.n:
            raise ValueError("size mismatch")
        if dense is None:
            dense = self._seen.get(filter_value, 0.0) > self.DENSE_ABOVE
        fn = _capi.lib().dbhip_copy_if_lt_dense_i32 if dense else _capi.lib().dbhip_copy_if_lt_i32
        self._last_filter = filter_value
        _capi.check(fn(src.data_ptr(), self.n, filter_value, self.out.data_ptr(), self.out_size.data_ptr(),
                       self.ws.data_ptr(), self.ws_bytes, _stream()), "copy_if_lt_i32")

    def result(self) -> torch.Tensor:
        _check_status(self.ws, "copy_if_lt_i32")
        m = int(self.out_size.item())
        if self._last_filter is not None and self.n:
            self._seen[self._last_filter] = m / self.n
        return self.out[:m]


def copy_if_lt(src: torch.Tensor, filter_value: int, dense: bool = False) -> torch.Tensor:
    plan = CopyIfLt(src.numel(), src.device)
    plan.launch(src, filter_value, dense=dense)
    return plan.result()


# ---------------------------------------------------------------------------------------------
# dwarf 2: radix sort
# ---------------------------------------------------------------------------------------------
class RadixSort:
    def __init__(self, n: int, radix_bits: int = 8, device="cuda"):
        self.n, self.bits = n, radix_bits
        self.ws_bytes = _capi.lib().dbhip_radix_sort_workspace_bytes(n, radix_bits)
        self.ws = _ws(self.ws_bytes, device)
        self.tmp = torch.empty(max(n, 1), dtype=torch.int32, device=device)

    def launch(self, keys: torch.Tensor, signed: bool = False) -> None:
        """Sorts `keys` (int32 storage) in place; signed=False orders the bits as uint32."""
        _need(keys, torch.int32, "keys")
        _need16(keys, "keys")
        if keys.numel() != self.n:
            raise ValueError("size mismatch")
        fn = _capi.lib().dbhip_radix_sort_i32 if signed else _capi.lib().dbhip_radix_sort_u32
        _capi.check(fn(keys.data_ptr(), self.tmp.data_ptr(), self.n, self.bits, self.ws.data_ptr(), self.ws_bytes,
                       _stream()), "radix_sort")


def radix_sort_prepare() -> int:
    """optional calibration (synchronises the current stream once): device-side self-test of the LDS-atomic ranking;
    returns the rank mode the sorts of this device use from now on (1 = LDS atomics, 0 = ballots)"""
    rc = _capi.lib().dbhip_radix_sort_prepare(_stream())
    if rc < 0 or rc > 1:
        _capi.check(rc if rc else -1, "radix_sort_prepare")
    return rc


def radix_sort_(keys: torch.Tensor, signed: bool = False, radix_bits: int = 8) -> torch.Tensor:
    plan = RadixSort(keys.numel(), radix_bits, keys.device)
    plan.launch(keys, signed)
    _check_status(plan.ws, "radix_sort")
    return keys


# ---------------------------------------------------------------------------------------------
# dwarf 3: group-by SUM
# ---------------------------------------------------------------------------------------------
class GroupBySum:
    def __init__(self, n: int, groups: int, device="cuda"):
        self.n, self.groups = n, groups
        self.ws_bytes = _capi.lib().dbhip_groupby_sum_u32_workspace_bytes(n, groups)
        self.ws = _ws(self.ws_bytes, device)
        self.out = torch.empty(max(groups, 1), dtype=torch.int32, device=device)

    def launch(self, keys: torch.Tensor, vals: torch.Tensor) -> None:
        _need(keys, torch.int32, "keys")
        _need(vals, torch.int32, "vals")
        _need16(keys, "keys")
        _need16(vals, "vals")
        if keys.numel() != self.n or vals.numel() != self.n:
            raise ValueError("size mismatch")
        _capi.check(_capi.lib().dbhip_groupby_sum_u32(keys.data_ptr(), vals.data_ptr(), self.n, self.groups,
                                                      self.out.data_ptr(), self.ws.data_ptr(), self.ws_bytes,
                                                      _stream()), "groupby_sum_u32")

    def partial(self, keys: torch.Tensor, vals: torch.Tensor, max_private_tables: int = 0) -> None:
        """phase 1 of groupby/groupby_local.cpp:52-83: privatised partial sums; 0 = let the library choose"""
        _need(keys, torch.int32, "keys")
        _need(vals, torch.int32, "vals")
        _need16(keys, "keys")
        _need16(vals, "vals")
        if keys.numel() != self.n or vals.numel() != self.n:
            raise ValueError("size mismatch")
        _capi.check(_capi.lib().dbhip_groupby_partial_u32(keys.data_ptr(), vals.data_ptr(), self.n, self.groups,
                                                          max_private_tables, self.ws.data_ptr(), self.ws_bytes,
                                                          _stream()), "groupby_partial_u32")

    def merge(self, max_private_tables: int = 0) -> None:
        """phase 2 (groupby_local.cpp:85-112): fold the private tables into the result"""
        _capi.check(_capi.lib().dbhip_groupby_merge_u32(self.groups, max_private_tables, self.out.data_ptr(),
                                                        self.ws.data_ptr(), _stream()), "groupby_merge_u32")

    def result(self) -> torch.Tensor:
        _check_status(self.ws, "groupby_sum_u32")
        return self.out[: self.groups]


def groupby_sum(keys: torch.Tensor, vals: torch.Tensor, groups: int) -> torch.Tensor:
    plan = GroupBySum(keys.numel(), groups, keys.device)
    plan.launch(keys, vals)
    return plan.result()


# ---------------------------------------------------------------------------------------------
# dwarf 4a: one-to-many hash join (JoinOmnisci semantics)
# ---------------------------------------------------------------------------------------------
class HashJoin:
    def __init__(self, n_build: int, n_probe: int, device="cuda"):
        self.nb, self.np = n_build, n_probe
        self.ws_bytes = _capi.lib().dbhip_join_workspace_bytes(n_build)
        self.ws = _ws(self.ws_bytes, device)
        self.ids = torch.empty(max(n_build, 1), dtype=torch.int32, device=device)
        self.pos = torch.empty(max(n_probe, 1), dtype=torch.int32, device=device)
        self.cnt = torch.empty(max(n_probe, 1), dtype=torch.int32, device=device)

    def build(self, build_keys: torch.Tensor, row_ids: torch.Tensor | None = None) -> None:
        """row_ids given: the id buffer receives those values (e.g. global row ids) instead of 0..n-1"""
        _need(build_keys, torch.int32, "build_keys")
        if row_ids is None:
            _capi.check(_capi.lib().dbhip_join_build_u32(build_keys.data_ptr(), self.nb, self.ids.data_ptr(),
                                                         self.ws.data_ptr(), self.ws_bytes, _stream()), "join_build_u32")
        else:
            _need(row_ids, torch.int32, "row_ids")
            _capi.check(_capi.lib().dbhip_join_build_pairs_u32(build_keys.data_ptr(), row_ids.data_ptr(), self.nb,
                                                               self.ids.data_ptr(), self.ws.data_ptr(), self.ws_bytes,
                                                               _stream()), "join_build_pairs_u32")

    def probe(self, probe_keys: torch.Tensor) -> None:
        _need(probe_keys, torch.int32, "probe_keys")
        _capi.check(_capi.lib().dbhip_join_probe_u32(probe_keys.data_ptr(), self.np, self.ws.data_ptr(),
                                                     self.nb, self.pos.data_ptr(), self.cnt.data_ptr(), _stream()),
                    "join_probe_u32")

    def result(self):
        _check_status(self.ws, "join")
        return self.pos[: self.np], self.cnt[: self.np], self.ids[: self.nb]


class RadixJoin:
    """dwarf 4a without the probe-row-order guarantee: both sides partitioned alike, one fused LDS build + probe launch.
    Results: probe_row_ids / pos / cnt in the probe side's partition order + the id buffer (dbhip_join_radix_*)."""

    def __init__(self, n_build: int, n_probe: int, device="cuda"):
        self.nb, self.np = n_build, n_probe
        self.ws_bytes = _capi.lib().dbhip_join_radix_workspace_bytes(n_build, n_probe)
        self.ws = _ws(self.ws_bytes, device)
        self.ids = torch.empty(max(n_build, 1), dtype=torch.int32, device=device)
        self.rid = torch.empty(max(n_probe, 1), dtype=torch.int32, device=device)
        self.pos = torch.empty(max(n_probe, 1), dtype=torch.int32, device=device)
        self.cnt = torch.empty(max(n_probe, 1), dtype=torch.int32, device=device)

    def _partition(self, side: int, keys: torch.Tensor, row_ids: torch.Tensor | None) -> None:
        _need(keys, torch.int32, "keys")
        if row_ids is not None:
            _need(row_ids, torch.int32, "row_ids")
        _capi.check(_capi.lib().dbhip_join_radix_partition_u32(side, keys.data_ptr(),
                                                               row_ids.data_ptr() if row_ids is not None else None,
                                                               keys.numel(), self.nb, self.np, self.ws.data_ptr(),
                                                               self.ws_bytes, _stream()), "join_radix_partition_u32")

    def partition_build(self, keys: torch.Tensor, row_ids: torch.Tensor | None = None) -> None:
        self._partition(0, keys, row_ids)

    def partition_probe(self, keys: torch.Tensor, row_ids: torch.Tensor | None = None) -> None:
        self._partition(1, keys, row_ids)

    def match(self) -> None:
        _capi.check(_capi.lib().dbhip_join_radix_match_u32(self.nb, self.np, self.ids.data_ptr(), self.rid.data_ptr(),
                                                           self.pos.data_ptr(), self.cnt.data_ptr(), self.ws.data_ptr(),
                                                           self.ws_bytes, _stream()), "join_radix_match_u32")

    def result(self):
        """-> probe_row_ids, pos, cnt (probe partition order), ids"""
        _check_status(self.ws, "join_radix")
        return self.rid[: self.np], self.pos[: self.np], self.cnt[: self.np], self.ids[: self.nb]


def radix_join(build_keys: torch.Tensor, probe_keys: torch.Tensor, build_row_ids: torch.Tensor | None = None,
               probe_row_ids: torch.Tensor | None = None):
    plan = RadixJoin(build_keys.numel(), probe_keys.numel(), build_keys.device)
    plan.partition_build(build_keys, build_row_ids)
    plan.partition_probe(probe_keys, probe_row_ids)
    plan.match()
    return plan.result()


def join_answers(ids: torch.Tensor, pos: torch.Tensor, cnt: torch.Tensor) -> torch.Tensor:
    """(n_probe, 2) int64: the reference's JoinOneToMany records {device pointer into ids, size}
    (common/dpcpp/omnisci_hashtable.hpp:12-17)"""
    _need(ids, torch.int32, "ids")
    _need(pos, torch.int32, "pos")
    _need(cnt, torch.int32, "cnt")
    out = torch.empty((max(pos.numel(), 1), 2), dtype=torch.int64, device=pos.device)
    _capi.check(_capi.lib().dbhip_join_answers_u32(ids.data_ptr(), pos.data_ptr(), cnt.data_ptr(), pos.numel(),
                                                   out.data_ptr(), _stream()), "join_answers_u32")
    return out[: pos.numel()]


def hash_join(build_keys: torch.Tensor, probe_keys: torch.Tensor):
    plan = HashJoin(build_keys.numel(), probe_keys.numel(), build_keys.device)
    plan.build(build_keys)
    plan.probe(probe_keys)
    return plan.result()


# ---------------------------------------------------------------------------------------------
# dwarf 4b: unique-key payload join (Join semantics)
# ---------------------------------------------------------------------------------------------
class UniqueJoin:
    def __init__(self, n_build: int, n_probe: int, device="cuda"):
        self.nb, self.np = n_build, n_probe
        self.ws_bytes = _capi.lib().dbhip_ujoin_workspace_bytes(n_build)
        self.ws = _ws(self.ws_bytes, device)
        self.out_key = torch.empty(max(n_probe, 1), dtype=torch.int32, device=device)
        self.out_bval = torch.empty_like(self.out_key)
        self.out_pval = torch.empty_like(self.out_key)

    def build(self, keys: torch.Tensor, vals: torch.Tensor) -> None:
        _need(keys, torch.int32, "build_keys")
        _need(vals, torch.int32, "build_vals")
        _capi.check(_capi.lib().dbhip_ujoin_build_u32(keys.data_ptr(), vals.data_ptr(), self.nb, self.ws.data_ptr(),
                                                      self.ws_bytes, _stream()), "ujoin_build_u32")

    def probe(self, keys: torch.Tensor, vals: torch.Tensor) -> None:
        _need(keys, torch.int32, "probe_keys")
        _need(vals, torch.int32, "probe_vals")
        _capi.check(_capi.lib().dbhip_ujoin_probe_u32(keys.data_ptr(), vals.data_ptr(), self.np, self.ws.data_ptr(),
                                                      self.nb, self.out_key.data_ptr(), self.out_bval.data_ptr(),
                                                      self.out_pval.data_ptr(), _stream()), "ujoin_probe_u32")

    def result(self):
        _check_status(self.ws, "ujoin")
        return self.out_key[: self.np], self.out_bval[: self.np], self.out_pval[: self.np]


# ---------------------------------------------------------------------------------------------
# multi-GPU partitioned join: device pieces (the exchange lives in pjoin.py)
# ---------------------------------------------------------------------------------------------
def partition_by_hash(keys: torch.Tensor, first_row_id: int, parts: int):
    """-> (keys bucket-major, global row ids bucket-major, counts[parts] int64 on the device)"""
    _need(keys, torch.int32, "keys")
    n = keys.numel()
    lib = _capi.lib()
    ws_bytes = lib.dbhip_pjoin_partition_workspace_bytes(n, parts)
    ws = _ws(ws_bytes, keys.device)
    out_keys = torch.empty(max(n, 1), dtype=torch.int32, device=keys.device)
    out_rids = torch.empty(max(n, 1), dtype=torch.int32, device=keys.device)
    counts = torch.zeros(parts, dtype=torch.int64, device=keys.device)
    _capi.check(lib.dbhip_pjoin_partition_u32(keys.data_ptr(), n, first_row_id, parts, out_keys.data_ptr(),
                                              out_rids.data_ptr(), counts.data_ptr(), ws.data_ptr(), ws_bytes,
                                              _stream()), "pjoin_partition_u32")
    return out_keys[:n], out_rids[:n], counts


def gather_u32(table: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    _need(table, torch.int32, "table")
    _need(idx, torch.int32, "idx")
    out = torch.empty(max(idx.numel(), 1), dtype=torch.int32, device=idx.device)
    _capi.check(_capi.lib().dbhip_gather_u32(table.data_ptr(), idx.data_ptr(), idx.numel(), out.data_ptr(), _stream()),
                "gather_u32")
    return out[: idx.numel()]


# ---------------------------------------------------------------------------------------------
# reduce and nested-loop join (reduce/reduce.cpp, join/nested_join.cpp)
# ---------------------------------------------------------------------------------------------
def reduce_sum(src: torch.Tensor) -> torch.Tensor:
    """one-element int32 tensor: the wrap-around sum of src"""
    _need(src, torch.int32, "src")
    out = torch.empty(1, dtype=torch.int32, device=src.device)
    _capi.check(_capi.lib().dbhip_reduce_sum_i32(src.data_ptr(), src.numel(), out.data_ptr(), _stream()), "reduce_sum_i32")
    return out


def nested_join(a_keys: torch.Tensor, a_vals: torch.Tensor, b_keys: torch.Tensor, b_vals: torch.Tensor):
    """dense (n_a x n_b) cell matrices (key, a_val, b_val); empty cells are (0, -1, -1)"""
    for t, name in ((a_keys, "a_keys"), (a_vals, "a_vals"), (b_keys, "b_keys"), (b_vals, "b_vals")):
        _need(t, torch.int32, name)
    na, nb = a_keys.numel(), b_keys.numel()
    if a_vals.numel() != na or b_vals.numel() != nb:
        raise ValueError("size mismatch")
    outs = [torch.empty(max(na * nb, 1), dtype=torch.int32, device=a_keys.device) for _ in range(3)]
    _capi.check(_capi.lib().dbhip_nested_join_u32(a_keys.data_ptr(), a_vals.data_ptr(), b_keys.data_ptr(),
                                                  b_vals.data_ptr(), na, nb, outs[0].data_ptr(), outs[1].data_ptr(),
                                                  outs[2].data_ptr(), _stream()), "nested_join_u32")
    return tuple(o[: na * nb].view(na, nb) if na and nb else o[:0] for o in outs)


# ---------------------------------------------------------------------------------------------
# bitmask-claimed table (SimpleNonOwningHashTable counterpart)
# ---------------------------------------------------------------------------------------------
class BitmaskTable:
    def __init__(self, table_size: int, hash_kind: int = 1, seed: int = 0, device="cuda"):
        self.size, self.kind, self.seed = table_size, hash_kind, seed
        self.ws_bytes = _capi.lib().dbhip_bitmask_table_workspace_bytes(table_size)
        self.ws = _ws(self.ws_bytes, device)
        self.reset()

    def reset(self) -> None:
        _capi.check(_capi.lib().dbhip_bitmask_table_reset(self.ws.data_ptr(), self.ws_bytes, self.size, _stream()),
                    "bitmask_table_reset")

    def insert(self, keys: torch.Tensor, vals: torch.Tensor, serial: bool = False) -> None:
        _need(keys, torch.int32, "keys")
        _need(vals, torch.int32, "vals")
        _capi.check(_capi.lib().dbhip_bitmask_table_insert_u32(keys.data_ptr(), vals.data_ptr(), keys.numel(),
                                                               self.ws.data_ptr(), self.ws_bytes, self.size, self.kind,
                                                               self.seed, int(serial), _stream()), "bitmask_table_insert")

    def lookup(self, keys: torch.Tensor):
        _need(keys, torch.int32, "keys")
        n = keys.numel()
        vals = torch.empty(max(n, 1), dtype=torch.int32, device=keys.device)
        found = torch.empty(max(n, 1), dtype=torch.int32, device=keys.device)
        _capi.check(_capi.lib().dbhip_bitmask_table_lookup_u32(keys.data_ptr(), n, self.ws.data_ptr(), self.size,
                                                               self.kind, self.seed, vals.data_ptr(), found.data_ptr(),
                                                               _stream()), "bitmask_table_lookup")
        return vals[:n], found[:n]

    def check(self) -> None:
        """raise if an insert found the table full (DEV_TABLE_FULL)"""
        _check_status(self.ws, "bitmask_table")

    def slot_values(self) -> torch.Tensor:
        """the payload array of the table (for slot-level known-answer tests)"""
        off = 256 + 4 * self.size
        return self.ws[off: off + 4 * self.size].view(torch.int32)


# ---------------------------------------------------------------------------------------------
# exclusive prefix sum (scan/scan.cl:44-66, tests/scan_tests.cpp:14-21, dpl_wrapper.hpp:18-25)
# ---------------------------------------------------------------------------------------------
class ExclusiveScan:
    """launch() is asynchronous; result() synchronises and raises on a device-side status (the single-launch path waits
    on other workgroups with a bounded spin: DBHIP_DEV_SPIN_TIMEOUT means the prefix that was written is wrong)"""

    def __init__(self, n: int, device="cuda"):
        self.n = n
        self.ws_bytes = _capi.lib().dbhip_exclusive_scan_u32_workspace_bytes(n)
        self.ws = _ws(self.ws_bytes, device)
        self.out = None

    def launch(self, src: torch.Tensor, init: int = 0, out: torch.Tensor | None = None) -> None:
        _need(src, torch.int32, "src")
        assert src.numel() == self.n
        if out is None:
            out = torch.empty(max(self.n, 1), dtype=torch.int32, device=src.device)[: self.n]
        _need(out, torch.int32, "out")
        self.out = out
        _capi.check(_capi.lib().dbhip_exclusive_scan_u32(src.data_ptr(), self.n, init & 0xFFFFFFFF, out.data_ptr(),
                                                         self.ws.data_ptr(), self.ws_bytes, _stream()), "exclusive_scan_u32")

    def result(self) -> torch.Tensor:
        _check_status(self.ws, "exclusive_scan_u32")
        return self.out


def exclusive_scan(src: torch.Tensor, init: int = 0, out: torch.Tensor | None = None) -> torch.Tensor:
    """out[0] = init, out[i] = init + src[0] + ... + src[i-1] (uint32 wrap-around); out may be src.  Checked: reads the
    workspace's status word back (one synchronisation); use ExclusiveScan for asynchronous launches."""
    plan = ExclusiveScan(src.numel(), src.device)
    plan.launch(src, init, out)
    return plan.result()


# ---------------------------------------------------------------------------------------------
# device-side validators (what the ...Hip dwarfs use for Result::valid at large sizes)
# ---------------------------------------------------------------------------------------------
def _result(words: int, device) -> torch.Tensor:
    return torch.empty(words, dtype=torch.int64, device=device)


def _u64(t: torch.Tensor):
    return [int(x) & 0xFFFFFFFFFFFFFFFF for x in t.cpu().tolist()]


def check_fingerprint_lt(src: torch.Tensor, filter_value: int):
    """-> (fingerprint, length) of the subsequence of src with x < filter_value, order-sensitive"""
    _need(src, torch.int32, "src")
    lib = _capi.lib()
    ws_bytes = lib.dbhip_check_fingerprint_workspace_bytes(src.numel())
    ws = _ws(ws_bytes, src.device)
    res = _result(2, src.device)
    _capi.check(lib.dbhip_check_fingerprint_lt_i32(src.data_ptr(), src.numel(), filter_value, res.data_ptr(), ws.data_ptr(),
                                                   ws_bytes, _stream()), "check_fingerprint_lt_i32")
    return tuple(_u64(res))


def check_sorted(keys: torch.Tensor, signed: bool = False):
    """-> (descents, multiset fingerprint, key sum)"""
    _need(keys, torch.int32, "keys")
    res = _result(3, keys.device)
    _capi.check(_capi.lib().dbhip_check_sorted_u32(keys.data_ptr(), keys.numel(), int(signed), res.data_ptr(), _stream()),
                "check_sorted_u32")
    return tuple(_u64(res))


def check_weighted_sum(keys: torch.Tensor | None, vals: torch.Tensor):
    """-> two 32-bit weighted sums of vals, weights drawn from keys (None: the index)"""
    _need(vals, torch.int32, "vals")
    if keys is not None:
        _need(keys, torch.int32, "keys")
    res = _result(2, vals.device)
    _capi.check(_capi.lib().dbhip_check_weighted_sum_u32(keys.data_ptr() if keys is not None else None, vals.data_ptr(),
                                                         vals.numel(), res.data_ptr(), _stream()), "check_weighted_sum_u32")
    return tuple(_u64(res))


def check_permutation(ids: torch.Tensor) -> int:
    """-> number of entries that are out of range or repeated (0 iff ids is a permutation of 0..n-1)"""
    _need(ids, torch.int32, "ids")
    lib = _capi.lib()
    ws_bytes = lib.dbhip_check_permutation_workspace_bytes(ids.numel())
    ws = _ws(ws_bytes, ids.device)
    res = _result(1, ids.device)
    _capi.check(lib.dbhip_check_permutation_u32(ids.data_ptr(), ids.numel(), res.data_ptr(), ws.data_ptr(), ws_bytes,
                                                _stream()), "check_permutation_u32")
    return _u64(res)[0]


def check_join(sorted_build: torch.Tensor, probe: torch.Tensor, pos: torch.Tensor, cnt: torch.Tensor, ids: torch.Tensor,
               build_keys: torch.Tensor | None = None, gen=(0, 0, 0)):
    """-> (bad probe rows, sum of counts); build_keys None: ids are global row ids of gen = (seed, lo, hi)"""
    for t, name in ((sorted_build, "sorted_build"), (probe, "probe"), (pos, "pos"), (cnt, "cnt"), (ids, "ids")):
        _need(t, torch.int32, name)
    res = _result(2, probe.device)
    _capi.check(_capi.lib().dbhip_check_join_u32(sorted_build.data_ptr(), sorted_build.numel(), probe.data_ptr(),
                                                 probe.numel(), pos.data_ptr(), cnt.data_ptr(), ids.data_ptr(),
                                                 build_keys.data_ptr() if build_keys is not None else None,
                                                 gen[0], gen[1], gen[2], res.data_ptr(), _stream()), "check_join_u32")
    return tuple(_u64(res))


def check_ujoin(sorted_build: torch.Tensor, build_vals: torch.Tensor, probe: torch.Tensor, probe_vals: torch.Tensor,
                out_key: torch.Tensor, out_bval: torch.Tensor, out_pval: torch.Tensor):
    """-> (bad probe rows, hits)"""
    res = _result(2, probe.device)
    _capi.check(_capi.lib().dbhip_check_ujoin_u32(sorted_build.data_ptr(), build_vals.data_ptr(), sorted_build.numel(),
                                                  probe.data_ptr(), probe_vals.data_ptr(), probe.numel(),
                                                  out_key.data_ptr(), out_bval.data_ptr(), out_pval.data_ptr(),
                                                  res.data_ptr(), _stream()), "check_ujoin_u32")
    return tuple(_u64(res))


def check_gen_uniform(values: torch.Tensor, seed: int, lo: int, hi: int, first_index: int = 0,
                      indices: torch.Tensor | None = None) -> int:
    """-> number of values that differ from lo + mix64(seed, index) % (hi - lo + 1)"""
    _need(values, torch.int32, "values")
    res = _result(1, values.device)
    _capi.check(_capi.lib().dbhip_check_gen_uniform_u32(values.data_ptr(), indices.data_ptr() if indices is not None else None,
                                                        values.numel(), seed, first_index, lo, hi, res.data_ptr(),
                                                        _stream()), "check_gen_uniform_u32")
    return _u64(res)[0]
