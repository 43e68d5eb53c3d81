"""Test-only helpers for the partitioned join: a numpy/oracle backend for CPU rehearsals of the exchange, the
hash the device partitioner uses restated in numpy, and the global checker.  Never imported by the product."""
import numpy as np
import torch

from oracle import pyoracle as po


def fmix32(k: np.ndarray) -> np.ndarray:
    h = k.astype(np.uint64)
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x85ebca6b)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(13)
    h = (h * np.uint64(0xc2b2ae35)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(16)
    return h


def dest_of(keys: np.ndarray, parts: int) -> np.ndarray:
    """(fmix32(key * 0x9E3779B1 + 0x7F4A7C15) * parts) >> 32, as jl_rank_of in csrc/join_lds.hip: a hash independent
    of the one the local join partitions by"""
    pre = (keys.astype(np.uint64) * np.uint64(0x9E3779B1) + np.uint64(0x7F4A7C15)) & np.uint64(0xFFFFFFFF)
    return ((fmix32(pre) * np.uint64(parts)) >> np.uint64(32)).astype(np.int64)


class OracleBackend:
    """CPU stand-in for HipBackend: same contracts, numpy + the oracle's OmniSci restatement."""

    def partition(self, keys, first_row_id, parts):
        k = keys.numpy().view(np.uint32)
        d = dest_of(k, parts)
        order = np.argsort(d, kind="stable")
        counts = np.bincount(d, minlength=parts).astype(np.int64)
        rid = (np.arange(k.size, dtype=np.int64) + first_row_id).astype(np.uint32)
        return (torch.from_numpy(k[order].view(np.int32).copy()), torch.from_numpy(rid[order].view(np.int32).copy()),
                torch.from_numpy(counts))

    def build(self, build_keys, build_row_ids, n_probe):
        return (build_keys, build_row_ids)

    def probe(self, plan, probe_keys, probe_row_ids=None):
        build_keys, build_row_ids = plan
        b = build_keys.numpy().view(np.uint32)
        p = probe_keys.numpy().view(np.uint32)
        pos, cnt, ids = po.join_omnisci(b, p)
        if build_row_ids is not None:
            ids = build_row_ids.numpy().view(np.uint32)[ids.astype(np.int64)]
        t = lambda a: torch.from_numpy(a.astype(np.uint32).view(np.int32).copy())
        if probe_row_ids is None:
            probe_row_ids = t(np.arange(p.size, dtype=np.uint32))
        return probe_row_ids, t(pos), t(cnt), t(ids)

    def local_join(self, build_keys, probe_keys, build_row_ids=None, probe_row_ids=None):
        return self.probe(self.build(build_keys, build_row_ids, probe_keys.numel()), probe_keys, probe_row_ids)

    def column_sum(self, col):
        return int(col.numpy().view(np.uint32).astype(np.uint64).sum()) & 0xFFFFFFFF


class LossyBackend(OracleBackend):
    """reports a wrong sum for one received column on rank 1: the conservation check must notice"""

    def __init__(self, rank):
        self.rank, self.calls = rank, 0

    def column_sum(self, col):
        self.calls += 1
        s = super().column_sum(col)
        return (s + 1) & 0xFFFFFFFF if (self.rank == 1 and self.calls == 6) else s


def check_global(results, build_all: np.ndarray, probe_all: np.ndarray):
    """results: list over ranks of (probe_row_ids, pos, cnt, build_row_ids) numpy uint32 arrays.
    Canonical form: per global probe row the count, and the sorted list of global build row ids."""
    bc, off, bids = po.join_bruteforce(build_all, probe_all)
    seen = np.zeros(probe_all.size, dtype=bool)
    for rid, pos, cnt, ids in results:
        assert not seen[rid].any(), "a probe row was delivered to two ranks"
        seen[rid] = True
        assert np.array_equal(cnt.astype(np.uint64), bc[rid])
        for i in range(rid.size):
            got = np.sort(ids[pos[i]: pos[i] + cnt[i]]).astype(np.uint64)
            assert np.array_equal(got, bids[int(off[rid[i]]): int(off[rid[i] + 1])]), (int(rid[i]),)
    assert seen.all(), "some probe rows were lost in the exchange"
