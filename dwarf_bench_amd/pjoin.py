"""Radix-partitioned hash join across the GPUs of one node (one process per GPU, torch.distributed over
RCCL/xGMI).  No reference counterpart — the reference is single-device; SURVEY 8(e) defines the path.

Per rank, with local shards of the build (R) and probe (S) key columns and their global row offsets:
  1. partition  both shards by destination GPU = mixed hash of the key (dbhip_pjoin_partition_u32):
                bucket-major (key, global row id) pairs + per-bucket counts;
  2. counts     one all_to_all of the P-entry count vectors (who sends how much to whom);
  3. exchange   all_to_all of the pairs (RCCL: every GPU sends 1/P of its rows to each peer, one peer per
                xGMI link, all links busy at once) — keys and row ids of one relation travel as one
                int32 [n, 2]... two columns, one collective per column;
  4. local join dwarf 4a on the received pairs (dbhip_join_build_pairs_u32 / dbhip_join_probe_u32): the id
                buffer holds GLOBAL build row ids.
Results stay sharded by key hash: per rank (probe global row id, offset, count) + the id buffer.

The compute steps go through a small backend object so that the orchestration (split sizes, collectives,
bookkeeping) can be exercised on CPU/gloo in tests with a test-only backend; the product backend is
HipBackend below and nothing else is ever chosen implicitly.
"""
from __future__ import annotations

from dataclasses import dataclass

import torch
import torch.distributed as dist


class HipBackend:
    """device steps on the local GPU through the C ABI (dwarf_bench_amd.ops)"""

    def partition(self, keys: torch.Tensor, first_row_id: int, parts: int):
        from . import ops
        return ops.partition_by_hash(keys, first_row_id, parts)

    def local_join(self, build_keys: torch.Tensor, probe_keys: torch.Tensor, build_row_ids: torch.Tensor | None = None):
        """-> pos, cnt, ids; ids hold build_row_ids values when given (global row ids), else local indices"""
        from . import ops
        plan = ops.HashJoin(build_keys.numel(), probe_keys.numel(), build_keys.device)
        plan.build(build_keys, build_row_ids)
        plan.probe(probe_keys)
        return plan.result()


@dataclass
class PartitionedJoinResult:
    probe_row_ids: torch.Tensor  # global probe row id of every probe row this rank received
    pos: torch.Tensor            # offset into build_row_ids
    cnt: torch.Tensor            # number of matching build rows
    build_row_ids: torch.Tensor  # global build row ids grouped by key
    sent_rows: int               # rows this rank shipped to other ranks (both relations)
    recv_build_rows: int
    recv_probe_rows: int


def _a2a(out: torch.Tensor, inp: torch.Tensor, out_splits, in_splits, group) -> None:
    """all_to_all_single; a gloo group (CPU rehearsal of the exchange) only moves host memory, so device
    tensors are staged through the host there.  With the nccl (= RCCL) backend this is one collective on
    device memory over xGMI."""
    if inp.is_cuda and dist.get_backend(group) == "gloo":
        host_out = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(host_out, inp.cpu(), output_split_sizes=out_splits, input_split_sizes=in_splits, group=group)
        out.copy_(host_out)
        return
    dist.all_to_all_single(out, inp, output_split_sizes=out_splits, input_split_sizes=in_splits, group=group)


def _exchange(column: torch.Tensor, send_counts, recv_counts, group) -> torch.Tensor:
    out = torch.empty(int(sum(recv_counts)), dtype=column.dtype, device=column.device)
    _a2a(out, column.contiguous(), [int(x) for x in recv_counts], [int(x) for x in send_counts], group)
    return out


def partitioned_join(build_keys: torch.Tensor, probe_keys: torch.Tensor, build_first_row: int, probe_first_row: int,
                     group=None, backend=None) -> PartitionedJoinResult:
    """Join this rank's shards; collective over `group` (default: WORLD).  World size 1 = plain local join."""
    backend = backend or HipBackend()
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if world == 1:
        pos, cnt, ids = backend.local_join(build_keys, probe_keys)
        dev = probe_keys.device
        rid = torch.arange(probe_first_row, probe_first_row + probe_keys.numel(), dtype=torch.int64, device=dev).to(torch.int32)
        ids_global = ids if build_first_row == 0 else (ids.to(torch.int64) + build_first_row).to(torch.int32)
        return PartitionedJoinResult(rid, pos, cnt, ids_global, 0, build_keys.numel(), probe_keys.numel())

    rk, rr, rc = backend.partition(build_keys, build_first_row, world)
    sk, sr, sc = backend.partition(probe_keys, probe_first_row, world)
    # counts matrix: row = sender.  One small all_to_all tells every rank what it will receive.
    send_counts = torch.stack([rc, sc]).to(torch.int64)           # [2, P]
    recv_flat = torch.empty(2 * world, dtype=torch.int64, device=send_counts.device)
    _a2a(recv_flat, _interleave(send_counts, world), None, None, group)
    recv_counts = _deinterleave(recv_flat, world)
    send_h, recv_h = send_counts.cpu().tolist(), recv_counts.cpu().tolist()  # split sizes must be host ints

    rk_in = _exchange(rk, send_h[0], recv_h[0], group)
    rr_in = _exchange(rr, send_h[0], recv_h[0], group)
    sk_in = _exchange(sk, send_h[1], recv_h[1], group)
    sr_in = _exchange(sr, send_h[1], recv_h[1], group)

    pos, cnt, ids_global = backend.local_join(rk_in, sk_in, rr_in)
    sent = int(sum(send_h[0]) - send_h[0][rank] + sum(send_h[1]) - send_h[1][rank])
    return PartitionedJoinResult(sr_in, pos, cnt, ids_global, sent, rk_in.numel(), sk_in.numel())


def _interleave(counts: torch.Tensor, world: int) -> torch.Tensor:
    """[2, P] -> flat [P*2] with the two relations' counts for peer p adjacent (one all_to_all of 2 ints per peer)"""
    return counts.t().contiguous().view(-1)


def _deinterleave(flat: torch.Tensor, world: int) -> torch.Tensor:
    return flat.view(world, 2).t().contiguous()
