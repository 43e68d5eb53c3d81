"""Radix-partitioned hash join across the GPUs of one node (one process per GPU, torch.distributed over
RCCL/xGMI).  No reference counterpart — the reference is single-device; SURVEY 8(e) defines the path.

Per rank, with local shards of the build (R) and probe (S) key columns and their global row offsets:
  1. partition  both shards by destination GPU = mixed hash of the key (dbhip_pjoin_partition_u32):
                bucket-major (key, global row id) pairs + per-bucket counts;
  2. counts     one all_to_all of the P-entry count vectors (who sends how much to whom);
  3. exchange   all_to_all of the pairs (RCCL: every GPU sends 1/P of its rows to each peer, one peer per
                xGMI link, all links busy at once), one collective per column (keys, row ids);
  4. local join dwarf 4a on the received pairs (dbhip_join_build_pairs_u32 / dbhip_join_probe_u32): the id
                buffer holds GLOBAL build row ids.
Results stay sharded by key hash: per rank (probe global row id, offset, count) + the id buffer.

The steps of the two relations are interleaved so that the xGMI exchange hides behind HBM-bound kernels
(collectives are issued async: they run on RCCL's own stream, `wait()` only makes the compute stream wait):

    compute stream :  partition R |            partition S | build R            | probe S
    RCCL stream    :          counts R | exchange R | counts S | exchange S |

The host blocks twice, on the two tiny count exchanges (split sizes must be host integers).

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

    def build(self, build_keys: torch.Tensor, build_row_ids: torch.Tensor | None, n_probe: int):
        """-> plan; the id buffer holds build_row_ids values when given (global row ids), else local indices"""
        from . import ops
        plan = ops.HashJoin(build_keys.numel(), n_probe, build_keys.device)
        plan.build(build_keys, build_row_ids)
        return plan

    def probe(self, plan, probe_keys: torch.Tensor):
        """-> pos, cnt, ids"""
        plan.probe(probe_keys)
        return plan.result()

    def local_join(self, build_keys: torch.Tensor, probe_keys: torch.Tensor, build_row_ids: torch.Tensor | None = None):
        return self.probe(self.build(build_keys, build_row_ids, probe_keys.numel()), probe_keys)


@dataclass
class PartitionedJoinResult:
    probe_row_ids: torch.Tensor  # global probe row id of every probe row this rank received
    pos: torch.Tensor            # offset into build_row_ids
    cnt: torch.Tensor            # number of matching build rows
    build_row_ids: torch.Tensor  # global build row ids grouped by key
    sent_rows: int               # rows this rank shipped to other ranks (both relations)
    recv_build_rows: int
    recv_probe_rows: int


_MAX_MESSAGE_BYTES = 1 << 31


class _Done:
    """handle of an exchange that has already completed"""

    def __init__(self, out):
        self.out = out

    def wait(self):
        return self.out


class _Pending:
    """handle of a collective in flight on the backend's own stream; wait() orders the current stream after it"""

    def __init__(self, work, out, keep):
        self.work, self.out, self.keep = work, out, keep  # `keep`: the send buffer must outlive the collective

    def wait(self):
        self.work.wait()
        self.keep = None
        return self.out


def _a2a(inp: torch.Tensor, out_splits, in_splits, group):
    """all_to_all_single -> handle.  With the nccl (= RCCL) backend this is one async collective on device
    memory over xGMI.  A gloo group (CPU rehearsal of the exchange) only moves host memory, so device tensors
    are staged through the host there, synchronously."""
    n_out = int(sum(out_splits)) if out_splits is not None else inp.numel()
    inp = inp.contiguous()
    # one send/recv of 2 GiB or more was measured to deliver garbage without an error through RCCL 2.27.7
    # (PartitionedJoinHip, which therefore cuts its messages into 1 GiB pieces); torch's all_to_all_single cannot be
    # cut from here, so refuse instead of corrupting: 2^30 x 2^30 over >= 2 ranks needs 1 GiB per message at most
    biggest = max(list(out_splits or [0]) + list(in_splits or [0]) + [0]) * inp.element_size()
    if biggest >= _MAX_MESSAGE_BYTES:
        raise ValueError(f"a {biggest} byte message to one peer exceeds the {_MAX_MESSAGE_BYTES} byte limit of the exchange")
    if inp.is_cuda and dist.get_backend(group) == "gloo":
        host_out = torch.empty(n_out, dtype=inp.dtype)
        dist.all_to_all_single(host_out, inp.cpu(), output_split_sizes=out_splits, input_split_sizes=in_splits, group=group)
        return _Done(host_out.to(inp.device))
    out = torch.empty(n_out, dtype=inp.dtype, device=inp.device)
    work = dist.all_to_all_single(out, inp, output_split_sizes=out_splits, input_split_sizes=in_splits, group=group,
                                  async_op=True)
    return _Pending(work, out, inp)


def _exchange_counts(counts: torch.Tensor, group):
    """counts: this rank's P send counts (device or host int64/uint64).  One tiny all_to_all tells every rank what
    it will receive.  -> (send, recv) as host int lists — blocks the host until `counts` is computed and exchanged."""
    send = counts.to(torch.int64)
    recv = _a2a(send, None, None, group).wait()
    return [int(x) for x in send.cpu().tolist()], [int(x) for x in recv.cpu().tolist()]


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

    # R: partition, learn the split sizes, start its exchange
    rk, rr, rc = backend.partition(build_keys, build_first_row, world)
    r_send, r_recv = _exchange_counts(rc, group)
    rk_x = _a2a(rk, r_recv, r_send, group)
    rr_x = _a2a(rr, r_recv, r_send, group)
    # S: partition while R is on the links
    sk, sr, sc = backend.partition(probe_keys, probe_first_row, world)
    s_send, s_recv = _exchange_counts(sc, group)
    sk_x = _a2a(sk, s_recv, s_send, group)
    sr_x = _a2a(sr, s_recv, s_send, group)
    # build on the received R pairs while S is on the links, then probe
    rk_in, rr_in = rk_x.wait(), rr_x.wait()
    plan = backend.build(rk_in, rr_in, int(sum(s_recv)))
    sk_in, sr_in = sk_x.wait(), sr_x.wait()
    pos, cnt, ids_global = backend.probe(plan, sk_in)
    sent = int(sum(r_send) - r_send[rank] + sum(s_send) - s_send[rank])
    return PartitionedJoinResult(sr_in, pos, cnt, ids_global, sent, rk_in.numel(), sk_in.numel())
