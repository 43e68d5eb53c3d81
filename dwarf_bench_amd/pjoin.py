"""Radix-partitioned hash join across the GPUs of one node (one process per GPU, torch.distributed over
RCCL/xGMI).  No reference counterpart — the reference is single-device; SURVEY 8(e) defines the path.

Per rank, with local shards of the build (R) and probe (S) key columns and their global row offsets:
  1. partition  both shards by destination GPU = mixed hash of the key (dbhip_pjoin_partition_u32):
                bucket-major (key, global row id) pairs + per-bucket counts;
  2. counts     one all_gather of the P-entry count vectors: every rank holds the P x P matrix (who sends how much to
                whom), hence its receive sizes and — identically on all ranks — the number of exchange rounds;
  3. exchange   all_to_all of the pairs (RCCL: every GPU sends 1/P of its rows to each peer, one peer per
                xGMI link, all links busy at once), one collective per column (keys, row ids);
  4. local join dwarf 4a on the received pairs as a radix join (dbhip_join_radix_*): the id buffer holds GLOBAL
                build row ids, the probe rows come back with their global row ids.
Results stay sharded by key hash: per rank (probe global row id, offset, count) + the id buffer.

The steps of the two relations are interleaved so that the xGMI exchange hides behind HBM-bound kernels
(collectives are issued async: they run on RCCL's own stream, `wait()` only makes the compute stream wait):

    compute stream :  partition R |            partition S | local partition R' | local partition S' + match
    RCCL stream    :          counts R | exchange R | counts S | exchange S |

The host blocks twice, on the two tiny count gathers (split sizes must be host integers).

The compute steps go through a small backend object so that the orchestration (split sizes, collectives,
bookkeeping) can be exercised on CPU/gloo in tests with a test-only backend; the product backend is
HipBackend below and nothing else is ever chosen implicitly.
"""
from __future__ import annotations

from dataclasses import dataclass

import torch
import torch.distributed as dist


class HipBackend:
    """device steps on the local GPU through the C ABI (dwarf_bench_amd.ops).  The local join is the radix join
    (dbhip_join_radix_*): the received pairs of both sides are partitioned once more with the same geometry and ONE
    launch builds and probes every partition's sub-table in LDS — the results come in the probe side's partition
    order together with their row ids, which is all a sharded result needs."""

    def partition(self, keys: torch.Tensor, first_row_id: int, parts: int):
        from . import ops
        return ops.partition_by_hash(keys, first_row_id, parts)

    def build(self, build_keys: torch.Tensor, build_row_ids: torch.Tensor | None, n_probe: int):
        """-> plan; the id buffer holds build_row_ids values when given (global row ids), else local indices"""
        from . import ops
        plan = ops.RadixJoin(build_keys.numel(), n_probe, build_keys.device)
        plan.partition_build(build_keys, build_row_ids)
        return plan

    def probe(self, plan, probe_keys: torch.Tensor, probe_row_ids: torch.Tensor | None):
        """-> probe row ids (in result order), pos, cnt, ids"""
        plan.partition_probe(probe_keys, probe_row_ids)
        plan.match()
        return plan.result()

    def local_join(self, build_keys: torch.Tensor, probe_keys: torch.Tensor, build_row_ids: torch.Tensor | None = None,
                   probe_row_ids: torch.Tensor | None = None):
        return self.probe(self.build(build_keys, build_row_ids, probe_keys.numel()), probe_keys, probe_row_ids)

    def column_sum(self, col: torch.Tensor) -> int:
        """wrap-around (mod 2^32) sum of a column, on the device (dbhip_reduce_sum_i32)"""
        from . import ops
        return int(ops.reduce_sum(col).item()) & 0xFFFFFFFF if col.numel() else 0


@dataclass
class PartitionedJoinResult:
    probe_row_ids: torch.Tensor  # global probe row id of every probe row this rank received
    pos: torch.Tensor            # offset into build_row_ids
    cnt: torch.Tensor            # number of matching build rows
    build_row_ids: torch.Tensor  # global build row ids grouped by key
    sent_rows: int               # rows this rank shipped to other ranks (both relations)
    recv_build_rows: int
    recv_probe_rows: int


# Largest single message (one peer's segment of one column) the exchange hands to the backend, in ELEMENTS.
# An ncclSend/ncclRecv pair of more than 2^28 uint32 (1 GiB) through RCCL 2.27.7 completes without an error and delivers
# garbage (PartitionedJoinHip --gpus 1 at 2^30 rows, caught by its conservation check).  Round 4 pinned the limit with the
# C++ engine's DWARF_BENCH_PJOIN_PIECE knob (host/pjoin_engine.cpp): 2^28 elements intact; 2^28 + 1, 3 * 2^27, 2^29 - 1
# and 2^29 damaged, and only the pieces above 2^28 — 2^30 BYTES per message is the limit, not the signed 32-bit byte
# count at 2^31.  Messages here are at most 2^27 elements (512 MiB), half the largest size seen to work.  A hash bucket
# of 2^30 rows over 2 ranks holds 2^28 +- ~12K rows: such a segment goes out in three rounds.  Both sides derive the
# rounds from the same split sizes, so the r-th piece sent to a peer meets the r-th piece received from it.
MAX_MESSAGE_ELEMS = 1 << 27


class _Done:
    """handle of an exchange that has already completed"""

    def __init__(self, out):
        self.out = out

    def wait(self):
        return self.out


class _Pending:
    """handle of collectives in flight on the backend's own stream; wait() orders the current stream after them and,
    when the exchange was cut into rounds, moves every round's pieces to their place in the output"""

    def __init__(self, rounds, out, keep):
        self.rounds, self.out, self.keep = rounds, out, keep  # `keep`: send buffers must outlive the collectives

    def wait(self):
        for work, stage, pieces in self.rounds:
            work.wait()
            if stage is not None:
                at = 0
                for dst_off, length in pieces:
                    self.out[dst_off: dst_off + length].copy_(stage[at: at + length])
                    at += length
        self.rounds, self.keep = [], None
        return self.out


def _offsets(splits):
    off, acc = [], 0
    for x in splits:
        off.append(acc)
        acc += int(x)
    return off


def _a2a(inp: torch.Tensor, out_splits, in_splits, group, largest: int, max_elems: int | None = None):
    """all_to_all_single -> handle.  `largest` = the largest segment between ANY two ranks (from the gathered count
    matrix): the number of rounds must be the same on every rank of the collective.  With the nccl (= RCCL) backend
    these are async collectives on device memory over xGMI.  A gloo group (CPU rehearsal of the exchange) only moves host memory, so device tensors are staged
    through the host there, synchronously.  No message exceeds `max_elems` elements (default MAX_MESSAGE_ELEMS):
    larger segments are sent in rounds, round r carrying elements [r*max, (r+1)*max) of every peer's segment."""
    max_elems = int(max_elems or MAX_MESSAGE_ELEMS)
    inp = inp.contiguous()
    out_splits, in_splits = [int(x) for x in out_splits], [int(x) for x in in_splits]
    n_out = sum(out_splits)
    n_rounds = max(1, -(-int(largest) // max_elems))
    staged_on_host = inp.is_cuda and dist.get_backend(group) == "gloo"
    src = inp.cpu() if staged_on_host else inp
    out = torch.empty(n_out, dtype=inp.dtype, device=src.device)
    if n_rounds == 1:
        if staged_on_host:
            dist.all_to_all_single(out, src, output_split_sizes=out_splits, input_split_sizes=in_splits, group=group)
            return _Done(out.to(inp.device))
        work = dist.all_to_all_single(out, src, output_split_sizes=out_splits, input_split_sizes=in_splits, group=group,
                                      async_op=True)
        return _Pending([(work, None, None)], out, src)
    in_off, out_off = _offsets(in_splits), _offsets(out_splits)
    rounds, keep = [], [src]
    for r in range(n_rounds):
        lo = r * max_elems
        send_len = [max(0, min(max_elems, n - lo)) for n in in_splits]
        recv_len = [max(0, min(max_elems, n - lo)) for n in out_splits]
        send = torch.cat([src[in_off[q] + lo: in_off[q] + lo + send_len[q]] for q in range(len(in_splits))])
        stage = torch.empty(sum(recv_len), dtype=inp.dtype, device=src.device)
        pieces = [(out_off[q] + lo, recv_len[q]) for q in range(len(out_splits))]
        work = dist.all_to_all_single(stage, send, output_split_sizes=recv_len, input_split_sizes=send_len, group=group,
                                      async_op=not staged_on_host)
        keep.append(send)
        if staged_on_host:
            rounds.append((_Done(None), stage, pieces))
        else:
            rounds.append((work, stage, pieces))
    handle = _Pending(rounds, out, keep)
    if staged_on_host:
        return _Done(handle.wait().to(inp.device))
    return handle


def _exchange_counts(counts: torch.Tensor, group):
    """counts: this rank's P send counts (device or host int64/uint64).  One tiny all_gather gives every rank the
    whole P x P count matrix (row = sender): what it will receive, and — the same on every rank — the largest
    segment anywhere, from which all ranks derive the same number of exchange rounds.
    -> (send, recv, largest) as host ints — blocks the host until `counts` is computed and gathered."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    send = counts.to(torch.int64)
    if send.is_cuda and dist.get_backend(group) == "gloo":
        send = send.cpu()
    rows = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(rows, send, group=group)
    matrix = torch.stack(rows).cpu()
    return ([int(x) for x in matrix[rank].tolist()], [int(x) for x in matrix[:, rank].tolist()], int(matrix.max().item()))


class ExchangeError(RuntimeError):
    """the all-to-all did not deliver what was sent"""


def _verify_exchange(backend, sent_cols, recv_cols, group) -> None:
    """Conservation check of the exchange (the C++ dwarf's always-on check, hip_dwarfs.cpp): the wrap-around sums of
    the four columns over everything SENT by all ranks equal the sums over everything RECEIVED by all ranks.  One
    all_reduce of 8 words; raises ExchangeError on every rank when a column was not conserved."""
    sums = torch.tensor([backend.column_sum(t) for t in list(sent_cols) + list(recv_cols)], dtype=torch.int64)
    dev = sent_cols[0].device
    if dist.get_backend(group) != "gloo":
        sums = sums.to(dev)
    dist.all_reduce(sums, group=group)
    tot = [int(x) & 0xFFFFFFFF for x in sums.cpu().tolist()]
    bad = [c for c in range(4) if tot[c] != tot[4 + c]]
    if bad:
        raise ExchangeError(f"the exchange did not conserve column(s) {bad} (0/1 = build key / row id, 2/3 = probe): "
                            f"sent sums {tot[:4]}, received sums {tot[4:]}")


def partitioned_join(build_keys: torch.Tensor, probe_keys: torch.Tensor, build_first_row: int, probe_first_row: int,
                     group=None, backend=None, verify_exchange: bool = True,
                     max_message_elems: int | None = None) -> PartitionedJoinResult:
    """Join this rank's shards; collective over `group` (default: WORLD).  World size 1 = plain local join.
    verify_exchange: run the conservation check after the join (a few column reductions + one tiny all_reduce;
    benchmarks switch it off for the timed steps and keep it on for the warm-up ones)."""
    backend = backend or HipBackend()
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if world == 1:
        rid, pos, cnt, ids = backend.local_join(build_keys, probe_keys)  # row ids = local indices, in result order
        rid = rid if probe_first_row == 0 else (rid.to(torch.int64) + probe_first_row).to(torch.int32)
        ids_global = ids if build_first_row == 0 else (ids.to(torch.int64) + build_first_row).to(torch.int32)
        return PartitionedJoinResult(rid, pos, cnt, ids_global, 0, build_keys.numel(), probe_keys.numel())

    # R: partition, learn the split sizes, start its exchange
    rk, rr, rc = backend.partition(build_keys, build_first_row, world)
    r_send, r_recv, r_big = _exchange_counts(rc, group)
    rk_x = _a2a(rk, r_recv, r_send, group, r_big, max_message_elems)
    rr_x = _a2a(rr, r_recv, r_send, group, r_big, max_message_elems)
    # S: partition while R is on the links
    sk, sr, sc = backend.partition(probe_keys, probe_first_row, world)
    s_send, s_recv, s_big = _exchange_counts(sc, group)
    sk_x = _a2a(sk, s_recv, s_send, group, s_big, max_message_elems)
    sr_x = _a2a(sr, s_recv, s_send, group, s_big, max_message_elems)
    # build on the received R pairs while S is on the links, then probe
    rk_in, rr_in = rk_x.wait(), rr_x.wait()
    plan = backend.build(rk_in, rr_in, int(sum(s_recv)))
    sk_in, sr_in = sk_x.wait(), sr_x.wait()
    rid_out, pos, cnt, ids_global = backend.probe(plan, sk_in, sr_in)
    if verify_exchange:
        _verify_exchange(backend, (rk, rr, sk, sr), (rk_in, rr_in, sk_in, sr_in), group)
    sent = int(sum(r_send) - r_send[rank] + sum(s_send) - s_send[rank])
    return PartitionedJoinResult(rid_out, pos, cnt, ids_global, sent, rk_in.numel(), sk_in.numel())
