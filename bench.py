#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X dwarf backend.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--dwarf scan|sort|groupby|join|all] [--no-cpu] [--no-pjoin] [--no-sweep]

One "step" = one pass of the hot path over one batch of synthetic input already resident in HBM.
N = 1 (default): the configuration BASELINE.json's metric is quoted on — TwoPassScan (stream compaction
x < 5 over int32 uniform in [1, 10000]) at 2^28 rows, with a cold-source control (`value_cold`: the same steps over
five 1 GiB copies in rotation).  The other single-GPU configurations of BASELINE.json (radix sort 2^24, group-by 2^26
rows / 2^16 groups, hash join 2^26 x 2^26 — row-ordered probe and radix join) and the 2^30 x 2^30 join on one GPU
(`pjoin_p1`, the P = 1 point of the partitioned join) are measured in the same run with a few steps each and reported
under "dwarfs" (they are not the headline value).
N > 1, one rank per GPU — either under torch.distributed.run (RANK / WORLD_SIZE in the environment) or as plain
`python bench.py --gpus N`, which then starts its own N rank processes before it imports torch or touches a GPU
(self_launch; a WORLD_SIZE that differs from --gpus, or a rank without a GPU of its own, ends non-zero instead of
printing a line with another n_gpus): the headline stays the same metric — scan does not
shard ("replicas only"), so every rank runs the 2^28 scan on its own GPU and `value` is the aggregate (weak
scaling) — and the one part of the path that does shard, the radix-partitioned hash join 2^30 x 2^30 with its
RCCL all-to-all, is measured in the same run (strong scaling) and reported under "pjoin": on the C++ engine of the
PartitionedJoinHip dwarf (libdbench.so, one rank per process, RCCL called from C++; device-side checks of the
exchange and of the result in the warm-up) and, under "pjoin.torch_distributed_host", on the torch.distributed host —
each with its single-GPU time taken in this run on rank 0, i.e. the speed-up over one GPU is in the line itself.  The
section runs in CHILD processes (every rank starts `bench.py --pjoin-child`, own process group one port up): a leg that
hangs is cut by a watchdog (DBENCH_PJOIN_DEADLINE_S, default 600 s), a leg that takes its process down (a GPU memory
fault aborts the process) costs `pjoin` an `error` entry, never the contract line (tests/test_gpu_bench_launcher.py).
Rehearsal knobs (not for reported numbers): DBENCH_BACKEND=gloo lets several ranks share one GPU (the C++ RCCL leg
is skipped), DBENCH_PJOIN_LOG2 shrinks the partitioned join.

Prints ONE JSON line on rank 0 with the contract fields plus "roofline" and "cpu_baseline".
The oracle (oracle/) is used ONLY for the cpu_baseline leg and a one-off result check; the timed path is
libdbhip.so through the C ABI.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300-6970 GB/s is what a bare stream reaches
# Two SELF-MEASURED peaks (secondary rooflines of the join probe and the group-by; the contract `roofline` divides by the
# HBM spec above, never by these).  Source: tools/ubench.hip / tools/ubench_lds.hip, raw output tracked under profiles/
# (written by tools/profile_round.sh); _peak_source() names the file next to every figure that uses them.
RANDOM_GATHER_PEAK_G = 53.0  # G random 4..16-byte gathers/s into a table >= 64 MiB ("random 2^26 ops, table 256 MiB" line)
LDS_ATOMIC_PEAK_G = 4000.0   # G random ds_add/s chip-wide (indices from registers; the 830 G/s figure of round 1 came
                             # from a loop that was bound by its 4-byte index loads)


def _dist_env():
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    return rank, world, local


def _event_times_us(fn, steps: int):
    """device time of every step (HIP events on the launch stream = torch's current stream)"""
    import torch
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    return [a.elapsed_time(b) * 1e3 for a, b in evs]


def _drop_max_mean(xs):
    """the reference notebook's statistic: drop the slowest, mean of the rest (scripts/report-sample.ipynb:143-176)"""
    xs = sorted(xs)
    if len(xs) > 1:
        xs = xs[:-1]
    return sum(xs) / len(xs)


def _traffic_for(name: str):
    """HBM bytes per launch from committed rocprofv3 PMC passes (profiles/hbm_traffic.json), or None.
    The counters need their own profiler passes (MI355X_MICROARCH.md): they are NOT measured in this run."""
    p = ROOT / "profiles" / "hbm_traffic.json"
    if not p.exists():
        return None
    try:
        return json.loads(p.read_text()).get(name, {}).get("hbm_bytes_per_launch")
    except Exception:
        return None


def _traffic_source():
    """where `roofline.traffic` / `pmc_traffic_bytes` come from: file + the provenance block the profile tool wrote"""
    p = ROOT / "profiles" / "hbm_traffic.json"
    if not p.exists():
        return None
    try:
        src = json.loads(p.read_text()).get("_source")
    except Exception:
        src = None
    # the counters belong to ONE state of the device code: its content hash is in the file, this run's is computed here
    try:
        from dwarf_bench_amd.build import kernel_tree_sha256
        now = kernel_tree_sha256()
    except Exception:
        now = None
    then = (src or {}).get("kernel_tree_sha256")
    return {"file": "profiles/hbm_traffic.json", "measured_in_this_run": False, "provenance": src,
            "kernel_tree_sha256_now": now, "collected_on_this_kernel_tree": bool(now and then and now == then)}


def _peak_source():
    """where RANDOM_GATHER_PEAK_G / LDS_ATOMIC_PEAK_G come from: the newest tracked micro-benchmark output"""
    files = sorted((ROOT / "profiles").glob("r*_ubench.txt"))
    if not files:
        return {"file": None, "tool": "tools/ubench.hip, tools/ubench_lds.hip", "measured_in_this_run": False}
    head = files[-1].read_text().splitlines()[0:1]
    return {"file": f"profiles/{files[-1].name}", "tool": "tools/ubench.hip, tools/ubench_lds.hip (tools/profile_round.sh)",
            "measured_in_this_run": False, "provenance": head[0].lstrip("# ") if head else None}


# ---------------------------------------------------------------------------------------------------
# single-GPU dwarfs
# ---------------------------------------------------------------------------------------------------
def bench_scan(steps, warmup, log2n=28, filt=5, cold=False):
    import torch
    from dwarf_bench_amd import ops
    n = 1 << log2n
    src = ops.gen_uniform_u32(n, 42, 1, 10000)  # the reference's distribution (common/common.hpp:31-40)
    plan = ops.CopyIfLt(n)
    run = lambda: plan.launch(src, filt)
    for _ in range(warmup):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        run()
    torch.cuda.synchronize()
    wall_ms = (time.perf_counter() - t0) * 1e3 / steps
    ev = _event_times_us(run, steps)
    matches = plan.result().numel()
    alg_bytes = 4 * n + 4 * matches  # 4*N*(1+s)
    avg_us = sum(ev) / len(ev)
    # cold-buffer control: the timed loop above scans the SAME 1 GiB every step; here the steps rotate over five
    # distinct 1 GiB copies of the column (5 GiB >> the 256 MiB Infinity Cache), so no step can find its input on-die
    cold_ms = None
    if cold:
        copies = [src] + [src.clone() for _ in range(4)]
        for i in range(len(copies)):
            plan.launch(copies[i], filt)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            plan.launch(copies[i % len(copies)], filt)
        torch.cuda.synchronize()
        cold_ms = (time.perf_counter() - t0) * 1e3 / steps
        del copies
    return {
        "cold_ms_per_step": cold_ms,
        "rows": n, "ms_per_step": wall_ms, "kernel_us_avg": avg_us, "kernel_us_dropmax_mean": _drop_max_mean(ev),
        "kernel_us_min": min(ev), "mrows_per_s": n / (wall_ms * 1e3), "matches": matches,
        "algorithmic_bytes": alg_bytes, "achieved_gbs": alg_bytes / avg_us / 1e3,
        "workload": f"TwoPassScan copy_if(x<{filt}) 2^{log2n} int32 uniform[1,10000]",
        "src": src, "plan": plan,
    }


def bench_sort(steps, warmup, log2n=24, bits=8, reference_range=False):
    """full-range uint32 keys (BASELINE config), or reference_range: the reference's own data — `int` keys uniform in
    [1, 10000] (sort/radix.cpp:19), signed order; the two upper bytes are constant and their passes are skipped"""
    import torch
    from dwarf_bench_amd import ops
    n = 1 << log2n
    keys0 = ops.gen_uniform_u32(n, 42, 1, 10000) if reference_range else ops.gen_uniform_u32(n, 42, 0, 2**32 - 1)
    keys = keys0.clone()
    plan = ops.RadixSort(n, bits)

    def run():
        keys.copy_(keys0)  # every step sorts unsorted data (sort/radix.cpp:31)
        plan.launch(keys, signed=reference_range)

    for _ in range(warmup):
        run()
    ev = _event_times_us(run, steps)
    cp = _event_times_us(lambda: keys.copy_(keys0), steps)
    us = _drop_max_mean(ev) - _drop_max_mean(cp)
    assert ops.workspace_status(plan.ws) == 0
    passes = 32 // bits
    return {"rows": n, "kernel_us": us, "mkeys_per_s": n / us, "radix_bits": bits,
            "compulsory_bytes": 8 * n, "pass_model_bytes": 4 * n + passes * 8 * n,
            "frac_of_hbm_peak_compulsory": 8 * n / us / 1e3 / HBM_PEAK_GBS,
            "frac_of_hbm_peak_pass_model": (4 * n + passes * 8 * n) / us / 1e3 / HBM_PEAK_GBS,
            "workload": (f"Radix sort 2^{log2n} int32 keys in [1,10000] (reference data), {bits}-bit LSD digits"
                         if reference_range else f"Radix sort 2^{log2n} uint32 full-range keys, {bits}-bit LSD digits")}


def bench_groupby(steps, warmup, log2n=26, groups=1 << 16):
    from dwarf_bench_amd import ops
    n = 1 << log2n
    keys = ops.gen_uniform_u32(n, 42, 0, groups - 1)
    vals = ops.gen_uniform_u32(n, 43, 1, 10000)
    plan = ops.GroupBySum(n, groups)
    run = lambda: plan.launch(keys, vals)
    for _ in range(warmup):
        run()
    ev = _event_times_us(run, steps)
    plan.result()
    us = _drop_max_mean(ev)
    alg = 8 * n + 4 * groups
    return {"rows": n, "groups": groups, "kernel_us": us, "mrows_per_s": n / us, "algorithmic_bytes": alg,
            "achieved_gbs": alg / us / 1e3, "frac_of_hbm_peak": alg / us / 1e3 / HBM_PEAK_GBS,
            # one ds_add per row: far from the LDS atomic rate of the chip (so NOT what bounds the kernel: with 2^16
            # groups every row is read by two workgroups, one per 128 KiB key range, at 16 waves per CU)
            "lds_atomic_rate": {"kernel": "gb_aggregate_kernel", "achieved": n / us / 1e3, "peak": LDS_ATOMIC_PEAK_G,
                                "unit": "G ds_add/s", "frac": n / us / 1e3 / LDS_ATOMIC_PEAK_G,
                                "peak_source": _peak_source()},
            "workload": f"GroupBy SUM 2^{log2n} rows / {groups} groups"}


def bench_join(steps, warmup, log2n=26):
    from dwarf_bench_amd import ops
    n = 1 << log2n
    build = ops.gen_uniform_u32(n, 42, 0, n - 1)
    probe = ops.gen_uniform_u32(n, 43, 0, n - 1)
    plan = ops.HashJoin(n, n)
    for _ in range(max(1, warmup // 2)):
        plan.build(build)
        plan.probe(probe)
    b = _event_times_us(lambda: plan.build(build), steps)
    p = _event_times_us(lambda: plan.probe(probe), steps)
    plan.result()
    bu, pu = _drop_max_mean(b), _drop_max_mean(p)
    # the chip's random-gather rate MEASURED IN THIS RUN with the library's own gather (dbhip_gather_u32: out[i] =
    # table[idx[i]], 2^26 random 4-byte reads from a 256 MiB table — the access pattern of the probe without its hashing
    # and its stores of two columns); the figure the fraction below divides by stays the tracked micro-benchmark's
    import torch
    table = ops.gen_uniform_u32(1 << 26, 7, 0, 2**32 - 1)
    idx = ops.gen_uniform_u32(n, 8, 0, (1 << 26) - 1)
    gu = _drop_max_mean(_event_times_us(lambda: ops.gather_u32(table, idx), max(3, steps)))
    del table, idx
    torch.cuda.empty_cache()
    alg = 20 * n  # keys in both sides + ids + (count,pos)
    return {"rows": 2 * n, "build_us": bu, "probe_us": pu, "kernel_us": bu + pu, "mrows_per_s": 2 * n / (bu + pu),
            "algorithmic_bytes": alg, "achieved_gbs": alg / (bu + pu) / 1e3,
            "frac_of_hbm_peak": alg / (bu + pu) / 1e3 / HBM_PEAK_GBS,
            # what the probe's access pattern allows: one random table access per probe row; the chip serves ~53 G
            # random 16-B gathers/s from tables >= 64 MiB (tools/ubench.hip), whatever the HBM byte rate
            "roofline_gather": {"bound": "random_gather", "kernel": "jl_probe_kernel", "achieved": n / pu / 1e3,
                                "peak": RANDOM_GATHER_PEAK_G, "unit": "G gathers/s", "frac": n / pu / 1e3 / RANDOM_GATHER_PEAK_G,
                                "peak_source": _peak_source(),
                                "gather_rate_measured_in_this_run": {"kernel": "dbhip_gather_u32, 2^26 random 4-byte reads, 256 MiB table",
                                                                     "g_per_s": n / gu / 1e3, "us": gu}},
            "workload": f"HashJoin build+probe 2^{log2n} x 2^{log2n} uint32 keys (JoinOmnisci semantics)"}


def bench_join_radix(steps, warmup, log2n=26):
    """the same join as a radix join (dbhip_join_radix_*): both sides partitioned alike, one fused LDS build + probe
    launch; results in the probe side's partition order with row ids — what the partitioned multi-GPU join runs per rank"""
    from dwarf_bench_amd import ops
    n = 1 << log2n
    build = ops.gen_uniform_u32(n, 42, 0, n - 1)
    probe = ops.gen_uniform_u32(n, 43, 0, n - 1)
    plan = ops.RadixJoin(n, n)

    def run():
        plan.partition_build(build)
        plan.partition_probe(probe)
        plan.match()

    for _ in range(max(1, warmup)):
        run()
    us = _drop_max_mean(_event_times_us(run, steps))
    plan.result()
    alg = 24 * n  # keys of both sides in, ids out, (row id, position, count) per probe row out
    return {"rows": 2 * n, "kernel_us": us, "mrows_per_s": 2 * n / us, "algorithmic_bytes": alg,
            "achieved_gbs": alg / us / 1e3, "frac_of_hbm_peak": alg / us / 1e3 / HBM_PEAK_GBS,
            "workload": f"HashJoin 2^{log2n} x 2^{log2n} uint32 keys as a radix join (results with row ids, partition order)"}


def bench_crowded_keys(steps, uniform):
    """The dwarfs on keys that crowd — the reference's generators are uniform, a caller's data need not be: the worst
    shape found for each dwarf by tools/ab.py sort-shapes | groupby-skew | join-skew, next to the uniform time of the
    same dwarf from this run (lane copies of the LDS histograms, ballot ranking where a wave sees a
    crowd, a cross-wave sum before the LDS add, giant join partitions shared by all workgroups)."""
    import torch
    from dwarf_bench_amd import ops
    g = torch.Generator(device="cuda").manual_seed(7)
    out = {}

    def med(fn):
        fn()
        ts = sorted(_event_times_us(fn, steps))
        return ts[len(ts) // 2]

    n = 1 << 24
    spread = ops.gen_uniform_u32(n, 42, 0, 2**32 - 1)
    hot = torch.rand(n, device="cuda", generator=g) < 0.9
    keys0 = torch.where(hot, torch.full_like(spread, 0x1E3779B9), spread)
    keys = keys0.clone()
    copy_us = med(lambda: keys.copy_(keys0))
    for bits in (8, 4):
        plan = ops.RadixSort(n, bits)

        def run():
            keys.copy_(keys0)
            plan.launch(keys)

        us = med(run) - copy_us
        assert ops.workspace_status(plan.ws) == 0
        base = uniform.get(f"sort_{bits}bit", {}).get("kernel_us")
        out[f"sort_{bits}bit_90pct_one_value"] = {"rows": n, "kernel_us": us, "uniform_kernel_us": base,
                                                  "over_uniform": us / base if base else None}
    del spread, hot, keys0, keys
    n = 1 << 26
    vals = ops.gen_uniform_u32(n, 43, 1, 10000)
    for groups in (1 << 15, 1 << 16):
        one = torch.full((n,), groups // 3, dtype=torch.int32, device="cuda")
        plan = ops.GroupBySum(n, groups)
        us = med(lambda: plan.launch(one, vals))
        plan.result()
        entry = {"rows": n, "groups": groups, "kernel_us": us}
        if groups == uniform.get("groupby", {}).get("groups"):
            entry["uniform_kernel_us"] = uniform["groupby"]["kernel_us"]
            entry["over_uniform"] = us / uniform["groupby"]["kernel_us"]
        out[f"groupby_{groups}_groups_every_row_one_group"] = entry
        del one, plan
    del vals
    torch.cuda.empty_cache()
    build = ops.gen_uniform_u32(n, 42, 0, n - 1)
    build[::2] = 12345
    plan = ops.HashJoin(n, n)
    us = med(lambda: plan.build(build))
    assert ops.workspace_status(plan.ws) == 0
    base = uniform.get("join", {}).get("build_us")
    out["join_build_every_other_row_one_key"] = {"rows": n, "build_us": us, "uniform_build_us": base,
                                                 "over_uniform": us / base if base else None}
    return out


def bench_pjoin(steps, warmup, log2_total=30, dist=None, group=None):
    """Radix-partitioned hash join of 2^log2_total x 2^log2_total rows over all ranks (strong scaling: the total
    is fixed, every rank holds a contiguous 1/P shard of both key columns, generated in place)."""
    import torch
    from dwarf_bench_amd import ops, pjoin
    world = dist.get_world_size() if dist is not None else 1
    rank = dist.get_rank() if dist is not None else 0
    total = 1 << log2_total
    per = total // world
    lo = rank * per
    n_local = total - lo if rank == world - 1 else per
    build = ops.gen_uniform_u32(n_local, 42, 0, total - 1, first_index=lo)
    probe = ops.gen_uniform_u32(n_local, 43, 0, total - 1, first_index=lo)

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # `group`: with dist=None inside a multi-rank job, a one-rank group that makes this call a plain local join.
    # The previous result is dropped before each step so the caching allocator reuses its blocks: holding it
    # alive forces fresh hipMallocs of tens of GiB inside the timed region (measured: 2.9 s instead of 70 ms).
    res = None
    # warm-up steps run with the exchange's conservation check (sent == received column sums over all ranks; raises
    # on a mismatch), the timed steps without it
    for _ in range(warmup):
        res = None
        res = pjoin.partitioned_join(build, probe, lo, lo, group=group, verify_exchange=True)
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        res = None
        res = pjoin.partitioned_join(build, probe, lo, lo, group=group, verify_exchange=False)
    sync()
    ms = (time.perf_counter() - t0) * 1e3 / steps
    if dist is not None:
        t = torch.tensor([ms], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ms = float(t.item())
    matches = int(res.cnt.to(torch.int64).sum().item())
    stats = torch.tensor([matches, res.sent_rows, res.recv_build_rows + res.recv_probe_rows], dtype=torch.int64, device="cuda")
    mx = stats.clone()
    if dist is not None:
        dist.all_reduce(stats)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    return {"rows": 2 * total, "ms_per_step": ms, "mrows_per_s": 2 * total / (ms * 1e3), "matches": int(stats[0]),
            "rows_exchanged": int(stats[1]), "exchange_conserved_in_warmup": bool(warmup > 0 and world > 1),
            "max_over_mean_rows_per_rank": float(mx[2]) * world / max(int(stats[2]), 1),
            "workload": f"radix-partitioned HashJoin 2^{log2_total} x 2^{log2_total} uint32 keys over {world} GPU(s)"}


def bench_pjoin_native(steps, warmup, log2_total, dist, rank, world, local, solo=False, sub_joins=None):
    """The same join on the C++ host the `PartitionedJoinHip` dwarf runs on (dwarf_bench_amd/host/pjoin_engine.cpp
    through libdbench.so): one process per GPU, per-rank compute + exchange streams, counts by ncclAllGather, one
    RCCL send/recv group per relation.  solo: this process alone joins the whole input on its GPU (plain local join —
    the P = 1 point of the scaling curve); the other ranks of the job do not take part."""
    import torch
    from dwarf_bench_amd import pjoin_native
    total = 1 << log2_total
    multi = dist is not None and world > 1 and not solo
    nid = None
    if multi:
        box = [pjoin_native.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        nid = box[0]
    # sub_joins: the engine cuts every rank's rows into that many independent joins whose exchanges follow each other on
    # the links while the one before is joined locally (host/pjoin_engine.hpp Options::sub_joins; default 2); the knob
    # travels through the environment, read when the engine is created — the same on every rank
    if sub_joins is not None:
        os.environ["DWARF_BENCH_PJOIN_SUBJOINS"] = str(sub_joins)
    else:
        os.environ.pop("DWARF_BENCH_PJOIN_SUBJOINS", None)
    eng = pjoin_native.NativePartitionedJoin(total, rank if multi else 0, world if multi else 1, local, nid,
                                             direct_single=not multi)

    def sync():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    info = eng.info()
    last = None
    for _ in range(max(1, warmup)):
        last = eng.step()
    chk = eng.check()  # device-side checks + conservation over all ranks (collective), outside the timed region
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        last = eng.step()
    sync()
    ms = (time.perf_counter() - t0) * 1e3 / steps
    eng.close()
    words = torch.tensor([chk["bad_pairs"], chk["bad_route"], chk["bad_rows"], chk["matches"], chk["sent_rows"],
                          chk["recv_build"] + chk["recv_probe"]], dtype=torch.int64, device="cuda")
    mx = words.clone()
    if multi:
        t = torch.tensor([ms], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ms = float(t.item())
        dist.all_reduce(words)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    w = [int(x) for x in words.cpu().tolist()]
    n_ranks = world if multi else 1
    # what RCCL itself reports (ncclCommCount on every rank's communicator: the smallest answer over the ranks) and the GPU
    # every rank ran on — so that "the exchange ran over N ranks on N GPUs" can be read off the line
    seen = torch.tensor([info["rccl_ranks_seen"]], dtype=torch.int64, device="cuda")
    devices = [f"rank {rank}: cuda:{info['device']} {info['device_name']}"]
    if multi:
        dist.all_reduce(seen, op=dist.ReduceOp.MIN)
        box = [None] * world
        dist.all_gather_object(box, devices[0])
        devices = box
    exchange = None
    if multi and last:
        # what the first real multi-GPU line needs to be readable: bytes per xGMI link and the rate the two send/recv
        # groups reached (rank 0's last step; a group's span on the exchange stream includes waiting for its peers).
        # Uniform keys: every rank sends 1/P of its shard of each relation to every rank, 8 bytes per row.
        link_bytes = total / n_ranks / n_ranks * 8
        exchange = {"links_per_gpu": n_ranks - 1, "bytes_per_link_per_relation": link_bytes,
                    "xgmi_link_peak_gbs": 153.0, "min_us_per_relation_at_link_peak": link_bytes / 153e3}
        for rel in ("r", "s"):
            us = last.get(f"exchange_{rel}_us") or 0.0
            exchange[f"exchange_{rel}_us"] = round(us, 1)
            exchange[f"achieved_gbs_per_link_{rel}"] = (link_bytes / us / 1e3) if us > 0 else None
    return {"rows": 2 * total, "sub_joins": 1 if not multi else (sub_joins or 2), "rccl_ranks_seen": int(seen.item()),
            "rank_devices": devices, "exchange_links": exchange, "ms_per_step": ms, "mrows_per_s": 2 * total / (ms * 1e3), "matches": w[3],
            "rows_exchanged": w[4], "bytes_sent_per_gpu": w[4] * 8 / n_ranks,
            "max_over_mean_rows_per_rank": float(mx[5]) * n_ranks / max(w[5], 1),
            "checks": {"damaged_pairs": w[0], "misrouted_keys": w[1], "wrong_probe_rows": w[2],
                       "exchange_conserved": bool(chk["conserved"]), "all_rows_delivered": w[5] == 2 * total},
            "phase_us_rank0_last_step": {k: round(v, 1) for k, v in (last or {}).items()},
            "host": "C++ pjoin::Engine (libdbench.so, the PartitionedJoinHip dwarf's host): per-rank compute + exchange "
                    "HIP streams, ncclAllGather of the count matrix, one ncclSend/ncclRecv group per relation"
                    if multi else "C++ pjoin::Engine, one rank: plain local join (no partition, no exchange)",
            "workload": f"radix-partitioned HashJoin 2^{log2_total} x 2^{log2_total} uint32 keys over {n_ranks} GPU(s)"}


class _Watchdog:
    """A multi-GPU leg that hangs (an RCCL call that never returns cannot be interrupted from Python) must not cost
    the whole line: after `seconds` rank 0 prints what has been measured so far, with the hung leg named, and every
    rank leaves."""

    def __init__(self, seconds, rank, out):
        import threading
        self.rank, self.out, self.leg, self.seconds = rank, out, "start", seconds
        self.timer = threading.Timer(seconds, self._bail)
        self.timer.daemon = True
        self.timer.start()

    def _bail(self):
        # runs on the timer thread while the main thread may still be filling `out`: serialise a snapshot (retry if the
        # dict changed under the copy), print it on rank 0, and leave with a NON-ZERO code on every rank — a hung process
        # that has touched the GPU is a failure (run_pjoin_children turns the child's exit code into pjoin.error)
        if self.rank == 0:
            import copy
            line = None
            for _ in range(5):
                try:
                    snap = copy.deepcopy(self.out)
                    snap.setdefault("pjoin", {})["error"] = (f"watchdog: leg '{self.leg}' did not finish within "
                                                             f"{self.seconds} s")
                    line = json.dumps(snap)
                    break
                except RuntimeError:  # dictionary changed size during iteration
                    time.sleep(0.05)
            print(line or json.dumps({"pjoin": {"error": f"watchdog: leg '{self.leg}' hung"}}), flush=True)
            sys.stdout.flush()
        os._exit(3)

    def cancel(self):
        self.timer.cancel()


def scan_selectivity_sweep(src, plan, n):
    """SURVEY 8(d)'s secondary sweep: filter in {101, 1001, 5001, 10001} (selectivity ~1 %, 10 %, 50 %, 100 %), 9
    iterations each, the reference notebook's drop-max-mean; algorithmic bytes 4*n*(1+s).  Both entry points are timed
    (dbhip_copy_if_lt_i32: two launches through a staging buffer; dbhip_copy_if_lt_dense_i32: one launch, final
    positions at first write); `kernel_us` is the one a plan picks by itself from the selectivity of its previous
    call (ops.CopyIfLt.DENSE_ABOVE), named in `variant`."""
    out = {}
    for filt in (101, 1001, 5001, 10001):
        t = {}
        for name, dense in (("two_launch", False), ("dense", True)):
            run = lambda: plan.launch(src, filt, dense=dense)
            run()
            t[name] = _drop_max_mean(_event_times_us(run, 9))
        m = plan.result().numel()
        variant = "dense" if m / n > plan.DENSE_ABOVE else "two_launch"
        us = t[variant]
        out[str(filt)] = {"selectivity": m / n, "variant": variant, "kernel_us": us, "two_launch_us": t["two_launch"],
                          "dense_us": t["dense"], "mrows_per_s": n / us,
                          "frac_of_hbm_peak": (4 * n + 4 * m) / us / 1e3 / HBM_PEAK_GBS}
    return out


def cpu_baseline_scan(src_dev, filt, budget_s=12.0):
    """The oracle's chunked scan (scan.cl restated, T chunks on T threads) on the WHOLE column of the timed
    configuration (scan/scan.cpp:107-128 times the full input), repeated for about budget_s seconds."""
    import numpy as np
    from oracle import pyoracle as po
    cores = os.cpu_count() or 1
    m = src_dev.numel()  # the full 2^28 rows (1 GiB on the host)
    host = src_dev.cpu().numpy()
    out = np.empty(m, dtype=np.int32)
    po.chunked_scan(host, filt, cores, out)  # warm (page faults)
    reps, t_total = 0, 0.0
    while t_total < budget_s and reps < 200:
        t0 = time.perf_counter()
        _, k = po.chunked_scan(host, filt, cores, out)
        t_total += time.perf_counter() - t0
        reps += 1
    # single-thread figure on a smaller slice
    m1 = min(m, 1 << 24)
    t0 = time.perf_counter()
    po.chunked_scan(host[:m1], filt, 1, out)
    t1 = time.perf_counter() - t0
    return {"value": m * reps / t_total / 1e6, "unit": "Mrows/s", "cores": cores, "kind": "port",
            "sample": f"the whole column (2^{m.bit_length() - 1} rows), {reps} passes, {cores} threads "
                      f"(oracle/dbo.c dbo_chunked_scan_i32 = scan/scan.cl:3-42 with T chunks)",
            "single_thread_mrows_per_s": m1 / t1 / 1e6}


def cpu_baselines_dwarfs(which):
    """CPU baselines of the other dwarfs: the oracle's multi-threaded restatements of the reference
    algorithms on bounded samples of the same synthetic columns, all host cores.  Reported, never a target."""
    import numpy as np
    from oracle import pyoracle as po
    cores = os.cpu_count() or 1
    out = {}
    if "sort" in which:
        n = 1 << 24
        keys0 = po.gen_uniform_u32(n, 42, 0, 2**32 - 1)
        tmp = np.empty_like(keys0)
        best, reps, t_total = None, 0, 0.0
        while t_total < 4.0 and reps < 20:
            k = keys0.copy()
            t0 = time.perf_counter()
            po.radix_sort_u32_mt(k, tmp, cores)
            dt = time.perf_counter() - t0
            t_total += dt
            reps += 1
            best = dt if best is None else min(best, dt)
        out["sort"] = {"value": n / (t_total / reps) / 1e6, "unit": "Mkeys/s", "cores": cores, "kind": "port",
                       "sample": f"the full 2^24 full-range keys, {reps} sorts (oracle dbo_radix_sort_u32_mt: parallel LSD "
                                 f"radix, the role TBBSort/std::sort play in sort/tbbsort.cpp:22, sort/radix.cpp:8-12)"}
    if "groupby" in which:
        n, groups = 1 << 26, 1 << 16  # the full configuration (groupby/groupby.cpp:57-94 times the full input)
        keys = po.gen_uniform_u32(n, 42, 0, groups - 1)
        vals = po.gen_uniform_u32(n, 43, 1, 10000)
        reps, t_total = 0, 0.0
        while t_total < 6.0 and reps < 20:
            t0 = time.perf_counter()
            po.groupby_hash(keys, vals, groups, threads=cores)
            t_total += time.perf_counter() - t0
            reps += 1
        out["groupby"] = {"value": n / (t_total / reps) / 1e6, "unit": "Mrows/s", "cores": cores, "kind": "port",
                          "sample": f"the whole 2^26-row columns, 2^16 groups, {reps} passes (oracle dbo_groupby_hash_u32 = the "
                                    f"reference's algorithm as it stands: ONE global CAS + fetch_add table of n slots, "
                                    f"groupby/groupby.cpp:58-93, hashtable.hpp:136-153 — every row of a group contends for "
                                    f"the same slot, which is what this figure measures on {cores} cores)"}
    if "join" in which:
        n = 1 << 26  # the full configuration (join/join_omnisci.cpp:78-88 times the full input); ~3 GiB of host tables
        build = po.gen_uniform_u32(n, 42, 0, n - 1)
        probe = po.gen_uniform_u32(n, 43, 0, n - 1)
        bs, ps, _ = po.join_omnisci_timings(build, probe, cores)
        out["join"] = {"value": 2 * n / (bs + ps) / 1e6, "unit": "Mrows/s", "cores": cores, "kind": "port",
                       "build_s": bs, "probe_s": ps,
                       "sample": "the whole 2^26-row key columns of the timed configuration, one build + probe (oracle "
                                 "dbo_join_build/dbo_join_probe = omnisci_hashtable.hpp:80-192 with std::atomic; ht_size = "
                                 "2*distinct(build) computed outside the timed part as join_omnisci.cpp:69 does)"}
    return out


# ---------------------------------------------------------------------------------------------------
def pjoin_section(args, dist, rank, world, local, barrier, out):
    """The one part of the hot path that shards (BASELINE north_star): the hash join, radix-partitioned across the
    ranks with an RCCL all-to-all bucket exchange.  STRONG scaling: 2^30 x 2^30 in total whatever N is.  Two hosts
    drive the same device kernels and are both reported: the C++ engine of the PartitionedJoinHip dwarf (`pjoin`, the
    headline of this section) and the torch.distributed one (`pjoin.torch_distributed_host`).  Each one's single-GPU
    reference point is measured in this same run, on rank 0's GPU, while the other ranks wait — so the speed-ups are
    self-contained in the line.  Fills out["pjoin"] on rank 0."""
    import torch
    n_gpus = world
    torch.cuda.empty_cache()
    pj_steps, pj_warm = max(1, min(args.steps, 5)), max(1, min(args.warmup, 2))
    pj_log2 = int(os.environ.get("DBENCH_PJOIN_LOG2", "30"))
    dog = _Watchdog(int(os.environ.get("DBENCH_PJOIN_DEADLINE_S", "600")), rank, out)
    section = {"metric": f"Mrows/s, radix-partitioned hash join 2^{pj_log2} x 2^{pj_log2} (build+probe rows / s, all GPUs)",
               "scaling": "strong", "n_gpus": n_gpus, "steps": pj_steps, "warmup": pj_warm}
    if rank == 0:
        out["pjoin"] = section
    try:
        dog.leg = "torch.distributed host, all ranks"
        pj = bench_pjoin(pj_steps, pj_warm, pj_log2, dist)
        torch.cuda.empty_cache()
        dog.leg = "torch.distributed host, one GPU"
        solo = dist.new_group(ranks=[0])  # collective call; only rank 0 uses it: its join below is purely local
        p1 = bench_pjoin(2, 1, pj_log2, None, group=solo) if rank == 0 else None
        barrier()
        torch.cuda.empty_cache()
        if rank == 0:
            section["torch_distributed_host"] = {
                "parallelism": f"hash-partitioned over {n_gpus} ranks, all_to_all bucket exchange (RCCL over xGMI) "
                               "overlapped with partition/build, local LDS-partitioned join per rank",
                **pj, "bytes_sent_per_gpu": pj["rows_exchanged"] * 8 / n_gpus,
                "single_gpu_ms_per_step": p1["ms_per_step"], "single_gpu_mrows_per_s": p1["mrows_per_s"],
                "speedup_vs_1gpu": p1["ms_per_step"] / pj["ms_per_step"],
                "matches_equal_single_gpu": pj["matches"] == p1["matches"]}
    except Exception as e:  # keep the line: the other host is still to come
        if rank == 0:
            section["torch_distributed_host"] = {"error": repr(e)}
    if rank == 0 and getattr(args, "pjoin_child", False):  # what is known so far, should the next leg take the process down
        print(json.dumps(out), flush=True)
        if os.environ.get("DBENCH_TEST_CHILD_FAULT"):  # test hook: the child of rank 0 dies like a process with a GPU fault
            import signal
            os.kill(os.getpid(), signal.SIGSEGV)
    if getattr(args, "pjoin_child", False) and os.environ.get("DBENCH_TEST_CHILD_HANG"):
        dog.leg = "test hook: a leg that never returns"  # every rank's child hangs like a collective that never completes
        time.sleep(10 ** 6)
    if os.environ.get("DBENCH_BACKEND", "nccl") == "nccl":  # the C++ engine talks RCCL: needs one GPU per rank
        try:
            dog.leg = "C++ engine, all ranks"
            cx = bench_pjoin_native(pj_steps, pj_warm, pj_log2, dist, rank, world, local)
            torch.cuda.empty_cache()
            dog.leg = "C++ engine, one GPU"
            c1 = bench_pjoin_native(2, 1, pj_log2, dist, rank, world, local, solo=True) if rank == 0 else None
            barrier()
            if rank == 0:
                section.update(cx)
                section.update({"single_gpu_ms_per_step": c1["ms_per_step"], "single_gpu_mrows_per_s": c1["mrows_per_s"],
                                "speedup_vs_1gpu": c1["ms_per_step"] / cx["ms_per_step"],
                                "matches_equal_single_gpu": cx["matches"] == c1["matches"]})
        except Exception as e:
            if rank == 0:
                section["error"] = repr(e)
            cx = None
        # the same join with one and with four sub-joins per step (the figures above are the default, two): how much of
        # the exchange the pipeline hides on THIS node — one join per step is the schedule of rounds 1-3.  A leg of its
        # own: whatever happens here leaves the figures above as they are.
        if os.environ.get("DBENCH_PJOIN_NO_SWEEP") is None:
            try:
                dog.leg = "C++ engine, all ranks, sub-join sweep"
                agreed = torch.tensor([0 if cx is None else 1], dtype=torch.int64, device="cuda")
                dist.all_reduce(agreed, op=dist.ReduceOp.MIN)  # every rank runs the sweep, or none does
                if int(agreed.item()) == 0:
                    raise RuntimeError("skipped: the leg above failed on some rank")
                sweep = {"2": {"ms_per_step": cx["ms_per_step"], "matches": cx["matches"]}}
                for sj in (1, 4):
                    sx = bench_pjoin_native(pj_steps, pj_warm, pj_log2, dist, rank, world, local, sub_joins=sj)
                    torch.cuda.empty_cache()
                    sweep[str(sj)] = {"ms_per_step": sx["ms_per_step"], "matches": sx["matches"],
                                      "exchange_links": sx["exchange_links"], "checks": sx["checks"]}
                if rank == 0:
                    section["sub_joins_sweep"] = sweep
                # The number of sub-joins is a knob of the engine (Options::sub_joins, DWARF_BENCH_PJOIN_SUBJOINS); which
                # value hides most of the exchange depends on the node's links.  The figures above (the default, two)
                # are safe by now: if another count was clearly faster in the sweep, the join is measured once more
                # with it, at the full number of steps, and THAT becomes the section's figure — with the count and the
                # default's time stated.  (Every rank sees the same all-reduced times, so every rank decides alike.)
                margin = float(os.environ.get("DBENCH_PJOIN_RERUN_MARGIN", "0.97"))
                best = min(("1", "4"), key=lambda k: sweep[k]["ms_per_step"])
                rerun = sweep[best]["ms_per_step"] < margin * cx["ms_per_step"]
            except Exception as e:
                rerun = False
                if rank == 0:
                    section["sub_joins_sweep"] = {"error": repr(e)}
            if rerun:
                try:
                    dog.leg = f"C++ engine, all ranks, {best} sub-joins"
                    bx = bench_pjoin_native(pj_steps, pj_warm, pj_log2, dist, rank, world, local, sub_joins=int(best))
                    torch.cuda.empty_cache()
                    if rank == 0 and bx["matches"] == cx["matches"] and bx["ms_per_step"] < cx["ms_per_step"]:
                        single = section.get("single_gpu_ms_per_step")
                        section.update(bx)
                        section["sub_joins_chosen_by"] = (f"the sweep on this node: {best} sub-joins per step; the engine's default "
                                                          f"(2) took {cx['ms_per_step']:.3f} ms per step in this run")
                        if single:
                            section["speedup_vs_1gpu"] = single / bx["ms_per_step"]
                except Exception as e:
                    if rank == 0:
                        section["sub_joins_rerun_error"] = repr(e)
    elif rank == 0:
        section["note"] = "rehearsal backend: ranks share GPUs, the C++ RCCL engine needs one GPU per rank and is skipped"
    dog.cancel()



def run_pjoin_children(args, rank):
    """every rank starts `bench.py --pjoin-child` with its own RANK / LOCAL_RANK / WORLD_SIZE and MASTER_PORT + 1,
    waits for it (deadline: DBENCH_PJOIN_DEADLINE_S + 60 s) and rank 0 returns the child's section"""
    import subprocess
    env = dict(os.environ)
    env["MASTER_PORT"] = str(int(os.environ.get("MASTER_PORT", "29500")) + 1)
    # under torch.distributed.run the ranks are clients of the launcher's store; the children's rank 0 must host its own
    env["TORCHELASTIC_USE_AGENT_STORE"] = "False"
    deadline = int(os.environ.get("DBENCH_PJOIN_DEADLINE_S", "600")) + 60
    cmd = [sys.executable, os.path.abspath(__file__), "--pjoin-child", "--gpus", str(args.gpus), "--steps", str(args.steps),
           "--warmup", str(args.warmup)]
    print(f"[bench] rank {rank}: starting the partitioned-join child (port {env['MASTER_PORT']}, deadline {deadline} s)", file=sys.stderr, flush=True)
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE if rank == 0 else subprocess.DEVNULL, text=True)
    try:
        text, _ = proc.communicate(timeout=deadline)
    except subprocess.TimeoutExpired:
        proc.kill()
        proc.communicate()
        return {"error": f"the partitioned-join child processes did not finish within {deadline} s"}
    if rank != 0:
        return None
    for line in reversed((text or "").strip().splitlines()):
        try:
            got = json.loads(line)
        except ValueError:
            continue
        if isinstance(got, dict) and "pjoin" in got:
            if proc.returncode != 0:
                got["pjoin"].setdefault("error", f"child exit code {proc.returncode}")
                got["pjoin"]["child_exit_code"] = proc.returncode  # 3 = the child's watchdog cut a hung leg
            return got["pjoin"]
    return {"error": f"the partitioned-join child of rank 0 ended with exit code {proc.returncode} and no result line"}


def _free_port_pair():
    """a port p with p and p + 1 both free on 127.0.0.1 (the partitioned-join children rendezvous one port up)"""
    import socket
    for _ in range(64):
        with socket.socket() as a:
            a.bind(("127.0.0.1", 0))
            p = a.getsockname()[1]
            if p >= 65535:
                continue
            with socket.socket() as b:
                try:
                    b.bind(("127.0.0.1", p + 1))
                except OSError:
                    continue
        return p
    raise SystemExit("bench.py: found no pair of free rendezvous ports on 127.0.0.1")


def self_launch(n, argv):
    """`python bench.py --gpus N` with no launcher around it (no WORLD_SIZE in the environment): this process becomes
    the launcher — it starts the N ranks as CHILD processes of itself (never an exec: nothing here has touched the GPU or
    imported torch, and nothing will), one per GPU, with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set the
    way torch.distributed.run sets them, passes rank 0's stdout through (the ONE JSON line), and leaves with a non-zero
    code if any rank did.  The reference's CLI starts everything it needs from one command too (main.cpp:13-96)."""
    import signal
    import subprocess
    port = int(os.environ.get("MASTER_PORT") or _free_port_pair())
    me = os.path.abspath(__file__)
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "GROUP_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
                    "DBENCH_SELF_LAUNCHED": "1"})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, me] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    print(f"[bench] --gpus {n} without a launcher: started ranks 0..{n - 1} as child processes "
          f"(pids {[p.pid for p in procs]}, rendezvous 127.0.0.1:{port})", file=sys.stderr, flush=True)
    # rank 0's stdout is passed through line by line on a thread, so a rank that dies early is noticed while rank 0
    # still waits in a collective for it
    import threading
    relay = threading.Thread(target=lambda: [print(l, end="", flush=True) for l in procs[0].stdout], daemon=True)
    relay.start()
    failed = None
    alive = set(range(n))
    while alive and failed is None:
        for r in sorted(alive):
            rc = procs[r].poll()
            if rc is None:
                continue
            alive.discard(r)
            if rc != 0:
                failed = (r, rc)
                break
        time.sleep(0.2)
    if failed is not None:
        # the others may sit in a collective that will never complete: give them a moment, then end exactly the
        # processes started above
        t_end = time.time() + 20
        while time.time() < t_end and any(procs[r].poll() is None for r in alive):
            time.sleep(0.2)
        for r in alive:
            if procs[r].poll() is None:
                procs[r].send_signal(signal.SIGTERM)
        for r in alive:
            try:
                procs[r].wait(timeout=10)
            except subprocess.TimeoutExpired:
                procs[r].kill()
    relay.join(timeout=10)
    if failed is not None:
        print(f"[bench] rank {failed[0]} ended with exit code {failed[1]}: this run measured nothing valid", file=sys.stderr, flush=True)
        return failed[1] if 0 < failed[1] < 256 else 1
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--dwarf", default="all", choices=["all", "scan", "sort", "groupby", "join"])
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-pjoin", action="store_true", help="skip the single-GPU 2^30 x 2^30 join (profiling passes)")
    ap.add_argument("--no-sweep", action="store_true", help="skip the scan selectivity sweep and the reference-range "
                    "sorts (counter passes: keeps every scan / sort dispatch at the BASELINE configuration)")
    ap.add_argument("--pjoin-child", action="store_true", help=argparse.SUPPRESS)  # see run_pjoin_children
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if args.gpus > 1 and not ("WORLD_SIZE" in os.environ and "RANK" in os.environ) and not args.pjoin_child:
        # no launcher around this process: be the launcher (before torch is imported or the GPU touched)
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))
    rank, world, local = _dist_env()
    if world != args.gpus:
        # a line that says n_gpus = WORLD_SIZE under a command that says --gpus N would be a silent wrong measurement
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE = {world}: start it as `python bench.py --gpus N` "
                         f"(it starts its own ranks) or under torch.distributed.run with --nproc-per-node N")
    # the host driver of this pool only supports dmabuf IPC: without this RCCL's buffer sharing between the ranks
    # fails with hipIpcGetMemHandle: invalid argument (already exported on the boxes; kept for any other launcher)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if os.environ.get("DBENCH_FAULT_DUMP"):  # debugging aid: dump every thread's stack after N seconds and exit
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["DBENCH_FAULT_DUMP"]), exit=True)
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product has no CPU path)")
    if os.environ.get("DBENCH_BACKEND", "nccl") != "nccl":  # rehearsal: ranks may share a GPU
        local %= torch.cuda.device_count()
    elif local >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} has no GPU of its own ({torch.cuda.device_count()} visible, --gpus "
                         f"{args.gpus}): RCCL needs one device per rank (DBENCH_BACKEND=gloo rehearses with shared GPUs)")
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("DBENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    n_gpus = max(world, 1)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if args.pjoin_child:  # the partitioned-join section only; rank 0 prints {"pjoin": section}
        child_out = {}
        print(f"[bench] child of rank {rank}: process group up", file=sys.stderr, flush=True)
        pjoin_section(args, dist, rank, world, local, barrier, child_out)
        if rank == 0:
            print(json.dumps(child_out), flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- headline: scan 2^28 (BASELINE.json's metric configuration)
    barrier()
    scan = bench_scan(args.steps, args.warmup, cold=(n_gpus == 1 and not args.no_sweep))
    barrier()
    ms = scan["ms_per_step"]
    if dist is not None:
        t = torch.tensor([ms], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ms = float(t.item())
    value = n_gpus * scan["rows"] / (ms * 1e3)  # Mrows/s over all GPUs

    out = {
        "metric": "Mrows/s per dwarf at 2^28 int32; achieved HBM GB/s vs peak",
        "value": value, "unit": "Mrows/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "int32", "data": "synthetic",
        "config": {"workload": scan["workload"], "rows": scan["rows"], "selectivity": scan["matches"] / scan["rows"],
                   "parallelism": "single GPU" if n_gpus == 1 else f"{n_gpus} independent replicas (scan does not shard)"},
        "roofline": {"bound": "hbm", "achieved": scan["achieved_gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": scan["achieved_gbs"] / HBM_PEAK_GBS, "traffic": _traffic_for("scan"),
                     "traffic_source": _traffic_source(),
                     "kernel": "scan_chunk_kernel (+ scan_move_kernel in the same event bracket)",
                     "kernel_us_avg": scan["kernel_us_avg"], "algorithmic_bytes": scan["algorithmic_bytes"],
                     "kernel_us_avg_note": "HIP events around every call in a second loop of `steps` calls (event "
                                           "records between the calls); ms_per_step is the wall clock of the "
                                           "back-to-back loop — two loops, so the two figures differ by ~1 %"},
    }
    if scan["cold_ms_per_step"] is not None:
        # the same scan with NO step finding its input in the Infinity Cache (five 1 GiB copies in rotation)
        out["value_cold"] = n_gpus * scan["rows"] / (scan["cold_ms_per_step"] * 1e3)
        out["cold_over_warm_time"] = scan["cold_ms_per_step"] / scan["ms_per_step"]

    if rank == 0:
        # one-off result check against the oracle on a bounded prefix (never inside the timed region)
        try:
            import numpy as np
            from oracle import pyoracle as po
            from dwarf_bench_amd import ops
            m = 1 << 22
            got = ops.copy_if_lt(scan["src"][:m], 5).cpu().numpy()
            out["parity_check"] = bool(np.array_equal(got, po.copy_if_lt(scan["src"][:m].cpu().numpy(), 5)))
        except Exception as e:  # pragma: no cover
            out["parity_check"] = f"skipped: {e}"
        if not args.no_cpu and n_gpus == 1:
            out["cpu_baseline"] = cpu_baseline_scan(scan["src"], 5)
        else:
            out["cpu_baseline"] = None
    sweep = (scan_selectivity_sweep(scan["src"], scan["plan"], scan["rows"])
             if (rank == 0 and n_gpus == 1 and not args.no_sweep) else None)
    del scan["src"], scan["plan"]
    torch.cuda.empty_cache()

    if rank == 0 and n_gpus == 1 and args.dwarf in ("all", "sort", "groupby", "join"):
        k = max(3, min(args.steps, 9))  # the reference scripts use --iterations=9
        dwarfs = {"scan": {x: scan[x] for x in ("rows", "kernel_us_avg", "kernel_us_min", "mrows_per_s", "achieved_gbs")}}
        dwarfs["scan"]["selectivity_sweep"] = sweep
        if args.dwarf in ("all", "sort"):
            dwarfs["sort_8bit"] = bench_sort(k, 2, 24, 8)
            dwarfs["sort_4bit"] = bench_sort(k, 2, 24, 4)
            if not args.no_sweep:  # (the profiler passes keep every sort dispatch at the BASELINE configuration)
                dwarfs["sort_8bit_reference_range"] = bench_sort(k, 2, 24, 8, reference_range=True)
                dwarfs["sort_4bit_reference_range"] = bench_sort(k, 2, 24, 4, reference_range=True)
        if args.dwarf in ("all", "groupby"):
            dwarfs["groupby"] = bench_groupby(k, 2)
        if args.dwarf in ("all", "join"):
            dwarfs["join"] = bench_join(max(3, k // 2), 2)
            torch.cuda.empty_cache()
            if not args.no_pjoin:  # (the counter passes run without it: it launches the same partition kernels as the build)
                dwarfs["join_radix"] = bench_join_radix(max(3, k // 2), 2)
                torch.cuda.empty_cache()
            if not args.no_pjoin:
                # the single-GPU point of the partitioned join's scaling curve: C++ host (the dwarf's engine), and the
                # torch.distributed host beside it
                dwarfs["pjoin_p1"] = bench_pjoin_native(3, 1, 30, None, 0, 1, local, solo=True)
                torch.cuda.empty_cache()
                dwarfs["pjoin_p1_torch_host"] = bench_pjoin(2, 1, 30, None)
        if args.dwarf == "all" and not args.no_sweep:
            torch.cuda.empty_cache()
            dwarfs["crowded_keys"] = bench_crowded_keys(5, dwarfs)
            torch.cuda.empty_cache()
        if not args.no_cpu:
            want = [w for w in ("sort", "groupby", "join") if args.dwarf in ("all", w)]
            for name, base in cpu_baselines_dwarfs(want).items():
                dwarfs["sort_8bit" if name == "sort" else name]["cpu_baseline"] = base
        for name, d in dwarfs.items():  # HBM bytes per call from the committed PMC passes, where collected
            t = _traffic_for(name)
            if t is not None:
                d["pmc_traffic_bytes"] = t
        out["dwarfs"] = dwarfs

    if n_gpus > 1:
        # The partitioned join runs in CHILD processes (one per rank, their own process group one port up): its RCCL
        # legs have only ever been rehearsed on one GPU, and a native fault there (a GPU memory fault aborts the
        # process; a hung collective cannot be interrupted) must not cost the contract line above.  Rank 0 reads its
        # child's JSON; anything else becomes pjoin.error.  DBENCH_PJOIN_INPROCESS=1 runs the section in this process.
        torch.cuda.empty_cache()
        if os.environ.get("DBENCH_PJOIN_INPROCESS"):
            pjoin_section(args, dist, rank, world, local, barrier, out)
        else:
            barrier()
            section = run_pjoin_children(args, rank)
            if rank == 0:
                out["pjoin"] = section
            barrier()

    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
