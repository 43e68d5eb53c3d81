"""development aid: randomized parity runs of the HIP paths against numpy (sizes, alignments, value distributions the
fixed test lists do not cover).  python tools/fuzz_gpu.py [seconds] [seed]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dwarf_bench_amd import ops

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
t_end = time.time() + budget
done = {"scan": 0, "dense": 0, "sort": 0, "groupby": 0, "xscan": 0, "reduce": 0, "join": 0, "radix_join": 0, "crowded": 0, "ujoin": 0}


def rand_n():
    kind = rng.integers(0, 4)
    if kind == 0:
        return int(rng.integers(0, 5000))
    if kind == 1:
        return int(rng.choice([4096, 8192, 32768, 65536, 1 << 20])) + int(rng.integers(-3, 4))
    if kind == 2:
        return int(rng.integers(5000, 3_000_000))
    return int(rng.integers(3_000_000, 20_000_000))


def rand_values(n, dtype):
    kind = rng.integers(0, 6)
    if kind == 0:
        v = rng.integers(0, 2**32, n, dtype=np.uint64)
    elif kind == 1:
        v = rng.integers(1, 10001, n, dtype=np.uint64)
    elif kind == 2:
        v = np.full(n, int(rng.integers(0, 2**32)), dtype=np.uint64)
    elif kind == 3:
        v = rng.integers(0, 4, n, dtype=np.uint64) << int(rng.integers(0, 30))
    elif kind == 4:
        v = np.sort(rng.integers(0, 2**32, n, dtype=np.uint64))[::-1].copy()
    else:
        v = rng.integers(0, 2**32, n, dtype=np.uint64) & np.uint64(0xFFFF00FF)
    return v.astype(np.uint32).view(dtype)


def fmix32_inv(h):
    """inverse of the joins' key hash (murmur3's 32-bit finaliser is a bijection)"""
    h = h.astype(np.uint64) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x7ED1B41D)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(13)
    h ^= h >> np.uint64(26)
    h = (h * np.uint64(0xA5CB9243)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(16)
    return h.astype(np.uint32)


def crowd(count, bits=None, top=None):
    """`count` distinct keys whose hashes share their top `bits` bits: whatever the join's geometry, they meet in one
    partition (or in a few neighbours when it has more than 2^bits partitions) — more keys than an LDS sub-table holds"""
    bits = int(rng.integers(12, 22)) if bits is None else bits
    top = int(rng.integers(0, 1 << bits)) if top is None else top
    space = 1 << (32 - bits)
    low = rng.choice(space, size=min(count, space), replace=False).astype(np.uint64)
    keys = fmix32_inv((np.uint64(top) << np.uint64(32 - bits)) | low)
    return keys[keys != 0xFFFFFFFF], bits, top


it = 0
while time.time() < t_end:
    it += 1
    n = rand_n()
    off = int(rng.integers(0, 4))
    # ---- scan, both entry points, unaligned views
    host = rand_values(n + off, np.int32)
    dev = torch.from_numpy(host).cuda()[off:]
    filt = int(rng.choice([np.iinfo(np.int32).min, -5, 0, 5, 5001, 2**30, np.iinfo(np.int32).max,
                           int(host[rng.integers(0, len(host))]) if len(host) else 1]))
    want = host[off:][host[off:] < filt]
    for dense in (False, True):
        got = ops.copy_if_lt(dev, filt, dense=dense).cpu().numpy()
        assert np.array_equal(got, want), ("scan", n, off, filt, dense)
        done["dense" if dense else "scan"] += 1
    # ---- reduce, exclusive scan
    assert int(ops.reduce_sum(dev)) & 0xFFFFFFFF == int(host[off:].astype(np.int64).sum()) & 0xFFFFFFFF, ("reduce", n, off)
    done["reduce"] += 1
    if n:
        u = torch.from_numpy(host.view(np.uint32).astype(np.int64).astype(np.int32)).cuda()[off:]
        exp = np.concatenate((np.zeros(1, np.uint64), np.cumsum(host[off:].view(np.uint32).astype(np.uint64))[:-1])) & np.uint64(0xFFFFFFFF)
        got = ops.exclusive_scan(u).cpu().numpy().view(np.uint32)
        assert np.array_equal(got, exp.astype(np.uint32)), ("xscan", n, off)
        done["xscan"] += 1
    # ---- sort (16-byte aligned columns), both widths, signed and unsigned
    if n:
        keys = rand_values(n, np.uint32)
        for bits in (8, 4):
            signed = bool(rng.integers(0, 2))
            d = torch.from_numpy(keys.view(np.int32).copy()).cuda()
            ops.radix_sort_(d, signed=signed, radix_bits=bits)
            got = d.cpu().numpy()
            exp = np.sort(keys.view(np.int32)) if signed else np.sort(keys).view(np.int32)
            assert np.array_equal(got, exp), ("sort", n, bits, signed)
            done["sort"] += 1
    # ---- group-by
    if n:
        groups = int(rng.choice([1, 7, 64, 1000, 4096, 40000, 65536, 100000]))
        gk = (rand_values(n, np.uint32).astype(np.uint64) % groups).astype(np.uint32)
        gv = rand_values(n, np.uint32)
        got = ops.groupby_sum(torch.from_numpy(gk.view(np.int32)).cuda(), torch.from_numpy(gv.view(np.int32)).cuda(), groups)
        exp = np.zeros(groups, dtype=np.uint64)
        np.add.at(exp, gk, gv.astype(np.uint64))
        assert np.array_equal(got.cpu().numpy().view(np.uint32), (exp & 0xFFFFFFFF).astype(np.uint32)), ("groupby", n, groups)
        done["groupby"] += 1
    # ---- one-to-many join, both forms: counts for every probe row, id lists for a sample of them
    nb, npb = min(rand_n(), 4_000_000), min(rand_n(), 4_000_000)
    if nb and npb:
        dom = int(rng.choice([4, 1000, max(nb // 4, 1), nb, 4 * nb, 2**32 - 1]))
        bk = rng.integers(0, dom, nb, dtype=np.uint64).astype(np.uint32)
        pk = rng.integers(0, dom, npb, dtype=np.uint64).astype(np.uint32)
        if dom <= 4 and nb > 200_000:  # a handful of keys with hundreds of thousands of rows each: keep it bounded
            bk = bk[:200_000].copy(); nb = bk.size
        if it % 3 == 0:  # a crowd: thousands of distinct keys in one partition (the spill tables), each with a few rows
            mine, bits, top = crowd(int(rng.integers(100, 30000)))
            per = int(rng.integers(1, 12))
            rows = np.repeat(mine[: max(1, min(mine.size, nb // per))], per)[:nb]
            bk[: rows.size] = rows
            bk = rng.permutation(bk)
            others, _, _ = crowd(2000, bits, top)  # the same partition: some of them build keys, most of them misses
            pk[::3] = mine[rng.integers(0, mine.size, pk[::3].size)]
            pk[1::3] = others[rng.integers(0, others.size, pk[1::3].size)]
            dom = -mine.size
            done["crowded"] += 1
        uniq, inv_cnt = np.unique(bk, return_counts=True)
        idx = np.searchsorted(uniq, pk)
        idx[idx >= uniq.size] = 0
        want_cnt = np.where(uniq[idx] == pk, inv_cnt[idx], 0).astype(np.uint32)
        order = np.argsort(bk, kind="stable")
        starts = np.concatenate((np.zeros(1, np.int64), np.cumsum(inv_cnt)[:-1]))
        d_bk, d_pk = torch.from_numpy(bk.view(np.int32)).cuda(), torch.from_numpy(pk.view(np.int32)).cuda()
        hj = ops.HashJoin(nb, npb)
        hj.build(d_bk); hj.probe(d_pk)
        pos, cnt, ids = (t.cpu().numpy() for t in hj.result())
        assert np.array_equal(cnt.view(np.uint32), want_cnt), ("join cnt", nb, npb, dom)
        for r in rng.integers(0, npb, 64):
            c = int(want_cnt[r])
            if c:
                exp_rows = order[starts[idx[r]]: starts[idx[r]] + c]
                assert np.array_equal(np.sort(ids[pos[r]: pos[r] + c]), np.sort(exp_rows)), ("join ids", nb, npb, dom, int(r))
        done["join"] += 1
        rid, rpos, rcnt, rids = (t.cpu().numpy() for t in ops.radix_join(d_bk, d_pk))
        assert np.array_equal(np.sort(rid), np.arange(npb)), ("radix join rows", nb, npb, dom)
        assert np.array_equal(rcnt.view(np.uint32), want_cnt[rid]), ("radix join cnt", nb, npb, dom)
        for j in rng.integers(0, npb, 64):
            c, r = int(rcnt[j]), int(rid[j])
            if c:
                exp_rows = order[starts[idx[r]]: starts[idx[r]] + c]
                assert np.array_equal(np.sort(rids[rpos[j]: rpos[j] + c]), np.sort(exp_rows)), ("radix join ids", nb, npb, dom, r)
        done["radix_join"] += 1
    # ---- unique-key payload join: hits carry (key, build value, probe value), misses three 0xFFFFFFFF
    nu = min(rand_n(), 3_000_000)
    if nu:
        if it % 2:
            uk, _, _ = crowd(min(nu, 200_000), bits=int(rng.integers(8, 14)))
        else:
            uk = np.unique(rng.integers(0, int(rng.choice([4 * nu, 2**32 - 1])), nu, dtype=np.uint64).astype(np.uint32))
        uk = rng.permutation(uk)
        uv = rng.integers(0, 2**32 - 1, uk.size, dtype=np.uint64).astype(np.uint32)
        m = min(rand_n(), 3_000_000) or 1
        qk = rng.integers(0, 2**32 - 1, m, dtype=np.uint64).astype(np.uint32)
        qk[::2] = uk[rng.integers(0, uk.size, qk[::2].size)]
        qv = rng.integers(0, 2**32 - 1, m, dtype=np.uint64).astype(np.uint32)
        uj = ops.UniqueJoin(uk.size, m)
        uj.build(torch.from_numpy(uk.view(np.int32)).cuda(), torch.from_numpy(uv.view(np.int32)).cuda())
        uj.probe(torch.from_numpy(qk.view(np.int32)).cuda(), torch.from_numpy(qv.view(np.int32)).cuda())
        ok_, ob, op_ = (t.cpu().numpy().view(np.uint32) for t in uj.result())
        srt = np.argsort(uk, kind="stable")
        at = np.searchsorted(uk[srt], qk)
        at[at >= uk.size] = 0
        hit = uk[srt][at] == qk
        miss = np.uint32(0xFFFFFFFF)
        assert np.array_equal(ok_, np.where(hit, qk, miss)), ("ujoin keys", uk.size, m)
        assert np.array_equal(ob, np.where(hit, uv[srt][at], miss)), ("ujoin build vals", uk.size, m)
        assert np.array_equal(op_, np.where(hit, qv, miss)), ("ujoin probe vals", uk.size, m)
        done["ujoin"] += 1
    if it % 10 == 0:
        print(f"{it} iterations {done}", flush=True)
print("fuzz ok", it, done)
