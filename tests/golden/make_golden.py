"""Generates tests/golden/ref_vectors.json by CALLING THE REFERENCE's own code — the few sources that
compile here with plain g++ (oracle/_ref/libdbref.so, built by oracle/Makefile from /root/reference where it
lies): hashfunctions.hpp (Murmur3 / Simple / Polynomial hashers), join_helpers.hpp (seq_join, order-insensitive
table equality), common/{result,options}.cpp (CSV writer, Result printing, DeviceType parse/print).

Run in the build container only (needs /root/reference); the JSON it writes is data (inputs + expected outputs)
and is what travels.  Usage:  python tests/golden/make_golden.py
"""
from __future__ import annotations

import ctypes as C
import json
import os
import sys
import tempfile
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import pyoracle  # noqa: E402


def main() -> None:
    R = pyoracle.ref_lib()
    if R is None:
        raise SystemExit("oracle/_ref/libdbref.so missing: run `make -C oracle ref` where /root/reference exists")
    out = {"_generator": "tests/golden/make_golden.py via oracle/_ref/libdbref.so (reference code)"}

    # ---- hashers (common/dpcpp/hashfunctions.hpp)
    keys = [0, 1, 2, 9, 10, 42, 99, 100, 12345, 65535, 65536, 99999, 100000, 123456, 1 << 20, 2147483647,
            2147483648, 4294967294]
    mur = []
    for seed in (0, 1, 42, 1000, 0xDEADBEEF):
        for sz in (64, 1000, 1 << 20, (1 << 27) - 1):
            mur.append({"seed": seed, "sz": sz, "keys": keys,
                        "hash": [R.ref_murmur3_x86_32(k, seed, sz) for k in keys]})
    out["murmur3"] = mur
    out["simple"] = [{"sz": sz, "keys": keys, "hash": [R.ref_simple_hash(k, sz) for k in keys]}
                     for sz in (1, 64, 20000, 1 << 27)]
    poly = []
    pkeys = [0, 1, 5, 9, 10, 19, 63, 64, 100, 999, 1000, 9999, 10000, 65535, 99999]
    for p in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37, 41, 43):
        for sz in (2048, 4096, 1 << 20):
            hv = [R.ref_polynomial_hash(k, p, sz) for k in pkeys]
            assert 0xFFFFFFFF not in hv, "prime not recovered"
            poly.append({"p": p, "sz": sz, "keys": pkeys, "hash": hv})
    out["polynomial"] = poly

    # ---- seq_join (join/join_helpers/join_helpers.hpp:86-104) on seeded inputs
    rng = np.random.default_rng(20240601)
    joins = []
    for na, nb, hi in ((7, 7, 8), (50, 40, 30), (200, 300, 100), (64, 64, 1000)):
        ak = rng.integers(0, hi, na, dtype=np.uint32)
        av = rng.integers(0, 1000, na, dtype=np.uint32)
        bk = rng.integers(0, hi, nb, dtype=np.uint32)
        bv = rng.integers(0, 1000, nb, dtype=np.uint32)
        n = R.ref_seq_join(ak.ctypes, av.ctypes, C.c_size_t(na), bk.ctypes, bv.ctypes, C.c_size_t(nb), None, None,
                           None)
        ok, o1, o2 = (np.zeros(max(n, 1), dtype=np.uint32) for _ in range(3))
        R.ref_seq_join(ak.ctypes, av.ctypes, C.c_size_t(na), bk.ctypes, bv.ctypes, C.c_size_t(nb), ok.ctypes,
                       o1.ctypes, o2.ctypes)
        joins.append({"a_keys": ak.tolist(), "a_vals": av.tolist(), "b_keys": bk.tolist(), "b_vals": bv.tolist(),
                      "rows": np.stack([ok[:n], o1[:n], o2[:n]], 1).tolist()})
    out["seq_join"] = joins

    # ---- CSV writer + Result printing (common/result.cpp:5-93)
    csv_cases = []
    cases = [
        ("TwoPassScan", "CPU", "", [0, 0, 0], [1024, 1024, 4096],
         [[2289.4, 12000.0, 0, 0], [268.0, 1999.9, 0, 0], [200.6, 0.0, 0, 0]]),
        ("JoinOmnisci", "GPU", "", [1, 1], [2048, 67108864], [[5123.7, 0.0, 4000.2, 1123.5], [99.99, 0, 50.5, 49.49]]),
        ("GroupByLocal", "CPU", "total_time,group_by_time,reduction_time", [2], [128], [[1500.0, 0.0, 1200.4, 299.6]]),
    ]
    for name, dev, header, kinds, sizes, times in cases:
        with tempfile.TemporaryDirectory() as td:
            path = os.path.join(td, "r.csv")
            printed = C.create_string_buffer(4096)
            k = (C.c_int * len(kinds))(*kinds)
            s = (C.c_uint64 * len(sizes))(*sizes)
            flat = [x for row in times for x in row]
            t = (C.c_double * len(flat))(*flat)
            for _ in range(2):  # second call appends without a header (result.cpp:60-66)
                rc = R.ref_write_csv(path.encode(), name.encode(), dev.encode(), header.encode(), k, s, t,
                                     C.c_size_t(len(kinds)), printed, C.c_size_t(4096))
                assert rc == 0
            csv_cases.append({"dwarf": name, "device_type": dev, "header": header, "kinds": kinds, "buf_sizes": sizes,
                              "times_us": times, "csv_after_two_writes": open(path).read(),
                              "printed": printed.value.decode()})
    out["csv"] = csv_cases

    # ---- DeviceType parse/print (common/options.cpp:3-33)
    dts = []
    for s in ("cpu", "CPU", "gpu", "GPU", "igpu", "iGPU", "hip", "whatever"):
        buf = C.create_string_buffer(32)
        enum_val = R.ref_device_type_roundtrip(s.encode(), buf, C.c_size_t(32))
        dts.append({"in": s, "enum": enum_val, "to_string": buf.value.decode()})
    out["device_type"] = dts

    dst = Path(__file__).with_name("ref_vectors.json")
    dst.write_text(json.dumps(out, indent=1))
    print("wrote", dst, dst.stat().st_size, "bytes")


if __name__ == "__main__":
    main()
