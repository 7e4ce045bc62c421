#!/usr/bin/env python3
"""PCIe-inclusive throughput of the host-pointer entry points (what the Rust
shim calls): H2D of the data shards + kernel + D2H of the parity, per batch.
This is NOT bench.py's `value` (which is device-resident); DESIGN.md quotes it
separately.  usage: host_api_bench.py [nblocks] [reps]"""
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

import garage_amd as g  # noqa: E402
from garage_amd._lib import check, lib  # noqa: E402


def main():
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    k, m, L = 10, 4, 1 << 20
    S = g.shard_len(k, L)
    rs = g.ReedSolomon(k, m)
    rng = np.random.default_rng(1)
    blocks = [rng.integers(0, 256, L, dtype=np.uint8) for _ in range(nb)]
    outs = [np.empty((m, S), dtype=np.uint8) for _ in range(nb)]
    lens = (ctypes.c_size_t * nb)(*[L] * nb)
    ptrs = (ctypes.c_void_p * nb)(*[b.ctypes.data for b in blocks])
    optrs = (ctypes.c_void_p * nb)(*[o.ctypes.data for o in outs])
    check(lib.gec_encode_batch(rs._h, nb, ptrs, lens, S, optrs), "warm")
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        check(lib.gec_encode_batch(rs._h, nb, ptrs, lens, S, optrs), "encode")
        ts.append(time.perf_counter() - t0)
    best, med = min(ts), sorted(ts)[len(ts) // 2]
    # reconstruct_data through the raw C ABI: data shards 0 and 3 of every block lost
    n = k + m
    padded = [np.concatenate([blocks[b], np.zeros(k * S - L, dtype=np.uint8)]) for b in range(nb)]
    rec = [np.empty((2, S), dtype=np.uint8) for _ in range(nb)]
    sp = (ctypes.c_void_p * (nb * n))()
    op = (ctypes.c_void_p * (nb * n))()
    for b in range(nb):
        for j in range(n):
            if j in (0, 3):
                sp[b * n + j] = None
                op[b * n + j] = rec[b][0 if j == 0 else 1].ctypes.data
            else:
                sp[b * n + j] = padded[b].ctypes.data + j * S if j < k else outs[b].ctypes.data + (j - k) * S
    check(lib.gec_reconstruct_batch(rs._h, nb, sp, op, S, 1), "warm")
    tr = []
    for _ in range(reps):
        t0 = time.perf_counter()
        check(lib.gec_reconstruct_batch(rs._h, nb, sp, op, S, 1), "reconstruct")
        tr.append(time.perf_counter() - t0)
    assert np.array_equal(rec[0][0], padded[0][:S]) and np.array_equal(rec[-1][1], padded[-1][3 * S:4 * S])
    trec = min(tr)
    print(json.dumps({
        "what": "host-pointer API, RS(10,4), 1 MiB blocks, PCIe-inclusive (H2D data + kernel + D2H parity)",
        "nblocks": nb,
        "encode_GiBps_best": round(nb * L / best / 2**30, 2),
        "encode_GiBps_median": round(nb * L / med / 2**30, 2),
        "reconstruct_data_2_lost_GiBps_best": round(nb * L / trec / 2**30, 2),
    }))


if __name__ == "__main__":
    main()
