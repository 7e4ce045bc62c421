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
    # reconstruct: 2 data shards lost per block
    st = [[None if j in (0, 3) else (np.ascontiguousarray(blocks[b][j * S:(j + 1) * S]) if j < k and (j + 1) * S <= L
            else (np.concatenate([blocks[b][j * S:], np.zeros((j + 1) * S - L, dtype=np.uint8)]) if j < k else outs[b][j - k]))
           for j in range(k + m)] for b in range(min(nb, 64))]
    t0 = time.perf_counter()
    rec = rs.reconstruct_data(st)
    trec = time.perf_counter() - t0
    assert np.array_equal(rec[0][0], blocks[0][:S])
    print(json.dumps({
        "what": "host-pointer API, RS(10,4), 1 MiB blocks, PCIe-inclusive (H2D data + kernel + D2H parity)",
        "nblocks": nb,
        "encode_GiBps_best": round(nb * L / best / 2**30, 2),
        "encode_GiBps_median": round(nb * L / med / 2**30, 2),
        "reconstruct_data_GiBps_python_wrapper": round(len(st) * L / trec / 2**30, 2),
    }))


if __name__ == "__main__":
    main()
