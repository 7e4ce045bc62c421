#!/usr/bin/env python3
"""One device trip on pinned memory, by block count: gec_encode_hash_batch (a put batch: parity + 14 checksums) and
gec_decode_verify_batch (a get batch without block checksums: healthy, and with 4 of 14 shards of every block gone),
RS(10,4), 1 MiB blocks.  Median of `reps` calls after warm-up.  Which path a count takes is decided by the library
(GEC_FUSED_MAX_LEAVES, GEC_FUSED_GET_MAX_LEAVES); the A/B is by environment.  usage: trip_bench.py [reps]"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

import garage_amd as g  # noqa: E402
from garage_amd import _lib  # noqa: E402
from garage_amd.codec import host_alloc  # noqa: E402


def med(f, reps):
    for _ in range(3):
        f()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        f()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2]


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 25
    k, m, L = 10, 4, 1 << 20
    n = k + m
    S = g.shard_len(k, L)
    lib = _lib.lib
    u8 = ctypes.POINTER(ctypes.c_uint8)
    c = g.ReedSolomon(k, m)
    NB = 64
    arena = host_alloc(NB * n * S)
    arena[:] = np.random.default_rng(5).integers(0, 256, arena.size, dtype=np.uint8)
    out = host_alloc(NB * m * S)
    tag = " ".join(f"{v}={os.environ[v]}" for v in ("GEC_FUSED_MAX_LEAVES", "GEC_FUSED_GET_MAX_LEAVES") if v in os.environ)
    print(f"# trip_bench RS(10,4) 1 MiB blocks, pinned memory, median of {reps}; {tag or 'defaults'}")
    print("# blocks   put ms  GiB/s   get ms  GiB/s   degraded get ms  GiB/s")
    for nb in (1, 2, 3, 4, 6, 8, 12, 16, 20, 24, 32, 48, 64):
        ptrs = (ctypes.c_void_p * nb)(*[arena.ctypes.data + b * n * S for b in range(nb)])
        optrs = (ctypes.c_void_p * nb)(*[arena.ctypes.data + (b * n + k) * S for b in range(nb)])
        clens = (ctypes.c_size_t * nb)(*[L] * nb)
        sums = np.zeros((nb, n, 32), dtype=np.uint8)
        t_put = med(lambda: _lib.check(lib.gec_encode_hash_batch(c._h, nb, ptrs, clens, S, optrs, sums.ctypes.data_as(u8)), "put"), reps)
        sp = (ctypes.c_void_p * (nb * n))(*[arena.ctypes.data + i * S for i in range(nb * n)])
        op = (ctypes.c_void_p * (nb * n))()
        blen = (ctypes.c_size_t * nb)(*[k * S] * nb)
        t_get = med(lambda: _lib.check(lib.gec_decode_verify_batch(c._h, nb, sp, S, blen, op, sums.ctypes.data_as(u8), None), "get"), reps)
        lost = (0, 3, 7, 12)
        for b in range(nb):
            i = 0
            for j in lost:
                sp[b * n + j] = None
                if j < k:
                    op[b * n + j] = out.ctypes.data + (b * m + i) * S
                    i += 1
        t_deg = med(lambda: _lib.check(lib.gec_decode_verify_batch(c._h, nb, sp, S, blen, op, sums.ctypes.data_as(u8), None), "get"), reps)
        gib = nb * L / 2**30
        print(f"  {nb:4d}   {t_put * 1e3:7.3f} {gib / t_put:6.1f}  {t_get * 1e3:7.3f} {gib / t_get:6.1f}   {t_deg * 1e3:7.3f} {gib / t_deg:6.1f}", flush=True)


if __name__ == "__main__":
    main()
