#!/usr/bin/env python3
"""Call latency of the host-pointer encode for SMALL batches (what a coalescing queue
in front of the FFI sees when few PutObject requests are in flight), beside the time the
oracle's AVX2 path needs for the same blocks on one host thread.  Answers "from how many
blocks per call does the GPU trip pay off".  usage: latency_bench.py [reps]"""
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
from oracle import rs_oracle as O  # noqa: E402  (tools/ bench only: the CPU column)


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    k, m, L = 10, 4, 1 << 20
    S = g.shard_len(k, L)
    rs = g.ReedSolomon(k, m)
    co = O.COracle()
    rng = np.random.default_rng(1)
    rows = []
    for nb in (1, 2, 3, 4, 8, 16, 32, 64, 256):
        blocks = [rng.integers(0, 256, L, dtype=np.uint8) for _ in range(nb)]
        outs = [np.empty((m, S), dtype=np.uint8) for _ in range(nb)]
        lens = (ctypes.c_size_t * nb)(*[L] * nb)
        ptrs = (ctypes.c_void_p * nb)(*[b.ctypes.data for b in blocks])
        optrs = (ctypes.c_void_p * nb)(*[o.ctypes.data for o in outs])
        for _ in range(3):
            check(lib.gec_encode_batch(rs._h, nb, ptrs, lens, S, optrs), "warm")
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            check(lib.gec_encode_batch(rs._h, nb, ptrs, lens, S, optrs), "encode")
            ts.append(time.perf_counter() - t0)
        ts.sort()
        # the same call with every buffer in pinned memory (gec_host_alloc): copy kernels, no staging memcpy
        from garage_amd.codec import host_alloc, host_free

        pb = [host_alloc(k * S) for _ in range(nb)]
        po = [host_alloc(m * S) for _ in range(nb)]
        for x, y in zip(pb, blocks):
            x[:L] = y
            x[L:] = 0
        pptrs = (ctypes.c_void_p * nb)(*[b.ctypes.data for b in pb])
        poptrs = (ctypes.c_void_p * nb)(*[o.ctypes.data for o in po])
        for _ in range(3):
            check(lib.gec_encode_batch(rs._h, nb, pptrs, lens, S, poptrs), "warm")
        tp = []
        for _ in range(reps):
            t0 = time.perf_counter()
            check(lib.gec_encode_batch(rs._h, nb, pptrs, lens, S, poptrs), "encode")
            tp.append(time.perf_counter() - t0)
        tp.sort()
        assert all(np.array_equal(po[i].reshape(m, S), outs[i]) for i in range(nb))
        for x in pb + po:
            host_free(x)
        cpu1 = co.bench_encode(k, m, S, nb, 5, co.AVX2 if co.has_avx2() else co.SCALAR, 1)
        rows.append({"blocks": nb, "gpu_call_us_median": round(ts[len(ts) // 2] * 1e6, 1), "gpu_call_us_min": round(ts[0] * 1e6, 1),
                     "gpu_GiBps": round(nb * L / ts[len(ts) // 2] / 2**30, 2),
                     "gpu_pinned_call_us_median": round(tp[len(tp) // 2] * 1e6, 1),
                     "cpu_1thread_us": round(cpu1 * 1e6, 1), "cpu_1thread_GiBps": round(nb * L / cpu1 / 2**30, 2)})
    print(f"{'blocks':>6} {'gpu call us (med/min)':>24} {'GiB/s':>8} {'pinned call us':>15} {'cpu 1 thread us':>16} {'GiB/s':>8}")
    for r in rows:
        print(f"{r['blocks']:>6} {r['gpu_call_us_median']:>14} /{r['gpu_call_us_min']:>8} {r['gpu_GiBps']:>8} {r['gpu_pinned_call_us_median']:>15} {r['cpu_1thread_us']:>16} {r['cpu_1thread_GiBps']:>8}")
    print(json.dumps(rows))


if __name__ == "__main__":
    main()
