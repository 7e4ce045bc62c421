#!/usr/bin/env python3
"""Soak of the host-pointer entry points on pinned memory (the zero-copy kernels): random codes, block counts, ragged
lengths and erasure patterns; every parity byte, rebuilt shard, verdict and checksum is compared with the CPU oracle /
hashlib.  usage: soak_host.py [seconds]   (test infrastructure: uses oracle/)"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

import garage_amd as g  # noqa: E402
from garage_amd import _lib  # noqa: E402
from garage_amd.codec import host_alloc, host_free  # noqa: E402
from oracle.rs_oracle import COracle  # noqa: E402


def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
    co = COracle()
    lib = _lib.lib
    rng = np.random.default_rng(2026)
    u8 = ctypes.POINTER(ctypes.c_uint8)
    codes = [(10, 4), (3, 1), (20, 8), (6, 10), (17, 3), (128, 2)]
    rs = {c: g.ReedSolomon(*c) for c in codes}
    t0, it, nbytes = time.time(), 0, 0
    while time.time() - t0 < secs:
        k, m = codes[it % len(codes)]
        n = k + m
        c = rs[(k, m)]
        L = int(rng.choice([1 << 20, 65536, 300_000, 4097 * k, 1_000_003])) if k < 64 else 64 * k * int(rng.integers(1, 40))
        nb = int(rng.integers(1, 90 if L > 500_000 else 300))
        if it % 3 == 0:          # a third of the iterations are small trips: the one-launch kernel (fused.hpp)
            nb = int(rng.integers(1, 9))
        S = g.shard_len(k, L)
        lens = [L if rng.random() < 0.7 else int(rng.integers(0, L + 1)) for _ in range(nb)]
        arena = host_alloc(nb * n * S)
        st = arena.reshape(nb, n, S)
        st[:] = 0x33
        padded = np.zeros((nb, k * S), dtype=np.uint8)
        for b in range(nb):
            padded[b, :lens[b]] = rng.integers(0, 256, lens[b], dtype=np.uint8)
            st[b, :k].reshape(-1)[:lens[b]] = padded[b, :lens[b]]
        want = co.encode_batch(k, m, padded.reshape(nb, k, S), co.AVX2, threads=16)
        ptrs = (ctypes.c_void_p * nb)(*[arena.ctypes.data + b * n * S for b in range(nb)])
        optrs = (ctypes.c_void_p * nb)(*[arena.ctypes.data + (b * n + k) * S for b in range(nb)])
        clens = (ctypes.c_size_t * nb)(*lens)
        sums = np.zeros((nb, n, 32), dtype=np.uint8)
        if it % 2:
            _lib.check(lib.gec_encode_hash_batch(c._h, nb, ptrs, clens, S, optrs, sums.ctypes.data_as(u8)), "encode_hash")
        else:
            _lib.check(lib.gec_encode_batch(c._h, nb, ptrs, clens, S, optrs), "encode")
        assert np.array_equal(st[:, k:], want), f"parity mismatch, iteration {it} RS({k},{m}) nb={nb} L={L}"
        st[:, :k] = padded.reshape(nb, k, S)          # the junk behind short blocks becomes the zero padding the shards carry
        if it % 2:
            b, j = int(rng.integers(nb)), int(rng.integers(n))
            assert sums[b, j].tobytes() == g.shardsum(st[b, j].tobytes()), f"shard checksum, iteration {it}"
        # scrub in one trip, one flipped bit
        vp = (ctypes.c_void_p * (nb * n))(*[arena.ctypes.data + i * S for i in range(nb * n)])
        ok = np.zeros(nb, dtype=np.uint8)
        bb, jj, off = int(rng.integers(nb)), int(rng.integers(n)), int(rng.integers(S))
        st[bb, jj, off] ^= 1 << int(rng.integers(8))
        _lib.check(lib.gec_verify_hash_batch(c._h, nb, vp, S, ok.ctypes.data_as(u8), sums.ctypes.data_as(u8)), "verify_hash")
        assert not ok[bb] and int((ok == 0).sum()) == 1, f"verdicts, iteration {it}"
        assert sums[bb, jj].tobytes() == g.shardsum(st[bb, jj].tobytes())
        st[bb, jj, off] = (padded.reshape(nb, k, S)[bb, jj, off] if jj < k else want[bb, jj - k, off])
        # rebuild in one trip: a random erasure pattern per block
        ref = st.copy()
        out = host_alloc(nb * m * S)
        sp = (ctypes.c_void_p * (nb * n))()
        op = (ctypes.c_void_p * (nb * n))()
        lost = []
        for b in range(nb):
            ls = sorted(int(x) for x in rng.choice(n, size=int(rng.integers(1, m + 1)), replace=False))
            lost.append(ls)
            for j in range(n):
                sp[b * n + j] = None if j in ls else arena.ctypes.data + (b * n + j) * S
            for i, j in enumerate(ls):
                op[b * n + j] = out.ctypes.data + (b * m + i) * S
        ins = np.zeros((nb, n, 32), dtype=np.uint8)
        outs = np.zeros((nb, n, 32), dtype=np.uint8)
        _lib.check(lib.gec_reconstruct_hash_batch(c._h, nb, sp, op, S, 0, ins.ctypes.data_as(u8), outs.ctypes.data_as(u8)), "reconstruct_hash")
        ob = out.reshape(nb, m, S)
        for b in range(nb):
            for i, j in enumerate(lost[b]):
                assert np.array_equal(ob[b, i], ref[b, j]), f"rebuilt shard, iteration {it} block {b} shard {j}"
        b = int(rng.integers(nb))
        j = lost[b][0]
        assert outs[b, j].tobytes() == g.shardsum(ref[b, j].tobytes()), f"checksum of a rebuilt shard, iteration {it}"
        # the read path in one trip (gec_decode_verify_batch, no block checksums): a random erasure pattern per block, blocks
        # that need no decode among them -- the one-launch kernel for small trips, pipelined pieces for big ones
        sp2 = (ctypes.c_void_p * (nb * n))()
        op2 = (ctypes.c_void_p * (nb * n))()
        lost2, reb = [], {}
        for b in range(nb):
            ls = sorted(int(x) for x in rng.choice(n, size=int(rng.integers(0, m + 1)), replace=False)) if rng.random() < 0.7 else []
            lost2.append(ls)
            i = 0
            for j in range(n):
                sp2[b * n + j] = None if j in ls else arena.ctypes.data + (b * n + j) * S
                if j in ls and j < k:
                    op2[b * n + j] = out.ctypes.data + (b * m + i) * S
                    reb[(b, j)] = (b, i)
                    i += 1
        out[:] = 0xAB
        ssums = np.zeros((nb, n, 32), dtype=np.uint8)
        blen = (ctypes.c_size_t * nb)(*[k * S] * nb)
        _lib.check(lib.gec_decode_verify_batch(c._h, nb, sp2, S, blen, op2, ssums.ctypes.data_as(u8), None), "decode_verify")
        for (b, j), (bb2, i) in reb.items():
            assert np.array_equal(ob[bb2, i], ref[b, j]), f"decode_verify: rebuilt shard, iteration {it} block {b} shard {j} nb={nb} RS({k},{m})"
        for b in set([0, nb - 1, int(rng.integers(nb))]):
            present = [j for j in range(n) if j not in lost2[b]][:k]
            for j in range(n):
                if j in present:
                    assert ssums[b, j].tobytes() == g.shardsum(ref[b, j].tobytes()), f"decode_verify: checksum, iteration {it} block {b} shard {j}"
                else:
                    assert not ssums[b, j].any(), f"decode_verify: checksum of a shard that was not read, iteration {it}"
        nbytes += nb * n * S * 4
        host_free(out)
        host_free(arena)
        it += 1
    print(f"soak_host OK: {it} iterations, {nbytes / 2**30:.1f} GiB through the link, {time.time() - t0:.1f} s, 0 mismatches")


if __name__ == "__main__":
    main()
