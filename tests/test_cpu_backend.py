"""Parity of libgarage_ec's CPU backend (GEC_BACKEND_CPU, garage_amd/csrc/ec_cpu.cpp) with the oracle, through the
C ABI's host-pointer entry points -- the same checks the HIP path gets in tests/test_gpu_parity.py, bit-exact.

This is BASELINE config 1's codec ("RS(3,1) on 64 KiB blocks, CPU ... path via BlockManager (plumbing, no GPU)") and
what a node without a GPU runs on.  The backend picks one of three kernels at run time (AVX-512 + GFNI bit matrices,
AVX2 split-nibble, scalar product table); every kernel the host supports is checked (in a subprocess: the choice is
read once per process from GEC_CPU_ISA)."""
import ctypes
import hashlib
import os
import subprocess
import sys

import numpy as np
import pytest

import garage_amd as g
from garage_amd import _lib
from oracle import rs_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sha(a) -> str:
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.uint8).tobytes()).hexdigest()


def rand_blocks(seed, nb, k, S):
    return O.splitmix64_bytes(0x6761726167650001 + seed, nb * k * S).reshape(nb, k, S)


def cpu(k, m, **kw):
    return g.ReedSolomon(k, m, backend="cpu", **kw)


def encode(rs, data):
    """data (nb, k, S) -> parity (nb, m, S) through gec_encode_batch"""
    nb, k, S = data.shape
    return np.stack(rs.encode_blocks([data[b].tobytes() for b in range(nb)], S))


# ------------------------------------------------------------ known answers
def test_kat_one_encode_5_5():
    # SURVEY.md Appendix A.4.3, shards padded to the 64-byte geometry
    data = np.zeros((1, 5, 64), dtype=np.uint8)
    data[0, :, :2] = [[0, 1], [4, 5], [2, 3], [6, 7], [8, 9]]
    par = encode(cpu(5, 5), data)
    assert par[0, :, :2].tolist() == [[12, 13], [10, 11], [14, 15], [90, 91], [94, 95]]
    assert not par[0, :, 2:].any()


@pytest.mark.parametrize("k,m,L,digest", [
    (3, 1, 64, "a1a2e6472297a6c8fc595265fbc01ecb82954e148bc46107d916c1f876f5dbb6"),
    (10, 4, 64, "716c5f64eecea82d320f527c7837757e9a9effaaf219169d6e52e2e5f80b021d"),
    (10, 4, 4096, "473009bcb1d7ca2a705463acf8a5788977d23115a3d6a3bb29b880dcbb7a5860"),
    (20, 8, 64, "9faa9f8192c1e5573f78dac342858bbc6b2cd98f46f118d467e84d3cfc2c42a5"),
])
def test_golden_encode_digests(k, m, L, digest):
    # SURVEY.md Appendix A.4.6
    par = encode(cpu(k, m), O.golden_pattern(k, L)[None])
    assert sha(par[0]) == digest


def test_rs3_1_parity_is_xor():
    data = rand_blocks(7, 4, 3, 21888)
    par = encode(cpu(3, 1), data)
    assert np.array_equal(par[:, 0], data[:, 0] ^ data[:, 1] ^ data[:, 2])


# ------------------------------------------------- encode == oracle, bytewise
ENCODE_CASES = [
    # (k, m, S, nblocks) -- the shapes of tests/test_gpu_parity.py
    (3, 1, 21888, 16),     # BASELINE config 1
    (10, 4, 104896, 3),    # config 2 shard length
    (20, 8, 209728, 2),    # config 5 shard length
    (10, 4, 64, 1), (10, 4, 4160, 5), (1, 1, 128, 2), (2, 3, 192, 3), (5, 5, 320, 2), (17, 3, 1024, 2),
    (11, 4, 4160, 2), (12, 3, 1024, 2), (13, 4, 4160, 2), (16, 4, 2048, 3), (4, 8, 512, 2), (7, 5, 1024, 2),
    (8, 8, 4160, 2), (10, 8, 104896, 2), (11, 7, 4096, 2),
    (40, 12, 2048, 2),     # two passes of 8 + 4 rows
    (3, 9, 640, 3), (64, 16, 4160, 2), (120, 16, 256, 2), (121, 16, 256, 1), (100, 20, 256, 1), (30, 40, 192, 2),
    (200, 56, 64, 1),      # k + m = 256
    (250, 6, 128, 1), (255, 1, 64, 2),
    (10, 4, 16448, 2),     # one byte range past a 16 KiB work item
    (10, 4, 32768 + 64, 1),
]


@pytest.mark.parametrize("k,m,S,nb", ENCODE_CASES)
def test_encode_matches_oracle(coracle, k, m, S, nb):
    data = rand_blocks(k * 31 + m, nb, k, S)
    data[0, 0, :] = 0          # all-zero shard
    data[-1, -1, :] = 0xFF     # all-0xFF shard
    want = coracle.encode_batch(k, m, data, coracle.SCALAR, threads=4)
    assert np.array_equal(encode(cpu(k, m), data), want)


def test_ragged_and_empty_blocks(coracle):
    """Blocks shorter than k*S (the last data shards are short or absent and read as zero), a 1-byte block, the
    empty batch, an empty block."""
    k, m = 10, 4
    rs = cpu(k, m)
    S = 4160
    lens = [k * S, k * S - 1, 5 * S + 17, S, 1, 63, 64, 65, 0]
    rng = np.random.default_rng(5)
    blocks = [rng.integers(0, 256, n, dtype=np.uint8).tobytes() for n in lens]
    par = np.stack(rs.encode_blocks(blocks, S))
    data = np.zeros((len(lens), k, S), dtype=np.uint8)
    for b, blk in enumerate(blocks):
        data[b].reshape(-1)[: len(blk)] = np.frombuffer(blk, dtype=np.uint8)
    assert np.array_equal(par, coracle.encode_batch(k, m, data, coracle.SCALAR))
    assert rs.encode_blocks([]) == []
    par2, sums = rs.encode_hash_blocks(blocks, S)
    assert np.array_equal(np.stack(par2), par)
    for b in range(len(lens)):
        for j in range(k + m):
            sh = data[b, j] if j < k else par[b, j - k]
            assert sums[b, j].tobytes() == g.shardsum(sh.tobytes())


# ------------------------------------------------- reconstruct == original
@pytest.mark.parametrize("lost", [(0,), (13,), (0, 3, 7, 9), (0, 3, 7, 11), (10, 11, 12, 13), (9, 10), (2, 5, 12)])
@pytest.mark.parametrize("data_only", [False, True])
def test_reconstruct_patterns(coracle, lost, data_only):
    k, m, S, nb = 10, 4, 4160, 4
    rs = cpu(k, m)
    data = rand_blocks(sum(lost) + 1, nb, k, S)
    st = np.concatenate([data, coracle.encode_batch(k, m, data, coracle.SCALAR)], axis=1)
    shards = [[None if j in lost else st[b, j] for j in range(k + m)] for b in range(nb)]
    rec = rs.reconstruct(shards, data_only=data_only)
    for b in range(nb):
        for j in range(k + m):
            if data_only and j >= k and j in lost:
                assert rec[b][j] is None
            else:
                assert np.array_equal(rec[b][j], st[b, j])


def test_reconstruct_mixed_patterns_in_one_call_and_unwanted_rows(coracle):
    """Every block its own erasure pattern (one decode plan per pattern, cached), and out entries left NULL are not
    computed (resync rebuilds only what is absent)."""
    k, m, S = 10, 4, 1024
    n = k + m
    rs = cpu(k, m)
    rng = np.random.default_rng(11)
    nb = 40
    data = rand_blocks(3, nb, k, S)
    st = np.concatenate([data, coracle.encode_batch(k, m, data, coracle.SCALAR)], axis=1)
    lib = _lib.lib
    sp = (ctypes.c_void_p * (nb * n))()
    op = (ctypes.c_void_p * (nb * n))()
    outs = {}
    for b in range(nb):
        lost = rng.choice(n, size=rng.integers(1, m + 1), replace=False)
        wanted = set(lost[: max(1, len(lost) // 2)].tolist())
        for j in range(n):
            if j in lost:
                sp[b * n + j] = None
                if j in wanted:
                    outs[(b, j)] = np.full(S, 0xAB, dtype=np.uint8)
                    op[b * n + j] = outs[(b, j)].ctypes.data
            else:
                sp[b * n + j] = st[b, j].ctypes.data
    _lib.check(lib.gec_reconstruct_batch(rs._h, nb, sp, op, S, 0), "gec_reconstruct_batch")
    for (b, j), buf in outs.items():
        assert np.array_equal(buf, st[b, j]), (b, j)
    cached, inversions = rs.cache_stats()
    assert 1 <= inversions <= nb
    # too few shards
    for j in range(5):
        sp[j] = None
    assert lib.gec_reconstruct_batch(rs._h, nb, sp, op, S, 0) == _lib.GEC_E_TOO_FEW_PRESENT


# ------------------------------------------------- verify, one-trip scrub / rebuild / read path
def test_verify_and_verify_hash(coracle):
    k, m, S, nb = 10, 4, 20480 + 64, 6
    rs = cpu(k, m)
    data = rand_blocks(21, nb, k, S)
    st = np.concatenate([data, coracle.encode_batch(k, m, data, coracle.SCALAR)], axis=1)
    assert rs.verify(st).all()
    bad = st.copy()
    bad[1, 3, S - 1] ^= 0x80      # last byte of a data shard
    bad[4, 12, 16384] ^= 1        # first byte of a parity shard's second work item
    ok, sums = rs.verify_hash(bad)
    assert ok.tolist() == [True, False, True, True, False, True]
    for b in range(nb):
        for j in range(k + m):
            assert sums[b, j].tobytes() == g.shardsum(bad[b, j].tobytes())


def test_decode_verify_batch_read_path(coracle):
    """gec_decode_verify_batch on the host cores: checksums of the first k shards in hand, rebuilt data shards, the
    block's own blake2sum -- against hashlib and the oracle."""
    k, m, S = 10, 4, 8256
    n = k + m
    rs = cpu(k, m)
    lib = _lib.lib
    nb = 5
    lens = [k * S, k * S - 100, 3 * S + 1, 1, k * S]
    data = np.zeros((nb, k, S), dtype=np.uint8)
    rng = np.random.default_rng(9)
    for b, L in enumerate(lens):
        data[b].reshape(-1)[:L] = rng.integers(0, 256, L, dtype=np.uint8)
    st = np.concatenate([data, coracle.encode_batch(k, m, data, coracle.SCALAR)], axis=1)
    losts = [(), (0,), (2, 9, 11), (0, 1, 2, 3), (13,)]
    sp = (ctypes.c_void_p * (nb * n))()
    rp = (ctypes.c_void_p * (nb * n))()
    reb = {}
    for b in range(nb):
        for j in range(n):
            if j in losts[b]:
                sp[b * n + j] = None
                if j < k:
                    reb[(b, j)] = np.zeros(S, dtype=np.uint8)
                    rp[b * n + j] = reb[(b, j)].ctypes.data
            else:
                sp[b * n + j] = st[b, j].ctypes.data
    ssums = np.zeros((nb, n, 32), dtype=np.uint8)
    bsums = np.zeros((nb, 32), dtype=np.uint8)
    blen = (ctypes.c_size_t * nb)(*lens)
    u8p = ctypes.POINTER(ctypes.c_uint8)
    _lib.check(lib.gec_decode_verify_batch(rs._h, nb, sp, S, blen, rp, ssums.ctypes.data_as(u8p), bsums.ctypes.data_as(u8p)),
               "gec_decode_verify_batch")
    for (b, j), buf in reb.items():
        assert np.array_equal(buf, st[b, j]), (b, j)
    for b in range(nb):
        used = [j for j in range(n) if j not in losts[b]][:k]
        for j in range(n):
            if j in used:
                assert ssums[b, j].tobytes() == g.shardsum(st[b, j].tobytes())
            else:
                assert not ssums[b, j].any()      # shards that were not read: untouched
        assert bsums[b].tobytes() == hashlib.blake2b(data[b].tobytes()[: lens[b]], digest_size=64).digest()[:32]
    # a missing data shard without an output buffer is an argument error
    rp[1 * n + 0] = None
    assert lib.gec_decode_verify_batch(rs._h, nb, sp, S, blen, rp, ssums.ctypes.data_as(u8p), None) == _lib.GEC_E_INVALID_ARG


def test_reconstruct_hash_batch(coracle):
    k, m, S, nb = 10, 4, 4160, 3
    n = k + m
    rs = cpu(k, m)
    data = rand_blocks(77, nb, k, S)
    st = np.concatenate([data, coracle.encode_batch(k, m, data, coracle.SCALAR)], axis=1)
    lost = (1, 12)
    sp = (ctypes.c_void_p * (nb * n))()
    op = (ctypes.c_void_p * (nb * n))()
    outs = {}
    for b in range(nb):
        for j in range(n):
            if j in lost:
                outs[(b, j)] = np.zeros(S, dtype=np.uint8)
                op[b * n + j] = outs[(b, j)].ctypes.data
            else:
                sp[b * n + j] = st[b, j].ctypes.data
    ins = np.zeros((nb, n, 32), dtype=np.uint8)
    osum = np.zeros((nb, n, 32), dtype=np.uint8)
    u8p = ctypes.POINTER(ctypes.c_uint8)
    _lib.check(_lib.lib.gec_reconstruct_hash_batch(rs._h, nb, sp, op, S, 0, ins.ctypes.data_as(u8p), osum.ctypes.data_as(u8p)),
               "gec_reconstruct_hash_batch")
    for b in range(nb):
        read = [j for j in range(n) if j not in lost][:k]
        for j in range(n):
            if j in lost:
                assert np.array_equal(outs[(b, j)], st[b, j])
                assert osum[b, j].tobytes() == g.shardsum(st[b, j].tobytes())
            elif j in read:
                assert ins[b, j].tobytes() == g.shardsum(st[b, j].tobytes())
            else:
                assert not ins[b, j].any()


@pytest.mark.parametrize("n", [0, 1, 127, 128, 129, 4095, 4096, 4097, 104896, (1 << 20) + 5])
def test_checksums_match_hashlib(n):
    rs = cpu(3, 1)
    d = bytes(O.splitmix64_bytes(n + 1, max(n, 8))[:n])
    assert rs.blake2sum_batch([d, b"abc"])[0] == hashlib.blake2b(d, digest_size=64).digest()[:32]
    assert rs.blake2sum_batch([b"abc"])[0].hex().startswith("ba80a53f981c4d0d6a2797b69f12f6e9")  # RFC 7693 appendix A
    assert rs.shardsum_batch([d])[0] == g.shardsum(d)


# ------------------------------------------------- errors mirror the HIP path's
def test_argument_errors():
    rs = cpu(10, 4)
    lib = _lib.lib
    S = 128
    blk = np.zeros(10 * S, dtype=np.uint8)
    par = np.zeros(4 * S, dtype=np.uint8)
    bp = (ctypes.c_void_p * 1)(blk.ctypes.data)
    pp = (ctypes.c_void_p * 1)(par.ctypes.data)
    ln = (ctypes.c_size_t * 1)(10 * S)
    assert lib.gec_encode_batch(rs._h, 1, bp, ln, 0, pp) == _lib.GEC_E_EMPTY_SHARD
    assert lib.gec_encode_batch(rs._h, 1, bp, ln, 100, pp) == _lib.GEC_E_INCORRECT_SHARD_SIZE
    assert lib.gec_encode_batch(rs._h, 1, bp, ln, 64, pp) == _lib.GEC_E_INCORRECT_SHARD_SIZE  # block longer than k*S
    assert lib.gec_encode_batch(rs._h, 1, None, ln, S, pp) == _lib.GEC_E_INVALID_ARG
    assert lib.gec_encode_batch(None, 1, bp, ln, S, pp) == _lib.GEC_E_INVALID_ARG
    assert lib.gec_encode_batch(rs._h, 0, None, None, S, None) == _lib.GEC_OK
    assert lib.gec_encode_hash_batch(rs._h, 1, bp, ln, S, pp, None) == _lib.GEC_E_INVALID_ARG
    sh = (ctypes.c_void_p * 14)(*[blk.ctypes.data] * 13 + [None])
    ok = (ctypes.c_uint8 * 1)()
    assert lib.gec_verify_batch(rs._h, 1, sh, S, ok) == _lib.GEC_E_TOO_FEW_SHARDS


# ------------------------------------------------- property sweep
def test_random_codes_round_trip(coracle):
    rng = np.random.default_rng(2024)
    for _ in range(25):
        k = int(rng.integers(1, 40))
        m = int(rng.integers(1, 20))
        S = int(rng.integers(1, 80)) * 64
        nb = int(rng.integers(1, 4))
        rs = cpu(k, m)
        data = rng.integers(0, 256, (nb, k, S), dtype=np.uint8)
        par = encode(rs, data)
        assert np.array_equal(par, coracle.encode_batch(k, m, data, coracle.SCALAR))
        st = np.concatenate([data, par], axis=1)
        lost = set(rng.choice(k + m, size=int(rng.integers(1, m + 1)), replace=False).tolist())
        rec = rs.reconstruct([[None if j in lost else st[b, j] for j in range(k + m)] for b in range(nb)])
        for b in range(nb):
            for j in lost:
                assert np.array_equal(rec[b][j], st[b, j])


def test_cauchy_family_round_trip():
    k, m, S = 10, 4, 1024
    rs = cpu(k, m, matrix="cauchy")
    data = rand_blocks(8, 2, k, S)
    par = encode(rs, data)
    C = O.build_matrix_cauchy(k, m)[k:]      # parity rows; numpy restatement of the product over GF(2^8)
    want = np.zeros((m, S), dtype=np.uint8)
    for r in range(m):
        for c in range(k):
            want[r] ^= O.MUL[C[r, c]][data[0, c]]
    assert np.array_equal(par[0], want)
    st = np.concatenate([data, par], axis=1)
    lost = (0, 5, 11, 13)
    rec = rs.reconstruct([[None if j in lost else st[b, j] for j in range(k + m)] for b in range(2)])
    for b in range(2):
        for j in lost:
            assert np.array_equal(rec[b][j], st[b, j])


# ------------------------------------------------- every kernel the host has
_ISA_SCRIPT = r"""
import sys, numpy as np
sys.path.insert(0, %r)
import garage_amd as g
from garage_amd._lib import lib
from oracle import rs_oracle as O
co = O.COracle()
print("isa", lib.gec_cpu_isa().decode())
for k, m, S, nb in [(3, 1, 21888, 16), (10, 4, 104896, 2), (20, 8, 4160, 2), (10, 4, 100 * 64 + 64, 3), (40, 12, 1024, 1)]:
    rs = g.ReedSolomon(k, m, backend="cpu")
    data = O.splitmix64_bytes(k + m, nb * k * S).reshape(nb, k, S)
    par = np.stack(rs.encode_blocks([data[b].tobytes()[: k * S - 37 * b] for b in range(nb)], S))
    d2 = data.copy()
    for b in range(nb):
        if b:
            d2[b].reshape(-1)[k * S - 37 * b:] = 0
    assert np.array_equal(par, co.encode_batch(k, m, d2, co.SCALAR)), (k, m)
    st = np.concatenate([d2, par], axis=1)
    lost = list(range(0, k + m, max(1, (k + m) // m)))[:m]
    rec = rs.reconstruct([[None if j in lost else st[b, j] for j in range(k + m)] for b in range(nb)])
    assert all(np.array_equal(rec[b][j], st[b, j]) for b in range(nb) for j in lost)
    assert rs.verify(st).all()
print("ok")
"""


@pytest.mark.parametrize("isa", ["scalar", "avx2", "gfni"])
def test_every_cpu_kernel(isa):
    env = dict(os.environ, GEC_CPU_ISA=isa, GEC_CPU_THREADS="3")
    r = subprocess.run([sys.executable, "-c", _ISA_SCRIPT % ROOT], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr
    got = r.stdout.split()[1]
    if isa == "scalar":
        assert got == "scalar"
    # a kernel the host cannot run falls back to the next one down; say which one was exercised
    print(f"requested {isa}: ran {got}")


# ------------------------------------------------- BASELINE config 1 through the C ABI, as SURVEY.md 8d words it
def test_baseline_config1_encode_and_reconstruct_one_erasure(coracle):
    """cfg 1: RS(3,1), L = 65536, S = 21888, batch 16, encode + reconstruct(1 erasure) via the C ABI, backend = cpu."""
    k, m, L, nb = 3, 1, 65536, 16
    S = g.shard_len(k, L)
    assert S == 21888
    rs = cpu(k, m)
    payload = O.splitmix64_bytes(0x6761726167650001 + 1, nb * L).reshape(nb, L)
    payload[0] = 0
    payload[1] = 0xFF
    par = np.stack(rs.encode_blocks([payload[b].tobytes() for b in range(nb)], S))
    data = np.zeros((nb, k * S), dtype=np.uint8)
    data[:, :L] = payload
    data = data.reshape(nb, k, S)
    assert np.array_equal(par, coracle.encode_batch(k, m, data, coracle.SCALAR))
    assert np.array_equal(par[:, 0], data[:, 0] ^ data[:, 1] ^ data[:, 2])   # RS(3,1): parity row is [1, 1, 1]
    st = np.concatenate([data, par], axis=1)
    for lost in range(k + m):
        rec = rs.reconstruct([[None if j == lost else st[b, j] for j in range(k + m)] for b in range(nb)])
        for b in range(nb):
            assert np.array_equal(rec[b][lost], st[b, lost])


def test_host_blake2b_eight_at_a_time_and_one_at_a_time_agree_with_hashlib():
    """blake2b_mb.hpp: the AVX-512 form (eight chains per core) and the scalar fallback, through the CPU codec's
    gec_blake2sum_batch / gec_shardsum_batch and libgarage_block's gbm_shardsum, on ragged batches."""
    import subprocess
    import sys

    code = r"""
import hashlib, sys
sys.path.insert(0, %r)
import numpy as np
import garage_amd as g
from garage_amd import block_native as bn
rs = g.ReedSolomon(10, 4, backend="cpu")
lens = list(range(0, 140)) + [255, 256, 257, 4095, 4096, 4097, 8191, 8192, 8193, 12288, 104896, 104897, 209728, (1 << 20) + 5]
msgs = [bytes((i * 7 + j * 13) & 255 for j in range(n)) for i, n in enumerate(lens)]
assert rs.blake2sum_batch(msgs) == [hashlib.blake2b(x, digest_size=64).digest()[:32] for x in msgs]
assert rs.shardsum_batch(msgs) == [g.shardsum(x) for x in msgs]
for x in msgs[::7]:
    assert bn.shardsum(x) == g.shardsum(x) and bn.blake2sum(x) == hashlib.blake2b(x, digest_size=64).digest()[:32]
# 1..17 equal messages (every group size), and the block checksum over k shard buffers (decode_verify on the CPU codec)
for n in range(1, 18):
    same = [msgs[-3]] * n
    assert rs.shardsum_batch(same) == [g.shardsum(msgs[-3])] * n
print("ok")
""" % ROOT
    for mode in ("auto", "scalar"):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, GEC_CPU_ISA=mode, GEC_CPU_THREADS="3"))
        assert r.returncode == 0 and "ok" in r.stdout, (mode, r.stdout, r.stderr[-2000:])


# ------------------------------------------------------------------ round 5: checksum kind 3 and per-block patterns on the host cores
def test_shardsum3_every_isa_path_agrees_with_the_oracle():
    """mlh64_host.hpp has an AVX-512, an AVX2 and a scalar form (GEC_CPU_ISA caps which one a process uses): all three -- each in a
    process of its own -- equal oracle/mlh64.py on lengths around every word, vector and leaf boundary."""
    import subprocess
    import sys

    code = r'''
import ctypes, sys
import numpy as np
sys.path.insert(0, %r)
from garage_amd import _lib
from oracle import mlh64
rng = np.random.default_rng(8)
for n in [0, 1, 3, 4, 5, 31, 32, 33, 63, 64, 65, 4092, 4093, 4095, 4096, 4097, 4160, 8191, 8192, 104896, 300001]:
    d = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
    out = ctypes.create_string_buffer(32)
    assert _lib.lib.gec_shardsum_host(3, d, n, out) == 0 and out.raw == mlh64.shardsum3(d), n
    assert _lib.lib.gec_shardsum_host(2, d, n, out) == 0 and out.raw != mlh64.shardsum3(d)
assert _lib.lib.gec_shardsum_host(7, b"x", 1, out) == _lib.GEC_E_INVALID_ARG
print("ok")
''' % ROOT
    for isa in ("auto", "avx2", "scalar"):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, GEC_CPU_ISA=isa), timeout=300)
        assert r.returncode == 0 and "ok" in r.stdout, (isa, r.stdout, r.stderr[-1500:])


def test_cpu_codec_checksum_kinds_and_siblings():
    from oracle import mlh64

    rs = g.ReedSolomon(10, 4, backend="cpu")
    assert rs.shardsum_kind == 3 and rs.background().shardsum_kind == 3
    rs2 = rs.with_shardsum(2)
    assert rs2.shardsum_kind == 2 and rs2.background().shardsum_kind == 2 and rs2.with_shardsum(3).shardsum_kind == 3
    blocks = [bytes(O.splitmix64_bytes(40 + i, n)) for i, n in enumerate([1 << 20, 70_001, 0, 4096 * 10])]
    S = g.shard_len(10, 1 << 20)
    p3, s3 = rs.encode_hash_blocks(blocks, S)
    p2, s2 = rs2.encode_hash_blocks(blocks, S)
    for b, blk in enumerate(blocks):
        shards = O.split_block(10, blk, S)
        assert np.array_equal(p3[b], p2[b])
        for j in range(14):
            payload = (shards[j] if j < 10 else p3[b][j - 10]).tobytes()
            assert s3[b, j].tobytes() == mlh64.shardsum3(payload) and s2[b, j].tobytes() == g.shardsum(payload, 2)
    h = ctypes.c_void_p()
    assert _lib.lib.gec_codec_create_ex2(10, 4, _lib.GEC_BACKEND_CPU, 0, 0, 5, ctypes.byref(h)) == _lib.GEC_E_INVALID_ARG   # no such kind
    assert _lib.lib.gec_codec_with_shardsum(None, 3, ctypes.byref(h)) == _lib.GEC_E_INVALID_ARG and _lib.lib.gec_codec_shardsum(None) == -1


@pytest.mark.parametrize("kind", [3, 2])
def test_put_trip_checksums_of_short_blocks_need_no_padded_copy(kind):
    """the CPU codec's put trip (gec_encode_hash_batch) used to zero-extend the one short data shard of
    every block into a fresh buffer, serially, before the pool started.  Checksum v3 now comes out of the encode's own pass (leaf
    sums of the bytes that exist; S is bound by the root), v2 extends a short shard inside the pool task that hashes it.  The
    results are those of the zero-extended shards: oracle/mlh64.py (v3) / the product's one-shard form (v2) on block lengths
    1, k*S - 1, k*S, 1 MiB, and lengths that end inside a leaf, on a leaf boundary and inside the first word of a chunk."""
    from oracle import mlh64

    k, m = 10, 4
    rs = g.ReedSolomon(k, m, backend="cpu", shardsum=kind)
    S = g.shard_len(k, 1 << 20)
    lens = [1, k * S - 1, k * S, 1 << 20, 3 * S + 4096, 3 * S + 4097, 7 * S + 16384 + 1, 5 * S, 0]
    blocks = [bytes(O.splitmix64_bytes(900 + i, n)) for i, n in enumerate(lens)]
    par, sums = rs.encode_hash_blocks(blocks, S)
    co = O.COracle()
    data = np.zeros((len(lens), k, S), dtype=np.uint8)
    for b, blk in enumerate(blocks):
        data[b].reshape(-1)[: len(blk)] = np.frombuffer(blk, dtype=np.uint8)
    want_par = co.encode_batch(k, m, data, co.SCALAR)
    assert np.array_equal(np.stack(par), want_par)
    for b in range(len(lens)):
        for j in range(k + m):
            payload = (data[b, j] if j < k else want_par[b, j - k]).tobytes()
            want = mlh64.shardsum3(payload) if kind == 3 else g.shardsum(payload, 2)
            assert sums[b, j].tobytes() == want, (lens[b], j)


@pytest.mark.parametrize("k,m,S,nb", [(10, 4, 4160, 40), (3, 1, 64, 9), (10, 12, 1088, 12)])
def test_reconstruct_dev_ex_on_host_memory(coracle, k, m, S, nb):
    """gec_reconstruct_batch_dev_ex over a CPU codec: the strided call on HOST memory, an erasure pattern per block."""
    import torch

    rs = g.ReedSolomon(k, m, backend="cpu")
    rng = np.random.default_rng(k + S)
    data = rng.integers(0, 256, (nb, k, S), dtype=np.uint8)
    full = np.concatenate([data, coracle.encode_batch(k, m, data, coracle.AVX2)], axis=1)
    pres = np.ones((nb, k + m), dtype=np.uint8)
    for b in range(1, nb):
        pres[b, rng.choice(k + m, size=int(rng.integers(0, m + 1)), replace=False)] = 0
    for data_only in (False, True):
        broken = full.copy()
        broken[pres == 0] = 0xEE
        st = torch.from_numpy(broken.copy())
        rs.reconstruct_dev_ex(st, pres, data_only=data_only)
        for b in range(nb):
            assert np.array_equal(st[b].numpy(), O.reconstruct(k, m, broken[b], pres[b], data_only=data_only)), (b, data_only)
    bad = pres.copy()
    bad[2, :m + 1] = 0
    with pytest.raises(g.GecError) as ei:
        rs.reconstruct_dev_ex(torch.from_numpy(full.copy()), bad)
    assert ei.value.code == _lib.GEC_E_TOO_FEW_PRESENT
