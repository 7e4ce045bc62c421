"""GPU blake2sum against the CPU oracle for this
row: Python's hashlib.blake2b (RFC 7693 reference implementation in CPython),
truncated to 32 bytes the way Garage's blake2sum does (src/util/data.rs:130-138).
Bit-exact."""
import hashlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import garage_amd as g  # noqa: E402
from oracle import rs_oracle as O  # noqa: E402


def ref(b: bytes) -> bytes:
    return hashlib.blake2b(b, digest_size=64).digest()[:32]


@pytest.fixture(scope="module")
def rs():
    return g.ReedSolomon(10, 4, shardsum=2)   # this file is about the BLAKE2b kernels: checksum kind 2 (kind 3: tests/test_gpu_shardsum3.py)


def test_rfc7693_abc(rs):
    # RFC 7693 Appendix A: BLAKE2b-512("abc")
    want = bytes.fromhex("ba80a53f981c4d0d6a2797b69f12f6e94c212f14685ac4b74b12bb6fdbffa2d1"
                         "7d87c5392aab792dc252d5de4533cc9518d38aa8dbf1925ab92386edd4009923")
    assert ref(b"abc") == want[:32]
    assert rs.blake2sum_batch([b"abc"]) == [want[:32]]


def test_ragged_lengths_host_api(rs):
    lens = [0, 1, 7, 8, 9, 15, 16, 17, 63, 64, 65, 127, 128, 129, 255, 256, 257, 1000, 3072, 65535, 65536, 104896,
            1 << 20, (1 << 20) + 3]
    msgs = [bytes(O.splitmix64_bytes(900 + i, n)) for i, n in enumerate(lens)]
    got = rs.blake2sum_batch(msgs)
    assert got == [ref(m) for m in msgs]
    # many small messages (more lanes than one wave, several workgroups)
    small = [bytes([i % 251]) * (i % 300) for i in range(1000)]
    assert rs.blake2sum_batch(small) == [ref(m) for m in small]


def test_uniform_device_api_shard_shape(rs):
    # config-2 shard shape: 14 shards of 104896 bytes per stripe
    n, S = 14 * 16, 104896
    data = O.splitmix64_bytes(4242, n * S).reshape(n, S)
    out = rs.blake2sum_dev(torch.from_numpy(data).to("cuda:0"))
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    for i in range(n):
        assert got[i].tobytes() == ref(data[i].tobytes()), i
    with pytest.raises(g.GecError):
        rs.blake2sum_dev(torch.zeros((4, 100), dtype=torch.uint8, device="cuda:0"))   # rows not 16-byte aligned


@pytest.mark.parametrize("k,m,S,nb", [(10, 4, 104896, 24), (3, 1, 64, 70), (20, 8, 4160, 5), (10, 4, 128, 1), (10, 4, 192, 3)])
def test_encode_hash_dev_fork_join(coracle, k, m, S, nb):
    """gec_encode_hash_batch_dev: the checksums of the data shards are computed on a second stream beside
    the RS kernel, the parity checksums behind it; parity vs the oracle, every checksum vs the hashlib restatement
    of the shard checksum (BLAKE2b tree mode, garage_amd.codec.shardsum) --
    repeated back to back so that a missing stream dependency would show as a stale checksum."""
    rs = g.ReedSolomon(k, m, shardsum=2)
    for rep in range(3):
        data = O.splitmix64_bytes(7000 + rep, nb * k * S).reshape(nb, k, S)
        st = torch.zeros((nb, k + m, S), dtype=torch.uint8, device="cuda:0")
        st[:, :k] = torch.from_numpy(data).to("cuda:0")
        sums = rs.encode_hash_dev(st)
        torch.cuda.synchronize()
        full = st.cpu().numpy()
        assert np.array_equal(full[:, k:], coracle.encode_batch(k, m, data, coracle.AVX2, threads=4))
        got = sums.cpu().numpy()
        for b in range(nb):
            for j in range(k + m):
                assert got[b, j].tobytes() == g.shardsum(full[b, j].tobytes(), 2), (rep, b, j)


def test_blake2_quad_and_lane_kernels_agree_on_tails(rs):
    """Both kernels on lengths around every block / quarter / word boundary (the quad kernel's lanes
    each fetch a 32-byte quarter: partial quarters, empty quarters, the empty message)."""
    import os
    import subprocess
    import sys

    code = """
import hashlib, sys
sys.path.insert(0, %r)
import garage_amd as g
g.set_kernel_variant(int(sys.argv[1]))   # 2 / 3: the BLAKE2b kernels with one lane / four lanes per message (include/garage_ec.h)
rs = g.ReedSolomon(10, 4, shardsum=2)
lens = list(range(0, 300)) + [383, 384, 385, 4095, 4096, 4097, 104896]
msgs = [bytes((i * 7 + j) & 255 for j in range(n)) for i, n in enumerate(lens)]
want = [hashlib.blake2b(x, digest_size=64).digest()[:32] for x in msgs]
assert rs.blake2sum_batch(msgs) == want
# the shard checksum's leaves and roots have the same two forms (four lanes per leaf / root below 40000 leaves)
tl = [0, 1, 63, 64, 65, 127, 128, 129, 4095, 4096, 4097, 8191, 8192, 8193, 12288, 12289, 104896, 209728, (1 << 20) + 5]
tm = [bytes((i * 11 + j * 3) & 255 for j in range(n)) for i, n in enumerate(tl)]
assert rs.shardsum_batch(tm) == [g.shardsum(x, 2) for x in tm]
print("ok")
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for kern, variant in (("quad", "3"), ("lane", "2")):
        r = subprocess.run([sys.executable, "-c", code, variant], capture_output=True, text=True)
        assert r.returncode == 0 and "ok" in r.stdout, (kern, r.stdout, r.stderr[-2000:])


def test_encode_hash_batch_sums_every_shard(coracle, rs):
    k, m = 10, 4
    lens = [1 << 20, 999_999, 65536, 1, 0, 500_000]
    blocks = [bytes(O.splitmix64_bytes(70 + i, n)) for i, n in enumerate(lens)]
    S = g.shard_len(k, max(lens))
    pars, sums = rs.encode_hash_blocks(blocks, S)
    for b, blk in enumerate(blocks):
        shards = O.split_block(k, blk, S)
        want_par = coracle.encode_batch(k, m, shards[None], coracle.AVX2)[0]
        assert np.array_equal(pars[b], want_par)
        for j in range(k + m):
            payload = shards[j] if j < k else want_par[j - k]
            assert sums[b, j].tobytes() == g.shardsum(payload.tobytes(), 2), (b, j)


@pytest.mark.parametrize("k,m,L,nb", [(10, 4, 1 << 20, 200), (3, 1, 300_000, 7), (10, 12, 70_000, 5)],
                         ids=["rs10_4_three_chunks", "rs3_1", "rs10_12_two_row_groups"])
def test_encode_hash_batch_zero_copy_pinned(coracle, k, m, L, nb):
    """gec_encode_hash_batch on pinned caller memory: one kernel reads the data shards over the link, writes parity
    back and lays everything down in HBM, where the shard checksums are computed chunk by chunk -- against the
    oracle's parity and hashlib's tree-mode checksums, ragged lengths, several chunks on both streams."""
    import ctypes

    from garage_amd import _lib
    from garage_amd.codec import host_alloc, host_free

    lib = _lib.lib
    rs_ = g.ReedSolomon(k, m, shardsum=2)
    n = k + m
    S = g.shard_len(k, L)
    rng = np.random.default_rng(nb)
    lens = [L if b % 5 else int(rng.integers(0, L + 1)) for b in range(nb)]
    lens[0] = L
    padded = np.zeros((nb, k * S), dtype=np.uint8)
    arena = host_alloc(nb * k * S)
    par = host_alloc(nb * m * S)
    arena[:] = 0x77
    for b in range(nb):
        padded[b, :lens[b]] = rng.integers(0, 256, lens[b], dtype=np.uint8)
        arena[b * k * S: b * k * S + lens[b]] = padded[b, :lens[b]]
    want = coracle.encode_batch(k, m, padded.reshape(nb, k, S), coracle.AVX2, threads=8)
    ptrs = (ctypes.c_void_p * nb)(*[arena.ctypes.data + b * k * S for b in range(nb)])
    optrs = (ctypes.c_void_p * nb)(*[par.ctypes.data + b * m * S for b in range(nb)])
    clens = (ctypes.c_size_t * nb)(*lens)
    sums = np.zeros((nb, n, 32), dtype=np.uint8)
    _lib.check(lib.gec_encode_hash_batch(rs_._h, nb, ptrs, clens, S, optrs, sums.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))),
               "gec_encode_hash_batch")
    assert np.array_equal(par.reshape(nb, m, S), want)
    shards = padded.reshape(nb, k, S)
    for b in list(range(0, nb, max(1, nb // 16))) + [nb - 1]:
        for j in range(n):
            payload = shards[b, j] if j < k else want[b, j - k]
            assert sums[b, j].tobytes() == g.shardsum(payload.tobytes(), 2), (b, j)
    host_free(arena)
    host_free(par)


@pytest.mark.parametrize("kernel", ["lane", "quad"])
def test_both_kernels_forced(kernel):
    """The host picks the one-lane or the four-lane kernel by batch size; force each
    (gec_set_kernel_variant 2 / 3, a process of its own) and check ragged + uniform inputs."""
    import os
    import subprocess
    import sys

    code = r'''
import hashlib, sys
import numpy as np, torch
import garage_amd as g
g.set_kernel_variant(int(sys.argv[1]))
from oracle import rs_oracle as O
rs = g.ReedSolomon(10, 4, shardsum=2)
ref = lambda b: hashlib.blake2b(b, digest_size=64).digest()[:32]
lens = [0, 1, 31, 32, 33, 64, 96, 127, 128, 129, 160, 255, 256, 257, 1000, 4097, 104896, 300001]
msgs = [bytes(O.splitmix64_bytes(50 + i, n)) for i, n in enumerate(lens)] + [bytes([i]) * (i * 7 % 400) for i in range(150)]
assert rs.blake2sum_batch(msgs) == [ref(m) for m in msgs], "ragged"
data = O.splitmix64_bytes(9, 70 * 4160).reshape(70, 4160)
out = rs.blake2sum_dev(torch.from_numpy(data).to("cuda:0")).cpu().numpy()
assert all(out[i].tobytes() == ref(data[i].tobytes()) for i in range(70)), "uniform"
print("OK")
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code, {"lane": "2", "quad": "3"}[kernel]], cwd=root, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout + r.stderr


def test_host_blake2sum_zero_copy_and_single_launch_paths(rs):
    """gec_blake2sum_batch has three ways in: (a) every message in pinned memory -> one launch reading host memory
    directly; (b) long messages in pageable memory -> everything staged into one device buffer, one launch; (c) many
    short pageable messages -> chunked pipeline.  Same digests from all of them."""
    import ctypes

    from garage_amd._lib import check, lib
    from garage_amd.codec import host_alloc, host_free

    lens = [1 << 20, (1 << 20) + 13, 3 << 20, 40 << 20, 0, 5, 65536, 999_999]
    rng = np.random.default_rng(77)
    msgs = [rng.integers(0, 256, n, dtype=np.uint8) for n in lens]
    want = [ref(m.tobytes()) for m in msgs]
    assert rs.blake2sum_batch([m.tobytes() for m in msgs]) == want            # (b): longest >= 256 KiB, pageable
    pinned = [host_alloc(max(n, 16)) for n in lens]                           # (a)
    for p_, m in zip(pinned, msgs):
        p_[:len(m)] = m
    n = len(lens)
    ptrs = (ctypes.c_void_p * n)(*[p_.ctypes.data for p_ in pinned])
    clens = (ctypes.c_size_t * n)(*lens)
    out = np.zeros((n, 32), dtype=np.uint8)
    check(lib.gec_blake2sum_batch(rs._h, n, ptrs, clens, out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))), "pinned blake2")
    assert [out[i].tobytes() for i in range(n)] == want
    # a slice in the middle of a pinned buffer, 16-byte aligned, and one that is not (falls back to staging)
    for off in (4096, 4099):
        p1 = (ctypes.c_void_p * 1)(pinned[3].ctypes.data + off)
        l1 = (ctypes.c_size_t * 1)(1 << 20)
        check(lib.gec_blake2sum_batch(rs._h, 1, p1, l1, out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))), "slice")
        assert out[0].tobytes() == ref(msgs[3][off:off + (1 << 20)].tobytes())
    for p_ in pinned:
        host_free(p_)



def test_shardsum_tree_mode_against_hashlib(rs):
    """The shard checksum (BLAKE2b tree mode: 4 KiB leaves, unlimited fanout, depth 2, inner 64) on the device vs
    hashlib with the same tree parameters: lengths around the leaf boundary, one / many leaves, ragged batches
    through the host API, uniform shard-shaped batches through the device API, pinned zero-copy input."""
    import ctypes

    from garage_amd._lib import check, lib
    from garage_amd.codec import host_alloc, host_free

    lens = [0, 1, 63, 64, 127, 128, 129, 4095, 4096, 4097, 8191, 8192, 8193, 12288, 104896, 209728, 1 << 20, (1 << 20) + 5]
    msgs = [bytes(O.splitmix64_bytes(300 + i, n)) for i, n in enumerate(lens)]
    assert rs.shardsum_batch(msgs) == [g.shardsum(x, 2) for x in msgs]
    assert rs.shardsum_batch([b"abc"])[0] != ref(b"abc"), "the shard checksum is not the plain hash"
    for S in (64, 4096, 4160, 104896):
        n = 97
        data = O.splitmix64_bytes(77 + S, n * S).reshape(n, S)
        out = rs.shardsum_dev(torch.from_numpy(data).to("cuda:0"))
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        for i in range(n):
            assert got[i].tobytes() == g.shardsum(data[i].tobytes(), 2), (S, i)
    pinned = [host_alloc(max(len(x), 16)) for x in msgs]
    for p_, x in zip(pinned, msgs):
        p_[:len(x)] = np.frombuffer(x, dtype=np.uint8)
    n = len(msgs)
    ptrs = (ctypes.c_void_p * n)(*[p_.ctypes.data for p_ in pinned])
    clens = (ctypes.c_size_t * n)(*lens)
    out = np.zeros((n, 32), dtype=np.uint8)
    check(lib.gec_shardsum_batch(rs._h, n, ptrs, clens, out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))), "pinned shardsum")
    assert [out[i].tobytes() for i in range(n)] == [g.shardsum(x, 2) for x in msgs]
    for p_ in pinned:
        host_free(p_)
