"""Shard checksum v3 (GEC_SHARDSUM_MLH64) on the GPU, against oracle/mlh64.py -- an independent restatement in Python
integers that shares no code or tables with garage_amd/csrc/mlh64*.hpp.  Bit-exact (integer arithmetic).

Covers the three producers of leaf sums (the streaming kernel, the RS kernel's SUM form, the pointer-table kernel's SUM form)
and the root kernel, on ragged lengths around every leaf boundary, and the corruption guarantees the format states."""
import ctypes

import numpy as np
import pytest

import garage_amd as g
from garage_amd._lib import check, lib
from oracle import mlh64 as M
from oracle import rs_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_mod():
    import torch

    assert torch.cuda.is_available()
    return torch


RAGGED = [0, 1, 3, 4, 5, 15, 16, 17, 31, 63, 64, 65, 1023, 1024, 1025, 4079, 4080, 4081, 4092, 4093, 4095, 4096, 4097, 4111, 4112,
          8191, 8192, 8193, 12287, 12288, 12289, 65535, 65536, 65537, 104896, 104897, 209728, 300001]


def test_host_buffers_of_ragged_lengths(torch_mod):
    """gec_shardsum_batch over pageable buffers: every length around a word, column, wave and leaf boundary"""
    rs = g.ReedSolomon(10, 4)
    assert rs.shardsum_kind == 3
    rng = np.random.default_rng(3)
    msgs = [rng.integers(0, 256, n, dtype=np.uint8).tobytes() for n in RAGGED]
    assert rs.shardsum_batch(msgs) == [M.shardsum3(x) for x in msgs]
    # the pure-Python definition agrees with the numpy form of the oracle on the small ones
    assert all(M.shardsum3_slow(x) == M.shardsum3(x) for x in msgs if len(x) < 20000)
    # ... and with the library's host implementation (what a node runs over a shard it serves)
    for x in msgs:
        out = ctypes.create_string_buffer(32)
        check(lib.gec_shardsum_host(3, x, len(x), out), "gec_shardsum_host")
        assert out.raw == M.shardsum3(x), len(x)


def test_shards_too_long_for_the_four_lane_roots(torch_mod):
    """The root kernel gives a shard four lanes while its message (16 + 8 bytes per leaf) fits the workgroup's LDS -- shards up to
    1.49 MiB -- and one lane beyond; a call's longest shard decides for the call.  Both sides of the limit, and the limit itself."""
    rs = g.ReedSolomon(10, 4)
    rng = np.random.default_rng(31)
    edge = (24 * 16 - 2) * 4096                       # 382 leaves: the last length with four lanes
    for lens in ([0, 5000, edge - 1, edge], [0, 5000, edge + 1], [17, 1 << 21, (1 << 21) + 4097]):
        msgs = [rng.integers(0, 256, n, dtype=np.uint8).tobytes() for n in lens]
        assert rs.shardsum_batch(msgs) == [M.shardsum3(x) for x in msgs], lens


def test_pinned_buffers_are_read_in_place(torch_mod):
    from garage_amd.codec import host_alloc, host_free

    rs = g.ReedSolomon(10, 4)
    rng = np.random.default_rng(4)
    lens = [0, 7, 4096, 5000, 104896, 104896, 33000, 1 << 20]
    bufs = [host_alloc(max(n, 16)) for n in lens]
    try:
        for b, n in zip(bufs, lens):
            b[:n] = rng.integers(0, 256, n, dtype=np.uint8)
        ptrs = (ctypes.c_void_p * len(lens))(*[b.ctypes.data for b in bufs])
        clens = (ctypes.c_size_t * len(lens))(*lens)
        out = np.zeros((len(lens), 32), dtype=np.uint8)
        check(lib.gec_shardsum_batch(rs._h, len(lens), ptrs, clens, out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))), "pinned")
        assert [out[i].tobytes() for i in range(len(lens))] == [M.shardsum3(b[:n].tobytes()) for b, n in zip(bufs, lens)]
    finally:
        for b in bufs:
            host_free(b)


@pytest.mark.parametrize("S", [64, 1024, 4096, 4160, 8192, 104896, 209728])
def test_device_resident_shards(torch_mod, S):
    torch = torch_mod
    rs = g.ReedSolomon(10, 4)
    rng = np.random.default_rng(S)
    data = rng.integers(0, 256, (37, S), dtype=np.uint8)
    got = rs.shardsum_dev(torch.from_numpy(data).to("cuda:0")).cpu().numpy()
    for i in range(37):
        assert got[i].tobytes() == M.shardsum3(data[i].tobytes()), (S, i)


SHAPES = [(10, 4, 104896, 9), (10, 4, 64, 5), (10, 4, 4096, 3), (10, 4, 4160, 4), (10, 4, 12352, 2), (3, 1, 21888, 7), (20, 8, 209728, 3),
          (20, 8, 8256, 5), (4, 2, 1024, 6), (12, 4, 6208, 3), (16, 4, 5056, 2), (17, 3, 4160, 2), (6, 6, 4352, 3), (10, 8, 8192, 2), (5, 12, 4160, 2),
          (30, 6, 4224, 2), (1, 1, 8256, 3)]


@pytest.mark.parametrize("k,m,S,nb", SHAPES, ids=[f"rs{k}_{m}-S{S}" for k, m, S, nb in SHAPES])
def test_encode_leaves_the_checksums_of_every_shard(torch_mod, k, m, S, nb):
    """gec_encode_hash_batch_dev: ONE pass -- the encode kernel's SUM form accumulates the leaf sums of the k shards it reads and
    the m rows it writes.  Parity vs the C oracle, checksums vs the Python oracle; every table width and batch geometry."""
    torch = torch_mod
    rs = g.ReedSolomon(k, m)
    data = O.splitmix64_bytes(1234 + k * 1000 + S, nb * k * S).reshape(nb, k, S)
    data[0] = 0
    if nb > 1:
        data[1] = 0xFF
    st = torch.zeros((nb, k + m, S), dtype=torch.uint8, device="cuda:0")
    st[:, :k] = torch.from_numpy(data).to("cuda:0")
    sums = rs.encode_hash_dev(st)
    torch.cuda.synchronize()
    full = st.cpu().numpy()
    co = O.COracle()
    assert np.array_equal(full[:, k:], co.encode_batch(k, m, data, co.AVX2 if co.has_avx2() else co.SCALAR))
    assert np.array_equal(full[:, :k], data)
    got = sums.cpu().numpy()
    for b in range(nb):
        for j in range(k + m):
            assert got[b, j].tobytes() == M.shardsum3(full[b, j].tobytes()), (b, j)


def test_both_kinds_side_by_side(torch_mod):
    """a codec produces ONE kind; its sibling the other; both over the same stripes"""
    torch = torch_mod
    rs3 = g.ReedSolomon(10, 4)
    rs2 = rs3.with_shardsum(2)
    assert (rs3.shardsum_kind, rs2.shardsum_kind) == (3, 2) and rs3.background().shardsum_kind == 3 and rs2.background().shardsum_kind == 2
    S, nb = 8256, 3
    st = torch.randint(0, 256, (nb, 14, S), dtype=torch.uint8, device="cuda:0")
    s3 = rs3.encode_hash_dev(st).cpu().numpy()
    s2 = rs2.encode_hash_dev(st).cpu().numpy()
    full = st.cpu().numpy()
    for b in range(nb):
        for j in range(14):
            assert s3[b, j].tobytes() == M.shardsum3(full[b, j].tobytes())
            assert s2[b, j].tobytes() == g.shardsum(full[b, j].tobytes(), 2)


def test_every_flip_is_detected(torch_mod):
    """10^5 random single-byte flips of one shard, all through the device kernel: every checksum differs from the clean one.
    (Guaranteed by the arithmetic: K[i] * delta != 0 mod 2^64 for a change confined to one 32-bit word.)"""
    torch = torch_mod
    rs = g.ReedSolomon(10, 4)
    S = 12352
    rng = np.random.default_rng(99)
    base = rng.integers(0, 256, S, dtype=np.uint8)
    clean = M.shardsum3(base.tobytes())
    total, batch = 100_000, 10_000
    for it in range(total // batch):
        pos = rng.integers(0, S, batch)
        xor = rng.integers(1, 256, batch, dtype=np.uint8)
        t = torch.from_numpy(np.broadcast_to(base, (batch, S)).copy()).to("cuda:0")
        idx = torch.arange(batch, device="cuda:0")
        t[idx, torch.from_numpy(pos).to("cuda:0")] ^= torch.from_numpy(xor).to("cuda:0")
        got = rs.shardsum_dev(t).cpu().numpy()
        clean_arr = np.frombuffer(clean, dtype=np.uint8)
        assert not (got == clean_arr).all(axis=1).any(), it
        # and the device's word for a few of them equals the oracle's
        for q in (0, batch // 2, batch - 1):
            mod = base.copy()
            mod[pos[q]] ^= xor[q]
            assert got[q].tobytes() == M.shardsum3(mod.tobytes())


def test_truncation_extension_and_leaf_swaps_are_detected():
    rng = np.random.default_rng(5)
    d = rng.integers(0, 256, 3 * 4096, dtype=np.uint8)
    s = M.shardsum3(d.tobytes())
    assert M.shardsum3(d[:-1].tobytes()) != s and M.shardsum3(d.tobytes() + b"\0") != s       # zero bytes change the length
    swapped = np.concatenate([d[4096:8192], d[:4096], d[8192:]])
    assert M.shardsum3(swapped.tobytes()) != s                                                 # same leaf sums, other order
    z = np.zeros(8192, dtype=np.uint8)
    assert M.shardsum3(z.tobytes()) != M.shardsum3(z[:4096].tobytes())


def test_a_healthy_get_launches_no_kernel(tmp_path):
    """Header version 3: the requester's cores check the shards (gec_shardsum_host's arithmetic on the manager's pool), so a healthy
    rpc_get_blocks / rpc_get_block / streaming get over a HIP codec issues NO device work at all -- asserted under rocprofv3: the
    traced process only reads (the store was written by another process), and its kernel trace holds no kernel of the library.
    A degraded get of the same blocks then launches exactly the decode."""
    import csv
    import os
    import subprocess
    import sys

    rocprof = "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        pytest.skip("rocprofv3 not installed")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    store = tmp_path / "store"
    common = r'''
import sys, os
sys.path.insert(0, %r)
import garage_amd as g
from garage_amd import block_native as bn
from tests.patterns import pattern_block
dirs = [os.path.join(%r, "node%%d" %% i) for i in range(16)]
blocks = [pattern_block(1 << 20, 900 + i) for i in range(24)]
hashes = [bn.blake2sum(b) for b in blocks]
codec = g.ReedSolomon(10, 4)
mgr = bn.NativeBlockManager(codec, 16, dirs)
assert mgr.shard_version == 3
''' % (root, str(store))
    r = subprocess.run([sys.executable, "-c", common + "mgr.rpc_put_blocks(list(zip(hashes, blocks)))\nmgr.close()\n"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]

    def traced(body, tag):
        out = tmp_path / tag
        rr = subprocess.run([rocprof, "--kernel-trace", "--output-format", "csv", "-d", str(out), "-o", "t", "--", sys.executable, "-c", common + body],
                            cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=900)
        assert rr.returncode == 0 and "GETS OK" in rr.stdout, rr.stdout[-1500:] + rr.stderr[-1500:]
        names = []
        for dirpath, _, files in os.walk(out):
            for f in files:
                if f.endswith("kernel_trace.csv"):
                    with open(os.path.join(dirpath, f)) as fh:
                        names += [row["Kernel_Name"] for row in csv.DictReader(fh)]
        return [x for x in names if "gec::" in x]

    healthy = r'''
got = mgr.rpc_get_blocks(hashes, 1 << 20)
assert [bytes(x) for x in got] == blocks
assert mgr.rpc_get_block(hashes[3]) == blocks[3]
assert b"".join(mgr.rpc_get_block_streaming(hashes[5])) == blocks[5]
print("GETS OK")
'''
    assert traced(healthy, "healthy") == []
    degraded = r'''
for h in hashes[:6]:
    who = mgr.storage_nodes_of(h)
    mgr.node_delete_shard(who[2], h, 2)
mgr.set_verify_block_hash("off")
got = mgr.rpc_get_blocks(hashes, 1 << 20)
assert [bytes(x) for x in got] == blocks
print("GETS OK")
'''
    names = traced(degraded, "degraded")
    assert len(names) == 1 and "gf_apply_ptrs" in names[0], names      # the decode of the six blocks that miss a data shard, nothing else
