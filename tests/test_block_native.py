"""libgarage_block.so (C++ BlockManager mirror, include/garage_block.h).
Symbols, blake2sum against hashlib, argument errors; then the put / get / failure / resync / scrub scenarios twice:
over libgarage_ec's CPU backend (backend "cpu": runs anywhere -- BASELINE config 1, "CPU path via BlockManager") and
over the HIP backend (backend "hip": marked gpu)."""
import ctypes
import hashlib
import os
import re

import numpy as np
import pytest

import garage_amd as g
from garage_amd import block_native as bn
from oracle import rs_oracle as O
from tests.patterns import pattern_block

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_symbols_match_header():
    src = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "garage_block.h")).read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(gbm_[a-z0-9_]+)\s*\(", src)))
    assert names == sorted(bn.SYMBOLS)
    lib = ctypes.CDLL(bn.LIB_PATH)
    for n in names:
        assert getattr(lib, n) is not None


@pytest.mark.parametrize("n", [0, 1, 64, 127, 128, 129, 256, 3072, 99999, 1 << 20])
def test_blake2sum_is_blake2b512_truncated(n):
    d = bytes(pattern_block(n, salt=n)) if n else b""
    assert bn.blake2sum(d) == hashlib.blake2b(d, digest_size=64).digest()[:32]
    assert bn.blake2sum(d) != hashlib.blake2b(d, digest_size=32).digest(), "NOT blake2b-256 (src/util/data.rs:130-138)"
    blocks = [bytes((i * 5 + j) & 255 for j in range(n)) for i, n in enumerate([0, 1, 127, 128, 129, 4096, 70001, 3] * 2 + [9])]
    assert bn.blake2sum_batch(blocks) == [hashlib.blake2b(b, digest_size=64).digest()[:32] for b in blocks]


@pytest.mark.parametrize("n", [0, 1, 64, 4095, 4096, 4097, 8192, 104896, 209728, (1 << 20) + 3])
def test_shard_checksums_by_header_version(n):
    """The shard checksums restated several times -- C++ (libgarage_block: gbm_shardsum_v), Python (garage_amd.codec.shardsum:
    hashlib with BLAKE2's tree parameters for version 2, numpy for version 3), the independent oracle (oracle/mlh64.py) and
    (GPU tests) the device kernels -- must agree; neither is the plain hash, which is what version 1 carried."""
    from oracle import mlh64

    d = bytes(pattern_block(n, salt=n + 1)) if n else b""
    assert bn.shardsum(d, 2) == g.shardsum(d, 2)
    assert bn.shardsum(d, 3) == g.shardsum(d, 3) == mlh64.shardsum3(d) == bn.shardsum(d)
    assert bn.shardsum(d, 1) == bn.blake2sum(d)
    assert len({bn.shardsum(d, 1), bn.shardsum(d, 2), bn.shardsum(d, 3)}) == 3


def test_create_rejects_null_codec():
    h = ctypes.c_void_p()
    assert bn.lib.gbm_create(None, 14, None, 0, ctypes.byref(h)) == bn.GBM_E_INVALID_ARG


# ----------------------------------------------------------------- scenarios, on both backends
@pytest.fixture(params=["cpu", pytest.param("hip", marks=pytest.mark.gpu)])
def backend(request):
    return request.param


@pytest.fixture(params=[(3, 1), (10, 4)], ids=["rs3_1", "rs10_4"])
def codec(request, backend):
    return g.ReedSolomon(*request.param, backend=backend)


def _mgr(codec, tmp_path=None, extra=2):
    n = codec.k + codec.m + extra
    dirs = [str(tmp_path / f"node{i}") for i in range(n)] if tmp_path else None
    return bn.NativeBlockManager(codec, n, dirs)


def test_native_put_get_roundtrip(codec, tmp_path):
    mgr = _mgr(codec, tmp_path)
    for size in (3073, 65536, 500_000, 1 << 20):
        data = pattern_block(size, salt=size)
        h = bn.blake2sum(data)
        mgr.rpc_put_block(h, data)
        assert mgr.rpc_get_block(h) == data
    h = bn.blake2sum(pattern_block(65536, salt=65536))
    who = mgr.storage_nodes_of(h)
    assert len(set(who)) == codec.k + codec.m
    for j, node in enumerate(who):
        assert mgr.node_has_shard(node, h, j)
    hx = h.hex()
    p = tmp_path / f"node{who[0]}" / hx[:2] / hx[2:4] / f"{hx}.s0"
    assert p.exists() and p.stat().st_size == 64 + g.shard_len(codec.k, 65536)
    # the on-disk shard format: a 64-byte header whose version names the checksum (3 = MLH64: what a default codec writes)
    from tests.patterns import parse_shard_header

    raw = p.read_bytes()
    hdr = parse_shard_header(raw)
    assert (hdr.version, hdr.k, hdr.m, hdr.idx, hdr.orig_len) == (3, codec.k, codec.m, 0, 65536) and mgr.shard_version == 3
    assert bn.shardsum(raw[64:]) == hdr.checksum == g.shardsum(raw[64:])
    assert mgr.metrics["blocks_put"] == 4 and mgr.metrics["blocks_get"] == 4


def test_native_survives_m_failures_and_quorum(codec):
    mgr = _mgr(codec)
    data = pattern_block(300_000, 7)
    h = bn.blake2sum(data)
    mgr.rpc_put_block(h, data)
    who = mgr.storage_nodes_of(h)
    for j in range(codec.m):                 # lose m nodes, data shards first
        mgr.node_set_down(who[j], True)
    assert mgr.rpc_get_block(h) == data
    assert mgr.metrics["ec_reconstructs"] == 1
    mgr.node_set_down(who[codec.m], True)    # one more: unrecoverable
    with pytest.raises(bn.MissingBlock):
        mgr.rpc_get_block(h)
    for j in range(codec.m + 1):
        mgr.node_set_down(who[j], False)
    # write quorum: k + ceil(m/2)
    quorum = codec.k + (codec.m + 1) // 2
    tolerated = codec.k + codec.m - quorum
    data2 = pattern_block(100_000, 9)
    h2 = bn.blake2sum(data2)
    who2 = mgr.storage_nodes_of(h2)
    for j in range(tolerated):
        mgr.node_set_down(who2[-1 - j], True)
    mgr.rpc_put_block(h2, data2)
    assert mgr.resync_queue_len() == (1 if tolerated else 0)
    assert mgr.rpc_get_block(h2) == data2
    mgr.node_set_down(who2[0], True)
    with pytest.raises(bn.Quorum) as ei:
        mgr.rpc_put_block(h2, data2)
    assert "Could not reach quorum" in str(ei.value)


def test_native_corruption_resync_scrub(codec, tmp_path):
    mgr = _mgr(codec, tmp_path)
    blocks = [pattern_block(200_000, s) for s in range(6)]
    hashes = [bn.blake2sum(b) for b in blocks]
    mgr.rpc_put_blocks(list(zip(hashes, blocks)))       # one coalesced device encode
    for h in hashes:
        mgr.block_incref(h)
    assert mgr.scrub(hashes) == []
    h = hashes[2]
    who = mgr.storage_nodes_of(h)
    mgr.node_corrupt_shard(who[1], h, 1, 1234, 0x55)    # checksum now wrong -> detected on read
    lost = 1
    if codec.m >= 2:
        mgr.node_delete_shard(who[codec.k], h, codec.k)
        lost = 2
    got = mgr.rpc_get_blocks(hashes, 200_000)           # batched get, one decode call
    assert got == blocks
    assert mgr.metrics["corruption_counter"] == 1
    hx = h.hex()
    assert (tmp_path / f"node{who[1]}" / hx[:2] / hx[2:4] / f"{hx}.s1.corrupted").exists()
    assert mgr.resync_all() == lost
    assert mgr.scrub(hashes) == []
    # silent corruption with a re-stamped checksum: only the RS verify finds it
    mgr.node_corrupt_shard(who[codec.k], h, codec.k, 77, 1, fix_checksum=True)
    assert mgr.scrub(hashes) == [h]
    # wrong content under a valid name: "always" -- the default over MLH64 shard checksums (header version 3), as the
    # reference's read path checks every Plain block against its name (block.rs:69-76, manager.rs:592) -- answers CorruptData;
    # "rebuilt" (the default over the cryptographic BLAKE2b-tree checksums, version 2) hashes only what a decode rebuilt
    evil = pattern_block(200_000, 99)
    mgr.rpc_put_block(hashes[0], evil)
    assert mgr.verify_block_hash == ("always" if mgr.shard_version == 3 else "rebuilt")
    mgr.set_verify_block_hash("rebuilt")
    assert mgr.rpc_get_block(hashes[0]) == evil   # nothing was rebuilt: not hashed
    mgr.set_verify_block_hash("always")
    with pytest.raises(bn.CorruptData):
        mgr.rpc_get_block(hashes[0])          # (small request: the block hash is checked on the host pool)
    mgr.set_host_block_hash_max(0)            # ... and on the device, in the decode's trip
    with pytest.raises(bn.CorruptData):
        mgr.rpc_get_block(hashes[0])
    assert mgr.rpc_get_block(hashes[2]) == blocks[2]
    mgr.set_host_block_hash_max(96)
    # rc -> 0: nothing is deleted inside BLOCK_GC_DELAY, every shard after it
    mgr.block_decref(hashes[5])
    assert mgr.resync_all() == 0 and mgr.rpc_get_block(hashes[5]) == blocks[5]
    mgr.clock_advance(bn.GBM_BLOCK_GC_DELAY_MS + 11_000)
    assert mgr.resync_all() >= codec.k + codec.m
    with pytest.raises(bn.MissingBlock):
        mgr.rpc_get_block(hashes[5])


def test_compressed_blocks(tmp_path, backend):
    """compression_level = Some(1) (Garage's default): blocks are zstd frames with the content checksum on before they are
    cut into shards; the raw get returns the frame as stored; a decode of the compressed payload, then zstd."""
    codec = g.ReedSolomon(10, 4, backend=backend)
    dirs = [str(tmp_path / f"node{i}") for i in range(14)]
    native = bn.NativeBlockManager(codec, 14, dirs, compression_level=1)
    a, b = pattern_block(1 << 20, 3), pattern_block(700_001, 4)     # compressible patterns
    ha, hb = bn.blake2sum(a), bn.blake2sum(b)
    native.rpc_put_block(ha, a)
    native.rpc_put_block(hb, b)
    hdr, raw = native.rpc_get_raw_block(ha)
    assert hdr.is_compressed() and len(raw) < len(a) // 4 and bn.zstd_decode(raw) == a
    assert native.rpc_get_block(ha) == a and native.rpc_get_block(hb) == b
    # shards are cut from the COMPRESSED payload: they are much smaller than for a plain block
    who = native.storage_nodes_of(ha)
    hx = ha.hex()
    shard_file = tmp_path / f"node{who[0]}" / hx[:2] / hx[2:4] / f"{hx}.s0"
    assert shard_file.stat().st_size < 64 + g.shard_len(10, len(a)) // 4
    # lose 4 shards incl. data shards -> decode of the compressed payload, then zstd
    for j in (0, 1, 2, 11):
        native.node_delete_shard(who[j], ha, j)
    assert native.rpc_get_block(ha) == a
    # random data: falls back to what zstd produces (still a frame)
    rnd = bytes(np.random.default_rng(1).integers(0, 256, 100_000, dtype=np.uint8))
    hr = bn.blake2sum(rnd)
    native.rpc_put_block(hr, rnd)
    assert native.rpc_get_block(hr) == rnd


def test_prevent_compression_order_tag_raw_and_streaming_gets(backend):
    """The reference's put/get surface (src/block/manager.rs:243-274,344-408): an "encrypted" (SSE-C) block is
    put with prevent_compression=True under compression level 1 (put.rs:576) and every shard header says
    Plain; rpc_get_raw_block returns the DataBlock as stored; the streaming forms deliver it in order."""
    codec = g.ReedSolomon(10, 4, backend=backend)
    mgr = bn.NativeBlockManager(codec, 16, compression_level=1)
    data = pattern_block(1 << 20, 12)                       # very compressible
    h = bn.blake2sum(data)
    who = mgr.storage_nodes_of(h)
    mgr.rpc_put_block(h, data, prevent_compression=True, order_tag=(99, 1))
    S = g.shard_len(10, len(data))
    for j, node in enumerate(who):
        hdr = mgr.node_shard_header(node, h, j)
        assert hdr[:4] == b"GECS" and hdr[7] == j and hdr[8] == 0, "compressed flag must be 0 in every shard header"
        assert int.from_bytes(hdr[12:20], "little") == len(data) and int.from_bytes(hdr[20:24], "little") == S
    hd, raw = mgr.rpc_get_raw_block(h)
    assert not hd.is_compressed() and raw == data
    mgr.rpc_put_block(h, data)                              # the same block without the flag: stored Compressed
    hd, raw = mgr.rpc_get_raw_block(h, order_tag=(99, 2))
    assert hd.is_compressed() and len(raw) < len(data) // 4 and raw[:4] == bytes.fromhex("28b52ffd")
    assert mgr.node_shard_header(who[3], h, 3)[8] == 1
    assert mgr.rpc_get_block(h, order_tag=(99, 3)) == data
    chunks = mgr.rpc_get_block_streaming(h, chunk_bytes=100_000)
    assert b"".join(chunks) == data and max(map(len, chunks)) == 100_000 and len(chunks) == 11
    hd2, rchunks = mgr.rpc_get_block_streaming(h, raw=True)
    assert hd2.is_compressed() and b"".join(rchunks) == raw
    with pytest.raises(bn.MissingBlock):
        mgr.rpc_get_block_streaming(bytes(32))
    # a batch handed over in reverse stream order reaches every node in order
    blocks = [pattern_block(300_000 + 4096 * i, 500 + i) for i in range(10)]
    hashes = [bn.blake2sum(b) for b in blocks]
    mgr.rpc_put_blocks(list(zip(hashes, blocks)), prevent_compression=[i % 2 == 0 for i in range(10)],
                       order_tags=[(7, 10 - i) for i in range(10)])
    assert all(mgr.node_order_violations(node) == 0 for node in range(16))
    assert mgr.rpc_get_blocks(hashes, 400_000) == blocks
    assert [mgr.node_shard_header(mgr.storage_nodes_of(hashes[i])[0], hashes[i], 0)[8] for i in range(10)] == [0, 1] * 5


def test_put_with_node_down_is_repaired_never_deleted(backend):
    """put reaches its quorum with a node down, no incref yet, resync runs: the straggler is
    rebuilt; the block is only deleted after a decref AND BLOCK_GC_DELAY."""
    codec = g.ReedSolomon(10, 4, backend=backend)
    mgr = bn.NativeBlockManager(codec, 16)
    data = pattern_block(400_000, 41)
    h = bn.blake2sum(data)
    who = mgr.storage_nodes_of(h)
    mgr.node_set_down(who[0], True)
    mgr.rpc_put_block(h, data)
    assert mgr.block_rc(h)[1] == "Deletable"                 # protected for BLOCK_GC_DELAY
    mgr.node_set_down(who[0], False)
    assert mgr.resync_all() == 1 and mgr.node_has_shard(who[0], h, 0)
    assert mgr.rpc_get_block(h) == data
    mgr.block_incref(h)
    mgr.clock_advance(bn.GBM_BLOCK_GC_DELAY_MS + 11_000)
    assert mgr.resync_all() == 0 and mgr.rpc_get_block(h) == data
    mgr.block_decref(h)
    assert mgr.resync_all() == 0 and mgr.rpc_get_block(h) == data      # inside the GC delay
    mgr.block_incref(h)                                                  # re-upload of identical content
    mgr.clock_advance(bn.GBM_BLOCK_GC_DELAY_MS + 11_000)
    assert mgr.resync_all() == 0 and mgr.rpc_get_block(h) == data
    mgr.block_decref(h)
    mgr.clock_advance(bn.GBM_BLOCK_GC_DELAY_MS + 11_000)
    assert mgr.resync_all() == 14
    with pytest.raises(bn.MissingBlock):
        mgr.rpc_get_block(h)
    assert mgr.block_rc(h)[1] == "Absent"


def test_resync_queue_batches_rebuilds_by_erasure_pattern(backend):
    """Row f3 as the survey wrote it: 1000 blocks are written while one node is dead; the node comes back
    empty; ONE pass over the time-ordered queue gathers k shards per block and rebuilds every absent shard
    with at most one device call -- and one matrix inversion -- per erasure pattern (<= k+m), with error
    back-off while the node is still away."""
    codec = g.ReedSolomon(10, 4, backend=backend)
    mgr = bn.NativeBlockManager(codec, 17)
    NB, L = 1000, 65536
    rng = np.random.default_rng(5)
    blocks = [rng.integers(0, 256, L, dtype=np.uint8).tobytes() for _ in range(NB)]
    hashes = codec.blake2sum_batch(blocks)
    dead = 4
    mgr.node_set_down(dead, True)
    mgr.rpc_put_blocks(list(zip(hashes, blocks)))
    for h in hashes:
        mgr.block_incref(h)
    affected = [h for h in hashes if dead in mgr.storage_nodes_of(h)]
    patterns = {mgr.storage_nodes_of(h).index(dead) for h in affected}
    assert 400 < len(affected) < NB and len(patterns) == 14
    st = mgr.resync_run(check=False)                          # node still away: every rebuild fails to land
    assert st["rc"] != 0 and st["errors"] == len(affected) and st["rebuilt"] == 0
    assert mgr.resync_errors_len() == len(affected)
    mgr.node_set_down(dead, False)
    assert mgr.resync_run()["taken"] == 0                     # inside the 60 s back-off
    mgr.clock_advance(bn.GBM_RESYNC_RETRY_DELAY_MS + 5)
    _, inv0 = codec.cache_stats()
    st = mgr.resync_run()
    _, inv1 = codec.cache_stats()
    assert st["taken"] == st["ok"] == st["rebuilt"] == len(affected) and st["errors"] == 0
    assert 1 <= st["device_calls"] <= len(patterns)
    assert inv1 - inv0 <= len(patterns)                       # one decode matrix per pattern, LRU-cached
    assert mgr.resync_errors_len() == 0
    assert all(mgr.node_has_shard(dead, h, mgr.storage_nodes_of(h).index(dead)) for h in affected)
    assert mgr.scrub(hashes) == []                            # every stripe RS-consistent on the device
    assert mgr.rpc_get_blocks(hashes[:50], L) == blocks[:50]


def test_layout_change_offloads_shards_to_their_new_owners(backend):
    codec = g.ReedSolomon(10, 4, backend=backend)
    mgr = bn.NativeBlockManager(codec, 20)
    blocks = [pattern_block(200_000, 700 + i) for i in range(40)]
    hashes = [bn.blake2sum(b) for b in blocks]
    mgr.rpc_put_blocks(list(zip(hashes, blocks)))
    for h in hashes:
        mgr.block_incref(h)
    old = [mgr.storage_nodes_of(h) for h in hashes]
    assert mgr.layout_update() == 1
    new = [mgr.storage_nodes_of(h) for h in hashes]
    assert old != new
    assert mgr.rpc_get_blocks(hashes, 200_000) == blocks      # reads fall back to the previous layout version
    for h in hashes:
        mgr.put_to_resync(h)
    st = mgr.resync_run()
    assert st["ok"] == 40 and st["offloaded"] > 0 and st["device_calls"] == 0
    for h, o, n_ in zip(hashes, old, new):
        for j in range(14):
            assert mgr.node_has_shard(n_[j], h, j)
            assert o[j] == n_[j] or not mgr.node_has_shard(o[j], h, j)
    mgr.layout_trim()
    assert mgr.rpc_get_blocks(hashes, 200_000) == blocks


def test_reads_walk_the_layout_versions_oldest_first():
    """block_read_nodes_of asks "the preferred node in all layout versions (older to newer)" (rpc_helper.rs:559-563).  Here the
    order is also what keeps a read safe beside the mover (PutShard at the new owner, THEN DeleteShard at the old one): the old
    holder is asked first.  Seen from outside: right after a layout change, before anything has moved, a read never touches a
    node that only the NEW version names: their request counters do not move (no clock in the assertion)."""

    codec = g.ReedSolomon(3, 1, backend="cpu")
    mgr = bn.NativeBlockManager(codec, 9)
    blocks = [pattern_block(50_000, 4200 + i) for i in range(12)]
    hashes = [bn.blake2sum(b) for b in blocks]
    mgr.rpc_put_blocks(list(zip(hashes, blocks)))
    old = [mgr.storage_nodes_of(h) for h in hashes]
    assert mgr.layout_update() == 1
    new = [mgr.storage_nodes_of(h) for h in hashes]
    picked = [i for i in range(len(blocks)) if set(new[i][:3]) - set(old[i])]   # a data shard's new owner holds nothing of the block
    assert picked
    for i in picked[:4]:
        new_only = set(new[i]) - set(old[i])
        before = {nd: mgr.node_requests(nd) for nd in new_only}
        assert mgr.rpc_get_block(hashes[i]) == blocks[i]
        assert b"".join(mgr.rpc_get_block_streaming(hashes[i])) == blocks[i]
        assert {nd: mgr.node_requests(nd) for nd in new_only} == before, "a node only the new layout version names was asked before the old holders"
    # a shard that HAS moved is found at its new owner once the old one has said no
    h, o, n_ = hashes[picked[0]], old[picked[0]], new[picked[0]]
    mgr.block_incref(h)
    mgr.put_to_resync(h)
    assert mgr.resync_run()["offloaded"] > 0
    assert all(mgr.node_has_shard(n_[j], h, j) and (o[j] == n_[j] or not mgr.node_has_shard(o[j], h, j)) for j in range(4))
    assert mgr.rpc_get_block(h) == blocks[picked[0]]
    mgr.close()


# ----------------------------------------------------------------- round 6: the default mode's big gets, shared between pool and device
@pytest.mark.parametrize("block_len", [40_000, 1 << 20], ids=["40KB", "1MiB"])
def test_a_big_always_get_is_shared_between_the_pool_and_the_device_trip(backend, block_len):
    """GBM_VERIFY_ALWAYS is the default over header version 3, and the hash of every block is what a big get then
    costs: ~11 ms of BLAKE2b chain per MiB on the device however few blocks the trip carries, with the pool idle beside it.
    fetch_blocks' shared form gives the pool as many healthy blocks as it checks, assembles and hashes in that time and the device
    the rest (every block that misses a data shard among them).  Both shares must behave as one get: the right bytes, CorruptData
    for content that does not match its name whichever share the block falls into, a shard that fails its checksum replaced, a
    compressed block untouched by the name check.  (1 MiB blocks: both shares are populated; 40 KB: the pool takes every healthy block.)"""
    codec = g.ReedSolomon(10, 4, backend=backend)
    mgr = bn.NativeBlockManager(codec, 16)
    assert mgr.verify_block_hash == "always"
    nb = 160 if block_len > 100_000 else 220
    mgr.set_host_block_hash_max(8)                      # (so that a batch of this size is not simply hashed by the pool afterwards)
    blocks = [bytes(O.splitmix64_bytes(6000 + i, block_len - (i % 7) * 13)) for i in range(nb)]
    hashes = [bn.blake2sum(b) for b in blocks]
    mgr.rpc_put_blocks(list(zip(hashes, blocks)))
    evil = {3: bytes(O.splitmix64_bytes(1, len(blocks[3]))), 97: bytes(O.splitmix64_bytes(2, len(blocks[97]))), nb - 1: b"x" * 5000}
    for i, content in evil.items():                     # wrong content under a valid name
        mgr.rpc_put_block(hashes[i], content)
    degraded = [5, 97, 120]                             # these miss a data shard: the device's share (97 is evil AND degraded)
    for i in degraded:
        who = mgr.storage_nodes_of(hashes[i])
        mgr.node_delete_shard(who[2], hashes[i], 2)
    who = mgr.storage_nodes_of(hashes[40])
    mgr.node_corrupt_shard(who[6], hashes[40], 6, 1000, 0x10)   # a healthy block of the pool's share with a rotten shard
    before = mgr.metrics["corruption_counter"]
    got = mgr.rpc_get_blocks(hashes, block_len)
    for i in range(nb):
        if i in evil:
            assert got[i] == bn.GBM_E_CORRUPT_DATA, i
        else:
            assert got[i] == blocks[i], i
    assert mgr.metrics["corruption_counter"] == before + 1
    # the same get in the other modes and through the unshared paths gives the same answers where the modes agree
    mgr.set_verify_block_hash("off")
    got_off = mgr.rpc_get_blocks(hashes, block_len)
    assert [got_off[i] for i in range(nb) if i not in evil] == [blocks[i] for i in range(nb) if i not in evil]
    assert got_off[3] == evil[3]
    mgr.set_verify_block_hash("always")
    mgr.set_host_block_hash_max(10_000)                 # everything hashed by the pool after the fetch
    got2 = mgr.rpc_get_blocks(hashes, block_len)
    assert got2 == got
    mgr.close()


# ----------------------------------------------------------------- a7: request_order / block_read_nodes_of on shards
def _reference_request_order(nodes, self_node, our_zone, zone_of, ping_of):
    """RpcHelper::request_order, /root/reference/src/rpc/rpc_helper.rs:621-660, restated: sort by (is another node, is another
    zone, avg ping or 10 s), stable."""
    return sorted(nodes, key=lambda to: (to != self_node, zone_of[to] != our_zone, ping_of.get(to) or 10_000_000))


def _reference_block_read_nodes_of(vernodes, self_node, our_zone, zone_of, ping_of, n):
    """block_read_nodes_of, rpc_helper.rs:570-619, restated on (node, shard) pairs: every layout version's holders in
    request_order, then "the preferred node in all layout versions (older to newer), then the second preferred one in all
    versions", no request twice, the requester itself in front when there are several versions."""
    ordered = []
    for ver, who in vernodes:                       # oldest version first; who[j] = the holder of shard j in that version
        pref = _reference_request_order(who, self_node, our_zone, zone_of, ping_of)   # (holders are distinct nodes: 14 of 24)
        ordered.append([(nd, who.index(nd), ver) for nd in pref])
    if len(ordered) == 1:
        return ordered[0]
    ret = []
    for i in range(n):
        for vn in ordered:
            node, j, ver = vn[i]
            if any((node, j) == (a, b) for a, b, _ in ret):
                continue
            if node == self_node:
                ret.insert(0, (node, j, ver))
            else:
                ret.append((node, j, ver))
    return ret


def test_read_order_is_the_references_request_order_applied_to_shards():
    """SURVEY section 8 row a7.  gbm_block_read_order (the order the gather walks) against a restatement of the reference's
    request_order + block_read_nodes_of on random zone / ping tables, with one and with two active layout versions, for a
    requester that is a storage node and for one that is not.  A manager that is told nothing asks the k data shards first."""
    import random

    codec = g.ReedSolomon(10, 4, backend="cpu")
    nn = 24
    mgr = bn.NativeBlockManager(codec, nn)
    blocks = [pattern_block(30_000, 3300 + i) for i in range(10)]
    hashes = [bn.blake2sum(b) for b in blocks]
    mgr.rpc_put_blocks(list(zip(hashes, blocks)))
    for h in hashes:                                                   # nothing known: shard-index order, data shards first
        who = mgr.storage_nodes_of(h)
        assert mgr.block_read_order(h) == [(who[j], j, 0) for j in range(14)]
    rnd = random.Random(11)
    old_holders = {h: mgr.storage_nodes_of(h) for h in hashes}
    for trial in range(40):
        if trial == 20:
            assert mgr.layout_update() == 1                            # from here on: two active versions
        zone_of = {nd: rnd.randrange(3) for nd in range(nn)}
        ping_of = {nd: rnd.choice([0, 0, 150, 150, 900, 12_000, 48_000]) for nd in range(nn)}   # 0 = unknown (10 s); ties on purpose
        self_node, our_zone = (rnd.randrange(nn), None) if trial % 2 else (-1, rnd.randrange(3))
        if our_zone is None:
            our_zone = zone_of[self_node]
        for nd in range(nn):
            mgr.node_set_zone(nd, zone_of[nd])
            mgr.node_set_ping(nd, ping_of[nd])
        mgr.set_self_node(self_node, our_zone)
        for h in hashes:
            vernodes = [(0, old_holders[h])] if trial < 20 else [(0, old_holders[h]), (1, mgr.storage_nodes_of(h))]
            want = _reference_block_read_nodes_of(vernodes, self_node, our_zone, zone_of, ping_of, 14)
            assert mgr.block_read_order(h) == want, (trial, self_node, our_zone)
    # the reads still work, whatever the order says
    assert mgr.rpc_get_blocks(hashes, 30_000) == blocks
    assert [b"".join(mgr.rpc_get_block_streaming(h)) for h in hashes] == blocks
    mgr.close()


def test_a_far_zone_costs_a_decode_not_its_round_trip(backend):
    """Which k of the k+m holders a read asks decides whether it pays a WAN round trip or a decode (VERDICT r05 missing #3).
    14 holders in three zones, the zone that holds data shards 5..9 is 60 ms away.  Told nothing, a hedged read asks the ten data
    shards' holders and waits for the far zone; told the zones and pings it asks the nine near holders first (four of them parity)
    and ONE far one, and the block comes back from whichever ten arrive first... which still includes one far shard -- so the
    far zone gets only as many holders as the near ones cannot cover, and with a near parity for every far data shard (RS(10,4),
    4 far holders) the read never waits for the far zone at all."""
    import time

    codec = g.ReedSolomon(10, 4, backend=backend)
    mgr = bn.NativeBlockManager(codec, 14)
    data = pattern_block(400_000, 5151)
    h = bn.blake2sum(data)
    mgr.rpc_put_block(h, data)
    who = mgr.storage_nodes_of(h)
    far = [who[j] for j in (6, 7, 8, 9)]                               # four DATA shards live in the far zone
    for nd in range(14):
        mgr.node_set_latency(nd, 60_000 if nd in far else 200)
    mgr.set_read_hedge(500_000)                                        # all requests of a round at once; no hedge timer in play

    def timed_get():
        t0 = time.perf_counter()
        assert mgr.rpc_get_block(h) == data
        return time.perf_counter() - t0

    before = mgr.metrics["ec_reconstructs"]
    assert timed_get() > 0.055                                         # told nothing: the ten data shards, four of them 60 ms away
    assert mgr.metrics["ec_reconstructs"] == before                    # ... and no decode
    for nd in range(14):
        mgr.node_set_zone(nd, 1 if nd in far else 0)
        mgr.node_set_ping(nd, 60_000 if nd in far else 200)
    mgr.set_self_node(who[12], 0)                                      # the requester holds parity shard 12 itself
    order = mgr.block_read_order(h)
    assert order[0] == (who[12], 12, 0) and {j for _, j, _ in order[:10]} == {0, 1, 2, 3, 4, 5, 10, 11, 12, 13}
    t = min(timed_get() for _ in range(3))
    assert t < 0.03, f"{t * 1e3:.1f} ms: the read waited for the far zone"   # near round trips + one decode
    assert mgr.metrics["ec_reconstructs"] > before
    assert b"".join(mgr.rpc_get_block_streaming(h)) == data            # the streaming form takes the general path here, same order
    mgr.close()


def test_a_block_whose_holders_were_all_down_is_asked_for_again_during_a_transition():
    """the retry of a Missing block while a layout change is being followed.  A walk that found NOTHING because
    every holder of the block was DOWN (and the new owners hold nothing yet) has not learnt that the block is absent: during a
    transition it is worth the two later walks (1 ms, then 5 ms apart) -- the old holders may be back, or the mover may have
    finished.  A block that every reachable holder of every version denies is final: no sleep, no second walk.  Counted in node
    requests (no clock in the assertions), and once for real with nodes that come back while the get is waiting."""
    import threading
    import time

    codec = g.ReedSolomon(3, 1, backend="cpu")
    mgr = bn.NativeBlockManager(codec, 12)
    blocks = [pattern_block(40_000, 8100 + i) for i in range(6)]
    hashes = [bn.blake2sum(b) for b in blocks]
    mgr.rpc_put_blocks(list(zip(hashes, blocks)))
    old = [mgr.storage_nodes_of(h) for h in hashes]
    assert mgr.layout_update() == 1
    new = [mgr.storage_nodes_of(h) for h in hashes]
    i = next(i for i in range(len(blocks)) if not set(new[i]) & set(old[i]))     # no node serves this block in both versions
    h, o, n_ = hashes[i], old[i], new[i]

    def asked(nodes):
        return sum(mgr.node_requests(nd) for nd in nodes)

    # (1) a block nobody has ever heard of, every holder up: ONE walk over both versions, final
    ghost = bn.blake2sum(b"never stored")
    gho, ghn = None, mgr.storage_nodes_of(ghost)
    before = asked(range(12))
    with pytest.raises(bn.MissingBlock):
        mgr.rpc_get_block(ghost)
    one_walk = asked(range(12)) - before
    assert 4 <= one_walk <= 8                                                     # each (version, shard) candidate at most once
    # (2) every OLD holder of a stored block down, its new owners empty: three walks (the new owners are asked three times)
    for nd in o:
        mgr.node_set_down(nd, True)
    before_new = asked(n_)
    with pytest.raises(bn.MissingBlock):
        mgr.rpc_get_block(h)
    assert asked(n_) - before_new == 3 * 4, "a block whose holders were all unreachable must be asked for again during a transition"
    # (3) ... and the holders come back while the get waits between its walks: the block is returned
    ok = []
    for attempt in range(20):                                                     # (a thread's start is not on a clock: try until one lands in the window)
        for nd in o:
            mgr.node_set_down(nd, True)

        def revive():
            time.sleep(0.003)
            for nd in o:
                mgr.node_set_down(nd, False)

        before_new = asked(n_)
        t = threading.Thread(target=revive)
        t.start()
        try:
            got = mgr.rpc_get_block(h)
        except bn.MissingBlock:
            got = None
        t.join()
        if got is not None and asked(n_) - before_new >= 4:                       # at least one fruitless walk came first
            ok.append(got)
            break
    assert ok and ok[0] == blocks[i]
    # (4) outside a transition the same outage is final at once (the reference leaves it to the client's retry)
    for nd in o:
        mgr.node_set_down(nd, False)
    for hh in hashes:
        mgr.block_incref(hh)
    mgr.resync_all()
    mgr.layout_trim()
    for nd in n_:
        mgr.node_set_down(nd, True)
    before_all = asked(range(12))
    with pytest.raises(bn.MissingBlock):
        mgr.rpc_get_block(h)
    assert asked(range(12)) - before_all <= 4
    mgr.close()


@pytest.mark.parametrize("seed,ndev", [(1, 1), (6, 1), (7, 2)])
def test_arbitrary_arguments_come_back_with_a_code(seed, ndev):
    """tests/c/bm_abi_fuzz.py, a process of its own: NULL manager / hash / data / outputs, node and shard indices out of range,
    zero capacities, ranges beyond the block, unknown hashes, a NULL sink -- 400 calls around a manager that holds blocks.  (It found
    gbm_node_has_shard / _delete_shard / _corrupt_shard building a Hash from NULL -- std::logic_error, terminate -- and a get
    writing through a NULL output buffer that came with a capacity.)"""
    import subprocess
    import sys

    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "c", "bm_abi_fuzz.py"), str(seed), "cpu", str(ndev)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().splitlines()[-1] == "done 400 unreadable 0", r.stdout[-500:] + r.stderr[-2000:]


@pytest.mark.parametrize("seed", [2, 11])
def test_arbitrary_arguments_to_the_rest_of_the_api(seed):
    """tests/c/bm_abi_fuzz2.py: creation, batched calls with NULL entries, helpers, listings, worker controls."""
    import subprocess
    import sys

    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "c", "bm_abi_fuzz2.py"), str(seed)], capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().splitlines()[-1] == "done", r.stdout[-500:] + r.stderr[-2000:]


def test_batcher_coalesces_concurrent_puts(backend):
    """16 caller threads (think: 16 PutObject requests) each put 6 blocks through the
    batcher; every call blocks until ITS block is stored; the worker coalesces them into
    far fewer device batches; everything reads back; a quorum failure reaches its caller."""
    import threading

    codec = g.ReedSolomon(10, 4, backend=backend)
    mgr = bn.NativeBlockManager(codec, 16)
    bt = bn.Batcher(mgr, max_blocks=32, max_wait_us=2000)
    T, PER = 16, 6
    blocks = [[pattern_block(262144 + 4096 * (t * PER + j), t * 100 + j) for j in range(PER)] for t in range(T)]
    hashes = [[bn.blake2sum(b) for b in row] for row in blocks]
    errors = []
    # (callers that never overlap have nothing to coalesce -- a block that arrives with nothing in flight goes at once -- and
    # sixteen Python threads storing 300 KB blocks in memory nodes may well never overlap: the nodes answer after 1 ms, like disks)
    for nd in range(16):
        mgr.node_set_latency(nd, 1000)

    def worker(t):
        try:
            for j in range(PER):
                bt.put_block(hashes[t][j], blocks[t][j])
                assert mgr.rpc_get_block(hashes[t][j]) == blocks[t][j]   # visible as soon as put returns
        except Exception as e:  # pragma: no cover
            errors.append(repr(e))

    th = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
    [x.start() for x in th]
    [x.join() for x in th]
    assert not errors, errors
    st = bt.stats()
    # (how many batches there are depends on how the callers' gets interleave and on when the linger sees arrivals
    # stop -- a timing property, tools/batcher_bench measures it -- but 16 callers released together must coalesce)
    assert st["blocks"] == T * PER and st["batches"] < T * PER and 2 <= st["max_batch"] <= 32, st
    for nd in range(16):
        mgr.node_set_latency(nd, 0)
    who = mgr.storage_nodes_of(hashes[0][0])
    for j in range(3):
        mgr.node_set_down(who[j], True)
    with pytest.raises(bn.Quorum):
        bt.put_block(hashes[0][0], blocks[0][0])
    bt.close()


@pytest.mark.parametrize("on_disk", [False, True], ids=["memory", "directories"])
def test_scrub_all_locates_and_repairs_silent_corruption(tmp_path, on_disk, backend):
    """ScrubWorker over everything stored (BlockStoreIterator = directory walk / memory stripes), device verify in
    batches.  A parity shard AND (in another block) a data shard rot *before* their checksums are taken: every
    checksum still matches, only the RS verify sees it; leave-one-out decodes say which shard it is; it is set aside
    and the resync that follows rebuilds it.  RepairWorker queues everything known."""
    codec = g.ReedSolomon(10, 4, backend=backend)
    dirs = [str(tmp_path / f"node{i}") for i in range(16)] if on_disk else None
    mgr = bn.NativeBlockManager(codec, 16, dirs)
    blocks = [pattern_block(300_000 + 64 * i, 1200 + i) for i in range(40)]
    hashes = [bn.blake2sum(b) for b in blocks]
    mgr.rpc_put_blocks(list(zip(hashes, blocks)))
    for h in hashes:
        mgr.block_incref(h)
    st = mgr.scrub_all(batch_blocks=16)
    assert st == {"scrubbed": 40, "corruptions": 0, "device_calls": st["device_calls"], "located": 0} and st["device_calls"] >= 3
    who7, who9 = mgr.storage_nodes_of(hashes[7]), mgr.storage_nodes_of(hashes[9])
    mgr.node_corrupt_shard(who7[12], hashes[7], 12, 4242, 0x80, fix_checksum=True)      # a parity shard
    mgr.node_corrupt_shard(who9[3], hashes[9], 3, 17, 0x01, fix_checksum=True)          # a data shard
    assert mgr.scrub(hashes) == [hashes[7], hashes[9]]
    st = mgr.scrub_all()
    assert st["corruptions"] == 2 and st["located"] == 2
    assert not mgr.node_has_shard(who7[12], hashes[7], 12) and not mgr.node_has_shard(who9[3], hashes[9], 3)
    if on_disk:
        hx = hashes[9].hex()
        assert (tmp_path / f"node{who9[3]}" / hx[:2] / hx[2:4] / f"{hx}.s3.corrupted").exists()
    assert mgr.resync_run()["rebuilt"] == 2
    assert mgr.scrub_all()["corruptions"] == 0 and mgr.scrub_state()[0] == 2
    assert mgr.rpc_get_blocks(hashes, 400_000) == blocks
    assert mgr.repair_all() == 40 and mgr.resync_queue_len() >= 40
    assert mgr.resync_run()["ok"] == 40


def test_hedged_read_decodes_around_a_slow_node(codec):
    """with a hedge delay, a read whose data-shard holder is slow asks the parity
    holders too and decodes from whichever k shards arrive first -- same bytes, without waiting for the slow node."""
    import time

    mgr = _mgr(codec)
    blocks = [pattern_block(200_000 + 4099 * i, 900 + i) + bytes([i]) for i in range(32)]
    hashes = [bn.blake2sum(b) for b in blocks]
    mgr.rpc_put_blocks(list(zip(hashes, blocks)))
    who = mgr.storage_nodes_of(hashes[0])
    mgr.node_set_latency(who[0], 400_000)
    t0 = time.perf_counter()
    assert mgr.rpc_get_block(hashes[0]) == blocks[0]
    assert time.perf_counter() - t0 >= 0.39 and mgr.hedged_reads == 0
    rec0 = mgr.metrics["ec_reconstructs"]
    mgr.set_read_hedge(10_000)
    t0 = time.perf_counter()
    assert mgr.rpc_get_block(hashes[0]) == blocks[0]
    assert time.perf_counter() - t0 < 0.35
    assert mgr.hedged_reads >= 1 and mgr.metrics["ec_reconstructs"] == rec0 + 1
    t0 = time.perf_counter()
    assert mgr.rpc_get_blocks(hashes, 1 << 19) == blocks
    assert time.perf_counter() - t0 < 0.39
    # hedging never lowers the bar: with m other holders down the slow shard is needed and is waited for
    down = [w for w in who[1:]][-codec.m:]
    for w in down:
        mgr.node_set_down(w, True)
    assert mgr.rpc_get_block(hashes[0]) == blocks[0]


# ----------------------------------------------------------------- round-2 advisor items
@pytest.mark.parametrize("writes", [3, 2])
def test_older_shard_headers_are_read_and_rewritten_unknown_versions_are_left_alone(tmp_path, backend, writes):
    """A manager writes ONE header version (its codec's checksum kind) and reads all three: version 1 (round 1: plain
    blake2sum) and the other of 2 / 3 are verified on the host with THEIR checksum the first time they are read and rewritten
    in the manager's own version; a version this build does not know is never renamed or deleted."""
    import hashlib

    codec = g.ReedSolomon(10, 4, backend=backend, shardsum=writes)
    other = 5 - writes
    mgr = _mgr(codec, tmp_path)
    assert mgr.shard_version == writes
    data = pattern_block(500_000, salt=77)
    h = bn.blake2sum(data)
    mgr.rpc_put_block(h, data)
    who = mgr.storage_nodes_of(h)
    hx = h.hex()

    def shard_file(j):
        return tmp_path / f"node{who[j]}" / hx[:2] / hx[2:4] / f"{hx}.s{j}"

    # shard 0 becomes a version-1 file: same payload, checksum = plain blake2sum
    raw = bytearray(shard_file(0).read_bytes())
    assert raw[:4] == b"GECS" and raw[4] == writes
    raw[4] = 1
    raw[28:60] = hashlib.blake2b(bytes(raw[64:]), digest_size=64).digest()[:32]
    shard_file(0).write_bytes(bytes(raw))
    # shard 1 claims a version from the future
    raw1 = bytearray(shard_file(1).read_bytes())
    raw1[4] = 9
    shard_file(1).write_bytes(bytes(raw1))
    # shards 2 and 12 (a data and a parity shard) carry the OTHER current format: what a store written by a manager of the
    # other kind looks like
    for j in (2, 12):
        r = bytearray(shard_file(j).read_bytes())
        r[4] = other
        r[28:60] = g.shardsum(bytes(r[64:]), other)
        shard_file(j).write_bytes(bytes(r))
    assert mgr.rpc_get_block(h) == data                  # shards 0 and 2 are used after their own checks, shard 1 is skipped: 13 >= k
    # a READ does not write to the store: the files keep their versions, nothing was migrated ...
    assert shard_file(0).read_bytes()[4] == 1 and shard_file(2).read_bytes()[4] == other and mgr.shards_migrated == 0
    # ... unless the operator asks for migration on read
    mgr.set_migrate_on_read(True)
    assert mgr.rpc_get_block(h) == data
    assert mgr.shards_migrated == 2
    mgr.set_migrate_on_read(False)
    for j in (0, 2):
        again = shard_file(j).read_bytes()
        assert again[4] == writes and again[28:60] == bn.shardsum(again[64:], writes)   # upgraded in place
    assert shard_file(1).exists() and shard_file(1).read_bytes()[4] == 9  # untouched: not *.corrupted, not deleted
    assert not list((tmp_path / f"node{who[1]}" / hx[:2] / hx[2:4]).glob("*.corrupted"))
    # scrub walks every shard: the parity shard of the other format is verified with its own checksum, and rewritten when that
    # format is the OLDER one -- maintenance migrates upwards only (a version-2 manager leaves version-3 shards as they are: two
    # managers of different kinds over one store converge instead of rewriting each other's shards at every scrub)
    assert shard_file(12).read_bytes()[4] == other
    mgr.scrub_all()
    again = shard_file(12).read_bytes()
    if other < writes:
        assert again[4] == writes and again[28:60] == bn.shardsum(again[64:], writes) and mgr.shards_migrated == 3
    else:
        assert again[4] == other and again[28:60] == g.shardsum(again[64:], other) and mgr.shards_migrated == 2
    # an old-format file whose payload does not match ITS checksum IS corrupt
    for j, ver in ((0, 1), (2, other)):
        r = bytearray(shard_file(j).read_bytes())
        r[4] = ver
        r[28:60] = g.shardsum(bytes(r[64:]), ver) if ver > 1 else hashlib.blake2b(bytes(r[64:]), digest_size=64).digest()[:32]
        r[70] ^= 1
        shard_file(j).write_bytes(bytes(r))
    assert mgr.rpc_get_block(h) == data
    assert not shard_file(0).exists() and not shard_file(2).exists()      # renamed *.corrupted, queued for resync
    # ... and resync rebuilds them in the manager's version
    mgr.resync_all()
    for j in (0, 2):
        again = shard_file(j).read_bytes()
        assert again[4] == writes and again[28:60] == bn.shardsum(again[64:], writes)
    assert mgr.rpc_get_block(h) == data
    mgr.close()


def test_a_reput_with_another_geometry_replaces_shards_only_at_quorum(backend):
    """The same block put again with a different compression setting cuts shards of another geometry.  A node parks such
    a shard beside the one in place; the manager commits it once the new stripe has its write quorum and drops it
    otherwise -- a put that fails half way must not leave 7 shards of each kind."""
    codec = g.ReedSolomon(10, 4, backend=backend)
    mgr = _mgr(codec)
    data = bytes(pattern_block(600_000, salt=5))
    h = bn.blake2sum(data)
    mgr.rpc_put_block(h, data)                                       # Plain
    assert not mgr.rpc_get_raw_block(h)[0].is_compressed()
    who = mgr.storage_nodes_of(h)
    assert bn.lib.gbm_set_compression_level(mgr._h, 1, 1) == 0
    for node in who[:7]:                                             # 7 of the 14 holders unreachable: quorum (12) fails
        mgr.node_set_down(node, True)
    with pytest.raises(bn.Quorum):
        mgr.rpc_put_block(h, data)                                   # Compressed this time
    for node in who[:7]:
        mgr.node_set_down(node, False)
    hdr, stored = mgr.rpc_get_raw_block(h)
    assert not hdr.is_compressed() and stored == data                # all 14 Plain shards are still in place
    assert mgr.rpc_get_block(h) == data
    mgr.rpc_put_block(h, data)                                       # now it reaches everybody: committed
    hdr, stored = mgr.rpc_get_raw_block(h)
    assert hdr.is_compressed() and len(stored) < len(data) and mgr.rpc_get_block(h) == data
    mgr.close()


def test_resync_never_deletes_what_a_concurrent_put_acknowledges(backend):
    """A block that is deletable (its protection ran out) is put again while resync is deciding to delete it: whatever the
    interleaving, a put that returned OK leaves a readable block (the delete branch re-reads the refcount under the hash's
    mutation lock; the put stamps its protection under that lock before it writes its first shard)."""
    import threading

    codec = g.ReedSolomon(10, 4, backend=backend)
    mgr = _mgr(codec)
    mgr.set_timing(gc_delay_ms=1000)
    blocks = [bytes(pattern_block(70_000, salt=900 + i)) for i in range(12)]
    hashes = [bn.blake2sum(b) for b in blocks]
    for rnd in range(6):
        mgr.rpc_put_blocks(list(zip(hashes, blocks)))
        mgr.clock_advance(5000)                                       # every stamp is in the past: all deletable
        for h in hashes:
            mgr.put_to_resync(h, 0)
        errs = []

        def putter():
            try:
                for h, b in zip(hashes, blocks):
                    mgr.rpc_put_block(h, b)
            except Exception as e:  # noqa: BLE001
                errs.append(e)

        def resyncer():
            try:
                mgr.resync_all()
            except Exception as e:  # noqa: BLE001
                errs.append(e)

        ts = [threading.Thread(target=putter), threading.Thread(target=resyncer)]
        [t.start() for t in ts]
        [t.join() for t in ts]
        assert not errs, errs
        for h, b in zip(hashes, blocks):
            assert mgr.rpc_get_block(h) == b, f"round {rnd}: an acknowledged put lost its shards"
    mgr.close()


def test_batcher_read_side_coalesces_concurrent_gets(backend):
    """gbm_batcher_get_block: many readers, one block each at a time (GetObject's prefetch slots) -> shared batches; the
    bytes, a missing block and a short buffer come back like from rpc_get_block."""
    import threading

    codec = g.ReedSolomon(4, 2, backend=backend)
    mgr = bn.NativeBlockManager(codec, 8)
    bt = bn.Batcher(mgr, max_blocks=64, max_wait_us=2000)
    rng = np.random.default_rng(11)
    blocks = [rng.integers(0, 256, 20000 + 997 * i, dtype=np.uint8).tobytes() for i in range(24)]
    hashes = [bn.blake2sum(b) for b in blocks]
    mgr.rpc_put_blocks(list(zip(hashes, blocks)))
    got = [None] * len(blocks)
    # (a block that arrives alone goes at once, and six Python threads fetching 20 KB blocks from memory never overlap:
    # the nodes answer after 2 ms, like disks, so that the readers do)
    for nd in range(8):
        mgr.node_set_latency(nd, 2000)

    def reader(t):
        for i in range(t, len(blocks), 6):
            got[i] = bt.get_block(hashes[i], 1 << 16)

    th = [threading.Thread(target=reader, args=(t,)) for t in range(6)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert got == blocks
    st = bt.get_stats()
    assert st["blocks"] == len(blocks) and st["batches"] <= st["blocks"] and st["max_batch"] >= 2
    for nd in range(8):
        mgr.node_set_latency(nd, 0)
    with pytest.raises(bn.BlockError) as e:
        bt.get_block(b"\x5a" * 32, 100)
    assert e.value.code == bn.GBM_E_MISSING_BLOCK
    with pytest.raises(bn.BlockError) as e:
        bt.get_block(hashes[0], 8)
    assert e.value.code == bn.GBM_E_BUFFER_TOO_SMALL
    bt.close()
    mgr.close()


# ----------------------------------------------------------------- round 4: the gets stream; the block hash is a mode
def _timed_stream(mgr, h, chunk_bytes=65536, raw=False):
    """(chunks, arrival time of each chunk in seconds since the call) of a streaming get."""
    import time

    chunks, times = [], []
    t0 = [0.0]

    def sink(_ctx, p, n):
        times.append(time.perf_counter() - t0[0])
        chunks.append(ctypes.string_at(p, n))
        return 0

    cb = bn.CHUNK_FN(sink)
    t0[0] = time.perf_counter()
    if raw:
        hdr = bn.DataBlockHeader()
        rc = bn.lib.gbm_rpc_get_raw_block_streaming(mgr._h, h, None, ctypes.byref(hdr), chunk_bytes, cb, None)
    else:
        rc = bn.lib.gbm_rpc_get_block_streaming(mgr._h, h, None, chunk_bytes, cb, None)
    return rc, chunks, times


def test_streaming_get_first_chunk_long_before_the_last(backend):
    """rpc_get_block_streaming hands the stream through (manager.rs:344-363): data shard 0 leaves as soon as IT has
    arrived and ITS checksum has matched -- the k shards are asked for at once, and the first byte waits for one node
    and one checksum, not for the slowest of k nodes.  A 4 MiB block, RS(10,4), the node that holds data shard 7 answers
    after 30 ms (a slow disk, a far zone): the first chunk -- and all of shards 0..6 -- must be out long before that
    node has answered, in every mode; the hash, when it is on, runs behind the stream."""
    codec = g.ReedSolomon(10, 4, backend=backend)
    mgr = bn.NativeBlockManager(codec, 16)
    data = bytes(np.random.default_rng(8).integers(0, 256, 4 << 20, dtype=np.uint8))
    h = bn.blake2sum(data)
    mgr.rpc_put_block(h, data)
    S = g.shard_len(10, len(data))
    who = mgr.storage_nodes_of(h)
    for mode in ("off", "rebuilt", "always"):
        mgr.set_verify_block_hash(mode)
        rc, chunks, times = _timed_stream(mgr, h)      # (the first call also starts the manager's async pool)
        assert rc == 0 and b"".join(chunks) == data
        assert max(len(c) for c in chunks) <= 65536 and len(chunks[0]) == 65536
    mgr.node_set_latency(who[7], 30_000)
    for mode in ("off", "rebuilt", "always"):
        mgr.set_verify_block_hash(mode)
        rc, chunks, times = _timed_stream(mgr, h, chunk_bytes=16384)
        assert rc == 0 and b"".join(chunks) == data
        per_shard = -(-S // 16384)
        assert times[-1] >= 0.030                                  # the stream ends when the slow node has answered ...
        assert times[0] < 0.40 * times[-1], (mode, times[0], times[-1])   # ... the first chunk left long before (< 40 %)
        assert times[7 * per_shard - 1] < 0.5 * times[-1]          # and so did everything in front of the slow shard
        assert times[7 * per_shard] >= 0.030                       # shard 7 itself could not
    mgr.node_set_latency(who[7], 0)
    # degraded (data shard 3 gone): shards 0..2 still leave at once, the rebuilt one follows, bytes identical
    mgr.node_delete_shard(who[3], h, 3)
    for mode in ("off", "rebuilt", "always"):
        mgr.set_verify_block_hash(mode)
        rc, chunks, times = _timed_stream(mgr, h)
        assert rc == 0 and b"".join(chunks) == data
    assert mgr.metrics["ec_reconstructs"] == 3


def test_verify_modes_and_corrupt_shards_in_every_mode(backend):
    """Shard checksums are verified in every mode (they are the serving node's check, manager.rs:577-609): a corrupt
    shard is detected, set aside and made up for from the others, and when too few good shards are left the answer is
    CorruptData -- off / rebuilt / always alike, through rpc_get_block, the streaming form and the batcher.  The
    end-to-end block hash only differs for content that is wrong under intact checksums."""
    codec = g.ReedSolomon(10, 4, backend=backend)
    mgr = bn.NativeBlockManager(codec, 16)
    bat = bn.Batcher(mgr, 8, 100)
    k, m = 10, 4
    for i, mode in enumerate(("off", "rebuilt", "always")):
        mgr.set_verify_block_hash(mode)
        assert mgr.verify_block_hash == mode
        data = pattern_block(700_000 + 4096 * i, 31 + i)
        h = bn.blake2sum(data)
        mgr.rpc_put_block(h, data)
        who = mgr.storage_nodes_of(h)
        # one corrupt shard: detected, replaced by a decode, the caller sees the right bytes
        c0 = mgr.metrics["corruption_counter"]
        mgr.node_corrupt_shard(who[2], h, 2, 1000, 0x08)
        assert mgr.rpc_get_block(h) == data
        assert mgr.metrics["corruption_counter"] == c0 + 1 and not mgr.node_has_shard(who[2], h, 2)
        mgr.node_corrupt_shard(who[5], h, 5, 7, 0x01)
        assert b"".join(mgr.rpc_get_block_streaming(h)) == data       # found mid-stream
        mgr.node_corrupt_shard(who[7], h, 7, 70, 0x10)
        assert bat.get_block(h, len(data)) == data
        assert mgr.metrics["corruption_counter"] == c0 + 3
        # three shards are gone now; one more node down and another corrupt shard: k - 1 good ones -> CorruptData
        mgr.node_set_down(who[k], True)
        for getter in (lambda: mgr.rpc_get_block(h), lambda: mgr.rpc_get_block_streaming(h), lambda: bat.get_block(h, len(data))):
            mgr.rpc_put_block(h, data) if False else None
            mgr.node_set_down(who[k], False)
            mgr.rpc_put_block(h, data)                                 # all 14 shards back
            for j in (2, 5, 7):
                mgr.node_delete_shard(who[j], h, j)
            mgr.node_set_down(who[k], True)
            mgr.node_corrupt_shard(who[0], h, 0, 3, 0x40)
            with pytest.raises(bn.CorruptData):
                getter()
        mgr.node_set_down(who[k], False)
        mgr.rpc_put_block(h, data)
        # wrong content under a valid name, every shard checksum intact
        evil = pattern_block(len(data), 900 + i)
        mgr.rpc_put_block(h, evil)
        if mode == "always":
            with pytest.raises(bn.CorruptData):
                mgr.rpc_get_block(h)
            rc, chunks, _ = _timed_stream(mgr, h)
            assert rc == bn.GBM_E_CORRUPT_DATA and b"".join(chunks) == evil   # delivered first, failed at the tail
        else:
            assert mgr.rpc_get_block(h) == evil
            rc, chunks, _ = _timed_stream(mgr, h)
            assert rc == 0 and b"".join(chunks) == evil
        mgr.node_delete_shard(who[1], h, 1)        # now the block goes through a decode
        if mode == "off":
            assert mgr.rpc_get_block(h) == evil
        else:
            with pytest.raises(bn.CorruptData):
                mgr.rpc_get_block(h)
            rc, chunks, _ = _timed_stream(mgr, h)
            assert rc == bn.GBM_E_CORRUPT_DATA and b"".join(chunks) == evil
    bat.close()


def test_streaming_compressed_block_is_decoded_incrementally(backend):
    codec = g.ReedSolomon(10, 4, backend=backend)
    mgr = bn.NativeBlockManager(codec, 14, compression_level=1)
    data = pattern_block(3 << 20, 5)                       # compressible
    h = bn.blake2sum(data)
    mgr.rpc_put_block(h, data)
    hdr, stored = mgr.rpc_get_raw_block(h)
    assert hdr.is_compressed() and len(stored) < len(data)
    rc, chunks, _ = _timed_stream(mgr, h, chunk_bytes=100_000)
    assert rc == 0 and b"".join(chunks) == data and all(len(c) == 100_000 for c in chunks[:-1])
    rc, chunks, _ = _timed_stream(mgr, h, raw=True)
    assert rc == 0 and b"".join(chunks) == stored
    # silent damage inside the frame: the frame checksum fails the stream's tail (block.rs:78-83)
    who = mgr.storage_nodes_of(h)
    mgr.node_corrupt_shard(who[0], h, 0, 64, 0x02, fix_checksum=True)
    rc, chunks, _ = _timed_stream(mgr, h)
    assert rc == bn.GBM_E_CORRUPT_DATA


def test_range_get_reads_only_the_shards_it_touches(codec):
    """body_from_blocks_range (src/api/s3/get.rs:650-743): the reference streams the whole block and cuts it; here a
    range of a Plain block is served from the data shards it touches -- and from the whole-block stream when a shard
    is missing, does not match, or the block is stored Compressed."""
    k, m = codec.k, codec.m
    mgr = bn.NativeBlockManager(codec, k + m + 2)
    L = 1_000_003
    data = bytes(np.random.default_rng(41).integers(0, 256, L, dtype=np.uint8))
    h = bn.blake2sum(data)
    mgr.rpc_put_block(h, data)
    S = g.shard_len(k, L)
    who = mgr.storage_nodes_of(h)

    def ranged(b, e, **kw):
        before = mgr.metrics["bytes_read"]
        chunks = mgr.rpc_get_block_range(h, L, b, e, **kw)
        return b"".join(chunks), mgr.metrics["bytes_read"] - before, chunks

    # inside one shard / across a shard boundary / to the ragged end / the whole block / nothing
    got, read, _ = ranged(S + 5, S + 1005)
    assert got == data[S + 5:S + 1005] and read == S
    got, read, chunks = ranged(S - 100, 2 * S + 100, chunk_bytes=4096)
    assert got == data[S - 100:2 * S + 100] and read == 3 * S and max(map(len, chunks)) <= 4096
    got, read, _ = ranged(L - 10, L + 500)                 # the end is clamped to the block
    assert got == data[L - 10:] and read == S
    got, read, _ = ranged(0, L)
    assert got == data and read == k * S
    got, read, _ = ranged(777, 777)
    assert got == b"" and read == 0
    got, read, _ = ranged(L + 5, L + 50)                   # beyond the block: the stored block decides -- nothing
    assert got == b""
    with pytest.raises(bn.BlockError):
        mgr.rpc_get_block_range(h, L, 10, 5)
    with pytest.raises(bn.MissingBlock):
        mgr.rpc_get_block_range(bytes(32), L, 0, 10)

    # the shard the range needs is not there: the whole-block stream (parity + decode) serves the same bytes
    rebuilt = mgr.metrics["ec_reconstructs"]
    mgr.node_set_down(who[1], True)
    got, read, _ = ranged(S + 5, S + 1005)
    assert got == data[S + 5:S + 1005] and mgr.metrics["ec_reconstructs"] == rebuilt + 1
    # ... and takes over mid-range, from the byte the range had reached
    got, _, _ = ranged(10, 2 * S + 7)
    assert got == data[10:2 * S + 7]
    mgr.node_set_down(who[1], False)

    # a shard that does not match its checksum is never delivered: set aside, queued, served by the decode
    mgr.node_corrupt_shard(who[0], h, 0, 100, 0x40, fix_checksum=False)
    corrupt = mgr.metrics["corruption_counter"]
    got, _, _ = ranged(50, 5000)
    assert got == data[50:5000] and mgr.metrics["corruption_counter"] == corrupt + 1
    assert mgr.resync_queue_len() >= 1

    # the caller's block_size is wrong (a stale version table): the stored geometry wins, the bytes are still right
    got, _, _ = ranged(S + 5, S + 1005)
    assert mgr.rpc_get_block_range(h, L + 4096 * k, S + 5, S + 1005) == [data[S + 5:S + 1005]]


def test_range_get_of_a_compressed_block_goes_through_the_decoder(backend):
    codec = g.ReedSolomon(10, 4, backend=backend)
    mgr = bn.NativeBlockManager(codec, 14, compression_level=1)
    data = pattern_block(2 << 20, 9)
    h = bn.blake2sum(data)
    mgr.rpc_put_block(h, data)
    assert mgr.rpc_get_raw_block(h)[0].is_compressed()
    for b, e in [(0, 100), (123_456, 987_654), ((2 << 20) - 77, 2 << 20), (5, (2 << 20) + 9)]:
        assert b"".join(mgr.rpc_get_block_range(h, len(data), b, e, chunk_bytes=50_000)) == data[b:e]
    # the consumer stops the stream: GBM_E_ABORTED, as for the whole-block forms
    seen = []

    def sink(_ctx, p, n):
        seen.append(n)
        return 1

    cb = bn.CHUNK_FN(sink)
    rc = bn.lib.gbm_rpc_get_block_range_streaming(mgr._h, h, None, len(data), 10, 500_000, 4096, cb, None)
    assert rc == bn.GBM_E_ABORTED and seen == [4086]     # the first chunk of the stream, cut to the range


def test_zstd_encode_is_the_reexport_the_ssec_path_uses():
    """garage_block::zstd_encode (src/block/block.rs:99-106, re-exported at lib.rs:13): EncryptionParams::encrypt_block
    compresses with it before it encrypts and stores the result with prevent_compression (encryption.rs:303-316).  One frame,
    content checksum on: the stored form of a Compressed DataBlock is byte for byte what this returns."""
    data = pattern_block(700_000, 5)
    frame = bn.zstd_encode(data, 3)
    assert frame[:4] == b"\x28\xb5\x2f\xfd" and frame[4] & 0x04, "zstd magic + the content-checksum flag of the frame header"
    assert len(frame) < len(data) // 4 and bn.zstd_decode(frame) == data
    assert bn.zstd_encode(b"", 1)[:4] == b"\x28\xb5\x2f\xfd" and bn.zstd_decode(bn.zstd_encode(b"", 1)) == b""
    # what DataBlock::from_buffer stores for this block at this level is the same frame
    codec = g.ReedSolomon(3, 1, backend="cpu")
    mgr = bn.NativeBlockManager(codec, 4, compression_level=3)
    h = bn.blake2sum(data)
    mgr.rpc_put_block(h, data)
    hdr, raw = mgr.rpc_get_raw_block(h)
    assert hdr.is_compressed() and raw == frame
    # a flipped bit fails the frame checksum (DataBlock::verify, block.rs:78-83); a short buffer says how much is needed
    for pos in (len(frame) - 1, 9):                   # the checksum itself; the first block's header
        bad = bytearray(frame)
        bad[pos] ^= 0x10
        with pytest.raises(bn.CorruptData):
            bn.zstd_decode(bytes(bad))
    n = ctypes.c_size_t()
    buf = ctypes.create_string_buffer(1000)
    assert bn.lib.gbm_zstd_decode(frame, len(frame), buf, 1000, ctypes.byref(n)) == bn.GBM_E_BUFFER_TOO_SMALL and n.value == len(data)
    assert bn.lib.gbm_zstd_encode(data, len(data), 3, buf, 1000, ctypes.byref(n)) == bn.GBM_E_BUFFER_TOO_SMALL and n.value == len(frame)
    # the SSE-C shape: the caller's own (compressed, then encrypted) bytes go in with prevent_compression and come back untouched
    blob = bytes(x ^ 0x5A for x in frame)
    hb = bn.blake2sum(blob)
    mgr.rpc_put_block(hb, blob, prevent_compression=True)
    hdr, raw = mgr.rpc_get_raw_block(hb)
    assert not hdr.is_compressed() and raw == blob


@pytest.mark.parametrize("on_disk", [False, True], ids=["memory", "directories"])
def test_resync_replaces_a_shard_of_a_stale_geometry(tmp_path, on_disk, backend):
    """A node that was down while a block was put AGAIN with another compression setting comes back holding a shard of the
    old geometry: present, readable, useless beside the thirteen new ones.  The presence scan learns every shard's geometry
    (NeedShardReply carries the header), the resync replaces the odd one -- PutShard parks the new shard beside it,
    CommitShard swaps them -- and the block scrubs clean again.  (Found by tools/soak_manager.py: the scan only asked
    "is a shard there?", so such a block was flagged by every scrub and repaired by none.)"""
    codec = g.ReedSolomon(10, 4, backend=backend)
    dirs = [str(tmp_path / f"node{i}") for i in range(16)] if on_disk else None
    mgr = bn.NativeBlockManager(codec, 16, dirs, compression_level=1)
    data = pattern_block(300_000, 77)
    h = bn.blake2sum(data)
    mgr.rpc_put_block(h, data, prevent_compression=True)
    mgr.block_incref(h)
    who = mgr.storage_nodes_of(h)
    plain_hdr = mgr.node_shard_header(who[10], h, 10)
    mgr.node_set_down(who[10], True)
    mgr.node_set_down(who[3], True)
    mgr.rpc_put_block(h, data)                       # Compressed now; twelve nodes take it: the write quorum
    mgr.node_set_down(who[10], False)
    mgr.node_set_down(who[3], False)
    assert mgr.node_shard_header(who[10], h, 10) == plain_hdr and mgr.rpc_get_block(h) == data
    assert mgr.scrub([h]) == [h]                     # fourteen shards, two of them of another geometry
    mgr.put_to_resync(h, 0)
    st = mgr.resync_run()
    assert st["rebuilt"] == 2 and st["errors"] == 0
    new_hdr = mgr.node_shard_header(who[10], h, 10)
    assert new_hdr != plain_hdr and new_hdr[8] == 1                     # the compressed flag of the header
    assert new_hdr[12:24] == mgr.node_shard_header(who[0], h, 0)[12:24]   # orig_len, shard_len: the stripe's
    assert mgr.scrub([h]) == [] and mgr.rpc_get_block(h) == data
    assert mgr.rpc_get_raw_block(h)[0].is_compressed()
    if on_disk:
        left = [f for nd in range(16) for _, _, fs in os.walk(tmp_path / f"node{nd}") for f in fs if not re.fullmatch(r"[0-9a-f]{64}\.s\d+", f)]
        assert left == [], left                      # no .parked / .tmp files stay behind
    assert mgr.resync_run()["rebuilt"] == 0


def test_hedged_reads_walk_on_to_the_next_holders_when_requests_fail(backend):
    """try_call_many_inner (rpc_helper.rs:323-411): "start another on each failure".  The hedged gather launched more requests
    only for holders that were SLOW; when the first k holders all answered "no such shard" at once -- every shard still on
    the previous layout version's nodes -- the round ended empty-handed and the block was Missing to every hedged read
    (found by tools/soak_manager.py).  Absent shards, down nodes and an older layout version, each with hedging on."""
    codec = g.ReedSolomon(10, 4, backend=backend)
    mgr = bn.NativeBlockManager(codec, 16)
    blocks = [pattern_block(60_000 + 64 * i, 900 + i) for i in range(12)]
    hashes = [bn.blake2sum(b) for b in blocks]
    mgr.rpc_put_blocks(list(zip(hashes, blocks)))
    for h in hashes:
        mgr.block_incref(h)
    mgr.set_read_hedge(300)
    # data shards absent on nodes that answer at once: the parity holders are asked in the same round
    who = mgr.storage_nodes_of(hashes[0])
    for j in (0, 4, 9):
        mgr.node_delete_shard(who[j], hashes[0], j)
    assert mgr.rpc_get_block(hashes[0]) == blocks[0]
    # nodes that cannot be contacted at all
    mgr.node_set_down(who[1], True)
    assert mgr.rpc_get_block(hashes[0]) == blocks[0] and b"".join(mgr.rpc_get_block_streaming(hashes[0])) == blocks[0]
    mgr.node_set_down(who[1], False)
    # the layout moves on: every shard is on the PREVIOUS version's nodes until resync has offloaded it
    mgr.layout_update()
    assert [mgr.rpc_get_block(h) for h in hashes] == blocks
    assert mgr.rpc_get_blocks(hashes, 100_000) == blocks
    assert b"".join(mgr.rpc_get_block_streaming(hashes[3])) == blocks[3]
    assert b"".join(mgr.rpc_get_block_range(hashes[3], len(blocks[3]), 100, 40_000)) == blocks[3][100:40_000]
    assert mgr.scrub(hashes[1:]) == []               # all n shards of a block, over both versions
    mgr.repair_all()
    mgr.resync_all()
    mgr.layout_trim()
    assert [mgr.rpc_get_block(h) for h in hashes] == blocks and mgr.scrub(hashes) == []


def test_reconstruct_hash_batch_leaves_a_block_nothing_is_wanted_of_untouched(backend):
    """The contract the resync relies on (include/garage_ec.h): a block of which nothing is wanted -- every out entry NULL, or
    only entries of shards that are PRESENT as well -- is not read at all, and its in_sums / out_sums entries stay as they
    were.  The resync's rebuild pass used to send such blocks along (a wanted shard that the gather had fetched from another
    holder after all), compared the untouched in_sums with the shard headers, and set aside all k good shards of the block
    (found by tools/soak_manager.py).  It now hands a shard it has in hand to its owner instead."""
    import numpy as np
    from garage_amd import _lib
    k, m, S, nb = 10, 4, 4160, 3
    n = k + m
    rs = g.ReedSolomon(k, m, backend=backend)
    rng = np.random.default_rng(5)
    data = rng.integers(0, 256, (nb, k, S), dtype=np.uint8)
    parity = np.zeros((nb, m, S), dtype=np.uint8)
    u8pp = ctypes.POINTER(ctypes.c_uint8)
    dp = (ctypes.c_void_p * nb)(*[data[b].ctypes.data for b in range(nb)])
    pp = (ctypes.c_void_p * nb)(*[parity[b].ctypes.data for b in range(nb)])
    lens = (ctypes.c_size_t * nb)(*[k * S] * nb)
    _lib.check(_lib.lib.gec_encode_batch(rs._h, nb, dp, lens, S, pp), "gec_encode_batch")
    full = np.concatenate([data, parity], axis=1)
    sp = (ctypes.c_void_p * (nb * n))()
    op = (ctypes.c_void_p * (nb * n))()
    outs = np.full((nb, n, S), 0xEE, dtype=np.uint8)
    # block 0: shard 3 lost and wanted; block 1: nothing lost, shard 5 "wanted" although it is present; block 2: shard 12 lost, not wanted
    for b in range(nb):
        for j in range(n):
            sp[b * n + j] = full[b, j].ctypes.data
    sp[0 * n + 3] = None
    op[0 * n + 3] = outs[0, 3].ctypes.data
    op[1 * n + 5] = outs[1, 5].ctypes.data
    sp[2 * n + 12] = None
    ins = np.full((nb, n, 32), 0xAA, dtype=np.uint8)
    osum = np.full((nb, n, 32), 0xBB, dtype=np.uint8)
    _lib.check(_lib.lib.gec_reconstruct_hash_batch(rs._h, nb, sp, op, S, 0, ins.ctypes.data_as(u8pp), osum.ctypes.data_as(u8pp)),
               "gec_reconstruct_hash_batch")
    assert np.array_equal(outs[0, 3], full[0, 3]) and osum[0, 3].tobytes() == g.shardsum(full[0, 3].tobytes())
    read0 = [j for j in range(n) if j != 3][:k]
    assert all(ins[0, j].tobytes() == g.shardsum(full[0, j].tobytes()) for j in read0)
    for b in (1, 2):                                   # nothing wanted: not read, nothing written
        assert (ins[b] == 0xAA).all() and (osum[b] == 0xBB).all() and (outs[b] == 0xEE).all()


def test_resync_hands_over_a_wanted_shard_that_turned_up_in_hand(backend):
    """The scan finds shard 2 absent; before the gather a put of the same block (a client's retry) writes it again; the
    gather -- which asks the first k holders, shard 2's among them -- brings it in.  Present AND wanted: the codec skips
    such a block (nothing is wanted of it) and leaves its in_sums untouched; the rebuild pass used to compare those with
    the headers of the k shards it had read and set all ten of them aside.  The window is held open with a slow node."""
    import threading
    import time as _t
    codec = g.ReedSolomon(10, 4, backend=backend)
    mgr = bn.NativeBlockManager(codec, 16)
    data = pattern_block(200_000, 31)
    h = bn.blake2sum(data)
    mgr.rpc_put_block(h, data)
    mgr.block_incref(h)
    who = mgr.storage_nodes_of(h)
    mgr.node_delete_shard(who[2], h, 2)
    before = mgr.block_metrics()
    mgr.node_set_latency(who[13], 150_000)           # the scan asks the nodes in shard order: 150 ms between "2 is absent" and the gather
    mgr.put_to_resync(h, 0)

    def retry_put():
        _t.sleep(0.03)
        mgr.rpc_put_block(h, data)

    t = threading.Thread(target=retry_put)
    t.start()
    st = mgr.resync_run()
    t.join()
    mgr.node_set_latency(who[13], 0)
    assert st["errors"] == 0
    assert all(mgr.node_has_shard(who[j], h, j) for j in range(14)), [mgr.node_has_shard(who[j], h, j) for j in range(14)]
    after = mgr.block_metrics()
    assert after["corruption_counter"] == before["corruption_counter"] and after["unconfirmed_verdicts"] == 0
    assert mgr.scrub([h]) == [] and mgr.rpc_get_block(h) == data


@pytest.mark.parametrize("compress", [False, True], ids=["plain", "compressed"])
def test_scrub_locates_silent_rot_with_one_parity_shard(backend, compress):
    """RS(3,1): a stripe that is RS-inconsistent with every checksum intact cannot be settled by comparing re-derived
    stripes (there is no second parity shard to compare with) -- but the block's name is the hash of its bytes, and a
    compressed block is a zstd frame with a content checksum: the shard without which the rest gives back what the name
    promises is the culprit.  Without this such a block was flagged by every scrub and repaired by none (tools/soak_manager.py
    with RS(3,1))."""
    codec = g.ReedSolomon(3, 1, backend=backend)
    mgr = bn.NativeBlockManager(codec, 5, compression_level=1 if compress else None)
    blocks = [pattern_block(90_000 + 1000 * i, 60 + i) for i in range(6)]
    hashes = [bn.blake2sum(b) for b in blocks]
    mgr.rpc_put_blocks(list(zip(hashes, blocks)))
    for h in hashes:
        mgr.block_incref(h)
    assert mgr.rpc_get_raw_block(hashes[0])[0].is_compressed() == compress
    for victim, idx in ((1, 0), (2, 2), (4, 3)):      # a data shard, another data shard, the parity shard
        who = mgr.storage_nodes_of(hashes[victim])
        mgr.node_corrupt_shard(who[idx], hashes[victim], idx, 7, 0x20, fix_checksum=True)
    assert sorted(mgr.scrub(hashes)) == sorted([hashes[1], hashes[2], hashes[4]])
    st = mgr.scrub_all()
    assert st["corruptions"] == 3 and st["located"] == 3, st
    for victim, idx in ((1, 0), (2, 2), (4, 3)):
        assert not mgr.node_has_shard(mgr.storage_nodes_of(hashes[victim])[idx], hashes[victim], idx)
    assert mgr.resync_run()["rebuilt"] == 3
    assert mgr.scrub(hashes) == [] and mgr.rpc_get_blocks(hashes, 200_000) == blocks


def test_a_hedged_round_satisfied_by_a_corrupt_parity_shard_asks_the_slow_holder_again(backend):
    """RS(3,1), hedging on, data shard 2's node slow, the parity shard corrupt.  The hedge timer fires, the parity holder answers,
    the round has its three answers and abandons the slow request -- then the parity shard fails its checksum.  The slow
    holder was "already asked", so the read gave the block up as corrupt with a good shard still out there (found by
    tools/soak_manager.py on RS(3,1)); an abandoned request's holder is asked again now."""
    codec = g.ReedSolomon(3, 1, backend=backend)
    mgr = bn.NativeBlockManager(codec, 6)
    data = pattern_block(120_000, 17)
    h = bn.blake2sum(data)
    mgr.rpc_put_block(h, data)
    mgr.block_incref(h)
    who = mgr.storage_nodes_of(h)
    mgr.node_corrupt_shard(who[3], h, 3, 11, 0x02, fix_checksum=False)
    mgr.node_set_latency(who[2], 40_000)
    mgr.set_read_hedge(300)
    for _ in range(3):
        assert mgr.rpc_get_block(h) == data
    assert b"".join(mgr.rpc_get_block_streaming(h)) == data
    mgr.node_set_latency(who[2], 0)
    assert mgr.hedged_reads >= 1 and mgr.metrics["corruption_counter"] >= 1


def test_a_put_trip_whose_checksums_the_host_cannot_reproduce_stores_nothing(backend):
    """gbm_set_put_spot_check: the device computes parity AND every shard's checksum; a device that got a checksum wrong would
    stamp shards with checksums nobody can ever confirm.  With the check on every trip, a trip that comes back with a
    falsified checksum (test hook) is refused -- GBM_E_EC, nothing reaches a node, the counters say so -- both for a direct put
    and for callers of the coalescing queue; the next trip goes through."""
    codec = g.ReedSolomon(10, 4, backend=backend)
    mgr = bn.NativeBlockManager(codec, 16)
    blocks = [pattern_block(150_000 + 64 * i, 500 + i) for i in range(9)]
    hashes = [bn.blake2sum(b) for b in blocks]
    mgr.set_put_spot_check(1)
    mgr.rpc_put_blocks(list(zip(hashes[:3], blocks[:3])))
    m0 = mgr.block_metrics()
    assert m0["put_spot_checks"] == 1 and m0["put_spot_check_failures"] == 0
    assert bn.lib.gbm_test_corrupt_put_sums(mgr._h, 1) == 0
    with pytest.raises(bn.BlockError, match="not what the host computes"):
        mgr.rpc_put_blocks(list(zip(hashes[3:6], blocks[3:6])))
    for h in hashes[3:6]:
        who = mgr.storage_nodes_of(h)
        assert not any(mgr.node_has_shard(who[j], h, j) for j in range(14))
        with pytest.raises(bn.MissingBlock):
            mgr.rpc_get_block(h)
    bt = bn.Batcher(mgr, max_blocks=8, max_wait_us=100)
    assert bn.lib.gbm_test_corrupt_put_sums(mgr._h, 1) == 0
    with pytest.raises(bn.BlockError, match="not what the host computes"):
        bt.put_block(hashes[6], blocks[6])
    bt.put_block(hashes[6], blocks[6])                  # the retry: a trip the host can vouch for
    mgr.rpc_put_blocks(list(zip(hashes[3:6], blocks[3:6])))
    assert mgr.rpc_get_blocks(hashes[:7], 200_000) == blocks[:7]
    m1 = mgr.block_metrics()
    assert m1["put_spot_check_failures"] == 2 and m1["put_spot_checks"] == 5 and m1["blocks_put"] == 3 + 1 + 3
    bt.close()
    # the default rate: one trip in sixteen is looked at
    mgr2 = bn.NativeBlockManager(codec, 16)
    for h, b in zip(hashes, blocks):
        for _ in range(4):
            mgr2.rpc_put_block(h, b)
    assert mgr2.block_metrics()["put_spot_checks"] == 3    # trips 0, 16, 32 of 36
