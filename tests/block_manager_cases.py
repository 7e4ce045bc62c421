"""Shared BlockManager scenarios, run with the oracle-backed stub on CPU
(tests/test_block_manager.py) and with the real GPU codec
(tests/test_gpu_block_manager.py).  Style follows the reference: deterministic
byte patterns, put -> get -> assert_eq (src/api/s3/encryption.rs:555-596)."""
import pytest

from garage_amd.block_manager import (BlockManager, CorruptData, DataBlock, DataBlockHeader, DirShardStore,
                                      MemoryShardStore, MissingBlock, Quorum, ShardHeader)
from garage_amd.codec import shardsum
from garage_amd.partition import block_hash


def pattern_block(n: int, salt: int = 0) -> bytes:
    # the reference's test pattern: runs of (i % 256) of length (i*37) % 1024 (encryption.rs:561-566)
    out = bytearray()
    i = salt
    while len(out) < n:
        out += bytes([i % 256]) * ((i * 37) % 1024)
        i += 1
    return bytes(out[:n])


def make_manager(codec, nstores=None, tmp_path=None, **kw):
    n = codec.k + codec.m
    nstores = nstores or n + 2
    stores = [DirShardStore(str(tmp_path / f"node{i}")) if tmp_path else MemoryShardStore() for i in range(nstores)]
    return BlockManager(codec, stores, **kw), stores


def scenario_put_get_roundtrip(codec, tmp_path):
    mgr, stores = make_manager(codec, tmp_path=tmp_path)
    for size in (INLINE + 1, 65536, 500_000, 1 << 20):
        data = pattern_block(size, salt=size)
        h = block_hash(data)
        mgr.rpc_put_block(h, data)
        assert mgr.rpc_get_block(h) == data
        raw = mgr.rpc_get_raw_block(h)
        assert raw.header is DataBlockHeader.Plain and raw.elem == data
    # every node holds exactly one shard of each block, with the Garage directory naming
    h = block_hash(pattern_block(65536, salt=65536))
    who = mgr.storage_nodes_of(h)
    assert len(set(who)) == codec.k + codec.m
    for j, node in enumerate(who):
        rawshard = stores[node].get(h, j)
        hdr = ShardHeader.unpack(rawshard)
        assert (hdr.k, hdr.m, hdr.idx, hdr.orig_len) == (codec.k, codec.m, j, 65536)
        assert shardsum(rawshard[ShardHeader.SIZE:]) == hdr.checksum
    hx = h.hex()
    assert (tmp_path / f"node{who[0]}" / hx[:2] / hx[2:4] / f"{hx}.s0").exists()


INLINE = 3072


def scenario_survives_m_failures(codec):
    mgr, stores = make_manager(codec)
    data = pattern_block(300_000, 7)
    h = block_hash(data)
    mgr.rpc_put_block(h, data)
    who = mgr.storage_nodes_of(h)
    # lose m nodes, data shards first (worst case: needs a decode)
    for j in range(codec.m):
        stores[who[j]].down = True
    assert mgr.rpc_get_block(h) == data
    assert mgr.metrics["ec_reconstructs"] == 1
    # one more failure: unrecoverable -> MissingBlock like "no node returned a valid block"
    stores[who[codec.m]].down = True
    with pytest.raises(MissingBlock):
        mgr.rpc_get_block(h)


def scenario_write_quorum(codec):
    mgr, stores = make_manager(codec)
    data = pattern_block(100_000, 9)
    h = block_hash(data)
    who = mgr.storage_nodes_of(h)
    # tolerate floor(m/2) down nodes on write; stragglers queued for resync
    tolerated = codec.k + codec.m - mgr.write_quorum
    for j in range(tolerated):
        stores[who[-1 - j]].down = True
    mgr.rpc_put_block(h, data)
    assert (h in mgr.resync_queue) == (tolerated > 0)
    assert mgr.rpc_get_block(h) == data
    stores[who[0]].down = True
    with pytest.raises(Quorum) as ei:
        mgr.rpc_put_block(h, data)
    assert ei.value.ok == mgr.write_quorum - 1 and "Could not reach quorum" in str(ei.value)


def scenario_corrupt_shard_detected_and_resynced(codec, tmp_path):
    mgr, stores = make_manager(codec, tmp_path=tmp_path)
    data = pattern_block(200_000, 11)
    h = block_hash(data)
    mgr.rpc_put_block(h, data)
    mgr.block_incref(h)
    who = mgr.storage_nodes_of(h)
    # flip a byte in data shard 1 on disk, delete parity shard k
    p = stores[who[1]]._path(h, 1)
    raw = bytearray(open(p, "rb").read())
    raw[ShardHeader.SIZE + 1234] ^= 0x55
    open(p, "wb").write(raw)
    lost = 1
    if codec.m >= 2:                                  # a second loss only if the code can take it
        stores[who[codec.k]].delete(h, codec.k)
        lost = 2
    assert mgr.rpc_get_block(h) == data               # read repairs around it
    assert mgr.metrics["corruption_counter"] == 1
    assert (tmp_path / f"node{who[1]}" / h.hex()[:2] / h.hex()[2:4] / f"{h.hex()}.s1.corrupted").exists()
    fixed = mgr.resync_all()
    assert fixed == lost
    assert mgr.scrub([h]) == []
    for j, node in enumerate(who):
        assert stores[node].get(h, j) is not None
    # rc -> 0: nothing is deleted inside BLOCK_GC_DELAY, every shard after it
    mgr.block_decref(h)
    assert mgr.resync_all() == 0 and mgr.rpc_get_block(h) == data
    mgr.clock_advance(mgr.gc_delay_ms + 11_000)
    assert mgr.resync_all() == codec.k + codec.m
    with pytest.raises(MissingBlock):
        mgr.rpc_get_block(h)


def scenario_scrub_finds_silent_corruption(codec):
    mgr, stores = make_manager(codec)
    blocks = [pattern_block(150_000, s) for s in range(5)]
    hashes = [block_hash(b) for b in blocks]
    mgr.rpc_put_blocks(list(zip(hashes, blocks)))      # one batched device call
    assert mgr.scrub(hashes) == []
    # corrupt a parity shard *consistently with its checksum* (bit rot before checksumming)
    h = hashes[3]
    who = mgr.storage_nodes_of(h)
    j = codec.k
    raw = bytearray(stores[who[j]].get(h, j))
    raw[ShardHeader.SIZE + 77] ^= 1
    hdr = ShardHeader.unpack(bytes(raw))
    hdr.checksum = shardsum(bytes(raw[ShardHeader.SIZE:]))
    stores[who[j]].put(h, j, hdr.pack() + bytes(raw[ShardHeader.SIZE:]))
    assert mgr.scrub(hashes) == [h]
    # wrong content under a valid name is caught by the block hash on read of a plain block
    evil = pattern_block(150_000, 99)
    mgr.rpc_put_block(hashes[0], evil)
    with pytest.raises(CorruptData):
        mgr.rpc_get_block(hashes[0])


def scenario_datablock_api():
    b = DataBlock.plain(b"abc")
    assert b.into_parts() == (DataBlockHeader.Plain, b"abc") and not b.header.is_compressed()
    assert DataBlock.from_parts(DataBlockHeader.Compressed, b"x").header.is_compressed()
    b.verify(block_hash(b"abc"))
    with pytest.raises(CorruptData):
        b.verify(block_hash(b"abd"))
    # no zstd in this image: from_buffer falls back to Plain exactly like an encoder error would
    assert DataBlock.from_buffer(b"hello" * 100, None).header is DataBlockHeader.Plain
    blk = DataBlock.from_buffer(b"hello" * 100, 1)
    assert blk.header in (DataBlockHeader.Plain, DataBlockHeader.Compressed)
    hdr = ShardHeader(10, 4, 3, False, 12345, 1280, bytes(range(32)))
    assert len(hdr.pack()) == 64 and ShardHeader.unpack(hdr.pack()) == hdr
    with pytest.raises(ValueError):
        ShardHeader.unpack(b"\0" * 64)


def scenario_geometry_is_a_function_of_the_block(codec):
    """Regression (found by ASan in tests/c/block_manager_host_test): a block put in a
    batch with a larger block must get the same shards as when put alone, and a
    half-failed re-put must not poison reads."""
    mgr, stores = make_manager(codec)
    small, big = pattern_block(200_000, 21), pattern_block(1 << 20, 22)
    hs, hb = block_hash(small), block_hash(big)
    mgr.rpc_put_blocks([(hb, big), (hs, small)])
    who = mgr.storage_nodes_of(hs)
    batched = [stores[who[j]].get(hs, j) for j in range(codec.k + codec.m)]
    mgr.rpc_put_block(hs, small)
    alone = [stores[who[j]].get(hs, j) for j in range(codec.k + codec.m)]
    assert batched == alone
    import garage_amd

    assert ShardHeader.unpack(alone[0]).shard_len == garage_amd.shard_len(codec.k, len(small))
    mgr.block_incref(hs)
    hdr = ShardHeader(codec.k, codec.m, 0, False, 1000, 64, shardsum(bytes(64)))
    stores[who[0]].put(hs, 0, hdr.pack() + bytes(64))     # stale shard of another geometry
    assert mgr.rpc_get_block(hs) == small                  # majority geometry wins
    assert mgr.resync_all() >= 1                           # and resync overwrites the stray shard
    assert ShardHeader.unpack(stores[who[0]].get(hs, 0)).orig_len == len(small)


def scenario_put_with_node_down_is_repaired_not_deleted(codec):
    """a put that reached its quorum with a node down queues the block for resync; nobody
    has incref'ed it yet (PutObject does that concurrently).  Resync must rebuild the straggler, never
    delete the block."""
    mgr, stores = make_manager(codec, write_quorum=codec.k)
    data = pattern_block(250_000, 31)
    h = block_hash(data)
    who = mgr.storage_nodes_of(h)
    stores[who[-1]].down = True
    mgr.rpc_put_block(h, data)
    stores[who[-1]].down = False
    assert mgr.resync_all() == 1                       # the missing shard, rebuilt
    assert stores[who[-1]].get(h, codec.k + codec.m - 1) is not None
    assert mgr.rpc_get_block(h) == data
    mgr.block_incref(h)
    mgr.clock_advance(mgr.gc_delay_ms + 11_000)
    assert mgr.resync_all() == 0 and mgr.rpc_get_block(h) == data
