"""The (frozen) Python BlockManager mirror over libgarage_ec's own CPU backend: BASELINE config 1 -- RS(3,1) /
RS(10,4) "CPU path via BlockManager (plumbing, no GPU)" -- with product code end to end (no oracle in the path)."""
import pytest

import garage_amd as g
from tests import block_manager_cases as C


def cpu_codec(k, m):  # the PRODUCT's CPU backend
    return g.ReedSolomon(k, m, backend="cpu")


@pytest.fixture(params=[(3, 1), (10, 4)], ids=["rs3_1", "rs10_4"])
def codec(request):
    return cpu_codec(*request.param)


def test_put_get_roundtrip(codec, tmp_path):
    C.scenario_put_get_roundtrip(codec, tmp_path)


def test_survives_m_failures(codec):
    C.scenario_survives_m_failures(codec)


def test_write_quorum(codec):
    C.scenario_write_quorum(codec)


def test_corrupt_shard_detected_and_resynced(codec, tmp_path):
    C.scenario_corrupt_shard_detected_and_resynced(codec, tmp_path)


def test_scrub_finds_silent_corruption(codec):
    C.scenario_scrub_finds_silent_corruption(codec)


def test_geometry_is_a_function_of_the_block(codec):
    C.scenario_geometry_is_a_function_of_the_block(codec)


def test_datablock_api():
    C.scenario_datablock_api()


def test_needs_enough_nodes():
    from garage_amd.block_manager import BlockManager, Error, MemoryShardStore

    with pytest.raises(Error):
        BlockManager(cpu_codec(10, 4), [MemoryShardStore() for _ in range(13)])


def test_compressed_put_get_and_corruption():
    """compression_level = Some(1): zstd frame with content checksum; shards are cut from
    the compressed payload; a flipped byte inside the payload is caught by zstd's
    checksum (DataBlock::verify, src/block/block.rs:78-82)."""
    from garage_amd.block_manager import (BlockManager, CorruptData, DataBlock, DataBlockHeader, MemoryShardStore,
                                          ShardHeader)
    from garage_amd.partition import block_hash

    codec = cpu_codec(3, 1)
    stores = [MemoryShardStore() for _ in range(5)]
    mgr = BlockManager(codec, stores, compression_level=1)
    data = C.pattern_block(300_000, 5)
    h = block_hash(data)
    mgr.rpc_put_block(h, data)
    raw = mgr.rpc_get_raw_block(h)
    assert raw.header is DataBlockHeader.Compressed and len(raw.elem) < 20_000
    assert mgr.rpc_get_block(h) == data
    mgr.rpc_put_block(h, data, prevent_compression=True)
    assert mgr.rpc_get_raw_block(h).header is DataBlockHeader.Plain
    # corrupt the compressed payload consistently with the shard checksum
    mgr.rpc_put_block(h, data)
    who = mgr.storage_nodes_of(h)
    rawshard = bytearray(stores[who[0]].get(h, 0))
    rawshard[ShardHeader.SIZE + 40] ^= 0x10
    hdr = ShardHeader.unpack(bytes(rawshard))
    from garage_amd.codec import shardsum

    hdr.checksum = shardsum(bytes(rawshard[ShardHeader.SIZE:]))
    stores[who[0]].put(h, 0, hdr.pack() + bytes(rawshard[ShardHeader.SIZE:]))
    with pytest.raises(CorruptData):
        mgr.rpc_get_block(h)
    with pytest.raises(CorruptData):
        DataBlock.compressed(b"not a zstd frame").verify(h)


def test_put_with_node_down_is_repaired_not_deleted(codec):
    C.scenario_put_with_node_down_is_repaired_not_deleted(codec)
