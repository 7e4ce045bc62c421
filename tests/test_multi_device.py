"""libgarage_block over SEVERAL devices (gbm_create_multi): "blocks from a batched PutObject stream are
hash-partitioned across the GPUs of one node" as product code, not as a benchmark loop.

The rule is gec_device_of_hash(hash, ndev) = hash[4] % ndev (include/garage_ec.h; the reference places by hash bytes the
same way: partition_of, src/rpc/layout/version.rs:101-104; drives, src/block/layout.rs:278-284; mutation locks,
src/block/manager.rs:679-689).  CPU suite: four GEC_BACKEND_CPU codecs as four logical devices.  GPU suite: two HIP
codecs on device 0 (what GARAGE_DRYRUN_ONE_GPU=1 means everywhere else in this repo: one box, one GPU)."""
import ctypes
import threading

import numpy as np
import pytest

import garage_amd as g
from garage_amd import _lib
from garage_amd import block_native as bn
from garage_amd.partition import gpu_of_hash
from tests.patterns import pattern_block

K, M, NNODES = 10, 4, 16


@pytest.fixture(params=[("cpu", 4), ("cpu", 3), pytest.param(("hip", 2), marks=pytest.mark.gpu)], ids=["cpu-4dev", "cpu-3dev", "hip-2on1"])
def multi(request):
    backend, ndev = request.param
    codecs = [g.ReedSolomon(K, M, device=0, backend=backend) for _ in range(ndev)]
    mgr = bn.NativeBlockManager(codecs, NNODES)
    yield mgr, ndev
    mgr.close()


def _blocks(n, size, salt0=0):
    blocks = [pattern_block(size if i % 5 else size // 2 + 7, salt0 + i) for i in range(n)]
    return blocks, [bn.blake2sum(b) for b in blocks]


def test_device_of_hash_is_the_one_rule():
    """C ABI, libgarage_block and partition.py name the same device: byte 4 of the hash, modulo the device count."""
    rng = np.random.default_rng(4)
    hashes = rng.integers(0, 256, size=(512, 32), dtype=np.uint8)
    for ndev in (1, 2, 3, 4, 8):
        want = hashes[:, 4].astype(np.int64) % ndev
        assert np.array_equal(gpu_of_hash(hashes, ndev), want)
        assert [_lib.lib.gec_device_of_hash(bytes(h), ndev) for h in hashes[:32]] == list(want[:32])
    assert _lib.lib.gec_device_of_hash(bytes(32), 0) == -1
    assert _lib.lib.gec_device_of_hash(None, 4) == -1


def test_create_multi_rejects_bad_arguments():
    a, b = g.ReedSolomon(K, M, backend="cpu"), g.ReedSolomon(3, 1, backend="cpu")
    h = ctypes.c_void_p()
    arr = (ctypes.c_void_p * 2)(a._h.value, b._h.value)
    assert bn.lib.gbm_create_multi(arr, 2, NNODES, None, 0, ctypes.byref(h)) == bn.GBM_E_INVALID_ARG  # two different codes
    arr = (ctypes.c_void_p * 2)(a._h.value, a._h.value)
    assert bn.lib.gbm_create_multi(arr, 2, NNODES, None, 0, ctypes.byref(h)) == bn.GBM_E_INVALID_ARG  # the same codec twice
    assert bn.lib.gbm_create_multi(arr, 0, NNODES, None, 0, ctypes.byref(h)) == bn.GBM_E_INVALID_ARG
    assert bn.lib.gbm_create_multi(None, 2, NNODES, None, 0, ctypes.byref(h)) == bn.GBM_E_INVALID_ARG


def test_blocks_land_on_the_device_their_hash_names(multi):
    mgr, ndev = multi
    assert mgr.device_count == ndev
    blocks, hashes = _blocks(96, 70_000)
    mgr.rpc_put_blocks(list(zip(hashes, blocks)))                  # ONE call: ndev device trips side by side
    want = np.bincount(gpu_of_hash(np.frombuffer(b"".join(hashes), dtype=np.uint8).reshape(-1, 32), ndev), minlength=ndev)
    assert [mgr.device_metrics(d)["blocks_put"] for d in range(ndev)] == list(want)
    assert all(mgr.device_of_hash(h) == h[4] % ndev for h in hashes)
    assert mgr.metrics["blocks_put"] == 96
    got = mgr.rpc_get_blocks(hashes, 70_000)
    assert got == blocks
    assert [mgr.device_metrics(d)["blocks_get"] for d in range(ndev)] == list(want)
    # single-block calls go to the same lanes
    b1 = pattern_block(300_001, 777)
    h1 = bn.blake2sum(b1)
    mgr.rpc_put_block(h1, b1)
    assert mgr.device_metrics(h1[4] % ndev)["blocks_put"] == want[h1[4] % ndev] + 1
    assert mgr.rpc_get_block(h1) == b1
    hdr, raw = mgr.rpc_get_raw_block(h1)
    assert raw == b1 and not hdr.is_compressed()
    assert b"".join(mgr.rpc_get_block_streaming(h1, chunk_bytes=50_000)) == b1


def test_failures_refcounts_resync_and_scrub_follow_the_hash(multi):
    mgr, ndev = multi
    blocks, hashes = _blocks(40, 120_000, salt0=1000)
    mgr.rpc_put_blocks(list(zip(hashes, blocks)))
    for h in hashes:
        mgr.block_incref(h)
        assert mgr.block_rc(h)[:2] == (1, "Present")
    assert mgr.scrub(hashes) == []
    st = mgr.scrub_all(batch_blocks=8)
    assert st["scrubbed"] == 40 and st["corruptions"] == 0 and st["device_calls"] >= ndev
    # a node dies: every block that had a shard there is degraded, whichever device serves it
    mgr.node_set_down(3, True)
    assert mgr.rpc_get_blocks(hashes, 120_000) == blocks
    affected = sum(1 for h in hashes if 3 in mgr.storage_nodes_of(h))
    assert affected > 0
    mgr.node_set_down(3, False)
    # lose shards of blocks on different devices, let resync rebuild them (each device's queue, side by side)
    victims = {}
    for h in hashes:
        victims.setdefault(h[4] % ndev, h)
    assert len(victims) == ndev, "40 blocks should cover every device"
    for h in victims.values():
        who = mgr.storage_nodes_of(h)
        mgr.node_delete_shard(who[1], h, 1)
        mgr.node_delete_shard(who[K], h, K)
        mgr.put_to_resync(h)
    assert mgr.resync_queue_len() >= ndev
    assert mgr.resync_all() == 2 * ndev
    for h in victims.values():
        who = mgr.storage_nodes_of(h)
        assert mgr.node_has_shard(who[1], h, 1) and mgr.node_has_shard(who[K], h, K)
    # silent corruption (checksum re-stamped): only the scrub of the device that owns the hash can find it
    h = hashes[7]
    who = mgr.storage_nodes_of(h)
    mgr.node_corrupt_shard(who[2], h, 2, 99, 0x10, fix_checksum=True)
    st = mgr.scrub_all()
    assert st["corruptions"] == 1 and st["located"] == 1
    assert mgr.scrub_state()[0] == 1
    assert mgr.resync_all() == 1
    assert mgr.scrub(hashes) == []
    # repair_all queues every hash exactly once over all devices
    assert mgr.repair_all() == 40
    # rc -> 0 and the GC delay: the owning lane deletes
    mgr.block_decref(hashes[0])
    mgr.clock_advance(bn.GBM_BLOCK_GC_DELAY_MS + 11_000)
    assert mgr.resync_all() >= K + M
    with pytest.raises(bn.MissingBlock):
        mgr.rpc_get_block(hashes[0])
    assert mgr.rpc_get_block(hashes[1]) == blocks[1]


def test_layout_change_reaches_every_lane(multi):
    mgr, ndev = multi
    blocks, hashes = _blocks(24, 66_000, salt0=5000)
    mgr.rpc_put_blocks(list(zip(hashes, blocks)))
    for h in hashes:
        mgr.block_incref(h)
    before = [mgr.storage_nodes_of(h) for h in hashes]
    assert mgr.layout_update() == 1
    assert [mgr.storage_nodes_of(h) for h in hashes] != before
    assert mgr.rpc_get_blocks(hashes, 66_000) == blocks            # read from the older version's nodes
    for h in hashes:
        mgr.put_to_resync(h)
    assert mgr.resync_all() > 0                                      # offloaded to the new owners
    mgr.layout_trim()
    assert mgr.rpc_get_blocks(hashes, 66_000) == blocks


def test_batcher_one_queue_per_device_tagged_streams_stay_ordered(multi):
    """PutObject requests (<= 3 puts in flight each, OrderTag(stream, index), src/api/s3/put.rs:42,486-511) beside
    GetObject readers through the multi-device batcher: every device's queue coalesces its own blocks, blocks of one
    stream are encoded on different devices and still reach every node in `order` order."""
    mgr, ndev = multi
    bat = bn.Batcher(mgr, max_blocks=32, max_wait_us=300)
    R, NB = 8, 9
    objects = []
    for r in range(R):
        blocks, hashes = _blocks(NB, 90_000, salt0=10_000 + 100 * r)
        objects.append((blocks, hashes))
    errors = []

    def put_object(r):
        try:
            blocks, hashes = objects[r]
            pending = []
            for i in range(NB):   # futures created in block order, <= 3 pending: `buffered(PUT_BLOCKS_MAX_PARALLEL)`
                if len(pending) == 3:
                    bat.wait(pending.pop(0))
                pending.append(bat.submit(hashes[i], blocks[i], False, (r + 1, i)))
            for tk in pending:
                bat.wait(tk)
        except Exception as e:  # pragma: no cover
            errors.append(e)

    ths = [threading.Thread(target=put_object, args=(r,)) for r in range(R)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errors
    st = bat.stats()
    assert st["blocks"] == R * NB and st["batches"] < R * NB
    per = [bat.device_stats(d)["put"]["blocks"] for d in range(ndev)]
    allh = np.frombuffer(b"".join(h for _, hs in objects for h in hs), dtype=np.uint8).reshape(-1, 32)
    assert per == list(np.bincount(gpu_of_hash(allh, ndev), minlength=ndev))
    assert sum(mgr.node_order_violations(n) for n in range(NNODES)) == 0
    # read everything back through the read side of the same queues
    out = {}

    def reader(r):
        blocks, hashes = objects[r]
        out[r] = [bat.get_block(h, 90_000) for h in hashes]

    ths = [threading.Thread(target=reader, args=(r,)) for r in range(R)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    for r in range(R):
        assert out[r] == objects[r][0]
    gper = [bat.device_stats(d)["get"]["blocks"] for d in range(ndev)]
    assert gper == per
    bat.close()
