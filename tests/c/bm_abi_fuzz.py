"""Run by tests/test_block_native.py in a process of its own: calls into libgarage_block with arguments that are in contract as
far as memory goes and otherwise arbitrary -- NULL manager / hash / data / outputs, node and shard indices out of range, zero
capacities, ranges that start beyond their end or beyond the block, unknown hashes, a NULL sink -- around a manager that holds
real blocks.  Every call must come back with a code, and the blocks must still read back afterwards.
usage: bm_abi_fuzz.py <seed> [cpu|hip] [devices]; prints "done <calls>" """
import ctypes
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import garage_amd as g  # noqa: E402
from garage_amd import block_native as bn  # noqa: E402

rng = random.Random(int(sys.argv[1]))
backend = sys.argv[2] if len(sys.argv) > 2 else "cpu"
ndev = int(sys.argv[3]) if len(sys.argv) > 3 else 1
k, m = rng.choice([(3, 1), (10, 4), (4, 2)])
codec = g.ReedSolomon(k, m, backend=backend) if ndev == 1 else [g.ReedSolomon(k, m, backend=backend) for _ in range(ndev)]
mgr = bn.NativeBlockManager(codec, k + m + 2)
bt = bn.Batcher(mgr, max_blocks=8, max_wait_us=100)
L = bn.lib
blocks = [bytes([i]) * rng.choice([0, 1, 63, 4096, 70_001, 300_000]) + os.urandom(rng.choice([0, 7, 1000])) for i in range(12)]
hashes = [bn.blake2sum(b) for b in blocks]
if os.environ.get("FUZZ_TRACE"):
    print("setup", (k, m), [len(b) for b in blocks], flush=True)
mgr.rpc_put_blocks(list(zip(hashes, blocks)))
if os.environ.get("FUZZ_TRACE"):
    print("setup done", flush=True)
for h in hashes:
    mgr.block_incref(h)
H = mgr._h
SINK = bn.CHUNK_FN(lambda ctx, p, n: 0)
STOP = bn.CHUNK_FN(lambda ctx, p, n: 1)
sz = ctypes.c_size_t
calls = 0


damaged = set()   # blocks the walk has already taken a shard from (or rotted one of): one per block at most, so that m = 1 can repair


def hsh(harm=False):
    r = rng.random()
    if r < 0.15:
        return None
    if r < 0.35:
        return os.urandom(32)
    if not harm:
        return rng.choice(hashes)
    sound = [h for h in hashes if h not in damaged]
    if not sound:
        return os.urandom(32)
    h = rng.choice(sound)
    damaged.add(h)
    return h


def mg():
    return None if rng.random() < 0.1 else H


for it in range(400):
    op = rng.randrange(16)
    if os.environ.get("FUZZ_TRACE"):
        print(it, "op", op, flush=True)
    cap = rng.choice([0, 1, 64, 400_000])
    buf = ctypes.create_string_buffer(max(cap, 1))
    ln = sz(0)
    node, idx = rng.choice([-1, 0, 3, k + m + 1, k + m + 2, 9999]), rng.choice([-1, 0, k, k + m - 1, k + m, 500])
    if op == 0:
        L.gbm_rpc_get_block(mg(), hsh(), None, buf if rng.random() > 0.1 else None, cap, ctypes.byref(ln) if rng.random() > 0.1 else None)
    elif op == 1:
        d = os.urandom(rng.choice([0, 1, 5000]))
        L.gbm_rpc_put_block(mg(), (None if rng.random() < 0.2 else os.urandom(32)) if rng.random() < 0.5 else bn.blake2sum(d), d if rng.random() > 0.1 else None, len(d),
                            rng.choice([0, 1, 7]), None)
    elif op == 2:
        L.gbm_rpc_get_block_streaming(mg(), hsh(), None, rng.choice([0, 1, 4096]), rng.choice([SINK, STOP, bn.CHUNK_FN()]), None)
    elif op == 3:
        bsz = rng.choice([0, 1, 4096, 300_000, 1 << 40])
        a, b = rng.choice([0, 1, 4095, 300_000, 1 << 41]), rng.choice([0, 1, 4096, 299_999, 1 << 42])
        L.gbm_rpc_get_block_range_streaming(mg(), hsh(), None, bsz, a, b, rng.choice([0, 64]), rng.choice([SINK, STOP, bn.CHUNK_FN()]), None)
    elif op == 4:
        hd = bn.DataBlockHeader()
        L.gbm_rpc_get_raw_block(mg(), hsh(), None, ctypes.byref(hd) if rng.random() > 0.2 else None, buf, cap, ctypes.byref(ln))
    elif op == 5:
        L.gbm_node_set_down(mg(), node, rng.choice([0, 1, 5]))
        L.gbm_node_set_down(H, node, 0)
    elif op == 6:
        L.gbm_node_has_shard(mg(), node, hsh(), idx)
        out = ctypes.create_string_buffer(64)
        L.gbm_node_shard_header(mg(), node, hsh(), idx, out)
    elif op == 7:
        L.gbm_node_corrupt_shard(mg(), node, hsh(True), idx, rng.choice([0, 63, 1 << 40]), rng.choice([0, 1, 255]), rng.choice([0, 1]))
    elif op == 8:
        L.gbm_node_delete_shard(mg(), node, hsh(True), idx)
    elif op == 9:
        nodes = (ctypes.c_int * (k + m))()
        L.gbm_storage_nodes_of(mg(), hsh(), nodes if rng.random() > 0.1 else None)
    elif op == 10:
        n = rng.choice([0, 1, 3])
        bad = (ctypes.c_uint8 * max(n, 1))()
        hs = b"".join(hsh() or os.urandom(32) for _ in range(n))
        L.gbm_scrub(mg(), n, hs if rng.random() > 0.1 else None, bad if rng.random() > 0.1 else None)
    elif op == 11:
        L.gbm_put_to_resync(mg(), hsh(), rng.choice([0, 1, 1 << 62]))
        st = (ctypes.c_uint64 * 8)()
        L.gbm_resync_run(mg(), rng.choice([0, 1, 1000]), st if rng.random() > 0.2 else None)
    elif op == 12:
        L.gbm_batcher_get_block(bt._h if rng.random() > 0.1 else None, hsh(), buf, cap, ctypes.byref(ln))
    elif op == 13:
        d = os.urandom(rng.choice([0, 1, 5000]))
        L.gbm_batcher_put_block(bt._h if rng.random() > 0.1 else None, bn.blake2sum(d) if rng.random() > 0.3 else (None if rng.random() < 0.3 else os.urandom(32)), d, len(d), 0,
                                None)
    elif op == 14:
        L.gbm_set_threads(mg(), rng.choice([-1, 0, 1, 4, 300]))
        L.gbm_set_tranquility(mg(), rng.choice([-5, -1, 0, 3]), rng.choice([-5, -1, 0, 3]))
        L.gbm_set_verify_block_hash(mg(), rng.choice([-1, 0, 1, 2, 3]))
        L.gbm_set_read_hedge(mg(), rng.choice([0, 1, 1 << 60]))
        L.gbm_set_read_hedge(H, 0)
    else:
        L.gbm_block_incref(mg(), hsh())
        rc = ctypes.c_uint64()
        L.gbm_block_rc(mg(), hsh(), ctypes.byref(rc) if rng.random() > 0.1 else None)
        L.gbm_device_of_hash(mg(), hsh())
    calls += 1
# the walk took at most one shard from a block (deleted, flipped a byte, or rotted it under a valid checksum): the scrub finds the
# rot, the resync puts everything back, and every block reads back under the end-to-end hash
mgr.set_verify_block_hash("always")
for node in range(k + m + 2):
    L.gbm_node_set_down(H, node, 0)
mgr.scrub_all()
for _ in range(3):
    for h in hashes:
        mgr.put_to_resync(h)
    mgr.resync_run()
bad = 0
for h, b in zip(hashes, blocks):
    try:
        bad += mgr.rpc_get_block(h) != b
    except bn.BlockError as e:
        print("unreadable:", h.hex()[:16], e)
        bad += 1
bt.close()
mgr.close()
print("done", calls, "unreadable", bad)
sys.exit(1 if bad else 0)
