"""The second half of tests/c/bm_abi_fuzz.py (same rules: in contract as far as memory goes, otherwise arbitrary): manager creation
with impossible node counts / quorums / codec arrays, the batched put and get with NULL entries, zstd helpers, the hash batch, the
metrics and resync-error listings with tiny capacities, ScrubWorker commands, timing and clock jumps, worker counts, batchers
created and destroyed around a live manager, layout changes.  FUZZ_TRACE=1 prints every call.  usage: bm_abi_fuzz2.py <seed>"""
import ctypes, os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import garage_amd as g
from garage_amd import block_native as bn
rng = random.Random(int(sys.argv[1]))
L = bn.lib
k, m = rng.choice([(3, 1), (10, 4)])
rs = g.ReedSolomon(k, m, backend="cpu")
sz = ctypes.c_size_t
def P(*a):
    if os.environ.get("FUZZ_TRACE"):
        print(*a, flush=True)
# creation with odd arguments
for it in range(40):
    out = ctypes.c_void_p()
    nn = rng.choice([-1, 0, 1, k + m - 1, k + m, k + m + 3, 100000])
    wq = rng.choice([-1, 0, 1, k, k + m, k + m + 1, 1 << 20])
    P("create", nn, wq)
    rc = L.gbm_create(rs._h if rng.random() > 0.1 else None, nn, None, wq, ctypes.byref(out) if rng.random() > 0.1 else None)
    if rc == 0 and out.value:
        L.gbm_destroy(out)
    nd = rng.choice([-1, 0, 1, 2, 5])
    arr = (ctypes.c_void_p * max(nd, 1))(*[rs._h if rng.random() > 0.2 else None for _ in range(max(nd, 1))])
    out = ctypes.c_void_p()
    P("create_multi", nd, nn, wq)
    rc = L.gbm_create_multi(arr if rng.random() > 0.1 else None, nd, nn, None, wq, ctypes.byref(out))
    if rc == 0 and out.value:
        L.gbm_destroy(out)
L.gbm_destroy(None)
mgr = bn.NativeBlockManager(rs, k + m + 2)
H = mgr._h
blocks = [os.urandom(rng.choice([0, 1, 5000, 200_000])) for _ in range(6)]
hashes = [bn.blake2sum(b) for b in blocks]
mgr.rpc_put_blocks(list(zip(hashes, blocks)))
for it in range(300):
    op = rng.randrange(12)
    P("op", op)
    if op == 0:
        nb = rng.choice([0, 1, 3])
        hs = b"".join(rng.choice(hashes + [os.urandom(32)]) for _ in range(nb))
        caps_l = [rng.choice([0, 1, 300_000]) for _ in range(nb)]
        bufs = [ctypes.create_string_buffer(max(c, 1)) for c in caps_l]
        outp = (ctypes.c_void_p * max(nb, 1))(*[ctypes.addressof(b) if rng.random() > 0.1 else None for b in bufs] or [None])
        caps = (sz * max(nb, 1))(*caps_l or [0]); lens = (sz * max(nb, 1))(); rcs = (ctypes.c_int * max(nb, 1))()
        L.gbm_rpc_get_blocks(H if rng.random() > 0.1 else None, nb, hs if rng.random() > 0.1 else None, None, outp if rng.random() > 0.1 else None,
                             caps if rng.random() > 0.1 else None, lens if rng.random() > 0.1 else None, rcs if rng.random() > 0.1 else None)
    elif op == 1:
        nb = rng.choice([0, 1, 4])
        ds = [os.urandom(rng.choice([0, 1, 70_000])) for _ in range(nb)]
        hs = b"".join(bn.blake2sum(d) for d in ds)
        keep = [ctypes.create_string_buffer(d, max(len(d), 1)) for d in ds]
        dp = (ctypes.c_void_p * max(nb, 1))(*[ctypes.addressof(b) if rng.random() > 0.1 else None for b in keep] or [None])
        lens = (sz * max(nb, 1))(*[len(d) for d in ds] or [0])
        pc = bytes(rng.choice([0, 1]) for _ in range(nb))
        L.gbm_rpc_put_blocks(H if rng.random() > 0.1 else None, nb, hs if rng.random() > 0.1 else None, dp if rng.random() > 0.1 else None,
                             lens if rng.random() > 0.1 else None, pc if rng.random() > 0.5 else None, None)
    elif op == 2:
        d = os.urandom(rng.choice([0, 1, 100, 100_000]))
        cap = rng.choice([0, 1, 50, 200_000]); out = ctypes.create_string_buffer(max(cap, 1)); ln = sz()
        L.gbm_zstd_encode(d if rng.random() > 0.1 else None, len(d), rng.choice([-100, 0, 1, 3, 22, 1000]), out if rng.random() > 0.1 else None, cap, ctypes.byref(ln) if rng.random() > 0.1 else None)
        fr = out.raw[:ln.value] if rng.random() > 0.3 else os.urandom(rng.choice([0, 3, 50]))
        cap2 = rng.choice([0, 1, len(d), 1 << 20]); out2 = ctypes.create_string_buffer(max(cap2, 1)); ln2 = sz()
        L.gbm_zstd_decode(fr if rng.random() > 0.1 else None, len(fr), out2, cap2, ctypes.byref(ln2))
    elif op == 3:
        n = rng.choice([0, 1, 9])
        msgs = [os.urandom(rng.choice([0, 1, 128, 129, 10_000])) for _ in range(n)]
        arr = (ctypes.c_char_p * max(n, 1))(*[x if rng.random() > 0.1 else None for x in msgs] or [None])
        lens = (sz * max(n, 1))(*[len(x) for x in msgs] or [0])
        out = ctypes.create_string_buffer(max(32 * n, 1))
        L.gbm_blake2sum_batch(n, arr if rng.random() > 0.1 else None, lens if rng.random() > 0.1 else None, out if rng.random() > 0.1 else None)
    elif op == 4:
        cap = rng.choice([0, 1, 100, 1 << 16]); buf = ctypes.create_string_buffer(max(cap, 1)); ln = sz()
        L.gbm_metrics_prometheus(H if rng.random() > 0.1 else None, None, buf if rng.random() > 0.1 else None, cap, ctypes.byref(ln) if rng.random() > 0.1 else None)
        bmx = bn.BlockMetrics()
        L.gbm_block_metrics_get(H if rng.random() > 0.1 else None, None, ctypes.byref(bmx) if rng.random() > 0.1 else None)
    elif op == 5:
        cap = rng.choice([0, 1, 10]); arr = (bn.ResyncErrorInfo * max(cap, 1))(); n = sz()
        L.gbm_list_resync_errors(H if rng.random() > 0.1 else None, arr if rng.random() > 0.1 else None, cap, ctypes.byref(n) if rng.random() > 0.1 else None)
        L.gbm_resync_clear_backoff(H if rng.random() > 0.1 else None, rng.choice(hashes + [None, os.urandom(32)]))
        L.gbm_resync_config_persist(H if rng.random() > 0.1 else None, rng.choice([None, b"", b"/nonexistent/dir/x", b"/tmp/bmf2_resync_cfg"]))
    elif op == 6:
        L.gbm_scrub_worker_start(H if rng.random() > 0.1 else None, rng.choice([None, b"", b"/nonexistent/dir/x", b"/tmp/bmf2_scrub_info"]), rng.choice([0, 1, 16, 1 << 40]), rng.choice([0, 1, 50]))
        L.gbm_scrub_worker_command(H if rng.random() > 0.1 else None, rng.choice([-1, 0, 1, 2, 3, 4, 99]), rng.choice([0, 1, 1 << 62]))
        st = bn.ScrubStatus()
        L.gbm_scrub_worker_status(H if rng.random() > 0.1 else None, ctypes.byref(st) if rng.random() > 0.1 else None)
        if rng.random() < 0.5:
            L.gbm_scrub_worker_stop(H if rng.random() > 0.1 else None)
    elif op == 7:
        st = (ctypes.c_uint64 * 4)()
        L.gbm_scrub_all(H if rng.random() > 0.1 else None, rng.choice([0, 1, 1 << 40]), st if rng.random() > 0.1 else None)
        n = sz()
        L.gbm_repair_all(H if rng.random() > 0.1 else None, ctypes.byref(n) if rng.random() > 0.1 else None)
    elif op == 8:
        L.gbm_set_timing(H if rng.random() > 0.1 else None, rng.choice([-1, 0, 1 << 62]), rng.choice([-1, 0, 5]), rng.choice([-1, 0, 5]))
        L.gbm_clock_advance(H if rng.random() > 0.1 else None, rng.choice([0, 1, 1 << 63, (1 << 64) - 1]))
        L.gbm_set_timing(H, 600000, 60000, 3600000)
    elif op == 9:
        L.gbm_set_resync_workers(H if rng.random() > 0.1 else None, rng.choice([-1, 0, 1, 8, 9, 1000]))
        L.gbm_resync_worker_start(H if rng.random() > 0.1 else None)
        L.gbm_resync_worker_stop(H if rng.random() > 0.1 else None)
        L.gbm_set_put_spot_check(H if rng.random() > 0.1 else None, rng.choice([0, 1, 1 << 31]))
        L.gbm_set_compression_level(H if rng.random() > 0.1 else None, rng.choice([0, 1]), rng.choice([-1000, 0, 3, 22, 1000]))
        L.gbm_set_data_fsync(H if rng.random() > 0.1 else None, rng.choice([0, 1, 9]))
    elif op == 10:
        bt = ctypes.c_void_p()
        rc = L.gbm_batcher_create(H if rng.random() > 0.1 else None, rng.choice([0, 1, 8, 1 << 40]), rng.choice([0, 1, 100, 1 << 31]), ctypes.byref(bt) if rng.random() > 0.1 else None)
        if rc == 0 and bt.value:
            d = os.urandom(1000); tk = ctypes.c_void_p()
            L.gbm_batcher_submit(bt, bn.blake2sum(d), d, len(d), 0, None, ctypes.byref(tk) if rng.random() > 0.1 else None)
            if tk.value:
                L.gbm_batcher_wait(tk)
            st = (ctypes.c_uint64 * 8)()
            L.gbm_batcher_stats(bt, st if rng.random() > 0.1 else None)
            L.gbm_batcher_set_ram_buffer_max(bt, rng.choice([0, 1, 1 << 62]))
            L.gbm_batcher_destroy(bt)
        L.gbm_batcher_wait(None)
        L.gbm_batcher_destroy(None)
    else:
        L.gbm_layout_update(H if rng.random() > 0.1 else None)
        for h in hashes: L.gbm_put_to_resync(H, h, 0)
        L.gbm_resync_run(H, 1000, None)
        L.gbm_layout_trim(H if rng.random() > 0.1 else None)
mgr.close()
print("done")
