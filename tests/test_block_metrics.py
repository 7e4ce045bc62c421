"""BlockManagerMetrics (src/block/metrics.rs:10-143) out of libgarage_block: every instrument of the reference under its
name, the value recorders as histograms over the reference exporter's boundaries (src/garage/server.rs:36-44), and the
Prometheus text the admin API's /metrics would serve -- on the CPU backend everywhere, on the HIP backend on a GPU box."""
import re

import pytest

import garage_amd as g
from garage_amd import block_native as bn
from tests.patterns import pattern_block

# opentelemetry_prometheus::exporter().with_default_histogram_boundaries(..) of the reference, src/garage/server.rs:38-43
BOUNDS = [0.001, 0.0015, 0.002, 0.003, 0.005, 0.007, 0.01, 0.015, 0.02, 0.03, 0.05, 0.07, 0.1, 0.15, 0.2, 0.3, 0.5, 0.7, 1., 1.5, 2., 3., 5., 7.,
          10., 15., 20., 30., 40., 50., 60., 70., 100.]
# the names the reference's instruments get from the OTel Prometheus exporter (dots become underscores), metrics.rs:40-143
REFERENCE_NAMES = ["block_compression_level", "block_rc_size", "block_resync_queue_length", "block_resync_errored_blocks",
                   "block_ram_buffer_free_kb", "block_resync_counter", "block_resync_error_counter", "block_resync_duration",
                   "block_resync_send_counter", "block_resync_recv_counter", "block_bytes_read", "block_read_duration", "block_bytes_written",
                   "block_write_duration", "block_delete_counter", "block_corruption_counter"]


@pytest.fixture(params=["cpu", pytest.param("hip", marks=pytest.mark.gpu)])
def backend(request):
    return request.param


def _parse(text):
    """Prometheus text exposition -> {name: {labels: value}}, checking the format's grouping rule on the way."""
    out, types, seen_done = {}, {}, set()
    cur = None
    for line in text.splitlines():
        if line.startswith("# TYPE "):
            _, _, name, ty = line.split(" ", 3)
            assert name not in types, f"{name} declared twice"
            types[name] = ty
            if cur is not None:
                seen_done.add(cur)
            cur = name
            continue
        if line.startswith("#"):
            continue
        m = re.fullmatch(r"([a-zA-Z_:][a-zA-Z0-9_:]*)(\{[^}]*\})? ([0-9.eE+-]+|NaN)", line)
        assert m, line
        name, labels, val = m.group(1), m.group(2) or "", float(m.group(3))
        base = re.sub(r"_(bucket|sum|count)$", "", name) if types.get(cur) == "histogram" else name
        assert base == cur, f"sample {name} outside its metric's group ({cur})"
        assert base not in seen_done
        out.setdefault(name, {})[labels] = val
    return out, types


def test_histogram_bounds_are_the_reference_exporters():
    b = bn.lib.gbm_histogram_bounds()
    assert [b[i] for i in range(bn.HISTOGRAM_BUCKETS)] == BOUNDS


def test_every_reference_instrument_moves_with_the_path(backend):
    codec = g.ReedSolomon(10, 4, backend=backend)
    mgr = bn.NativeBlockManager(codec, 16, compression_level=None)
    bt = bn.Batcher(mgr, max_blocks=16, max_wait_us=100)
    m0 = mgr.block_metrics(bt)
    assert m0["compression_level"] == 0 and m0["rc_size"] == 0 and m0["resync_queue_length"] == 0 and m0["devices"] == 1
    assert m0["ram_buffer_free_kb"] == 256 * 1024                         # Config.block_ram_buffer_max's default, all of it free
    assert m0["block_write_duration"]["count"] == 0 and m0["block_read_duration"]["bucket"] == [0] * 34

    blocks = [pattern_block(200_000 + 64 * i, 8800 + i) for i in range(12)]
    hashes = [bn.blake2sum(b) for b in blocks]
    S = g.shard_len(10, 200_000)
    mgr.rpc_put_blocks(list(zip(hashes[:8], blocks[:8])))                 # one call ...
    for h, b in zip(hashes[8:], blocks[8:]):
        bt.put_block(h, b)                                                # ... and four through the queue
    for h in hashes:
        mgr.block_incref(h)
    m1 = mgr.block_metrics(bt)
    assert m1["blocks_put"] == 12 and m1["rc_size"] == 12
    assert m1["bytes_written"] == sum(14 * g.shard_len(10, len(b)) for b in blocks) and m1["bytes_written"] >= 12 * 14 * S
    w = m1["block_write_duration"]
    assert 2 <= w["count"] <= 5 and w["sum_s"] > 0 and w["bucket"][-1] == w["count"]          # cumulative: the last bucket is +Inf
    assert all(a <= b for a, b in zip(w["bucket"], w["bucket"][1:]))
    assert m1["batcher_put_blocks"] == 4 and 1 <= m1["batcher_put_batches"] <= 4
    assert m1["ram_buffer_free_kb"] == 256 * 1024                         # the permits are back once the sends are over

    assert mgr.rpc_get_blocks(hashes, 300_000) == blocks
    assert mgr.rpc_get_block(hashes[3]) == blocks[3]
    assert b"".join(mgr.rpc_get_block_streaming(hashes[4])) == blocks[4]
    assert b"".join(mgr.rpc_get_block_range(hashes[5], len(blocks[5]), 10, 5000)) == blocks[5][10:5000]
    m2 = mgr.block_metrics(bt)
    assert m2["blocks_get"] == 12 + 3 and m2["block_read_duration"]["count"] == 4
    assert m2["bytes_read"] > m1["bytes_read"] and m2["corruption_counter"] == 0

    # a node dies and comes back empty; resync rebuilds what it held: resync_counter / recv_counter / duration / ec_reconstructs
    who = mgr.storage_nodes_of(hashes[0])
    for j in (0, 5, 12):
        mgr.node_delete_shard(who[j], hashes[0], j)
    mgr.put_to_resync(hashes[0], 0)
    st = mgr.resync_run()
    assert st["rebuilt"] == 3 and all(mgr.node_has_shard(who[j], hashes[0], j) for j in range(14))
    m3 = mgr.block_metrics(bt)
    assert m3["resync_counter"] == m2["resync_counter"] + st["taken"] and m3["resync_recv_counter"] == 3
    assert m3["resync_duration"]["count"] == m2["resync_duration"]["count"] + 1 and m3["resync_error_counter"] == 0
    assert m3["ec_reconstructs"] == m2["ec_reconstructs"] + 1           # blocks that went through a decode

    # a corrupt shard met on a read: corruption_counter; an unneeded block past its GC delay: delete_counter, rc_size
    mgr.node_corrupt_shard(who[1], hashes[0], 1, 50, 0x08, fix_checksum=False)
    assert mgr.rpc_get_block(hashes[0]) == blocks[0]
    m4 = mgr.block_metrics(bt)
    assert m4["corruption_counter"] == m3["corruption_counter"] + 1 and m4["resync_queue_length"] >= 1
    mgr.resync_all()
    mgr.block_decref(hashes[7])
    mgr.clock_advance(600_000 + 20_000)                                   # BLOCK_GC_DELAY + the 10 s the resync is queued behind it
    mgr.resync_all()
    m5 = mgr.block_metrics(bt)
    assert m5["delete_counter"] == 14, m5["delete_counter"]               # the block's fourteen shards, one per node
    with pytest.raises(bn.MissingBlock):
        mgr.rpc_get_block(hashes[7])

    # an error in the resync loop: a needed block with too few shards left
    for j in range(14):
        mgr.node_delete_shard(mgr.storage_nodes_of(hashes[2])[j], hashes[2], j)
    mgr.put_to_resync(hashes[2], 0)
    assert mgr.resync_run(check=False)["rc"] == bn.GBM_E_MISSING_BLOCK
    m6 = mgr.block_metrics(bt)
    assert m6["resync_error_counter"] == m5["resync_error_counter"] + 1 and m6["resync_errored_blocks"] == 1

    # `garage block list-errors` / `garage block retry-now` (manager.rs:429-449, resync.rs:119-134)
    errs = mgr.list_resync_errors()
    assert [e["hash"] for e in errs] == [hashes[2]] and errs[0]["error_count"] == 1 and errs[0]["refcount"] == 1
    assert errs[0]["next_try_ms"] == errs[0]["last_try_ms"] + 60_000                   # RESYNC_RETRY_DELAY << (errors - 1)
    assert mgr.resync_run(check=False)["skipped"] >= 0 and mgr.block_metrics(bt)["resync_counter"] == m6["resync_counter"]   # inside its back-off
    with pytest.raises(bn.BlockError, match=f"Block {hashes[3].hex()} was not in an errored state"):
        mgr.resync_clear_backoff(hashes[3])
    mgr.resync_clear_backoff(hashes[2])                                                # tried again at once ...
    st = mgr.resync_run(check=False)
    assert st["taken"] == 1 and st["errors"] == 1                                      # ... and it fails again: the back-off doubles
    errs = mgr.list_resync_errors()
    assert errs[0]["error_count"] == 2 and errs[0]["next_try_ms"] == errs[0]["last_try_ms"] + 120_000
    m6 = mgr.block_metrics(bt)

    # ---- the same as Prometheus text
    text = mgr.metrics_prometheus(bt)
    samples, types = _parse(text)
    for name in REFERENCE_NAMES:
        assert name in types, name
    assert types["block_read_duration"] == "histogram" and types["block_bytes_read"] == "counter" and types["block_rc_size"] == "gauge"
    assert samples["block_bytes_written"][""] == m6["bytes_written"] and samples["block_delete_counter"][""] == 14
    les = list(samples["block_write_duration_bucket"].keys())
    assert les == [f'{{le="{b:g}"}}' for b in BOUNDS] + ['{le="+Inf"}']
    assert samples["block_write_duration_bucket"]['{le="+Inf"}'] == samples["block_write_duration_count"][""] == m6["block_write_duration"]["count"]
    assert abs(samples["block_read_duration_sum"][""] - m6["block_read_duration"]["sum_s"]) < 1e-6
    assert "# HELP block_corruption_counter Data corruptions detected on block reads" in text
    # without the queue there is nothing to say about its permits
    assert "block_ram_buffer_free_kb" not in mgr.metrics_prometheus() and "block_ec_batcher_put_batches" not in mgr.metrics_prometheus()
    need = bn.ctypes.c_size_t()
    assert bn.lib.gbm_metrics_prometheus(mgr._h, None, None, 0, bn.ctypes.byref(need)) == bn.GBM_E_BUFFER_TOO_SMALL and need.value > 2000
    bt.close()


def test_compression_level_and_several_devices():
    codecs = [g.ReedSolomon(3, 1, backend="cpu") for _ in range(2)]
    mgr = bn.NativeBlockManager(codecs, 6, compression_level=3)
    blocks = [pattern_block(50_000 + i, 4400 + i) for i in range(20)]
    hashes = [bn.blake2sum(b) for b in blocks]
    mgr.rpc_put_blocks(list(zip(hashes, blocks)))
    assert mgr.rpc_get_blocks(hashes, 100_000) == blocks
    m = mgr.block_metrics()
    assert m["compression_level"] == 3 and m["devices"] == 2 and m["blocks_put"] == 20 and m["blocks_get"] == 20
    assert m["block_write_duration"]["count"] == 2 and m["block_read_duration"]["count"] == 2     # one trip per device
    samples, types = _parse(mgr.metrics_prometheus())
    per = samples["block_ec_device_blocks_put"]
    assert set(per) == {'{device="0"}', '{device="1"}'} and sum(per.values()) == 20 == samples["block_ec_blocks_put"][""]
    by_dev = [sum(1 for h in hashes if mgr.device_of_hash(h) == d) for d in range(2)]
    assert [per['{device="0"}'], per['{device="1"}']] == by_dev and min(by_dev) > 0
    assert sum(samples["block_ec_device_bytes_written"].values()) == samples["block_bytes_written"][""]


def test_a_snapshot_taken_while_requests_are_running_is_a_valid_histogram():
    """The exposition format wants the +Inf bucket to equal _count.  Readers and writers keep observing while gbm_block_metrics
    walks the buckets, so a snapshot's count is what its buckets say (the manager's soak found a snapshot whose separately kept
    count was one observation ahead of its buckets)."""
    import threading

    codec = g.ReedSolomon(4, 2, backend="cpu")
    mgr = bn.NativeBlockManager(codec, 8)
    blocks = [pattern_block(20_000 + 64 * i, 40 + i) for i in range(24)]
    hashes = [bn.blake2sum(b) for b in blocks]
    mgr.rpc_put_blocks(list(zip(hashes, blocks)))
    stop = threading.Event()

    def reader(off):
        i = off
        while not stop.is_set():
            assert mgr.rpc_get_block(hashes[i % len(hashes)]) == blocks[i % len(hashes)]
            i += 1

    def writer():
        i = 0
        while not stop.is_set():
            mgr.rpc_put_block(hashes[i % len(hashes)], blocks[i % len(hashes)])
            i += 1

    threads = [threading.Thread(target=reader, args=(q,)) for q in range(3)] + [threading.Thread(target=writer)]
    for t in threads:
        t.start()
    try:
        last = 0
        for _ in range(400):
            met = mgr.block_metrics()
            for name in ("block_read_duration", "block_write_duration"):
                h = met[name]
                assert h["bucket"][-1] == h["count"], (name, h)
                assert all(a <= b for a, b in zip(h["bucket"], h["bucket"][1:])), (name, h)
            assert met["block_read_duration"]["count"] >= last
            last = met["block_read_duration"]["count"]
    finally:
        stop.set()
        for t in threads:
            t.join()
    assert last > 0
    mgr.close()
