#!/usr/bin/env python3
"""Soak of the BlockManager mirror (libgarage_block) against a model: random puts (plain / compressed / through the
coalescing queue, ragged sizes down to empty), gets (whole, streaming, ranged, raw, through the queue), nodes going down
and coming back, shards deleted or corrupted under the readers, refcounts dropped and the clock moved past the GC delay
-- while THREE resync workers and the ScrubWorker run in the background the whole time.  Every byte read is compared
with what was put; at every quiesce point (all nodes up, resync drained) every live block must scrub clean and read
back, and the metrics must add up.
Over directory nodes the daemon is restarted now and then: a new manager over the same directories, the references
counted again from the model, the ScrubWorker carrying on from its record.
At some quiesce points the cluster layout changes (every block's nodes move; the old version is trimmed once a repair
pass has offloaded the strays) and a shard rots silently (checksum intact: only the scrub's RS verify can find it).
SOAK_READERS / SOAK_WRITERS: reader and writer threads beside the walk (2 / 1).
usage: soak_manager.py [seconds] [backend: hip|cpu] [max block bytes] [seed] [devices] [directory-nodes root or ""] [k m]"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

import garage_amd as g  # noqa: E402
from garage_amd import block_native as bn  # noqa: E402

# the walk's own patience (how long moves may take to settle, a resync to converge) is wall-clock time: sanitizer builds run the
# library ten to twenty times slower, and tools/soak_tsan.sh stretches it accordingly
TIME_SCALE = float(os.environ.get("SOAK_TIME_SCALE", "1"))


def soak(seconds: float = 30.0, backend: str = "hip", max_len: int = 1 << 20, seed: int = 2026, k: int = 10, m: int = 4,
         state_dir: str | None = None, verbose: bool = True, node_dirs_root: str | None = None, ndev: int = 1, layout_changes: bool = True,
         nreaders: int = 2, nwriters: int = 1) -> dict:
    rng = np.random.default_rng(seed)
    codec = g.ReedSolomon(k, m, backend=backend) if ndev == 1 else [g.ReedSolomon(k, m, backend=backend) for _ in range(ndev)]
    n, nnodes = k + m, k + m + 2
    max_down = m // 2                       # nodes down at a time; the other half of the budget of m is for damaged shards
    dirs = [os.path.join(node_dirs_root, f"node{i}") for i in range(nnodes)] if node_dirs_root else None
    layout_version = 0                      # how often the layout has changed (the manager's own counter does not survive a restart)

    def open_manager():
        m_ = bn.NativeBlockManager(codec, nnodes, dirs, compression_level=1)
        for _ in range(layout_version):    # the cluster layout is the cluster's: a restarted daemon learns the current version again
            m_.layout_update()
        m_.layout_trim()
        b_ = bn.Batcher(m_, max_blocks=32, max_wait_us=100)
        m_.set_tranquility(scrub=0, resync=0)
        m_.set_resync_workers(3)
        m_.set_put_spot_check(1)           # every put trip: one random shard + every block's first, hashed again on the host
        return m_, b_

    def start_workers():
        mgr.resync_worker_start()
        mgr.scrub_worker_start(os.path.join(state_dir, "scrub_info") if state_dir else None, batch_blocks=16, checkpoint_interval_ms=50)
        mgr.set_tranquility(scrub=0)       # (a persisted record's value wins at the start)

    mgr, bt = open_manager()
    start_workers()
    # the reader and writer threads stand still while the manager is being replaced (a daemon restart)
    gate = threading.Condition()
    active, pausing = [0], [False]

    def enter():
        with gate:
            while pausing[0]:
                gate.wait()
            active[0] += 1

    def leave():
        with gate:
            active[0] -= 1
            gate.notify_all()
    live: dict[bytes, bytes] = {}
    down: set[int] = set()
    damaged: dict[bytes, int] = {}          # shards of a live block deleted / corrupted since the last quiesce
    pinned: dict[bytes, bytes] = {}        # blocks that are never dereferenced: what the concurrent readers read
    reader_stop = threading.Event()
    reader_fail: list[str] = []
    fail_hashes: list[bytes] = []
    reader_ops = [0, 0]

    def reader(idx):
        """A GetObject beside everything else: whole, streaming, ranged and queued reads of blocks that stay referenced, while
        the main thread takes nodes down, damages shards, changes the layout and the background workers repair."""
        r2 = np.random.default_rng(seed * 1000 + idx)
        while not reader_stop.is_set():
            items = list(pinned.items())
            if not items:
                time.sleep(0.001)
                continue
            h, data = items[int(r2.integers(len(items)))]
            how = r2.random()
            enter()
            try:
                if how < 0.3:
                    got = mgr.rpc_get_block(h, max_len + 4096)
                elif how < 0.55:
                    got = b"".join(mgr.rpc_get_block_streaming(h, chunk_bytes=int(r2.choice([0, 8192]))))
                elif how < 0.75 and len(data):
                    b = int(r2.integers(0, len(data)))
                    e = int(r2.integers(b, len(data) + 1))
                    got = b"".join(mgr.rpc_get_block_range(h, len(data), b, e))
                    data = data[b:e]
                else:
                    got = bt.get_block(h, max_len + 4096)
                if got != data:
                    reader_fail.append(f"reader {idx}: wrong bytes for {h.hex()[:16]} ({len(got)} vs {len(data)})")
                    return
                reader_ops[0] += 1
                reader_ops[1] += len(data)
            except bn.BlockError as e:
                reader_fail.append(f"reader {idx}: {h.hex()[:16]}: {e}")
                fail_hashes.append(h)
                return
            finally:
                leave()

    writer_ops = [0]

    def writer(idx):
        """A PutObject beside everything else: its blocks go through the queue as futures, three in flight, tagged with the
        request's OrderTag stream (put.rs:486-511), each read back as soon as it is stored, then dereferenced again."""
        r3 = np.random.default_rng(seed * 77 + idx)
        stream = 1_000_000 * (idx + 1)
        while not reader_stop.is_set():
            stream += 1
            blocks = [r3.integers(0, 256, int(r3.integers(1, min(max_len, 200_000))), dtype=np.uint8).tobytes() for _ in range(int(r3.integers(1, 7)))]
            hashes = [bn.blake2sum(b) for b in blocks]
            pend = []
            enter()
            try:
                for order, (h, b) in enumerate(zip(hashes, blocks)):
                    pend.append(bt.submit(h, b, order_tag=(stream, order)))
                    mgr.block_incref(h)
                    if len(pend) == 3:
                        bt.wait(pend.pop(0))
                for tk in pend:
                    bt.wait(tk)
                for h, b in zip(hashes, blocks):
                    if mgr.rpc_get_block(h, max_len + 4096) != b:
                        reader_fail.append(f"writer {idx}: wrong bytes read back for {h.hex()[:16]}")
                        return
                    mgr.block_decref(h)
                writer_ops[0] += len(blocks)
            except bn.Quorum:
                for tk in pend:                                    # (the tickets own the queue's view of the buffers: wait them all out)
                    try:
                        bt.wait(tk)
                    except bn.BlockError:
                        pass
                for h in hashes:                                   # (more nodes down than the write quorum allows: the put is refused)
                    mgr.block_decref(h)
            except bn.BlockError as e:
                reader_fail.append(f"writer {idx}: {e}")
                fail_hashes.extend(hashes)
                return
            finally:
                leave()

    readers = [threading.Thread(target=reader, args=(i,)) for i in range(nreaders)] + [threading.Thread(target=writer, args=(i,)) for i in range(nwriters)]
    for t in readers:
        t.start()
    ops = dict(put=0, get=0, stream=0, range=0, raw=0, queue_get=0, queue_put=0, down=0, up=0, corrupt=0, delete=0, decref=0, clock=0,
               quiesce=0, scrub_start=0, layout_update=0, silent_rot=0)
    layout_pending = False
    settings = {"hedge_us": 0, "verify": "off"}
    nbytes = 0

    def new_block():
        kind = rng.random()
        if kind < 0.08:
            ln = int(rng.integers(0, 64))
        elif kind < 0.3:
            ln = int(rng.integers(64, 5000))
        else:
            ln = int(rng.integers(5000, max_len + 1))
        if rng.random() < 0.5:                                   # compressible: the block is stored as a zstd frame
            return bytes(np.repeat(rng.integers(0, 256, ln // 97 + 1, dtype=np.uint8), 97)[:ln])
        return rng.integers(0, 256, ln, dtype=np.uint8).tobytes()

    def check_read(h, data):
        nonlocal nbytes
        how = rng.random()
        nbytes += len(data)
        if how < 0.3:
            ops["get"] += 1
            assert mgr.rpc_get_block(h, max_len + 4096) == data, "rpc_get_block"
        elif how < 0.5:
            ops["stream"] += 1
            assert b"".join(mgr.rpc_get_block_streaming(h, chunk_bytes=int(rng.choice([0, 4096, 100_000])))) == data, "streaming get"
        elif how < 0.7 and len(data):
            ops["range"] += 1
            b = int(rng.integers(0, len(data)))
            e = int(rng.integers(b, len(data) + 1 + (rng.random() < 0.1) * 1000))
            assert b"".join(mgr.rpc_get_block_range(h, len(data), b, e, chunk_bytes=int(rng.choice([0, 8192])))) == data[b:e], "ranged get"
        elif how < 0.8:
            ops["raw"] += 1
            hdr, raw = mgr.rpc_get_raw_block(h, max_len + 4096)
            assert (bn.zstd_decode(raw, max_len + 4096) if hdr.is_compressed() else raw) == data, "raw get"
        else:
            ops["queue_get"] += 1
            assert bt.get_block(h, max_len + 4096) == data, "get through the queue"

    def diagnose(bad, what):
        """What a block that fails a check looks like right now (and a moment later: a repair may have been in flight)."""
        lines = [what]
        for h in bad[:4]:
            who = mgr.storage_nodes_of(h)
            hdrs = []
            for j in range(n):
                try:
                    b = mgr.node_shard_header(who[j], h, j)
                    hdrs.append((j, b[8], int.from_bytes(b[12:20], "little"), int.from_bytes(b[20:24], "little")))
                except bn.BlockError:
                    hdrs.append((j, None))
            if node_dirs_root:
                import glob
                files = sorted(os.path.relpath(f, node_dirs_root) for f in glob.glob(os.path.join(node_dirs_root, "node*", h.hex()[:2], h.hex()[2:4], h.hex() + "*")))
                lines.append(f"  files of {h.hex()[:16]}: {[f.replace(h.hex(), '<h>') for f in files]}")
            elsewhere = {j: [nd for nd in range(nnodes) if nd != who[j] and mgr.node_has_shard(nd, h, j)] for j in range(n)}
            lines.append(f"  shards on OTHER nodes than the current layout's: { {j: v for j, v in elsewhere.items() if v} }")
            lines.append(f"  {h.hex()[:16]} dev {mgr.device_of_hash(h)} rc {mgr.block_rc(h)} len {len(live.get(h, b''))} again-bad {mgr.scrub([h]) == [h]} "
                         f"queue {mgr.resync_queue_len()} errors {[e for e in mgr.list_resync_errors() if e['hash'] == h]} shards {hdrs}")
        time.sleep(0.3)
        lines.append(f"  0.3 s later still bad: {[h.hex()[:16] for h in mgr.scrub(bad[:4])]}; scrub worker {mgr.scrub_worker_status()}")
        return "\n".join(lines)

    def settled_scrub(hs):
        """gbm_scrub wants all n shards of a block in hand.  While a layout change is being followed the background workers
        MOVE shards (PutShard to the new owner, then DeleteShard at the old one): a gather that asked the new owner before the
        put and the old one after the delete misses that shard -- the block is whole the whole time (a read needs any k), the
        verdict "not all there" is a snapshot of a move.  What is still bad after the moves have settled is bad."""
        bad = mgr.scrub(hs)
        for _ in range(int(40 * TIME_SCALE)):
            if not bad:
                break
            ops["transient_scrub"] = ops.get("transient_scrub", 0) + 1
            time.sleep(0.02)
            bad = mgr.scrub(bad)
        return bad

    def quiesce():
        ops["quiesce"] += 1
        for nd in list(down):
            mgr.node_set_down(nd, False)
        down.clear()
        for h in damaged:
            mgr.put_to_resync(h, 0)
        deadline = time.time() + 20 * TIME_SCALE
        while True:                                              # the background workers (and this call) drain what is due
            mgr.resync_all()
            errs = mgr.list_resync_errors()
            for e in errs:                                       # a block that failed while nodes were down: retry now
                try:
                    mgr.resync_clear_backoff(e["hash"])
                except bn.BlockError:
                    pass                                         # (a background worker has just cleared it)
            if not errs and not mgr.scrub(list(damaged)):
                break
            assert time.time() < deadline, diagnose([e["hash"] for e in errs] + mgr.scrub(list(damaged)),
                                                    f"resync does not converge: errored {[(e['hash'].hex()[:12], e['error_count'], e['refcount']) for e in errs][:6]}, "
                                                    f"damaged still bad {len(mgr.scrub(list(damaged)))}, last error {bn.lib.gbm_last_error()!r}")
            time.sleep(0.01)
        damaged.clear()
        hs = list(live)
        bad = settled_scrub(hs)
        assert bad == [], diagnose(bad, "a live block does not scrub clean after the resync")
        for i in range(0, len(hs), 64):
            part = hs[i:i + 64]
            got = mgr.rpc_get_blocks(part, max_len + 4096)
            diff = [(h.hex()[:16], len(live[h]), x if isinstance(x, int) else len(x),
                     -1 if isinstance(x, int) else next((q for q in range(min(len(x), len(live[h]))) if x[q] != live[h][q]), -1))
                    for h, x in zip(part, got) if isinstance(x, int) or bytes(x) != live[h]]
            assert not diff, diagnose([bytes.fromhex(d[0]) for d in diff][:0], f"bulk get after the resync: (hash, put length, got length or error code, first differing byte) {diff[:6]}")
        met = mgr.block_metrics(bt)
        assert met["resync_errored_blocks"] == 0 and met["rc_size"] >= len(live)
        assert met["block_write_duration"]["count"] > 0 and met["block_read_duration"]["bucket"][-1] == met["block_read_duration"]["count"]
        st = mgr.scrub_worker_status()
        assert st["errors"] == 0, st
        nonlocal layout_pending, layout_version
        if layout_pending:
            # everything stored has been walked since the layout changed: strays offloaded to their new owners, the old
            # version can go (reads stop consulting it) -- and every block must still be all there
            mgr.repair_all()
            for attempt in range(200):
                mgr.resync_all()
                errs = mgr.list_resync_errors()       # (a writer's block whose put has not landed yet is "missing" to a pass)
                if not errs:
                    break
                for e in errs:
                    try:
                        mgr.resync_clear_backoff(e["hash"])
                    except bn.BlockError:
                        pass
                time.sleep(0.01)
            assert not errs, diagnose([e["hash"] for e in errs], "the repair pass after a layout change leaves errors")
            mgr.layout_trim()
            layout_pending = False
            bad = settled_scrub(hs)
            assert bad == [], diagnose(bad, "a block is not whole on its new nodes after the layout change")
        if hs and ops["quiesce"] % 3 == 0:
            # bit rot BEFORE checksumming in one shard of one block (the checksum still matches): no read in the default mode
            # can see it, the scrub's RS verify does -- it locates the shard, sets it aside, the resync rebuilds it
            h = hs[int(rng.integers(len(hs)))]
            if len(live[h]):
                ops["silent_rot"] += 1
                who = mgr.storage_nodes_of(h)
                j = int(rng.integers(k, n))                      # a parity shard: healthy reads do not touch it
                seen_before = mgr.scrub_state()[0]
                mgr.node_corrupt_shard(who[j], h, j, int(rng.integers(0, 64)), 1 << int(rng.integers(8)), fix_checksum=True)
                mgr.scrub_all(64)
                # found by this pass or, a moment earlier, by the ScrubWorker's own: either way it has been counted
                assert mgr.scrub_state()[0] >= seen_before + 1, (mgr.scrub_state(), seen_before)
                mgr.resync_all()
                assert settled_scrub([h]) == [] and mgr.rpc_get_block(h, max_len + 4096) == live[h]
        if layout_changes and ops["quiesce"] % 4 == 1:
            ops["layout_update"] += 1
            mgr.layout_update()                                  # every block's nodes move; reads consult both versions
            layout_version += 1
            layout_pending = True
        elif node_dirs_root and state_dir and not layout_pending and ops["quiesce"] % 5 == 2:
            restart()

    def restart():
        """The daemon goes down and comes back over the same directories: the shard files, scrub_info and resync_cfg are what
        survives; the refcounts are counted again from the model (the reference's block_ref table) BEFORE any worker runs."""
        nonlocal mgr, bt
        ops["restart"] = ops.get("restart", 0) + 1
        with gate:
            pausing[0] = True
            while active[0]:
                gate.wait()
        try:
            mgr.scrub_worker_stop()
            mgr.resync_worker_stop()
            bt.close()
            mgr.close()
            mgr, bt = open_manager()
            try:
                mgr.repair_all()
                raise AssertionError("a repair over an empty refcount table was not refused")
            except bn.BlockError:
                pass
            for h in live:
                mgr.block_incref(h)
            mgr.set_read_hedge(settings["hedge_us"])
            mgr.set_verify_block_hash(settings["verify"])
            start_workers()
            hs = list(live)
            for i in range(0, len(hs), 64):
                part = hs[i:i + 64]
                assert mgr.rpc_get_blocks(part, max_len + 4096) == [live[h] for h in part], "bulk get after the restart"
            assert settled_scrub(hs) == [], "a block does not scrub clean after the restart"
        finally:
            with gate:
                pausing[0] = False
                gate.notify_all()

    t0 = time.time()
    it = 0
    try:
        while time.time() - t0 < seconds:
            it += 1
            r = rng.random()
            if r < 0.30 or len(live) < 8:
                data = new_block()
                h = bn.blake2sum(data)
                pc = bool(rng.random() < 0.3)
                if rng.random() < 0.4:
                    ops["queue_put"] += 1
                    bt.put_block(h, data, prevent_compression=pc)
                else:
                    ops["put"] += 1
                    mgr.rpc_put_block(h, data, prevent_compression=pc)
                if h not in live:
                    mgr.block_incref(h)
                if down:
                    # the shards of the nodes that are down did not get written (the put went through on its write quorum; they
                    # are queued as stragglers): until the next quiesce this block is as damaged as the model lets a block be
                    damaged[h] = max(damaged.get(h, 0), len(down))
                live[h] = data
                if len(pinned) < 64:
                    pinned[h] = data
                nbytes += len(data)
            elif r < 0.62:
                h = list(live)[int(rng.integers(len(live)))]
                try:
                    check_read(h, live[h])
                except (bn.BlockError, AssertionError) as e:
                    raise AssertionError(diagnose([h], f"a read of a live block failed: {e!r}; nodes down {sorted(down)}, damaged {damaged.get(h, 0)}, "
                                                       f"settings {settings}")) from e
            elif r < 0.67 and len(down) < max_down:
                nd = int(rng.integers(nnodes))
                if nd not in down:
                    ops["down"] += 1
                    down.add(nd)
                    mgr.node_set_down(nd, True)
            elif r < 0.72 and down:
                ops["up"] += 1
                nd = down.pop()
                mgr.node_set_down(nd, False)
            elif r < 0.80:
                # one shard of a live block goes away or goes bad (its checksum no longer matches): with <= 2 nodes down and
                # <= 1 damaged shard per block since the last quiesce, every block keeps >= k good shards
                h = list(live)[int(rng.integers(len(live)))]
                if damaged.get(h, 0) == 0 and len(live[h]) > 0 and m - max_down >= 1:
                    who = mgr.storage_nodes_of(h)
                    j = int(rng.integers(n))
                    if who[j] not in down and mgr.node_has_shard(who[j], h, j):
                        damaged[h] = 1
                        if rng.random() < 0.5:
                            ops["delete"] += 1
                            mgr.node_delete_shard(who[j], h, j)
                        else:
                            ops["corrupt"] += 1
                            mgr.node_corrupt_shard(who[j], h, j, int(rng.integers(0, 64)), 1 << int(rng.integers(8)), fix_checksum=False)
            elif r < 0.84 and len(live) > 16:
                ops["decref"] += 1
                h = list(live)[int(rng.integers(len(live)))]
                if h in pinned:
                    continue
                mgr.block_decref(h)
                del live[h]
                damaged.pop(h, None)
            elif r < 0.88:
                ops["clock"] += 1
                mgr.clock_advance(int(rng.integers(1000, 400_000)))     # GC delays, back-offs and the scrub's pauses run out
            elif r < 0.91:
                try:
                    mgr.scrub_worker_command(int(rng.choice([bn.SCRUB_START, bn.SCRUB_START, bn.SCRUB_PAUSE, bn.SCRUB_RESUME, bn.SCRUB_CANCEL])), 20)
                    ops["scrub_start"] += 1
                except bn.BlockError:
                    pass                                                # does not fit the worker's state: refused, nothing changes
            if it % 150 == 0:
                quiesce()
                settings["hedge_us"] = int(rng.choice([0, 0, 300]))      # the hedged gather on some stretches
                settings["verify"] = str(rng.choice(["off", "off", "rebuilt", "always"]))
                mgr.set_read_hedge(settings["hedge_us"])
                mgr.set_verify_block_hash(settings["verify"])
            assert not reader_fail, diagnose(fail_hashes, f"{reader_fail}; nodes down {sorted(down)}, settings {settings}")
        quiesce()
    finally:
        reader_stop.set()                                        # (also when a check fails: the threads must not outlive the walk)
        for t in readers:
            t.join()
    assert not reader_fail, reader_fail
    st = mgr.scrub_worker_status()
    met = mgr.block_metrics(bt)
    # every "checksum does not match" a trip reported was confirmed by the host before a shard was set aside: a verdict the host
    # does not confirm means the checker was wrong (a block the codec skipped, a device fault) -- none must have happened
    assert met["unconfirmed_verdicts"] == 0, f"{met['unconfirmed_verdicts']} checksum verdicts were not confirmed by the host"
    assert met["put_spot_check_failures"] == 0 and met["put_spot_checks"] > 0, (met["put_spot_checks"], met["put_spot_check_failures"])
    violations = sum(mgr.node_order_violations(nd) for nd in range(nnodes))
    assert violations == 0, f"{violations} PutShard deliveries out of their stream's order"
    mgr.scrub_worker_stop()
    mgr.resync_worker_stop()
    bt.close()
    res = {"seconds": round(time.time() - t0, 1), "backend": backend, "iterations": it, "live_blocks": len(live), "GiB_checked": round(nbytes / 2**30, 2),
           "concurrent_readers": {"threads": nreaders, "reads": reader_ops[0], "GiB": round(reader_ops[1] / 2**30, 2)},
           "concurrent_writers": {"threads": nwriters, "blocks_put_tagged_and_read_back": writer_ops[0], "order_violations": violations},
           "ops": ops, "scrub_worker": {x: st[x] for x in ("blocks_scrubbed", "corruptions_detected", "checkpoints_saved", "errors")},
           "metrics": {x: met[x] for x in ("blocks_put", "blocks_get", "ec_reconstructs", "corruption_counter", "resync_counter", "resync_error_counter",
                                           "resync_recv_counter", "resync_send_counter", "delete_counter", "unconfirmed_verdicts", "put_spot_checks",
                                           "put_spot_check_failures")}}
    mgr.close()
    if verbose:
        print("soak_manager OK:", res)
    return res


if __name__ == "__main__":
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
    backend = sys.argv[2] if len(sys.argv) > 2 else "hip"
    max_len = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 20
    seed = int(sys.argv[4]) if len(sys.argv) > 4 else 2026
    ndev = int(sys.argv[5]) if len(sys.argv) > 5 else 1
    root = (sys.argv[6] or None) if len(sys.argv) > 6 else None  # directory nodes under this path (a tmpfs, preferably); "" = memory
    k, m = (int(sys.argv[7]), int(sys.argv[8])) if len(sys.argv) > 8 else (10, 4)
    soak(secs, backend, max_len, seed, k=k, m=m, ndev=ndev, node_dirs_root=root, state_dir=root, nreaders=int(os.environ.get("SOAK_READERS", "2")),
         nwriters=int(os.environ.get("SOAK_WRITERS", "1")))
