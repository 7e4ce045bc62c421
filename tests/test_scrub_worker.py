"""The continuously running ScrubWorker (src/block/repair.rs:156-500) in libgarage_block: the state machine and its
commands (Start / Pause / Resume / Cancel with the reference's refusals), the schedule (SCRUB_INTERVAL + 0..10 days, driven
here by the manager's clock), the persisted record (tranquility, times, corruptions, CHECKPOINT) and a restart that
carries on from the checkpoint -- on the CPU backend everywhere and on the HIP backend on a GPU box."""
import os
import time

import pytest

import garage_amd as g
from garage_amd import block_native as bn
from tests.patterns import pattern_block

DAY = 24 * 3600 * 1000


@pytest.fixture(params=["cpu", pytest.param("hip", marks=pytest.mark.gpu)])
def backend(request):
    return request.param


def _store(mgr, n, size=40_000, salt=7000):
    blocks = [pattern_block(size + 64 * i, salt + i) for i in range(n)]
    hashes = [bn.blake2sum(b) for b in blocks]
    mgr.rpc_put_blocks(list(zip(hashes, blocks)))
    for h in hashes:
        mgr.block_incref(h)
    return hashes, blocks


def _wait(cond, what, timeout=30.0):
    t0 = time.time()
    while time.time() - t0 < timeout:
        if cond():
            return
        time.sleep(0.005)
    raise AssertionError(f"timed out waiting for: {what}")


def test_commands_follow_the_reference_state_machine(backend, tmp_path):
    codec = g.ReedSolomon(10, 4, backend=backend)
    mgr = bn.NativeBlockManager(codec, 16)
    hashes, _ = _store(mgr, 60)
    assert mgr.scrub_worker_status()["state"] == bn.SCRUB_NO_WORKER
    with pytest.raises(bn.BlockError, match="no scrub worker"):
        mgr.scrub_worker_command(bn.SCRUB_START)
    state = str(tmp_path / "scrub_info")
    t0 = int(time.time() * 1000)
    mgr.scrub_worker_start(state, batch_blocks=8)
    st = mgr.scrub_worker_status()
    # ScrubWorkerPersisted::default: nothing done, the first run 25..35 days away, INITIAL_SCRUB_TRANQUILITY
    assert st["state"] == bn.SCRUB_FINISHED and st["progress"] == 1.0 and st["time_last_complete_scrub_ms"] == 0
    assert t0 + 25 * DAY - 1000 <= st["time_next_run_scrub_ms"] <= t0 + 35 * DAY + 60_000
    assert st["tranquility"] == 4
    mgr.set_tranquility(scrub=0)
    # commands that do not fit the state are refused with the reference's words (repair.rs:349,370,383,397)
    for cmd, msg in [(bn.SCRUB_PAUSE, "Cannot pause scrub worker: not running!"), (bn.SCRUB_RESUME, "Cannot resume scrub worker: not paused!"),
                     (bn.SCRUB_CANCEL, "Cannot cancel scrub worker: not running!")]:
        with pytest.raises(bn.BlockError, match=msg):
            mgr.scrub_worker_command(cmd)
    with pytest.raises(bn.BlockError, match="unknown scrub worker command"):
        mgr.scrub_worker_command(17)

    # Start: a pass over everything, then Finished again with the times moved on
    mgr.scrub_worker_command(bn.SCRUB_START)
    _wait(lambda: mgr.scrub_worker_status()["time_last_complete_scrub_ms"] > 0, "the pass to complete")
    st = mgr.scrub_worker_status()
    assert st["state"] == bn.SCRUB_FINISHED and st["blocks_scrubbed"] == 60 and st["corruptions_detected"] == 0 and st["errors"] == 0
    assert st["time_last_complete_scrub_ms"] >= t0 and st["time_next_run_scrub_ms"] >= st["time_last_complete_scrub_ms"] + 25 * DAY
    assert mgr.scrub_state()[1] == st["time_last_complete_scrub_ms"]

    # Pause right behind Start: the pass stands still where it is, Start is refused, Pause may be repeated, Resume ends it
    mgr.set_tranquility(scrub=50)       # slow steps: the pause lands inside the pass
    mgr.scrub_worker_command(bn.SCRUB_START)
    with pytest.raises(bn.BlockError, match="Cannot start scrub worker: already running!"):
        mgr.scrub_worker_command(bn.SCRUB_START)
    mgr.scrub_worker_command(bn.SCRUB_PAUSE, 3600_000)
    st = mgr.scrub_worker_status()
    assert st["state"] == bn.SCRUB_PAUSED and st["progress"] < 1.0 and st["resume_at_ms"] >= int(time.time() * 1000) + 3500_000
    with pytest.raises(bn.BlockError, match="already running"):
        mgr.scrub_worker_command(bn.SCRUB_START)
    mgr.scrub_worker_command(bn.SCRUB_PAUSE, 7200_000)
    time.sleep(0.05)
    frozen = mgr.scrub_worker_status()
    time.sleep(0.1)
    again = mgr.scrub_worker_status()
    assert again["blocks_scrubbed"] == frozen["blocks_scrubbed"] and again["progress"] == frozen["progress"]
    mgr.set_tranquility(scrub=0)
    mgr.scrub_worker_command(bn.SCRUB_RESUME)
    _wait(lambda: mgr.scrub_worker_status()["state"] == bn.SCRUB_FINISHED, "the resumed pass to complete")
    assert mgr.scrub_worker_status()["blocks_scrubbed"] >= 120          # every block once per pass (a dropped step is done again)

    # a pause ends by itself when its time is up (wait_for_work, :500-505) -- the manager's clock is moved
    mgr.set_tranquility(scrub=50)
    mgr.scrub_worker_command(bn.SCRUB_START)
    mgr.scrub_worker_command(bn.SCRUB_PAUSE, 60_000)
    mgr.set_tranquility(scrub=0)
    assert mgr.scrub_worker_status()["state"] == bn.SCRUB_PAUSED
    mgr.clock_advance(61_000)
    _wait(lambda: mgr.scrub_worker_status()["state"] == bn.SCRUB_FINISHED, "the pause to end and the pass to complete")

    # Cancel drops the pass and the checkpoint
    mgr.set_tranquility(scrub=50)
    mgr.scrub_worker_command(bn.SCRUB_START)
    mgr.scrub_worker_command(bn.SCRUB_CANCEL)
    st = mgr.scrub_worker_status()
    assert st["state"] == bn.SCRUB_FINISHED and st["progress"] == 1.0
    mgr.set_tranquility(scrub=0)

    # the schedule: nothing happens before time_next_run_scrub, a pass starts by itself once it is reached
    before = mgr.scrub_worker_status()
    mgr.clock_advance(24 * DAY)
    time.sleep(0.1)
    assert mgr.scrub_worker_status()["blocks_scrubbed"] == before["blocks_scrubbed"]
    mgr.clock_advance(12 * DAY)
    _wait(lambda: mgr.scrub_worker_status()["time_last_complete_scrub_ms"] > before["time_last_complete_scrub_ms"], "the scheduled pass")
    st = mgr.scrub_worker_status()
    assert st["blocks_scrubbed"] == before["blocks_scrubbed"] + 60
    assert st["time_next_run_scrub_ms"] >= st["time_last_complete_scrub_ms"] + 25 * DAY
    mgr.scrub_worker_stop()
    assert mgr.scrub_worker_status()["state"] == bn.SCRUB_NO_WORKER
    assert os.path.getsize(state) == 72


@pytest.mark.parametrize("on_disk", [False, True], ids=["memory", "directories"])
def test_a_restart_carries_on_from_the_checkpoint(backend, tmp_path, on_disk):
    """ScrubWorker::new over a state file that holds a checkpoint: Running from there (repair.rs:307-326).  The second
    worker object scrubs what the first had not reached -- not the whole store again -- and the counters it inherits
    (corruptions_detected, tranquility) are the persisted ones."""
    codec = g.ReedSolomon(10, 4, backend=backend)
    dirs = [str(tmp_path / f"node{i}") for i in range(16)] if on_disk else None
    mgr = bn.NativeBlockManager(codec, 16, dirs)
    hashes, blocks = _store(mgr, 96, size=30_000)
    order = sorted(hashes)
    # one block far down the walk is silently wrong (checksum "fixed"): only the device's RS verify can see it
    victim = order[80]
    who = mgr.storage_nodes_of(victim)
    mgr.node_corrupt_shard(who[11], victim, 11, 99, 0x10, fix_checksum=True)
    state = str(tmp_path / "scrub_info")
    mgr.set_tranquility(scrub=7)
    mgr.scrub_worker_start(state, batch_blocks=8, checkpoint_interval_ms=1)
    mgr.set_tranquility(scrub=200)          # (after the start: the steps crawl, the stop lands mid-pass)
    mgr.scrub_worker_command(bn.SCRUB_START)
    _wait(lambda: mgr.scrub_worker_status()["blocks_scrubbed"] >= 16, "two steps")
    mgr.scrub_worker_stop()                 # the process "dies" here; the state file holds the walk's position
    mgr.set_tranquility(scrub=7)
    first = os.path.getsize(state)
    assert first == 72

    mgr.scrub_worker_start(state, batch_blocks=8)
    st = mgr.scrub_worker_status()
    assert st["tranquility"] == 200                                     # the persisted value wins (repair.rs:26-27)
    mgr.set_tranquility(scrub=0)
    _wait(lambda: mgr.scrub_worker_status()["state"] == bn.SCRUB_FINISHED, "the carried-on pass to complete")
    st = mgr.scrub_worker_status()
    assert 0 < st["blocks_scrubbed"] <= 96 - 16, st                      # only what was left
    assert st["corruptions_detected"] == 1 and st["time_last_complete_scrub_ms"] > 0
    assert not mgr.node_has_shard(who[11], victim, 11)                   # located, set aside ...
    assert mgr.resync_run()["rebuilt"] == 1                              # ... and rebuilt by the resync it queued
    assert mgr.rpc_get_blocks(hashes, 60_000) == blocks
    mgr.scrub_worker_stop()

    # a third worker object: no checkpoint in the file any more -> Finished, the counters still there
    mgr.scrub_worker_start(state)
    st3 = mgr.scrub_worker_status()
    assert st3["state"] == bn.SCRUB_FINISHED and st3["corruptions_detected"] == 1 and st3["tranquility"] == 0
    assert st3["time_last_complete_scrub_ms"] == st["time_last_complete_scrub_ms"]
    assert st3["time_next_run_scrub_ms"] == st["time_next_run_scrub_ms"]
    mgr.scrub_worker_stop()

    # a state file that does not decode is ignored (Persister::load's error -> Default, persister.rs:97-101)
    with open(state, "wb") as f:
        f.write(b"not a scrub record")
    mgr.scrub_worker_start(state)
    st4 = mgr.scrub_worker_status()
    assert st4["state"] == bn.SCRUB_FINISHED and st4["corruptions_detected"] == 0 and st4["time_last_complete_scrub_ms"] == 0


def test_one_worker_per_device_of_a_multi_device_manager(tmp_path):
    """gbm_create_multi: a worker and a state file per device, each walking the hashes gec_device_of_hash gives it."""
    codecs = [g.ReedSolomon(3, 1, backend="cpu") for _ in range(3)]
    mgr = bn.NativeBlockManager(codecs, 6)
    hashes, _ = _store(mgr, 45, size=9_000)
    state = str(tmp_path / "scrub_info")
    mgr.set_tranquility(scrub=0)
    mgr.scrub_worker_start(state, batch_blocks=4)
    assert mgr.scrub_worker_status()["state"] == bn.SCRUB_FINISHED
    mgr.scrub_worker_command(bn.SCRUB_START)
    _wait(lambda: mgr.scrub_worker_status()["state"] == bn.SCRUB_FINISHED and mgr.scrub_worker_status()["blocks_scrubbed"] == 45,
          "every device's pass")
    st = mgr.scrub_worker_status()
    assert st["corruptions_detected"] == 0 and st["time_last_complete_scrub_ms"] > 0 and st["progress"] == 1.0
    mgr.scrub_worker_stop()
    assert sorted(os.listdir(tmp_path)) == ["scrub_info.dev0", "scrub_info.dev1", "scrub_info.dev2"]


# ------------------------------------------------------------------ the resync workers' variables (resync.rs:58-71,136-166)
def test_resync_worker_count_and_persisted_config(tmp_path):
    """`resync-worker-count` (1..MAX_RESYNC_WORKERS) and `resync-tranquility`, kept in `resync_cfg` over restarts
    (ResyncPersistedConfig); several workers share one queue without ever holding the same block (the busy set)."""
    cfg = str(tmp_path / "resync_cfg")
    codec = g.ReedSolomon(10, 4, backend="cpu")
    mgr = bn.NativeBlockManager(codec, 16)
    assert mgr.resync_workers == 1
    for n in (0, 9, -3):
        with pytest.raises(bn.BlockError, match="Invalid number of resync workers, must be between 1 and 8"):
            mgr.set_resync_workers(n)
    mgr.resync_config_persist(cfg)                       # no record yet: ResyncPersistedConfig::default is written
    assert mgr.get_tranquility()[1] == 2 and mgr.resync_workers == 1 and os.path.getsize(cfg) == 16
    mgr.set_resync_workers(4)
    mgr.set_tranquility(resync=5)
    mgr2 = bn.NativeBlockManager(codec, 16)
    mgr2.set_tranquility(resync=9)
    mgr2.resync_config_persist(cfg)                      # the persisted values win
    assert mgr2.resync_workers == 4 and mgr2.get_tranquility()[1] == 5
    mgr2.set_tranquility(resync=0)
    mgr3 = bn.NativeBlockManager(codec, 16)
    mgr3.resync_config_persist(cfg)
    assert mgr3.resync_workers == 4 and mgr3.get_tranquility()[1] == 0
    with open(cfg, "wb") as f:
        f.write(b"garbage")
    mgr4 = bn.NativeBlockManager(codec, 16)
    mgr4.resync_config_persist(cfg)                      # does not decode: defaults, and the file is made good again
    assert mgr4.resync_workers == 1 and mgr4.get_tranquility()[1] == 2 and os.path.getsize(cfg) == 16

    # four workers over one queue: 120 blocks each lost a shard on one node and one on another; every block is
    # resynced exactly once, everything is back, the workers can be re-counted while they run
    hashes, blocks = _store(mgr, 120, size=20_000, salt=300)
    for h in hashes:
        who = mgr.storage_nodes_of(h)
        mgr.node_delete_shard(who[2], h, 2)
        mgr.node_delete_shard(who[11], h, 11)
    before = mgr.block_metrics()
    mgr.set_tranquility(resync=0)
    mgr.resync_worker_start()
    for i, h in enumerate(hashes):
        mgr.put_to_resync(h, 0)
        if i == 60:
            mgr.set_resync_workers(2)                    # restarts the running workers with the new count
    _wait(lambda: mgr.block_metrics()["resync_recv_counter"] == before["resync_recv_counter"] + 240, "the workers to rebuild every shard")
    assert mgr.resync_queue_len() == 120                 # what is left are block_incref's presence checks, 2 x rpc_timeout away
    mgr.resync_worker_stop()
    after = mgr.block_metrics()
    assert after["resync_counter"] == before["resync_counter"] + 120 and after["resync_error_counter"] == before["resync_error_counter"]
    assert mgr.scrub(hashes) == [] and mgr.rpc_get_blocks(hashes, 60_000) == blocks
    assert mgr.resync_workers == 2


def test_a_new_manager_over_the_same_directories(tmp_path):
    """A daemon restart: the manager object is gone, the shard files and the workers' records stay.  A new manager over the same
    node directories reads every block (reads need no refcount), carries the scrub on from the checkpoint the old one left,
    keeps the resync variables -- and refuses to repair before the references have been counted again."""
    codec = g.ReedSolomon(10, 4, backend="cpu")
    dirs = [str(tmp_path / f"node{i}") for i in range(16)]
    state, cfg = str(tmp_path / "scrub_info"), str(tmp_path / "resync_cfg")
    mgr = bn.NativeBlockManager(codec, 16, dirs, compression_level=1)
    hashes, blocks = _store(mgr, 80, size=25_000, salt=4100)
    mgr.resync_config_persist(cfg)
    mgr.set_resync_workers(3)
    mgr.set_tranquility(scrub=100, resync=1)
    mgr.scrub_worker_start(state, batch_blocks=8, checkpoint_interval_ms=1)
    mgr.scrub_worker_command(bn.SCRUB_START)
    _wait(lambda: mgr.scrub_worker_status()["blocks_scrubbed"] >= 16, "two steps")
    mgr.close()                                         # gbm_destroy stops the workers; the scrub's record holds its position

    mgr2 = bn.NativeBlockManager(codec, 16, dirs, compression_level=1)
    assert mgr2.rpc_get_blocks(hashes, 60_000) == blocks
    mgr2.resync_config_persist(cfg)
    assert mgr2.resync_workers == 3 and mgr2.get_tranquility()[1] == 1
    mgr2.scrub_worker_start(state, batch_blocks=8)
    st = mgr2.scrub_worker_status()
    assert st["state"] == bn.SCRUB_RUNNING and st["tranquility"] == 100 and 0 < st["progress"] < 1
    mgr2.set_tranquility(scrub=0)
    _wait(lambda: mgr2.scrub_worker_status()["state"] == bn.SCRUB_FINISHED, "the carried-on pass")
    st = mgr2.scrub_worker_status()
    assert 0 < st["blocks_scrubbed"] <= 80 - 16 and st["corruptions_detected"] == 0
    # The refcount table did not survive (the reference keeps it in its metadata DB; the mirror's is in memory): until the
    # references are counted again every stored block looks unneeded, and the resync DELETES what nothing references
    # (RcEntry::Absent is deletable, rc.rs:222-228).  A repair in that state is refused; after the increfs it repairs.
    who = mgr2.storage_nodes_of(hashes[5])
    mgr2.node_delete_shard(who[7], hashes[5], 7)
    with pytest.raises(bn.BlockError, match="the refcount table is empty while blocks are stored"):
        mgr2.repair_all()
    for h in hashes:
        mgr2.block_incref(h)
    assert mgr2.repair_all() == 80
    st = mgr2.resync_run()
    assert st["rebuilt"] == 1 and st["deleted"] == 0 and mgr2.scrub(hashes) == []
    assert mgr2.rpc_get_blocks(hashes, 60_000) == blocks
