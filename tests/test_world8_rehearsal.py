"""World-8 rehearsal: what the driver's 8-GPU pass will run, on ONE GPU (GARAGE_DRYRUN_ONE_GPU=1: every
rank / codec on device 0, gloo instead of RCCL -- labelled as such in the line; numbers meaningless, control flow identical).
tools/world8_rehearsal.py runs the whole matrix (N = 2, 4, 8; four invocations each) and records wall times in
profiles/r06_world8_rehearsal.txt; these tests assert the world-8 column and the fault injections in about two minutes."""
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")
DRY = {"GARAGE_DRYRUN_ONE_GPU": "1"}
DRIVER_FLAGS = ["--steps", "20", "--warmup", "5"]   # what the driver passes (BENCH_r04.json: cmd)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(cmd, env=None, timeout=300):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        e.pop(k, None)
    e.update(env or {})
    t0 = time.time()
    r = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=timeout)
    wall = time.time() - t0
    assert r.returncode == 0, f"{' '.join(cmd)}\n{r.stdout[-2000:]}\n{r.stderr[-3000:]}"
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, f"expected ONE JSON line on stdout, got {len(lines)}:\n{r.stdout[-3000:]}"
    assert wall < 300, f"{wall:.0f} s: the rehearsal's budget per invocation is 300 s (the driver's limit is 1800 s)"
    return json.loads(lines[0])


def _torchrun(n, *args):
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
            "--master-port", str(_free_port()), BENCH, "--gpus", str(n), *args]


def _expected_partition(world: int, batch: int = 1024):
    """hash[4] % world on Garage-style hashes of the synthetic stream -- computed here with hashlib, not with the product's
    gec_device_of_hash (src/rpc/layout/version.rs:101-104 is the reference's analogue: placement by hash bytes)"""
    import hashlib
    import struct

    owners = [hashlib.blake2b(struct.pack("<QQ", 0x6761726167650004, i), digest_size=64).digest()[4] % world for i in range(batch * world)]
    return np.bincount(np.array(owners), minlength=world).tolist()


def _check_encode_line(d, world):
    assert d["n_gpus"] == world and d["scaling"] == "weak" and d["metric"].startswith("RS(10,4) encode") and d["unit"] == "GiB/s"
    per = d["config"]["blocks_per_rank"]
    assert per == _expected_partition(world) and sum(per) == 1024 * world == d["config"]["blocks_total"]
    assert d["parity_checked_blocks"] == 1024 * world                       # every block of every rank against the CPU oracle
    fr = d["roofline_frac_per_gpu"]
    assert len(fr) == world and all(f is not None and 0 < f < 1 for f in fr)
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["algorithmic_bytes_per_launch"] == 14 * 104896 * per[0]
    hf = d["host_fed"]
    assert hf["n_gpus"] == world and len(hf["numa_node_per_gpu"]) == world      # (round 6: each lane's memory node, -1 = not placed)
    for kind in ("pinned", "pageable"):
        assert len(hf[kind]["per_gpu_GiBps"]) == world and hf[kind]["bit_exact_vs_oracle"] is True
    mm = hf["block_manager_multi"]                                          # (d) the product's multi-device manager over `world` codecs
    assert "error" not in mm and mm["n_devices"] == world and mm["routing_follows_gec_device_of_hash"] is True and mm["every_byte_compared"] is True


@pytest.mark.gpu
def test_world8_procs_as_the_driver_launches_it():
    d = _run(_torchrun(8, *DRIVER_FLAGS), DRY)
    _check_encode_line(d, 8)
    assert d["rccl_ranks"] == 8 and d["collective_backend"] == "gloo"       # (the dry run's stand-in, named in the line)
    sd = d["striped_decode"]
    assert sd["bit_exact"] is True and "oracle" in sd["bit_exact_against"] and sd["exchange"]["alltoall"]["bit_exact"] is True
    assert sd["rccl_ranks"] == 8 and "DRY RUN" in sd["collective_backend"] and "gec_group_allgather_decode" in sd["config"]["collective"]
    assert "256 x 4 MiB" in sd["config"]["workload"]                         # BASELINE config 5 at full size


@pytest.mark.gpu
def test_world8_threads_mode():
    d = _run([sys.executable, BENCH, "--gpus", "8", "--mode", "threads", *DRIVER_FLAGS], DRY)
    _check_encode_line(d, 8)
    assert d["rccl_ranks"] is None and "threads" in d["config"]["mode"] and len(d["kernel_ms_per_gpu"]) == 8
    sd = d["striped_decode"]
    assert sd["bit_exact"] is True and sd["ranks"] == 8 and "oracle" in sd["bit_exact_against"]


@pytest.mark.gpu
def test_world8_striped_decode_op_at_full_config5_size():
    d = _run(_torchrun(8, "--op", "striped-decode", *DRIVER_FLAGS), DRY)
    assert d["n_gpus"] == 8 and d["bit_exact"] is True and "oracle" in d["bit_exact_against"]
    # every object of the TIMED batch, on every rank, after each of the three exchanges' timed loops
    assert d["bit_exact_objects"] == 256 == d["timed_batch_objects"] and d["timed_batch"].endswith(": ok")
    for name in ("allgather", "alltoall", "peer"):
        ex = d["exchange"][name]
        assert ex["bit_exact"] is True and ex["bit_exact_objects"] == 256 and ex["verify_batch_dev"] is True and ex["oracle_sample_objects"] >= 4, (name, ex)
        assert ex["small_batch_bit_exact"] is True
    assert d["rccl_ranks"] == 8 and "DRY RUN" in d["collective_backend"]
    cfg = d["config"]
    assert (cfg["k"], cfg["m"], cfg["shard_len"], cfg["slots_per_rank"]) == (20, 8, 209728, 4) and "256 x 4 MiB" in cfg["workload"]
    # all-gather: 7 peers' slot buffers; all-to-all: only this rank's byte range of the k valid shards -- an order less
    # (plus the second step both share: the 7 other ranks' rebuilt ranges of the 8 missing shards, padded to the longest range)
    second = 7 * 8 * 256 * (-(-(209728 // 16) // 8)) * 16
    assert d["exchange"]["allgather"]["bytes_received_per_rank"] == 7 * 256 * 4 * 209728 + second
    assert (d["exchange"]["alltoall"]["bytes_received_per_rank"] - second) * 7 < d["exchange"]["allgather"]["bytes_received_per_rank"] - second   # 1/8 of it (slots padded alike)
    # the peer-pointer form across 8 PROCESSES: every rank maps the 7 others' slot buffers (HIP IPC) and its decode launch reads
    # its byte range of the 20 survivors out of them -- 1/8 of 17-18 remote shards per object instead of 7 ranks' whole slot buffers
    peer = d["exchange"]["peer"]
    assert peer["bit_exact"] is True and 0 < peer["bytes_read_from_peers_memory_per_rank"] * 10 < d["exchange"]["allgather"]["bytes_received_per_rank"]


@pytest.mark.gpu
@pytest.mark.parametrize("exchange", ["allgather", "alltoall", "peer"])
def test_one_flipped_byte_in_one_ranks_slot_buffer_turns_the_line_red(exchange):
    """Negative test of the timed batch's check: GARAGE_BENCH_STRIPED_FLIP_RANK flips ONE byte of one surviving shard in one
    rank's slot buffer between the warm-up and the timed loop of one exchange.  That exchange must report bit_exact false with
    255 of 256 objects intact, the line carries `error`, and the two other exchanges (same buffers, byte restored) stay green --
    i.e. a cross-device visibility bug on real hardware cannot print a fast, wrong, green line."""
    d = _run(_torchrun(4, "--op", "striped-decode", "--striped-objects", "64", "--steps", "20"),
             dict(DRY, GARAGE_BENCH_STRIPED_FLIP_RANK="2", GARAGE_BENCH_STRIPED_FLIP_EXCHANGE=exchange))
    assert "timed batch" in d["error"]
    for name in ("allgather", "alltoall", "peer"):
        ex = d["exchange"][name]
        if name == exchange:
            assert ex["bit_exact"] is False and ex["bit_exact_objects"] == 63 and ex["verify_batch_dev"] is False, ex
        else:
            assert ex["bit_exact"] is True and ex["bit_exact_objects"] == 64, (name, ex)
    assert d["bit_exact"] is (exchange != "allgather")


@pytest.mark.gpu
def test_headline_survives_an_rccl_that_cannot_be_loaded():
    d = _run([sys.executable, BENCH, "--gpus", "1", *DRIVER_FLAGS, "--striped", "--no-host-path", "--no-cpu-baseline"],
             {"GEC_RCCL_LIB": "/nonexistent/librccl.so"})
    assert d["value"] > 0 and d["roofline"]["frac"] > 0
    assert "RCCL is not available" in d["striped_decode"]["error"]


@pytest.mark.gpu
def test_headline_survives_a_rank_that_never_reaches_the_collective():
    d = _run(_torchrun(2, *DRIVER_FLAGS, "--striped-timeout", "8", "--no-host-fed"), dict(DRY, GARAGE_BENCH_STRIPED_HANG_RANK="1"))
    assert d["value"] > 0 and d["n_gpus"] == 2 and len(d["roofline_frac_per_gpu"]) == 2
    assert "watchdog" in d["striped_decode"]["error"]


@pytest.mark.gpu
def test_a_rank_that_cannot_map_its_peers_costs_only_the_peer_form():
    """The peer-pointer exchange needs HIP IPC mappings between the ranks -- the one step of the striped decode that has only ever
    run on one device.  A rank that fails there says so inside the exchange of handles, every rank drops the peer form together,
    and the all-gather / all-to-all figures and their oracle checks are reported as usual (no watchdog, no hang)."""
    d = _run(_torchrun(2, "--op", "striped-decode", "--striped-objects", "16", "--steps", "20"), dict(DRY, GARAGE_BENCH_PEER_FAIL_RANK="1"))
    assert d["bit_exact"] is True and d["exchange"]["alltoall"]["bit_exact"] is True
    assert "could not map" in d["exchange"]["peer"]["error"], d["exchange"]["peer"]


def test_the_recorded_rehearsal_is_complete():
    """profiles/r06_world8_rehearsal.txt (tools/world8_rehearsal.py on one MI355X): N = 2, 4, 8 x four invocations, the N = 1
    line, the two fault injections that must leave the headline intact and the three that must turn one exchange red -- every one
    rc 0, one JSON line, within its 300 s; every striped decode of the matrix with 256 of 256 objects of its TIMED batch exact
    after each of the three exchanges."""
    path = os.path.join(ROOT, "profiles", "r06_world8_rehearsal.txt")
    rows = [json.loads(ln) for ln in open(path) if ln.startswith("{")]
    assert len(rows) == 18 and all(r["ok"] and r["rc"] == 0 and r["json_lines"] == 1 and r["wall_s"] < 300 for r in rows)
    for n in (2, 4, 8):
        kinds = [r["what"][:3] for r in rows if r["n"] == n and r["what"].startswith("(")]
        assert sorted(kinds) == ["(a'", "(a)", "(b)", "(c)"], (n, kinds)
        for r in rows:
            if r["n"] == n and r["what"].startswith(("(a", "(b")):
                assert r["summary"]["blocks_per_rank"] == _expected_partition(n)
                assert r["summary"]["striped_decode"]["bit_exact"] is True and len(r["summary"]["roofline_frac_per_gpu"]) == n
            if r["n"] == n and r["what"].startswith(("(a", "(c")):
                sd = r["summary"]["striped_decode"]
                assert sd["timed_batch_objects"] == 256 and sd["bit_exact_objects_per_exchange"] == {"allgather": 256, "alltoall": 256, "peer": 256}
    flips = [r for r in rows if "flipped" in r["what"]]
    assert len(flips) == 3
    for r, hit in zip(flips, ("allgather", "alltoall", "peer")):
        per = r["summary"]["striped_decode"]["bit_exact_objects_per_exchange"]
        assert hit in r["what"] and per[hit] == 63 and sorted(per.values()) == [63, 64, 64]
