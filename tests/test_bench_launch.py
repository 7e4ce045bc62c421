"""bench.py must start however it is launched: under
torch.distributed.run (what the driver does), plainly with --gpus N (it re-launches itself),
and as one process with N codecs / N host threads (--mode threads).  On CPU only the launch
plumbing runs (--launch-check: rank discovery, process group over gloo, one reduction); on a GPU
box the real encode bench runs in both modes with every rank / codec on device 0
(GARAGE_DRYRUN_ONE_GPU=1 -- numbers meaningless, control flow identical)."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(cmd, env=None, timeout=600):
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None)
    e.pop("RANK", None)
    e.pop("LOCAL_RANK", None)
    e.update(env or {})
    r = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, f"{' '.join(cmd)}\n{r.stdout}\n{r.stderr[-3000:]}"
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, f"expected ONE JSON line on stdout, got {len(lines)}:\n{r.stdout}"
    return json.loads(lines[0])


def test_plain_invocation_self_launches_n_ranks():
    d = _run([sys.executable, BENCH, "--gpus", "2", "--launch-check"])
    assert d == {"launch_check": True, "mode": "procs", "world": 2, "ranks_seen": 2, "backend": "gloo",
                 "ranks": [0, 1], "self_launched": True}


def test_under_torch_distributed_run():
    d = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3",
              "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), BENCH, "--gpus", "3", "--launch-check"])
    assert d["world"] == 3 and d["ranks_seen"] == 3 and d["ranks"] == [0, 1, 2] and d["self_launched"] is False


def test_threads_mode_launch_check():
    d = _run([sys.executable, BENCH, "--gpus", "4", "--mode", "threads", "--launch-check"])
    assert d["mode"] == "threads" and d["world"] == 4 and d["ranks_seen"] == 4


def test_world_size_mismatch_is_an_error():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launch-check"], cwd=ROOT, capture_output=True, text=True,
                       env=dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert r.returncode != 0 and "does not match --gpus" in (r.stderr + r.stdout)


def test_a_ranks_child_group_gets_a_rendezvous_of_its_own():
    """The striped decode runs in a child process per rank with its own process group (so that nothing in there can cost the encode
    line).  Under torch.distributed.run the parent's environment says "the agent hosts the store": that must not reach the child."""
    sys.path.insert(0, ROOT)
    import bench

    parent = {"TORCHELASTIC_USE_AGENT_STORE": "True", "TORCHELASTIC_RUN_ID": "x", "GROUP_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29500",
              "RANK": "3", "LOCAL_RANK": "3", "WORLD_SIZE": "8", "LOCAL_WORLD_SIZE": "8", "PATH": "/bin", "GARAGE_DRYRUN_ONE_GPU": "1"}
    env = bench.child_group_env(parent, 3, 8, 29523)
    assert not [k for k in env if k.startswith("TORCHELASTIC_")] and "GROUP_RANK" not in env
    assert (env["RANK"], env["LOCAL_RANK"], env["WORLD_SIZE"], env["MASTER_PORT"], env["GARAGE_DRYRUN_ONE_GPU"]) == ("3", "3", "8", "29523", "1")


SMALL = ["--steps", "5", "--warmup", "2", "--precondition-ms", "0", "--batch", "64", "--no-cpu-baseline"]


@pytest.mark.gpu
@pytest.mark.parametrize("launcher", ["self", "torchrun"])
def test_gpu_procs_mode_two_ranks_on_one_gpu(launcher):
    if launcher == "self":
        cmd = [sys.executable, BENCH, "--gpus", "2"] + SMALL
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
               "127.0.0.1", "--master-port", str(_free_port()), BENCH, "--gpus", "2"] + SMALL
    d = _run(cmd, env={"GARAGE_DRYRUN_ONE_GPU": "1"})
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["scaling"] == "weak"
    per = d["config"]["blocks_per_rank"]
    assert len(per) == 2 and sum(per) == 128 == d["config"]["blocks_total"]
    assert d["parity_checked_blocks"] >= 32 and d["roofline"]["frac"] > 0 and "decode" in d
    assert len(d["roofline_frac_per_gpu"]) == 2 and all(0 < f < 1 for f in d["roofline_frac_per_gpu"])
    # every GPU fed from host memory at once (both ranks push their blocks through gec_encode_hash_batch), oracle-checked
    hf = d["host_fed"]
    assert hf.get("n_gpus") == 2, hf
    for kind in ("pinned", "pageable"):
        assert len(hf[kind]["per_gpu_GiBps"]) == 2 and hf[kind]["aggregate_GiBps"] > 0
        assert hf[kind]["bit_exact_vs_oracle"] is True and hf[kind]["blocks_checked"] >= 4
        assert hf[kind]["host_memory_traffic_GBps_model"] > hf[kind]["aggregate_GiBps"]
    # the same node through the product's multi-device manager, from one process (rank 0 runs tools/multi_bench)
    mm = hf["block_manager_multi"]
    assert "error" not in mm, mm
    assert mm["n_devices"] == 2 and mm["routing_follows_gec_device_of_hash"] is True and mm["every_byte_compared"] is True


@pytest.mark.gpu
def test_gpu_threads_mode_two_codecs_on_one_gpu():
    d = _run([sys.executable, BENCH, "--gpus", "2", "--mode", "threads"] + SMALL, env={"GARAGE_DRYRUN_ONE_GPU": "1"})
    assert d["n_gpus"] == 2 and d["rccl_ranks"] is None and "threads" in d["config"]["mode"]
    per = d["config"]["blocks_per_rank"]
    assert len(per) == 2 and sum(per) == 128 and len(d["kernel_ms_per_gpu"]) == 2
    assert d["parity_checked_blocks"] >= 32
    # two host threads, two codecs: the host-fed path, pinned and pageable, oracle-checked
    hf = d["host_fed"]
    assert "error" not in hf, hf
    for kind in ("pinned", "pageable"):
        assert len(hf[kind]["per_gpu_GiBps"]) == 2 and hf[kind]["bit_exact_vs_oracle"] is True
    # ... and through the PRODUCT's multi-device manager (gbm_create_multi, one coalescing queue per device, native callers)
    mm = hf["block_manager_multi"]
    assert "error" not in mm, mm
    assert mm["n_devices"] == 2 and mm["routing_follows_gec_device_of_hash"] is True and mm["every_byte_compared"] is True
    assert len(mm["per_device"]) == 2 and all(dv["blocks_put"] > 0 and dv["blocks_get"] == dv["blocks_put"] for dv in mm["per_device"])
    assert mm["put_GiBps"] > 0 and mm["get_GiBps"] > 0
    # a one-process gec_group over the (logical) ranks: the striped decode against the ORACLE's stripes, both exchanges
    sd = d["striped_decode"]
    assert sd.get("bit_exact") is True and sd["ranks"] == 2 and "oracle" in sd["bit_exact_against"], sd


@pytest.mark.gpu
def test_gpu_striped_decode_is_checked_against_the_oracle():
    """--op striped-decode at world 1 (RCCL with one rank): the rebuilt shards are compared with stripes the CPU oracle
    encoded, not with the GPU's own encode."""
    d = _run([sys.executable, BENCH, "--op", "striped-decode", "--steps", "100", "--striped-objects", "32"])
    assert d["bit_exact"] is True and "oracle" in d["bit_exact_against"]
    assert d["exchange"]["alltoall"]["bit_exact"] is True


@pytest.mark.gpu
def test_gpu_n1_modes_agree():
    """N=1: the two modes run the same kernel on the same batch; their values agree within noise
    (both short runs, so the bound is loose) and both carry the contract objects."""
    a = _run([sys.executable, BENCH, "--steps", "200", "--warmup", "20", "--no-cpu-baseline", "--no-host-path"])
    b = _run([sys.executable, BENCH, "--steps", "200", "--warmup", "20", "--no-cpu-baseline", "--mode", "threads"])
    assert a["config"]["blocks_total"] == b["config"]["blocks_total"] == 1024
    assert abs(a["value"] - b["value"]) / a["value"] < 0.10, (a["value"], b["value"])
    for d in (a, b):
        assert d["roofline"]["bound"] == "hbm" and 0.3 < d["roofline"]["frac"] < 1.0
        assert d["roofline"]["cold_burst_frac"] > 0
    # round 5: the put trip (config 2 + the checksum of every shard) and config 5's code ride in the driver-run line, each checked
    # against its oracles after its own timed loop
    eh, r2 = a["encode_hash"], a["rs20_8_encode"]
    assert eh["bit_exact"] is True and eh["checksums_per_launch"] == 1024 * 14 and eh["roofline"]["algorithmic_bytes_per_launch"] == 1503789056
    assert eh["ms"] < 2.0 * eh["encode_alone_ms"] and 0.3 < eh["roofline"]["frac"] < a["roofline"]["frac"] + 0.05
    assert eh["binding_resource"]["valu_busy_frac"] > 0 and eh["traffic"] >= eh["roofline"]["algorithmic_bytes_per_launch"]
    assert r2["bit_exact"] is True and r2["blocks"] == 256 and r2["shard_len"] == 209728 and 0.3 < r2["roofline"]["frac"] < 1.0
    assert r2["roofline"]["traffic"] >= r2["roofline"]["algorithmic_bytes_per_launch"] == 28 * 209728 * 256
