"""N>1 paths on CPU: world_size-2 (and 3) `gloo` process groups exercise the striped-object decode -- the torch
mirror's exchange logic (garage_amd/striped.py) AND the library's own (gec_group_* through the C ABI over a gloo-backed
transport, tests/test_group_multiprocess.py) -- and the hash partition.  The arithmetic is the PRODUCT's: a
GEC_BACKEND_CPU codec, whose strided reconstruct runs on host memory; the oracle only encodes the stripes and judges the
result.  The GPU version of the same flow is tests/test_gpu_striped.py."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from garage_amd.partition import block_hash, gpu_of_hash, partition
from garage_amd.striped import (StripeLayout, gather_stripes, scatter_stripes, striped_reconstruct,
                                striped_reconstruct_alltoall)
from oracle import rs_oracle as O


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, k, m, S, nobj, lost, data_only, complete, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import garage_amd as g

        codec = g.ReedSolomon(k, m, backend="cpu")
        layout = StripeLayout(k, m, world)
        data = O.splitmix64_bytes(77, nobj * k * S).reshape(nobj, k, S)
        full = np.concatenate([data, np.stack([O.encode(k, m, d) for d in data])], axis=1)
        present = [j not in lost for j in range(k + m)]
        broken = full.copy()
        broken[:, list(lost)] = 0xEE
        mine = scatter_stripes(torch.from_numpy(broken), layout, rank)
        out = striped_reconstruct(codec, mine, present, layout, data_only=data_only, complete=complete)
        got = gather_stripes(out, layout).numpy()
        want = full.copy()
        if data_only:
            for j in lost:
                if j >= k:
                    want[:, j] = 0xEE
        if complete:
            ok = np.array_equal(got, want)
        else:
            off, ln = layout.byte_range(rank, S)
            ok = np.array_equal(got[:, :, off:off + ln], want[:, :, off:off + ln])
            present_idx = [j for j in range(k + m) if present[j]]
            ok = ok and np.array_equal(got[:, present_idx], full[:, present_idx])
        q.put((rank, bool(ok), ""))
        dist.destroy_process_group()
    except Exception as e:  # pragma: no cover
        import traceback

        q.put((rank, False, traceback.format_exc()))


def _worker_a2a(rank, world, port, k, m, S, nobj, lost, data_only, complete, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import garage_amd as g

        codec = g.ReedSolomon(k, m, backend="cpu")
        layout = StripeLayout(k, m, world)
        data = O.splitmix64_bytes(78, nobj * k * S).reshape(nobj, k, S)
        full = np.concatenate([data, np.stack([O.encode(k, m, d) for d in data])], axis=1)
        present = [j not in lost for j in range(k + m)]
        broken = full.copy()
        broken[:, list(lost)] = 0xEE
        mine = scatter_stripes(torch.from_numpy(broken), layout, rank)
        reb = striped_reconstruct_alltoall(codec, mine, present, layout, data_only=data_only, complete=complete).numpy()
        wanted = [j for j in lost if not (data_only and j >= k)]
        ok = reb.shape == (len(wanted), nobj, S)
        off, ln = layout.byte_range(rank, S)
        for i, j in enumerate(wanted):
            if complete:
                ok = ok and np.array_equal(reb[i], full[:, j])
            else:
                ok = ok and np.array_equal(reb[i][:, off:off + ln], full[:, j, off:off + ln])
        q.put((rank, bool(ok), ""))
        dist.destroy_process_group()
    except Exception:  # pragma: no cover
        import traceback

        q.put((rank, False, traceback.format_exc()))


def _run_a2a(world, **kw):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_a2a, args=(r, world, port, kw["k"], kw["m"], kw["S"], kw["nobj"], kw["lost"],
                                                   kw.get("data_only", False), kw.get("complete", True), q))
             for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=30)
    for rank, ok, err in res:
        assert ok, f"rank {rank} failed: {err}"


@pytest.mark.timeout(180)
def test_striped_alltoall_world2_and_world3():
    """The all-to-all exchange (each rank receives only its byte range of the k valid shards) over gloo."""
    _run_a2a(2, k=20, m=8, S=1088, nobj=3, lost=(0, 1, 5, 9, 13, 19, 21, 27))
    _run_a2a(3, k=10, m=4, S=832, nobj=2, lost=(1, 4, 13))             # ragged ranges, padded slots, surplus survivors
    _run_a2a(2, k=10, m=4, S=192, nobj=2, lost=(0, 3, 7, 11), complete=False)
    _run_a2a(2, k=10, m=4, S=192, nobj=2, lost=(2, 12), data_only=True)


def _run(world, **kw):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, kw["k"], kw["m"], kw["S"], kw["nobj"], kw["lost"],
                                               kw.get("data_only", False), kw.get("complete", True), q))
             for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=30)
    for rank, ok, err in res:
        assert ok, f"rank {rank} failed: {err}"


@pytest.mark.timeout(180)
def test_striped_decode_world2_rs_20_8():
    # config 5 code, scaled-down shard length; 8 erasures (6 data + 2 parity)
    _run(2, k=20, m=8, S=1088, nobj=3, lost=(0, 1, 5, 9, 13, 19, 21, 27))


@pytest.mark.timeout(180)
def test_striped_decode_world2_partial_and_data_only():
    _run(2, k=10, m=4, S=192, nobj=2, lost=(0, 3, 7, 11), complete=False)
    _run(2, k=10, m=4, S=192, nobj=2, lost=(2, 12), data_only=True)


@pytest.mark.timeout(180)
def test_striped_decode_world3_ragged_ranges():
    # 3 ranks: n=14 shards -> 5 slots with padding; S/16 = 13 columns does not divide by 3
    _run(3, k=10, m=4, S=832, nobj=2, lost=(1, 4, 13))


def test_layout_geometry():
    lay = StripeLayout(20, 8, 8)
    assert lay.slots == 4 and lay.shards_of(0) == [0, 8, 16, 24] and lay.shards_of(7) == [7, 15, 23]
    S = 209728
    ranges = [lay.byte_range(r, S) for r in range(8)]
    assert sum(ln for _, ln in ranges) == S and all(off % 16 == 0 and ln % 16 == 0 for off, ln in ranges)
    assert ranges[0][0] == 0 and all(ranges[i][0] + ranges[i][1] == ranges[i + 1][0] for i in range(7))
    offs = lay.shard_offsets(256, S)
    assert offs[0] == 0 and offs[8] == S and offs[1] == 256 * 4 * S and len(set(offs)) == 28
    lay3 = StripeLayout(10, 4, 3)
    assert lay3.slots == 5 and [len(lay3.shards_of(r)) for r in range(3)] == [5, 5, 4]


def test_hash_partition_properties():
    hashes = np.frombuffer(b"".join(block_hash(i.to_bytes(8, "little")) for i in range(4096)), dtype=np.uint8).reshape(-1, 32)
    assert block_hash(b"") == __import__("hashlib").blake2b(b"", digest_size=64).digest()[:32]
    for n in (1, 2, 4, 8):
        parts = partition(hashes, n)
        allidx = np.sort(np.concatenate(parts))
        assert np.array_equal(allidx, np.arange(4096)), "every block on exactly one GPU"
        assert all(np.all(np.diff(p) > 0) for p in parts if len(p) > 1), "stream order kept"
        counts = np.array([len(p) for p in parts])
        assert counts.max() <= 4096 / n * 1.15, counts
        assert gpu_of_hash(bytes(hashes[5]), n) == gpu_of_hash(hashes, n)[5]
    # the GPU byte is independent of the cluster-partition byte (0) and drive bytes (2,3)
    h = hashes.copy()
    h[:, [0, 2, 3]] ^= 0xFF
    assert np.array_equal(gpu_of_hash(h, 8), gpu_of_hash(hashes, 8))


# ------------------------------------------- bench.py's multi-rank plumbing
def _dist_worker(rank, world, port, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                          WORLD_SIZE=str(world))
        from garage_amd import distrib

        R = distrib.init_from_env(force_backend="gloo")
        assert (R.rank, R.world, R.distributed) == (rank, world, True)
        distrib.barrier(R)
        mx = distrib.max_over_ranks(R, 1.5 + rank)
        hashes = np.frombuffer(b"".join(block_hash(i.to_bytes(8, "little")) for i in range(512 * world)),
                               dtype=np.uint8).reshape(-1, 32)
        mine = int((gpu_of_hash(hashes, world) == rank).sum())
        total = distrib.sum_over_ranks(R, mine)
        distrib.shutdown(R)
        q.put((rank, mx == 1.5 + world - 1 and total == 512 * world and mine > 0, f"mx={mx} total={total}"))
    except Exception:  # pragma: no cover
        import traceback

        q.put((rank, False, traceback.format_exc()))


@pytest.mark.timeout(180)
def test_bench_rank_plumbing_world2():
    """The barrier / max-over-ranks / sum-of-units contract bench.py uses at N>1."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dist_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=30)
    for rank, ok, msg in res:
        assert ok, f"rank {rank}: {msg}"
