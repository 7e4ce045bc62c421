"""gec_group_* (C ABI striped decode, BASELINE config 5 / SURVEY.md section 8b+8e) on the GPU:
  * N logical ranks as threads of this process on the one visible device, exchanging through
    the loopback gec_allgather_fn of tests/c/loopback_transport.cpp (RCCL refuses two ranks
    on one device): the full flow incl. the range pack / second exchange / unpack;
  * the real RCCL transport at world size 1 (ncclCommInitRank + ncclAllGather from the
    library itself, no torch.distributed).
Bit-exact against the oracle's stripes."""
import ctypes
import os
import subprocess
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import garage_amd as g  # noqa: E402
from garage_amd.striped import StripeLayout, gather_stripes, scatter_stripes  # noqa: E402
from oracle import rs_oracle as O  # noqa: E402

DEV = "cuda:0"
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def loopback():
    so = os.path.join(HERE, "c", "libgec_loopback.so")
    if not os.path.exists(so):
        r = subprocess.run(["make", "-C", os.path.join(HERE, "c"), "libgec_loopback.so"], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
    lb = ctypes.CDLL(so)
    lb.lb_create.restype = ctypes.c_void_p
    lb.lb_create.argtypes = [ctypes.c_int]
    lb.lb_destroy.argtypes = [ctypes.c_void_p]
    lb.lb_rank_ctx.restype = ctypes.c_void_p
    lb.lb_rank_ctx.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lb.lb_all_gather_ptr.restype = ctypes.c_void_p
    return lb


def _stripes(coracle, k, m, S, nobj, seed):
    data = O.splitmix64_bytes(seed, nobj * k * S).reshape(nobj, k, S)
    return np.concatenate([data, coracle.encode_batch(k, m, data, coracle.AVX2, threads=4)], axis=1)


def _run_logical_ranks(lb, rs, world, broken, present, data_only, complete):
    """-> per-rank gathered buffers (world, nobj, slots, S) as numpy arrays"""
    layout = StripeLayout(rs.k, rs.m, world)
    handle = lb.lb_create(world)
    fn = lb.lb_all_gather_ptr()
    outs, errs = [None] * world, []

    def rank_main(r):
        try:
            with torch.cuda.stream(torch.cuda.Stream(DEV)):
                grp = g.Group(rs, r, world, transport=(fn, lb.lb_rank_ctx(handle, r)))
                local = scatter_stripes(broken, layout, r)
                local[:, [layout.slot(j) for j in layout.shards_of(r) if not present[j]]] = 0xA0 + r  # junk in erased slots
                out = grp.allgather_decode(local, present, data_only=data_only, complete=complete)
                torch.cuda.current_stream().synchronize()
                outs[r] = out.cpu().numpy()
                grp.close()
        except Exception as e:  # noqa: BLE001
            errs.append((r, e))

    ts = [threading.Thread(target=rank_main, args=(r,), daemon=True) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in ts), "a logical rank hung"
    lb.lb_destroy(handle)
    assert not errs, errs
    return outs, layout


@pytest.mark.parametrize("k,m,world,S,lost", [
    (10, 4, 2, 4160, (0, 3, 7, 9)),
    (10, 4, 4, 4160, (1, 2, 11, 13)),
    (20, 8, 8, 4096 + 64, (0, 1, 5, 9, 13, 19, 21, 27)),   # config 5's shape, short shards
    (3, 1, 8, 64, (2,)),                                    # 4 columns over 8 ranks: most ranks own an EMPTY range
    (10, 4, 3, 1984, (4, 12)),                              # world does not divide k+m nor the columns
])
def test_group_allgather_decode_logical_ranks(coracle, loopback, k, m, world, S, lost):
    nobj = 5
    full = _stripes(coracle, k, m, S, nobj, 900 + world)
    present = [j not in lost for j in range(k + m)]
    broken = torch.from_numpy(full).to(DEV)
    rs = g.ReedSolomon(k, m)
    outs, layout = _run_logical_ranks(loopback, rs, world, broken, present, data_only=False, complete=True)
    for r in range(world):
        got = gather_stripes(torch.from_numpy(outs[r]), layout).numpy()
        assert np.array_equal(got, full), f"rank {r}"


from hypothesis import HealthCheck, given, settings  # noqa: E402
from hypothesis import strategies as hst  # noqa: E402


@settings(max_examples=20, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(k=hst.integers(1, 12), m=hst.integers(1, 6), world=hst.integers(1, 8), cols4=hst.integers(1, 40),
       nobj=hst.integers(1, 4), seed=hst.integers(0, 2**31), data_only=hst.booleans())
def test_group_random_geometry(coracle, loopback, k, m, world, cols4, nobj, seed, data_only):
    """Random code, world size, shard length (any multiple of 64) and erasure pattern: the byte-range
    split, the pack / second exchange / unpack index math and the padding slots, against the oracle."""
    S = 64 * cols4
    rng = np.random.default_rng(seed)
    nlost = int(rng.integers(0, m + 1))
    lost = tuple(sorted(rng.choice(k + m, size=nlost, replace=False).tolist()))
    full = _stripes(coracle, k, m, S, nobj, seed)
    present = [j not in lost for j in range(k + m)]
    broken = torch.from_numpy(full).to(DEV)
    rs = g.ReedSolomon(k, m)
    outs, layout = _run_logical_ranks(loopback, rs, world, broken, present, data_only=data_only, complete=True)
    want_rebuilt = [j for j in lost if not (data_only and j >= k)]
    keep = [j for j in range(k + m) if present[j] or j in want_rebuilt]
    for r in range(world):
        got = gather_stripes(torch.from_numpy(outs[r]), layout).numpy()
        assert np.array_equal(got[:, keep], full[:, keep]), f"rank {r} lost {lost}"
    rs.close()


def test_group_partial_and_data_only(coracle, loopback):
    """complete=0: only the rank's own byte range of each missing shard is rebuilt;
    data_only: missing parity is left alone."""
    k, m, world, S, nobj = 10, 4, 4, 2048, 3
    lost = (2, 6, 10)
    full = _stripes(coracle, k, m, S, nobj, 77)
    present = [j not in lost for j in range(k + m)]
    broken = torch.from_numpy(full).to(DEV)
    rs = g.ReedSolomon(k, m)
    outs, layout = _run_logical_ranks(loopback, rs, world, broken, present, data_only=True, complete=False)
    for r in range(world):
        got = gather_stripes(torch.from_numpy(outs[r]), layout).numpy()
        off, ln = layout.byte_range(r, S)
        for j in (2, 6):
            assert np.array_equal(got[:, j, off:off + ln], full[:, j, off:off + ln])
        present_idx = [j for j in range(k + m) if present[j]]
        assert np.array_equal(got[:, present_idx], full[:, present_idx])
        assert not np.array_equal(got[:, 10], full[:, 10])  # parity shard 10 was not asked for


def test_group_full_size_config5_vs_oracle(coracle, loopback):
    """BASELINE config 5 geometry at full shard size (RS(20,8), 4 MiB objects, 8 ranks): the stripes are encoded by the
    CPU ORACLE (35 MB: milliseconds), eight shards are erased -- zeroed on the device, so a decode that did nothing
    cannot pass -- and every rank's group decode must give the oracle's stripes back, byte for byte."""
    k, m, world, nobj = 20, 8, 8, 6
    S = g.shard_len(k, 4 << 20)
    rs = g.ReedSolomon(k, m)
    full = _stripes(coracle, k, m, S, nobj, 2005)
    lost = (0, 1, 5, 9, 13, 19, 21, 27)
    present = [j not in lost for j in range(k + m)]
    st = torch.from_numpy(full).to(DEV)
    assert bool(rs.verify_dev(st).all())          # the kernel's own check of the oracle's parity
    st[:, list(lost)] = 0
    outs, layout = _run_logical_ranks(loopback, rs, world, st, present, data_only=False, complete=True)
    for r in (0, 3, 7):
        assert np.array_equal(gather_stripes(torch.from_numpy(outs[r]), layout).numpy(), full)


def test_group_rccl_world1(coracle):
    """The library's own RCCL path: unique id, ncclCommInitRank, ncclAllGather."""
    k, m, S, nobj = 10, 4, 4160, 3
    full = _stripes(coracle, k, m, S, nobj, 58)
    lost = (0, 3, 7, 9)
    present = [j not in lost for j in range(k + m)]
    broken = torch.from_numpy(full).to(DEV)
    broken[:, list(lost)] = 0
    rs = g.ReedSolomon(k, m)
    uid = g.Group.unique_id()
    assert len(uid) == 128 and any(uid)
    grp = g.Group(rs, 0, 1, uid)
    assert grp.slots == k + m
    layout = StripeLayout(k, m, 1)
    out = grp.allgather_decode(scatter_stripes(broken, layout, 0), present)
    torch.cuda.synchronize()
    assert np.array_equal(gather_stripes(out, layout).cpu().numpy(), full)
    grp.close()


def test_group_argument_errors(loopback):
    rs = g.ReedSolomon(10, 4)
    with pytest.raises(g.GecError):
        g.Group(rs, 2, 2, transport=(loopback.lb_all_gather_ptr(), None))      # rank out of range
    with pytest.raises(g.GecError):
        g.Group(rs, 0, 1, unique_id=b"short")
    handle = loopback.lb_create(1)
    grp = g.Group(rs, 0, 1, transport=(loopback.lb_all_gather_ptr(), loopback.lb_rank_ctx(handle, 0)))
    local = torch.zeros((2, 14, 64), dtype=torch.uint8, device=DEV)
    with pytest.raises(g.GecError):
        grp.allgather_decode(local, [1] * 9 + [0] * 5)                          # fewer than k present
    with pytest.raises(TypeError):
        grp.allgather_decode(local[:, :13], [1] * 14)                           # wrong slot count
    grp.close()
    loopback.lb_destroy(handle)


# ------------------------------------------------------------- all-to-all exchange (gec_group_alltoall_decode)
def _run_logical_ranks_a2a(lb, rs, world, broken, present, data_only, complete):
    """-> per-rank (rebuilt (nmiss, nobj, S) numpy array, bytes received)"""
    layout = StripeLayout(rs.k, rs.m, world)
    handle = lb.lb_create(world)
    lb.lb_all_to_all_ptr.restype = ctypes.c_void_p
    outs, errs = [None] * world, []

    def rank_main(r):
        try:
            with torch.cuda.stream(torch.cuda.Stream(DEV)):
                grp = g.Group(rs, r, world, transport=(lb.lb_all_gather_ptr(), lb.lb_all_to_all_ptr(), lb.lb_rank_ctx(handle, r)))
                local = scatter_stripes(broken, layout, r)
                local[:, [layout.slot(j) for j in layout.shards_of(r) if not present[j]]] = 0xB0 + r
                out = grp.alltoall_decode(local, present, data_only=data_only, complete=complete)
                torch.cuda.current_stream().synchronize()
                outs[r] = (out.cpu().numpy(), grp.bytes_exchanged())
                grp.close()
        except Exception as e:  # noqa: BLE001
            errs.append((r, e))

    ts = [threading.Thread(target=rank_main, args=(r,), daemon=True) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in ts), "a logical rank hung"
    lb.lb_destroy(handle)
    assert not errs, errs
    return outs, layout


@pytest.mark.parametrize("k,m,world,S,lost", [
    (10, 4, 2, 4160, (0, 3, 7, 9)),
    (10, 4, 3, 1984, (4, 12)),                              # world divides neither k+m nor the columns
    (20, 8, 8, 4096 + 64, (0, 1, 5, 9, 13, 19, 21, 27)),   # config 5's shape, short shards
    (3, 1, 8, 64, (2,)),                                    # most ranks own an EMPTY byte range
    (10, 4, 1, 640, (1, 13)),
])
@pytest.mark.parametrize("data_only", [False, True])
def test_group_alltoall_decode_logical_ranks(coracle, loopback, k, m, world, S, lost, data_only):
    """The all-to-all exchange gives every rank the same rebuilt shards as the all-gather path (= the oracle's
    stripes), while receiving 1/N-th of the survivor bytes."""
    nobj = 5
    full = _stripes(coracle, k, m, S, nobj, 1900 + world)
    present = [j not in lost for j in range(k + m)]
    broken = torch.from_numpy(full).to(DEV)
    rs = g.ReedSolomon(k, m)
    want = [j for j in lost if not (data_only and j >= k)]
    outs, layout = _run_logical_ranks_a2a(loopback, rs, world, broken, present, data_only, True)
    for r in range(world):
        reb, nbytes = outs[r]
        assert reb.shape == (len(want), nobj, S)
        for i, j in enumerate(want):
            assert np.array_equal(reb[i], full[:, j]), f"rank {r} shard {j}"
    if world > 1 and want:
        ag, _ = _run_logical_ranks(loopback, rs, world, broken, present, data_only=data_only, complete=True)
        # bytes over the wire: all-to-all receives ~1/world of what the all-gather moves for the survivors
        slots = layout.slots
        assert outs[0][1] < nobj * slots * S * (world - 1)
    # complete=0: only the rank's own byte range of every wanted shard
    outs, layout = _run_logical_ranks_a2a(loopback, rs, world, broken, present, data_only, False)
    for r in range(world):
        off, ln = layout.byte_range(r, S)
        for i, j in enumerate(want):
            assert np.array_equal(outs[r][0][i][:, off:off + ln], full[:, j, off:off + ln]), f"rank {r} shard {j}"


def test_group_alltoall_full_size_config5_vs_oracle_and_traffic(coracle, loopback):
    """BASELINE config 5 geometry at full shard size: the rebuilt shards equal the ORACLE's (the stripes are its encode,
    the lost shards zeroed on the device), and the traffic claim of the header: 11x fewer bytes received per rank than
    the all-gather."""
    k, m, world, nobj = 20, 8, 8, 4
    S = g.shard_len(k, 4 << 20)
    rs = g.ReedSolomon(k, m)
    full = _stripes(coracle, k, m, S, nobj, 2006)
    lost = (0, 1, 5, 9, 13, 19, 21, 27)
    present = [j not in lost for j in range(k + m)]
    st = torch.from_numpy(full).to(DEV)
    st[:, list(lost)] = 0
    outs, layout = _run_logical_ranks_a2a(loopback, rs, world, st, present, False, True)
    for r in (0, 3, 7):
        for i, j in enumerate(lost):
            assert np.array_equal(outs[r][0][i], full[:, j])
    ag, _ = _run_logical_ranks(loopback, rs, world, st, present, data_only=False, complete=True)
    a2a_bytes = outs[0][1]
    ag_bytes = nobj * layout.slots * S * (world - 1) + len(lost) * nobj * (S // 8) * (world - 1)
    assert a2a_bytes * 3 < ag_bytes, (a2a_bytes, ag_bytes)


def test_group_without_alltoall_transport_is_refused(loopback):
    rs = g.ReedSolomon(10, 4)
    handle = loopback.lb_create(1)
    grp = g.Group(rs, 0, 1, transport=(loopback.lb_all_gather_ptr(), loopback.lb_rank_ctx(handle, 0)))
    with pytest.raises(g.GecError):
        grp.alltoall_decode(torch.zeros((1, 14, 64), dtype=torch.uint8, device=DEV), [0] + [1] * 13)
    grp.close()
    loopback.lb_destroy(handle)


def test_group_rccl_world1_alltoall(coracle):
    """The library's own RCCL all-to-all (grouped ncclSend / ncclRecv) at world size 1."""
    k, m, S, nobj = 10, 4, 4160, 3
    full = _stripes(coracle, k, m, S, nobj, 59)
    lost = (0, 3, 7, 9)
    present = [j not in lost for j in range(k + m)]
    broken = torch.from_numpy(full).to(DEV)
    broken[:, list(lost)] = 0
    rs = g.ReedSolomon(k, m)
    grp = g.Group(rs, 0, 1, g.Group.unique_id())
    out = grp.alltoall_decode(scatter_stripes(broken, StripeLayout(k, m, 1), 0), present)
    torch.cuda.synchronize()
    for i, j in enumerate(lost):
        assert np.array_equal(out[i].cpu().numpy(), full[:, j])
    grp.close()


# ------------------------------------------------------------------ the peer-pointer exchange (gec_group_peer_decode)
from tests.test_group_peer import CASES as PEER_CASES, check as peer_check, run_thread_ranks  # noqa: E402


@pytest.mark.parametrize("k,m,world,S,nobj,lost,data_only,complete", PEER_CASES + [(20, 8, 8, 209728, 4, (0, 1, 5, 9, 13, 19, 21, 27), False, True)])
def test_peer_decode_logical_ranks_on_one_device(loopback, k, m, world, S, nobj, lost, data_only, complete, peer_route):
    """N logical ranks as threads on the one visible device: every rank's decode launch reads its byte range of the survivors
    straight out of the OTHER ranks' slot buffers (all pointers local here: the table, the ranges and the second step are what
    is under test), the loopback transport carries the barriers and the rebuilt ranges.  Against the oracle's stripes; the last
    case is config 5's full shard length."""
    handle = loopback.lb_create(world)
    fn = loopback.lb_all_gather_ptr()

    def transport(r):
        return (fn, None, loopback.lb_rank_ctx(handle, r))

    outs, full, layout = run_thread_ranks(lambda: g.ReedSolomon(k, m), k, m, world, S, nobj, lost, data_only, complete,
                                          to_device=lambda t: t.to(DEV), transport_factory=transport)
    loopback.lb_destroy(handle)
    peer_check(outs, full, layout, k, S, lost, data_only, complete)


@pytest.fixture(params=[0, 5], ids=["strided_when_addressable", "pointer_tables"])
def peer_route(request):
    """gec_group_peer_decode reads the peers' buffers with the strided kernel when they are within 64 GiB of each other (always on one
    device) and through pointer tables otherwise; gec_set_kernel_variant(5) is the tests' route to the latter."""
    g.set_kernel_variant(request.param)
    yield request.param
    g.set_kernel_variant(0)


@pytest.mark.parametrize("complete", [True, False])
def test_peer_decode_reuses_its_pointer_tables_and_stores_in_place(coracle, complete, peer_route):
    """Round 6: a peer decode out of the same slot buffers, with the same pattern, geometry, destination and
    stream, finds its pointer tables on the device from the call before -- the steady-state call is the launch and the barriers --
    and a group of one (or complete=0) stores the rebuilt ranges straight into the destination, no unpack pass.  The tables hold
    POINTERS: new contents in the same buffers decode to the new contents; another pattern, another destination or another batch
    size rebuilds the tables.  Every result against the oracle's stripes."""
    k, m, S, nobj = 20, 8, 4160, 6
    rs = g.ReedSolomon(k, m)
    grp = g.Group(rs, 0, 1, g.Group.unique_id())
    layout = StripeLayout(k, m, 1)
    lost_a, lost_b = (0, 1, 5, 9, 13, 19, 21, 27), (2, 3, 4, 20)

    def stripes(seed, n=nobj):
        return _stripes(coracle, k, m, S, n, seed)

    def put(buf, full, lost):
        t = torch.from_numpy(full).to(DEV)
        t[:, list(lost)] = 0xEE
        buf.copy_(scatter_stripes(t, layout, 0))

    def check(out, full, lost):
        torch.cuda.synchronize()
        for i, j in enumerate(sorted(lost)):
            assert np.array_equal(out[i].cpu().numpy(), full[:, j]), (i, j)

    local = torch.empty((nobj, layout.slots, S), dtype=torch.uint8, device=DEV)
    out = torch.zeros((len(lost_a), nobj, S), dtype=torch.uint8, device=DEV)
    pres_a = [j not in lost_a for j in range(k + m)]
    for seed in (81, 82, 83):                       # same buffers, same pattern, NEW contents: calls 2 and 3 run on cached tables
        full = stripes(seed)
        put(local, full, lost_a)
        out.fill_(0)
        grp.peer_decode(local, [local.data_ptr()], pres_a, complete=complete, out=out)
        check(out, full, lost_a)
    # another pattern (other row count, other survivors), then back
    full = stripes(84)
    put(local, full, lost_b)
    out_b = torch.zeros((len(lost_b), nobj, S), dtype=torch.uint8, device=DEV)
    grp.peer_decode(local, [local.data_ptr()], [j not in lost_b for j in range(k + m)], complete=complete, out=out_b)
    check(out_b, full, lost_b)
    put(local, full, lost_a)
    grp.peer_decode(local, [local.data_ptr()], pres_a, complete=complete, out=out)
    check(out, full, lost_a)
    # another destination, then a smaller batch out of a prefix of the same buffer
    out2 = torch.zeros_like(out)
    grp.peer_decode(local, [local.data_ptr()], pres_a, complete=complete, out=out2)
    check(out2, full, lost_a)
    small = local[:2].contiguous()
    out3 = torch.zeros((len(lost_a), 2, S), dtype=torch.uint8, device=DEV)
    grp.peer_decode(small, [small.data_ptr()], pres_a, complete=complete, out=out3)
    check(out3, full[:2], lost_a)
    # another stream: its own upload (the tables of a call are ordered on that call's stream)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        out.fill_(0)
        grp.peer_decode(local, [local.data_ptr()], pres_a, complete=complete, out=out)
    side.synchronize()
    check(out, full, lost_a)
    grp.close()


def test_alltoall_decode_past_one_gib_per_peer():
    """Found by bench.py's check of the timed batch in round 6: ncclSend / ncclRecv of 2^30 bytes or more in one call delivers
    garbage (RCCL 2.26), and a group of one sends a peer's whole share to itself -- 256 objects of config 5 are 1 073 807 360
    bytes, 250 are not.  rccl_all_to_all now goes in pieces of 512 MiB.  All 300 objects on both sides of the boundary against the
    stripes they were cut from (valid stripes built on the device: the oracle checks the encode that makes them elsewhere)."""
    import bench

    k, m, L = 20, 8, 4 << 20
    S = g.shard_len(k, L)
    rs = g.ReedSolomon(k, m)
    grp = g.Group(rs, 0, 1, g.Group.unique_id())
    lost = (0, 1, 5, 9, 13, 19, 21, 27)
    present = [j not in lost for j in range(k + m)]
    layout = StripeLayout(k, m, 1)
    dev = torch.device(DEV)
    for nobj in (250, 300):
        full = bench.striped_valid_stripes(torch, rs, dev, nobj, k, m, S, L)
        assert bool(rs.verify_dev(full).all())
        local = scatter_stripes(full.index_fill(1, torch.as_tensor(sorted(lost), device=dev), 0xEE), layout, 0)
        out = torch.zeros((len(lost), nobj, S), dtype=torch.uint8, device=dev)
        grp.alltoall_decode(local, present, out=out)
        torch.cuda.synchronize()
        assert 20 * nobj * S > (1 << 30) or nobj == 250
        for i, j in enumerate(sorted(lost)):
            assert torch.equal(out[i], full[:, j]), (nobj, j)
        del full, local, out
    grp.close()


def test_peer_decode_two_processes_through_ipc_handles(tmp_path):
    """Two PROCESSES on the one device: each exports its slot buffer (gec_ipc_export = hipIpcGetMemHandle), opens the other's
    (gec_ipc_open), and decodes out of it; the barriers and rebuilt ranges go over gloo through a host-staging transport.
    What the Rust host does across the ranks of a node."""
    code = r'''
import ctypes, os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %r)
import garage_amd as g
from garage_amd import _lib
from garage_amd.group import ipc_export, ipc_open, ipc_close
from garage_amd.striped import StripeLayout, scatter_stripes
from oracle import rs_oracle as O
rank, world = int(os.environ["RANK"]), 2
dist.init_process_group("gloo", rank=rank, world_size=world)
k, m, S, nobj = 10, 4, 4160, 6
lost = (0, 3, 7, 11)
data = O.splitmix64_bytes(321, nobj * k * S).reshape(nobj, k, S)
full = np.concatenate([data, np.stack([O.encode(k, m, d) for d in data])], axis=1)
broken = full.copy(); broken[:, list(lost)] = 0xEE
layout = StripeLayout(k, m, world)
local = scatter_stripes(torch.from_numpy(broken), layout, rank).contiguous().to("cuda:0")
torch.cuda.synchronize()
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
hip.hipStreamSynchronize.argtypes = [ctypes.c_void_p]
@_lib.ALLGATHER_FN
def all_gather(_ctx, send, recv, nbytes, stream):
    hip.hipStreamSynchronize(stream)
    mine = np.empty(nbytes, dtype=np.uint8)
    hip.hipMemcpy(mine.ctypes.data, send, nbytes, 2)
    got = np.empty(nbytes * world, dtype=np.uint8)
    dist.all_gather(list(torch.from_numpy(got).chunk(world)), torch.from_numpy(mine))
    hip.hipMemcpy(recv, got.ctypes.data, got.size, 1)
    return 0
handles = [None, None]
dist.all_gather_object(handles, ipc_export(local))
peer = ipc_open(handles[1 - rank], 0)
ptrs = [0, 0]
ptrs[1 - rank] = peer
rs = g.ReedSolomon(k, m)
grp = g.Group(rs, rank, world, transport=(all_gather, None, None))
reb = grp.peer_decode(local, ptrs, [j not in lost for j in range(k + m)])
torch.cuda.synchronize()
got = reb.cpu().numpy()
ok = all(np.array_equal(got[i], full[:, j]) for i, j in enumerate(sorted(lost)))
moved = grp.bytes_exchanged()
grp.close()
dist.barrier()
ipc_close(peer)
flags = [None, None]
dist.all_gather_object(flags, (bool(ok), int(moved)))
if rank == 0:
    print("RESULT", flags)
dist.destroy_process_group()
''' % os.path.dirname(HERE)
    import socket
    import sys

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, "-c", code], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(2)]
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-2000:] for o in outs]
    line = [ln for ln in outs[0][0].splitlines() if ln.startswith("RESULT")][0]
    flags = eval(line[len("RESULT "):])
    assert all(ok for ok, _ in flags), flags
    # each rank read its half of the 5 valid shards the other rank holds, then received the other's rebuilt ranges
    assert all(moved > 0 for _, moved in flags), flags
