"""The LIBRARY's own striped-object exchange code (garage_amd/csrc/ec_hip_group.cpp: range split, packs and unpacks, the
rebuild of a rank's byte range) with N > 1 REAL processes, on a box without a GPU: gec_group_allgather_decode and
gec_group_alltoall_decode through the C ABI, over a GEC_BACKEND_CPU codec and a gloo-backed caller transport
(gec_group_create_with_transport2), checked against stripes the ORACLE encoded.  VERDICT r03 item 4(c).

BASELINE config 5's geometry (RS(20,8), shard j on rank j mod N) at world 2 and 3, ragged byte ranges, padded slots,
data_only and complete=0."""
import ctypes
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

from oracle import rs_oracle as O


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, cases, q):
    try:
        import torch
        import torch.distributed as dist

        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import garage_amd as g
        from garage_amd import _lib
        from garage_amd.striped import StripeLayout, gather_stripes, scatter_stripes

        def view(ptr, nbytes):
            return torch.from_numpy(np.ctypeslib.as_array((ctypes.c_uint8 * nbytes).from_address(ptr)))

        calls = {"ag": 0, "a2a": 0}

        @_lib.ALLGATHER_FN
        def all_gather(_ctx, send, recv, nbytes, _stream):   # gec_allgather_fn: host pointers, no stream
            calls["ag"] += 1
            if nbytes:
                dist.all_gather(list(view(recv, nbytes * world).chunk(world)), view(send, nbytes))
            return 0

        @_lib.ALLGATHER_FN
        def all_to_all(_ctx, send, recv, nbytes, _stream):   # gec_alltoall_fn: nbytes per peer
            calls["a2a"] += 1
            if nbytes:
                dist.all_to_all_single(view(recv, nbytes * world), view(send, nbytes * world))
            return 0

        ok, why = True, ""
        for (k, m, S, nobj, lost, data_only, complete) in cases:
            codec = g.ReedSolomon(k, m, backend="cpu")
            grp = g.Group(codec, rank, world, transport=(all_gather, all_to_all, None))
            layout = StripeLayout(k, m, world)
            data = O.splitmix64_bytes(1000 + k + S, nobj * k * S).reshape(nobj, k, S)
            full = np.concatenate([data, np.stack([O.encode(k, m, d) for d in data])], axis=1)   # the ORACLE's stripes
            present = [j not in lost for j in range(k + m)]
            broken = full.copy()
            broken[:, list(lost)] = 0xEE
            mine = scatter_stripes(torch.from_numpy(broken), layout, rank)
            want = full.copy()
            if data_only:
                for j in lost:
                    if j >= k:
                        want[:, j] = 0xEE
            off, ln = layout.byte_range(rank, S)
            # ---- all-gather decode
            got = gather_stripes(grp.allgather_decode(mine, present, data_only=data_only, complete=complete), layout).numpy()
            if complete:
                good = np.array_equal(got, want)
            else:
                good = np.array_equal(got[:, :, off:off + ln], want[:, :, off:off + ln])
            if not good:
                ok, why = False, f"allgather_decode {(k, m, S, nobj, lost, data_only, complete)}"
            ag_bytes = grp.bytes_exchanged()
            # ---- all-to-all decode
            wanted = [j for j in lost if not (data_only and j >= k)]
            reb = grp.alltoall_decode(mine, present, data_only=data_only, complete=complete).numpy()
            good = reb.shape == (len(wanted), nobj, S)
            for i, j in enumerate(wanted):
                if complete:
                    good = good and np.array_equal(reb[i], full[:, j])
                else:
                    good = good and np.array_equal(reb[i][:, off:off + ln], full[:, j, off:off + ln])
            if not good:
                ok, why = False, f"alltoall_decode {(k, m, S, nobj, lost, data_only, complete)}"
            if wanted and complete and world > 1 and not grp.bytes_exchanged() < ag_bytes:
                ok, why = False, f"all-to-all did not move fewer bytes: {grp.bytes_exchanged()} vs {ag_bytes}"
            grp.close()
        ok = ok and calls["ag"] > 0 and calls["a2a"] > 0
        q.put((rank, bool(ok), why))
        dist.destroy_process_group()
    except Exception:  # pragma: no cover
        import traceback

        q.put((rank, False, traceback.format_exc()))


def _run(world, cases):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, cases, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=30)
    for rank, ok, err in res:
        assert ok, f"rank {rank} failed: {err}"


CASES = [
    # k, m, S, nobj, lost, data_only, complete
    (20, 8, 1088, 3, (0, 1, 5, 9, 13, 19, 21, 27), False, True),   # config 5's code, 8 erasures (6 data + 2 parity)
    (10, 4, 832, 2, (1, 4, 13), False, True),                      # S/16 = 52 columns; surplus survivors
    (10, 4, 192, 2, (0, 3, 7, 11), False, False),                  # complete=0: only the rank's own byte range
    (10, 4, 192, 2, (2, 12), True, True),                          # data_only: the missing parity shard is left alone
    (3, 1, 64, 4, (1,), False, True),                              # 4 columns per shard: some ranks own none at world 3
    (10, 4, 4160, 1, (), False, True),                             # nothing missing: the exchange alone
]


@pytest.mark.timeout(300)
def test_library_group_code_world2():
    _run(2, CASES)


@pytest.mark.timeout(300)
def test_library_group_code_world3_ragged():
    """3 ranks: 14 shards -> 5 slots with padding, 28 shards -> 10 slots; column counts that do not divide by 3."""
    _run(3, CASES)


def test_cpu_group_argument_errors():
    import garage_amd as g
    from garage_amd import _lib

    codec = g.ReedSolomon(10, 4, backend="cpu")
    h = ctypes.c_void_p()
    ident = (ctypes.c_uint8 * _lib.GEC_GROUP_ID_BYTES)()
    # RCCL moves device memory: an RCCL group over a CPU codec is refused, loudly
    assert _lib.lib.gec_group_create(codec._h, 0, 1, ident, ctypes.byref(h)) == _lib.GEC_E_DEVICE
    assert _lib.lib.gec_group_create_with_transport(codec._h, 0, 1, None, None, ctypes.byref(h)) == _lib.GEC_E_INVALID_ARG
