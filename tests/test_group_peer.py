"""gec_group_peer_decode -- the peer-pointer form of the striped-object exchange (every rank's decode reads ITS byte range of
the k surviving shards straight out of the other ranks' memory, one launch, no staging) -- on a box without a GPU: N ranks as
threads of one process over a GEC_BACKEND_CPU codec (the "peers' slot buffers" are plain host pointers), the group's
transport carrying only the two barriers and the rebuilt ranges.  Against stripes the ORACLE encoded.
The GPU forms (logical ranks on one device; two processes through HIP IPC handles) are in tests/test_gpu_group.py."""
import ctypes
import threading

import numpy as np
import pytest
import torch

import garage_amd as g
from garage_amd import _lib
from garage_amd.striped import StripeLayout, scatter_stripes
from oracle import rs_oracle as O


def run_thread_ranks(codec_factory, k, m, world, S, nobj, lost, data_only, complete, to_device=lambda t: t, transport_factory=None):
    """-> [(rebuilt ndarray, bytes_exchanged)] per rank, and the oracle's full stripes"""
    data = O.splitmix64_bytes(5 + k + S, nobj * k * S).reshape(nobj, k, S)
    full = np.concatenate([data, np.stack([O.encode(k, m, d) for d in data])], axis=1)     # the ORACLE's stripes
    present = [j not in lost for j in range(k + m)]
    broken = full.copy()
    broken[:, list(lost)] = 0xEE
    layout = StripeLayout(k, m, world)
    bar = threading.Barrier(world)
    sends = [None] * world

    def host_transport(r):
        @_lib.ALLGATHER_FN
        def ag(_ctx, send, recv, nbytes, _stream):
            sends[r] = send
            bar.wait()
            for q in range(world):
                ctypes.memmove(recv + q * nbytes, sends[q], nbytes)
            bar.wait()
            return 0
        return (ag, None, None)

    make = transport_factory or host_transport
    locs = [to_device(scatter_stripes(torch.from_numpy(broken), layout, r).contiguous()) for r in range(world)]
    ptrs = [t.data_ptr() for t in locs]
    outs, errs = [None] * world, []
    transports = [make(r) for r in range(world)]

    def rank(r):
        try:
            rs = codec_factory()
            grp = g.Group(rs, r, world, transport=transports[r])
            reb = grp.peer_decode(locs[r], ptrs, present, data_only=data_only, complete=complete)
            if reb.is_cuda:
                torch.cuda.synchronize()
            outs[r] = (reb.cpu().numpy().copy(), grp.bytes_exchanged())
            grp.close()
        except BaseException as e:  # noqa: BLE001
            errs.append((r, e))
            bar.abort()

    ts = [threading.Thread(target=rank, args=(r,), daemon=True) for r in range(world)]
    [t.start() for t in ts]
    [t.join(timeout=120) for t in ts]
    assert not any(t.is_alive() for t in ts), "a rank hung"
    assert not errs, errs
    return outs, full, layout


def check(outs, full, layout, k, S, lost, data_only, complete):
    wanted = [j for j in lost if not (data_only and j >= k)]
    nobj = full.shape[0]
    for r, (reb, _) in enumerate(outs):
        assert reb.shape == (len(wanted), nobj, S)
        off, ln = layout.byte_range(r, S)
        for i, j in enumerate(wanted):
            if complete:
                assert np.array_equal(reb[i], full[:, j]), (r, j)
            else:
                assert np.array_equal(reb[i][:, off:off + ln], full[:, j, off:off + ln]), (r, j)


CASES = [
    # k, m, world, S, nobj, lost, data_only, complete
    (20, 8, 8, 1088, 3, (0, 1, 5, 9, 13, 19, 21, 27), False, True),   # config 5's code and placement, 8 erasures
    (10, 4, 3, 832, 2, (1, 4, 13), False, True),                      # 52 columns over 3 ranks: ragged ranges, padded slots
    (10, 4, 2, 192, 2, (0, 3, 7, 11), False, False),                  # complete=0: only the rank's own byte range
    (10, 4, 4, 192, 2, (2, 12), True, True),                          # data_only: the missing parity shard is not rebuilt
    (3, 1, 3, 64, 4, (1,), False, True),                              # 4 columns per shard over 3 ranks
    (10, 12, 4, 256, 2, tuple(range(2, 12)), False, True),            # ten rows: two row groups in the one table
    (10, 4, 4, 4160, 1, (), False, True),                             # nothing missing: nothing happens
]


@pytest.mark.parametrize("k,m,world,S,nobj,lost,data_only,complete", CASES)
def test_peer_decode_thread_ranks_on_host_memory(k, m, world, S, nobj, lost, data_only, complete):
    outs, full, layout = run_thread_ranks(lambda: g.ReedSolomon(k, m, backend="cpu"), k, m, world, S, nobj, lost, data_only, complete)
    check(outs, full, layout, k, S, lost, data_only, complete)
    if lost and world > 1:
        # what a rank read out of OTHER ranks' memory: its byte range of the valid shards it does not hold -- about 1/N of the
        # k survivors, not all of them (the all-gather moves slots * S * (N - 1) per object and rank)
        valid = [j for j in range(k + m) if j not in lost][:k]
        wanted = [j for j in lost if not (data_only and j >= k)]
        for r, (_, moved) in enumerate(outs):
            off, ln = layout.byte_range(r, S)
            remote = sum(1 for v in valid if v % world != r) * nobj * ln
            second = (world - 1) * len(wanted) * nobj * max(layout.byte_range(q, S)[1] for q in range(world)) if complete else 0
            assert moved == remote + second, (r, moved, remote, second)
            assert remote < layout.slots * S * (world - 1) * nobj


def test_peer_decode_argument_errors():
    rs = g.ReedSolomon(10, 4, backend="cpu")
    fn = _lib.ALLGATHER_FN(lambda *a: 0)
    grp = g.Group(rs, 0, 1, transport=(fn, None, None))
    loc = torch.zeros((1, 14, 64), dtype=torch.uint8)
    pres = np.ones(14, dtype=np.uint8)
    pres[:5] = 0
    with pytest.raises(g.GecError) as ei:
        grp.peer_decode(loc, [0], pres)
    assert ei.value.code == _lib.GEC_E_TOO_FEW_PRESENT
    out = ctypes.c_void_p()
    assert _lib.lib.gec_group_peer_decode(None, 1, None, 64, None, 0, 1, None, None) == _lib.GEC_E_INVALID_ARG
    assert _lib.lib.gec_ipc_export(None, None) == _lib.GEC_E_INVALID_ARG
    assert _lib.lib.gec_ipc_open(None, 0, ctypes.byref(out)) == _lib.GEC_E_INVALID_ARG
    assert _lib.lib.gec_ipc_close(None) == _lib.GEC_OK
    grp.close()
