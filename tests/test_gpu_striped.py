"""GPU side of the striped-object decode (BASELINE config 5): the scattered-
offset reconstruct kernel on the all-gather layout, with 8 logical ranks on the
one visible device, and the real RCCL path at world size 1."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import garage_amd as g  # noqa: E402
from garage_amd.striped import StripeLayout, gather_stripes, scatter_stripes, striped_reconstruct  # noqa: E402
from oracle import rs_oracle as O  # noqa: E402

DEV = "cuda:0"


def _stripes(coracle, k, m, S, nobj, seed):
    data = O.splitmix64_bytes(seed, nobj * k * S).reshape(nobj, k, S)
    return np.concatenate([data, coracle.encode_batch(k, m, data, coracle.AVX2, threads=4)], axis=1)


def test_config5_layout_8_logical_ranks(coracle):
    """RS(20,8), 4 MiB objects (S = 209728), shard j on rank j % 8: what each of
    the 8 GPUs computes after the all-gather, run rank by rank on one device."""
    k, m, world, nobj = 20, 8, 8, 4
    S = g.shard_len(k, 4 << 20)
    layout = StripeLayout(k, m, world)
    full = _stripes(coracle, k, m, S, nobj, 55)
    lost = (0, 1, 5, 9, 13, 19, 21, 27)
    present = [j not in lost for j in range(k + m)]
    broken = torch.from_numpy(full).to(DEV)
    broken[:, list(lost)] = 0xEE
    # the gathered buffer every rank holds after step (1)
    gathered = torch.stack([scatter_stripes(broken, layout, r) for r in range(world)]).contiguous()
    rs = g.ReedSolomon(k, m)
    offs = layout.shard_offsets(nobj, S)
    for r in range(world):
        off, ln = layout.byte_range(r, S)
        rs.reconstruct_scattered_dev(gathered.view(-1), nobj, layout.slots * S, offs, S, present, byte_range=(off, ln))
    torch.cuda.synchronize()
    got = gather_stripes(gathered, layout).cpu().numpy()
    assert np.array_equal(got, full)
    # padding slots (ranks 4..7 own 3 shards) must be untouched zeros
    assert not gathered[4:, :, 3].any()


def test_scattered_argument_errors():
    rs = g.ReedSolomon(10, 4)
    buf = torch.zeros(14 * 64, dtype=torch.uint8, device=DEV)
    offs = [j * 64 for j in range(14)]
    present = [1] * 13 + [0]
    rs.reconstruct_scattered_dev(buf, 1, 14 * 64, offs, 64, present)
    with pytest.raises(g.GecError):
        rs.reconstruct_scattered_dev(buf, 2, 14 * 64, offs, 64, present)          # past the end
    with pytest.raises(g.GecError):
        rs.reconstruct_scattered_dev(buf, 1, 14 * 64, offs[:-1] + [13 * 64 + 8], 64, present)  # misaligned offset
    with pytest.raises(g.GecError):
        rs.reconstruct_scattered_dev(buf, 1, 14 * 64, offs[:5], 64, present)      # wrong count


def test_striped_reconstruct_rccl_world1(coracle):
    import torch.distributed as dist

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        k, m, S, nobj = 10, 4, 4160, 3
        layout = StripeLayout(k, m, 1)
        full = _stripes(coracle, k, m, S, nobj, 56)
        lost = (0, 3, 7, 9)
        broken = torch.from_numpy(full).to(DEV)
        broken[:, list(lost)] = 0
        out = striped_reconstruct(g.ReedSolomon(k, m), scatter_stripes(broken, layout, 0),
                                  [j not in lost for j in range(k + m)], layout)
        torch.cuda.synchronize()
        assert np.array_equal(gather_stripes(out, layout).cpu().numpy(), full)
    finally:
        dist.destroy_process_group()
