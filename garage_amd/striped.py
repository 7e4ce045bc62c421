"""Multi-GPU decode of an object striped over the GPUs of one node
(BASELINE config 5).

Shard j of every object lives on rank ``j % world`` in slot ``j // world``
(ranks that own fewer shards pad to ``slots = ceil(n / world)``).  Decode has
the path's one real exchange step:

  1. ONE all-gather of every rank's slot buffer (equal-sized, uint8) --
     ``torch.distributed.all_gather_into_tensor``; backend "nccl" is RCCL over
     xGMI on MI355X, "gloo" in the CPU tests;
  2. every rank rebuilds its 1/world byte-range of every missing shard, in place
     inside the gathered buffer, with ``gec_reconstruct_scattered_dev`` (no
     permute copy: the kernel takes per-shard offsets);
  3. optionally a second all-gather of just the rebuilt ranges so that every
     rank holds the complete missing shards.

Whole (non-striped) blocks never come through here: they are independent units,
hash-partitioned over ranks with no collective (``garage_amd.partition``).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Sequence

import torch
import torch.distributed as dist


@dataclass(frozen=True)
class StripeLayout:
    k: int
    m: int
    world: int

    @property
    def n(self) -> int:
        return self.k + self.m

    @property
    def slots(self) -> int:
        return -(-self.n // self.world)

    def owner(self, j: int) -> int:
        return j % self.world

    def slot(self, j: int) -> int:
        return j // self.world

    def shards_of(self, rank: int) -> list[int]:
        return [j for j in range(self.n) if j % self.world == rank]

    def shard_offsets(self, nobjects: int, S: int) -> list[int]:
        """Byte offset of shard j of object 0 inside the gathered buffer
        [rank][object][slot][S]; consecutive objects are slots*S apart."""
        per_rank = nobjects * self.slots * S
        return [self.owner(j) * per_rank + self.slot(j) * S for j in range(self.n)]

    def byte_range(self, rank: int, S: int) -> tuple[int, int]:
        """This rank's 16-byte-aligned share [off, off+len) of every shard."""
        cols = S // 16
        lo = cols * rank // self.world
        hi = cols * (rank + 1) // self.world
        return lo * 16, (hi - lo) * 16


def striped_reconstruct(codec, local_slots: torch.Tensor, present: Sequence[int], layout: StripeLayout,
                        group: Optional[dist.ProcessGroup] = None, data_only: bool = False,
                        complete: bool = True) -> torch.Tensor:
    """local_slots: (nobjects, slots, S) uint8 -- this rank's shards (slot s holds
    shard ``s*world + rank``; contents of slots whose shard is erased or padding
    are ignored).  ``present``: the k+m flags of the object's erasure pattern,
    identical on all ranks.

    Returns the gathered buffer, shape (world, nobjects, slots, S): shard j is
    ``out[j % world, :, j // world]``.  With ``complete`` every missing shard is
    fully rebuilt on every rank; without it only this rank's byte range of each
    missing shard is valid (enough when the next stage is also range-partitioned).

    ``codec`` must provide ``reconstruct_scattered_dev`` (``garage_amd.ReedSolomon``).
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if world != layout.world:
        raise ValueError(f"layout is for {layout.world} ranks, group has {world}")
    if local_slots.dim() != 3 or local_slots.shape[1] != layout.slots or local_slots.dtype != torch.uint8:
        raise ValueError(f"local_slots must be uint8 (nobjects, {layout.slots}, S)")
    if len(present) != layout.n:
        raise ValueError("present must have k+m entries")
    nobj, slots, S = local_slots.shape
    if S % 64:
        raise ValueError("S must be a multiple of 64")
    local_slots = local_slots.contiguous()
    gathered = torch.empty((world, nobj, slots, S), dtype=torch.uint8, device=local_slots.device)
    # (1) the exchange step: survivors' slots to everybody
    dist.all_gather_into_tensor(gathered.view(-1), local_slots.view(-1), group=group)

    missing = [j for j in range(layout.n) if not present[j] and not (data_only and j >= layout.k)]
    if not missing:
        return gathered
    # (2) my byte range of every missing shard, in place
    off, ln = layout.byte_range(rank, S)
    if ln:
        codec.reconstruct_scattered_dev(gathered.view(-1), nobj, slots * S, layout.shard_offsets(nobj, S), S,
                                        present, data_only=data_only, byte_range=(off, ln))
    if not complete:
        return gathered
    # (3) exchange the rebuilt ranges: ranks have ranges of different length when
    # S/16 is not a multiple of world, so pad to the longest
    max_ln = max(layout.byte_range(r, S)[1] for r in range(world))
    mine = torch.zeros((len(missing), nobj, max_ln), dtype=torch.uint8, device=gathered.device)
    for i, j in enumerate(missing):
        mine[i, :, :ln] = gathered[layout.owner(j), :, layout.slot(j), off:off + ln]
    parts = torch.empty((world,) + tuple(mine.shape), dtype=torch.uint8, device=gathered.device)
    dist.all_gather_into_tensor(parts.view(-1), mine.view(-1), group=group)
    for r in range(world):
        roff, rln = layout.byte_range(r, S)
        if r == rank or rln == 0:
            continue
        for i, j in enumerate(missing):
            gathered[layout.owner(j), :, layout.slot(j), roff:roff + rln] = parts[r, i, :, :rln]
    return gathered


def striped_reconstruct_alltoall(codec, local_slots: torch.Tensor, present: Sequence[int], layout: StripeLayout,
                                 group: Optional[dist.ProcessGroup] = None, data_only: bool = False,
                                 complete: bool = True) -> torch.Tensor:
    """The all-to-all form of the exchange (gec_group_alltoall_decode in the C ABI): every rank receives only ITS
    byte range of the k shards the decode reads -- (world-1)/world^2 of k*S per object instead of the whole
    survivor set -- rebuilds its range of the missing shards, and (``complete``) the rebuilt ranges are
    all-gathered.  Returns the rebuilt shards only, shape (nmiss, nobjects, S), in ascending index order of the
    missing (``data_only``: missing data) shards; survivors stay where they are."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if world != layout.world:
        raise ValueError(f"layout is for {layout.world} ranks, group has {world}")
    nobj, slots, S = local_slots.shape
    if slots != layout.slots or S % 64 or len(present) != layout.n:
        raise ValueError("bad local_slots / present")
    n, k = layout.n, layout.k
    valid = [j for j in range(n) if present[j]][:k]
    if len(valid) < k:
        raise ValueError("fewer than k shards present")
    wanted = [j for j in range(n) if not present[j] and not (data_only and j >= k)]
    dev = local_slots.device
    if not wanted:
        return torch.zeros((0, nobj, S), dtype=torch.uint8, device=dev)
    valid_of = [[v for v in valid if layout.owner(v) == r] for r in range(world)]
    nvs_max = max(len(v) for v in valid_of)
    ranges = [layout.byte_range(r, S) for r in range(world)]
    pitch = -(-max(ln for _, ln in ranges) // 64) * 64          # column pitch of the exchanged ranges
    send = torch.zeros((world, nvs_max, nobj, pitch), dtype=torch.uint8, device=dev)
    for peer, (off, ln) in enumerate(ranges):
        for vs, v in enumerate(valid_of[rank]):
            send[peer, vs, :, :ln] = local_slots[:, layout.slot(v), off:off + ln]
    recv = torch.empty_like(send)
    dist.all_to_all_single(recv.view(-1), send.view(-1), group=group)   # (1) the exchange step
    # (2) my byte range of every shard the decode can produce, into an area behind the received ranges
    others = [j for j in range(n) if j not in valid]
    buf = torch.cat([recv.view(-1), torch.zeros(len(others) * nobj * pitch, dtype=torch.uint8, device=dev)])
    shard_off = [0] * n
    for r in range(world):
        for vs, v in enumerate(valid_of[r]):
            shard_off[v] = (r * nvs_max + vs) * nobj * pitch
    for i, j in enumerate(others):
        shard_off[j] = recv.numel() + i * nobj * pitch
    off, ln = ranges[rank]
    if ln:
        codec.reconstruct_scattered_dev(buf, nobj, pitch, shard_off, pitch, [j in valid for j in range(n)],
                                        data_only=False, byte_range=(0, ln))
    mine = torch.stack([buf[shard_off[j]: shard_off[j] + nobj * pitch].view(nobj, pitch) for j in wanted])
    rebuilt = torch.zeros((len(wanted), nobj, S), dtype=torch.uint8, device=dev)
    if complete and world > 1:                                  # (3) exchange the rebuilt ranges
        parts = torch.empty((world,) + tuple(mine.shape), dtype=torch.uint8, device=dev)
        dist.all_gather_into_tensor(parts.view(-1), mine.contiguous().view(-1), group=group)
        for r, (roff, rln) in enumerate(ranges):
            rebuilt[:, :, roff:roff + rln] = parts[r][:, :, :rln]
    else:
        rebuilt[:, :, off:off + ln] = mine[:, :, :ln]
    return rebuilt


def scatter_stripes(stripes: torch.Tensor, layout: StripeLayout, rank: int) -> torch.Tensor:
    """Test/bench helper: the slot buffer rank `rank` would hold for full stripes
    (nobjects, n, S)."""
    nobj, n, S = stripes.shape
    out = torch.zeros((nobj, layout.slots, S), dtype=torch.uint8, device=stripes.device)
    for j in layout.shards_of(rank):
        out[:, layout.slot(j)] = stripes[:, j]
    return out


def gather_stripes(gathered: torch.Tensor, layout: StripeLayout) -> torch.Tensor:
    """(world, nobjects, slots, S) -> (nobjects, n, S) contiguous stripes."""
    return torch.stack([gathered[layout.owner(j), :, layout.slot(j)] for j in range(layout.n)], dim=1)
