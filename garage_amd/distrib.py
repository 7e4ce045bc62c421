"""One process per GPU: the few collectives the bench / multi-GPU paths need.

Launched as `python -m torch.distributed.run --nproc-per-node N ...`; reads
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment.  Backend "nccl"
(= RCCL over xGMI on MI355X) when a GPU is visible, "gloo" otherwise (CPU tests).
The encode path itself needs NO collective: blocks are independent units.
"""
from __future__ import annotations

import os
from dataclasses import dataclass

import torch


@dataclass
class Ranks:
    rank: int
    world: int
    local_rank: int
    device: torch.device
    backend: str | None  # None: single process, no process group

    @property
    def distributed(self) -> bool:
        return self.backend is not None


def init_from_env(force_backend: str | None = None) -> Ranks:
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_cuda = torch.cuda.is_available() and force_backend != "gloo"
    # Dry-run hook for boxes with ONE GPU: every rank uses device 0 and the collectives go
    # through gloo (RCCL refuses two ranks on one device).  Exercises the N>1 control flow of
    # bench.py end to end; the numbers it prints are meaningless.
    dryrun = os.environ.get("GARAGE_DRYRUN_ONE_GPU") == "1"
    if dryrun:
        local_rank = 0
        force_backend = "gloo"
    if use_cuda:
        # a launcher may already have narrowed each rank to its own GPU (HIP_VISIBLE_DEVICES):
        # then every rank sees one device, index 0
        ndev = max(1, torch.cuda.device_count())
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
        if not dryrun and ndev > 1 and ndev < local_world:
            raise RuntimeError(f"{local_world} ranks on this node but only {ndev} GPUs are visible: RCCL refuses two ranks "
                               "on one device (GARAGE_DRYRUN_ONE_GPU=1 gives a plumbing-only dry run on device 0)")
        local_rank = local_rank % ndev
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)
    else:
        device = torch.device("cpu")
    if world == 1:
        return Ranks(rank, world, local_rank, device, None)
    import torch.distributed as dist

    backend = force_backend or ("nccl" if use_cuda else "gloo")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    # RCCL prints a version banner on STDOUT at NCCL_DEBUG=VERSION/INFO; bench.py's contract
    # is ONE JSON line on stdout, so keep RCCL at WARN unless the caller insists.
    # (its WARN lines go to stdout too -- e.g. "alt_rsmi.cc NCCL WARN Could not read node" on
    # some boxes -- so send RCCL's log to stderr.)
    quiet_rccl()
    if backend == "nccl":
        dist.init_process_group(backend, rank=rank, world_size=world, device_id=device)
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    return Ranks(rank, world, local_rank, device, backend)


def quiet_rccl() -> None:
    """Keep RCCL's own output off stdout (bench.py prints ONE JSON line there)."""
    if os.environ.get("GARAGE_KEEP_NCCL_DEBUG") is None:
        os.environ["NCCL_DEBUG"] = "WARN"
        os.environ.setdefault("NCCL_DEBUG_FILE", "/dev/stderr")


def barrier(r: Ranks) -> None:
    if r.distributed:
        import torch.distributed as dist

        dist.barrier()


def max_over_ranks(r: Ranks, value: float) -> float:
    if not r.distributed:
        return float(value)
    import torch.distributed as dist

    t = torch.tensor([value], dtype=torch.float64, device=r.device if r.backend == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(r: Ranks, value: int) -> int:
    if not r.distributed:
        return int(value)
    import torch.distributed as dist

    t = torch.tensor([value], dtype=torch.int64, device=r.device if r.backend == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())


def gather_ints(r: Ranks, value: int) -> list[int]:
    """value of every rank, in rank order, on every rank."""
    if not r.distributed:
        return [int(value)]
    import torch.distributed as dist

    dev = r.device if r.backend == "nccl" else "cpu"
    mine = torch.tensor([value], dtype=torch.int64, device=dev)
    out = torch.empty((r.world,), dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(out, mine)
    return [int(x) for x in out.tolist()]


def count_ranks(r: Ranks) -> dict:
    """How many ranks the collective backend actually connected: an all-reduce (SUM) of ones --
    on the device over RCCL when the backend is "nccl" -- so "RCCL saw N ranks" is a measured
    statement, not WORLD_SIZE read back from the environment."""
    if not r.distributed:
        return {"ranks": 1, "backend": "none (single process)"}
    import torch.distributed as dist

    t = torch.ones(1, dtype=torch.int32, device=r.device if r.backend == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    name = "rccl (torch.distributed backend nccl)" if r.backend == "nccl" else r.backend
    return {"ranks": int(t.item()), "backend": name}


def shutdown(r: Ranks) -> None:
    if r.distributed:
        import torch.distributed as dist

        dist.destroy_process_group()
