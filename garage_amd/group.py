"""Multi-GPU decode of a striped object through the C ABI (gec_group_*,
include/garage_ec.h): the same three steps as ``garage_amd.striped`` -- all-gather
of the slot buffers, per-rank byte-range reconstruct in place, exchange of the
rebuilt ranges -- but with RCCL driven by libgarage_ec itself (``ncclAllGather``
over xGMI), which is what a host without torch (Garage's Rust shim) calls.
torch is only used here for device tensors and, in ``from_torch_distributed``,
to carry rank 0's RCCL unique id to the other ranks.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Sequence

import numpy as np

from . import _lib
from ._lib import GecError, check, lib
from .codec import ReedSolomon, _stream_handle, _u8p


class Group:
    def __init__(self, codec: ReedSolomon, rank: int, nranks: int, unique_id: Optional[bytes] = None,
                 transport: Optional[tuple[int, int]] = None):
        """RCCL group (``unique_id`` from ``Group.unique_id()`` on rank 0, identical on
        all ranks; collective) or, with ``transport=(fn_ptr, ctx_ptr)``, a group over a
        caller-supplied all-gather (``gec_allgather_fn``)."""
        self.codec = codec
        self.rank, self.nranks = rank, nranks
        h = ctypes.c_void_p()
        if transport is not None:
            # (all_gather_fn, ctx) or (all_gather_fn, all_to_all_fn, ctx)
            if len(transport) == 3:
                fn, a2a, ctx = transport
                check(lib.gec_group_create_with_transport2(codec._h, rank, nranks, fn, a2a, ctx, ctypes.byref(h)),
                      "gec_group_create_with_transport2")
            else:
                fn, ctx = transport
                check(lib.gec_group_create_with_transport(codec._h, rank, nranks, fn, ctx, ctypes.byref(h)),
                      "gec_group_create_with_transport")
        else:
            if unique_id is None or len(unique_id) != _lib.GEC_GROUP_ID_BYTES:
                raise GecError(_lib.GEC_E_INVALID_ARG, "unique_id", f"must be {_lib.GEC_GROUP_ID_BYTES} bytes")
            buf = (ctypes.c_uint8 * _lib.GEC_GROUP_ID_BYTES).from_buffer_copy(unique_id)
            check(lib.gec_group_create(codec._h, rank, nranks, buf, ctypes.byref(h)), "gec_group_create")
        self._h = h

    @staticmethod
    def unique_id() -> bytes:
        buf = (ctypes.c_uint8 * _lib.GEC_GROUP_ID_BYTES)()
        check(lib.gec_group_unique_id(buf), "gec_group_unique_id")
        return bytes(buf)

    @classmethod
    def from_torch_distributed(cls, codec: ReedSolomon, group=None) -> "Group":
        """One rank per process, already inside a torch.distributed job: rank 0 draws
        the RCCL unique id and the job's own (any-backend) process group carries it."""
        import torch.distributed as dist

        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [cls.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        return cls(codec, rank, world, box[0])

    @property
    def slots(self) -> int:
        return int(lib.gec_group_slots(self._h))

    def allgather_decode(self, local_slots, present: Sequence[int], data_only: bool = False, complete: bool = True,
                         out=None):
        """local_slots: (nobjects, slots, S) uint8 CUDA tensor (this rank's shards, slot s =
        shard s*nranks + rank).  Returns the gathered buffer (nranks, nobjects, slots, S),
        missing shards rebuilt (see gec_group_allgather_decode); stream-ordered on the
        current torch stream."""
        import torch

        on_host = self.codec.backend == "cpu"   # a group over a CPU codec (caller transport) works on CPU tensors
        if not (isinstance(local_slots, torch.Tensor) and local_slots.is_cuda != on_host and local_slots.dtype == torch.uint8
                and local_slots.dim() == 3 and local_slots.shape[1] == self.slots):
            raise TypeError(f"local_slots must be a uint8 {'CPU' if on_host else 'CUDA'} tensor (nobjects, {self.slots}, S)")
        if not on_host and local_slots.device.index != self.codec.device:
            raise GecError(_lib.GEC_E_INVALID_ARG, "local_slots", "tensor is on a different device than the codec")
        local_slots = local_slots.contiguous()
        nobj, slots, S = local_slots.shape
        pres = np.ascontiguousarray(np.asarray(present, dtype=np.uint8))
        if pres.size != self.codec.n:
            raise GecError(_lib.GEC_E_INVALID_INDEX, "present", "must have k+m entries")
        if out is None:
            out = torch.empty((self.nranks, nobj, slots, S), dtype=torch.uint8, device=local_slots.device)
        elif not (out.is_cuda != on_host and out.dtype == torch.uint8 and out.is_contiguous() and out.numel() == self.nranks * local_slots.numel()):
            raise TypeError("out must be a contiguous uint8 tensor of nranks*nobjects*slots*S bytes, where the slots are")
        check(lib.gec_group_allgather_decode(self._h, nobj, local_slots.data_ptr(), S, _u8p(pres), int(bool(data_only)),
                                             int(bool(complete)), out.data_ptr(), None if on_host else _stream_handle(self.codec.device)),
              "gec_group_allgather_decode")
        return out

    def alltoall_decode(self, local_slots, present: Sequence[int], data_only: bool = False, complete: bool = True, out=None):
        """All-to-all exchange (gec_group_alltoall_decode): every rank receives only its byte range of the k shards
        the decode reads.  Returns the rebuilt shards (nmiss, nobjects, S), nmiss = missing (data_only: missing data)
        shards in ascending index order; with complete=False only this rank's byte range of them is valid."""
        import torch

        on_host = self.codec.backend == "cpu"
        if not (isinstance(local_slots, torch.Tensor) and local_slots.is_cuda != on_host and local_slots.dtype == torch.uint8
                and local_slots.dim() == 3 and local_slots.shape[1] == self.slots):
            raise TypeError(f"local_slots must be a uint8 {'CPU' if on_host else 'CUDA'} tensor (nobjects, {self.slots}, S)")
        local_slots = local_slots.contiguous()
        nobj, slots, S = local_slots.shape
        pres = np.ascontiguousarray(np.asarray(present, dtype=np.uint8))
        if pres.size != self.codec.n:
            raise GecError(_lib.GEC_E_INVALID_INDEX, "present", "must have k+m entries")
        nmiss = sum(1 for j in range(self.codec.n) if not pres[j] and not (data_only and j >= self.codec.k))
        if out is None:
            out = torch.zeros((nmiss, nobj, S), dtype=torch.uint8, device=local_slots.device)
        if nmiss == 0:   # nothing to rebuild: no exchange either (the pattern is the same on every rank)
            return out
        check(lib.gec_group_alltoall_decode(self._h, nobj, local_slots.data_ptr(), S, _u8p(pres), int(bool(data_only)),
                                            int(bool(complete)), out.data_ptr(), None if on_host else _stream_handle(self.codec.device)),
              "gec_group_alltoall_decode")
        return out

    def peer_decode(self, local_slots, peer_ptrs: Sequence[int], present: Sequence[int], data_only: bool = False, complete: bool = True,
                    out=None):
        """Peer-pointer exchange (gec_group_peer_decode): ``peer_ptrs[q]`` is rank q's slot buffer as an address this process can
        use on the codec's device (``tensor.data_ptr()`` of a thread-rank's buffer, or what ``ipc_open`` returned; entry [rank] is
        overwritten with ``local_slots``).  Returns the rebuilt shards (nmiss, nobjects, S) like ``alltoall_decode``."""
        import torch

        on_host = self.codec.backend == "cpu"
        if not (isinstance(local_slots, torch.Tensor) and local_slots.is_cuda != on_host and local_slots.dtype == torch.uint8
                and local_slots.dim() == 3 and local_slots.shape[1] == self.slots and local_slots.is_contiguous()):
            raise TypeError(f"local_slots must be a contiguous uint8 {'CPU' if on_host else 'CUDA'} tensor (nobjects, {self.slots}, S)")
        nobj, slots, S = local_slots.shape
        pres = np.ascontiguousarray(np.asarray(present, dtype=np.uint8))
        if pres.size != self.codec.n or len(peer_ptrs) != self.nranks:
            raise GecError(_lib.GEC_E_INVALID_INDEX, "present / peer_ptrs", "need k+m flags and one pointer per rank")
        ptrs = list(peer_ptrs)
        ptrs[self.rank] = local_slots.data_ptr()
        arr = (ctypes.c_void_p * self.nranks)(*ptrs)
        nmiss = sum(1 for j in range(self.codec.n) if not pres[j] and not (data_only and j >= self.codec.k))
        if out is None:
            out = torch.zeros((nmiss, nobj, S), dtype=torch.uint8, device=local_slots.device)
        if nmiss == 0:
            return out
        check(lib.gec_group_peer_decode(self._h, nobj, arr, S, _u8p(pres), int(bool(data_only)), int(bool(complete)), out.data_ptr(),
                                        None if on_host else _stream_handle(self.codec.device)), "gec_group_peer_decode")
        return out

    def bytes_exchanged(self) -> int:
        """bytes this rank received from other ranks in the last decode call"""
        return int(lib.gec_group_bytes_exchanged(self._h))

    def close(self) -> None:
        if getattr(self, "_h", None):
            lib.gec_group_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # pragma: no cover - interpreter shutdown
            pass


def ipc_export(tensor) -> bytes:
    """gec_ipc_export: a 72-byte handle another process passes to ipc_open to address this CUDA tensor's memory."""
    buf = ctypes.create_string_buffer(_lib.GEC_IPC_HANDLE_BYTES)
    check(lib.gec_ipc_export(tensor.data_ptr(), buf), "gec_ipc_export")
    return buf.raw


def ipc_open(handle: bytes, device: int = 0) -> int:
    """gec_ipc_open: the device address (int) of the memory another process exported; release with ipc_close."""
    p = ctypes.c_void_p()
    check(lib.gec_ipc_open(handle, device, ctypes.byref(p)), "gec_ipc_open")
    return int(p.value)


def ipc_close(ptr: int) -> None:
    check(lib.gec_ipc_close(ptr), "gec_ipc_close")
