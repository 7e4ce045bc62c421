"""Host-side mirror of the `reed_solomon_erasure::galois_8::ReedSolomon` API
[EXT] over libgarage_ec's C ABI, so that parity tests read like the crate's own
tests (`new`, `encode_sep`, `verify`, `reconstruct`, `reconstruct_data`).

PyTorch is used only as plumbing for device memory and streams: tensors are
handed to the C ABI as raw pointers + the current HIP stream.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Sequence

import numpy as np

from . import _lib
from ._lib import GecError, check, lib


def shard_len(k: int, block_len: int) -> int:
    """S = round_up(ceil(L/k), 64) -- gec_shard_len."""
    return int(lib.gec_shard_len(k, block_len))


MATRIX_KINDS = {"vandermonde": _lib.GEC_MATRIX_VANDERMONDE, "cauchy": _lib.GEC_MATRIX_CAUCHY}
BACKENDS = {"cpu": _lib.GEC_BACKEND_CPU, "hip": _lib.GEC_BACKEND_HIP, "auto": _lib.GEC_BACKEND_AUTO}


def build_matrix(k: int, m: int, matrix: str = "vandermonde") -> np.ndarray:
    out = np.zeros((max(k + m, 1), max(k, 1)), dtype=np.uint8)
    check(lib.gec_build_matrix_ex(k, m, MATRIX_KINDS[matrix], out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))),
          "gec_build_matrix_ex")
    return out


def build_decode_matrix(k: int, m: int, present: Sequence[int]):
    pres = np.ascontiguousarray(np.asarray(present, dtype=np.uint8))
    if pres.size != k + m:
        raise GecError(_lib.GEC_E_INVALID_ARG, "gec_build_decode_matrix", "present must have k+m entries")
    valid = (ctypes.c_int32 * max(k, 1))()
    out = np.zeros((max(k, 1), max(k, 1)), dtype=np.uint8)
    check(lib.gec_build_decode_matrix(k, m, pres.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), valid,
                                      out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))),
          "gec_build_decode_matrix")
    return list(valid)[:k], out


def _u8p(a: np.ndarray):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))


def _stream_handle(device_index: int) -> int:
    import torch

    return int(torch.cuda.current_stream(device_index).cuda_stream)


class ReedSolomon:
    """`ReedSolomon::new(data_shards, parity_shards)` bound to one GPU (backend="hip", the default) or to the
    host cores (backend="cpu": the library's own CPU data path, host-pointer methods only); "auto" picks the
    GPU when there is one."""

    def __init__(self, data_shards: int, parity_shards: int, device: int = 0, matrix: str = "vandermonde",
                 backend: str = "hip", shardsum: int = 0, _handle=None):
        """matrix="vandermonde" is the crate-compatible default; "cauchy" is the extra family
        of include/garage_ec.h (not interchangeable with the default).  shardsum: which shard checksum the *_hash
        methods produce -- 3 = MLH64 (the default, 0), 2 = BLAKE2b tree mode (shard header version 2)."""
        h = ctypes.c_void_p()
        if _handle is not None:
            h = _handle
        else:
            check(lib.gec_codec_create_ex2(data_shards, parity_shards, BACKENDS[backend], device, MATRIX_KINDS[matrix],
                                           shardsum, ctypes.byref(h)), "gec_codec_create_ex2")
        self.matrix = matrix
        self._h = h
        self.shardsum_kind = int(lib.gec_codec_shardsum(h))
        self.k = data_shards
        self.m = parity_shards
        self.n = data_shards + parity_shards
        self.backend = {v: k for k, v in BACKENDS.items()}[int(lib.gec_codec_backend(h))]
        self.device = int(lib.gec_codec_device(h))

    def background(self) -> "ReedSolomon":
        """gec_codec_background: a sibling codec whose work is classed BACKGROUND (scrub, resync)."""
        h = ctypes.c_void_p()
        check(lib.gec_codec_background(self._h, ctypes.byref(h)), "gec_codec_background")
        return ReedSolomon(self.k, self.m, self.device, self.matrix, self.backend, _handle=h)

    def with_shardsum(self, kind: int) -> "ReedSolomon":
        """gec_codec_with_shardsum: a sibling codec that produces the other kind of shard checksum."""
        h = ctypes.c_void_p()
        check(lib.gec_codec_with_shardsum(self._h, kind, ctypes.byref(h)), "gec_codec_with_shardsum")
        return ReedSolomon(self.k, self.m, self.device, self.matrix, self.backend, _handle=h)

    @property
    def qos_class(self) -> int:
        return int(lib.gec_codec_class(self._h))

    # ---- NUMA placement of the codec's host side (include/garage_ec.h, gec_codec_numa_node)
    @property
    def numa_node(self) -> int:
        """the memory node the codec keeps its copy threads and pinned memory on (-1: CPU codec, one node, GEC_NUMA=0)"""
        return int(lib.gec_codec_numa_node(self._h))

    @property
    def numa_cpus(self) -> list[int]:
        cnt = ctypes.c_size_t()
        check(lib.gec_codec_numa_cpus(self._h, 0, None, ctypes.byref(cnt)), "gec_codec_numa_cpus")
        arr = (ctypes.c_int * max(cnt.value, 1))()
        check(lib.gec_codec_numa_cpus(self._h, cnt.value, arr, ctypes.byref(cnt)), "gec_codec_numa_cpus")
        return list(arr[: cnt.value])

    def host_alloc(self, nbytes: int) -> np.ndarray:
        """gec_host_alloc_near: pinned memory on this codec's memory node (release with garage_amd.host_free)"""
        p = lib.gec_host_alloc_near(self._h, nbytes)
        if not p:
            raise GecError(_lib.GEC_E_NOMEM, "gec_host_alloc_near", (lib.gec_last_error() or b"").decode("utf-8", "replace"))
        return np.ctypeslib.as_array((ctypes.c_uint8 * max(nbytes, 1)).from_address(p))[:nbytes]

    def close(self) -> None:
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                lib.gec_codec_destroy(h)
            except Exception:  # interpreter shutdown: the library may already be gone
                pass

    __del__ = close

    # -- introspection -----------------------------------------------------
    def data_shard_count(self) -> int:
        return int(lib.gec_codec_k(self._h))

    def parity_shard_count(self) -> int:
        return int(lib.gec_codec_m(self._h))

    def total_shard_count(self) -> int:
        return self.n

    def parity_matrix(self) -> np.ndarray:
        out = np.zeros((self.m, self.k), dtype=np.uint8)
        check(lib.gec_parity_matrix(self._h, _u8p(out)), "gec_parity_matrix")
        return out

    def cache_stats(self) -> tuple[int, int]:
        a, b = ctypes.c_uint64(), ctypes.c_uint64()
        check(lib.gec_codec_cache_stats(self._h, ctypes.byref(a), ctypes.byref(b)), "gec_codec_cache_stats")
        return int(a.value), int(b.value)

    # -- device-resident (torch uint8 CUDA tensors) ---------------------------
    def _check_dev(self, t, shards: int, name: str):
        import torch

        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.uint8):
            raise TypeError(f"{name} must be a uint8 CUDA tensor")
        if t.dim() != 3 or t.shape[1] != shards or not t.is_contiguous():
            raise GecError(_lib.GEC_E_INCORRECT_SHARD_SIZE, name, f"expected contiguous (nblocks, {shards}, S)")
        if t.device.index != self.device:
            raise GecError(_lib.GEC_E_INVALID_ARG, name, "tensor is on a different device than the codec")

    def encode_sep_dev(self, data, parity=None):
        """data: (nblocks, k, S) uint8 on the codec's GPU -> parity (nblocks, m, S).
        Asynchronous on torch's current stream."""
        import torch

        self._check_dev(data, self.k, "data")
        nb, _, S = data.shape
        if parity is None:
            parity = torch.empty((nb, self.m, S), dtype=torch.uint8, device=data.device)
        else:
            self._check_dev(parity, self.m, "parity")
            if parity.shape[0] != nb or parity.shape[2] != S:
                raise GecError(_lib.GEC_E_INCORRECT_SHARD_SIZE, "parity", "shape mismatch with data")
        check(lib.gec_encode_batch_dev(self._h, nb, data.data_ptr(), self.k * S, S, parity.data_ptr(),
                                       self.m * S, _stream_handle(self.device)), "gec_encode_batch_dev")
        return parity

    def encode_dev(self, stripes):
        """stripes: (nblocks, k+m, S); parity rows are overwritten in place
        (`ReedSolomon::encode(&mut shards)`)."""
        self._check_dev(stripes, self.n, "stripes")
        nb, _, S = stripes.shape
        base = stripes.data_ptr()
        check(lib.gec_encode_batch_dev(self._h, nb, base, self.n * S, S, base + self.k * S, self.n * S,
                                       _stream_handle(self.device)), "gec_encode_batch_dev")
        return stripes

    def verify_dev(self, stripes):
        """-> bool tensor (nblocks,), True where parity is consistent."""
        import torch

        self._check_dev(stripes, self.n, "stripes")
        nb, _, S = stripes.shape
        bad = torch.empty((nb,), dtype=torch.int32, device=stripes.device)
        check(lib.gec_verify_batch_dev(self._h, nb, stripes.data_ptr(), self.n * S, S, bad.data_ptr(),
                                       _stream_handle(self.device)), "gec_verify_batch_dev")
        return bad == 0

    def reconstruct_dev(self, stripes, present: Sequence[int], data_only: bool = False,
                        byte_range: Optional[tuple[int, int]] = None):
        """Rebuild the shards with present[j]==0 in place, one pattern per batch.
        byte_range=(off, len) restricts the work to that slice of every shard."""
        self._check_dev(stripes, self.n, "stripes")
        nb, _, S = stripes.shape
        pres = np.ascontiguousarray(np.asarray(present, dtype=np.uint8))
        if pres.size != self.n:
            raise GecError(_lib.GEC_E_INVALID_INDEX, "present", "must have k+m entries")
        off, ln = (0, S) if byte_range is None else byte_range
        check(lib.gec_reconstruct_range_dev(self._h, nb, stripes.data_ptr(), self.n * S, S, _u8p(pres),
                                            int(bool(data_only)), off, ln, _stream_handle(self.device)),
              "gec_reconstruct_range_dev")
        return stripes

    def reconstruct_dev_ex(self, stripes, present, data_only: bool = False):
        """gec_reconstruct_batch_dev_ex: `present` is (nblocks, k+m) -- an erasure pattern PER BLOCK, rebuilt in place by one
        launch.  (A CPU codec runs the same call on a CPU tensor.)"""
        import torch

        on_host = self.backend == "cpu"
        if on_host:
            if not (isinstance(stripes, torch.Tensor) and not stripes.is_cuda and stripes.dtype == torch.uint8 and stripes.dim() == 3
                    and stripes.shape[1] == self.n and stripes.is_contiguous()):
                raise TypeError("stripes must be a contiguous uint8 CPU tensor (nblocks, k+m, S)")
        else:
            self._check_dev(stripes, self.n, "stripes")
        nb, _, S = stripes.shape
        pres = np.ascontiguousarray(np.asarray(present, dtype=np.uint8))
        if pres.shape != (nb, self.n):
            raise GecError(_lib.GEC_E_INVALID_INDEX, "present", "must be (nblocks, k+m)")
        check(lib.gec_reconstruct_batch_dev_ex(self._h, nb, stripes.data_ptr(), self.n * S, S, _u8p(pres), int(bool(data_only)),
                                               None if on_host else _stream_handle(self.device)), "gec_reconstruct_batch_dev_ex")
        return stripes

    def reconstruct_scattered_dev(self, buf, nblocks: int, block_stride: int, shard_off: Sequence[int], S: int,
                                  present: Sequence[int], data_only: bool = False,
                                  byte_range: Optional[tuple[int, int]] = None):
        """Shard j of block b at buf.data_ptr() + b*block_stride + shard_off[j]
        (gec_reconstruct_scattered_dev).  `buf` is any uint8 CUDA tensor that
        covers those addresses; missing shards are rebuilt in place."""
        import torch

        on_host = self.backend == "cpu"   # a CPU codec runs the same strided call on HOST memory (CPU tensors)
        if not (isinstance(buf, torch.Tensor) and buf.is_cuda != on_host and buf.dtype == torch.uint8 and buf.is_contiguous()):
            raise TypeError("buf must be a contiguous uint8 %s tensor" % ("CPU" if on_host else "CUDA"))
        if not on_host and buf.device.index != self.device:
            raise GecError(_lib.GEC_E_INVALID_ARG, "buf", "tensor is on a different device than the codec")
        if len(shard_off) != self.n:
            raise GecError(_lib.GEC_E_INVALID_INDEX, "shard_off", "must have k+m entries")
        top = (nblocks - 1) * block_stride + max(shard_off) + S
        if nblocks > 0 and top > buf.numel():
            raise GecError(_lib.GEC_E_INCORRECT_SHARD_SIZE, "buf", "layout reaches past the end of the tensor")
        pres = np.ascontiguousarray(np.asarray(present, dtype=np.uint8))
        if pres.size != self.n:
            raise GecError(_lib.GEC_E_INVALID_INDEX, "present", "must have k+m entries")
        offs = (ctypes.c_size_t * self.n)(*[int(o) for o in shard_off])
        off, ln = (0, S) if byte_range is None else byte_range
        check(lib.gec_reconstruct_scattered_dev(self._h, nblocks, buf.data_ptr(), block_stride, offs, S, _u8p(pres),
                                                int(bool(data_only)), off, ln, None if on_host else _stream_handle(self.device)),
              "gec_reconstruct_scattered_dev")
        return buf

    # -- host buffers (what the Rust shim calls) -----------------------------
    def encode_blocks(self, blocks: Sequence[bytes], S: Optional[int] = None) -> list[np.ndarray]:
        """blocks: byte strings (any lengths) -> per block a (m, S) parity array."""
        nb = len(blocks)
        if nb == 0:
            return []
        if S is None:
            S = max(shard_len(self.k, len(b)) for b in blocks)
        bufs = [np.frombuffer(bytes(b), dtype=np.uint8) if len(b) else np.zeros(1, dtype=np.uint8) for b in blocks]
        lens = (ctypes.c_size_t * nb)(*[len(b) for b in blocks])
        ptrs = (ctypes.c_void_p * nb)(*[a.ctypes.data for a in bufs])
        outs = [np.empty((self.m, S), dtype=np.uint8) for _ in range(nb)]
        optrs = (ctypes.c_void_p * nb)(*[o.ctypes.data for o in outs])
        check(lib.gec_encode_batch(self._h, nb, ptrs, lens, S, optrs), "gec_encode_batch")
        return outs

    def encode_hash_blocks(self, blocks: Sequence[bytes], S: Optional[int] = None):
        """encode_blocks + the blake2sum of every shard, hashed on the GPU while the
        stripe is resident: -> (parities, sums) with sums shape (nblocks, k+m, 32)."""
        nb = len(blocks)
        if nb == 0:
            return [], np.zeros((0, self.n, 32), dtype=np.uint8)
        if S is None:
            S = max(shard_len(self.k, len(b)) for b in blocks)
        bufs = [np.frombuffer(bytes(b), dtype=np.uint8) if len(b) else np.zeros(1, dtype=np.uint8) for b in blocks]
        lens = (ctypes.c_size_t * nb)(*[len(b) for b in blocks])
        ptrs = (ctypes.c_void_p * nb)(*[a.ctypes.data for a in bufs])
        outs = [np.empty((self.m, S), dtype=np.uint8) for _ in range(nb)]
        optrs = (ctypes.c_void_p * nb)(*[o.ctypes.data for o in outs])
        sums = np.empty((nb, self.n, 32), dtype=np.uint8)
        check(lib.gec_encode_hash_batch(self._h, nb, ptrs, lens, S, optrs, _u8p(sums)), "gec_encode_hash_batch")
        return outs, sums

    def blake2sum_batch(self, msgs: Sequence[bytes]) -> list[bytes]:
        """Garage's blake2sum (blake2b-512[..32]) of every message, on the GPU."""
        n = len(msgs)
        if n == 0:
            return []
        bufs = [np.frombuffer(bytes(x), dtype=np.uint8) if len(x) else np.zeros(1, dtype=np.uint8) for x in msgs]
        ptrs = (ctypes.c_void_p * n)(*[a.ctypes.data for a in bufs])
        lens = (ctypes.c_size_t * n)(*[len(x) for x in msgs])
        out = np.empty((n, 32), dtype=np.uint8)
        check(lib.gec_blake2sum_batch(self._h, n, ptrs, lens, _u8p(out)), "gec_blake2sum_batch")
        return [out[i].tobytes() for i in range(n)]

    def shardsum_batch(self, msgs: Sequence[bytes]) -> list[bytes]:
        """The shard checksum (BLAKE2b tree mode, `shardsum`) of every message, on the GPU."""
        n = len(msgs)
        if n == 0:
            return []
        bufs = [np.frombuffer(bytes(x), dtype=np.uint8) if len(x) else np.zeros(1, dtype=np.uint8) for x in msgs]
        ptrs = (ctypes.c_void_p * n)(*[a.ctypes.data for a in bufs])
        lens = (ctypes.c_size_t * n)(*[len(x) for x in msgs])
        out = np.empty((n, 32), dtype=np.uint8)
        check(lib.gec_shardsum_batch(self._h, n, ptrs, lens, _u8p(out)), "gec_shardsum_batch")
        return [out[i].tobytes() for i in range(n)]

    def shardsum_dev(self, t):
        """t: (n, len) uint8 CUDA tensor, rows 16-byte aligned -> (n, 32) uint8 tensor of shard checksums."""
        import torch

        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.uint8 and t.dim() == 2 and t.is_contiguous()):
            raise TypeError("expected a contiguous 2-D uint8 CUDA tensor")
        n, ln = t.shape
        out = torch.empty((n, 32), dtype=torch.uint8, device=t.device)
        check(lib.gec_shardsum_batch_dev(self._h, n, t.data_ptr(), ln, ln, out.data_ptr(), _stream_handle(self.device)),
              "gec_shardsum_batch_dev")
        return out

    def blake2sum_dev(self, t):
        """t: (n, len) uint8 CUDA tensor, rows 16-byte aligned -> (n, 32) uint8 tensor."""
        import torch

        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.uint8 and t.dim() == 2 and t.is_contiguous()):
            raise TypeError("expected a contiguous 2-D uint8 CUDA tensor")
        n, ln = t.shape
        out = torch.empty((n, 32), dtype=torch.uint8, device=t.device)
        check(lib.gec_blake2sum_batch_dev(self._h, n, t.data_ptr(), ln, ln, out.data_ptr(), _stream_handle(self.device)),
              "gec_blake2sum_batch_dev")
        return out

    def encode_hash_dev(self, stripes):
        """encode_dev + the blake2sum of all k+m shards of every stripe, device-resident
        (gec_encode_hash_batch_dev): -> (nblocks, k+m, 32) uint8 CUDA tensor of checksums."""
        import torch

        self._check_dev(stripes, self.n, "stripes")
        nb, _, S = stripes.shape
        sums = torch.empty((nb, self.n, 32), dtype=torch.uint8, device=stripes.device)
        check(lib.gec_encode_hash_batch_dev(self._h, nb, stripes.data_ptr(), self.n * S, S, sums.data_ptr(),
                                            _stream_handle(self.device)), "gec_encode_hash_batch_dev")
        return sums

    def verify(self, stripes: np.ndarray) -> np.ndarray:
        """stripes: (nblocks, k+m, S) host array -> bool (nblocks,)."""
        st = np.ascontiguousarray(stripes, dtype=np.uint8)
        if st.ndim != 3 or st.shape[1] != self.n:
            raise GecError(_lib.GEC_E_TOO_FEW_SHARDS if st.ndim == 3 and st.shape[1] < self.n else _lib.GEC_E_TOO_MANY_SHARDS,
                           "verify", f"expected (nblocks, {self.n}, S)")
        nb, _, S = st.shape
        ptrs = (ctypes.c_void_p * (nb * self.n))(*[st[b, j].ctypes.data for b in range(nb) for j in range(self.n)])
        ok = np.zeros(nb, dtype=np.uint8)
        check(lib.gec_verify_batch(self._h, nb, ptrs, S, _u8p(ok)), "gec_verify_batch")
        return ok.astype(bool)

    def verify_hash(self, stripes: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
        """The scrub check in one trip (gec_verify_hash_batch): stripes (nblocks, k+m, S) host array (pinned or not)
        -> (ok bool (nblocks,), shard checksums uint8 (nblocks, k+m, 32))."""
        st = stripes if stripes.flags["C_CONTIGUOUS"] and stripes.dtype == np.uint8 else np.ascontiguousarray(stripes, dtype=np.uint8)
        if st.ndim != 3 or st.shape[1] != self.n:
            raise GecError(_lib.GEC_E_TOO_FEW_SHARDS, "verify_hash", f"expected (nblocks, {self.n}, S)")
        nb, _, S = st.shape
        base = st.ctypes.data
        ptrs = (ctypes.c_void_p * (nb * self.n))(*[base + i * S for i in range(nb * self.n)])
        ok = np.zeros(nb, dtype=np.uint8)
        sums = np.zeros((nb, self.n, 32), dtype=np.uint8)
        check(lib.gec_verify_hash_batch(self._h, nb, ptrs, S, _u8p(ok), _u8p(sums)), "gec_verify_hash_batch")
        return ok.astype(bool), sums

    def reconstruct(self, shards: Sequence[Sequence[Optional[np.ndarray]]], data_only: bool = False):
        """shards[b][j] is a uint8 array of S bytes or None (missing).  Returns a
        list of lists with the missing entries filled (`reconstruct` /
        `reconstruct_data` when data_only)."""
        nb = len(shards)
        if nb == 0:
            return []
        S = None
        for row in shards:
            if len(row) != self.n:
                raise GecError(_lib.GEC_E_TOO_FEW_SHARDS if len(row) < self.n else _lib.GEC_E_TOO_MANY_SHARDS,
                               "reconstruct", f"each block needs {self.n} shard slots")
            for s in row:
                if s is not None:
                    if S is None:
                        S = len(s)
                    elif len(s) != S:
                        raise GecError(_lib.GEC_E_INCORRECT_SHARD_SIZE, "reconstruct", "shards differ in length")
        if S is None:
            raise GecError(_lib.GEC_E_TOO_FEW_PRESENT, "reconstruct", "no shard present")
        keep, inp, outp, result = [], [], [], []
        for row in shards:
            res_row = []
            for j, s in enumerate(row):
                if s is not None:
                    a = np.ascontiguousarray(s, dtype=np.uint8)
                    keep.append(a)
                    inp.append(a.ctypes.data)
                    outp.append(None)
                    res_row.append(a)
                else:
                    inp.append(None)
                    if data_only and j >= self.k:
                        outp.append(None)
                        res_row.append(None)
                    else:
                        o = np.empty(S, dtype=np.uint8)
                        keep.append(o)
                        outp.append(o.ctypes.data)
                        res_row.append(o)
            result.append(res_row)
        ip = (ctypes.c_void_p * len(inp))(*inp)
        op = (ctypes.c_void_p * len(outp))(*outp)
        check(lib.gec_reconstruct_batch(self._h, nb, ip, op, S, int(bool(data_only))), "gec_reconstruct_batch")
        return result

    def reconstruct_data(self, shards):
        return self.reconstruct(shards, data_only=True)


SHARDSUM_LEAF = 4096
_MLH_KEYS = None


def shardsum3(data: bytes) -> bytes:
    """Shard checksum v3 (GEC_SHARDSUM_MLH64, include/garage_ec.h) in numpy + hashlib: the host mirror of the definition.
    (The tests' oracle for it is oracle/mlh64.py, which shares nothing with this.)"""
    import hashlib
    import struct

    global _MLH_KEYS
    if _MLH_KEYS is None:
        M = (1 << 64) - 1
        keys = []
        for i in range(SHARDSUM_LEAF // 4):
            x = (0x6761726167654D4C + (i + 1) * 0x9E3779B97F4A7C15) & M
            x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & M
            x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & M
            keys.append(((x ^ (x >> 31)) >> 32) | 1)
        _MLH_KEYS = np.array(keys, dtype=np.uint64)
    buf = np.frombuffer(bytes(data), dtype=np.uint8)
    msg = [b"GECSUM3\0", struct.pack("<Q", buf.size)]
    for lo in range(0, buf.size, SHARDSUM_LEAF):
        leaf = buf[lo:lo + SHARDSUM_LEAF]
        if leaf.size % 4:
            leaf = np.concatenate([leaf, np.zeros(-leaf.size % 4, dtype=np.uint8)])
        w = leaf.view("<u4").astype(np.uint64)
        with np.errstate(over="ignore"):
            msg.append(struct.pack("<Q", int((w * _MLH_KEYS[:w.size]).sum(dtype=np.uint64))))
    return hashlib.blake2b(b"".join(msg), digest_size=64).digest()[:32]


def shardsum(data: bytes, kind: int = 3) -> bytes:
    """The shard checksum of the given kind, restated in Python (host mirror): kind 3 = MLH64 (the default of every codec),
    kind 2 = BLAKE2b tree mode, 4 KiB leaves, unlimited fanout, depth 2, 64-byte inner digests, root truncated to 32 bytes
    (include/garage_ec.h)."""
    import hashlib

    if kind == 3:
        return shardsum3(data)
    if kind != 2:
        raise ValueError("shard checksum kinds are 2 (BLAKE2b tree) and 3 (MLH64)")
    n = max(1, -(-len(data) // SHARDSUM_LEAF))
    leaves = b"".join(
        hashlib.blake2b(data[i * SHARDSUM_LEAF:(i + 1) * SHARDSUM_LEAF], digest_size=64, fanout=0, depth=2, leaf_size=SHARDSUM_LEAF,
                        node_offset=i, node_depth=0, inner_size=64, last_node=(i == n - 1)).digest() for i in range(n))
    return hashlib.blake2b(leaves, digest_size=64, fanout=0, depth=2, leaf_size=SHARDSUM_LEAF, node_offset=0, node_depth=1,
                           inner_size=64, last_node=True).digest()[:32]


def host_alloc(nbytes: int) -> np.ndarray:
    """A uint8 array over pinned host memory from gec_host_alloc: buffers handed to the host-pointer
    calls from such arrays go over PCIe without the staging copy.  Release with host_free(arr)."""
    p = lib.gec_host_alloc(nbytes)
    if not p:
        raise GecError(_lib.GEC_E_NOMEM, "gec_host_alloc", (lib.gec_last_error() or b"").decode("utf-8", "replace"))
    return np.ctypeslib.as_array((ctypes.c_uint8 * max(nbytes, 1)).from_address(p))[:nbytes]


def host_free(arr: np.ndarray) -> None:
    lib.gec_host_free(ctypes.c_void_p(arr.ctypes.data))


def set_kernel_variant(v: int) -> None:
    check(lib.gec_set_kernel_variant(v), "gec_set_kernel_variant")
