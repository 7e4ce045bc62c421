"""ctypes binding of libgarage_ec.so (include/garage_ec.h).

The library is the product; this module only declares prototypes.  There is no
Python fallback of any kind: if the shared object is missing, import fails loudly
with the command that builds it.  (The library itself has two backends, chosen
per codec: GEC_BACKEND_HIP -- the gfx950 kernels -- and GEC_BACKEND_CPU, its own
host-core data path for nodes without a GPU.)
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgarage_ec.so")

GEC_OK = 0
GEC_E_TOO_FEW_SHARDS = -1
GEC_E_TOO_MANY_SHARDS = -2
GEC_E_TOO_FEW_DATA = -3
GEC_E_TOO_MANY_DATA = -4
GEC_E_TOO_FEW_PARITY = -5
GEC_E_TOO_MANY_PARITY = -6
GEC_E_INCORRECT_SHARD_SIZE = -7
GEC_E_TOO_FEW_PRESENT = -8
GEC_E_EMPTY_SHARD = -9
GEC_E_INVALID_INDEX = -10
GEC_E_DEVICE = -100
GEC_E_NOMEM = -101
GEC_E_INVALID_ARG = -102
GEC_MATRIX_VANDERMONDE, GEC_MATRIX_CAUCHY = 0, 1
GEC_BACKEND_CPU, GEC_BACKEND_HIP, GEC_BACKEND_AUTO = 0, 1, 2
GEC_CLASS_FOREGROUND, GEC_CLASS_BACKGROUND = 0, 1
GEC_SHARDSUM_DEFAULT, GEC_SHARDSUM_BLAKE2B_TREE, GEC_SHARDSUM_MLH64 = 0, 2, 3

# every symbol include/garage_ec.h declares (tests/test_cabi_host.py::test_every_declared_symbol_is_exported checks
# this list against the header and against the built library)
SYMBOLS = [
    "gec_version", "gec_device_count", "gec_device_of_hash", "gec_thread_link_release", "gec_strerror", "gec_last_error", "gec_env_table",
    "gec_cpu_isa",
    "gec_shard_len", "gec_build_matrix", "gec_build_matrix_ex", "gec_build_decode_matrix",
    "gec_codec_create", "gec_codec_create_ex", "gec_codec_create_ex2", "gec_codec_with_shardsum", "gec_codec_shardsum", "gec_shardsum_host",
    "gec_codec_destroy", "gec_codec_k", "gec_codec_m",
    "gec_codec_device", "gec_parity_matrix", "gec_codec_cache_stats",
    "gec_codec_background", "gec_codec_class", "gec_codec_backend", "gec_qos_yields", "gec_cu_masks_active",
    "gec_encode_batch", "gec_verify_batch", "gec_verify_hash_batch", "gec_reconstruct_batch", "gec_reconstruct_hash_batch",
    "gec_encode_batch_dev", "gec_verify_batch_dev", "gec_reconstruct_batch_dev", "gec_reconstruct_batch_dev_ex",
    "gec_reconstruct_range_dev", "gec_reconstruct_scattered_dev", "gec_blake2sum_batch_dev", "gec_blake2sum_batch",
    "gec_encode_hash_batch", "gec_set_kernel_variant", "gec_get_kernel_variant",
    "gec_group_unique_id", "gec_group_create", "gec_group_create_with_transport", "gec_group_destroy",
    "gec_group_rank", "gec_group_size", "gec_group_slots", "gec_group_allgather_decode",
    "gec_group_create_with_transport2", "gec_group_alltoall_decode", "gec_group_bytes_exchanged",
    "gec_group_peer_decode", "gec_ipc_export", "gec_ipc_open", "gec_ipc_close",
    "gec_launch_geometry",
    "gec_encode_hash_batch_dev", "gec_decode_verify_batch", "gec_shardsum_batch", "gec_shardsum_batch_dev", "gec_host_alloc", "gec_host_free", "gec_host_register", "gec_host_unregister", "gec_host_is_pinned",
    "gec_codec_numa_node", "gec_codec_numa_cpus", "gec_host_alloc_near", "gec_numa_bind_thread", "gec_numa_node_of",
]
GEC_GROUP_ID_BYTES = 128
GEC_IPC_HANDLE_BYTES = 72
# int (*gec_allgather_fn)(void *ctx, const void *d_send, void *d_recv, size_t bytes, void *hip_stream)
ALLGATHER_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p)


class GecError(RuntimeError):
    """A non-zero return code from libgarage_ec (mirrors reed_solomon_erasure::Error)."""

    def __init__(self, code: int, what: str, detail: str):
        super().__init__(f"{what}: {detail} (code {code})" if detail else f"{what} (code {code})")
        self.code = code


def _load() -> ctypes.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: libgarage_ec has not been built. Run "
            "`make -C garage_amd/csrc` (or `python -c 'import __graft_entry__ as g; g.build()'`). "
            "Nothing in Python stands in for it."
        )
    # One HIP runtime per process: PyTorch's ROCm wheel bundles its own
    # libamdhip64.so (SONAME libamdhip64.so.7).  If libgarage_ec were loaded first it
    # would pull /opt/rocm's copy through its RUNPATH, torch would then load its
    # bundled one as a SECOND runtime, and whichever initialises second sees "no
    # GPUs".  Importing torch first makes the dynamic linker satisfy our NEEDED
    # libamdhip64.so.7 with the already-loaded runtime.  (A C/Rust host without
    # torch simply gets /opt/rocm's runtime.)
    try:
        import torch  # noqa: F401

        # same reasoning for RCCL (gec_group_*, resolved with dlopen on first use): inside a
        # torch process it must be the RCCL that is linked against torch's HIP runtime
        bundled = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        if os.path.exists(bundled):
            os.environ.setdefault("GEC_RCCL_LIB", bundled)
    except ImportError:  # pragma: no cover - torch is plumbing, not required to load
        pass
    lib = ctypes.CDLL(LIB_PATH)
    vp, sz, ci = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
    u8p = ctypes.POINTER(ctypes.c_uint8)
    pp = ctypes.POINTER(ctypes.c_void_p)
    lib.gec_version.restype = ctypes.c_uint32
    lib.gec_device_count.restype = ci
    lib.gec_device_of_hash.argtypes = [ctypes.c_char_p, ci]
    lib.gec_device_of_hash.restype = ci
    lib.gec_strerror.restype = ctypes.c_char_p
    lib.gec_strerror.argtypes = [ci]
    lib.gec_last_error.restype = ctypes.c_char_p
    lib.gec_shard_len.restype = sz
    lib.gec_shard_len.argtypes = [ci, sz]
    lib.gec_build_matrix.argtypes = [ci, ci, u8p]
    lib.gec_build_matrix_ex.argtypes = [ci, ci, ci, u8p]
    lib.gec_codec_create_ex.argtypes = [ci, ci, ci, ci, ci, pp]
    lib.gec_codec_background.argtypes = [vp, pp]
    lib.gec_codec_create_ex2.argtypes = [ci, ci, ci, ci, ci, ci, pp]
    lib.gec_codec_with_shardsum.argtypes = [vp, ci, pp]
    lib.gec_codec_shardsum.argtypes = [vp]
    lib.gec_shardsum_host.argtypes = [ci, ctypes.c_char_p, sz, ctypes.c_char_p]
    lib.gec_env_table.restype = ctypes.c_char_p
    lib.gec_cpu_isa.restype = ctypes.c_char_p
    lib.gec_qos_yields.argtypes = [ci]
    lib.gec_qos_yields.restype = ctypes.c_uint64
    lib.gec_build_decode_matrix.argtypes = [ci, ci, u8p, ctypes.POINTER(ctypes.c_int32), u8p]
    lib.gec_codec_create.argtypes = [ci, ci, ci, ci, pp]
    lib.gec_codec_destroy.argtypes = [vp]
    lib.gec_codec_destroy.restype = None
    for f in ("gec_codec_k", "gec_codec_m", "gec_codec_device", "gec_codec_class", "gec_codec_backend"):
        getattr(lib, f).argtypes = [vp]
    lib.gec_parity_matrix.argtypes = [vp, u8p]
    lib.gec_codec_cache_stats.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]
    lib.gec_encode_batch.argtypes = [vp, sz, pp, ctypes.POINTER(sz), sz, pp]
    lib.gec_verify_batch.argtypes = [vp, sz, pp, sz, u8p]
    lib.gec_verify_hash_batch.argtypes = [vp, sz, pp, sz, u8p, u8p]
    lib.gec_reconstruct_batch.argtypes = [vp, sz, pp, pp, sz, ci]
    lib.gec_reconstruct_hash_batch.argtypes = [vp, sz, pp, pp, sz, ci, u8p, u8p]
    lib.gec_encode_batch_dev.argtypes = [vp, sz, vp, sz, sz, vp, sz, vp]
    lib.gec_verify_batch_dev.argtypes = [vp, sz, vp, sz, sz, vp, vp]
    lib.gec_reconstruct_batch_dev.argtypes = [vp, sz, vp, sz, sz, u8p, ci, vp]
    lib.gec_reconstruct_batch_dev_ex.argtypes = [vp, sz, vp, sz, sz, u8p, ci, vp]
    lib.gec_reconstruct_range_dev.argtypes = [vp, sz, vp, sz, sz, u8p, ci, sz, sz, vp]
    lib.gec_reconstruct_scattered_dev.argtypes = [vp, sz, vp, sz, ctypes.POINTER(sz), sz, u8p, ci, sz, sz, vp]
    lib.gec_blake2sum_batch_dev.argtypes = [vp, sz, vp, sz, sz, vp, vp]
    lib.gec_blake2sum_batch.argtypes = [vp, sz, pp, ctypes.POINTER(sz), u8p]
    lib.gec_shardsum_batch.argtypes = [vp, sz, pp, ctypes.POINTER(sz), u8p]
    lib.gec_shardsum_batch_dev.argtypes = [vp, sz, vp, sz, sz, vp, vp]
    lib.gec_encode_hash_batch.argtypes = [vp, sz, pp, ctypes.POINTER(sz), sz, pp, u8p]
    lib.gec_encode_hash_batch_dev.argtypes = [vp, sz, vp, sz, sz, vp, vp]
    lib.gec_decode_verify_batch.argtypes = [vp, sz, pp, sz, ctypes.POINTER(sz), pp, u8p, u8p]
    lib.gec_set_kernel_variant.argtypes = [ci]
    ip = ctypes.POINTER(ci)
    lib.gec_launch_geometry.argtypes = [ci, ci, ip, ip, ip, ip, ctypes.POINTER(sz)]
    lib.gec_host_alloc.argtypes = [sz]
    lib.gec_host_alloc.restype = vp
    lib.gec_host_free.argtypes = [vp]
    lib.gec_host_free.restype = None
    lib.gec_host_register.argtypes = [vp, sz]
    lib.gec_host_unregister.argtypes = [vp]
    lib.gec_host_is_pinned.argtypes = [vp, sz]
    lib.gec_codec_numa_node.argtypes = [vp]
    lib.gec_codec_numa_cpus.argtypes = [vp, sz, ctypes.POINTER(ci), ctypes.POINTER(sz)]
    lib.gec_host_alloc_near.argtypes = [vp, sz]
    lib.gec_host_alloc_near.restype = vp
    lib.gec_numa_bind_thread.argtypes = [vp]
    lib.gec_numa_node_of.argtypes = [vp]
    lib.gec_group_unique_id.argtypes = [u8p]
    lib.gec_group_create.argtypes = [vp, ci, ci, u8p, pp]
    lib.gec_group_create_with_transport.argtypes = [vp, ci, ci, vp, vp, pp]
    lib.gec_group_destroy.argtypes = [vp]
    lib.gec_group_destroy.restype = None
    lib.gec_group_rank.argtypes = [vp]
    lib.gec_group_size.argtypes = [vp]
    lib.gec_group_slots.argtypes = [vp]
    lib.gec_group_slots.restype = sz
    lib.gec_group_allgather_decode.argtypes = [vp, sz, vp, sz, u8p, ci, ci, vp, vp]
    lib.gec_group_create_with_transport2.argtypes = [vp, ci, ci, vp, vp, vp, pp]
    lib.gec_group_alltoall_decode.argtypes = [vp, sz, vp, sz, u8p, ci, ci, vp, vp]
    lib.gec_group_bytes_exchanged.argtypes = [vp]
    lib.gec_group_peer_decode.argtypes = [vp, sz, pp, sz, u8p, ci, ci, vp, vp]
    lib.gec_ipc_export.argtypes = [vp, ctypes.c_char_p]
    lib.gec_ipc_open.argtypes = [ctypes.c_char_p, ci, pp]
    lib.gec_ipc_close.argtypes = [vp]
    lib.gec_group_bytes_exchanged.restype = ctypes.c_uint64
    return lib


lib = _load()


def check(rc: int, what: str) -> None:
    if rc != GEC_OK:
        detail = (lib.gec_last_error() or b"").decode("utf-8", "replace")
        name = (lib.gec_strerror(rc) or b"").decode()
        raise GecError(rc, f"{what}: {name}", detail)
