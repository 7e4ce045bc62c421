"""Hash partitioning of blocks over the GPUs of one node.

Garage already places by hash bits at three levels: cluster partition = top 8
bits (src/rpc/layout/version.rs:101-104), drive = bytes 2-3 mod 1024
(src/block/layout.rs:278-284), mutation lock = bytes 0-1 mod 256
(src/block/manager.rs:679-689).  The GPU is chosen from byte 4 so that it is
independent of node and drive choice.
Blocks are independent units, so this needs no data-path collective.
"""
from __future__ import annotations

import hashlib

import numpy as np

_MASK64 = (1 << 64) - 1


def block_hash(data: bytes) -> bytes:
    """Garage's `blake2sum`: blake2b-512 truncated to 32 bytes
    (src/util/data.rs:130-138) -- NOT blake2b-256."""
    return hashlib.blake2b(data, digest_size=64).digest()[:32]


def gpu_of_hash(hashes, n_gpus: int):
    """hashes: (N, 32) uint8 array or a single 32-byte hash -> GPU index.

    The rule itself lives in the C ABI (`gec_device_of_hash`, include/garage_ec.h): this function
    only calls it, so that libgarage_block's multi-device manager, bench.py and the Python side
    cannot drift apart."""
    from ._lib import lib

    if n_gpus < 1:
        raise ValueError("n_gpus must be >= 1")
    if isinstance(hashes, (bytes, bytearray)):
        if len(hashes) != 32:
            raise ValueError("a block hash is 32 bytes")
        return lib.gec_device_of_hash(bytes(hashes), n_gpus)
    h = np.ascontiguousarray(hashes, dtype=np.uint8)
    if h.shape[-1] != 32:
        raise ValueError("a block hash is 32 bytes")
    # arrays: the rule written out (hash[4] % n) -- one vectorised expression instead of a C call per row (a PutObject batch or a
    # bench input is 1e5-1e6 hashes); tests/test_multi_device.py pins it to gec_device_of_hash on every row of a sample
    return (h[..., 4].astype(np.int64) % n_gpus)


def partition(hashes, n_gpus: int) -> list[np.ndarray]:
    """Indices of the blocks each GPU owns, in stream order."""
    owner = gpu_of_hash(hashes, n_gpus)
    return [np.nonzero(owner == r)[0] for r in range(n_gpus)]


def splitmix64_bytes(seed: int, nbytes: int) -> np.ndarray:
    """SplitMix64 stream as little-endian u64 (synthetic payloads / hashes)."""
    n = -(-nbytes // 8)
    with np.errstate(over="ignore"):
        idx = np.arange(1, n + 1, dtype=np.uint64)
        z = np.uint64(seed & _MASK64) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z.astype("<u8").view(np.uint8)[:nbytes].copy()
