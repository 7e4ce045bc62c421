"""Shard checksum v3 ("MLH64 tree"), restated independently in Python integers.  TEST INFRASTRUCTURE, like everything under
oracle/: only tests/, __graft_entry__.smoke() and bench.py's checks may import it; the product computes the checksum in
garage_amd/csrc/mlh64_host.hpp (host) and mlh64_dev.hpp (gfx950) and never calls this.

The definition is this project's own (shards are its storage format; Garage has none -- the reference's analogue is what
DataBlock::verify accepts as a block's integrity check, /root/reference/src/block/block.rs:69-83), written down in
garage_amd/csrc/mlh64.hpp.  This file shares no code and no tables with it:

    K[i]  = (splitmix64(SEED + (i+1)*GOLDEN) >> 32) | 1,   i < 1024
    s_l   = sum_{i<1024} K[i] * u32le(shard[4096 l + 4 i : +4])  mod 2^64          (shard zero-extended)
    sum   = blake2b-512(b"GECSUM3\\0" + u64le(len) + u64le(s_0) + ... )[:32]
"""
from __future__ import annotations

import hashlib
import struct

import numpy as np

MASK = (1 << 64) - 1
SEED = int.from_bytes(b"garageML", "big")   # 0x6761726167654d4c
GOLDEN = 0x9E3779B97F4A7C15
LEAF = 4096


def _splitmix64(x: int) -> int:
    x &= MASK
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & MASK
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & MASK
    return x ^ (x >> 31)


KEYS = [(_splitmix64(SEED + (i + 1) * GOLDEN) >> 32) | 1 for i in range(LEAF // 4)]
_KEYS_NP = np.array(KEYS, dtype=np.uint64)


def leaf_sums_slow(data: bytes) -> list[int]:
    """pure Python loops: the definition, for small cases"""
    out = []
    for lo in range(0, len(data), LEAF):
        leaf = data[lo:lo + LEAF]
        leaf += b"\0" * (-len(leaf) % 4)
        s = 0
        for i in range(len(leaf) // 4):
            s += KEYS[i] * int.from_bytes(leaf[4 * i:4 * i + 4], "little")
        out.append(s & MASK)
    return out


def leaf_sums(data) -> list[int]:
    """numpy: uint64 products of 32-bit factors never overflow; the sum wraps mod 2^64 as the definition says"""
    b = np.frombuffer(bytes(data), dtype=np.uint8)
    out = []
    for lo in range(0, b.size, LEAF):
        leaf = b[lo:lo + LEAF]
        if leaf.size % 4:
            leaf = np.concatenate([leaf, np.zeros(-leaf.size % 4, dtype=np.uint8)])
        w = leaf.view("<u4").astype(np.uint64)
        with np.errstate(over="ignore"):
            out.append(int((w * _KEYS_NP[:w.size]).sum(dtype=np.uint64)))
    return out


def root(length: int, sums) -> bytes:
    msg = b"GECSUM3\0" + struct.pack("<Q", length) + b"".join(struct.pack("<Q", s) for s in sums)
    return hashlib.blake2b(msg, digest_size=64).digest()[:32]


def shardsum3(data) -> bytes:
    data = bytes(data)
    return root(len(data), leaf_sums(data))


def shardsum3_slow(data: bytes) -> bytes:
    return root(len(data), leaf_sums_slow(data))
