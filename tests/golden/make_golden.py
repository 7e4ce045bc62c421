#!/usr/bin/env python3
"""Generates tests/golden/rs_golden.json: SHA-256 digests of parity and of
reconstructed shards for BASELINE-sized inputs, computed with the CPU oracle
(oracle/rs_oracle.c, AVX2 and scalar paths must agree) on SplitMix64 payloads.

The reference tree holds no vectors for this path and cannot be executed here
(SURVEY.md section 8c), so these fixtures are oracle outputs, pinned in turn by the
upstream known-answer vectors in tests/test_oracle_kat.py.  They let the GPU
tests check full-size configurations byte-for-byte without running the oracle.

usage: python tests/golden/make_golden.py   (rewrites rs_golden.json)
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import mlh64
from oracle import rs_oracle as O  # noqa: E402

SEED = 0x6761726167650001


def sha(a) -> str:
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.uint8).tobytes()).hexdigest()


def case(co, name, cfg, k, m, L, nb, lost):
    S = O.shard_len(k, L)
    data = np.zeros((nb, k * S), dtype=np.uint8)
    payload = O.splitmix64_bytes(SEED + cfg, nb * L).reshape(nb, L)
    data[:, :L] = payload
    data[0, :L] = 0            # one all-zero block
    if nb > 1:
        data[1, :L] = 0xFF     # one all-0xFF block
    data = data.reshape(nb, k, S)
    par = co.encode_batch(k, m, data, co.AVX2, threads=8)
    par_s = co.encode_batch(k, m, data[:1], co.SCALAR, threads=1)
    assert np.array_equal(par[:1], par_s)
    full = np.concatenate([data, par], axis=1)
    broken = full.copy()
    broken[:, list(lost)] = 0
    present = [j not in lost for j in range(k + m)]
    rec = co.reconstruct_batch(k, m, broken, present, threads=8)
    assert np.array_equal(rec, full)
    valid, D = O.decode_matrix(k, m, present)
    return {
        "name": name, "config": cfg, "k": k, "m": m, "block_len": L, "shard_len": S, "nblocks": nb,
        "seed": SEED + cfg, "payload_sha256": sha(data),
        "parity_sha256": sha(par), "parity_first16": par[min(2, nb - 1), 0, :16].tolist(),
        "lost": list(lost), "valid": valid, "decode_matrix_sha256": sha(D),
        "reconstructed_sha256": sha(rec[:, list(lost)]),
    }


SHARDSUM_LEAF = 4096


def shardsum_hashlib(data: bytes) -> bytes:
    """The shard checksum written out with hashlib only (BLAKE2b tree mode; include/garage_ec.h)."""
    n = max(1, -(-len(data) // SHARDSUM_LEAF))
    leaves = b"".join(
        hashlib.blake2b(data[i * SHARDSUM_LEAF:(i + 1) * SHARDSUM_LEAF], digest_size=64, fanout=0, depth=2, leaf_size=SHARDSUM_LEAF,
                        node_offset=i, node_depth=0, inner_size=64, last_node=(i == n - 1)).digest() for i in range(n))
    return hashlib.blake2b(leaves, digest_size=64, fanout=0, depth=2, leaf_size=SHARDSUM_LEAF, node_offset=0, node_depth=1,
                           inner_size=64, last_node=True).digest()[:32]


def checksum_cases():
    """Known answers for the two hashes of row f4: Garage's blake2sum (plain) and the shard checksum (tree mode), on
    SplitMix64 messages -- generated with hashlib, the RFC 7693 reference implementation in CPython."""
    out = []
    for i, n in enumerate([0, 1, 3, 64, 127, 128, 129, 4095, 4096, 4097, 8192, 12289, 104896, 209728, (1 << 20) + 3]):
        msg = bytes(O.splitmix64_bytes(SEED + 100 + i, n))
        out.append({"len": n, "seed": SEED + 100 + i, "blake2sum": hashlib.blake2b(msg, digest_size=64).digest()[:32].hex(),
                    "shardsum": shardsum_hashlib(msg).hex(),            # version 2: BLAKE2b tree mode
                    "shardsum3": mlh64.shardsum3_slow(msg).hex()})      # version 3: MLH64, the pure-Python definition (oracle/mlh64.py)
    return out


def main():
    co = O.COracle()
    cases = [
        case(co, "config1_rs3_1_64KiB", 1, 3, 1, 65536, 16, (1,)),
        case(co, "config2_3_rs10_4_1MiB", 2, 10, 4, 1 << 20, 8, (0, 3, 7, 9)),
        case(co, "config3_mixed_rs10_4_1MiB", 3, 10, 4, 1 << 20, 4, (0, 3, 7, 11)),
        case(co, "config5_rs20_8_4MiB", 5, 20, 8, 4 << 20, 2, (0, 1, 5, 9, 13, 19, 21, 27)),
        case(co, "ragged_rs10_4_999999", 6, 10, 4, 999_999, 3, (9, 13)),
        # the headline batch itself (BASELINE config 2 / 3): 1024 blocks of 1 MiB, one digest over all parity
        case(co, "config2_full_batch_rs10_4_1MiB_x1024", 7, 10, 4, 1 << 20, 1024, (0, 3, 7, 9)),
    ]
    with open(os.path.join(HERE, "rs_golden.json"), "w") as f:
        json.dump({"generator": "tests/golden/make_golden.py (CPU oracle; checksums: hashlib)", "cases": cases,
                   "checksums": checksum_cases()}, f, indent=1)
    print("wrote", len(cases), "cases")


if __name__ == "__main__":
    main()
