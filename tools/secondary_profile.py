#!/usr/bin/env python3
"""A few launches of every kernel that is NOT the headline one, for rocprofv3:
RS(20,8) 8-row encode, verify, 4-erasure decode, both blake2 kernels, clear_flags."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import garage_amd as g  # noqa: E402


def main():
    dev = "cuda:0"
    rs = g.ReedSolomon(20, 8)
    S = g.shard_len(20, 4 << 20)
    st = torch.randint(0, 256, (256, 28, S), dtype=torch.uint8, device=dev)
    for _ in range(5):
        rs.encode_dev(st)
    for _ in range(3):
        assert bool(rs.verify_dev(st).all())
    lost = (0, 1, 5, 9, 13, 19, 21, 27)
    pres = [j not in lost for j in range(28)]
    for _ in range(3):
        rs.reconstruct_dev(st, pres)
    rs10 = g.ReedSolomon(10, 4)
    S1 = g.shard_len(10, 1 << 20)
    t = torch.randint(0, 256, (1024 * 14, S1), dtype=torch.uint8, device=dev)
    for _ in range(3):
        rs10.blake2sum_dev(t)          # quad kernel (14336 messages)
    t4 = torch.randint(0, 256, (4096 * 14, S1), dtype=torch.uint8, device=dev)
    for _ in range(3):
        rs10.blake2sum_dev(t4)         # one-lane kernel (57344 messages)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
