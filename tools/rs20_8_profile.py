#!/usr/bin/env python3
"""The RS(20,8) encode kernel at BASELINE config 5's shape (4 MiB blocks, 256 of them, device-resident) for rocprofv3:
50 back-to-back launches after a warm-up burst.  usage: rs20_8_profile.py [launches]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import garage_amd as g  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    rs = g.ReedSolomon(20, 8)
    S = g.shard_len(20, 4 << 20)
    st = torch.randint(0, 256, (256, 28, S), dtype=torch.uint8, device="cuda:0")
    for _ in range(200):       # the power controller's transient
        rs.encode_dev(st)
    torch.cuda.synchronize()
    for _ in range(n):
        rs.encode_dev(st)
    torch.cuda.synchronize()
    print("algorithmic bytes per launch:", 256 * 28 * S)


if __name__ == "__main__":
    main()
