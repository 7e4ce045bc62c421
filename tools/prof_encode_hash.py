#!/usr/bin/env python3
"""A few launches of each device-resident put-path kernel for rocprofv3 (kernel trace or one PMC pass at a time):
RS(10,4) x 1024 x 1 MiB: encode, encode + 14 checksums (v3), checksums alone; RS(20,8) x 256 x 4 MiB: encode, encode + checksums."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import garage_amd as g  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for (k, m, L, nb) in ((10, 4, 1 << 20, 1024), (20, 8, 4 << 20, 256)):
    S = g.shard_len(k, L)
    st = torch.randint(0, 256, (nb, k + m, S), dtype=torch.uint8, device="cuda:0")
    rs = g.ReedSolomon(k, m)
    for _ in range(reps):
        rs.encode_dev(st)
    torch.cuda.synchronize()
    for _ in range(reps):
        rs.encode_hash_dev(st)
    torch.cuda.synchronize()
    for _ in range(reps):
        rs.shardsum_dev(st.view(nb * (k + m), S))
    torch.cuda.synchronize()
print("done")
