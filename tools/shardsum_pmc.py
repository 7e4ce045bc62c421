#!/usr/bin/env python3
"""Ten shard-checksum launches over one 1024-stripe batch (14336 shards of 104896 bytes), for rocprofv3 --pmc passes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import garage_amd as g  # noqa: E402

rs = g.ReedSolomon(10, 4)
t = torch.randint(0, 256, (1024 * 14, 104896), dtype=torch.uint8, device="cuda:0")
for _ in range(10):
    rs.shardsum_dev(t)
torch.cuda.synchronize()
print("ok")
