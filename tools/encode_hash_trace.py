#!/usr/bin/env python3
"""Five gec_encode_hash_batch_dev calls on one 1024-stripe batch, meant to run under
`rocprofv3 --kernel-trace` to see which kernels overlap (tools/rocprof_summary.py prints the timeline)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import garage_amd as g  # noqa: E402

rs = g.ReedSolomon(10, 4)
st = torch.randint(0, 256, (1024, 14, 104896), dtype=torch.uint8, device="cuda:0")
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(5):
        rs.encode_hash_dev(st)
    s.synchronize()
print("ok")
