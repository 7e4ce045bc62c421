#!/usr/bin/env python3
"""One BLAKE2b chain's speed: plain blake2sum of n device-resident messages of `len` bytes (quad kernel below 40000
messages): ms per launch and ns per 128-byte compression.  usage: chain_bench.py [n] [len]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import hashlib  # noqa: E402

import torch  # noqa: E402

import garage_amd as g  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ln = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
rs = g.ReedSolomon(10, 4)
x = torch.randint(0, 256, (n, ln), dtype=torch.uint8, device="cuda:0")
out = rs.blake2sum_dev(x)
assert out[3].cpu().numpy().tobytes() == hashlib.blake2b(x[3].cpu().numpy().tobytes(), digest_size=64).digest()[:32]
ts = []
for _ in range(7):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    rs.blake2sum_dev(x)
    b.record()
    torch.cuda.synchronize()
    ts.append(a.elapsed_time(b))
ms = min(ts)
print(f"{n} messages x {ln} B: {ms:.3f} ms per launch, {ms * 1e6 / ((ln + 127) // 128):.0f} ns per compression, {n * ln / ms / 1e6:.1f} GB/s")
