#!/usr/bin/env python3
"""Soak: sustained encode / verify / reconstruct on fresh random data, every result checked
(verify flags + byte-compare of rebuilt shards).  usage: soak.py [seconds]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import garage_amd as g  # noqa: E402


def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
    dev = "cuda:0"
    rng = np.random.default_rng(11)
    codes = [(10, 4, 1 << 20, 256), (20, 8, 4 << 20, 64), (3, 1, 65536, 2048), (6, 3, 300000, 512)]
    rs = {c[:2]: g.ReedSolomon(c[0], c[1]) for c in codes}
    t0, it, launches = time.time(), 0, 0
    while time.time() - t0 < secs:
        k, m, L, nb = codes[it % len(codes)]
        S = g.shard_len(k, L)
        st = torch.randint(0, 256, (nb, k + m, S), dtype=torch.uint8, device=dev)
        c = rs[(k, m)]
        for _ in range(20):
            c.encode_dev(st)
        launches += 20
        assert bool(c.verify_dev(st).all()), f"verify failed at iteration {it}"
        ref = st.clone()
        nlost = int(rng.integers(1, m + 1))
        lost = sorted(rng.choice(k + m, size=nlost, replace=False).tolist())
        st[:, lost] = 0x5A
        c.reconstruct_dev(st, [j not in lost for j in range(k + m)])
        assert torch.equal(st, ref), f"reconstruct mismatch at iteration {it}, lost {lost}"
        # a flipped byte must be caught, and only in its block
        b, j, off = int(rng.integers(nb)), int(rng.integers(k + m)), int(rng.integers(S))
        st[b, j, off] ^= 1 << int(rng.integers(8))
        ok = c.verify_dev(st)
        assert not bool(ok[b]) and int((~ok).sum()) == 1, f"corruption not localised at iteration {it}"
        it += 1
    torch.cuda.synchronize()
    print(f"soak OK: {it} iterations, {launches} encode launches, {time.time() - t0:.1f} s, 0 mismatches")


if __name__ == "__main__":
    main()
