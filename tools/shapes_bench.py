#!/usr/bin/env python3
"""Encode rate of the product path (device API) across code shapes and block sizes:
checks that the launch heuristics (load-batch size, table width, tile width) hold up
away from the headline RS(10,4) / 1 MiB configuration."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import garage_amd as g  # noqa: E402

SHAPES = [(3, 1), (4, 2), (6, 3), (8, 4), (10, 4), (12, 4), (16, 4), (17, 3), (20, 4), (10, 8), (20, 8), (32, 8), (64, 16)]
BLOCKS = [(1 << 20, "1MiB"), (64 << 10, "64KiB"), (4 << 20, "4MiB")]


def main():
    rows = []
    for k, m in SHAPES:
        rs = g.ReedSolomon(k, m)
        for L, tag in BLOCKS:
            if tag != "1MiB" and (k, m) not in ((10, 4), (3, 1), (20, 8)):
                continue
            S = g.shard_len(k, L)
            nb = max(8, (1 << 30) // L)
            st = torch.randint(0, 256, (nb, k + m, S), dtype=torch.uint8, device="cuda:0")
            for _ in range(30):
                rs.encode_dev(st)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 60
            e0.record()
            for _ in range(reps):
                rs.encode_dev(st)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            algo = (k + m) * S * nb
            rows.append({"code": f"RS({k},{m})", "block": tag, "nblocks": nb, "us": round(ms * 1e3, 1),
                         "payload_GiBps": round(nb * L / ms / 2**30 * 1e3, 1), "algorithmic_GBps": round(algo / ms / 1e6, 1),
                         "frac_of_8TBps": round(algo / ms / 1e6 / 8000, 3)})
            del st
        rs.close()
    print(f"{'code':>10} {'block':>6} {'nblocks':>7} {'us':>8} {'payload GiB/s':>14} {'algo GB/s':>10} {'of 8 TB/s':>9}")
    for r in rows:
        print(f"{r['code']:>10} {r['block']:>6} {r['nblocks']:>7} {r['us']:>8} {r['payload_GiBps']:>14} {r['algorithmic_GBps']:>10} {r['frac_of_8TBps']:>9}")
    print(json.dumps(rows))


if __name__ == "__main__":
    main()
