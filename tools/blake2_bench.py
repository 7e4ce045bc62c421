#!/usr/bin/env python3
"""GPU blake2sum rate on shard-shaped batches (device-resident, HIP events)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import garage_amd as g  # noqa: E402


def main():
    rs = g.ReedSolomon(10, 4)
    S = 104896
    res = {}
    for nblocks in (64, 256, 1024, 4096):
        n = nblocks * 14
        t = torch.randint(0, 256, (n, S), dtype=torch.uint8, device="cuda:0")
        for _ in range(2):
            rs.blake2sum_dev(t)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 5
        e0.record()
        for _ in range(reps):
            rs.blake2sum_dev(t)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        res[f"{nblocks}_stripes_{n}_shards"] = {"ms": round(ms, 3), "GBps": round(n * S / ms / 1e6, 1)}
        del t
    # encode + checksums of all 14 shards of every stripe, device-resident (gec_encode_hash_batch_dev): the data
    # shards' checksums run on a second stream beside the RS kernel
    for nblocks in (256, 1024):
        st = torch.randint(0, 256, (nblocks, 14, S), dtype=torch.uint8, device="cuda:0")
        for _ in range(3):
            rs.encode_hash_dev(st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10
        e0.record()
        for _ in range(reps):
            rs.encode_hash_dev(st)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        e0.record()
        for _ in range(reps):
            rs.encode_dev(st)
        e1.record()
        torch.cuda.synchronize()
        ms_enc = e0.elapsed_time(e1) / reps
        res[f"encode_hash_dev_{nblocks}_stripes"] = {"ms": round(ms, 3), "encode_only_ms": round(ms_enc, 3),
                                                      "hashed_GBps": round(nblocks * 14 * S / ms / 1e6, 1)}
        del st
    print(json.dumps({"what": "GPU blake2sum of 104896-byte shards, device-resident, kernel = " + "auto (quad < 40000 messages <= lane)", "results": res}))


if __name__ == "__main__":
    main()
