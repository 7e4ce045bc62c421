#!/usr/bin/env python3
"""Device-resident put trip on BASELINE config 2 (RS(10,4), 1024 x 1 MiB): the encode alone, the encode with all 14 shard
checksums (gec_encode_hash_batch_dev) and the checksums alone (gec_shardsum_batch_dev), for both checksum kinds;
RS(20,8) x 256 x 4 MiB beside it.  HIP events on the launch stream, warmed clocks (200 ms of the same launches first)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import garage_amd as g  # noqa: E402


def timed(fn, reps=50, warm_ms=200):
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < warm_ms:
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    res = {}
    for (k, m, L, nb) in ((10, 4, 1 << 20, 1024), (20, 8, 4 << 20, 256)):
        S = g.shard_len(k, L)
        st = torch.randint(0, 256, (nb, k + m, S), dtype=torch.uint8, device="cuda:0")
        algo = (k + m) * S * nb
        row = {"S": S, "blocks": nb, "algorithmic_bytes": algo}
        for kind in (3, 2):
            rs = g.ReedSolomon(k, m, shardsum=kind)
            enc = timed(lambda: rs.encode_dev(st))
            eh = timed(lambda: rs.encode_hash_dev(st))
            hs = timed(lambda: rs.shardsum_dev(st.view(nb * (k + m), S)))
            row[f"kind{kind}"] = {"encode_ms": round(enc, 4), "encode_hash_ms": round(eh, 4), "shardsum_only_ms": round(hs, 4),
                                  "encode_frac_of_8TBps": round(algo / enc / 1e6 / 8000, 4), "encode_hash_frac_of_8TBps": round(algo / eh / 1e6 / 8000, 4),
                                  "shardsum_GBps": round(algo / hs / 1e6, 1), "encode_hash_over_encode": round(eh / enc, 3)}
        res[f"rs{k}_{m}_x{nb}"] = row
    print(json.dumps(res))


if __name__ == "__main__":
    main()
