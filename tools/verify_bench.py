#!/usr/bin/env python3
"""Device-resident rate of gec_verify_batch_dev (ReedSolomon::verify; the scrub path): a pure read stream of
(k+m)*S bytes per block, compared with the encode of the same stripes."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import garage_amd as g  # noqa: E402


def timed(fn, reps=300):
    for _ in range(50):
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
    for k, m, L, nb in ((10, 4, 1 << 20, 1024), (20, 8, 4 << 20, 256), (3, 1, 65536, 16384)):
        rs = g.ReedSolomon(k, m)
        S = g.shard_len(k, L)
        st = torch.randint(0, 256, (nb, k + m, S), dtype=torch.uint8, device="cuda:0")
        rs.encode_dev(st)
        assert bool(rs.verify_dev(st).all())
        st[5, k, 77] ^= 1
        assert (~rs.verify_dev(st)).nonzero().flatten().tolist() == [5]
        st[5, k, 77] ^= 1
        bad = torch.empty((nb,), dtype=torch.int32, device="cuda:0")
        lib, h = g._lib.lib, rs._h
        n = k + m
        ms_v = timed(lambda: lib.gec_verify_batch_dev(h, nb, st.data_ptr(), n * S, S, bad.data_ptr(), None))
        ms_e = timed(lambda: lib.gec_encode_batch_dev(h, nb, st.data_ptr(), n * S, S, st.data_ptr() + k * S, n * S, None))
        by = n * S * nb
        res[f"RS({k},{m}) {L >> 10} KiB x{nb}"] = {"verify_us": round(ms_v * 1e3, 1), "verify_read_TBps": round(by / ms_v / 1e9, 3),
                                                  "verify_frac_of_8TBps": round(by / ms_v / 1e9 / 8, 4),
                                                  "encode_us": round(ms_e * 1e3, 1), "encode_frac": round(by / ms_e / 1e9 / 8, 4)}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
