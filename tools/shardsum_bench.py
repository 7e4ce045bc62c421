#!/usr/bin/env python3
"""Device-resident rate of the shard checksum (BLAKE2b tree mode) and of encode + checksums, with the shader
clock sampled from rocm-smi while the kernels run (VALU-bound kernels clock down under the power cap)."""
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import garage_amd as g  # noqa: E402


def sample_clock(stop, out):
    while not stop.is_set():
        try:
            r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=5)
            out.append(r.stdout.strip())
        except Exception as e:  # noqa: BLE001
            out.append(repr(e))
        time.sleep(0.05)


def main():
    rs = g.ReedSolomon(10, 4)
    S = 104896
    res = {}
    for nblocks in (64, 256, 1024, 4096):
        n = nblocks * 14
        t = torch.randint(0, 256, (n, S), dtype=torch.uint8, device="cuda:0")
        for _ in range(3):
            rs.shardsum_dev(t)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            rs.shardsum_dev(t)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        res[f"shardsum_{nblocks}_stripes_{n}_shards"] = {"ms": round(ms, 3), "GBps": round(n * S / ms / 1e6, 1)}
        del t
    st = torch.randint(0, 256, (1024, 14, S), dtype=torch.uint8, device="cuda:0")
    flat = st.view(1024 * 14, S)
    stop, samples = threading.Event(), []
    th = threading.Thread(target=sample_clock, args=(stop, samples))
    th.start()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 1.5:      # sustained hashing while rocm-smi samples
        for _ in range(50):
            rs.shardsum_dev(flat)
        torch.cuda.synchronize()
    stop.set()
    th.join()
    for _ in range(3):
        rs.encode_hash_dev(st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        rs.encode_hash_dev(st)
    e1.record()
    torch.cuda.synchronize()
    res["encode_hash_dev_1024_stripes_ms"] = round(e0.elapsed_time(e1) / 20, 3)
    res["rocm_smi_during_hash"] = samples[len(samples) // 2] if samples else None
    print(json.dumps({"what": "shard checksum (BLAKE2b tree mode) of 104896-byte shards, device-resident; GEC_B2_ADD=" + os.environ.get("GEC_B2_ADD", "default"), "results": res}))


if __name__ == "__main__":
    main()
