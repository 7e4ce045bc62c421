"""Where a bulk put / get through the BlockManager mirror spends HOST core time on the HIP codec: per-thread user + system
time (/proc/self/task/*/stat) over `reps` calls of `nb` 1 MiB blocks, grouped by thread name.  tools/host_path_bench.py
host_cpu gives the totals beside the CPU codec's; this says whose they are."""
import collections
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import garage_amd as g  # noqa: E402
from garage_amd import block_native as bn  # noqa: E402

TICK = os.sysconf("SC_CLK_TCK")


def per_thread():
    out = {}
    for tid in os.listdir("/proc/self/task"):
        try:
            st = open(f"/proc/self/task/{tid}/stat").read()
        except OSError:
            continue
        name = st[st.index("(") + 1:st.rindex(")")]
        f = st[st.rindex(")") + 2:].split()
        out[int(tid)] = (name, (int(f[11]), int(f[12])))   # utime, stime
    return out


def delta(a, b):
    by = collections.defaultdict(lambda: [0.0, 0.0, 0])
    for tid, (name, (u1, s1)) in b.items():
        u0, s0 = a.get(tid, (name, (0, 0)))[1]
        if u1 + s1 - u0 - s0 > 0:
            by[name][0] += (u1 - u0) / TICK
            by[name][1] += (s1 - s0) / TICK
            by[name][2] += 1
    return {k: {"user_s": round(v[0], 3), "sys_s": round(v[1], 3), "threads": v[2]} for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1][:2]))}


def main():
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    backend = sys.argv[3] if len(sys.argv) > 3 else "hip"
    L = 1 << 20
    codec = g.ReedSolomon(10, 4, backend=backend)
    mgr = bn.NativeBlockManager(codec, 16)
    rng = np.random.default_rng(11)
    blocks = [rng.integers(0, 256, L, dtype=np.uint8).tobytes() for _ in range(nb)]
    hashes = [bn.blake2sum(b) for b in blocks]
    items = list(zip(hashes, blocks))
    outs = [np.empty(L, dtype=np.uint8) for _ in range(nb)]
    for _ in range(3):
        mgr.rpc_put_blocks(items)
    mgr.rpc_get_blocks(hashes, L, out=outs)
    gib = nb * L / 2**30 * reps
    res = {"nblocks": nb, "reps": reps, "backend": backend, "GiB_per_phase": gib}
    for what, fn in (("put", lambda: mgr.rpc_put_blocks(items)), ("get", lambda: mgr.rpc_get_blocks(hashes, L, out=outs))):
        a, t0 = per_thread(), time.perf_counter()
        for _ in range(reps):
            fn()
        dt = time.perf_counter() - t0
        by = delta(a, per_thread())
        tot = sum(v["user_s"] + v["sys_s"] for v in by.values())
        res[what] = {"GiBps": round(gib / dt, 2), "core_s_per_GiB": round(tot / gib, 4), "by_thread_name": by}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
