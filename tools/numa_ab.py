#!/usr/bin/env python3
"""Same-box A/B of the NUMA placement of a device lane's host side -> profiles/r06_numa.txt.

Legs: GEC_NUMA=1 (near: the default), GEC_NUMA=0 (off: round 5's behaviour -- threads float, the runtime places pinned memory),
GEC_NUMA=far (test hook: the lane's threads and pinned memory on the node the device is NOT on: what `numactl --membind` of the far
node would force; the image has no numactl, so the library does it to itself), each with the CALLER (the process's main thread and
the buffers it first-touches: the blocks it puts, the buffers it gets into) on the device's node, on the other node, or unbound.
Every leg is a fresh process; 7 repetitions per figure, best / median / worst, so that a spread is visible as a spread.

usage: python tools/numa_ab.py [--nb 512] [--out gpurun_out/r06_numa.txt]"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import ctypes, json, os, sys, time
sys.path.insert(0, %(root)r)
caller = %(caller)r
def cpulist(node):
    out = []
    for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
        a, _, b = part.partition("-")
        out += list(range(int(a), int(b or a) + 1))
    return out
import numpy as np
import garage_amd as g
from garage_amd import _lib, block_native as bn
from garage_amd._lib import check, lib
hipl = ctypes.CDLL("libamdhip64.so")
buf = ctypes.create_string_buffer(64); hipl.hipDeviceGetPCIBusId(buf, 64, 0)
dev_node = int(open(f"/sys/bus/pci/devices/{buf.value.decode().lower()}/numa_node").read())
nnodes = len([d for d in os.listdir("/sys/devices/system/node") if d.startswith("node") and d[4:].isdigit()])
if caller != "unbound" and nnodes > 1 and dev_node >= 0:
    os.sched_setaffinity(0, cpulist(dev_node if caller == "near" else (dev_node + 1) %% nnodes))   # before any buffer is touched
K, M, L, nb = 10, 4, 1 << 20, %(nb)d
S = g.shard_len(K, L)
rs = g.ReedSolomon(K, M)
res = {"codec_node": rs.numa_node, "device_node": dev_node, "caller": caller, "caller_cpus": len(os.sched_getaffinity(0))}
rng = np.random.default_rng(1)
def stats(fn, reps=7):
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    gib = nb * L / 2**30
    ts.sort()
    return {"best": round(gib / ts[0], 1), "median": round(gib / ts[len(ts) // 2], 1), "worst": round(gib / ts[-1], 1)}
# -- the boundary: gec_encode_batch on pageable caller memory (staging slots + copy threads) and on pinned memory (link kernel)
for kind in ("pageable", "pinned"):
    alloc = (lambda sz: np.empty(sz, dtype=np.uint8)) if kind == "pageable" else rs.host_alloc
    blocks = [alloc(K * S) for _ in range(nb)]
    for b in blocks:
        b[:L] = rng.integers(0, 256, L, dtype=np.uint8); b[L:] = 0
    outs = [alloc(M * S) for _ in range(nb)]
    for o in outs:
        o[:] = 0
    lens = (ctypes.c_size_t * nb)(*[L] * nb)
    ptrs = (ctypes.c_void_p * nb)(*[b.ctypes.data for b in blocks])
    optrs = (ctypes.c_void_p * nb)(*[o.ctypes.data for o in outs])
    res[f"encode_{kind}_GiBps"] = stats(lambda: check(lib.gec_encode_batch(rs._h, nb, ptrs, lens, S, optrs), "enc"))
    res[f"{kind}_block0_on_node"] = lib.gec_numa_node_of(blocks[0].ctypes.data)
    if kind == "pinned":
        from garage_amd.codec import host_free
        for a in blocks + outs:
            host_free(a)
    del blocks, outs
# -- the manager: bulk put, bulk get in the modes that stay off the link (off / rebuilt) and the default
mgr = bn.NativeBlockManager(rs, 16)
blocks = [rng.integers(0, 256, L, dtype=np.uint8).tobytes() for _ in range(nb)]
hashes = rs.blake2sum_batch(blocks)
items = list(zip(hashes, blocks))
for _ in range(3):
    mgr.rpc_put_blocks(items)
res["rpc_put_blocks_GiBps"] = stats(lambda: mgr.rpc_put_blocks(items), 5)
outs = [np.empty(L, dtype=np.uint8) for _ in range(nb)]
for o in outs:
    o[:] = 0
res["verify_mode_default"] = mgr.verify_block_hash
for mode in ("rebuilt", "always"):
    mgr.set_verify_block_hash(mode)
    res[f"rpc_get_blocks_{mode}_GiBps"] = stats(lambda: mgr.rpc_get_blocks(hashes, L, out=outs))
    assert outs[5].tobytes() == blocks[5] and outs[-1].tobytes() == blocks[-1]
for node in range(4):
    mgr.node_set_down(node, True)
mgr.set_verify_block_hash("rebuilt")
res["rpc_get_blocks_4_nodes_down_rebuilt_GiBps"] = stats(lambda: mgr.rpc_get_blocks(hashes, L, out=outs), 5)
assert outs[5].tobytes() == blocks[5]
mgr.close()
print(json.dumps(res))
'''


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nb", type=int, default=512)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r06_numa.txt"))
    ap.add_argument("--repeat", type=int, default=3, help="fresh processes per leg, interleaved (a process keeps the cores and pages it happened to get)")
    a = ap.parse_args()
    rows = []
    for rep in range(a.repeat):
      for numa in ("1", "0", "far"):
        for caller in ("near", "far", "unbound"):
            t0 = time.time()
            r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "caller": caller, "nb": a.nb}], capture_output=True, text=True,
                               timeout=900, env=dict(os.environ, GEC_NUMA=numa))
            lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            row = {"GEC_NUMA": numa, "caller": caller, "process": rep, "wall_s": round(time.time() - t0, 1)}
            if r.returncode == 0 and lines:
                row.update(json.loads(lines[-1]))
            else:
                row["error"] = (r.stderr or r.stdout)[-500:]
            rows.append(row)
            print(json.dumps(row), flush=True)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    keys = ["encode_pageable_GiBps", "encode_pinned_GiBps", "rpc_put_blocks_GiBps", "rpc_get_blocks_rebuilt_GiBps", "rpc_get_blocks_always_GiBps",
            "rpc_get_blocks_4_nodes_down_rebuilt_GiBps"]
    with open(a.out, "w") as f:
        f.write("# tools/numa_ab.py on one MI355X box: RS(10,4), %d x 1 MiB blocks, payload GiB/s as best / median / worst of 7 (5) repetitions.\n"
                "# GEC_NUMA: 1 = the lane's threads and pinned memory on the device's node (default), 0 = off (round 5), far = forced onto the\n"
                "# other node (test hook).  caller = where the process's main thread runs and first-touches its own buffers.\n" % a.nb)
        f.write("%-9s %-8s %-5s " % ("GEC_NUMA", "caller", "node") + " ".join("%-26s" % k.replace("_GiBps", "").replace("rpc_", "")[:26] for k in keys) + "\n")
        for r in rows:
            if "error" in r:
                f.write("%-9s %-8s ERROR %s\n" % (r["GEC_NUMA"], r["caller"], r["error"][-200:].replace("\n", " ")))
                continue
            f.write("%-9s %-8s %-5s " % (r["GEC_NUMA"], r["caller"], r["codec_node"]) +
                    " ".join("%-26s" % ("%.1f / %.1f / %.1f" % (r[k]["best"], r[k]["median"], r[k]["worst"])) for k in keys) + "\n")
        # per leg over its processes: the median process's median, and the spread of the processes' medians
        f.write("# per leg over %d processes: median of the processes' medians [lowest .. highest]\n" % a.repeat)
        for numa in ("1", "0", "far"):
            for caller in ("near", "far", "unbound"):
                mine = [r for r in rows if r["GEC_NUMA"] == numa and r["caller"] == caller and "error" not in r]
                if not mine:
                    continue
                cells = []
                for k in keys:
                    v = sorted(r[k]["median"] for r in mine)
                    cells.append("%-26s" % ("%.1f [%.1f .. %.1f]" % (v[len(v) // 2], v[0], v[-1])))
                f.write("%-9s %-8s %-5s " % (numa, caller, mine[0]["codec_node"]) + " ".join(cells) + "\n")
        f.write("# raw rows\n")
        for r in rows:
            f.write(json.dumps(r) + "\n")
    return 0


if __name__ == "__main__":
    sys.exit(main())
