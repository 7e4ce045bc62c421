#!/usr/bin/env python3
"""What a device lane's host side can know and do about NUMA on this box: the GPU's node, whether the
memory-policy system calls are allowed in this container, where hipHostMalloc puts pages from a thread on either socket, and what
near / far placement is worth on the link and in the staging copy.  Prints one JSON object."""
import ctypes
import json
import os
import re
import sys
import threading
import time

libc = ctypes.CDLL(None, use_errno=True)
hip = ctypes.CDLL("libamdhip64.so")
SYS = {"mbind": 237, "set_mempolicy": 238, "get_mempolicy": 239, "move_pages": 279}
out = {}


def cpulist(s):
    cpus = []
    for part in s.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        cpus += list(range(int(a), int(b or a) + 1))
    return cpus


nodes = {}
for d in sorted(os.listdir("/sys/devices/system/node")):
    if re.fullmatch(r"node\d+", d):
        nodes[int(d[4:])] = cpulist(open(f"/sys/devices/system/node/{d}/cpulist").read())
out["nodes"] = {k: f"{len(v)} cpus {v[0]}..{v[-1]}" for k, v in nodes.items()}
buf = ctypes.create_string_buffer(64)
assert hip.hipSetDevice(0) == 0
assert hip.hipDeviceGetPCIBusId(buf, 64, 0) == 0
bdf = buf.value.decode().lower()
out["gpu0_bdf"] = bdf
try:
    out["gpu0_numa_node"] = int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read())
except OSError as e:
    out["gpu0_numa_node"] = str(e)
out["affinity_at_start"] = len(os.sched_getaffinity(0))

# ---- are the memory-policy calls allowed?
mode, mask = ctypes.c_int(), (ctypes.c_ulong * 16)()
libc.syscall.restype = ctypes.c_long
r = libc.syscall(SYS["get_mempolicy"], ctypes.byref(mode), mask, 1024, None, 0)
out["get_mempolicy"] = "ok" if r == 0 else os.strerror(ctypes.get_errno())
m1 = (ctypes.c_ulong * 16)()
m1[0] = 1
r = libc.syscall(SYS["set_mempolicy"], 1, m1, 1024)   # MPOL_PREFERRED node 0
out["set_mempolicy"] = "ok" if r == 0 else os.strerror(ctypes.get_errno())
libc.syscall(SYS["set_mempolicy"], 0, None, 0)


def node_of(addr, nbytes, samples=64):
    """move_pages in query mode over `samples` pages of the range -> {node: count}"""
    pages = (ctypes.c_void_p * samples)(*[addr + (i * (nbytes // samples)) // 4096 * 4096 for i in range(samples)])
    status = (ctypes.c_int * samples)()
    r = libc.syscall(SYS["move_pages"], 0, samples, pages, None, status, 0)
    if r != 0:
        return {"error": os.strerror(ctypes.get_errno())}
    hist = {}
    for s in status:
        hist[int(s)] = hist.get(int(s), 0) + 1
    return hist


def numa_maps_of(addr):
    for ln in open("/proc/self/numa_maps"):
        if ln.startswith("%x " % addr) or ln.startswith("%012x " % addr):
            return ln.strip()[:200]
    return None


hip.hipHostMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]
hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
hip.hipHostFree.argtypes = [ctypes.c_void_p]
N = 512 << 20
dev = ctypes.c_void_p()
assert hip.hipMalloc(ctypes.byref(dev), N) == 0


def on_node(node, fn):
    res = {}

    def run():
        os.sched_setaffinity(0, nodes[node])
        assert hip.hipSetDevice(0) == 0
        res["v"] = fn()

    t = threading.Thread(target=run)
    t.start()
    t.join()
    return res["v"]


def bw(host_ptr, kind):
    best = 1e9
    for _ in range(4):
        t0 = time.perf_counter()
        if kind == "h2d":
            assert hip.hipMemcpy(dev, host_ptr, N, 1) == 0
        else:
            assert hip.hipMemcpy(host_ptr, dev, N, 2) == 0
        best = min(best, time.perf_counter() - t0)
    return round(N / best / 2**30, 2)


place = {}
for flags, fname in ((0, "default"), (0x20000000, "hipHostMallocNumaUser")):
    for node in nodes:
        def alloc():
            p = ctypes.c_void_p()
            rc = hip.hipHostMalloc(ctypes.byref(p), N, flags)
            if rc:
                return {"error": rc}
            ctypes.memset(p.value, 1, N)
            r = {"pages_on": node_of(p.value, N), "numa_maps": numa_maps_of(p.value)}
            r["h2d_GiBps_from_this_thread"] = bw(p.value, "h2d")
            r["d2h_GiBps_from_this_thread"] = bw(p.value, "d2h")
            # the staging copy: pageable memory first-touched on this node -> this pinned buffer, by one thread on this node
            src = ctypes.create_string_buffer(N)
            ctypes.memset(src, 2, N)
            t0 = time.perf_counter()
            ctypes.memmove(p.value, src, N)
            r["memcpy_1thread_GiBps"] = round(N / (time.perf_counter() - t0) / 2**30, 2)
            r["_ptr"] = p.value
            return r
        info = on_node(node, alloc)
        ptr = info.pop("_ptr", None)
        if ptr:
            other = [n for n in nodes if n != node]
            if other:
                info["h2d_GiBps_from_other_node_thread"] = on_node(other[0], lambda: bw(ptr, "h2d"))
            hip.hipHostFree(ptr)
        place[f"{fname}, allocating thread on node {node}"] = info
out["hipHostMalloc_placement"] = place
print(json.dumps(out, indent=1))
