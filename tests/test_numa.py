"""NUMA placement of a device lane's host side (garage_amd/csrc/numa.hpp, include/garage_ec.h
gec_codec_numa_node).  No reference counterpart: Garage has no device.  CPU tests: the helpers under sanitizers, the C ABI on a CPU
codec (nothing is placed, nothing breaks).  GPU tests: the codec's node is what sysfs says of its PCI address, the threads a lane
starts run on that node's CPUs, its pinned memory -- staging slots, gec_host_alloc_near, the manager's shard buffers -- sits on
that node (move_pages in query mode), GEC_NUMA=0 switches all of it off, and results are the same bytes either way."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

import garage_amd as g
from garage_amd import _lib
from garage_amd.codec import host_free

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CDIR = os.path.join(ROOT, "tests", "c")


def test_numa_helpers_under_asan_ubsan():
    r = subprocess.run(["make", "-C", CDIR, "numa_san_test"], capture_output=True, text=True)
    if r.returncode != 0 and "cannot find" in (r.stdout + r.stderr):
        pytest.skip("sanitizer runtime not installed")
    assert r.returncode == 0, r.stdout + r.stderr
    r = subprocess.run([os.path.join(CDIR, "numa_san_test")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "numa_san_test ok" in r.stdout, r.stdout + r.stderr


def test_a_cpu_codec_places_nothing_and_the_calls_still_answer():
    rs = g.ReedSolomon(10, 4, backend="cpu")
    assert rs.numa_node == -1 and rs.numa_cpus == []
    before = os.sched_getaffinity(0)
    assert _lib.lib.gec_numa_bind_thread(rs._h) == 0 and os.sched_getaffinity(0) == before
    a = rs.host_alloc(1 << 20)
    a[:] = 7
    assert _lib.lib.gec_numa_node_of(a.ctypes.data) >= -1
    host_free(a)
    cnt = ctypes.c_size_t(99)
    assert _lib.lib.gec_codec_numa_cpus(None, 0, None, ctypes.byref(cnt)) == _lib.GEC_E_INVALID_ARG
    assert _lib.lib.gec_codec_numa_node(None) == -1 and _lib.lib.gec_numa_node_of(None) == -1
    assert "GEC_NUMA" in _lib.lib.gec_env_table().decode()


# ------------------------------------------------------------------------------------------------ on the GPU box
def _sysfs_node_of_device0():
    import torch

    bdf = torch.cuda.get_device_properties(0).pci_bus_id if hasattr(torch.cuda.get_device_properties(0), "pci_bus_id") else None
    hip = ctypes.CDLL("libamdhip64.so")
    buf = ctypes.create_string_buffer(64)
    assert hip.hipDeviceGetPCIBusId(buf, 64, 0) == 0
    bdf = buf.value.decode().lower()
    try:
        return int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read())
    except OSError:
        return -1


def _nodes():
    base = "/sys/devices/system/node"
    return sorted(int(d[4:]) for d in os.listdir(base) if d.startswith("node") and d[4:].isdigit()) if os.path.isdir(base) else []


def _cpulist(node):
    out = []
    for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
        a, _, b = part.partition("-")
        out += list(range(int(a), int(b or a) + 1))
    return out


_CHILD = r'''
import ctypes, json, os, sys, threading
sys.path.insert(0, %(root)r)
import numpy as np
import garage_amd as g
from garage_amd import _lib, block_native as bn
from garage_amd.codec import host_free
from oracle import rs_oracle as O

def threads_by_name():
    out = {}
    for tid in os.listdir("/proc/self/task"):
        try:
            name = open(f"/proc/self/task/{tid}/comm").read().strip()
            out.setdefault(name, []).append(sorted(os.sched_getaffinity(int(tid))))
        except OSError:
            pass
    return out

rs = g.ReedSolomon(10, 4, device=0)
res = {"node": rs.numa_node, "cpus": rs.numa_cpus}
a = rs.host_alloc(8 << 20); a[:] = 1
res["near_alloc_nodes"] = sorted({_lib.lib.gec_numa_node_of(a.ctypes.data + off) for off in range(0, 8 << 20, 1 << 20)})
# a lane: manager + batcher over the codec; a pageable put (staging slots + copy threads), a get, a put through the queue
mgr = bn.NativeBlockManager(rs, 16)
blocks = [bytes(O.splitmix64_bytes(70 + i, 1 << 20)) for i in range(48)]
hashes = [bn.blake2sum(b) for b in blocks]
mgr.rpc_put_blocks(list(zip(hashes, blocks)))
res["get_ok"] = mgr.rpc_get_blocks(hashes, 1 << 20) == blocks
bt = bn.Batcher(mgr, max_blocks=16, max_wait_us=200)
bt.put_block(hashes[0], blocks[0])
# the host-pointer encode on pageable memory runs the codec's copy threads
par = rs.encode_blocks(blocks[:8])
co = O.COracle()
S = g.shard_len(10, 1 << 20)
data = np.zeros((8, 10, S), dtype=np.uint8)
for b in range(8):
    data[b].reshape(-1)[: 1 << 20] = np.frombuffer(blocks[b], dtype=np.uint8)
res["parity_ok"] = bool(np.array_equal(np.stack(par), co.encode_batch(10, 4, data, co.SCALAR)))
res["threads"] = threads_by_name()
# where the manager's shard buffers are: the pointers of a stored shard are not exposed, so sample what the lane's pool hands out
# through the same call the BufPool makes
b2 = rs.host_alloc(4 << 20); b2[:] = 2
res["pool_alloc_nodes"] = sorted({_lib.lib.gec_numa_node_of(b2.ctypes.data + off) for off in range(0, 4 << 20, 1 << 20)})
res["main_thread_affinity"] = len(os.sched_getaffinity(0))
bt.close(); mgr.close()
print(json.dumps(res))
'''


def _run_child(env_extra):
    import json

    r = subprocess.run([sys.executable, "-c", _CHILD % {"root": ROOT}], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, **env_extra))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])


LANE_THREADS = ("gec-pool", "gbm-pool", "gbm-batch-put", "gbm-batch-get")


@pytest.mark.gpu
def test_a_lane_runs_and_allocates_on_its_devices_node():
    want = _sysfs_node_of_device0()
    nodes = _nodes()
    res = _run_child({})
    assert res["get_ok"] and res["parity_ok"]
    if want < 0 or len(nodes) < 2:
        assert res["node"] == -1 and res["cpus"] == []          # one node (or a platform that does not say): nothing is placed
        return
    cpus = _cpulist(want)
    assert res["node"] == want and res["cpus"] == cpus
    assert res["near_alloc_nodes"] == [want] and res["pool_alloc_nodes"] == [want]
    seen = 0
    for name in LANE_THREADS:
        for aff in res["threads"].get(name, []):
            assert set(aff) <= set(cpus), (name, aff[:4], "...")
            seen += 1
    assert seen >= 8, sorted(res["threads"])                    # copy threads, pool, batcher workers all exist and are bound
    assert res["main_thread_affinity"] == len(os.sched_getaffinity(0))   # the caller's own thread is never touched


@pytest.mark.gpu
def test_gec_numa_0_switches_placement_off_and_far_is_the_other_node():
    nodes = _nodes()
    want = _sysfs_node_of_device0()
    off = _run_child({"GEC_NUMA": "0"})
    assert off["node"] == -1 and off["cpus"] == [] and off["get_ok"] and off["parity_ok"]
    allowed = len(os.sched_getaffinity(0))
    for name in LANE_THREADS:
        for aff in off["threads"].get(name, []):
            assert len(aff) == allowed, (name, len(aff))        # unbound: what the process may use
    if want < 0 or len(nodes) < 2:
        return
    far = _run_child({"GEC_NUMA": "far"})                        # the test hook of the A/B (profiles/r06_numa.txt)
    other = (want + 1) % len(nodes)
    assert far["node"] == other and far["near_alloc_nodes"] == [other] and far["get_ok"] and far["parity_ok"]
    for name in LANE_THREADS:
        for aff in far["threads"].get(name, []):
            assert set(aff) <= set(_cpulist(other))
