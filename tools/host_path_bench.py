#!/usr/bin/env python3
"""The path above the kernel, bounded to a few seconds: (a) PCIe-inclusive rate of the
host-pointer C ABI (what the Rust shim calls) and (b) put/get rate of the C++ BlockManager
mirror on in-memory nodes.  bench.py embeds both objects in its JSON line at N=1
(`pcie_inclusive`, `block_manager`) -- next to the device-resident `value`, never as it
(SURVEY.md section 8d).  usage: host_path_bench.py [nblocks]"""
from __future__ import annotations

import ctypes
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

K, M, L = 10, 4, 1 << 20


def _best(fn, reps):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return min(ts), sorted(ts)[len(ts) // 2]


def pcie_inclusive_rates(nb: int = 512, reps: int = 5) -> dict:
    """gec_encode_batch / gec_reconstruct_batch on host buffers: H2D of the data shards + kernel +
    D2H of the parity (rebuilt shards).  Two flavours of caller memory: ordinary pageable buffers
    (the library stages them through its own pinned slots) and buffers from gec_host_alloc (pinned:
    DMA straight from / to the caller's memory, no staging copy)."""
    import garage_amd as g
    from garage_amd._lib import check, lib

    S = g.shard_len(K, L)
    n = K + M
    rs = g.ReedSolomon(K, M)
    rng = np.random.default_rng(1)
    out = {"what": "host-pointer C ABI, RS(10,4), 1 MiB blocks: H2D data + kernel + D2H parity, payload GiB/s",
           "nblocks": nb, "pcie_gen5_x16_spec_GBps": 63}

    def run(kind, alloc, free):
        blocks = [alloc(K * S) for _ in range(nb)]      # padded to k*S so that reconstruct can use them as shards
        for b in blocks:
            b[:L] = rng.integers(0, 256, L, dtype=np.uint8)
            b[L:] = 0
        outs = [alloc(M * S) for _ in range(nb)]
        lens = (ctypes.c_size_t * nb)(*[L] * nb)
        ptrs = (ctypes.c_void_p * nb)(*[b.ctypes.data for b in blocks])
        optrs = (ctypes.c_void_p * nb)(*[o.ctypes.data for o in outs])
        enc = lambda: check(lib.gec_encode_batch(rs._h, nb, ptrs, lens, S, optrs), "gec_encode_batch")  # noqa: E731
        enc()
        best, med = _best(enc, reps)
        out[f"encode_{kind}_GiBps"] = round(nb * L / best / 2**30, 2)
        out[f"encode_{kind}_median_GiBps"] = round(nb * L / med / 2**30, 2)
        # reconstruct_data: data shards 0 and 3 of every block lost
        rec = [alloc(2 * S) for _ in range(nb)]
        sp = (ctypes.c_void_p * (nb * n))()
        op = (ctypes.c_void_p * (nb * n))()
        for b in range(nb):
            for j in range(n):
                if j in (0, 3):
                    sp[b * n + j] = None
                    op[b * n + j] = rec[b].ctypes.data + (0 if j == 0 else S)
                else:
                    sp[b * n + j] = blocks[b].ctypes.data + j * S if j < K else outs[b].ctypes.data + (j - K) * S
        dec = lambda: check(lib.gec_reconstruct_batch(rs._h, nb, sp, op, S, 1), "gec_reconstruct_batch")  # noqa: E731
        dec()
        best, _ = _best(dec, reps)
        assert np.array_equal(rec[0][:S], blocks[0][:S]) and np.array_equal(rec[-1][S:], blocks[-1][3 * S:4 * S])
        out[f"reconstruct_data_2_lost_{kind}_GiBps"] = round(nb * L / best / 2**30, 2)
        # encode + the checksums of all k+m shards in the same trip (what the BlockManager's put calls)
        sums = np.zeros((nb, n, 32), dtype=np.uint8)
        u8 = ctypes.POINTER(ctypes.c_uint8)
        eh = lambda: check(lib.gec_encode_hash_batch(rs._h, nb, ptrs, lens, S, optrs, sums.ctypes.data_as(u8)), "gec_encode_hash_batch")  # noqa: E731
        eh()
        best, _ = _best(eh, reps)
        out[f"encode_hash_{kind}_GiBps"] = round(nb * L / best / 2**30, 2)
        # verify (scrub): all k+m shards cross the link, payload rate
        vp = (ctypes.c_void_p * (nb * n))(*[blocks[b].ctypes.data + j * S if j < K else outs[b].ctypes.data + (j - K) * S
                                            for b in range(nb) for j in range(n)])
        okf = np.zeros(nb, dtype=np.uint8)
        ver = lambda: check(lib.gec_verify_batch(rs._h, nb, vp, S, okf.ctypes.data_as(u8)), "gec_verify_batch")  # noqa: E731
        ver()
        best, _ = _best(ver, reps)
        assert okf.all()
        out[f"verify_{kind}_GiBps"] = round(nb * L / best / 2**30, 2)
        for a in blocks + outs + rec:
            free(a)

    run("pageable", lambda sz: np.empty(sz, dtype=np.uint8), lambda a: None)
    if hasattr(lib, "gec_host_alloc"):
        from garage_amd.codec import host_alloc, host_free

        run("pinned", host_alloc, host_free)
    return out


def block_manager_rates(nb: int = 512, threads: int = 48) -> dict:
    """libgarage_block (C++ BlockManager mirror) on 16 in-memory nodes: coalesced put, get, get with
    4 nodes down (every block needs a decode), and concurrent single-block puts through the batcher (48 callers = 16 PutObject requests x PUT_BLOCKS_MAX_PARALLEL = 3, src/api/s3/put.rs:42)."""
    import garage_amd as g
    from garage_amd import block_native as bn

    codec = g.ReedSolomon(K, M)
    mgr = bn.NativeBlockManager(codec, 16)
    rng = np.random.default_rng(3)
    blocks = [rng.integers(0, 256, L, dtype=np.uint8).tobytes() for _ in range(nb)]
    hashes = codec.blake2sum_batch(blocks)          # Garage's block names, computed on the GPU
    items = list(zip(hashes, blocks))
    for _ in range(3):                              # warm: three generations of shard buffers size the pinned pool
        mgr.rpc_put_blocks(items)                   # (a put that still allocates pinned memory runs at 25 instead of 40 GiB/s)
    mgr.rpc_get_blocks(hashes, L)
    gib = nb * L / 2**30
    t_put, t_put_med = _best(lambda: mgr.rpc_put_blocks(items), 5)
    # the caller's receive buffers exist before the call (as the reference's would: the stream is consumed into
    # the response body); allocating 512 MiB of fresh Python buffers per call is not what is being measured
    outs = [np.empty(L, dtype=np.uint8) for _ in range(nb)]
    check = lambda r: all(x == L for x in r) and all(outs[i].tobytes() == blocks[i] for i in (0, 1, nb // 2, nb - 1))  # noqa: E731
    res_ = []
    # the requester's end-to-end block hash is a mode (gbm_set_verify_block_hash): off = the reference's read path and the
    # default; rebuilt = only blocks that went through a decode; always = round 3's behaviour
    t_mode, t_deg_mode = {}, {}
    default_mode = mgr.verify_block_hash          # what a manager created over this codec runs with (always over checksum v3)
    for mode in ("off", "rebuilt", "always"):
        mgr.set_verify_block_hash(mode)
        t_mode[mode], _ = _best(lambda: res_.__setitem__(slice(None), mgr.rpc_get_blocks(hashes, L, out=outs)), 3)
        assert check(res_)
    for node in range(4):
        mgr.node_set_down(node, True)
    for mode in ("off", "rebuilt", "always"):
        mgr.set_verify_block_hash(mode)
        for o in outs:
            o[:] = 0
        t_deg_mode[mode], _ = _best(lambda: res_.__setitem__(slice(None), mgr.rpc_get_blocks(hashes, L, out=outs)), 3)
        assert check(res_) and all(outs[i].tobytes() == blocks[i] for i in range(0, nb, 37))
    for node in range(4):
        mgr.node_set_down(node, False)
    mgr.set_verify_block_hash("off")
    t_get, t_get_nv, t_deg = t_mode["always"], t_mode["off"], t_deg_mode["off"]
    bt = bn.Batcher(mgr, max_blocks=128, max_wait_us=300)
    per = max(1, nb // threads)

    def worker(t):
        for j in range(per):
            i = t * per + j
            bt.put_block(hashes[i], blocks[i])

    th = [threading.Thread(target=worker, args=(t,)) for t in range(threads)]
    t0 = time.perf_counter()
    [x.start() for x in th]
    [x.join() for x in th]
    t_bat = time.perf_counter() - t0
    bstats = bt.stats()
    bt.close()
    native = native_batcher_rate(threads)
    native96 = native_batcher_rate(2 * threads)
    res = {
        "what": "libgarage_block (C++ BlockManager mirror over the C ABI), RS(10,4), 1 MiB blocks, 16 in-memory nodes, payload GiB/s",
        "nblocks": nb,
        "rpc_put_blocks_GiBps": round(gib / t_put, 2),
        "rpc_put_blocks_median_GiBps": round(gib / t_put_med, 2),
        "rpc_get_blocks_GiBps": round(gib / t_mode[default_mode], 2),   # the DEFAULT mode of this manager (named below)
        "rpc_get_blocks_by_verify_mode_GiBps": {mo: round(gib / t_mode[mo], 2) for mo in t_mode},
        "rpc_get_blocks_4_nodes_down_GiBps": round(gib / t_deg_mode[default_mode], 2),   # (the default mode as well)
        "rpc_get_blocks_4_nodes_down_by_verify_mode_GiBps": {mo: round(gib / t_deg_mode[mo], 2) for mo in t_deg_mode},
        "rpc_get_blocks_with_block_hash_always_GiBps": round(gib / t_get, 2),
        "verify_mode_default": default_mode + " (shard checksums are verified in every mode; over MLH64 shard checksums -- header version 3 -- every Plain block is "
                               "also hashed against its name by default, as the reference's read path does; rebuilt / off are the operator's modes)",
        # the batcher under 48 native callers (tools/batcher_bench, C: no interpreter between the callers and the
        # library) is the figure; the same load from Python threads is kept beside it -- the GIL hand-offs between
        # 48 threads cost it a fifth
        f"batcher_{threads}_threads_put_GiBps": native.get("GiBps", round(threads * per * L / 2**30 / t_bat, 2)),
        f"batcher_{threads}_threads_put_source": "tools/batcher_bench (native callers)" if "GiBps" in native else "python threads (tools/batcher_bench not built)",
        f"batcher_{threads}_python_threads_put_GiBps": round(threads * per * L / 2**30 / t_bat, 2),
        "batcher_native": native,
        f"batcher_{2 * threads}_threads_put_GiBps": native96.get("GiBps"),
        f"batcher_native_{2 * threads}": native96,
        "small_trips": small_trip_rates(),
        "batcher_stats": bstats,
        "ec_reconstructs": mgr.metrics["ec_reconstructs"],
        "messages_hashed_on_gpu": mgr.gpu_hashed(),
    }
    mgr.close()
    return res


def host_cpu_cost(nb: int = 256) -> dict:
    """What the offload is for: the host-core time a put and a get of the same blocks cost through the SAME manager code
    on the HIP codec and on the product's CPU codec (GFNI / AVX-512 on these hosts), process-wide (every pool thread
    counted: getrusage).  On the HIP codec the cores copy payload into pinned shard buffers, fan shards out and verify
    nothing themselves; on the CPU codec they also encode, decode and hash."""
    import resource

    import garage_amd as g
    from garage_amd import block_native as bn

    rng = np.random.default_rng(11)
    blocks = [rng.integers(0, 256, L, dtype=np.uint8).tobytes() for _ in range(nb)]
    gib = nb * L / 2**30
    out = {"what": "host core-seconds per GiB of payload (user + system, all threads) and GiB/s, RS(10,4), 1 MiB blocks, 16 memory nodes",
           "nblocks": nb, "host_cores": os.cpu_count()}

    def cpu_s():
        r = resource.getrusage(resource.RUSAGE_SELF)
        return r.ru_utime + r.ru_stime

    for backend in ("hip", "cpu"):
        try:
            codec = g.ReedSolomon(K, M, backend=backend)
            mgr = bn.NativeBlockManager(codec, 16)
            hashes = [bn.blake2sum(b) for b in blocks]
            items = list(zip(hashes, blocks))
            outs = [np.empty(L, dtype=np.uint8) for _ in range(nb)]
            for _ in range(3):
                mgr.rpc_put_blocks(items)
            mgr.rpc_get_blocks(hashes, L, out=outs)
            reps = 4
            c0, t0 = cpu_s(), time.perf_counter()
            for _ in range(reps):
                mgr.rpc_put_blocks(items)
            c1, t1 = cpu_s(), time.perf_counter()
            for _ in range(reps):
                mgr.rpc_get_blocks(hashes, L, out=outs)
            c2, t2 = cpu_s(), time.perf_counter()
            assert outs[5].tobytes() == blocks[5]
            out[backend] = {"put_core_s_per_GiB": round((c1 - c0) / (reps * gib), 4), "put_GiBps": round(reps * gib / (t1 - t0), 2),
                            "get_core_s_per_GiB": round((c2 - c1) / (reps * gib), 4), "get_GiBps": round(reps * gib / (t2 - t1), 2)}
            mgr.close()
        except Exception as e:  # noqa: BLE001
            out[backend] = {"error": f"{type(e).__name__}: {e}"[:200]}
    return out


def small_trip_rates() -> dict:
    """tools/small_trip_bench (native): one put / a PutObject's three through the batcher, one get per verify mode (healthy and
    degraded), streaming gets (first / last chunk), 48 readers through the batcher per mode."""
    import re
    import subprocess

    exe = os.path.join(ROOT, "tools", "small_trip_bench")
    if not os.path.exists(exe):
        r = subprocess.run(["make", "-C", os.path.join(ROOT, "tools"), "small_trip_bench"], capture_output=True, text=True)
        if r.returncode != 0 or not os.path.exists(exe):
            return {"error": "tools/small_trip_bench is not built"}
    try:
        r = subprocess.run([exe, "48", "20"], capture_output=True, text=True, timeout=300)
    except subprocess.SubprocessError as e:
        return {"error": f"{type(e).__name__}"}
    out = {"source": "tools/small_trip_bench (native callers), RS(10,4), 1 MiB blocks, 16 in-memory nodes, medians"}
    txt = r.stdout
    m = re.findall(r"put, 1 caller through the batcher, pass \d: median ([0-9.]+) ms", txt)
    if m:
        out["put_one_block_ms"] = float(m[-1])
    m = re.search(r"a PutObject's three in flight:\s+median ([0-9.]+) ms", txt)
    if m:
        out["put_three_blocks_ms"] = float(m.group(1))
    for state in ("healthy", "degraded"):
        for mo, key in (("off", "off"), ("rebuilt-only", "rebuilt"), ("always", "always")):
            m = re.search(rf"get, one 1 MiB block, {state}\s+mode {mo}\s*: median ([0-9.]+) ms", txt)
            if m:
                out.setdefault(f"get_one_block_{state}_ms", {})[key] = float(m.group(1))
    for mib in (1, 4):
        for mo, key in (("off", "off"), ("rebuilt-only", "rebuilt"), ("always", "always")):
            m = re.search(rf"streaming get, {mib} MiB block, mode {mo}\s*: first chunk ([0-9.]+) ms, last chunk ([0-9.]+) ms, call returns ([0-9.]+) ms", txt)
            if m:
                out.setdefault(f"streaming_get_{mib}MiB_ms", {})[key] = {"first_chunk": float(m.group(1)), "last_chunk": float(m.group(2)),
                                                                             "call_returns": float(m.group(3))}
    for m in re.finditer(r"ranged get,\s+(\d+) bytes of a 4 MiB block: median ([0-9.]+) ms, (\d+) KiB of shards read", txt):
        out.setdefault("ranged_get_of_4MiB_block", {})[f"{m.group(1)}_bytes"] = {"median_ms": float(m.group(2)), "shard_KiB_read": int(m.group(3))}
    for mo, key in (("off", "off"), ("rebuilt-only", "rebuilt"), ("always", "always")):
        m = re.search(rf"48 readers x 20 gets through the batcher, mode {mo}\s*: ([0-9.]+) GiB/s, median ([0-9.]+) ms, p99 ([0-9.]+) ms", txt)
        if m:
            out.setdefault("batcher_48_readers", {})[key] = {"GiBps": float(m.group(1)), "median_ms": float(m.group(2)), "p99_ms": float(m.group(3))}
    if len(out) == 1:
        out["error"] = (r.stderr or r.stdout)[-200:]
    return out


def native_batcher_rate(threads: int = 48, puts: int = 20) -> dict:
    """tools/batcher_bench: `threads` native callers, each putting `puts` blocks of 1 MiB one after the other through
    gbm_batcher_put_block; best of its three repetitions."""
    import re
    import subprocess

    exe = os.path.join(ROOT, "tools", "batcher_bench")
    if not os.path.exists(exe):
        r = subprocess.run(["make", "-C", os.path.join(ROOT, "tools"), "batcher_bench"], capture_output=True, text=True)
        if r.returncode != 0 or not os.path.exists(exe):
            return {"error": "tools/batcher_bench is not built"}
    try:
        r = subprocess.run([exe, str(threads), str(puts)], capture_output=True, text=True, timeout=120)
    except subprocess.SubprocessError as e:
        return {"error": f"{type(e).__name__}"}
    best = None
    for line in r.stdout.splitlines():
        m = re.search(r"= ([0-9.]+) GiB/s; (\d+) batches, largest (\d+); put latency median ([0-9.]+) ms, p99 ([0-9.]+) ms", line)
        if m and (best is None or float(m.group(1)) > best["GiBps"]):
            best = {"GiBps": float(m.group(1)), "batches": int(m.group(2)), "largest_batch": int(m.group(3)),
                    "put_latency_median_ms": float(m.group(4)), "put_latency_p99_ms": float(m.group(5)), "callers": threads,
                    "puts_per_caller": puts}
    return best or {"error": (r.stderr or r.stdout)[-200:]}


def maintenance_rates(nb: int = 512, root: str = "") -> dict:
    """Rows f2 / f3 of SURVEY.md section 8 as rates: scrub of everything stored (gbm_scrub_all: every stripe's k+m shards
    through gec_verify_batch) and the resync of one node's worth of lost shards (gbm_resync_run: presence scan, gather k,
    one gec_reconstruct_batch per erasure pattern, PutShard).  Memory nodes, and directory nodes under `root` (tmpfs on
    the GPU box: the file format and the directory walk, not a disk)."""
    import shutil
    import tempfile

    import garage_amd as g
    from garage_amd import block_native as bn

    codec = g.ReedSolomon(K, M)
    rng = np.random.default_rng(5)
    blocks = [rng.integers(0, 256, L, dtype=np.uint8).tobytes() for _ in range(nb)]
    hashes = codec.blake2sum_batch(blocks)
    items = list(zip(hashes, blocks))
    gib = nb * L / 2**30
    res = {"what": "libgarage_block maintenance paths, RS(10,4), 1 MiB blocks, 16 nodes, payload GiB/s of the blocks concerned", "nblocks": nb}
    for kind in ("memory", "directories"):
        tmp = None
        if kind == "directories":
            need = 3 * nb * (K + M) * (g.shard_len(K, L) + 64)   # the shard files, with room for the .tmp copies
            shm = "/dev/shm" if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > need else None
            tmp = tempfile.mkdtemp(prefix="gbm_bench_", dir=root or shm)
            mgr = bn.NativeBlockManager(codec, 16, [os.path.join(tmp, f"n{i}") for i in range(16)])
        else:
            mgr = bn.NativeBlockManager(codec, 16)
        try:
            t_put, _ = _best(lambda: mgr.rpc_put_blocks(items), 4)   # (the first two size the pinned buffer pool)
            outs = [np.empty(L, dtype=np.uint8) for _ in range(nb)]
            t_get, _ = _best(lambda: mgr.rpc_get_blocks(hashes, L, out=outs), 3)
            assert outs[7].tobytes() == blocks[7]
            st = {}
            t_scrub, _ = _best(lambda: st.update(mgr.scrub_all(256)), 2)
            assert st["scrubbed"] == nb and st["corruptions"] == 0
            # the same store through the ScrubWorker (directory-by-directory iterator, checkpoints, the worker thread): one pass
            mgr.set_tranquility(scrub=0)
            mgr.scrub_worker_start(None, 256)
            t_worker = None
            for _ in range(2):
                last = mgr.scrub_worker_status()["time_last_complete_scrub_ms"]
                t0 = time.perf_counter()
                mgr.scrub_worker_command(bn.SCRUB_START)
                while True:
                    ws = mgr.scrub_worker_status()
                    if ws["state"] == bn.SCRUB_FINISHED and ws["time_last_complete_scrub_ms"] > last:
                        break
                    if time.perf_counter() - t0 > 60:
                        raise TimeoutError(f"the ScrubWorker's pass did not complete: {ws}")
                    time.sleep(0.0005)
                dt = time.perf_counter() - t0
                t_worker = dt if t_worker is None else min(t_worker, dt)
                time.sleep(0.002)   # (time_last_complete is in milliseconds: the next pass must end in a later one)
            assert ws["blocks_scrubbed"] == 2 * nb and ws["corruptions_detected"] == 0 and ws["errors"] == 0
            mgr.scrub_worker_stop()
            # lose everything node 3 holds, queue every block, one resync pass rebuilds it
            lost = 0
            for h in hashes:
                who = mgr.storage_nodes_of(h)
                if 3 in who:
                    mgr.node_delete_shard(3, h, who.index(3))
                    lost += 1
                mgr.put_to_resync(h, 0)
            t0 = time.perf_counter()
            rs_ = mgr.resync_run(0)
            t_resync = time.perf_counter() - t0
            assert mgr.scrub_all(256)["corruptions"] == 0 and all(mgr.node_has_shard(3, h, mgr.storage_nodes_of(h).index(3))
                                                                  for h in hashes[:50] if 3 in mgr.storage_nodes_of(h))
            res[kind] = {
                **({"rpc_put_blocks_GiBps": round(gib / t_put, 2), "rpc_get_blocks_GiBps": round(gib / t_get, 2)}
                   if kind == "directories" else {}),   # memory nodes: block_manager_rates' numbers
                "scrub_all_GiBps": round(gib / t_scrub, 2),
                "scrub_worker_pass_GiBps": round(gib / t_worker, 2),
                "resync_one_lost_node": {"blocks_queued": nb, "shards_rebuilt": lost, "seconds": round(t_resync, 4),
                                         "GiBps_of_blocks_repaired": round(lost * L / 2**30 / t_resync, 2),
                                         "device_calls": rs_.get("device_calls", rs_.get("batches"))},
            }
        except Exception as e:  # noqa: BLE001 -- one kind of node failing (a full tmpfs) must not cost the other's numbers
            res[kind] = {"error": f"{type(e).__name__}: {e}"[:300]}
        finally:
            mgr.close()
            if tmp:
                shutil.rmtree(tmp, ignore_errors=True)
    return res


if __name__ == "__main__":
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    if len(sys.argv) > 2 and sys.argv[2] == "maintenance":
        print(json.dumps({"maintenance": maintenance_rates(nb)}))
    elif len(sys.argv) > 2 and sys.argv[2] == "host_cpu":
        print(json.dumps({"host_cpu": host_cpu_cost(nb)}))
    else:
        print(json.dumps({"pcie_inclusive": pcie_inclusive_rates(nb), "block_manager": block_manager_rates(nb)}))
