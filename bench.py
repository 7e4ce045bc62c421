#!/usr/bin/env python3
"""bench.py -- RS(10,4) encode of 1 MiB Garage blocks on MI355X (BASELINE.json).

A "step" is one pass of the hot path over one batch: RS(10,4) encode of the
rank's batch of 1 MiB blocks already resident in HBM (one kernel launch through
the C ABI).  At N=1 the workload is BASELINE config 2 (batch 1024).  At N>1 it
is config 4: a stream of N*1024 blocks hash-partitioned across the GPUs with
`hash[4] % N` on Garage-style 32-byte block hashes, no data-path collective
(weak scaling).

Launching (any of these gives the same JSON line):
  * `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`
    (what the driver does): one process per GPU, RCCL for barrier / reductions;
  * `python bench.py --gpus N` with no WORLD_SIZE in the environment: bench.py
    re-launches itself under torch.distributed.run on a free local port;
  * `python bench.py --gpus N --mode threads`: ONE process, N codecs and N host
    threads through the C ABI -- how a single Garage daemon per node would drive
    the GPUs.  No process group, no RCCL.

Prints ONE JSON line on rank 0 (contract in the task statement), including
`roofline` (HIP-event kernel time vs the 8 TB/s HBM peak, with the secondary
LDS / VALU bounds beside it), `cpu_baseline` (the oracle's C restatement timed
on this host's cores), a `decode` object for BASELINE config 3 (4 erasures),
`pcie_inclusive` and `block_manager` (host-pointer API and BlockManager mirror;
never the `value`) at N=1, and at N>1 a `striped_decode` object for BASELINE
config 5 (RS(20,8), RCCL all-gather decode) with its own bit-exact check.

Defaults (--steps 1000 --warmup 100, ~0.3 s of GPU time) measure the steady
state: MI355X's power management slows the first few milliseconds of a burst of
this kernel by up to 1.5x before settling (profiles/r01_early_20step_burst_dvfs_transient.txt);
the cold burst is reported next to it as `cold_burst_frac`.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# The checks after the timed regions bring stripes back with torch's plain `.cpu()` (synchronous copies into pageable memory).  The
# HIP runtime's default for those is to lock the caller's pages for the DMA engine, and that path faulted about once in eight GPU
# suite runs on these boxes (profiles/r06_gpu_suite_abort.txt: the runtime's own log, nothing of this repository between the lock
# and the fault); with this floor (KiB; read once, when the runtime initialises; inherited by the child processes) the bytes go
# through the runtime's pinned staging buffer instead.  No timed region contains such a copy.
os.environ.setdefault("GPU_PINNED_MIN_XFER_SIZE", "1048576")

K, M = 10, 4
BLOCK_LEN = 1 << 20
BATCH = 1024
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
ORACLE_SAMPLE = 0      # blocks of every rank's timed batch compared with the CPU oracle after the timed region (0 = every block)


def synthetic_hashes(n_total: int):
    """Block hashes of the synthetic stream: Garage's blake2sum (blake2b-512
    truncated to 32 bytes, src/util/data.rs:130-138) of (seed, block index).
    Hashing the 1 MiB payloads themselves would only add ~1 s/GiB of host time
    outside the timed region; the partition statistics are the same."""
    import struct

    import numpy as np

    from garage_amd.partition import block_hash

    raw = b"".join(block_hash(struct.pack("<QQ", 0x6761726167650004, i)) for i in range(n_total))
    return np.frombuffer(raw, dtype=np.uint8).reshape(n_total, 32)


def measured_traffic(nblocks: int):
    """HBM bytes per launch from the committed rocprofv3 PMC passes of exactly this
    workload (profiles/pmc_traffic.json); None for any other batch size.  A STATIC
    figure (PMC counters cannot be collected inside this process); `traffic_source`
    in the JSON line says so."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            rec = json.load(f)["rs10_4_encode_1MiB_x1024"]
    except (OSError, KeyError, ValueError):
        return None, None
    if nblocks != BATCH:
        return None, None
    return rec["traffic_bytes"], f"static: profiles/pmc_traffic.json ({rec.get('source', 'rocprofv3 PMC passes')}); 2*FETCH_SIZE + WRITE_SIZE per launch"


def secondary_bounds():
    """LDS / VALU cycles per tile next to the HBM cycles (hard part 1):
    static, from the committed SQ counter pass."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            return json.load(f).get("rs10_4_secondary_bounds")
    except (OSError, ValueError):
        return None


def host_description() -> dict:
    """What the CPU figures were measured on: lscpu model / sockets / threads, nproc, the OpenMP environment."""
    info = {"nproc": os.cpu_count(), "omp_env": {k: v for k, v in os.environ.items() if k.startswith(("OMP_", "GOMP_", "KMP_"))}}
    try:
        out = subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout
        for line in out.splitlines():
            key, _, val = line.partition(":")
            key, val = key.strip(), val.strip()
            if key in ("Model name", "Socket(s)", "Core(s) per socket", "Thread(s) per core", "NUMA node(s)", "CPU(s)"):
                info["lscpu_" + key.lower().replace("(s)", "s").replace(" ", "_")] = val
    except (OSError, subprocess.SubprocessError):
        pass
    return info


def physical_cores() -> int:
    try:
        out = subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout
        vals = {}
        for line in out.splitlines():
            key, _, val = line.partition(":")
            vals[key.strip()] = val.strip()
        return int(vals["Socket(s)"]) * int(vals["Core(s) per socket"])
    except (OSError, KeyError, ValueError, subprocess.SubprocessError):
        return 0


def cpu_threads_probe(nthreads: int) -> None:
    """`bench.py --cpu-threads-probe N`: the oracle's encode rate at exactly N threads under this process's OpenMP
    environment (cpu_baseline_only runs it with different wait policies to name what starves the all-CPUs point)."""
    from oracle import rs_oracle as O

    co = O.COracle()
    S = ((BLOCK_LEN + K - 1) // K + 63) // 64 * 64
    sec = co.bench_encode(K, M, S, 2048, 3, co.AVX2 if co.has_avx2() else co.SCALAR, nthreads)
    print(json.dumps({"threads": nthreads, "GiBps": round(2048 * BLOCK_LEN / sec / 2**30, 2)}), flush=True)


def cpu_baseline(S: int, quick: bool = False):
    """Oracle C restatement (split-nibble AVX2 when available, OpenMP over blocks,
    buffers first-touched by the threads that encode them) on a bounded sample of
    the same workload; thread count swept and the best reported with its count.
    quick: the two largest thread counts only, 3 repetitions (the in-process cross-check)."""
    from oracle import rs_oracle as O

    co = O.COracle()
    maxthr = co.max_threads()
    variant = co.AVX2 if co.has_avx2() else co.SCALAR
    t0 = time.perf_counter()
    best = None
    sweep = {}
    # thread counts: the physical cores first (the best point on every box seen so far), then every logical CPU (SMT
    # siblings included: on these boxes that point collapses, see cpu_baseline_only), then fractions
    phys = physical_cores() or maxthr
    phys = min(phys, maxthr)
    counts = [phys] + [t for t in (maxthr, phys // 2, phys // 4, 32, 16) if 1 <= t <= maxthr and t != phys]
    counts = list(dict.fromkeys(counts))
    for thr in (counts[:2] if quick else counts):
        nb = 2048  # 3 GB of stripes: well past the host's L3 (2 x 256 MB on the EPYC 9575F box)
        sec = co.bench_encode(K, M, S, nb, 3 if quick else 5, variant, thr)
        rate = nb * BLOCK_LEN / sec / 2**30
        sweep[str(thr)] = round(rate, 2)
        if best is None or rate > best[0]:
            best = (rate, thr, nb)
        if time.perf_counter() - t0 > 20:
            break
    if quick:
        return {"value": round(best[0], 2), "unit": "GiB/s", "cores": best[1], "threads_sweep_GiBps": sweep}
    scalar1 = 16 * BLOCK_LEN / co.bench_encode(K, M, S, 16, 3, co.SCALAR, 1) / 2**30
    simd1 = 64 * BLOCK_LEN / co.bench_encode(K, M, S, 64, 5, variant, 1) / 2**30
    return {
        "value": round(best[0], 2),
        "unit": "GiB/s",
        "cores": best[1],
        "kind": "port",
        "sample": f"{best[2]} blocks x 1 MiB RS(10,4) encode, median of 5 reps, "
                  f"{'avx2 split-nibble' if variant else 'scalar'} + OpenMP, every thread on the blocks it first-touched (NUMA) and then on what others have left; "
                  "C restatement of reed-solomon-erasure (not the Rust crate)",
        "threads_sweep_GiBps": sweep,
        "host_threads_available": maxthr,
        "one_thread_scalar_GiBps": round(scalar1, 3),
        "one_thread_simd_GiBps": round(simd1, 3),
    }


def cpu_backend_rate(S: int, nb: int = 512) -> dict:
    """The PRODUCT's own CPU backend (GEC_BACKEND_CPU, garage_amd/csrc/ec_cpu.cpp) on the same sample, through the C
    ABI (gec_encode_batch on a CPU codec): what a node without a GPU falls back to.  Reported beside cpu_baseline, never
    as it (the baseline is the oracle's restatement of the crate)."""
    import ctypes

    import numpy as np

    import garage_amd as g
    from garage_amd._lib import check, lib

    rs = g.ReedSolomon(K, M, backend="cpu")
    rng = np.random.default_rng(3)
    big = rng.integers(0, 256, nb * K * S, dtype=np.uint8)      # one arena for the blocks, one for the parity
    outb = np.zeros(nb * M * S, dtype=np.uint8)
    ptrs = (ctypes.c_void_p * nb)(*[big.ctypes.data + b * K * S for b in range(nb)])
    optrs = (ctypes.c_void_p * nb)(*[outb.ctypes.data + b * M * S for b in range(nb)])
    lens = (ctypes.c_size_t * nb)(*[BLOCK_LEN] * nb)
    # the encode alone and the encode with the 14 shard checksums of every stripe (what a put costs a node without a GPU), turn and
    # turn about: on a shared host the rate drifts by a quarter within seconds, and two figures taken one after the other would
    # mostly measure that
    sums = np.zeros(nb * (K + M) * 32, dtype=np.uint8)
    sp = sums.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))
    check(lib.gec_encode_batch(rs._h, nb, ptrs, lens, S, optrs), "gec_encode_batch")
    best = hbest = None
    for _ in range(6):
        t0 = time.perf_counter()
        check(lib.gec_encode_batch(rs._h, nb, ptrs, lens, S, optrs), "gec_encode_batch")
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
        try:
            t0 = time.perf_counter()
            check(lib.gec_encode_hash_batch(rs._h, nb, ptrs, lens, S, optrs, sp), "gec_encode_hash_batch")
            dt = time.perf_counter() - t0
            hbest = dt if hbest is None else min(hbest, dt)
        except Exception:  # noqa: BLE001
            hbest = None
            break
    return {"value": round(nb * BLOCK_LEN / best / 2**30, 2), "unit": "GiB/s", "kernel": lib.gec_cpu_isa().decode(),
            "encode_plus_14_checksums_GiBps": round(nb * BLOCK_LEN / hbest / 2**30, 2) if hbest else None,
            "cpus_allowed": len(os.sched_getaffinity(0)),
            "threads": int(os.environ.get("GEC_CPU_THREADS", "0")) or min(os.cpu_count() or 1, 16),
            "sample": f"{nb} blocks x 1 MiB RS(10,4) gec_encode_batch / gec_encode_hash_batch on a GEC_BACKEND_CPU codec, best of 6 each, interleaved"}


def _proc_snapshot():
    """(jiffies per field of the aggregate `cpu` line of /proc/stat, {pid: (comm, utime + stime)}) -- who else runs."""
    fields = ("user", "nice", "system", "idle", "iowait", "irq", "softirq", "steal")
    cpu = {}
    try:
        with open("/proc/stat") as f:
            parts = f.readline().split()
        cpu = {k: int(v) for k, v in zip(fields, parts[1:1 + len(fields)])}
    except (OSError, ValueError):
        pass
    procs = {}
    try:
        for pid in os.listdir("/proc"):
            if not pid.isdigit():
                continue
            try:
                with open(f"/proc/{pid}/stat") as f:
                    s = f.read()
                comm = s[s.index("(") + 1:s.rindex(")")]
                rest = s[s.rindex(")") + 2:].split()
                procs[int(pid)] = (comm, int(rest[11]) + int(rest[12]))
            except (OSError, ValueError, IndexError):
                continue
    except OSError:
        pass
    return cpu, procs


def host_load_during(before, after, wall_s: float) -> dict:
    """What the box did besides the oracle while it was timed: CPU seconds by kind (steal = taken by the hypervisor from
    this VM's vCPUs) and the other processes that used the most CPU -- the threads a collapsed team was waiting for."""
    hz = os.sysconf("SC_CLK_TCK") if hasattr(os, "sysconf") else 100
    cpu0, p0 = before
    cpu1, p1 = after
    me = os.getpid()
    out = {"wall_s": round(wall_s, 2)}
    if cpu0 and cpu1:
        out["cpu_seconds"] = {k: round((cpu1[k] - cpu0[k]) / hz, 2) for k in cpu1 if k in cpu0 and k in ("user", "system", "idle", "steal")}
    others = []
    for pid, (comm, t1) in p1.items():
        if pid == me:
            continue
        dt = (t1 - p0.get(pid, (comm, 0))[1]) / hz
        if dt >= 0.2:
            others.append((dt, comm, pid))
    others.sort(reverse=True)
    out["other_processes_cpu_s"] = [{"comm": c, "pid": p, "cpu_s": round(d, 2)} for d, c, p in others[:5]]
    try:
        out["loadavg_after"] = open("/proc/loadavg").read().split()[:3]
    except OSError:
        pass
    return out


def cpu_baseline_only() -> None:
    """`bench.py --cpu-baseline-only`: the CPU figures in a process of their own -- nothing of HIP or torch is loaded
    while the oracle is timed (the in-process figure of round 2 fell from 388 to 259 GiB/s at 64 threads and from 333
    to 21 at 128 with an unchanged oracle: whatever else the bench process had running took part)."""
    S = (BLOCK_LEN + K - 1) // K
    S = (S + 63) // 64 * 64
    snap0, t0 = _proc_snapshot(), time.time()
    out = cpu_baseline(S)
    out["host_load_during_sweep"] = host_load_during(snap0, _proc_snapshot(), time.time() - t0)
    out["host"] = host_description()
    out["process"] = "fresh subprocess, before any HIP / torch initialisation; OMP_PROC_BIND=close OMP_PLACES=cores"
    # The point with one thread per LOGICAL CPU collapsed by 16x in round 2's line (and does here whenever it is run):
    # with every CPU of the box inside a libgomp team that spin-waits at its barriers, anything else that becomes
    # runnable (the gpurun agent, the driver's rocm-smi sampler, kernel threads) preempts one team member and the
    # others spin for it.  Shown by running that one point again with a passive wait policy and with one CPU left free.
    allcpus = out["host_threads_available"]
    rate_all = out["threads_sweep_GiBps"].get(str(allcpus))
    if rate_all is not None and rate_all < 0.5 * out["value"]:
        probe = {}
        for name, env_add, n in (("passive_wait", {"OMP_WAIT_POLICY": "passive"}, allcpus),
                                 ("two_cpus_left_free", {}, max(1, allcpus - 2))):
            try:
                env = dict(os.environ, **env_add)
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-threads-probe", str(n)], env=env, capture_output=True,
                                   text=True, timeout=120)
                probe[name] = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
            except Exception as e:  # noqa: BLE001
                probe[name] = {"error": f"{type(e).__name__}: {e}"[:200]}
        out["all_logical_cpus_point"] = {"threads": allcpus, "GiBps_active_wait": rate_all, **probe,
                                         "reading": "every logical CPU inside a spin-waiting OpenMP team: any other runnable thread on the box "
                                                    "preempts a team member and the rest spin at the barrier; not a property of the encode loop"}
    print(json.dumps(out), flush=True)


def cpu_baseline_subprocess():
    """Runs cpu_baseline_only() in a fresh interpreter with an explicit OpenMP placement; None on failure."""
    env = dict(os.environ, OMP_PROC_BIND="close", OMP_PLACES="cores")
    env.pop("OMP_NUM_THREADS", None)
    env.setdefault("GEC_CPU_THREADS", str(min(os.cpu_count() or 1, 64)))
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only"], env=env, capture_output=True, text=True,
                           timeout=300)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
        out = json.loads(line)
    except Exception as e:  # noqa: BLE001 -- a reported baseline must never cost the headline line
        return {"error": f"{type(e).__name__}: {e}"[:300]}
    # The product's own CPU backend, in a process of ITS own started from here: libgomp binds the main thread of the
    # oracle's process to the first place (OMP_PROC_BIND=close), and both the threads created from it and the
    # processes forked from it inherit that one-core affinity mask -- a CPU codec's pool measured there runs on one core.
    try:
        env = {k: v for k, v in env.items() if not k.startswith(("OMP_", "GOMP_"))}
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-backend-only"], env=env, capture_output=True, text=True, timeout=240)
        out["cpu_backend"] = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    except Exception as e:  # noqa: BLE001
        out["cpu_backend"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    return out


def cpu_baseline_object(args, S: int):
    """cpu_baseline of the line: the subprocess figure (taken before this process touched the GPU), with the
    in-process figure -- same oracle, same sample, measured now, HIP runtime and torch loaded -- beside it."""
    out = getattr(args, "cpu_pre", None)
    if not out or "error" in out:
        base = cpu_baseline(S)
        base["process"] = "in-process (the subprocess measurement failed: " + str((out or {}).get("error")) + ")"
        return base
    try:
        out["in_process"] = cpu_baseline(S, quick=True)
        out["in_process"]["process"] = "the bench process itself, after the GPU work (HIP runtime, torch and the codec's threads loaded)"
    except Exception as e:  # noqa: BLE001
        out["in_process"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    return out


def oracle_check_sample(st, nb: int, S: int) -> int:
    """After (and outside) the timed region: the parity the timed kernel left in HBM for a
    strided sample of this rank's batch, byte for byte against the CPU oracle.  Returns the
    number of blocks compared; raises on any mismatch."""
    import numpy as np
    import torch

    from oracle import rs_oracle as O

    if nb == 0:
        return 0
    take = nb if ORACLE_SAMPLE <= 0 else min(ORACLE_SAMPLE, nb)
    idx = sorted(set(np.linspace(0, nb - 1, take).astype(int).tolist()))
    co = O.COracle()
    kind = co.AVX2 if co.has_avx2() else co.SCALAR
    threads = max(1, min(16, os.cpu_count() or 1))
    for i0 in range(0, len(idx), 128):  # 128 blocks = 188 MB of stripes at a time
        part = idx[i0:i0 + 128]
        host = st[torch.as_tensor(part, device=st.device)].cpu().numpy()
        want = co.encode_batch(K, M, np.ascontiguousarray(host[:, :K]), kind, threads=threads)
        if not np.array_equal(host[:, K:], want):
            raise AssertionError("bench output differs from the CPU oracle")
    return len(idx)


# --------------------------------------------------------------------------- workload
def make_stripes(dev, rank: int, nb: int, S: int):
    """This rank's batch in the stripe layout, payload = seeded random bytes, the two edge
    blocks of SURVEY.md section 8d (all-zero, all-0xFF) first."""
    import torch

    n = K + M
    gen = torch.Generator(device=dev)
    gen.manual_seed(0x6761726167650002 + rank)
    st = torch.zeros((nb, n, S), dtype=torch.uint8, device=dev)
    if nb:
        payload = st[:, :K].reshape(nb, K * S)
        payload[:, :BLOCK_LEN] = torch.randint(0, 256, (nb, BLOCK_LEN), dtype=torch.uint8, device=dev, generator=gen)
    if nb >= 2:
        st[0, :K] = 0
        st[1, :K].reshape(-1)[:BLOCK_LEN] = 0xFF
    return st


class EncodeJob:
    """One GPU's share of the encode bench: buffers, codec, the step, and the timed loop.
    Used by one process per GPU (procs mode) or one thread per GPU (threads mode)."""

    def __init__(self, args, device_index: int, rank: int, nb: int, stream=None):
        import torch

        import garage_amd as g

        self.torch, self.g = torch, g
        self.args, self.rank, self.nb = args, rank, nb
        self.dev = torch.device("cuda", device_index)
        self.S = g.shard_len(K, BLOCK_LEN)
        self.rs = g.ReedSolomon(K, M, device=device_index)
        self.st = make_stripes(self.dev, rank, nb, self.S)
        self.stream = stream if stream is not None else torch.cuda.current_stream(self.dev)
        self.lib, self.h = g._lib.lib, self.rs._h
        self.base = self.st.data_ptr() if nb else 0

    def step(self):
        n, S = K + M, self.S
        rc = self.lib.gec_encode_batch_dev(self.h, self.nb, self.base, n * S, S, self.base + K * S, n * S,
                                           self.stream.cuda_stream)
        if rc:
            self.g._lib.check(rc, "gec_encode_batch_dev")

    def sync(self):
        self.stream.synchronize()

    def timed_burst(self, steps: int) -> float:
        """avg ms per launch over `steps` back-to-back launches (HIP events on the launch stream)"""
        ev0, ev1 = self.torch.cuda.Event(enable_timing=True), self.torch.cuda.Event(enable_timing=True)
        ev0.record(self.stream)
        for _ in range(steps):
            self.step()
        ev1.record(self.stream)
        self.sync()
        return ev0.elapsed_time(ev1) / steps

    def cold_burst(self):
        """The first 50 launches from an idle device: module load and first-touch are paid by three
        untimed launches, then the device idles 0.3 s so the clocks drop back (DESIGN.md 'DVFS transient')."""
        for _ in range(3):
            self.step()
        self.sync()
        time.sleep(0.3)
        return self.timed_burst(50)

    def precondition_and_warm(self):
        a = self.args
        if a.precondition_ms > 0:
            tpre = time.perf_counter()
            while (time.perf_counter() - tpre) * 1e3 < a.precondition_ms:
                for _ in range(20):
                    self.step()
                self.sync()
        for _ in range(a.warmup):
            self.step()
        self.sync()

    def decode_setup(self, lost=(0, 3, 7, 9)):
        """Erase the shards of `lost` in every block of the (encoded) batch; keeps what a decode must bring back."""
        import numpy as np

        self.lost = tuple(lost)
        self.present = np.array([j not in self.lost for j in range(K + M)], dtype=np.uint8)
        self.lost_ref = self.st[:, list(self.lost)].clone()   # the shards about to be erased: what a decode must return
        self.st[:, list(self.lost)] = 0

    def decode_step(self):
        with self.torch.cuda.stream(self.stream):
            self.rs.reconstruct_dev(self.st, self.present)

    def decode_erase(self):
        self.st[:, list(self.lost)] = 0


def algo_bytes(nb: int, S: int) -> int:
    return (K + M) * S * nb  # read k*S + write m*S per block


def roofline_obj(args, nb: int, S: int, kern_ms: float, cold_ms):
    ab = algo_bytes(nb, S)
    achieved = ab / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
    traffic, source = measured_traffic(nb) if args.variant == 0 else (None, None)
    r = {
        "bound": "hbm",
        "achieved": round(achieved, 1),
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 4),
        "traffic": traffic,
        "traffic_source": source,
        "kernel": "gf_apply_nibble<1,0,10,1,true,256>" if args.variant == 0 else "gf_apply_logexp<0>",
        "kernel_ms": round(kern_ms, 4),
        "algorithmic_bytes_per_launch": ab,
        "secondary": secondary_bounds() if args.variant == 0 else None,
    }
    if cold_ms:
        r["cold_burst_frac"] = round(ab / (cold_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        r["cold_burst"] = "first 50 launches from an idle device, no pre-conditioning (DVFS transient, DESIGN.md section 4)"
    return r


def base_line(args, world: int, elapsed: float, blocks_all: int, per_rank, nb0: int, S: int, mode: str):
    total_blocks = args.batch * world
    return {
        "metric": "RS(10,4) encode payload throughput, 1 MiB blocks",
        "value": round(blocks_all * BLOCK_LEN * args.steps / elapsed / 2**30, 2),
        "unit": "GiB/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "preconditioning_ms": args.precondition_ms,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u8",
        "data": "synthetic",
        "config": {
            "workload": "BASELINE config 2: RS(10,4) encode, 1 MiB blocks, batch 1024 per GPU, device-resident"
                        if world == 1 else
                        "BASELINE config 4: RS(10,4) encode, 1 MiB blocks, hash-partitioned stream of "
                        f"{total_blocks} blocks over {world} GPUs (hash[4] % N), no collective",
            "k": K, "m": M, "block_len": BLOCK_LEN, "shard_len": S,
            "blocks_total": blocks_all, "blocks_rank0": nb0, "blocks_per_rank": per_rank,
            "kernel_variant": args.variant,
            "parallelism": f"hash-partition x{world}",
            "mode": mode,
        },
    }


def _timed_on_stream(torch, stream, fn, steps: int, warm_ms: float) -> float:
    """avg ms per call of fn over `steps` back-to-back calls, HIP events on the launch stream, behind `warm_ms` of the same calls"""
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < warm_ms:
        for _ in range(10):
            fn()
        stream.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record(stream)
    for _ in range(steps):
        fn()
    ev1.record(stream)
    stream.synchronize()
    return ev0.elapsed_time(ev1) / steps


def static_pmc(key: str):
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            return json.load(f).get(key)
    except (OSError, ValueError):
        return None


def encode_hash_object(args, job) -> dict:
    """The device-resident PUT trip on config 2: RS(10,4) encode AND the checksum of all 14 shards of every block
    (gec_encode_hash_batch_dev) -- one pass: the encode kernel's SUM form leaves the MLH64 leaf sums from its registers, one lane
    per shard makes the roots.  Checked after the timed loop: parity against the C oracle and all 14 checksums against
    oracle/mlh64.py on a strided sample of blocks."""
    import numpy as np

    from oracle import mlh64
    from oracle import rs_oracle as O

    torch, g = job.torch, job.g
    n, S, nb = K + M, job.S, job.nb
    sums = torch.empty((nb, n, 32), dtype=torch.uint8, device=job.dev)
    rs3 = job.rs if job.rs.shardsum_kind == 3 else job.rs.with_shardsum(3)

    def step():
        rc = job.lib.gec_encode_hash_batch_dev(rs3._h, nb, job.base, n * S, S, sums.data_ptr(), job.stream.cuda_stream)
        if rc:
            g._lib.check(rc, "gec_encode_hash_batch_dev")

    job.st[:, K:] = 0
    steps = max(20, min(args.steps, 200))
    ms = _timed_on_stream(torch, job.stream, step, steps, min(args.precondition_ms, 100.0))
    enc_ms = _timed_on_stream(torch, job.stream, job.step, steps, 0.0)   # the encode alone, same clocks, for the ratio
    # -- after the timed loops
    idx = sorted(set(np.linspace(0, nb - 1, 16).astype(int).tolist()))
    host = job.st[torch.as_tensor(idx, device=job.dev)].cpu().numpy()
    hs = sums[torch.as_tensor(idx, device=job.dev)].cpu().numpy()
    co = O.COracle()
    want = co.encode_batch(K, M, np.ascontiguousarray(host[:, :K]), co.AVX2 if co.has_avx2() else co.SCALAR, threads=4)
    ok = bool(np.array_equal(host[:, K:], want))
    ok = ok and all(hs[i, j].tobytes() == mlh64.shardsum3(host[i, j].tobytes()) for i in range(len(idx)) for j in range(n))
    assert bool(job.rs.verify_dev(job.st).all()), "verify failed after encode_hash"
    ab = algo_bytes(nb, S)
    out = {
        "what": "BASELINE config 2 with the checksum of every shard: gec_encode_hash_batch_dev, RS(10,4), 1 MiB blocks, device-resident "
                "(shard checksum v3 = MLH64: leaf sums accumulated by the encode kernel from its registers + one root kernel)",
        "value": round(nb * BLOCK_LEN / (ms * 1e-3) / 2**30, 2), "unit": "GiB/s", "ms": round(ms, 4), "blocks": nb, "steps": steps,
        "checksums_per_launch": nb * n, "encode_alone_ms": round(enc_ms, 4), "over_encode_alone": round(ms / enc_ms, 3) if enc_ms else None,
        "roofline": {"bound": "hbm", "achieved": round(ab / (ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(ab / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_launch": ab,
                     "note": "same algorithmic bytes as the encode (read k*S, write m*S): the checksums read nothing twice; "
                             "both launches of the trip are inside `ms`",
                     "kernels": ["gf_apply_nibble_sum<1,0,10,true,256>", "mlh_roots_quad"]},
        "binding_resource": static_pmc("rs10_4_encode_hash_secondary_bounds"),
        "traffic": (static_pmc("rs10_4_encode_hash_1MiB_x1024") or {}).get("traffic_bytes"),
        "bit_exact": ok, "checked": f"{len(idx)} strided blocks: parity vs the C oracle, all {n} checksums vs oracle/mlh64.py; gec_verify_batch_dev over the batch",
    }
    if not ok:
        out["error"] = "encode_hash output differs from the oracles"
    return out


def rs20_8_object(args, dev, stream) -> dict:
    """BASELINE config 5's code on one GPU: RS(20,8) encode of 256 x 4 MiB blocks, device-resident; every block's parity against
    the C oracle after the timed loop."""
    import numpy as np
    import torch

    import garage_amd as g
    from oracle import rs_oracle as O

    k, m, L, nb = 20, 8, 4 << 20, 256
    n, S = k + m, g.shard_len(k, 4 << 20)
    rs = g.ReedSolomon(k, m, device=dev.index)
    gen = torch.Generator(device=dev)
    gen.manual_seed(0x6761726167650005)
    st = torch.zeros((nb, n, S), dtype=torch.uint8, device=dev)
    st[:, :k].reshape(nb, k * S)[:, :L] = torch.randint(0, 256, (nb, L), dtype=torch.uint8, device=dev, generator=gen)
    base = st.data_ptr()
    lib = g._lib.lib

    def step():
        rc = lib.gec_encode_batch_dev(rs._h, nb, base, n * S, S, base + k * S, n * S, stream.cuda_stream)
        if rc:
            g._lib.check(rc, "gec_encode_batch_dev")

    steps = max(20, min(args.steps, 200))
    ms = _timed_on_stream(torch, stream, step, steps, min(args.precondition_ms, 100.0))
    host = st.cpu().numpy()
    co = O.COracle()
    want = co.encode_batch(k, m, np.ascontiguousarray(host[:, :k]), co.AVX2 if co.has_avx2() else co.SCALAR, threads=max(1, min(16, os.cpu_count() or 1)))
    ok = bool(np.array_equal(host[:, k:], want))
    ab = n * S * nb
    rec = static_pmc("rs20_8_encode_4MiB_x256") or {}
    out = {
        "what": "BASELINE config 5's code on one GPU: RS(20,8) encode, 256 x 4 MiB blocks, device-resident",
        "value": round(nb * L / (ms * 1e-3) / 2**30, 2), "unit": "GiB/s", "kernel_ms": round(ms, 4), "blocks": nb, "steps": steps, "shard_len": S,
        "roofline": {"bound": "hbm", "achieved": round(ab / (ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(ab / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_launch": ab,
                     "traffic": rec.get("traffic_bytes"), "traffic_source": "static: profiles/pmc_traffic.json (" + rec.get("source", "") + ")" if rec else None,
                     "kernel": "gf_apply_nibble<2,0,5,1,true,512>", "secondary": static_pmc("rs20_8_secondary_bounds")},
        "bit_exact": ok, "checked": f"parity of all {nb} blocks vs the C oracle, after the timed loop",
    }
    if not ok:
        out["error"] = "RS(20,8) parity differs from the C oracle"
    return out


DECODE_PATTERNS = ((0, 3, 7, 9), (0, 3, 7, 11))


def decode_object(args, job, R, distrib, barrier, blocks_all: int, world: int) -> dict:
    """BASELINE config 3 on the rank's resident batch, once per erasure pattern: erase, ONE cold call (host 10x10 inversion
    included), a timed loop of warm calls (cached decode matrix; wall clock MAX over ranks for `value`, HIP events on the launch
    stream for the roofline), and after the loop every rebuilt shard of every block against the shards that were erased, the
    kernel's own compare mode over the completed stripes, and the CPU oracle's reconstruct on a strided sample of blocks."""
    import numpy as np
    import torch

    from oracle import rs_oracle as O

    S, nb = job.S, job.nb
    dsteps = max(5, args.steps // 2)
    per = []
    for lost in DECODE_PATTERNS:
        job.decode_setup(lost)
        torch.cuda.synchronize()
        c0 = time.perf_counter()
        job.decode_step()                     # cold: includes the host 10x10 inversion
        torch.cuda.synchronize()
        cold_dec_ms = (time.perf_counter() - c0) * 1e3
        assert torch.equal(job.st[:, list(lost)], job.lost_ref), "reconstruct mismatch (first call)"
        job.decode_erase()
        barrier()
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        d0 = time.perf_counter()
        ev0.record(job.stream)
        for _ in range(dsteps):
            job.decode_step()                 # warm: cached decode matrix
        ev1.record(job.stream)
        torch.cuda.synchronize()
        barrier()
        dt = distrib.max_over_ranks(R, time.perf_counter() - d0)
        kern_ms = ev0.elapsed_time(ev1) / dsteps
        # -- after the timed loop
        exact = bool(torch.equal(job.st[:, list(lost)], job.lost_ref))
        exact = exact and bool(job.rs.verify_dev(job.st).all())
        idx = sorted(set(np.linspace(0, nb - 1, min(nb, 16)).astype(int).tolist()))
        host = job.st[torch.as_tensor(idx, device=job.dev)].cpu().numpy()
        co = O.COracle()
        broken = host.copy()  # the oracle rebuilds the same erasures from the survivors alone
        broken[:, list(lost)] = 0
        want = co.reconstruct_batch(K, M, broken, [j not in lost for j in range(K + M)], threads=4)
        exact = exact and bool(np.array_equal(want, host))
        exact_all = distrib.sum_over_ranks(R, int(exact)) == world
        ab = (K + len(lost)) * S * nb     # read k surviving shards, write the 4 lost ones: the encode's algorithmic bytes
        per.append({
            "lost": list(lost),
            "value": round(blocks_all * BLOCK_LEN * dsteps / dt / 2**30, 2), "unit": "GiB/s",
            "ms_per_step": round(dt / dsteps * 1e3, 4), "kernel_ms": round(kern_ms, 4), "cold_first_call_ms": round(cold_dec_ms, 3),
            "roofline": {"bound": "hbm", "achieved": round(ab / (kern_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(ab / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_launch": ab,
                         "kernel": "gf_apply_nibble<1,0,10,1,true,256>", "kernel_ms": round(kern_ms, 4),
                         "note": "one launch rebuilds all four shards (a lost parity row is composed with the decode matrix on the host)"},
            "bit_exact": exact_all,
        })
    first = per[0]
    out = {
        "workload": "BASELINE config 3: RS(10,4) reconstruct with 4 erasures, 1 MiB blocks, the rank's resident batch; patterns "
                    + ", ".join("{" + ",".join(map(str, p["lost"])) + "}" for p in per),
        "value": first["value"], "unit": "GiB/s", "ms_per_step": first["ms_per_step"], "cold_first_call_ms": first["cold_first_call_ms"],
        "roofline_frac": first["roofline"]["frac"], "roofline": first["roofline"],
        "bit_exact": all(p["bit_exact"] for p in per),
        "checked": "after each timed loop: every rebuilt shard of every block equals the erased one, gec_verify_batch_dev over the "
                   "completed stripes, and the CPU oracle's reconstruct of 16 strided blocks from the survivors alone",
        "patterns": per,
    }
    if not out["bit_exact"]:
        out["error"] = "a decode result differs from the erased shards / the oracle"
    return out


# --------------------------------------------------------------- one process per GPU
def run_procs(args) -> None:
    import numpy as np
    import torch

    import garage_amd as g
    from garage_amd import distrib
    from garage_amd.partition import gpu_of_hash

    assert torch.cuda.is_available(), "bench.py needs a GPU (libgarage_ec has no CPU path)"
    R = distrib.init_from_env()
    world, rank = R.world, R.rank
    if world != args.gpus:
        sys.exit(f"WORLD_SIZE={world} does not match --gpus {args.gpus}")
    g.set_kernel_variant(args.variant)
    if args.op == "striped-decode":
        # the same watchdog as the striped_decode object of the default run: a collective that never returns must not
        # keep N processes alive until the driver's own limit
        box, done = {}, threading.Event()

        def give_up():
            if not done.wait(args.striped_timeout):
                if rank == 0:
                    print(json.dumps({"metric": "RS(20,8) striped-object decode", "n_gpus": world,
                                      "error": f"no result within {args.striped_timeout:.0f} s (watchdog)", **box}), flush=True)
                os._exit(0)

        threading.Thread(target=give_up, daemon=True).start()
        try:
            out = striped_decode_bench(args, R, distrib, box)
        except Exception as e:  # noqa: BLE001
            done.set()
            if rank == 0:
                print(json.dumps({"metric": "RS(20,8) striped-object decode", "n_gpus": world,
                                  "error": f"{type(e).__name__}: {e}"[:400], **box}), flush=True)
            sys.stdout.flush()
            os._exit(0)   # peers may be inside a collective: do not wait for them in the teardown
        done.set()
        if rank == 0:
            print(json.dumps(out), flush=True)
        distrib.shutdown(R)
        return

    # ---- hash-partition the (synthetic) PutObject block stream over the GPUs
    total_blocks = args.batch * world
    hashes = synthetic_hashes(total_blocks)
    mine = np.nonzero(gpu_of_hash(hashes, world) == rank)[0]
    nb = int(mine.size)
    job = EncodeJob(args, R.local_rank, rank, nb)
    S = job.S

    def barrier():
        distrib.barrier(R)

    cold_ms = job.cold_burst() if (rank == 0 and nb) else None
    barrier()
    job.precondition_and_warm()
    torch.cuda.synchronize()
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record(job.stream)
    for _ in range(args.steps):
        job.step()
    ev1.record(job.stream)
    torch.cuda.synchronize()
    barrier()
    t1 = time.perf_counter()
    kern_ms = ev0.elapsed_time(ev1) / args.steps  # avg launch duration, HIP events on the launch stream

    elapsed = distrib.max_over_ranks(R, t1 - t0)      # MAX over ranks
    blocks_all = distrib.sum_over_ranks(R, nb)        # units all ranks processed
    per_rank = distrib.gather_ints(R, nb)
    kern_ns_all = distrib.gather_ints(R, int(round(kern_ms * 1e6)))   # every GPU's own average launch duration
    rccl = distrib.count_ranks(R)                     # an all-reduce of ones on the device: RCCL saw this many ranks

    # correctness gates, after and outside the timed region: the kernel's own compare mode over the
    # whole batch, and the CPU oracle byte for byte on a strided sample
    assert bool(job.rs.verify_dev(job.st).all()), "verify failed on bench output"
    checked = 0 if args.no_oracle_check else oracle_check_sample(job.st, nb, S)
    checked_all = distrib.sum_over_ranks(R, checked)

    # ---- decode (BASELINE config 3): 4 shards lost, reconstruct in place -- both patterns BASELINE.md section 3 names:
    #      four data shards {0,3,7,9}, and three data + one parity {0,3,7,11} (the decode matrix known answer of SURVEY.md A.4.7)
    decode = None
    if not args.no_decode and nb:
        decode = decode_object(args, job, R, distrib, barrier, blocks_all, world)

    # ---- N = 1: the put trip (config 2 + the checksum of every shard) and config 5's code, beside the headline
    extra = {}
    if world == 1 and rank == 0 and nb and not args.no_extra and args.variant == 0:
        for key, fn in (("encode_hash", lambda: encode_hash_object(args, job)), ("rs20_8_encode", lambda: rs20_8_object(args, R.device, job.stream))):
            try:
                extra[key] = fn()
            except Exception as e:  # noqa: BLE001 -- a secondary object must never cost the headline line
                extra[key] = {"error": f"{type(e).__name__}: {e}"[:400]}

    out = None
    if rank == 0:
        out = base_line(args, world, elapsed, blocks_all, per_rank, nb, S,
                        "one process per GPU (torch.distributed.run)" + (", self-launched" if os.environ.get("GARAGE_BENCH_SELF_LAUNCHED") else ""))
        out["roofline"] = roofline_obj(args, nb, S, kern_ms, cold_ms)
        if world > 1:  # the fraction of the HBM roofline on every GPU, not only rank 0's
            out["kernel_ms_per_gpu"] = [round(x / 1e6, 4) for x in kern_ns_all]
            out["roofline_frac_per_gpu"] = [round(algo_bytes(per_rank[i], S) / (kern_ns_all[i] / 1e9) / 1e9 / HBM_PEAK_GBS, 4)
                                            if kern_ns_all[i] else None for i in range(world)]
        out["parity_checked_blocks"] = checked_all
        out["parity_check"] = ("CPU oracle byte-for-byte on "
                               + ("every block" if checked_all == blocks_all else f"{checked_all} strided blocks")
                               + " of every rank's timed batch (the parity the timed kernel left in HBM), after the timed region"
                               "; gec_verify_batch_dev over every block beside it")
        out["rccl_ranks"] = rccl["ranks"]
        out["collective_backend"] = rccl["backend"]
        if decode:
            out["decode"] = decode
        out.update(extra)
    del job
    torch.cuda.empty_cache()

    # ---- every GPU fed from host memory at once (N>1: the question one host with N x16 links has to answer)
    if (world > 1 and not args.no_host_fed) or args.host_fed:
        try:
            rs_h = g.ReedSolomon(K, M, device=R.local_rank)
            reps = max(2, min(8, args.steps // 100 or 2))
            sec = host_fed_section(rs_h, args.host_blocks, reps, barrier, seed=100 + rank)
            per = []
            cols = {}
            for kind in ("pinned", "pageable"):
                cols[kind] = {"t0": distrib.gather_ints(R, int(sec[kind]["t0"] * 1e6)), "t1": distrib.gather_ints(R, int(sec[kind]["t1"] * 1e6)),
                              "rate": distrib.gather_ints(R, int(sec[kind]["GiBps"] * 1000)),
                              "exact": distrib.gather_ints(R, int(sec[kind]["bit_exact"])), "checked": distrib.gather_ints(R, sec[kind]["checked"])}
            lane_nodes = distrib.gather_ints(R, rs_h.numa_node)      # where each rank's codec keeps its host side (-1: not placed)
            if rank == 0:
                for q in range(world):
                    per.append({kind: {"t0": cols[kind]["t0"][q] / 1e6, "t1": cols[kind]["t1"][q] / 1e6, "GiBps": cols[kind]["rate"][q] / 1000,
                                       "bit_exact": bool(cols[kind]["exact"][q]), "checked": cols[kind]["checked"][q]}
                                for kind in ("pinned", "pageable")})
                out["host_fed"] = host_fed_object(per, args.host_blocks, reps)
                out["host_fed"]["numa_node_per_gpu"] = lane_nodes
                out["host_fed"]["numa"] = ("each codec's copy threads and pinned staging memory are on its device's memory node "
                                           "(GEC_NUMA, garage_amd/csrc/numa.hpp); -1 = not placed (one node, or switched off)")
            del rs_h
        except Exception as e:  # noqa: BLE001 -- a secondary object must never cost the headline line
            if rank == 0:
                out["host_fed"] = {"error": f"{type(e).__name__}: {e}"[:400]}
        # ... and the node's N devices through the PRODUCT's multi-device manager: ONE process (rank 0 starts it; the
        # other ranks idle at the barrier meanwhile), gbm_create_multi over the N codecs, one coalescing queue per device,
        # native callers (tools/multi_bench) -- "blocks from a batched PutObject stream are hash-partitioned across the
        # GPUs of one node" as the daemon would run it, per device and summed, beside the raw per-rank figures above
        try:
            if rank == 0 and isinstance(out.get("host_fed"), dict):
                out["host_fed"]["block_manager_multi"] = multi_manager_rates(world, os.environ.get("GARAGE_DRYRUN_ONE_GPU") == "1")
        except Exception as e:  # noqa: BLE001
            if rank == 0:
                out["host_fed"]["block_manager_multi"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        barrier()

    # ---- BASELINE config 5 beside it when there is more than one GPU (or on request): the path's one
    # real exchange step.  Guarded by a watchdog so that a collective that hangs cannot take the
    # headline line with it.
    # Every rank runs it in a CHILD process (its own process group on another port): the exchanges have never met N > 1 hardware, and
    # neither a collective that hangs (the child's watchdog), nor one that cannot start, nor a process that dies inside RCCL or on
    # a peer's memory may take the headline line -- which is not printed yet -- with it.
    # (GARAGE_DRYRUN_ONE_GPU=1: the same flow over a gloo-backed caller transport, labelled as such in the object)
    if (world > 1 and not args.no_striped) or args.striped:
        res = striped_decode_in_children(args, R)
        if rank == 0:
            out["striped_decode"] = res
        barrier()

    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_object(args, S)
        if world == 1 and not args.no_host_path:
            out.update(host_path_objects())
        print(json.dumps(out), flush=True)
    distrib.shutdown(R)


# ------------------------------------------------- one process, N codecs, N host threads
def run_threads(args) -> None:
    import numpy as np
    import torch

    import garage_amd as g
    from garage_amd.partition import gpu_of_hash

    assert torch.cuda.is_available(), "bench.py needs a GPU (libgarage_ec has no CPU path)"
    world = args.gpus
    ndev = torch.cuda.device_count()
    dry = os.environ.get("GARAGE_DRYRUN_ONE_GPU") == "1"
    if ndev < world and not dry:
        sys.exit(f"--gpus {world} --mode threads needs {world} visible GPUs, found {ndev} "
                 "(GARAGE_DRYRUN_ONE_GPU=1 puts every codec on device 0: plumbing only, numbers meaningless)")
    g.set_kernel_variant(args.variant)
    total_blocks = args.batch * world
    owner = gpu_of_hash(synthetic_hashes(total_blocks), world)
    S = g.shard_len(K, BLOCK_LEN)
    res = [None] * world
    err = []
    bar = threading.Barrier(world)
    hf_reps = max(2, min(8, args.steps // 100 or 2))

    def worker(t: int):
        try:
            d = 0 if dry else t
            torch.cuda.set_device(d)
            stream = torch.cuda.Stream(device=d)
            nb = int(np.count_nonzero(owner == t))
            with torch.cuda.stream(stream):
                job = EncodeJob(args, d, t, nb, stream=stream)
            stream.synchronize()
            cold = job.cold_burst() if (t == 0 and nb) else None
            bar.wait()
            job.precondition_and_warm()
            bar.wait()
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            ev0.record(stream)
            for _ in range(args.steps):
                job.step()
            ev1.record(stream)
            stream.synchronize()
            t1 = time.perf_counter()
            bar.wait()
            with torch.cuda.stream(stream):
                ok = bool(job.rs.verify_dev(job.st).all())
                checked = 0 if args.no_oracle_check else oracle_check_sample(job.st, nb, S)
            if not ok:
                raise AssertionError("verify failed on bench output")
            hf = None
            if (world > 1 and not args.no_host_fed) or args.host_fed:
                # N host threads, N codecs, N links at once: how a single Garage daemon would feed the node's GPUs
                job.st = None
                hf = host_fed_section(job.rs, args.host_blocks, hf_reps, bar.wait, seed=100 + t)
            res[t] = {"t0": t0, "t1": t1, "nb": nb, "kern_ms": ev0.elapsed_time(ev1) / args.steps, "cold": cold,
                      "checked": checked, "host_fed": hf, "numa_node": job.rs.numa_node}
        except BaseException as e:  # noqa: BLE001
            err.append(f"thread {t}: {type(e).__name__}: {e}")
            bar.abort()

    th = [threading.Thread(target=worker, args=(t,)) for t in range(world)]
    [x.start() for x in th]
    [x.join() for x in th]
    if err:
        sys.exit("; ".join(err))
    elapsed = max(r["t1"] for r in res) - min(r["t0"] for r in res)  # barrier release -> last thread done
    blocks_all = sum(r["nb"] for r in res)
    out = base_line(args, world, elapsed, blocks_all, [r["nb"] for r in res], res[0]["nb"], S,
                    f"threads: one process, {world} codecs, {world} host threads through the C ABI"
                    + (" (DRY RUN: all on device 0)" if dry else ""))
    out["roofline"] = roofline_obj(args, res[0]["nb"], S, res[0]["kern_ms"], res[0]["cold"])
    out["kernel_ms_per_gpu"] = [round(r["kern_ms"], 4) for r in res]
    out["roofline_frac_per_gpu"] = [round(algo_bytes(r["nb"], S) / (r["kern_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if r["kern_ms"] else None
                                    for r in res]
    out["parity_checked_blocks"] = sum(r["checked"] for r in res)
    out["parity_check"] = ("CPU oracle byte-for-byte on "
                           + ("every block" if out["parity_checked_blocks"] == sum(r["nb"] for r in res) else f"{out['parity_checked_blocks']} strided blocks")
                           + " of every GPU's timed batch, after the timed region; gec_verify_batch_dev over every block beside it")
    out["rccl_ranks"] = None
    out["collective_backend"] = "none (single process; the encode path has no collective)"
    if all(r["host_fed"] for r in res):
        out["host_fed"] = host_fed_object([r["host_fed"] for r in res], args.host_blocks, hf_reps)
        out["host_fed"]["numa_node_per_gpu"] = [r["numa_node"] for r in res]
        # ... and the same node through the PRODUCT's multi-device manager: gbm_create_multi over the N codecs, one
        # coalescing queue per device, native callers through gbm_batcher_put_block / _get_block (tools/multi_bench), beside
        # the raw gec_encode_hash_batch figures above
        out["host_fed"]["block_manager_multi"] = multi_manager_rates(world, dry)
    # the path's one collective, from this one process: a gec_group over the N devices (N threads, N codecs), an
    # oracle-checked striped decode through it -- `rccl_ranks` is what RCCL connected
    if (world > 1 and not args.no_striped) or args.striped:
        try:
            grp = threads_group_check(world, dry)
            out["striped_decode"] = grp
            out["rccl_ranks"] = grp.get("rccl_ranks")
            out["collective_backend"] = grp.get("transport")
        except Exception as e:  # noqa: BLE001
            out["striped_decode"] = {"error": f"{type(e).__name__}: {e}"[:400]}
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_object(args, S)
    print(json.dumps(out), flush=True)


def threads_group_check(world: int, dry: bool) -> dict:
    """One process, `world` threads, one RS(20,8) codec and one gec_group rank per device: create the group over RCCL
    (ncclCommInitRank from every thread with rank 0's unique id), decode 8 oracle-encoded 4 MiB objects striped over
    the ranks through both exchanges, compare every rebuilt shard on every rank with the oracle's stripes.
    dry (GARAGE_DRYRUN_ONE_GPU=1: every codec on device 0, where RCCL refuses a second rank): the same flow over the
    loopback gec_allgather_fn of tests/c (test transport, plumbing only)."""
    import ctypes

    import numpy as np
    import torch

    import garage_amd as g
    from garage_amd.striped import StripeLayout, gather_stripes, scatter_stripes
    from oracle import rs_oracle as O

    k, m, L, nobj = 20, 8, 4 << 20, 8
    S = g.shard_len(k, L)
    layout = StripeLayout(k, m, world)
    lost = (0, 1, 5, 9, 13, 19, 21, 27)
    present = [j not in lost for j in range(k + m)]
    data = np.zeros((nobj, k * S), dtype=np.uint8)
    data[:, :L] = O.splitmix64_bytes(0x6761726167650005, nobj * L).reshape(nobj, L)
    data = data.reshape(nobj, k, S)
    co = O.COracle()
    full_np = np.concatenate([data, co.encode_batch(k, m, data, co.AVX2 if co.has_avx2() else co.SCALAR, threads=4)], axis=1)
    lb = handle = None
    if dry:
        so = os.path.join(ROOT, "tests", "c", "libgec_loopback.so")
        if not os.path.exists(so):
            return {"skipped": "dry run on one device and tests/c/libgec_loopback.so is not built (RCCL refuses two ranks on one device)"}
        lb = ctypes.CDLL(so)
        lb.lb_create.restype = ctypes.c_void_p
        lb.lb_create.argtypes = [ctypes.c_int]
        lb.lb_destroy.argtypes = [ctypes.c_void_p]
        lb.lb_rank_ctx.restype = ctypes.c_void_p
        lb.lb_rank_ctx.argtypes = [ctypes.c_void_p, ctypes.c_int]
        lb.lb_all_gather_ptr.restype = ctypes.c_void_p
        lb.lb_all_to_all_ptr.restype = ctypes.c_void_p
        handle = lb.lb_create(world)
    uid = None if dry else g.Group.unique_id()
    oks, errs, times = [None] * world, [], [None] * world
    bar = threading.Barrier(world)
    slot_ptrs = [0] * world   # every rank's slot buffer, for the peer-pointer form (one process: plain device pointers)

    def rank_main(r: int):
        try:
            d = 0 if dry else r
            torch.cuda.set_device(d)
            dev = torch.device("cuda", d)
            with torch.cuda.stream(torch.cuda.Stream(dev)):
                rs = g.ReedSolomon(k, m, device=d)
                if dry:
                    grp = g.Group(rs, r, world, transport=(lb.lb_all_gather_ptr(), lb.lb_all_to_all_ptr(), lb.lb_rank_ctx(handle, r)))
                else:
                    grp = g.Group(rs, r, world, uid)
                full = torch.from_numpy(full_np).to(dev)
                broken = full.clone()
                broken[:, list(lost)] = 0xEE
                mine = scatter_stripes(broken, layout, r)
                got = gather_stripes(grp.allgather_decode(mine, present), layout)
                torch.cuda.current_stream().synchronize()
                ok = bool(torch.equal(got, full))
                reb = grp.alltoall_decode(mine, present)
                torch.cuda.current_stream().synchronize()
                ok = ok and all(bool(torch.equal(reb[i], full[:, j])) for i, j in enumerate(sorted(lost)))
                # the peer-pointer form: the decode launch reads the other ranks' slot buffers in place (peer access over xGMI)
                slot_ptrs[r] = mine.data_ptr()
                bar.wait()
                reb = grp.peer_decode(mine, list(slot_ptrs), present)
                torch.cuda.current_stream().synchronize()
                ok = ok and all(bool(torch.equal(reb[i], full[:, j])) for i, j in enumerate(sorted(lost)))
                bar.wait()
                t0 = time.perf_counter()
                for _ in range(5):
                    grp.allgather_decode(mine, present)
                torch.cuda.current_stream().synchronize()
                times[r] = (time.perf_counter() - t0) / 5
                oks[r] = ok
                grp.close()
        except BaseException as e:  # noqa: BLE001
            errs.append(f"rank {r}: {type(e).__name__}: {e}")
            bar.abort()

    ts = [threading.Thread(target=rank_main, args=(r,), daemon=True) for r in range(world)]
    [x.start() for x in ts]
    [x.join(timeout=120) for x in ts]
    hung = any(x.is_alive() for x in ts)
    if lb is not None and not hung:
        lb.lb_destroy(handle)
    if errs or hung:
        return {"error": ("a rank hung; " if hung else "") + "; ".join(errs)[:400]}
    return {"rccl_ranks": None if dry else world, "ranks": world,
            "transport": "loopback test transport (DRY RUN on one device)" if dry else f"RCCL: gec_group_create over {world} devices from one process",
            "bit_exact": all(oks), "bit_exact_objects": nobj, "bit_exact_against": "stripes encoded by the CPU oracle; all three exchanges (all-gather, all-to-all, peer pointers), every rank",
            "allgather_decode_ms": round(max(times) * 1e3, 3),
            "config": {"workload": f"BASELINE config 5 check: RS(20,8), {nobj} x 4 MiB objects striped over {world} ranks, 8 erasures", "shard_len": S}}


def child_group_env(parent_env, rank: int, world: int, port: int) -> dict:
    """The environment of a rank's child that forms a process group of its own.  torch.distributed.run's agent variables must NOT
    travel: with TORCHELASTIC_USE_AGENT_STORE set, env:// rendezvous expects the AGENT to host the store at MASTER_PORT and rank 0
    never starts one -- on a port of our own every child would wait for a server that nobody runs."""
    env = {k: v for k, v in parent_env.items() if not k.startswith("TORCHELASTIC_") and k not in ("GROUP_RANK", "ROLE_RANK", "ROLE_WORLD_SIZE", "GROUP_WORLD_SIZE")}
    env.update(RANK=str(rank), LOCAL_RANK=parent_env.get("LOCAL_RANK", "0"), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=parent_env.get("LOCAL_WORLD_SIZE", str(world)),
               MASTER_ADDR=parent_env.get("MASTER_ADDR", "127.0.0.1"), MASTER_PORT=str(port), GARAGE_BENCH_CHILD="1")
    return env


def striped_decode_in_children(args, R) -> dict:
    """`bench.py --op striped-decode` as a child process of every rank, with the ranks' own RANK / LOCAL_RANK / WORLD_SIZE and a
    process group of its own (MASTER_PORT + 23); rank 0's child prints the object, which is returned on rank 0 (other ranks: {}).
    Whatever happens to a child -- a watchdog exit, a non-zero return code, a signal, no output -- becomes an `error` entry."""
    port = int(os.environ.get("MASTER_PORT", "29500")) + 23 if R.distributed else free_port()
    env = child_group_env(os.environ, R.rank, R.world, port)
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(R.world), "--op", "striped-decode", "--steps", str(args.steps), "--warmup", str(args.warmup),
           "--striped-objects", str(args.striped_objects), "--striped-timeout", str(args.striped_timeout), "--collective", args.collective]
    if args.striped_steps:
        cmd += ["--striped-steps", str(args.striped_steps)]
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=args.striped_timeout + 60)
    except subprocess.TimeoutExpired:
        return {"error": f"the striped-decode child did not end within {args.striped_timeout + 60:.0f} s"}
    except OSError as e:
        return {"error": f"could not start the striped-decode child: {e}"[:300]}
    if R.rank != 0:
        return {}
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not lines:
        return {"error": f"striped-decode child: rc {r.returncode}, {len(lines)} JSON lines; stderr tail: {r.stderr[-300:]}"}
    try:
        res = json.loads(lines[-1])
    except ValueError as e:
        return {"error": f"striped-decode child printed no JSON: {e}"[:200]}
    res["ran_in"] = "a child process per rank (own process group): a failure in here cannot cost the encode line"
    return res


def striped_valid_stripes(torch, rs, dev, nobj: int, k: int, m: int, S: int, L: int):
    """(nobj, k+m, S) uint8 on `dev`: nobj complete RS(k,m) stripes of L-byte objects.  Payload word i (64-bit) is a splitmix-style
    mix of i in wrapping int64 arithmetic -- no generator state, so every rank of a job builds the same bytes whatever its device;
    parity by the encode kernel."""
    full = torch.zeros((nobj, k + m, S), dtype=torch.uint8, device=dev)
    wpo = L // 8                              # 64-bit words per object
    pay = full[:, :k].reshape(nobj, k * S)    # a view: the k data shards of an object are contiguous
    for o0 in range(0, nobj, 32):
        o1 = min(nobj, o0 + 32)
        x = torch.arange(o0 * wpo, o1 * wpo, dtype=torch.int64, device=dev)
        x = x * -7046029254386353131 + 0x6761726167650005      # 0x9E3779B97F4A7C15 as int64
        x ^= x >> 30
        x *= -4658895280553007687                              # 0xBF58476D1CE4E5B9
        x ^= x >> 27
        x *= -7723592293110705685                              # 0x94D049BB133111EB
        x ^= x >> 31
        pay[o0:o1, :L] = x.view(torch.uint8).view(o1 - o0, L)
        del x
    rs.encode_dev(full)
    torch.cuda.synchronize()
    return full


# ------------------------------------------------------------------ BASELINE config 5
def striped_decode_bench(args, R, distrib, progress=None) -> dict:
    """BASELINE config 5 (secondary to the headline metric): RS(20,8), 4 MiB objects, shard j on
    rank j % N; 8 shards erased; the survivors' slots are exchanged over RCCL, every rank rebuilds
    its 1/N byte range of each missing shard in place on the gathered buffer, then the rebuilt
    ranges are exchanged.  First a bit-exact check on real data (every rank encodes the same seeded
    objects, keeps only its own slots, and compares what comes back with the full stripes), then
    the timing on 256 objects."""
    import torch
    import torch.distributed as dist

    import garage_amd as g
    from garage_amd.striped import StripeLayout, gather_stripes, scatter_stripes, striped_reconstruct

    progress = progress if progress is not None else {}
    own_pg = False
    if not R.distributed:  # world 1 still goes through RCCL so the code path is the same
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(free_port()))
        distrib.quiet_rccl()  # keep RCCL's banner and warnings off stdout
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=R.device)
        own_pg = True
    k, m, L, nobj = 20, 8, 4 << 20, args.striped_objects
    S = g.shard_len(k, L)
    layout = StripeLayout(k, m, R.world)
    rs = g.ReedSolomon(k, m, device=R.local_rank)
    lost = (0, 1, 5, 9, 13, 19, 21, 27)
    present = [j not in lost for j in range(k + m)]
    dry = os.environ.get("GARAGE_DRYRUN_ONE_GPU") == "1" and R.distributed
    # fault injection for the rehearsal tests (tests/test_world8_rehearsal.py): one rank never reaches the collective
    hang_rank = os.environ.get("GARAGE_BENCH_STRIPED_HANG_RANK")
    if hang_rank is not None and int(hang_rank) == R.rank:
        progress["stage"] = f"rank {R.rank} sleeps (GARAGE_BENCH_STRIPED_HANG_RANK)"
        time.sleep(3600)
    use_cabi = args.collective == "cabi" and dist.get_backend() == "nccl"
    keep_alive = None
    if dry and args.collective == "cabi":
        # DRY RUN (every rank on device 0, where RCCL refuses a second rank): the PRODUCT's group code (range split, packs,
        # byte-range rebuild, second exchange) over a caller transport that stages through host memory and gloo
        grp, keep_alive = dry_gloo_group(g, rs, R)
    else:
        grp = g.Group.from_torch_distributed(rs) if use_cabi else None
    progress["stage"] = "group created"

    from garage_amd.striped import striped_reconstruct_alltoall

    lost_sorted = sorted(lost)

    # the peer-pointer form: every rank opens every other rank's slot buffer (HIP IPC handles carried by the job's own process
    # group) and its decode launch reads its byte range of the survivors straight out of them
    from garage_amd.group import ipc_close, ipc_export, ipc_open

    def open_peers(t):
        """-> (pointer per rank, pointers to close).  Control flow stays collective: a rank that cannot export its buffer or map a
        peer's (first contact with N > 1 hardware) says so in the exchange, and then EVERY rank gives the peer form up -- one rank
        raising alone would leave the others inside the decode's barriers until the watchdog."""
        if grp is None:
            return None, []
        if R.world == 1:
            return [t.data_ptr()], []
        torch.cuda.synchronize()
        err = None
        try:
            if os.environ.get("GARAGE_BENCH_PEER_FAIL_RANK") == str(R.rank):  # (fault injection: tests/test_world8_rehearsal.py)
                raise RuntimeError("injected: this rank cannot export its buffer")
            mine_h = ipc_export(t)
        except Exception as e:  # noqa: BLE001
            mine_h, err = None, f"export: {type(e).__name__}: {e}"
        handles = [None] * R.world
        dist.all_gather_object(handles, mine_h)
        ptrs, opened = [], []
        for q in range(R.world):
            if q == R.rank:
                ptrs.append(t.data_ptr())
            elif err is None and handles[q] is not None:
                try:
                    p_ = ipc_open(handles[q], R.device.index or 0)
                    ptrs.append(p_)
                    opened.append(p_)
                except Exception as e:  # noqa: BLE001
                    err = f"open rank {q}'s buffer: {type(e).__name__}: {e}"
            elif err is None:
                err = f"rank {q} could not export its buffer"
        bad = distrib.sum_over_ranks(R, int(err is not None))
        if bad:
            close_peers(opened)
            raise RuntimeError(f"{bad} of {R.world} ranks could not map their peers' buffers ({err or 'this rank could'})"[:280])
        return ptrs, opened

    def close_peers(opened):
        torch.cuda.synchronize()
        distrib.barrier(R)        # nobody unmaps a buffer a peer's kernel may still be reading
        for p_ in opened:
            ipc_close(p_)

    def run(local, out=None):
        if grp is not None:
            return grp.allgather_decode(local, present, out=out)
        return striped_reconstruct(rs, local, present, layout)

    def run_a2a(local, out=None):  # the all-to-all exchange: only the rebuilt shards come back, (nmiss, nobj, S)
        if grp is not None:
            return grp.alltoall_decode(local, present, out=out)
        return striped_reconstruct_alltoall(rs, local, present, layout)

    # -- bit-exact check: same seeded objects on every rank; the stripes they must decode to are encoded by the CPU
    #    ORACLE on the host (8 x 4 MiB: milliseconds), never by the kernel under test
    import numpy as np

    from oracle import rs_oracle as O

    ncheck = 8
    payload = O.splitmix64_bytes(0x6761726167650005, ncheck * L).reshape(ncheck, L)
    data = np.zeros((ncheck, k * S), dtype=np.uint8)
    data[:, :L] = payload
    data = data.reshape(ncheck, k, S)
    co = O.COracle()
    parity = co.encode_batch(k, m, data, co.AVX2 if co.has_avx2() else co.SCALAR, threads=4)
    full = torch.from_numpy(np.concatenate([data, parity], axis=1)).to(R.device)
    broken = full.clone()
    broken[:, list(lost)] = 0xEE
    mine = scatter_stripes(broken, layout, R.rank)
    got = gather_stripes(run(mine), layout)
    torch.cuda.synchronize()
    ok = bool(torch.equal(got, full))
    progress["stage"] = "all-gather check done"
    reb = run_a2a(mine)
    torch.cuda.synchronize()
    ok_a2a = all(bool(torch.equal(reb[i], full[:, j])) for i, j in enumerate(lost_sorted))
    ok_all = distrib.sum_over_ranks(R, int(ok)) == R.world
    ok_a2a_all = distrib.sum_over_ranks(R, int(ok_a2a)) == R.world
    ok_peer_all, peer_err = None, None
    if grp is not None:
        try:
            ptrs, opened = open_peers(mine)
            reb = grp.peer_decode(mine, ptrs, present)
            torch.cuda.synchronize()
            ok_peer = all(bool(torch.equal(reb[i], full[:, j])) for i, j in enumerate(lost_sorted))
            close_peers(opened)
            ok_peer_all = distrib.sum_over_ranks(R, int(ok_peer)) == R.world
        except Exception as e:  # noqa: BLE001 -- the third exchange must not cost the other two their figures
            peer_err = f"{type(e).__name__}: {e}"[:300]
    del full, broken, got, reb, mine
    progress["stage"] = "bit-exact checks done"

    # -- timing, on VALID stripes, and every object of the timed batch checked on every rank after each exchange's loop.
    #    The payload is integer arithmetic on the byte index (no RNG: identical on every device by construction); the parity is
    #    the encode kernel's (gec_encode_batch_dev, itself checked against the C oracle on a strided sample right here); the
    #    erased shards are overwritten with 0xEE and scattered: `local` is what this rank would hold of 256 degraded objects.
    full = striped_valid_stripes(torch, rs, R.device, nobj, k, m, S, L)
    sample = sorted(set(np.linspace(0, nobj - 1, min(nobj, 4)).astype(int).tolist()))
    sample_t = torch.as_tensor(sample, device=R.device)
    host_full = full[sample_t].cpu().numpy()
    want_par = co.encode_batch(k, m, np.ascontiguousarray(host_full[:, :k]), co.AVX2 if co.has_avx2() else co.SCALAR, threads=4)
    timed_batch_ok = bool(np.array_equal(host_full[:, k:], want_par))     # the batch the loops run on IS a batch of RS(20,8) stripes
    lost_t = torch.as_tensor(lost_sorted, device=R.device)
    local = scatter_stripes(full.index_fill(1, lost_t, 0xEE), layout, R.rank)
    steps = args.striped_steps or (1 if dry else max(5, min(50, args.steps // 20)))
    nwarm = 0 if dry else 3
    # fault injection (tests/test_world8_rehearsal.py): one byte of one rank's slot buffer flipped between the warm-up and the timed
    # loop of the named exchange ("all": every exchange) -- the line must then say bit_exact false for it
    flip_rank = os.environ.get("GARAGE_BENCH_STRIPED_FLIP_RANK")
    flip_what = os.environ.get("GARAGE_BENCH_STRIPED_FLIP_EXCHANGE", "all")
    mine_present = [j for j in layout.shards_of(R.rank) if present[j]]

    def flip(name, undo=False):
        if flip_rank is None or int(flip_rank) != R.rank or flip_what not in ("all", name):
            return
        if not mine_present:
            raise RuntimeError(f"rank {R.rank} holds no surviving shard to corrupt")
        torch.cuda.synchronize()
        local[nobj // 2, layout.slot(mine_present[0]), S // 2 + 5] ^= 0x40
        torch.cuda.synchronize()

    def timed(name, fn, out):
        for _ in range(nwarm):
            fn(local, out)
        flip(name)
        torch.cuda.synchronize()
        distrib.barrier(R)
        t0 = time.perf_counter()
        for _ in range(steps):
            fn(local, out)
        torch.cuda.synchronize()
        distrib.barrier(R)
        dt_ = distrib.max_over_ranks(R, time.perf_counter() - t0)
        flip(name, undo=True)
        return dt_

    def check_rebuilt(reb_of):
        """What the LAST timed call left behind against the stripes the batch was cut from.  reb_of(i, j) -> the (nobj, S) view of
        rebuilt shard j (i-th missing).  Returns {objects whose 8 rebuilt shards all match, on the rank where fewest do;
        gec_verify_batch_dev over the completed stripes; the C oracle's own reconstruct of a strided sample}."""
        bad = torch.zeros(nobj, dtype=torch.bool, device=R.device)
        done = full.clone()
        for i, j in enumerate(lost_sorted):
            r_ = reb_of(i, j)
            bad |= (r_ != full[:, j]).any(dim=1)
            done[:, j] = r_
        n_ok = nobj - int(bad.sum().item())
        verified = bool(rs.verify_dev(done).all())
        # the oracle rebuilds the sample from the survivors alone: no kernel of this project made `want`
        broken = host_full.copy()
        broken[:, lost_sorted] = 0
        want = co.reconstruct_batch(k, m, broken, present, threads=4)
        got_s = done[sample_t][:, lost_t].cpu().numpy()
        oracle_ok = bool(np.array_equal(got_s, want[:, lost_sorted]))
        del done
        n_ok_all = -distrib.max_over_ranks(R, -n_ok)            # the worst rank's count
        all_ok = distrib.sum_over_ranks(R, int(n_ok == nobj and verified and oracle_ok and timed_batch_ok)) == R.world
        return {"bit_exact": all_ok, "bit_exact_objects": int(n_ok_all), "of": nobj,
                "verify_batch_dev": verified, "oracle_sample_objects": len(sample) if oracle_ok else 0}

    CHECKED = ("after the timed loop, on every rank: all rebuilt shards of all objects of the TIMED batch against the stripes it was cut "
               "from, gec_verify_batch_dev over the completed stripes, and the C oracle's reconstruct of a strided sample from the survivors alone")
    out = torch.empty((R.world, nobj, layout.slots, S), dtype=torch.uint8, device=R.device) if grp is not None else None
    if grp is None:
        box = {}
        dt = timed("allgather", lambda l_, o_: box.__setitem__("out", run(l_)), None)
        out = box.pop("out")
    else:
        dt = timed("allgather", run, out)
    ag_bytes = grp.bytes_exchanged() if grp is not None else nobj * layout.slots * S * (R.world - 1)
    ag_check = check_rebuilt(lambda i, j: out[layout.owner(j), :, layout.slot(j)])
    if ag_check["bit_exact"]:   # the all-gather form also returns the survivors: they must be the ones that were sent
        surv_ok = all(bool(torch.equal(out[layout.owner(j), :, layout.slot(j)], full[:, j])) for j in range(k + m) if present[j])
        ag_check["bit_exact"] = distrib.sum_over_ranks(R, int(surv_ok)) == R.world
    del out
    progress["stage"] = "all-gather timed and checked"
    out2 = torch.empty((len(lost), nobj, S), dtype=torch.uint8, device=R.device)
    if grp is None:
        box = {}
        dt2 = timed("alltoall", lambda l_, o_: box.__setitem__("out", run_a2a(l_)), None)
        out2 = box.pop("out")
    else:
        dt2 = timed("alltoall", run_a2a, out2)
    a2a_bytes = grp.bytes_exchanged() if grp is not None else None
    a2a_check = check_rebuilt(lambda i, j: out2[i])
    progress["stage"] = "all-to-all timed and checked"
    peer = {"skipped": "needs the C ABI group (--collective cabi)"} if grp is None else {"error": peer_err} if peer_err else None
    if peer is None:
        try:
            out2.fill_(0)
            ptrs, opened = open_peers(local)
            dt3 = timed("peer", lambda l_, o_: grp.peer_decode(l_, ptrs, present, out=o_), out2)
            peer_bytes = grp.bytes_exchanged()
            peer_check = check_rebuilt(lambda i, j: out2[i])
            close_peers(opened)
            peer = {"ms_per_step": round(dt3 / steps * 1e3, 3), "GiBps": round(nobj * L * steps / dt3 / 2**30, 2),
                    "bytes_received_per_rank": peer_bytes, "bytes_read_from_peers_memory_per_rank": peer_bytes - (R.world - 1) * len(lost) * nobj * (-(-(S // 16) // R.world)) * 16 if R.world > 1 else 0,
                    **peer_check, "small_batch_bit_exact": ok_peer_all,
                    "what": "gec_group_peer_decode: ONE decode launch per rank reads its byte range of the k valid shards out of the other "
                            "ranks' slot buffers (HIP IPC mappings, xGMI), no pack / unpack / staging; the transport carries two 16-byte "
                            "barriers and the rebuilt ranges"}
        except Exception as e:  # noqa: BLE001
            peer = {"error": f"{type(e).__name__}: {e}"[:300]}
    del out2, full, local
    timed_ok = ag_check["bit_exact"] and a2a_check["bit_exact"] and (peer.get("bit_exact", True) is not False)
    res = {
        "metric": "RS(20,8) striped-object decode payload throughput, 4 MiB objects (all-gather + range reconstruct + range exchange)",
        "value": round(nobj * L * steps / dt / 2**30, 2), "unit": "GiB/s", "n_gpus": R.world, "steps": steps,
        "ms_per_step": round(dt / steps * 1e3, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "u8", "data": "synthetic",
        "bit_exact": ok_all and ag_check["bit_exact"], "bit_exact_objects": ag_check["bit_exact_objects"], "timed_batch_objects": nobj,
        "bit_exact_against": f"(1) before the timing, {ncheck} objects whose stripes the CPU oracle (oracle/rs_oracle.c) encoded on the host; (2) " + CHECKED,
        "timed_batch": "valid RS(20,8) stripes: payload = integer arithmetic on the byte index (the same on every rank), parity by "
                       "gec_encode_batch_dev, checked against the C oracle on a strided sample: " + ("ok" if timed_batch_ok else "MISMATCH"),
        "rccl_ranks": (grp.nranks if grp is not None else dist.get_world_size()),
        "collective_backend": ("DRY RUN: gloo through a host-staging gec_allgather_fn / gec_alltoall_fn (every rank on device 0; "
                               "timings meaningless)" if dry and grp is not None
                               else "rccl (gec_group_create: ncclCommInitRank from libgarage_ec)" if grp is not None
                               else f"torch.distributed {dist.get_backend()}"),
        "config": {"workload": f"BASELINE config 5: RS(20,8), {nobj} x 4 MiB objects striped over the ranks, 8 erasures",
                   "k": k, "m": m, "shard_len": S, "slots_per_rank": layout.slots,
                   "allgather_bytes_per_rank": nobj * layout.slots * S, "parallelism": f"stripe x{R.world}",
                   "collective": ("gec_group_allgather_decode (C ABI, caller transport: gloo, DRY RUN)" if dry and grp is not None
                                  else "gec_group_allgather_decode (C ABI, RCCL ncclAllGather)") if grp is not None
                                 else f"torch.distributed all_gather_into_tensor ({dist.get_backend()}) + gec_reconstruct_scattered_dev"},
        # the two exchanges side by side (all-gather = the project brief's, and the `value` above)
        "exchange": {
            "allgather": {"ms_per_step": round(dt / steps * 1e3, 3), "GiBps": round(nobj * L * steps / dt / 2**30, 2),
                          "bytes_received_per_rank": ag_bytes, **ag_check, "small_batch_bit_exact": ok_all},
            "alltoall": {"ms_per_step": round(dt2 / steps * 1e3, 3), "GiBps": round(nobj * L * steps / dt2 / 2**30, 2),
                         "bytes_received_per_rank": a2a_bytes, **a2a_check, "small_batch_bit_exact": ok_a2a_all,
                         "what": "gec_group_alltoall_decode: each rank receives only its byte range of the k valid shards "
                                 "(grouped ncclSend/ncclRecv); returns the rebuilt shards only"},
            "peer": peer,
        },
    }
    if grp is not None:
        grp.close()
    del keep_alive   # the dry run's transport callbacks had to outlive the group
    if own_pg:
        dist.destroy_process_group()
    if not (ok_all and ok_a2a_all) or ok_peer_all is False or not timed_ok:
        res["error"] = "striped decode result differs from the oracle's stripes" + ("" if timed_ok else " (the timed batch)")
    return res


def dry_gloo_group(g, rs, R):
    """GARAGE_DRYRUN_ONE_GPU=1 with N > 1 processes: a gec_group over gec_group_create_with_transport2 whose all-gather and
    all-to-all stage the device buffers through host memory and torch.distributed's gloo group.  Plumbing only -- it lets the
    N-rank control flow of the product's group code run end to end on one GPU.  Returns (group, objects to keep alive)."""
    import ctypes

    import numpy as np
    import torch
    import torch.distributed as dist

    from garage_amd import _lib

    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    hip.hipStreamSynchronize.argtypes = [ctypes.c_void_p]
    D2H, H2D = 2, 1
    world = R.world

    def exchange(send, recv, nbytes, stream, a2a):
        if not nbytes:
            return 0
        if hip.hipStreamSynchronize(stream):
            return 1
        mine = np.empty(nbytes * (world if a2a else 1), dtype=np.uint8)
        if hip.hipMemcpy(mine.ctypes.data, send, mine.size, D2H):
            return 1
        got = np.empty(nbytes * world, dtype=np.uint8)
        if a2a:
            dist.all_to_all_single(torch.from_numpy(got), torch.from_numpy(mine))
        else:
            dist.all_gather(list(torch.from_numpy(got).chunk(world)), torch.from_numpy(mine))
        return 1 if hip.hipMemcpy(recv, got.ctypes.data, got.size, H2D) else 0

    @_lib.ALLGATHER_FN
    def all_gather(_ctx, send, recv, nbytes, stream):
        return exchange(send, recv, nbytes, stream, False)

    @_lib.ALLGATHER_FN
    def all_to_all(_ctx, send, recv, nbytes, stream):
        return exchange(send, recv, nbytes, stream, True)

    return g.Group(rs, R.rank, world, transport=(all_gather, all_to_all, None)), (all_gather, all_to_all, hip)


# ------------------------------------------------- every GPU fed from host memory at once
HOST_TRAFFIC_MODEL = {
    # host-memory bytes moved per payload byte, RS(10,4) encode + all 14 shard checksums through gec_encode_hash_batch
    "pinned": 1.0 * (10 * 104896 / 1048576) + 0.4 * (10 * 104896 / 1048576),   # the kernel reads k*S, writes m*S in place over the link
    "pageable": (1.0 + 1.0 + 1.0 + 0.4 + 0.4 + 0.4) * (10 * 104896 / 1048576),  # memcpy in (r+w), DMA read, DMA write, memcpy out (r+w)
}


def host_fed_section(rs, nb: int, reps: int, barrier, seed: int) -> dict:
    """One GPU's share of the host-fed measurement: `nb` 1 MiB blocks pushed through gec_encode_hash_batch (encode +
    the checksum of every shard, what rpc_put_block needs) `reps` times, first from pinned caller memory (read in
    place over the link), then from ordinary pageable memory (staged), between barriers so that every GPU of the node
    pulls on the host's memory system at the same time.  Parity and checksums of a strided sample are compared with
    the CPU oracle / hashlib afterwards."""
    import ctypes

    import numpy as np

    import garage_amd as g
    from garage_amd._lib import check, lib
    from garage_amd.codec import host_alloc, host_free
    from oracle import rs_oracle as O

    S = g.shard_len(K, BLOCK_LEN)
    n = K + M
    rng = np.random.default_rng(seed)
    res = {}
    for kind in ("pinned", "pageable"):
        alloc = (lambda nbytes: host_alloc(nbytes)) if kind == "pinned" else (lambda nbytes: np.empty(nbytes, dtype=np.uint8))
        blocks = [alloc(K * S) for _ in range(nb)]
        outs = [alloc(M * S) for _ in range(nb)]
        for b in blocks:
            b[:BLOCK_LEN] = rng.integers(0, 256, BLOCK_LEN, dtype=np.uint8)
            b[BLOCK_LEN:] = 0
        lens = (ctypes.c_size_t * nb)(*[BLOCK_LEN] * nb)
        ptrs = (ctypes.c_void_p * nb)(*[b.ctypes.data for b in blocks])
        optrs = (ctypes.c_void_p * nb)(*[o.ctypes.data for o in outs])
        sums = np.zeros((nb, n, 32), dtype=np.uint8)
        sp = sums.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))
        run = lambda: check(lib.gec_encode_hash_batch(rs._h, nb, ptrs, lens, S, optrs, sp), "gec_encode_hash_batch")  # noqa: E731
        run()
        run()
        barrier()
        t0 = time.time()   # wall clock: comparable across the ranks of one node
        for _ in range(reps):
            run()
        t1 = time.time()
        barrier()
        # the oracle's word on a strided sample of what the last call left in the caller's buffers
        co = O.COracle()
        idx = sorted(set(np.linspace(0, nb - 1, min(8, nb)).astype(int).tolist()))
        data = np.stack([np.asarray(blocks[i]).reshape(K, S) for i in idx])
        want = co.encode_batch(K, M, data, co.AVX2 if co.has_avx2() else co.SCALAR, threads=2)
        got = np.stack([np.asarray(outs[i]).reshape(M, S) for i in idx])
        exact = bool(np.array_equal(got, want))
        for q, i in enumerate(idx[:2]):
            for j in (0, K - 1, K, n - 1):
                sh = data[q, j] if j < K else want[q, j - K]
                exact = exact and sums[i, j].tobytes() == g.shardsum(sh.tobytes())
        res[kind] = {"t0": t0, "t1": t1, "GiBps": nb * reps * BLOCK_LEN / (t1 - t0) / 2**30, "bit_exact": exact, "checked": len(idx)}
        if kind == "pinned":
            for a in blocks + outs:
                host_free(a)
        del blocks, outs
    return res


def multi_manager_rates(ndev: int, dry: bool, callers_per_device: int = 48, puts: int = 20) -> dict:
    """tools/multi_bench in a subprocess: libgarage_block's multi-device manager under native load, per device and summed."""
    import subprocess

    root = os.path.dirname(os.path.abspath(__file__))
    exe = os.path.join(root, "tools", "multi_bench")
    try:
        if not os.path.exists(exe):
            r = subprocess.run(["make", "-C", os.path.join(root, "tools"), "multi_bench"], capture_output=True, text=True, timeout=120)
            if r.returncode != 0:
                return {"error": "tools/multi_bench is not built: " + (r.stderr or r.stdout)[-200:]}
        r = subprocess.run([exe, str(ndev), str(callers_per_device), str(puts), "1" if dry else "0"], capture_output=True, text=True, timeout=300)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not line:
            return {"error": (r.stderr or r.stdout)[-300:]}
        return json.loads(line[-1])
    except Exception as e:  # noqa: BLE001 -- a secondary figure must never cost the headline line
        return {"error": f"{type(e).__name__}: {e}"[:300]}


def host_fed_object(per_gpu: list, nb: int, reps: int) -> dict:
    """per_gpu: host_fed_section results, one per GPU -> the `host_fed` object of the line."""
    out = {"what": "every GPU fed from host memory at once: gec_encode_hash_batch (RS(10,4) encode + 14 shard checksums per block), "
                   "1 MiB blocks, payload GiB/s -- PCIe-inclusive, never the `value`",
           "blocks_per_gpu_and_call": nb, "calls": reps, "n_gpus": len(per_gpu)}
    for kind in ("pinned", "pageable"):
        rates = [r[kind]["GiBps"] for r in per_gpu]
        span = max(r[kind]["t1"] for r in per_gpu) - min(r[kind]["t0"] for r in per_gpu)
        agg = len(per_gpu) * nb * reps * BLOCK_LEN / span / 2**30
        out[kind] = {"per_gpu_GiBps": [round(x, 2) for x in rates], "aggregate_GiBps": round(agg, 2),
                     "host_memory_traffic_GBps_model": round(agg * 2**30 / 1e9 * HOST_TRAFFIC_MODEL[kind], 1),
                     "host_bytes_per_payload_byte_model": round(HOST_TRAFFIC_MODEL[kind], 2),
                     "bit_exact_vs_oracle": all(r[kind]["bit_exact"] for r in per_gpu),
                     "blocks_checked": sum(r[kind]["checked"] for r in per_gpu)}
    return out


# ------------------------------------------------- the path above the kernel (N=1, rank 0)
def host_path_objects() -> dict:
    """PCIe-inclusive rate of the host-pointer C ABI and the C++ BlockManager mirror's put/get rate on
    in-memory nodes -- reported next to the device-resident `value`, never as it."""
    out = {}
    try:
        from tools.host_path_bench import block_manager_rates, pcie_inclusive_rates

        out["pcie_inclusive"] = pcie_inclusive_rates()
        out["block_manager"] = block_manager_rates()
    except Exception as e:  # noqa: BLE001 -- secondary numbers must never cost the headline line
        out.setdefault("pcie_inclusive", {"error": f"{type(e).__name__}: {e}"[:300]})
    try:
        from tools.host_path_bench import maintenance_rates

        out["block_manager"]["maintenance"] = maintenance_rates(256)   # scrub / resync of the mirror
    except Exception as e:  # noqa: BLE001
        out.setdefault("block_manager", {})["maintenance"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    return out


# --------------------------------------------------------------------------- launching
def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(args) -> int:
    """`python bench.py --gpus N` with no launcher: run ourselves under torch.distributed.run, one
    rank per GPU, on a free local port.  stdout/stderr are inherited, so rank 0's JSON line is ours."""
    env = dict(os.environ, GARAGE_BENCH_SELF_LAUNCHED="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def launch_check(args) -> None:
    """--launch-check: only the launch plumbing (rank discovery, process group, one reduction), no
    GPU work -- runs on a CPU-only box with gloo.  tests/test_bench_launch.py."""
    if args.mode == "threads":
        seen = []
        th = [threading.Thread(target=lambda t=t: seen.append(t)) for t in range(args.gpus)]
        [x.start() for x in th]
        [x.join() for x in th]
        print(json.dumps({"launch_check": True, "mode": "threads", "world": args.gpus, "ranks_seen": len(seen)}), flush=True)
        return
    from garage_amd import distrib

    R = distrib.init_from_env()
    if R.world != args.gpus:
        sys.exit(f"WORLD_SIZE={R.world} does not match --gpus {args.gpus}")
    c = distrib.count_ranks(R)
    per = distrib.gather_ints(R, R.rank)
    if R.rank == 0:
        print(json.dumps({"launch_check": True, "mode": "procs", "world": R.world, "ranks_seen": c["ranks"],
                          "backend": c["backend"], "ranks": per,
                          "self_launched": bool(os.environ.get("GARAGE_BENCH_SELF_LAUNCHED"))}), flush=True)
    distrib.shutdown(R)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--batch", type=int, default=BATCH, help="blocks per GPU (default 1024 = BASELINE config 2)")
    ap.add_argument("--mode", choices=["procs", "threads"], default="procs",
                    help="procs = one process per GPU (torch.distributed.run; self-launched when WORLD_SIZE is unset); "
                         "threads = one process, N codecs, N host threads through the C ABI")
    ap.add_argument("--variant", type=int, default=0, help="0 nibble product tables (default), 1 log/antilog baseline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-decode", action="store_true")
    ap.add_argument("--no-oracle-check", action="store_true")
    ap.add_argument("--no-host-path", action="store_true", help="skip the pcie_inclusive / block_manager objects (N=1)")
    ap.add_argument("--no-extra", action="store_true", help="N=1: skip the encode_hash / rs20_8_encode objects")
    ap.add_argument("--no-striped", action="store_true", help="N>1: skip the BASELINE config 5 striped_decode object")
    ap.add_argument("--striped", action="store_true", help="N=1: also run the striped_decode object (RCCL with one rank)")
    ap.add_argument("--striped-objects", type=int, default=256)
    ap.add_argument("--striped-steps", type=int, default=0, help="timed steps of each striped-decode exchange (0 = from --steps)")
    ap.add_argument("--striped-timeout", type=float, default=150.0,
                    help="watchdog for the striped_decode object: after this many seconds the line is printed without it")
    ap.add_argument("--precondition-ms", type=float, default=200.0,
                    help="untimed device pre-conditioning before the W warm-up steps: the same encode launches for this "
                         "long, so a short K/W does not measure MI355X's idle-to-load DVFS transient (0 disables)")
    ap.add_argument("--op", choices=["encode", "striped-decode"], default="encode",
                    help="encode = the BASELINE metric (default); striped-decode = only BASELINE config 5: RS(20,8), 4 MiB "
                         "objects striped over the ranks, all-gather + per-rank byte-range reconstruct")
    ap.add_argument("--collective", choices=["torch", "cabi"], default="cabi",
                    help="striped decode: who drives RCCL -- libgarage_ec's gec_group_* C ABI (default) or torch.distributed")
    ap.add_argument("--launch-check", action="store_true", help="exercise only the launch plumbing (no GPU needed)")
    ap.add_argument("--cpu-baseline-only", action="store_true",
                    help="print only the cpu_baseline object (what the default run executes in a fresh subprocess before it touches the GPU)")
    ap.add_argument("--cpu-threads-probe", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-backend-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-host-fed", action="store_true",
                    help="N>1: skip the host_fed object (every GPU fed from host memory through gec_encode_hash_batch at once)")
    ap.add_argument("--host-fed", action="store_true", help="N=1: also run the host_fed object")
    ap.add_argument("--host-blocks", type=int, default=256, help="host_fed: 1 MiB blocks per GPU and call")
    args = ap.parse_args()
    if args.gpus < 1:
        sys.exit("--gpus must be >= 1")
    if args.cpu_threads_probe:
        cpu_threads_probe(args.cpu_threads_probe)
        return
    if args.cpu_backend_only:
        print(json.dumps(cpu_backend_rate(((BLOCK_LEN + K - 1) // K + 63) // 64 * 64)), flush=True)
        return
    if args.cpu_baseline_only:
        cpu_baseline_only()
        return
    # the CPU baseline first, in a process of its own, before anything of HIP / torch is loaded here (N=1, rank 0)
    args.cpu_pre = None
    if (args.gpus == 1 and not args.no_cpu_baseline and not args.launch_check and args.op == "encode"
            and int(os.environ.get("RANK", "0")) == 0):
        args.cpu_pre = cpu_baseline_subprocess()

    if args.mode == "procs" and args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    if args.launch_check:
        launch_check(args)
    elif args.mode == "threads":
        run_threads(args)
    else:
        run_procs(args)


if __name__ == "__main__":
    main()
