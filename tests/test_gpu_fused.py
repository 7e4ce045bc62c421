"""Small trips on pinned caller memory -- gec_encode_hash_batch and gec_decode_verify_batch for batches small enough that
launches, not bytes, are what they cost -- for BOTH shard checksum kinds:
  kind 3 (MLH64, the default): the link kernel leaves the leaf sums itself (gf_apply_ptrs SUM form) + one root kernel;
  kind 2 (BLAKE2b tree): the one-launch kernel of round 4 (garage_amd/csrc/fused.hpp).
Parity and rebuilt shards against the CPU oracle, every checksum against the restatement of its kind, the same calls again
with gec_set_kernel_variant(4) (the streaming paths) as an A/B, and kernel counts from rocprofv3 where it is installed."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

import garage_amd as g
from garage_amd import _lib
from garage_amd.codec import host_alloc, host_free

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
u8 = ctypes.POINTER(ctypes.c_uint8)


def _put(rs, k, m, S, lens, seed):
    """gec_encode_hash_batch on pinned memory -> (padded data [nb,k,S], parity [nb,m,S], sums [nb,n,32])."""
    nb, n = len(lens), k + m
    rng = np.random.default_rng(seed)
    padded = np.zeros((nb, k * S), dtype=np.uint8)
    arena, par = host_alloc(nb * k * S), host_alloc(nb * m * S)
    arena[:] = 0x77          # what lies behind a block's end must not leak into parity or checksums
    par[:] = 0xEE
    for b in range(nb):
        padded[b, :lens[b]] = rng.integers(0, 256, lens[b], dtype=np.uint8)
        arena[b * k * S: b * k * S + lens[b]] = padded[b, :lens[b]]
    ptrs = (ctypes.c_void_p * nb)(*[arena.ctypes.data + b * k * S for b in range(nb)])
    optrs = (ctypes.c_void_p * nb)(*[par.ctypes.data + b * m * S for b in range(nb)])
    sums = np.zeros((nb, n, 32), dtype=np.uint8)
    _lib.check(_lib.lib.gec_encode_hash_batch(rs._h, nb, ptrs, (ctypes.c_size_t * nb)(*lens), S, optrs, sums.ctypes.data_as(u8)),
               "gec_encode_hash_batch")
    out = par.reshape(nb, m, S).copy()
    host_free(arena)
    host_free(par)
    return padded.reshape(nb, k, S), out, sums


@pytest.mark.parametrize("k,m,L,nb", [(10, 4, 1 << 20, 1), (10, 4, 1 << 20, 3), (10, 4, 1 << 20, 48), (3, 1, 65536, 16), (10, 4, 70_000, 20),
                                      (20, 8, 4 << 20, 2), (6, 3, 4096 * 6, 5), (10, 4, 640, 9), (10, 12, 70_000, 5), (10, 4, 1 << 20, 200)],
                         ids=["one_block", "a_putobjects_three", "48_callers", "config1_rs3_1", "small_blocks", "rs20_8_4mib", "one_leaf_shards", "64_byte_shards",
                              "rs10_12_two_row_groups", "200_blocks"])
@pytest.mark.parametrize("kind", [3, 2], ids=["mlh64", "blake2b_tree"])
def test_put_in_one_launch_parity_and_checksums(coracle, k, m, L, nb, kind):
    rs = g.ReedSolomon(k, m, shardsum=kind)
    S = g.shard_len(k, L)
    rng = np.random.default_rng(nb + k)
    lens = [L if b % 4 != 1 else int(rng.integers(0, L + 1)) for b in range(nb)]
    if nb > 2:
        lens[2] = 0                       # an empty block: all-zero shards, still k + m checksums
    data, par, sums = _put(rs, k, m, S, lens, seed=nb)
    want = coracle.encode_batch(k, m, data, coracle.AVX2, threads=4)
    assert np.array_equal(par, want)
    for b in range(nb):
        for j in range(k + m):
            payload = data[b, j] if j < k else want[b, j - k]
            assert sums[b, j].tobytes() == g.shardsum(payload.tobytes(), kind), (b, j)
    # twice more on the same codec: the per-block tile counters must be back at zero after every launch
    for _ in range(2):
        _, par2, sums2 = _put(rs, k, m, S, lens, seed=nb)
        assert np.array_equal(par2, want) and np.array_equal(sums2, sums)


def _get(rs, k, m, S, full, lost, with_block_sums=False):
    """gec_decode_verify_batch on pinned shards; lost[b] = shard indices not in hand.  -> (sums, rebuilt dict)."""
    nb, n = full.shape[0], k + m
    bufs, fresh = [], {}
    sp, op = (ctypes.c_void_p * (nb * n))(), (ctypes.c_void_p * (nb * n))()
    for b in range(nb):
        for j in range(n):
            if j in lost[b]:
                if j < k:
                    fresh[(b, j)] = host_alloc(S)
                    fresh[(b, j)][:] = 0xCC
                    op[b * n + j] = fresh[(b, j)].ctypes.data
            else:
                a = host_alloc(S)
                a[:] = full[b, j]
                bufs.append(a)
                sp[b * n + j] = a.ctypes.data
    lens = (ctypes.c_size_t * nb)(*[k * S] * nb)
    ssums = np.zeros((nb, n, 32), dtype=np.uint8)
    bsums = np.zeros((nb, 32), dtype=np.uint8)
    _lib.check(_lib.lib.gec_decode_verify_batch(rs._h, nb, sp, S, lens, op, ssums.ctypes.data_as(u8),
                                                bsums.ctypes.data_as(u8) if with_block_sums else None), "gec_decode_verify_batch")
    out = {key: v.copy() for key, v in fresh.items()}
    for a in bufs + list(fresh.values()):
        host_free(a)
    return ssums, out


@pytest.mark.parametrize("k,m,L,nb", [(10, 4, 1 << 20, 1), (10, 4, 1 << 20, 14), (10, 4, 1 << 20, 20), (10, 4, 1 << 20, 24),
                                      (3, 1, 65536, 12), (20, 8, 1 << 20, 6), (10, 4, 1 << 20, 96), (10, 4, 300_000, 130), (20, 8, 4 << 20, 50)],
                         ids=["one_block", "14_blocks_many_patterns", "20_blocks_streaming", "24_blocks_in_two_pieces", "rs3_1", "rs20_8",
                              "96_blocks_in_pieces", "130_blocks_in_pieces", "rs20_8_50_blocks_in_pieces"])
@pytest.mark.parametrize("kind", [3, 2], ids=["mlh64", "blake2b_tree"])
def test_get_in_one_launch_many_erasure_patterns(coracle, k, m, L, nb, kind):
    """One launch serves a batch whose blocks lost DIFFERENT shards (a coefficient set per block): checksums of exactly
    the first k shards in hand, every missing data shard rebuilt, blocks that need no decode beside blocks that do.
    (Past GEC_FUSED_GET_MAX_LEAVES the streaming path takes over, and from 24 blocks on the trip goes in PIECES -- upload / checksums +
    decode / rebuilt shards home pipelined on three streams, ec_hip_host.cpp: the same assertions.)"""
    rs = g.ReedSolomon(k, m, shardsum=kind)
    n, S = k + m, g.shard_len(k, L)
    rng = np.random.default_rng(7 * k + nb)
    data = rng.integers(0, 256, (nb, k, S), dtype=np.uint8)
    full = np.concatenate([data, coracle.encode_batch(k, m, data, coracle.AVX2, threads=4)], axis=1)
    lost = []
    for b in range(nb):
        cnt = int(rng.integers(0, m + 1)) if b % 3 else 0          # every third block is healthy
        lost.append(tuple(sorted(rng.choice(n, size=cnt, replace=False).tolist())))
    if nb > 2:
        lost[1] = tuple(range(min(m, k)))                             # the first m data shards: the worst case
        lost[2] = tuple(range(k, n))                                  # every parity shard: nothing to rebuild
    ssums, rebuilt = _get(rs, k, m, S, full, lost)
    for b in range(nb):
        present = [j for j in range(n) if j not in lost[b]][:k]
        for j in range(n):
            if j in present:
                assert ssums[b, j].tobytes() == g.shardsum(full[b, j].tobytes(), kind), (b, j, lost[b])
            else:
                assert not ssums[b, j].any(), (b, j)
        for j in lost[b]:
            if j < k:
                assert np.array_equal(rebuilt[(b, j)], full[b, j]), (b, j, lost[b])


def test_fused_and_streaming_paths_agree_and_fused_launches_fewer_kernels(tmp_path):
    """A/B in two subprocesses (gec_set_kernel_variant(4) = the streaming route): identical bytes and checksums either way; under
    rocprofv3 the small put is ONE kernel launch and the degraded small get ONE (three to five and four-plus before)."""
    code = r'''
import sys, hashlib, json
import numpy as np
sys.path.insert(0, %r)
import garage_amd as g
from oracle import rs_oracle as O
from tests.test_gpu_fused import _put, _get
g.set_kernel_variant(0 if sys.argv[1] == "1" else 4)
k, m = 10, 4
rs = g.ReedSolomon(k, m, shardsum=2)
S = g.shard_len(k, 1 << 20)
data, par, sums = _put(rs, k, m, S, [1 << 20, 1 << 20, 777_777], seed=5)
full = np.concatenate([data, par], axis=1)
lost = [(0, 5), (), (3, 11, 12)]
ss, reb = _get(rs, k, m, S, full, lost)
h = hashlib.sha256()
for x in (par, sums, ss):
    h.update(x.tobytes())
for key in sorted(reb):
    h.update(reb[key].tobytes())
    assert np.array_equal(reb[key], full[key[0], key[1]])
print("DIGEST", h.hexdigest())
''' % ROOT
    digests = {}
    for fused in ("1", "0"):
        r = subprocess.run([sys.executable, "-c", code, fused], cwd=ROOT, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        digests[fused] = [ln for ln in r.stdout.splitlines() if ln.startswith("DIGEST")][0]
    assert digests["1"] == digests["0"]
    rocprof = "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        pytest.skip("rocprofv3 not installed")
    counts = {}
    for fused in ("1", "0"):
        out = tmp_path / f"prof{fused}"
        r = subprocess.run([rocprof, "--kernel-trace", "--stats", "-d", str(out), "-o", "t", "--output-format", "csv", "--", sys.executable, "-c", code, fused],
                           cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        names = []
        for dirpath, _, files in os.walk(out):
            for f in files:
                if f.endswith("kernel_trace.csv"):
                    import csv

                    with open(os.path.join(dirpath, f)) as fh:   # (the runtime's own fill / copy kernels are set-up, not trips)
                        names += [row["Kernel_Name"] for row in csv.DictReader(fh) if not row["Kernel_Name"].startswith("__amd_rocclr")]
        counts[fused] = names
    fused_names = [x for x in counts["1"] if "gf_ptrs_hash" in x]
    assert len(fused_names) == 2, counts["1"]                       # one put + one get
    assert len(counts["1"]) == 2, counts["1"]                       # ... and nothing else
    assert len(counts["0"]) >= len(counts["1"]) + 4, (len(counts["0"]), len(counts["1"]))


def test_a_v3_put_is_the_link_kernel_and_a_root_kernel(tmp_path):
    """Checksum kind 3 under rocprofv3: a put of three blocks on pinned memory is exactly TWO launches -- the pointer-table
    kernel in its SUM form and mlh_roots -- and no BLAKE2b leaf kernel, no mirror copy, nothing else."""
    rocprof = "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        pytest.skip("rocprofv3 not installed")
    code = r'''
import sys
import numpy as np
sys.path.insert(0, %r)
import garage_amd as g
from oracle import mlh64
from tests.test_gpu_fused import _put
k, m = 10, 4
rs = g.ReedSolomon(k, m)
assert rs.shardsum_kind == 3
S = g.shard_len(k, 1 << 20)
data, par, sums = _put(rs, k, m, S, [1 << 20, 1 << 20, 777_777], seed=5)
full = np.concatenate([data, par], axis=1)
assert all(sums[b, j].tobytes() == mlh64.shardsum3(full[b, j].tobytes()) for b in range(3) for j in range(14))
''' % ROOT
    out = tmp_path / "prof"
    r = subprocess.run([rocprof, "--kernel-trace", "--stats", "-d", str(out), "-o", "t", "--output-format", "csv", "--", sys.executable, "-c", code],
                       cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    names = []
    for dirpath, _, files in os.walk(out):
        for f in files:
            if f.endswith("kernel_trace.csv"):
                import csv

                with open(os.path.join(dirpath, f)) as fh:
                    names += [row["Kernel_Name"] for row in csv.DictReader(fh) if not row["Kernel_Name"].startswith("__amd_rocclr")]
    assert len(names) == 2 and "gf_apply_ptrs<1, 5, false, false, true>" in names[0] and "mlh_roots" in names[1], names
