"""Run by tests/test_cabi_host.py in a process of its own: a call into libgarage_ec that cannot get host memory must come back
with GEC_E_NOMEM -- not unwind across the C ABI, not std::terminate on a pool thread -- and the codec must work afterwards.
usage: oom_probe.py <margin MiB>;  prints one line: "<rc of the starved call> <rc of the same call afterwards> <1 = that one's parity and checksums equal an
undisturbed call's> <the starved call's gec_last_error()>" """
import ctypes
import os
import resource
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import garage_amd as g  # noqa: E402
from garage_amd import _lib  # noqa: E402

k, m, nb, L = 10, 4, 96, 1 << 20
rs = g.ReedSolomon(k, m, backend="cpu")
S = int(_lib.lib.gec_shard_len(k, L))
blocks = [np.random.default_rng(i).integers(0, 256, L, dtype=np.uint8) for i in range(nb)]
U8P = ctypes.POINTER(ctypes.c_uint8)


def encode(nblocks, calls_for=None):
    """buffers for a call over `nblocks` blocks; calls_for > nblocks: the argument arrays name them cyclically (the starved call:
    what it would write is never looked at, only that it comes back with a code)"""
    par = [np.zeros(m * S, dtype=np.uint8) for _ in range(nblocks)]   # parity[b]: m consecutive shards
    n = calls_for or nblocks
    sums = np.zeros(n * (k + m) * 32, dtype=np.uint8)
    bp = (ctypes.c_void_p * n)(*[blocks[i % nblocks].ctypes.data for i in range(n)])
    pp = (ctypes.c_void_p * n)(*[par[i % nblocks].ctypes.data for i in range(n)])
    lens = (ctypes.c_size_t * n)(*([L] * n))
    return bp, pp, lens, par, sums


# every thread of the codec's pool has run once (glibc aborts by itself when a thread cannot get its thread-local block)
w = encode(2)
assert _lib.lib.gec_encode_hash_batch(rs._h, 2, w[0], w[2], S, w[1], w[4].ctypes.data_as(U8P)) == 0
# the starved call names 8192 blocks: its leaf sums alone (8192 x 14 shards x 26 leaves x 8 bytes = 23 MiB) are beyond every margin
# tried (since round 6 a put trip no longer copies a padded shard per block, which is what used to run out first)
NSTARVE = 8192
a, b = encode(nb, NSTARVE), encode(nb)
with open("/proc/self/statm") as f:
    vm = int(f.read().split()[0]) * 4096
soft, hard = resource.getrlimit(resource.RLIMIT_AS)
resource.setrlimit(resource.RLIMIT_AS, (vm + (int(sys.argv[1]) << 20), hard))
rc1 = _lib.lib.gec_encode_hash_batch(rs._h, NSTARVE, a[0], a[2], S, a[1], a[4].ctypes.data_as(U8P))
err = _lib.lib.gec_last_error()
resource.setrlimit(resource.RLIMIT_AS, (soft, hard))
rc2 = _lib.lib.gec_encode_hash_batch(rs._h, nb, b[0], b[2], S, b[1], b[4].ctypes.data_as(U8P))
same = int(rc2 == 0 and all((x == y).all() for x, y in zip(w[3], b[3][:2])) and (w[4] == b[4][:2 * (k + m) * 32]).all())
print(rc1, rc2, same, err.decode() if isinstance(err, bytes) else err)
