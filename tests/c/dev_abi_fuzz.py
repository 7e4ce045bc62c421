"""The device-resident (strided) entry points of libgarage_ec under arbitrary geometry: one buffer of 64 MiB, and every call's
parameters are either rejected or describe accesses INSIDE that buffer -- S of 0 / 63 / 64 / 4096 / 65536, strides that are
zero, short of a stripe, no multiple of 16, base pointers off by 8, byte ranges that leave the shard, erasure patterns with too
few shards present, scattered offsets.  A HIP codec works on device memory (torch allocates it), a CPU codec's reconstruct forms
on host memory.  Every call must come back with a code and the device must still answer afterwards.
usage: dev_abi_fuzz.py <seed> [cpu|hip]; prints "done <calls> ok=<calls that returned GEC_OK>" """
import ctypes
import os
import random
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import garage_amd as g  # noqa: E402
from garage_amd import _lib  # noqa: E402

rng = random.Random(int(sys.argv[1]))
backend = sys.argv[2] if len(sys.argv) > 2 else "cpu"
L = _lib.lib
TOTAL = 64 << 20
if backend == "hip":
    import torch

    buf = torch.randint(0, 256, (TOTAL,), dtype=torch.uint8, device="cuda:0")
    aux = torch.zeros(1 << 20, dtype=torch.uint8, device="cuda:0")
    base, auxp = buf.data_ptr(), aux.data_ptr()
    sync = torch.cuda.synchronize
else:
    buf = np.random.default_rng(1).integers(0, 256, TOTAL, dtype=np.uint8)
    aux = np.zeros(1 << 20, dtype=np.uint8)
    base, auxp = buf.ctypes.data, aux.ctypes.data

    def sync():
        pass
U8P = ctypes.POINTER(ctypes.c_uint8)
sz = ctypes.c_size_t
calls = oks = 0
for it in range(400):
    k, m = rng.choice([(3, 1), (10, 4), (20, 8), (1, 1)])
    rs = g.ReedSolomon(k, m, backend=backend)
    n = k + m
    S = rng.choice([0, 63, 64, 128, 4096, 65536, 65600])
    nb = rng.choice([0, 1, 2, 8])
    stride = rng.choice([0, 16, max(n * S - 16, 0), n * S, n * S + 16, n * S + 8, n * S + 4096])
    off = rng.choice([0, 0, 0, 8, 16, 4096])
    if off + (max(nb, 1) - 1) * stride + (n + 1) * max(S, 1) + 65600 > TOTAL:
        continue
    present = (ctypes.c_uint8 * n)(*[0 if rng.random() < rng.choice([0.0, 0.2, 0.6]) else 1 for _ in range(n)])
    data_only = rng.choice([0, 1])
    b0, bl = rng.choice([0, 16, 8, S, 2 * S]), rng.choice([0, 16, 24, S, max(S - 16, 0), 2 * S])
    op = rng.randrange(8)
    p = ctypes.c_void_p(base + off) if rng.random() > 0.05 else None
    if os.environ.get("FUZZ_TRACE"):
        print(it, "op", op, (k, m), "S", S, "nb", nb, "stride", stride, "off", off, list(present), data_only, b0, bl, flush=True)
    h = rs._h if rng.random() > 0.03 else None
    if op == 0:
        pstride = rng.choice([0, m * S, m * S + 16, m * S + 8])
        rc = L.gec_encode_batch_dev(h, nb, p, stride, S, ctypes.c_void_p(base + (32 << 20) + rng.choice([0, 8])), pstride, None)
    elif op == 1:
        rc = L.gec_verify_batch_dev(h, nb, p, stride, S, ctypes.c_void_p(auxp + rng.choice([0, 0, 2])), None)
    elif op == 2:
        rc = L.gec_reconstruct_batch_dev(h, nb, p, stride, S, present if rng.random() > 0.05 else None, data_only, None)
    elif op == 3:
        rc = L.gec_reconstruct_range_dev(h, nb, p, stride, S, present, data_only, b0, bl, None)
    elif op == 4:
        offs = [j * S for j in range(n)]
        rng.shuffle(offs)
        if rng.random() < 0.2 and n > 1:
            offs[0] += rng.choice([8, 1])
        so = (sz * n)(*offs)
        rc = L.gec_reconstruct_scattered_dev(h, nb, p, stride, so if rng.random() > 0.05 else None, S, present, data_only, b0, bl, None)
    elif op == 5:
        ln = rng.choice([0, 1, 127, 128, 4096, S])
        rc = L.gec_blake2sum_batch_dev(h, nb * n, p, rng.choice([0, 16, max(S, 16), S + 8]), ln, ctypes.c_void_p(auxp), None)
    elif op == 6:
        rc = L.gec_shardsum_batch_dev(h, nb * n, p, rng.choice([0, 16, max(S, 16), S + 8]), S, ctypes.c_void_p(auxp), None)
    else:
        rc = L.gec_encode_hash_batch_dev(h, nb, p, stride, S, ctypes.c_void_p(auxp + rng.choice([0, 0, 8])), None)
    sync()
    calls += 1
    oks += rc == 0
sync()
# the device (or the host codec) still answers: one honest round trip
rs = g.ReedSolomon(10, 4, backend=backend)
blk = os.urandom(200_000)
par = rs.encode_blocks([blk])[0]
st = np.concatenate([np.frombuffer(blk + bytes(10 * par.size // 4 - len(blk)), dtype=np.uint8).reshape(10, -1), par.reshape(4, -1)])
assert rs.verify(st[None])[0], "the codec no longer verifies its own stripe"
print("done", calls, "ok=%d" % oks)
