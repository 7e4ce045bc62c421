"""Run by tests/test_cabi_host.py in a process of its own (a crash must not take pytest with it): 300 calls into the host-pointer
entry points of libgarage_ec with arguments that are IN CONTRACT as far as memory goes (every array as long as the header says,
every buffer as large) and otherwise arbitrary -- S of 0 / 1 / 63 / 65, no blocks, NULL blocks / shards / outputs, blocks longer
than k*S, too few shards present, a rebuilt pointer missing where one is due.  Every call must come back with a code.
usage: abi_fuzz.py <seed> [cpu|hip]; prints "done k m" """
import ctypes, sys, random
import numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import garage_amd as g
from garage_amd import _lib
L = _lib.lib
U8P = ctypes.POINTER(ctypes.c_uint8)
rng = random.Random(int(sys.argv[1]))
k, m = rng.choice([(3, 1), (10, 4), (5, 5), (1, 1), (20, 8)])
rs = g.ReedSolomon(k, m, backend=sys.argv[2] if len(sys.argv) > 2 else "cpu")
n = k + m
h = rs._h
def arr(nbytes): return np.random.default_rng(rng.randrange(1 << 30)).integers(0, 256, max(nbytes, 1), dtype=np.uint8)
for it in range(300):
    S = rng.choice([0, 1, 63, 64, 128, 4096, 65, 1 << 16])
    nb = rng.choice([0, 1, 2, 7])
    op = rng.choice(["enc", "enchash", "verify", "recon", "reconhash", "decver", "hash"])
    keep = []
    if os.environ.get("FUZZ_TRACE"):
        print(it, op, "S", S, "nb", nb, flush=True)
    def ptrs(count, size, null_p=0.0):
        a = (ctypes.c_void_p * max(count, 1))()
        for i in range(count):
            if rng.random() < null_p:
                a[i] = None
            else:
                b = arr(size); keep.append(b); a[i] = b.ctypes.data
        return a
    try:
        if op in ("enc", "enchash"):
            lens_l = [rng.choice([0, 1, k * S, k * S + 1, max(k * S - 5, 0)]) for _ in range(nb)]
            blocks = (ctypes.c_void_p * max(nb, 1))()
            for i, ln in enumerate(lens_l):
                b = arr(ln); keep.append(b); blocks[i] = b.ctypes.data if rng.random() > 0.05 else None
            lens = (ctypes.c_size_t * max(nb, 1))(*lens_l) if nb else (ctypes.c_size_t * 1)()
            par = ptrs(nb, m * max(S, 1), 0.05)
            sums = arr(nb * n * 32)
            if op == "enc":
                rc = L.gec_encode_batch(h, nb, blocks, lens, S, par)
            else:
                rc = L.gec_encode_hash_batch(h, nb, blocks, lens, S, par, sums.ctypes.data_as(U8P) if rng.random() > 0.1 else None)
        elif op == "verify":
            sh = ptrs(nb * n, max(S, 1), 0.05)
            ok = arr(nb)
            rc = L.gec_verify_batch(h, nb, sh, S, ok.ctypes.data_as(U8P))
        elif op in ("recon", "reconhash"):
            sh = ptrs(nb * n, max(S, 1), 0.3)
            out = ptrs(nb * n, max(S, 1), 0.3)
            if op == "recon":
                rc = L.gec_reconstruct_batch(h, nb, sh, out, S, rng.choice([0, 1]))
            else:
                a, b = arr(nb * n * 32), arr(nb * n * 32)
                rc = L.gec_reconstruct_hash_batch(h, nb, sh, out, S, rng.choice([0, 1]), a.ctypes.data_as(U8P), b.ctypes.data_as(U8P))
        elif op == "decver":
            sh = ptrs(nb * n, max(S, 1), 0.3)
            reb = ptrs(nb * n, max(S, 1), 0.2)
            lens = (ctypes.c_size_t * max(nb, 1))(*[rng.choice([0, 1, k * S, max(k * S - 3, 0)]) for _ in range(nb)])
            a, b = arr(nb * n * 32), arr(nb * 32)
            rc = L.gec_decode_verify_batch(h, nb, sh, S, lens, reb, a.ctypes.data_as(U8P), b.ctypes.data_as(U8P) if rng.random() > 0.3 else None)
        else:
            cnt = rng.choice([0, 1, 5, 33])
            lens_l = [rng.choice([0, 1, 127, 128, 129, 4096, 70000]) for _ in range(cnt)]
            msgs = (ctypes.c_void_p * max(cnt, 1))()
            for i, ln in enumerate(lens_l):
                b = arr(ln); keep.append(b); msgs[i] = b.ctypes.data
            lens = (ctypes.c_size_t * max(cnt, 1))(*lens_l) if cnt else (ctypes.c_size_t * 1)()
            out = arr(cnt * 32)
            fn = rng.choice([L.gec_blake2sum_batch, L.gec_shardsum_batch])
            rc = fn(h, cnt, msgs, lens, out.ctypes.data_as(U8P))
    except ctypes.ArgumentError as e:
        print("argerr", op, e); continue
print("done", k, m)
