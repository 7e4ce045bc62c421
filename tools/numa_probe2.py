import ctypes, os, json, time, threading, mmap
libc = ctypes.CDLL(None, use_errno=True)
hip = ctypes.CDLL("libamdhip64.so")
libc.syscall.restype = ctypes.c_long
hip.hipHostMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]
hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
hip.hipHostRegister.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint]
hip.hipHostFree.argtypes = [ctypes.c_void_p]
N = 512 << 20
assert hip.hipSetDevice(0) == 0
dev = ctypes.c_void_p(); assert hip.hipMalloc(ctypes.byref(dev), N) == 0
def node_of(addr, nbytes, samples=64):
    pages = (ctypes.c_void_p * samples)(*[addr + (i * (nbytes // samples)) // 4096 * 4096 for i in range(samples)])
    status = (ctypes.c_int * samples)()
    r = libc.syscall(279, 0, samples, pages, None, status, 0)
    h = {}
    for s in status: h[int(s)] = h.get(int(s), 0) + 1
    return h if r == 0 else os.strerror(ctypes.get_errno())
def bw(p, kind):
    best = 1e9
    for _ in range(4):
        t0 = time.perf_counter()
        assert (hip.hipMemcpy(dev, p, N, 1) if kind == "h2d" else hip.hipMemcpy(p, dev, N, 2)) == 0
        best = min(best, time.perf_counter() - t0)
    return round(N / best / 2**30, 2)
out = {}
for node in (0, 1):
    mask = (ctypes.c_ulong * 16)(); mask[0] = 1 << node
    for mode, mname in ((2, "MPOL_BIND"), (1, "MPOL_PREFERRED")):
        for flags, fname in ((0x20000000 | 1, "NumaUser|Portable"), (1, "Portable")):
            r = libc.syscall(238, mode, mask, 1024)
            p = ctypes.c_void_p()
            rc = hip.hipHostMalloc(ctypes.byref(p), N, flags)
            libc.syscall(238, 0, None, 0)
            if rc: out[f"{mname} node {node}, {fname}"] = {"hipHostMalloc rc": rc}; continue
            ctypes.memset(p.value, 1, N)
            out[f"{mname} node {node}, {fname}"] = {"set_mempolicy": r, "pages_on": node_of(p.value, N), "h2d": bw(p.value, "h2d"), "d2h": bw(p.value, "d2h")}
            hip.hipHostFree(p)
    # mmap + mbind + first touch + hipHostRegister
    libc.mmap.restype = ctypes.c_void_p
    libc.mmap.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_long]
    a = libc.mmap(None, N, 3, 0x22, -1, 0)
    r = libc.syscall(237, ctypes.c_void_p(a), ctypes.c_ulong(N), 2, mask, 1024, 0)
    ctypes.memset(a, 1, N)
    rc = hip.hipHostRegister(a, N, 1 | 2)
    out[f"mmap+mbind node {node}+hipHostRegister"] = {"mbind": r, "register rc": rc, "pages_on": node_of(a, N), "h2d": bw(a, "h2d") if rc == 0 else None, "d2h": bw(a, "d2h") if rc == 0 else None}
print(json.dumps(out, indent=1))
