"""CPU ORACLE (test infrastructure, NOT product code).

numpy restatement of `reed-solomon-erasure::galois_8` [EXT] following SURVEY.md
Appendix A, plus a ctypes loader for the C restatement (`librs_oracle.so`).
The two are written independently so that they check each other.

PARITY UNPINNED by /root/reference (Garage has no erasure coding,
doc/book/design/goals.md:27; the crate is not vendored).  Pinned instead to the
upstream known-answer vectors of SURVEY.md Appendix A.4 and to the worked 4 + 2 example of
Backblaze's article on JavaReedSolomon (tests/test_oracle_kat.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  garage_amd/ never does.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from functools import lru_cache

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

# --------------------------------------------------------------------- field
# Appendix A.1: GF(2^8), polynomial 0x11D, generator 2.


def _build_tables():
    exp = np.zeros(512, dtype=np.uint8)
    log = np.zeros(256, dtype=np.uint8)
    x = 1
    for i in range(255):
        exp[i] = x
        log[x] = i
        x <<= 1
        if x & 0x100:
            x ^= 0x11D
    exp[255:510] = exp[0:255]
    exp[510:512] = exp[0:2]
    a = np.arange(256)
    mul = exp[(log[a][:, None].astype(np.int32) + log[a][None, :].astype(np.int32))]
    mul[0, :] = 0
    mul[:, 0] = 0
    return exp, log, mul.astype(np.uint8)


EXP, LOG, MUL = _build_tables()


def gf_mul(a: int, b: int) -> int:
    return int(MUL[a, b])


def gf_div(a: int, b: int) -> int:
    if b == 0:
        raise ZeroDivisionError("GF(2^8) division by zero")
    if a == 0:
        return 0
    return int(EXP[(int(LOG[a]) - int(LOG[b])) % 255])


def gf_exp(a: int, n: int) -> int:
    """[EXT] galois_8::exp — 1 if n==0, 0 if a==0, else EXP[(LOG[a]*n) mod 255]."""
    if n == 0:
        return 1
    if a == 0:
        return 0
    return int(EXP[(int(LOG[a]) * n) % 255])


# ------------------------------------------------------------------ matrices


def vandermonde(rows: int, cols: int) -> np.ndarray:
    """[EXT] matrix.rs vandermonde: V[r][c] = exp(r, c) (Appendix A.2)."""
    return np.array([[gf_exp(r, c) for c in range(cols)] for r in range(rows)], dtype=np.uint8)


def mat_mul(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    out = np.zeros((a.shape[0], b.shape[1]), dtype=np.uint8)
    for t in range(a.shape[1]):
        out ^= MUL[a[:, t][:, None], b[t, :][None, :]]
    return out


def invert(mat: np.ndarray) -> np.ndarray:
    """[EXT] matrix.rs invert/gaussian_elim (Gauss-Jordan, Appendix A.2)."""
    n = mat.shape[0]
    a = np.concatenate([mat.astype(np.uint8), np.eye(n, dtype=np.uint8)], axis=1)
    for r in range(n):
        if a[r, r] == 0:
            for rb in range(r + 1, n):
                if a[rb, r] != 0:
                    a[[r, rb]] = a[[rb, r]]
                    break
            else:
                raise ValueError("singular matrix")
        if a[r, r] != 1:
            s = gf_div(1, int(a[r, r]))
            a[r] = MUL[s, a[r]]
        for rb in range(r + 1, n):
            f = int(a[rb, r])
            if f:
                a[rb] ^= MUL[f, a[r]]
    for d in range(n):
        for ra in range(d):
            f = int(a[ra, d])
            if f:
                a[ra] ^= MUL[f, a[d]]
    return a[:, n:].copy()


@lru_cache(maxsize=None)
def _build_matrix_cached(k: int, m: int) -> bytes:
    v = vandermonde(k + m, k)
    return mat_mul(v, invert(v[:k])).tobytes()


def build_matrix(k: int, m: int) -> np.ndarray:
    """[EXT] core.rs build_matrix = vandermonde(n,k) x invert(top k rows)."""
    if k <= 0 or m <= 0 or k + m > 256:
        raise ValueError("bad (k, m)")
    return np.frombuffer(_build_matrix_cached(k, m), dtype=np.uint8).reshape(k + m, k).copy()


def build_matrix_cauchy(k: int, m: int) -> np.ndarray:
    """The project's extra matrix family (NOT the crate's): identity on top, parity row
    r / column c = 1 / ((k + r) ^ c).  Same formula as klauspost/reedsolomon's
    WithCauchyMatrix [EXT, recalled].  Restated here only so the GPU path has an
    independent check for this mode."""
    if k <= 0 or m <= 0 or k + m > 256:
        raise ValueError("bad (k, m)")
    M = np.zeros((k + m, k), dtype=np.uint8)
    M[:k] = np.eye(k, dtype=np.uint8)
    for r in range(m):
        for c in range(k):
            M[k + r, c] = gf_div(1, (k + r) ^ c)
    return M


def parity_matrix(k: int, m: int) -> np.ndarray:
    return build_matrix(k, m)[k:]


def decode_matrix(k: int, m: int, present) -> tuple[list[int], np.ndarray]:
    """[EXT] core.rs get_data_decode_matrix: first k present rows, inverted."""
    valid = [j for j in range(k + m) if present[j]][:k]
    if len(valid) < k:
        raise ValueError("too few shards present")
    return valid, invert(build_matrix(k, m)[valid])


# ---------------------------------------------------------------- operations


def _apply(rows: np.ndarray, inputs: np.ndarray) -> np.ndarray:
    """out[r] = XOR_i rows[r][i] * inputs[i]; inputs is (k, ...) uint8."""
    out = np.zeros((rows.shape[0],) + inputs.shape[1:], dtype=np.uint8)
    for r in range(rows.shape[0]):
        acc = out[r]
        for i in range(rows.shape[1]):
            c = int(rows[r, i])
            if c:
                acc ^= MUL[c][inputs[i]]
    return out


def encode(k: int, m: int, data: np.ndarray) -> np.ndarray:
    """data: (k, S) uint8 (or (k, ...) any trailing shape) -> parity (m, ...)."""
    data = np.asarray(data, dtype=np.uint8)
    assert data.shape[0] == k
    return _apply(parity_matrix(k, m), data)


def verify(k: int, m: int, shards: np.ndarray) -> bool:
    shards = np.asarray(shards, dtype=np.uint8)
    return bool(np.array_equal(encode(k, m, shards[:k]), shards[k:]))


def reconstruct(k: int, m: int, shards: np.ndarray, present, data_only: bool = False) -> np.ndarray:
    """shards: (k+m, S); rows with present[j]==False are ignored and rebuilt.
    Order follows the crate: missing data from the first k present shards, then
    missing parity re-encoded from the complete data."""
    shards = np.array(shards, dtype=np.uint8, copy=True)
    n = k + m
    present = [bool(p) for p in present]
    if all(present):
        return shards
    if sum(present) < k:
        raise ValueError("too few shards present")
    M = build_matrix(k, m)
    missing_data = [j for j in range(k) if not present[j]]
    if missing_data:
        valid, D = decode_matrix(k, m, present)
        shards[missing_data] = _apply(D[missing_data], shards[valid])
    if not data_only:
        missing_par = [j for j in range(k, n) if not present[j]]
        if missing_par:
            shards[missing_par] = _apply(M[missing_par], shards[:k])
    return shards


# --------------------------------------------------------- block <-> shards


def shard_len(k: int, block_len: int) -> int:
    """S = round_up(ceil(L/k), 64)  (SURVEY.md section 7 step 2)."""
    per = -(-max(block_len, 1) // k)
    return -(-per // 64) * 64


def split_block(k: int, block: bytes | np.ndarray, S: int | None = None) -> np.ndarray:
    b = np.frombuffer(bytes(block), dtype=np.uint8) if not isinstance(block, np.ndarray) else block
    if S is None:
        S = shard_len(k, b.size)
    out = np.zeros(k * S, dtype=np.uint8)
    out[: b.size] = b
    return out.reshape(k, S)


# ------------------------------------------------------------ inputs (8d)

_MASK64 = (1 << 64) - 1


def splitmix64_bytes(seed: int, nbytes: int) -> np.ndarray:
    """SplitMix64 stream written as little-endian u64."""
    n = -(-nbytes // 8)
    with np.errstate(over="ignore"):
        idx = np.arange(1, n + 1, dtype=np.uint64)
        z = np.uint64(seed & _MASK64) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z.astype("<u8").view(np.uint8)[:nbytes].copy()


def golden_pattern(k: int, L: int) -> np.ndarray:
    """Appendix A.4 item 6: d[s][i] = (131*s + 7*i + (i>>8) + 1) & 0xFF."""
    s = np.arange(k, dtype=np.int64)[:, None]
    i = np.arange(L, dtype=np.int64)[None, :]
    return ((131 * s + 7 * i + (i >> 8) + 1) & 0xFF).astype(np.uint8)


# ------------------------------------------------------------- C oracle lib


class COracle:
    """ctypes view of oracle/librs_oracle.so (built by oracle/Makefile)."""

    SCALAR, AVX2 = 0, 1

    def __init__(self, build: bool = True):
        path = os.path.join(_HERE, "librs_oracle.so")
        if build and not os.path.exists(path):
            subprocess.run(["make", "-C", _HERE], check=True, capture_output=True)
        self.lib = ctypes.CDLL(path)
        L = self.lib
        u8p = ctypes.POINTER(ctypes.c_uint8)
        L.rso_gf_mul.restype = ctypes.c_uint8
        L.rso_gf_mul.argtypes = [ctypes.c_uint8, ctypes.c_uint8]
        L.rso_gf_div.restype = ctypes.c_uint8
        L.rso_gf_div.argtypes = [ctypes.c_uint8, ctypes.c_uint8]
        L.rso_gf_exp.restype = ctypes.c_uint8
        L.rso_gf_exp.argtypes = [ctypes.c_uint8, ctypes.c_uint]
        L.rso_invert.argtypes = [ctypes.c_int, u8p, u8p]
        L.rso_build_matrix.argtypes = [ctypes.c_int, ctypes.c_int, u8p]
        L.rso_decode_matrix.argtypes = [ctypes.c_int, ctypes.c_int, u8p, ctypes.POINTER(ctypes.c_int), u8p]
        L.rso_encode_batch.argtypes = [
            ctypes.c_int, ctypes.c_int, ctypes.c_size_t, ctypes.c_size_t,
            ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t,
            ctypes.c_int, ctypes.c_int,
        ]
        L.rso_reconstruct_batch.argtypes = [
            ctypes.c_int, ctypes.c_int, ctypes.c_size_t, ctypes.c_size_t,
            ctypes.c_void_p, ctypes.c_size_t, u8p, ctypes.c_int, ctypes.c_int,
        ]
        L.rso_bench_encode.restype = ctypes.c_double
        L.rso_bench_encode.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int,
                                       ctypes.c_int, ctypes.c_int, ctypes.c_uint64, u8p]
        L.rso_has_avx2.restype = ctypes.c_int
        L.rso_max_threads.restype = ctypes.c_int

    @staticmethod
    def _p(a: np.ndarray):
        return a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))

    def has_avx2(self) -> bool:
        return bool(self.lib.rso_has_avx2())

    def max_threads(self) -> int:
        return int(self.lib.rso_max_threads())

    def bench_encode(self, k: int, m: int, S: int, nblocks: int, reps: int, variant: int, threads: int,
                     seed: int = 1) -> float:
        """seconds per encode of `nblocks` blocks (median of reps), NUMA first-touch inside C."""
        cs = ctypes.c_uint8()
        t = self.lib.rso_bench_encode(k, m, S, nblocks, reps, variant, threads, seed, ctypes.byref(cs))
        if t <= 0:
            raise ValueError("rso_bench_encode failed")
        self.last_bench_checksum = int(cs.value)  # XOR over a sample of the parity bytes: the same for any thread count
        return float(t)

    def invert(self, mat: np.ndarray) -> np.ndarray:
        mat = np.ascontiguousarray(mat, dtype=np.uint8)
        out = np.zeros_like(mat)
        rc = self.lib.rso_invert(mat.shape[0], self._p(mat), self._p(out))
        if rc:
            raise ValueError(f"rso_invert rc={rc}")
        return out

    def build_matrix(self, k: int, m: int) -> np.ndarray:
        out = np.zeros((k + m, k), dtype=np.uint8)
        rc = self.lib.rso_build_matrix(k, m, self._p(out))
        if rc:
            raise ValueError(f"rso_build_matrix rc={rc}")
        return out

    def decode_matrix(self, k: int, m: int, present) -> tuple[list[int], np.ndarray]:
        pres = np.ascontiguousarray(np.asarray(present, dtype=np.uint8))
        valid = (ctypes.c_int * k)()
        out = np.zeros((k, k), dtype=np.uint8)
        rc = self.lib.rso_decode_matrix(k, m, self._p(pres), valid, self._p(out))
        if rc:
            raise ValueError(f"rso_decode_matrix rc={rc}")
        return list(valid), out

    def encode_batch(self, k: int, m: int, data: np.ndarray, variant: int = 0, threads: int = 1) -> np.ndarray:
        """data: (nblocks, k, S) contiguous -> parity (nblocks, m, S)."""
        data = np.ascontiguousarray(data, dtype=np.uint8)
        nb, kk, S = data.shape
        assert kk == k
        parity = np.empty((nb, m, S), dtype=np.uint8)
        rc = self.lib.rso_encode_batch(k, m, S, nb, data.ctypes.data, k * S,
                                       parity.ctypes.data, m * S, variant, threads)
        if rc:
            raise ValueError(f"rso_encode_batch rc={rc}")
        return parity

    def reconstruct_batch(self, k: int, m: int, stripes: np.ndarray, present, data_only=False, threads: int = 1) -> np.ndarray:
        """stripes: (nblocks, k+m, S); returns a repaired copy."""
        st = np.array(stripes, dtype=np.uint8, copy=True, order="C")
        nb, n, S = st.shape
        assert n == k + m
        pres = np.ascontiguousarray(np.asarray(present, dtype=np.uint8))
        rc = self.lib.rso_reconstruct_batch(k, m, S, nb, st.ctypes.data, n * S,
                                            self._p(pres), int(bool(data_only)), threads)
        if rc:
            raise ValueError(f"rso_reconstruct_batch rc={rc}")
        return st
