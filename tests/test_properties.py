"""Property tests (hypothesis) on CPU: the two oracle restatements against each
other and the product's host logic against both, over random codes, shard
lengths and erasure patterns."""
import numpy as np
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

import garage_amd as g
from oracle import rs_oracle as O

SET = settings(max_examples=40, deadline=None, suppress_health_check=[HealthCheck.too_slow, HealthCheck.function_scoped_fixture])


@st.composite
def code_and_pattern(draw):
    k = draw(st.integers(1, 24))
    m = draw(st.integers(1, 8))
    nlost = draw(st.integers(0, m))
    lost = draw(st.lists(st.integers(0, k + m - 1), min_size=nlost, max_size=nlost, unique=True))
    return k, m, sorted(lost)


@SET
@given(cp=code_and_pattern(), cols=st.integers(1, 9), seed=st.integers(0, 2**32 - 1))
def test_c_and_numpy_oracles_agree(coracle, cp, cols, seed):
    k, m, lost = cp
    S = 64 * cols
    data = np.random.default_rng(seed).integers(0, 256, (2, k, S), dtype=np.uint8)
    par = coracle.encode_batch(k, m, data, coracle.AVX2)
    assert np.array_equal(par, coracle.encode_batch(k, m, data, coracle.SCALAR))
    assert np.array_equal(par[0], O.encode(k, m, data[0]))
    full = np.concatenate([data, par], axis=1)
    present = [j not in lost for j in range(k + m)]
    broken = full.copy()
    broken[:, lost] = 0x3C
    assert np.array_equal(coracle.reconstruct_batch(k, m, broken, present), full)
    assert np.array_equal(O.reconstruct(k, m, broken[1], present), full[1])


@SET
@given(cp=code_and_pattern())
def test_product_matrices_match_oracle(cp):
    k, m, lost = cp
    M = g.build_matrix(k, m)
    assert np.array_equal(M, O.build_matrix(k, m))
    present = [j not in lost for j in range(k + m)]
    valid, D = g.build_decode_matrix(k, m, present)
    v2, D2 = O.decode_matrix(k, m, present)
    assert valid == v2 and np.array_equal(D, D2)
    # D really inverts the chosen rows
    assert np.array_equal(O.mat_mul(D, M[valid]), np.eye(k, dtype=np.uint8))


@SET
@given(k=st.integers(1, 255), L=st.integers(0, 1 << 26))
def test_shard_len_properties(k, L):
    S = g.shard_len(k, L)
    assert S == O.shard_len(k, L)
    assert S % 64 == 0 and S >= 64 and k * S >= L
    assert k * (S - 64) < max(L, 1) or S == 64      # smallest multiple of 64 that fits


# ------------------------------------------------------------------------------------
# An independent characterisation of the code, sharing no code with the matrix path:
# vandermonde(n, k) * inverse(top k rows) means that for every byte column there is ONE
# polynomial p of degree < k over GF(2^8) with p(i) = data_i for i < k (the row index i
# used as the field element), and shard j of the codeword is p(j).  Parity computed by
# Lagrange interpolation (pure Python, log/antilog generated here from x -> 2x mod 0x11D)
# must equal the oracle's matrix encode -- a check of the oracle's matrix logic that does
# not go through Gauss-Jordan or a matrix product.
def _field():
    exp, log = [0] * 510, [0] * 256
    x = 1
    for i in range(255):
        exp[i] = exp[i + 255] = x
        log[x] = i
        x <<= 1
        if x & 0x100:
            x ^= 0x11D
    return exp, log


def _lagrange_parity(k, m, data):
    """data: k lists of ints (one per shard) -> m lists"""
    exp, log = _field()

    def mul(a, b):
        return 0 if a == 0 or b == 0 else exp[log[a] + log[b]]

    def inv(a):
        return exp[255 - log[a]]

    out = []
    for r in range(m):
        x = k + r
        # weights L_i(x) = prod_{j != i} (x - j) / (i - j); subtraction is XOR
        w = []
        for i in range(k):
            num, den = 1, 1
            for j in range(k):
                if j != i:
                    num = mul(num, x ^ j)
                    den = mul(den, i ^ j)
            w.append(mul(num, inv(den)))
        out.append([_xor_all(mul(w[i], data[i][b]) for i in range(k)) for b in range(len(data[0]))])
    return out


def _xor_all(it):
    acc = 0
    for v in it:
        acc ^= v
    return acc


@settings(max_examples=25, deadline=None, suppress_health_check=[HealthCheck.too_slow, HealthCheck.function_scoped_fixture])
@given(k=st.integers(1, 20), m=st.integers(1, 8), nbytes=st.integers(1, 24), seed=st.integers(0, 2**32 - 1))
def test_oracle_encode_equals_lagrange_interpolation(coracle, k, m, nbytes, seed):
    rng = np.random.default_rng(seed)
    data = rng.integers(0, 256, size=(1, k, 64), dtype=np.uint8)
    want = _lagrange_parity(k, m, [data[0, i, :nbytes].tolist() for i in range(k)])
    got = coracle.encode_batch(k, m, data, coracle.SCALAR)[0]
    assert got[:, :nbytes].tolist() == want
    # and the parity rows of the product's host logic are exactly the Lagrange weights
    mat = g.build_matrix(k, m)
    for c in range(k):
        e = [[1 if i == c else 0] for i in range(k)]
        col = _lagrange_parity(k, m, e)
        assert [mat[k + r][c] for r in range(m)] == [col[r][0] for r in range(m)]
