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
