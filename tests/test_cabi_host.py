"""CPU-only checks of the drop-in boundary: libgarage_ec.so loads, exports every
symbol include/garage_ec.h declares, and its host logic (geometry, matrices,
argument/error behaviour) agrees with the oracle.  No kernels run here."""
import ctypes
import os
import re

import numpy as np
import pytest

import garage_amd as g
from garage_amd import _lib
from oracle import rs_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "garage_ec.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gec_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported():
    names = header_functions()
    assert len(names) >= 20
    assert sorted(_lib.SYMBOLS) == names, "garage_amd/_lib.py SYMBOLS out of sync with the header"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert getattr(lib, n) is not None


def test_version_and_strerror():
    assert _lib.lib.gec_version() == 0x00040000
    assert _lib.lib.gec_strerror(0) == b"ok"
    assert b"present" in _lib.lib.gec_strerror(_lib.GEC_E_TOO_FEW_PRESENT)
    assert _lib.lib.gec_device_count() >= 0


@pytest.mark.parametrize("k,L,S", [(3, 65536, 21888), (10, 1048576, 104896), (20, 4194304, 209728),
                                   (10, 0, 64), (10, 1, 64), (10, 640, 64), (10, 641, 128), (1, 100, 128)])
def test_shard_len(k, L, S):
    assert g.shard_len(k, L) == S == O.shard_len(k, L)


def test_shard_len_bad_k():
    assert g.shard_len(0, 100) == 0


@pytest.mark.parametrize("k,m", [(3, 1), (10, 4), (20, 8), (5, 5), (1, 1), (2, 3), (17, 3), (100, 20), (200, 56), (255, 1)])
def test_encoding_matrix_matches_oracle(k, m):
    M = g.build_matrix(k, m)
    assert np.array_equal(M, O.build_matrix(k, m))
    assert np.array_equal(M[:k], np.eye(k, dtype=np.uint8))


def test_decode_matrix_matches_oracle_and_kat():
    present = [j not in (0, 3, 7, 11) for j in range(14)]
    valid, D = g.build_decode_matrix(10, 4, present)
    assert valid == [1, 2, 4, 5, 6, 8, 9, 10, 12, 13]
    assert D[0].tolist() == [204, 75, 104, 156, 114, 211, 108, 57, 186, 60]
    rng = np.random.default_rng(3)
    for k, m in [(3, 1), (10, 4), (20, 8), (6, 6)]:
        for _ in range(10):
            lost = rng.choice(k + m, size=rng.integers(0, m + 1), replace=False)
            present = [j not in lost for j in range(k + m)]
            v, D = g.build_decode_matrix(k, m, present)
            v2, D2 = O.decode_matrix(k, m, present)
            assert v == v2 and np.array_equal(D, D2)


def test_error_codes_mirror_the_crate():
    # ReedSolomon::new argument errors come before any device access
    for k, m, code in [(0, 4, _lib.GEC_E_TOO_FEW_DATA), (-1, 4, _lib.GEC_E_TOO_FEW_DATA),
                       (4, 0, _lib.GEC_E_TOO_FEW_PARITY), (200, 57, _lib.GEC_E_TOO_MANY_SHARDS)]:
        with pytest.raises(g.GecError) as ei:
            g.ReedSolomon(k, m)
        assert ei.value.code == code
        with pytest.raises(g.GecError) as ei:
            g.build_matrix(k, m)
        assert ei.value.code == code
    with pytest.raises(g.GecError) as ei:
        g.build_decode_matrix(4, 2, [1, 1, 1, 0, 0, 0])
    assert ei.value.code == _lib.GEC_E_TOO_FEW_PRESENT


def test_backends_without_a_gpu():
    """SURVEY.md Appendix B's `backend` argument: without a device a HIP codec is refused (and says what would
    work), AUTO resolves to the host cores, a CPU codec has no device-resident entry points."""
    if _lib.lib.gec_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(g.GecError) as ei:
        g.ReedSolomon(10, 4)  # backend="hip" is the default
    assert ei.value.code == _lib.GEC_E_DEVICE
    assert "GEC_BACKEND_CPU" in str(ei.value)
    rs = g.ReedSolomon(10, 4, backend="auto")
    assert rs.backend == "cpu" and rs.device == -1
    h = ctypes.c_void_p()
    assert _lib.lib.gec_codec_create(10, 4, 7, 0, ctypes.byref(h)) == _lib.GEC_E_INVALID_ARG
    # no device-resident encode / verify / hashing on a CPU codec, and no RCCL group (RCCL moves device memory) ...
    buf = np.zeros(14 * 64 + 64, dtype=np.uint8)
    base = (buf.ctypes.data + 15) // 16 * 16
    assert _lib.lib.gec_encode_batch_dev(rs._h, 1, base, 14 * 64, 64, base + 640, 14 * 64, None) == _lib.GEC_E_DEVICE
    ident = (ctypes.c_uint8 * _lib.GEC_GROUP_ID_BYTES)()
    assert _lib.lib.gec_group_create(rs._h, 0, 1, ident, ctypes.byref(h)) == _lib.GEC_E_DEVICE
    # ... but a group over a caller transport works on host buffers (round 4: a rank that lost its GPU stays in the group;
    # tests/test_group_multiprocess.py drives it with real processes), and so does the strided reconstruct
    fn = _lib.ALLGATHER_FN(lambda *a: 0)
    assert _lib.lib.gec_group_create_with_transport(rs._h, 0, 1, fn, None, ctypes.byref(h)) == _lib.GEC_OK
    _lib.lib.gec_group_destroy(h)
    # a background sibling keeps code and backend
    bg = rs.background()
    assert bg.backend == "cpu" and bg.qos_class == _lib.GEC_CLASS_BACKGROUND and rs.qos_class == _lib.GEC_CLASS_FOREGROUND
    assert np.array_equal(bg.parity_matrix(), rs.parity_matrix())


def test_env_tables_name_every_switch_the_sources_read():
    """One table per library (gec_env_table / gbm_env_table); every GEC_* / GBM_* name that appears in a getenv of the
    product sources is in it, and nothing reads the environment outside the two table files."""
    from garage_amd import block_native as bn

    gec = {ln.split("\t")[0] for ln in _lib.lib.gec_env_table().decode().splitlines()}
    gbm = {ln.split("\t")[0] for ln in bn.lib.gbm_env_table().decode().splitlines()}
    assert "GEC_UPLOAD_CUS" in gec and "GEC_CPU_ISA" in gec and "GBM_TRACE" in gbm
    csrc = os.path.join(ROOT, "garage_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if not f.endswith((".cpp", ".hip", ".hpp")):
            continue
        txt = open(os.path.join(csrc, f)).read()
        if "getenv" in txt:
            assert f in ("ec_env.cpp", "bm_core.cpp"), f"{f} reads the environment outside the tables"
        for name in set(re.findall(r'"(GEC_[A-Z0-9_]+|GBM_[A-Z0-9_]+)"', txt)):
            assert name in gec or name in gbm, f"{name} ({f}) is not in the environment tables"


def test_the_documents_list_exactly_the_switches_the_libraries_read():
    """DESIGN.md section 0 sorts every switch into what it is for, INTEGRATION.md section 6 prints the libraries' tables: both
    name exactly the switches of gec_env_table / gbm_env_table -- a switch added or removed in the sources shows up here."""
    from garage_amd import block_native as bn

    table = {ln.split("\t")[0] for ln in (_lib.lib.gec_env_table() + b"\n" + bn.lib.gbm_env_table()).decode().splitlines() if ln.strip()}
    design = open(os.path.join(ROOT, "DESIGN.md")).read()
    para = design[design.index("**Switches** ("):design.index("## 1. Scope")]
    assert int(re.search(r"\*\*Switches\*\* \((\d+);", para).group(1)) == len(table)
    assert set(re.findall(r"`(GEC_[A-Z0-9_]+|GBM_[A-Z0-9_]+)`", para)) == table
    integ = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    rows = re.findall(r"^\| `(GEC_[A-Z0-9_]+|GBM_[A-Z0-9_]+)` \|", integ, flags=re.M)
    assert set(rows) == table and len(rows) == len(table)


def test_group_entry_points_reject_bad_arguments_without_a_gpu():
    """gec_group_*: argument checks come before any device / RCCL work."""
    import ctypes

    lib = _lib.lib
    h = ctypes.c_void_p()
    fn = _lib.ALLGATHER_FN(lambda *a: 0)
    assert lib.gec_group_create_with_transport(None, 0, 1, fn, None, ctypes.byref(h)) == _lib.GEC_E_INVALID_ARG
    assert lib.gec_group_create(None, 0, 1, (ctypes.c_uint8 * 128)(), ctypes.byref(h)) == _lib.GEC_E_INVALID_ARG
    assert lib.gec_group_unique_id(None) == _lib.GEC_E_INVALID_ARG
    assert lib.gec_group_rank(None) == -1 and lib.gec_group_size(None) == 0 and lib.gec_group_slots(None) == 0
    assert lib.gec_group_allgather_decode(None, 1, None, 64, None, 0, 1, None, None) == _lib.GEC_E_INVALID_ARG
    lib.gec_group_destroy(None)


def test_launch_geometry_invariants_for_every_k_and_row_count():
    """Every (k, rows_left) the library can be asked for: the launch makes progress, its tables fit
    the 64 KiB of LDS a workgroup gets without opting in, the 16-row form stays within k <= 120
    (k*16 coefficient bytes in the 2 KiB argument array), and the loads per batch never exceed
    what the kernels are instantiated for."""
    import ctypes

    lib = _lib.lib
    rows, ent, kc, thr = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    lds = ctypes.c_size_t()
    seen = set()
    for k in range(1, 256):
        for left in list(range(1, 34)) + [64, 255]:
            assert lib.gec_launch_geometry(k, left, rows, ent, kc, thr, lds) == 0
            r, e, c, t, l = rows.value, ent.value, kc.value, thr.value, lds.value
            assert 1 <= r <= min(left, 16)
            assert e == (4 if r <= 4 else 8 if r <= 8 else 16)
            assert l <= 65536 and l == k * 32 * e + 768 + k * (16 if e == 16 else 8)
            assert t in (256, 512)
            if e == 16:
                assert k <= 120 and k * 16 <= 2048 and c == min(k, 4)
            elif e == 8:
                assert c in (1, 2, 3, 4, 5, 6, 10) and (c != 10 or t == 256)
            else:
                assert c in (1, 2, 3, 4, 5, 6, 10, 12, 16)
            assert c <= k or (c in (10, 12, 16) and k <= c)   # more loads than shards only in the single-batch forms
            if left > 8 and k <= 120:
                assert r == min(left, 16)          # one pass over the data for up to 16 rows
            seen.add((e, c, t))
    assert (4, 10, 256) in seen and (8, 5, 512) in seen and (8, 10, 256) in seen and (16, 4, 512) in seen
    assert lib.gec_launch_geometry(0, 1, rows, ent, kc, thr, lds) == _lib.GEC_E_INVALID_ARG
    assert lib.gec_launch_geometry(10, 0, rows, ent, kc, thr, lds) == _lib.GEC_E_INVALID_ARG


def test_product_never_imports_oracle():
    """Nothing under garage_amd/ (the CPU backend ec_cpu.cpp included) or include/ names the oracle, and neither
    shared library links it."""
    import subprocess

    for top in ("garage_amd", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, top)):
            for f in files:
                if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h", ".map")) or f == "Makefile":
                    txt = open(os.path.join(dp, f)).read()
                    assert "rs_oracle" not in txt and "import oracle" not in txt and "from oracle" not in txt, f
                    for line in txt.splitlines():  # no source includes, links or opens anything under oracle/
                        if "oracle" in line and not line.lstrip().startswith(("//", "*", "#", '"""')):
                            assert not re.search(r'#include|dlopen|-l|\.so|open\(', line), f"{f}: {line}"
    for so in ("libgarage_ec.so", "libgarage_block.so"):
        out = subprocess.run(["ldd", os.path.join(ROOT, "garage_amd", so)], capture_output=True, text=True).stdout
        assert "rs_oracle" not in out


# ------------------------------------------------- Cauchy family (extra mode)
def test_cauchy_matrix_matches_restatement_and_is_mds():
    import itertools

    for k, m in [(3, 1), (10, 4), (20, 8), (5, 5), (200, 56)]:
        M = g.build_matrix(k, m, "cauchy")
        assert np.array_equal(M, O.build_matrix_cauchy(k, m))
        assert np.array_equal(M[:k], np.eye(k, dtype=np.uint8))
    # exhaustive MDS check on a small code: every k rows of the (k+m) x k matrix are independent
    k, m = 4, 3
    M = g.build_matrix(k, m, "cauchy")
    for rows in itertools.combinations(range(k + m), k):
        O.invert(M[list(rows)])            # raises if singular
    # not interchangeable with the crate-compatible default
    assert not np.array_equal(g.build_matrix(10, 4, "cauchy"), g.build_matrix(10, 4))
    with pytest.raises(g.GecError):
        _lib.check(_lib.lib.gec_build_matrix_ex(10, 4, 7, M.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))), "kind")


def test_link_release_hook_fires_once_per_armed_call():
    """gec_thread_link_release: the hook fires exactly once for the next gec_encode_hash_batch / gec_decode_verify_batch of the
    arming thread (a CPU codec has no link phase: right before the call returns), on errors too, and never when disarmed."""
    lib = ctypes.CDLL(_lib.LIB_PATH)
    FN = ctypes.CFUNCTYPE(None, ctypes.c_void_p)
    fired = []
    cb = FN(lambda arg: fired.append(arg))
    lib.gec_thread_link_release.argtypes = [FN, ctypes.c_void_p]
    lib.gec_thread_link_release.restype = None
    rs = g.ReedSolomon(3, 1, backend="cpu")
    S = 64
    data = np.arange(3 * S, dtype=np.uint8)
    parity = np.zeros(S, dtype=np.uint8)
    sums = np.zeros(4 * 32, dtype=np.uint8)
    u8p = ctypes.POINTER(ctypes.c_uint8)
    blocks = (ctypes.c_void_p * 1)(data.ctypes.data)
    pars = (ctypes.c_void_p * 1)(parity.ctypes.data)
    lens = (ctypes.c_size_t * 1)(3 * S)

    def encode(sums_ptr):
        return lib.gec_encode_hash_batch(ctypes.c_void_p(rs._h.value if hasattr(rs._h, "value") else rs._h), ctypes.c_size_t(1), blocks, lens,
                                         ctypes.c_size_t(S), pars, sums_ptr)

    lib.gec_thread_link_release(cb, ctypes.c_void_p(7))
    assert encode(sums.ctypes.data_as(u8p)) == _lib.GEC_OK and fired == [7]
    assert encode(sums.ctypes.data_as(u8p)) == _lib.GEC_OK and fired == [7]  # one call per arming
    lib.gec_thread_link_release(cb, ctypes.c_void_p(8))
    assert encode(None) == _lib.GEC_E_INVALID_ARG and fired == [7, 8]  # a failing call still gives the turn back
    lib.gec_thread_link_release(cb, ctypes.c_void_p(9))
    lib.gec_thread_link_release(FN(), None)  # disarm
    assert encode(sums.ctypes.data_as(u8p)) == _lib.GEC_OK and fired == [7, 8]
    assert sums[:32].tobytes() == g.shardsum(data[:S].tobytes())


def test_a_call_that_cannot_get_memory_returns_nomem_and_the_codec_works_afterwards():
    """Nothing may unwind across the C ABI (the caller is Rust): every gec_* entry point that allocates is a function-try-block,
    and an item of the codec's fork-join pool that throws is carried to the calling thread instead of std::terminate on the worker
    or unwinding the caller's frame under the other workers.  tests/c/oom_probe.py starves one gec_encode_hash_batch of address
    space (RLIMIT_AS a few MiB above what the process holds) in a process of its own."""
    import subprocess
    import sys

    probe = os.path.join(ROOT, "tests", "c", "oom_probe.py")
    starved = 0
    for margin in (1, 2, 4, 8):
        r = subprocess.run([sys.executable, probe, str(margin)], capture_output=True, text=True, timeout=300)
        if r.returncode != 0:
            # glibc itself gives up when a thread's first touch of a thread-local block cannot be served; nothing of ours
            assert "cannot allocate memory for thread-local data" in r.stderr + r.stdout, r.stdout + r.stderr
            continue
        rc1, rc2, same = r.stdout.split()[:3]
        assert int(rc1) in (_lib.GEC_E_NOMEM, _lib.GEC_OK) and int(rc2) == _lib.GEC_OK and same == "1", r.stdout
        if int(rc1) == _lib.GEC_E_NOMEM:
            assert "memory" in r.stdout
            starved += 1
    assert starved >= 1


@pytest.mark.parametrize("seed", [3, 10, 27])
def test_arbitrary_arguments_come_back_with_a_code(seed):
    """tests/c/abi_fuzz.py: NULL pointers, S that is no multiple of 64, empty batches, blocks longer than k*S, fewer than k shards
    present ... through every host-pointer entry point of a CPU codec; the process must live to print its last line."""
    import subprocess
    import sys

    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "c", "abi_fuzz.py"), str(seed)], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0 and r.stdout.strip().splitlines()[-1].startswith("done"), r.stdout[-500:] + r.stderr[-2000:]
