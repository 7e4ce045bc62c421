"""ASan + UBSan builds of the product's host-side C++ logic (CPU only):
* garage_amd/csrc/gf256.hpp          -> tests/c/gf256_san_test.cpp
* garage_amd/csrc/block_manager.cpp  -> tests/c/block_manager_host_test.cpp, linked
  against an oracle-backed stand-in for the gec_* calls it makes (tests/c/gec_stub.cpp;
  the real libgarage_ec has no CPU path).
The reference gets memory safety from Rust; the C++ here does not, hence these."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CDIR = os.path.join(HERE, "c")


def _make(target):
    r = subprocess.run(["make", "-j8", "-C", CDIR, target], capture_output=True, text=True)
    if r.returncode != 0 and "fsanitize" in (r.stdout + r.stderr) and "cannot find" in (r.stdout + r.stderr):
        pytest.skip("sanitizer runtime not installed")
    assert r.returncode == 0, r.stdout + r.stderr


def test_gf256_under_asan_ubsan():
    _make("gf256_san_test")
    r = subprocess.run([os.path.join(CDIR, "gf256_san_test")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout + r.stderr


def test_block_manager_host_logic_under_asan_ubsan(tmp_path):
    _make("block_manager_host_test")
    r = subprocess.run([os.path.join(CDIR, "block_manager_host_test"), str(tmp_path)], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and "all scenarios OK" in r.stdout, r.stdout + r.stderr


def test_block_manager_batcher_under_tsan(tmp_path):
    """ThreadSanitizer build: 8 producer threads through the coalescing batcher plus 2
    concurrent readers."""
    _make("block_manager_host_test_tsan")
    r = subprocess.run([os.path.join(CDIR, "block_manager_host_test_tsan")], capture_output=True, text=True, timeout=900)
    if "FATAL: ThreadSanitizer: unexpected memory mapping" in r.stderr:
        pytest.skip("TSan cannot run in this container (ASLR/memory layout)")
    assert r.returncode == 0 and "all scenarios OK" in r.stdout, r.stdout + r.stderr


def test_manager_soak_under_tsan():
    """tools/soak_tsan.sh: the manager's soak (random walk + three resync workers + the ScrubWorker + reader and writer threads)
    over ThreadSanitizer builds of libgarage_block and of libgarage_ec's host path, loaded into python with libtsan preloaded.
    Passes when the soak passes AND ThreadSanitizer has nothing to report."""
    r = subprocess.run([os.path.join(HERE, "..", "tools", "soak_tsan.sh"), "8", "3"], capture_output=True, text=True, timeout=900)
    if r.returncode == 77:
        pytest.skip("TSan cannot run in this container")
    assert r.returncode == 0 and "thread sanitizer reports: 0" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


def test_manager_soak_under_asan_ubsan(tmp_path):
    """The same soak over AddressSanitizer + UBSan builds, on directory nodes with daemon restarts: a pinned buffer recycled
    while somebody still reads it, an overrun in a shard header, a destroyed manager's worker still running."""
    root = tmp_path / "nodes"
    root.mkdir()
    r = subprocess.run([os.path.join(HERE, "..", "tools", "soak_tsan.sh"), "8", "4", "2", str(root)], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, SAN="address"))
    if r.returncode == 77:
        pytest.skip("ASan cannot run in this container")
    assert r.returncode == 0 and "address sanitizer reports: 0" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
