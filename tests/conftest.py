import os
import sys

# Before anything initialises the HIP runtime (the variable is read once, at initialisation; subprocesses inherit it): the tests move
# numpy arrays to the device and back with torch's plain `.to()` / `.cpu()`, i.e. SYNCHRONOUS copies from / to PAGEABLE memory.  For
# those the runtime's default is to lock the caller's pages for the device ("Locking to pool ... HostPtr = <heap address>") and let
# the DMA engine read them in place -- and about one full GPU suite run in eight died right there: "Memory access fault by GPU ... on
# address <the first page just locked>", one millisecond after the lock, with nothing of this repository between the two lines
# (profiles/r06_gpu_suite_abort.txt has the runtime's own log of it and the probes).  With a floor this high the runtime takes its
# other path -- the bytes travel through ITS pinned staging buffer, no caller page is ever mapped for the device -- which is also
# what the library under test does with pageable caller memory (ec_hip_host.cpp: its own pinned slots).  Value in KiB.
os.environ.setdefault("GPU_PINNED_MIN_XFER_SIZE", "1048576")

import pytest  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun)")
    # A fresh checkout has no built artefacts (they are git-ignored): build them once.
    # hipcc cross-compiles gfx950 without a GPU; the oracle needs only gcc.
    import subprocess

    need = [os.path.join(ROOT, "garage_amd", "libgarage_ec.so"), os.path.join(ROOT, "garage_amd", "libgarage_block.so"),
            os.path.join(ROOT, "oracle", "librs_oracle.so")]
    if not all(os.path.exists(p) for p in need):
        for d in (os.path.join(ROOT, "garage_amd", "csrc"), os.path.join(ROOT, "oracle")):
            r = subprocess.run(["make", "-j8", "-C", d], capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"make -C {d} failed:\n{r.stdout}\n{r.stderr}")


@pytest.fixture(autouse=True)
def _cyclic_garbage_is_collected_between_tests():
    """Codecs, groups and managers own device resources and threads, and Python frees the ones caught in reference cycles (an
    exception's traceback -> frame -> locals is the usual one: every `pytest.raises` makes it) whenever its collector happens to run
    -- which can be in the middle of ANOTHER test: a gbm_destroy / gec_codec_destroy (hipFree, hipStreamDestroy, thread joins) inside
    a stream capture invalidates the capture.  Collect after every test instead: whatever a test leaves behind is torn down before
    the next one starts.  (This was first taken for the cause of the abort that ended one GPU suite run in eight; it was not -- that
    one is the runtime's, see the top of this file.)"""
    yield
    import gc

    gc.collect()


@pytest.fixture(scope="session")
def coracle():
    from oracle.rs_oracle import COracle

    return COracle()
