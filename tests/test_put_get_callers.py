"""SURVEY.md section 8 row a10 -- the callers of the block path, replayed natively (tests/c/put_get_callers.cpp):
R PutObject requests, each submitting its blocks in order with <= PUT_BLOCKS_MAX_PARALLEL = 3 in flight and an
OrderTag per block (/root/reference/src/api/s3/put.rs:42,486-511), beside GetObject readers with a 2-deep prefetch
(src/api/s3/get.rs:429), and UploadPartCopy requests that stream their source blocks in through
gbm_rpc_get_block_streaming (two in flight, in order), re-encrypt them and put them untagged under their new names while the
next one is read (src/api/s3/copy.rs:520-551,606-630; src/api/s3/encryption.rs:269-280), through gbm_batcher_submit /
gbm_batcher_wait and gbm_rpc_get_block, and ranged GetObjects (body_from_blocks_range, src/api/s3/get.rs:650-743) through
gbm_rpc_get_block_range_streaming.  The harness asserts
coalescing (gbm_batcher_stats), zero gbm_node_order_violations, RAM-permit back-pressure and that every byte round-trips;
here it runs (1) under ThreadSanitizer over the product's CPU backend, (2) against the real libraries on the CPU
backend, (3) on the GPU with 1 MiB blocks -- and each of the three again over a MULTI-DEVICE manager (gbm_create_multi:
four CPU-backend codecs as four logical devices; two HIP codecs on device 0): one queue per device, per-device block counts
that follow gec_device_of_hash exactly, streams whose blocks cross devices still in order."""
import os
import re
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CDIR = os.path.join(HERE, "c")


def _make(target):
    r = subprocess.run(["make", "-j8", "-C", CDIR, target], capture_output=True, text=True)
    if r.returncode != 0 and "fsanitize" in (r.stdout + r.stderr) and "cannot find" in (r.stdout + r.stderr):
        pytest.skip("sanitizer runtime not installed")
    assert r.returncode == 0, r.stdout + r.stderr


def _check(out: str, backend: str, ndev: int = 1):
    assert "all bytes round-trip: OK" in out and "0 order violations" in out, out
    assert f"backend {backend}, {ndev} device(s)" in out, out
    m = re.search(r"(\d+) blocks in (\d+) device batches \(largest (\d+)\)", out)
    assert m, out
    blocks, batches, largest = map(int, m.groups())
    assert batches < blocks and largest >= 2, out   # concurrent requests shared device batches


def test_callers_under_tsan_on_the_cpu_backend():
    _make("put_get_callers_tsan")
    r = subprocess.run([os.path.join(CDIR, "put_get_callers_tsan"), "8", "9", "65536", "3"], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, GEC_CPU_THREADS="4"))
    if "FATAL: ThreadSanitizer: unexpected memory mapping" in r.stderr:
        pytest.skip("TSan cannot run in this container (ASLR/memory layout)")
    assert r.returncode == 0 and "ThreadSanitizer" not in r.stderr, r.stdout + r.stderr
    _check(r.stdout, "cpu")


def test_callers_under_tsan_multi_device_cpu_backend():
    """a TSan run of the caller pattern over the multi manager -- four logical devices."""
    _make("put_get_callers_tsan")
    r = subprocess.run([os.path.join(CDIR, "put_get_callers_tsan"), "8", "9", "65536", "3", "4"], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, GEC_CPU_THREADS="2"))
    if "FATAL: ThreadSanitizer: unexpected memory mapping" in r.stderr:
        pytest.skip("TSan cannot run in this container (ASLR/memory layout)")
    assert r.returncode == 0 and "ThreadSanitizer" not in r.stderr, r.stdout + r.stderr
    _check(r.stdout, "cpu", 4)


def test_callers_multi_device_real_libraries_cpu_backend():
    _make("put_get_callers")
    env = dict(os.environ, HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="", GEC_CPU_THREADS="2")
    r = subprocess.run([os.path.join(CDIR, "put_get_callers"), "12", "8", "262144", "3", "4"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    _check(r.stdout, "cpu", 4)


def test_callers_on_the_real_libraries_cpu_backend():
    _make("put_get_callers")
    env = dict(os.environ, HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="")   # GEC_BACKEND_AUTO -> the host cores
    r = subprocess.run([os.path.join(CDIR, "put_get_callers"), "12", "8", "262144", "3"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    _check(r.stdout, "cpu")


@pytest.mark.gpu
def test_callers_on_the_gpu_one_mib_blocks():
    """16 PutObjects x 12 blocks of 1 MiB (48 puts in flight: the batcher's design load) beside 4 GetObjects; one batch of a
    device's queue on the link at a time (its turn ends when its codec call returns)."""
    _make("put_get_callers")
    r = subprocess.run([os.path.join(CDIR, "put_get_callers"), "16", "12", "1048576", "4"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    _check(r.stdout, "hip")
    print(r.stdout)


@pytest.mark.gpu
def test_callers_on_the_gpu_two_codecs_one_device():
    """The multi-device manager on the one GPU a box has: two HIP codecs (two lanes, two queues) on device 0."""
    _make("put_get_callers")
    r = subprocess.run([os.path.join(CDIR, "put_get_callers"), "16", "12", "1048576", "4", "2"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    _check(r.stdout, "hip", 2)
    print(r.stdout)
