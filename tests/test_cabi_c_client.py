"""Drives the plain-C client of the C ABI (tests/c/cabi_client.c): proof that a
compiled, non-Python host can use libgarage_ec.so through include/garage_ec.h
alone."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
EXE = os.path.join(HERE, "c", "cabi_client")


def _build():
    r = subprocess.run(["make", "-C", os.path.join(HERE, "c")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_c_client_host_logic():
    _build()
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr


def test_c_client_cpu_backend_and_threads():
    """The same sections on a GEC_BACKEND_CPU codec: RS(3,1) parity == XOR, verify, one-trip scrub / rebuild with their
    checksums, blake2sum KATs, 4 threads on one codec -- on a box without a GPU."""
    _build()
    r = subprocess.run([EXE, "cpu"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "CPU checks OK" in r.stdout


@pytest.mark.gpu
def test_c_client_gpu_and_threads():
    _build()
    r = subprocess.run([EXE, "gpu"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "GPU checks OK" in r.stdout
