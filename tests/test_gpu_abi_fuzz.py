"""The ABI fuzzers of tests/c on the HIP backend (each in a process of its own: a GPU memory fault must not take pytest with it).
Arguments are in contract as far as memory goes and otherwise arbitrary; every call must come back with a code, the device must
still answer afterwards, and -- for the manager -- every block must still read back (tests/c/*.py say what they throw)."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _run(script, *args):
    r = subprocess.run([sys.executable, os.path.join(HERE, "c", script), *map(str, args)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-500:] + r.stderr[-2000:]
    return r.stdout.strip().splitlines()[-1]


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [3, 10])
def test_host_pointer_entry_points(seed):
    assert _run("abi_fuzz.py", seed, "hip").startswith("done")


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1, 2])
def test_device_resident_entry_points(seed):
    assert _run("dev_abi_fuzz.py", seed, "hip").startswith("done 400")


@pytest.mark.gpu
@pytest.mark.parametrize("seed,ndev", [(1, 1), (7, 2)])
def test_block_manager_entry_points(seed, ndev):
    assert _run("bm_abi_fuzz.py", seed, "hip", ndev) == "done 400 unreadable 0"


def test_device_forms_of_a_cpu_codec():
    """the strided reconstruct forms work on host memory with a CPU codec; the others answer GEC_E_DEVICE"""
    assert _run("dev_abi_fuzz.py", 1, "cpu").startswith("done 400")
