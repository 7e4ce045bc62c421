"""tools/soak_manager.py as a test: random puts / gets (whole, streaming, ranged, raw, through the queue) / node outages /
shard damage / refcount drops / clock jumps against a model, with three resync workers and the ScrubWorker running in the
background, quiesce points where every live block must scrub clean and read back.  A short run per backend here; the
tool runs for as long as it is told to (profiles/r04_soak.txt)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import soak_manager  # noqa: E402


@pytest.mark.parametrize("seed", [7, 2026])
def test_soak_on_the_cpu_backend(seed, tmp_path):
    res = soak_manager.soak(6.0, "cpu", 120_000, seed, state_dir=str(tmp_path), verbose=False)
    assert res["ops"]["quiesce"] >= 2 and res["scrub_worker"]["errors"] == 0 and res["iterations"] > 300
    assert res["metrics"]["ec_reconstructs"] > 0 and res["metrics"]["resync_recv_counter"] > 0


@pytest.mark.gpu
def test_soak_on_the_hip_backend(tmp_path):
    res = soak_manager.soak(12.0, "hip", 1 << 20, 11, state_dir=str(tmp_path), verbose=False)
    assert res["ops"]["quiesce"] >= 2 and res["scrub_worker"]["errors"] == 0 and res["iterations"] > 300
    assert res["metrics"]["ec_reconstructs"] > 0 and res["metrics"]["resync_recv_counter"] > 0


def test_soak_over_directory_nodes_with_daemon_restarts(tmp_path):
    """Directory nodes (two devices): now and then the manager is destroyed and a new one opened over the same directories --
    references counted again from the model, the ScrubWorker carrying on from its record -- with the reader and writer threads
    picking up where they stood."""
    root = tmp_path / "nodes"
    root.mkdir()
    res = soak_manager.soak(10.0, "cpu", 80_000, 5, state_dir=str(tmp_path), node_dirs_root=str(root), ndev=2, verbose=False)
    assert res["ops"].get("restart", 0) >= 1 and res["ops"]["quiesce"] >= 3 and res["scrub_worker"]["errors"] == 0
    assert res["concurrent_readers"]["reads"] > 100 and res["concurrent_writers"]["order_violations"] == 0
