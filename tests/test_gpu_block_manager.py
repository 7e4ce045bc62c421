"""The same BlockManager scenarios with the real GPU codec behind the C ABI's
host-pointer entry points (gec_encode_batch / gec_reconstruct_batch / gec_verify_batch)."""
import pytest

pytestmark = pytest.mark.gpu

import garage_amd as g  # noqa: E402
from tests import block_manager_cases as C  # noqa: E402


@pytest.fixture(params=[(3, 1), (10, 4)], ids=["rs3_1", "rs10_4"])
def codec(request):
    return g.ReedSolomon(*request.param)


def test_put_get_roundtrip(codec, tmp_path):
    C.scenario_put_get_roundtrip(codec, tmp_path)


def test_survives_m_failures(codec):
    C.scenario_survives_m_failures(codec)


def test_write_quorum(codec):
    C.scenario_write_quorum(codec)


def test_corrupt_shard_detected_and_resynced(codec, tmp_path):
    C.scenario_corrupt_shard_detected_and_resynced(codec, tmp_path)


def test_scrub_finds_silent_corruption(codec):
    C.scenario_scrub_finds_silent_corruption(codec)


def test_geometry_is_a_function_of_the_block(codec):
    C.scenario_geometry_is_a_function_of_the_block(codec)


def test_put_with_node_down_is_repaired_not_deleted(codec):
    C.scenario_put_with_node_down_is_repaired_not_deleted(codec)
