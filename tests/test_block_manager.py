"""Host logic of the EC BlockManager mirror on CPU (arithmetic by the oracle stub)."""
import pytest

from tests import block_manager_cases as C
from tests.oracle_codec import OracleCodec


@pytest.fixture(params=[(3, 1), (10, 4)], ids=["rs3_1", "rs10_4"])
def codec(request):
    return OracleCodec(*request.param)


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


def test_datablock_api():
    C.scenario_datablock_api()


def test_needs_enough_nodes():
    from garage_amd.block_manager import BlockManager, Error, MemoryShardStore

    with pytest.raises(Error):
        BlockManager(OracleCodec(10, 4), [MemoryShardStore() for _ in range(13)])
