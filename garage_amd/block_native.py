"""ctypes binding of libgarage_block.so (include/garage_block.h): the C++
host-side mirror of garage_block::BlockManager with EC fan-out.  Method names
follow the reference (rpc_put_block, rpc_get_block, block_incref, ...)."""
from __future__ import annotations

import ctypes
import os
from typing import Optional, Sequence

from . import _lib  # noqa: F401  -- loads libgarage_ec.so first (single HIP runtime, see _lib.py)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgarage_block.so")

GBM_OK, GBM_E_MISSING_BLOCK, GBM_E_CORRUPT_DATA, GBM_E_QUORUM = 0, -1, -2, -3
GBM_E_INVALID_ARG, GBM_E_EC, GBM_E_IO, GBM_E_BUFFER_TOO_SMALL, GBM_E_ABORTED = -4, -5, -6, -7, -8
GBM_BLOCK_GC_DELAY_MS, GBM_RESYNC_RETRY_DELAY_MS = 600_000, 60_000

SYMBOLS = [
    "gbm_last_error", "gbm_blake2sum", "gbm_shardsum", "gbm_shardsum_v", "gbm_shard_version", "gbm_node_requests", "gbm_blake2sum_batch", "gbm_create", "gbm_destroy", "gbm_set_compression_level", "gbm_set_data_fsync",
    "gbm_set_verify_block_hash", "gbm_set_threads", "gbm_set_timing", "gbm_clock_advance",
    "gbm_storage_nodes_of", "gbm_layout_update", "gbm_layout_trim",
    "gbm_rpc_put_block", "gbm_rpc_put_blocks", "gbm_rpc_get_block", "gbm_rpc_get_blocks",
    "gbm_rpc_get_raw_block", "gbm_rpc_get_block_streaming", "gbm_rpc_get_raw_block_streaming",
    "gbm_block_incref", "gbm_block_decref", "gbm_block_rc",
    "gbm_put_to_resync", "gbm_resync_run", "gbm_resync_block", "gbm_resync_all", "gbm_resync_queue_len",
    "gbm_resync_errors_len", "gbm_resync_worker_start", "gbm_resync_worker_stop",
    "gbm_scrub", "gbm_scrub_all", "gbm_scrub_state", "gbm_repair_all", "gbm_node_set_down", "gbm_node_has_shard", "gbm_node_delete_shard",
    "gbm_node_corrupt_shard", "gbm_node_shard_header", "gbm_node_order_violations", "gbm_metrics", "gbm_gpu_hashed",
    "gbm_set_read_hedge", "gbm_hedged_reads", "gbm_node_set_latency", "gbm_set_host_block_hash_max",
    "gbm_batcher_create", "gbm_batcher_destroy", "gbm_batcher_put_block", "gbm_batcher_set_ram_buffer_max", "gbm_batcher_stats",
    "gbm_env_table", "gbm_set_tranquility", "gbm_tranquilized_ms", "gbm_background_codec",
    "gbm_batcher_submit", "gbm_batcher_wait", "gbm_set_maintenance_class", "gbm_batcher_get_block", "gbm_batcher_get_stats",
    "gbm_create_multi", "gbm_device_count", "gbm_device_of_hash", "gbm_device_codec", "gbm_device_background_codec",
    "gbm_device_metrics", "gbm_batcher_device_stats", "gbm_get_verify_block_hash", "gbm_rpc_get_block_range_streaming",
    "gbm_scrub_worker_start", "gbm_scrub_worker_stop", "gbm_scrub_worker_command", "gbm_scrub_worker_status",
    "gbm_block_metrics_get", "gbm_histogram_bounds", "gbm_metrics_prometheus", "gbm_list_resync_errors", "gbm_resync_clear_backoff",
    "gbm_zstd_encode", "gbm_zstd_decode", "gbm_set_resync_workers", "gbm_get_resync_workers", "gbm_resync_config_persist",
    "gbm_get_tranquility", "gbm_set_put_spot_check", "gbm_test_corrupt_put_sums", "gbm_set_migrate_on_read", "gbm_shards_migrated",
    "gbm_node_set_zone", "gbm_node_set_ping", "gbm_set_self_node", "gbm_block_read_order",
]


class OrderTag(ctypes.Structure):
    """OrderTag(stream, order), src/net/message.rs:66-89."""
    _fields_ = [("stream_id", ctypes.c_uint64), ("order", ctypes.c_uint64)]


class ScrubStatus(ctypes.Structure):
    """gbm_scrub_status: WorkerStatus of the ScrubWorker + ScrubWorkerPersisted (src/block/repair.rs:185-194,411-447)."""
    _fields_ = [("state", ctypes.c_int32), ("tranquility", ctypes.c_uint32), ("progress", ctypes.c_double),
                ("corruptions_detected", ctypes.c_uint64), ("time_last_complete_scrub_ms", ctypes.c_uint64),
                ("time_next_run_scrub_ms", ctypes.c_uint64), ("resume_at_ms", ctypes.c_uint64), ("blocks_scrubbed", ctypes.c_uint64),
                ("checkpoints_saved", ctypes.c_uint64), ("errors", ctypes.c_uint64)]


class ResyncErrorInfo(ctypes.Structure):
    """BlockResyncErrorInfo (src/block/manager.rs:105-111)."""
    _fields_ = [("hash", ctypes.c_uint8 * 32), ("refcount", ctypes.c_uint64), ("error_count", ctypes.c_uint64),
                ("last_try_ms", ctypes.c_uint64), ("next_try_ms", ctypes.c_uint64)]


HISTOGRAM_BUCKETS = 33


class Histogram(ctypes.Structure):
    """gbm_histogram: a value recorder over the reference exporter's boundaries (src/garage/server.rs:36-44), cumulative."""
    _fields_ = [("count", ctypes.c_uint64), ("sum_s", ctypes.c_double), ("bucket", ctypes.c_uint64 * (HISTOGRAM_BUCKETS + 1))]


class BlockMetrics(ctypes.Structure):
    """gbm_block_metrics: BlockManagerMetrics (src/block/metrics.rs:10-143) + this engine's own counters."""
    _fields_ = ([(n, ctypes.c_uint64) for n in ("compression_level", "rc_size", "resync_queue_length", "resync_errored_blocks", "ram_buffer_free_kb",
                                               "resync_counter", "resync_error_counter", "resync_send_counter", "resync_recv_counter",
                                               "bytes_read", "bytes_written", "delete_counter", "corruption_counter")]
                + [(n, Histogram) for n in ("resync_duration", "block_read_duration", "block_write_duration")]
                + [(n, ctypes.c_uint64) for n in ("ec_reconstructs", "blocks_put", "blocks_get", "gpu_hashed", "hedged_reads", "unconfirmed_verdicts",
                                                 "put_spot_checks", "put_spot_check_failures",
                                                 "scrub_corruptions_detected", "scrub_time_last_complete_ms", "tranquilized_ms",
                                                 "batcher_put_batches", "batcher_put_blocks", "batcher_get_batches", "batcher_get_blocks")]
                + [("devices", ctypes.c_uint32)])


SCRUB_START, SCRUB_PAUSE, SCRUB_RESUME, SCRUB_CANCEL = 0, 1, 2, 3            # ScrubWorkerCommand (repair.rs:300-305)
SCRUB_NO_WORKER, SCRUB_FINISHED, SCRUB_RUNNING, SCRUB_PAUSED = -1, 0, 1, 2   # ScrubWorkerState (:272-286)
SCRUB_INTERVAL_MS = 25 * 24 * 3600 * 1000


class DataBlockHeader(ctypes.Structure):
    """DataBlockHeader::{Plain, Compressed}, src/block/block.rs:12-22."""
    _fields_ = [("kind", ctypes.c_int)]
    PLAIN, COMPRESSED = 0, 1

    def is_compressed(self) -> bool:
        return self.kind == self.COMPRESSED


CHUNK_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint8), ctypes.c_size_t)


class BlockError(RuntimeError):
    def __init__(self, code: int, what: str, detail: str):
        super().__init__(f"{what}: {detail} (code {code})")
        self.code = code


class MissingBlock(BlockError):
    pass


class CorruptData(BlockError):
    pass


class Quorum(BlockError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: run `make -C garage_amd/csrc`")
    lib = ctypes.CDLL(LIB_PATH)
    vp, sz, ci = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
    u8p = ctypes.POINTER(ctypes.c_uint8)
    pp = ctypes.POINTER(ctypes.c_void_p)
    lib.gbm_last_error.restype = ctypes.c_char_p
    lib.gbm_blake2sum.argtypes = [ctypes.c_char_p, sz, ctypes.c_char_p]
    lib.gbm_blake2sum.restype = None
    lib.gbm_shardsum.argtypes = [ctypes.c_char_p, sz, ctypes.c_char_p]
    lib.gbm_shardsum.restype = None
    lib.gbm_shardsum_v.argtypes = [ctypes.c_int, ctypes.c_char_p, sz, ctypes.c_char_p]
    lib.gbm_shard_version.argtypes = [ctypes.c_void_p]
    lib.gbm_node_requests.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.gbm_node_requests.restype = ctypes.c_uint64
    lib.gbm_blake2sum_batch.argtypes = [sz, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_size_t), ctypes.c_char_p]
    lib.gbm_blake2sum_batch.restype = ctypes.c_int
    lib.gbm_create.argtypes = [vp, ci, ctypes.POINTER(ctypes.c_char_p), ci, pp]
    lib.gbm_destroy.argtypes = [vp]
    lib.gbm_destroy.restype = None
    lib.gbm_set_compression_level.argtypes = [vp, ci, ci]
    lib.gbm_set_data_fsync.argtypes = [vp, ci]
    lib.gbm_storage_nodes_of.argtypes = [vp, ctypes.c_char_p, ctypes.POINTER(ci)]
    tagp = ctypes.POINTER(OrderTag)
    hdrp = ctypes.POINTER(DataBlockHeader)
    lib.gbm_rpc_put_block.argtypes = [vp, ctypes.c_char_p, ctypes.c_char_p, sz, ci, tagp]
    lib.gbm_rpc_put_blocks.argtypes = [vp, sz, ctypes.c_char_p, pp, ctypes.POINTER(sz), ctypes.c_char_p, tagp]
    lib.gbm_rpc_get_block.argtypes = [vp, ctypes.c_char_p, tagp, vp, sz, ctypes.POINTER(sz)]
    lib.gbm_rpc_get_blocks.argtypes = [vp, sz, ctypes.c_char_p, tagp, pp, ctypes.POINTER(sz), ctypes.POINTER(sz), ctypes.POINTER(ci)]
    lib.gbm_rpc_get_raw_block.argtypes = [vp, ctypes.c_char_p, tagp, hdrp, vp, sz, ctypes.POINTER(sz)]
    lib.gbm_rpc_get_block_streaming.argtypes = [vp, ctypes.c_char_p, tagp, sz, CHUNK_FN, vp]
    lib.gbm_rpc_get_raw_block_streaming.argtypes = [vp, ctypes.c_char_p, tagp, hdrp, sz, CHUNK_FN, vp]
    lib.gbm_rpc_get_block_range_streaming.argtypes = [vp, ctypes.c_char_p, tagp, sz, sz, sz, sz, CHUNK_FN, vp]
    for f in ("gbm_block_incref", "gbm_block_decref"):
        getattr(lib, f).argtypes = [vp, ctypes.c_char_p]
    lib.gbm_block_rc.argtypes = [vp, ctypes.c_char_p, ctypes.POINTER(ctypes.c_uint64)]
    lib.gbm_set_verify_block_hash.argtypes = [vp, ci]
    lib.gbm_get_verify_block_hash.argtypes = [vp]
    lib.gbm_set_migrate_on_read.argtypes = [vp, ci]
    lib.gbm_shards_migrated.argtypes = [vp]
    lib.gbm_shards_migrated.restype = ctypes.c_uint64
    lib.gbm_node_set_zone.argtypes = [vp, ci, ci]
    lib.gbm_node_set_ping.argtypes = [vp, ci, ctypes.c_uint64]
    lib.gbm_set_self_node.argtypes = [vp, ci, ci]
    lib.gbm_block_read_order.argtypes = [vp, ctypes.c_char_p, sz, ctypes.POINTER(ci), ctypes.POINTER(ci), ctypes.POINTER(ci), ctypes.POINTER(sz)]
    lib.gbm_set_threads.argtypes = [vp, ci]
    lib.gbm_set_timing.argtypes = [vp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64]
    lib.gbm_clock_advance.argtypes = [vp, ctypes.c_uint64]
    lib.gbm_layout_update.argtypes = [vp]
    lib.gbm_layout_trim.argtypes = [vp]
    lib.gbm_put_to_resync.argtypes = [vp, ctypes.c_char_p, ctypes.c_uint64]
    lib.gbm_resync_run.argtypes = [vp, sz, ctypes.POINTER(ctypes.c_uint64)]
    lib.gbm_resync_errors_len.argtypes = [vp]
    lib.gbm_resync_errors_len.restype = sz
    lib.gbm_resync_worker_start.argtypes = [vp]
    lib.gbm_resync_worker_stop.argtypes = [vp]
    lib.gbm_node_shard_header.argtypes = [vp, ci, ctypes.c_char_p, ci, ctypes.c_char_p]
    lib.gbm_node_order_violations.argtypes = [vp, ci]
    lib.gbm_node_order_violations.restype = ctypes.c_uint64
    lib.gbm_set_read_hedge.argtypes = [vp, ctypes.c_uint64]
    lib.gbm_hedged_reads.argtypes = [vp]
    lib.gbm_hedged_reads.restype = ctypes.c_uint64
    lib.gbm_node_set_latency.argtypes = [vp, ci, ctypes.c_uint64]
    lib.gbm_set_host_block_hash_max.argtypes = [vp, sz]
    lib.gbm_resync_block.argtypes = [vp, ctypes.c_char_p, ctypes.POINTER(ci)]
    lib.gbm_resync_all.argtypes = [vp, ctypes.POINTER(ci)]
    lib.gbm_resync_queue_len.argtypes = [vp]
    lib.gbm_resync_queue_len.restype = sz
    lib.gbm_scrub.argtypes = [vp, sz, ctypes.c_char_p, u8p]
    lib.gbm_scrub_all.argtypes = [vp, sz, ctypes.POINTER(ctypes.c_uint64)]
    lib.gbm_scrub_state.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64)]
    lib.gbm_repair_all.argtypes = [vp, ctypes.POINTER(sz)]
    lib.gbm_scrub_worker_start.argtypes = [vp, ctypes.c_char_p, sz, ctypes.c_uint64]
    lib.gbm_scrub_worker_stop.argtypes = [vp]
    lib.gbm_scrub_worker_command.argtypes = [vp, ci, ctypes.c_uint64]
    lib.gbm_scrub_worker_status.argtypes = [vp, ctypes.POINTER(ScrubStatus)]
    lib.gbm_zstd_encode.argtypes = [ctypes.c_char_p, sz, ci, ctypes.c_char_p, sz, ctypes.POINTER(sz)]
    lib.gbm_zstd_decode.argtypes = [ctypes.c_char_p, sz, ctypes.c_char_p, sz, ctypes.POINTER(sz)]
    lib.gbm_set_put_spot_check.argtypes = [vp, ctypes.c_uint]
    lib.gbm_test_corrupt_put_sums.argtypes = [vp, ci]
    lib.gbm_get_tranquility.argtypes = [vp, ctypes.POINTER(ctypes.c_uint32)]
    lib.gbm_set_resync_workers.argtypes = [vp, ci]
    lib.gbm_get_resync_workers.argtypes = [vp]
    lib.gbm_resync_config_persist.argtypes = [vp, ctypes.c_char_p]
    lib.gbm_list_resync_errors.argtypes = [vp, ctypes.POINTER(ResyncErrorInfo), sz, ctypes.POINTER(sz)]
    lib.gbm_resync_clear_backoff.argtypes = [vp, ctypes.c_char_p]
    lib.gbm_block_metrics_get.argtypes = [vp, vp, ctypes.POINTER(BlockMetrics)]
    lib.gbm_histogram_bounds.argtypes = []
    lib.gbm_histogram_bounds.restype = ctypes.POINTER(ctypes.c_double)
    lib.gbm_metrics_prometheus.argtypes = [vp, vp, ctypes.c_char_p, sz, ctypes.POINTER(sz)]
    lib.gbm_node_set_down.argtypes = [vp, ci, ci]
    lib.gbm_node_has_shard.argtypes = [vp, ci, ctypes.c_char_p, ci]
    lib.gbm_node_delete_shard.argtypes = [vp, ci, ctypes.c_char_p, ci]
    lib.gbm_node_corrupt_shard.argtypes = [vp, ci, ctypes.c_char_p, ci, sz, ctypes.c_uint8, ci]
    lib.gbm_metrics.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64)]
    lib.gbm_batcher_create.argtypes = [vp, sz, ctypes.c_uint, pp]
    lib.gbm_batcher_destroy.argtypes = [vp]
    lib.gbm_batcher_destroy.restype = None
    lib.gbm_batcher_put_block.argtypes = [vp, ctypes.c_char_p, ctypes.c_char_p, sz, ci, tagp]
    lib.gbm_batcher_stats.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64)]
    lib.gbm_batcher_set_ram_buffer_max.argtypes = [vp, ctypes.c_size_t]
    lib.gbm_gpu_hashed.argtypes = [vp]
    lib.gbm_gpu_hashed.restype = ctypes.c_uint64
    lib.gbm_env_table.restype = ctypes.c_char_p
    lib.gbm_set_tranquility.argtypes = [vp, ci, ci]
    lib.gbm_tranquilized_ms.argtypes = [vp]
    lib.gbm_tranquilized_ms.restype = ctypes.c_uint64
    lib.gbm_background_codec.argtypes = [vp]
    lib.gbm_background_codec.restype = vp
    lib.gbm_batcher_submit.argtypes = [vp, ctypes.c_char_p, ctypes.c_char_p, sz, ci, tagp, pp]
    lib.gbm_batcher_wait.argtypes = [vp]
    lib.gbm_batcher_get_block.argtypes = [vp, ctypes.c_char_p, vp, sz, ctypes.POINTER(ctypes.c_size_t)]
    lib.gbm_batcher_get_stats.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64)]
    lib.gbm_set_maintenance_class.argtypes = [vp, ci]
    lib.gbm_create_multi.argtypes = [pp, ci, ci, ctypes.POINTER(ctypes.c_char_p), ci, pp]
    lib.gbm_device_count.argtypes = [vp]
    lib.gbm_device_of_hash.argtypes = [vp, ctypes.c_char_p]
    for f in ("gbm_device_codec", "gbm_device_background_codec"):
        getattr(lib, f).argtypes = [vp, ci]
        getattr(lib, f).restype = vp
    lib.gbm_device_metrics.argtypes = [vp, ci, ctypes.POINTER(ctypes.c_uint64)]
    lib.gbm_batcher_device_stats.argtypes = [vp, ci, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]
    return lib


lib = _load()


def _check(rc: int, what: str) -> None:
    if rc == GBM_OK:
        return
    detail = (lib.gbm_last_error() or b"").decode("utf-8", "replace")
    cls = {GBM_E_MISSING_BLOCK: MissingBlock, GBM_E_CORRUPT_DATA: CorruptData, GBM_E_QUORUM: Quorum}.get(rc, BlockError)
    raise cls(rc, what, detail)


def blake2sum(data: bytes) -> bytes:
    out = ctypes.create_string_buffer(32)
    lib.gbm_blake2sum(data, len(data), out)
    return out.raw


def blake2sum_batch(blocks) -> list:
    """Content hashes of several blocks on one host core (eight at a time in AVX-512 lanes where there are any)."""
    n = len(blocks)
    ptrs = (ctypes.c_char_p * n)(*blocks)
    lens = (ctypes.c_size_t * n)(*[len(b) for b in blocks])
    out = ctypes.create_string_buffer(32 * n)
    _check(lib.gbm_blake2sum_batch(n, ptrs, lens, out), "gbm_blake2sum_batch")
    return [out.raw[32 * i:32 * i + 32] for i in range(n)]


def zstd_encode(data: bytes, level: int = 1) -> bytes:
    """garage_block::zstd_encode (src/block/block.rs:99-106): one frame with its content checksum."""
    n = ctypes.c_size_t()
    buf = ctypes.create_string_buffer(len(data) + len(data) // 128 + 1024)
    _check(lib.gbm_zstd_encode(data, len(data), level, buf, len(buf), ctypes.byref(n)), "gbm_zstd_encode")
    return buf.raw[:n.value]


def zstd_decode(frame: bytes, max_len: int = 1 << 26) -> bytes:
    n = ctypes.c_size_t()
    buf = ctypes.create_string_buffer(max(1, max_len))
    _check(lib.gbm_zstd_decode(frame, len(frame), buf, max_len, ctypes.byref(n)), "gbm_zstd_decode")
    return buf.raw[:n.value]


def shardsum(data: bytes, version: int = 3) -> bytes:
    """The checksum a shard header of `version` carries (3 = MLH64, the default of every codec; 2 = BLAKE2b tree mode;
    1 = plain blake2sum), computed by libgarage_block on the host (gbm_shardsum_v)."""
    out = ctypes.create_string_buffer(32)
    _check(lib.gbm_shardsum_v(version, data, len(data), out), "gbm_shardsum_v")
    return out.raw


class NativeBlockManager:
    """garage_block::BlockManager mirror (C++), `codec` is a garage_amd.ReedSolomon -- or a list of them, one per
    device: the manager then routes every block to the device gec_device_of_hash(hash, ndev) names (gbm_create_multi)."""

    METRICS = ("bytes_written", "bytes_read", "corruption_counter", "ec_reconstructs", "blocks_put", "blocks_get")

    def __init__(self, codec, nnodes: int, node_dirs: Optional[Sequence[str]] = None, write_quorum: int = 0,
                 compression_level: Optional[int] = None, data_fsync: bool = False):
        self.codecs = list(codec) if isinstance(codec, (list, tuple)) else None  # keep the borrowed codecs alive
        self.codec = codec = self.codecs[0] if self.codecs else codec
        self.k, self.m, self.n, self.nnodes = codec.k, codec.m, codec.k + codec.m, nnodes
        dirs = None
        if node_dirs is not None:
            assert len(node_dirs) == nnodes
            dirs = (ctypes.c_char_p * nnodes)(*[d.encode() for d in node_dirs])
        h = ctypes.c_void_p()
        if self.codecs is not None:
            arr = (ctypes.c_void_p * len(self.codecs))(*[c._h.value if isinstance(c._h, ctypes.c_void_p) else c._h for c in self.codecs])
            _check(lib.gbm_create_multi(arr, len(self.codecs), nnodes, dirs, write_quorum, ctypes.byref(h)), "gbm_create_multi")
        else:
            _check(lib.gbm_create(codec._h, nnodes, dirs, write_quorum, ctypes.byref(h)), "gbm_create")
        self._h = h
        if compression_level is not None:
            _check(lib.gbm_set_compression_level(self._h, 1, compression_level), "gbm_set_compression_level")
        if data_fsync:
            _check(lib.gbm_set_data_fsync(self._h, 1), "gbm_set_data_fsync")

    def close(self):
        if getattr(self, "_h", None) and lib is not None:   # (lib is None while the interpreter shuts down)
            lib.gbm_destroy(self._h)
            self._h = None

    __del__ = close

    @property
    def shard_version(self) -> int:
        """the shard-header version this manager writes (gbm_shard_version): its codec's checksum kind"""
        return int(lib.gbm_shard_version(self._h))

    def node_requests(self, node: int) -> int:
        """requests of any kind the node has been handed so far (test hook)"""
        return int(lib.gbm_node_requests(self._h, node))

    def storage_nodes_of(self, hash_: bytes) -> list[int]:
        out = (ctypes.c_int * self.n)()
        _check(lib.gbm_storage_nodes_of(self._h, hash_, out), "gbm_storage_nodes_of")
        return list(out)

    @staticmethod
    def _tag(order_tag):
        """order_tag: None, an OrderTag, or a (stream_id, order) pair."""
        if order_tag is None:
            return None
        if isinstance(order_tag, OrderTag):
            return ctypes.byref(order_tag)
        return ctypes.byref(OrderTag(int(order_tag[0]), int(order_tag[1])))

    def rpc_put_block(self, hash_: bytes, data: bytes, prevent_compression: bool = False, order_tag=None) -> None:
        """BlockManager::rpc_put_block(hash, data, prevent_compression, order_tag), src/block/manager.rs:366-408."""
        _check(lib.gbm_rpc_put_block(self._h, hash_, data, len(data), int(bool(prevent_compression)), self._tag(order_tag)),
               "rpc_put_block")

    def rpc_put_blocks(self, items: Sequence[tuple[bytes, bytes]], prevent_compression: Optional[Sequence[bool]] = None,
                       order_tags: Optional[Sequence[tuple[int, int]]] = None) -> None:
        n = len(items)
        hashes = b"".join(h for h, _ in items)
        # bytes objects are immutable and stay alive in `items`: hand their buffers over without a copy
        ptrs = (ctypes.c_void_p * n)(*[ctypes.cast(ctypes.c_char_p(d), ctypes.c_void_p).value for _, d in items])
        lens = (ctypes.c_size_t * n)(*[len(d) for _, d in items])
        pc = bytes(int(bool(x)) for x in prevent_compression) if prevent_compression is not None else None
        tags = (OrderTag * n)(*[OrderTag(int(s), int(o)) for s, o in order_tags]) if order_tags is not None else None
        _check(lib.gbm_rpc_put_blocks(self._h, n, hashes, ptrs, lens, pc, tags), "rpc_put_blocks")

    def rpc_get_block(self, hash_: bytes, max_len: int = 1 << 26, order_tag=None) -> bytes:
        ln = ctypes.c_size_t()
        buf = ctypes.create_string_buffer(min(max_len, 1 << 22))
        rc = lib.gbm_rpc_get_block(self._h, hash_, self._tag(order_tag), buf, len(buf), ctypes.byref(ln))
        if rc == GBM_E_BUFFER_TOO_SMALL and ln.value <= max_len:
            buf = ctypes.create_string_buffer(ln.value)
            rc = lib.gbm_rpc_get_block(self._h, hash_, self._tag(order_tag), buf, len(buf), ctypes.byref(ln))
        _check(rc, "rpc_get_block")
        return buf.raw[: ln.value]

    def rpc_get_raw_block(self, hash_: bytes, max_len: int = 1 << 26, order_tag=None) -> tuple[DataBlockHeader, bytes]:
        """BlockManager::rpc_get_raw_block: (DataBlockHeader, stored bytes) -- not decompressed."""
        ln = ctypes.c_size_t()
        hdr = DataBlockHeader()
        buf = ctypes.create_string_buffer(min(max_len, 1 << 22))
        rc = lib.gbm_rpc_get_raw_block(self._h, hash_, self._tag(order_tag), ctypes.byref(hdr), buf, len(buf), ctypes.byref(ln))
        if rc == GBM_E_BUFFER_TOO_SMALL and ln.value <= max_len:
            buf = ctypes.create_string_buffer(ln.value)
            rc = lib.gbm_rpc_get_raw_block(self._h, hash_, self._tag(order_tag), ctypes.byref(hdr), buf, len(buf), ctypes.byref(ln))
        _check(rc, "rpc_get_raw_block")
        return hdr, buf.raw[: ln.value]

    def rpc_get_block_streaming(self, hash_: bytes, order_tag=None, chunk_bytes: int = 0, raw: bool = False):
        """Generator-like: returns the list of chunks the sink received (rpc_get_block_streaming,
        or rpc_get_raw_block_streaming with raw=True, in which case (header, chunks))."""
        chunks: list[bytes] = []

        def sink(_ctx, p, n):
            chunks.append(ctypes.string_at(p, n))
            return 0

        cb = CHUNK_FN(sink)
        if raw:
            hdr = DataBlockHeader()
            _check(lib.gbm_rpc_get_raw_block_streaming(self._h, hash_, self._tag(order_tag), ctypes.byref(hdr), chunk_bytes, cb, None),
                   "rpc_get_raw_block_streaming")
            return hdr, chunks
        _check(lib.gbm_rpc_get_block_streaming(self._h, hash_, self._tag(order_tag), chunk_bytes, cb, None), "rpc_get_block_streaming")
        return chunks

    def rpc_get_block_range(self, hash_: bytes, block_size: int, begin: int, end: int, order_tag=None, chunk_bytes: int = 0) -> list:
        """Bytes [begin, end) of one block, as the chunks the sink received (body_from_blocks_range,
        src/api/s3/get.rs:650-743): only the data shards the range touches are read when the block is stored Plain."""
        chunks: list[bytes] = []

        def sink(_ctx, p, n):
            chunks.append(ctypes.string_at(p, n))
            return 0

        cb = CHUNK_FN(sink)
        _check(lib.gbm_rpc_get_block_range_streaming(self._h, hash_, self._tag(order_tag), block_size, begin, end, chunk_bytes, cb, None),
               "rpc_get_block_range_streaming")
        return chunks

    def rpc_get_blocks(self, hashes: Sequence[bytes], max_len: int, out: Optional[list] = None) -> list:
        """Batched get: returns bytes per block, or the error code on failure.  `out` (optional): a list of
        writable buffers (e.g. numpy arrays over pinned memory) that receive the blocks; then the result
        holds the lengths instead of copies."""
        n = len(hashes)
        bufs = out if out is not None else [ctypes.create_string_buffer(max_len) for _ in range(n)]
        if out is not None:
            ptrs = (ctypes.c_void_p * n)(*[b.ctypes.data for b in bufs])
        else:
            ptrs = (ctypes.c_void_p * n)(*[ctypes.addressof(b) for b in bufs])
        caps = (ctypes.c_size_t * n)(*[max_len] * n)
        lens = (ctypes.c_size_t * n)()
        rcs = (ctypes.c_int * n)()
        _check(lib.gbm_rpc_get_blocks(self._h, n, b"".join(hashes), None, ptrs, caps, lens, rcs), "rpc_get_blocks")
        if out is not None:
            return [int(lens[i]) if rcs[i] == 0 else rcs[i] for i in range(n)]
        return [bufs[i].raw[: lens[i]] if rcs[i] == 0 else rcs[i] for i in range(n)]

    def block_incref(self, hash_: bytes) -> None:
        _check(lib.gbm_block_incref(self._h, hash_), "block_incref")

    def block_decref(self, hash_: bytes) -> None:
        _check(lib.gbm_block_decref(self._h, hash_), "block_decref")

    def resync_block(self, hash_: bytes) -> int:
        c = ctypes.c_int()
        _check(lib.gbm_resync_block(self._h, hash_, ctypes.byref(c)), "resync_block")
        return c.value

    def resync_all(self) -> int:
        c = ctypes.c_int()
        _check(lib.gbm_resync_all(self._h, ctypes.byref(c)), "resync_all")
        return c.value

    def resync_queue_len(self) -> int:
        return int(lib.gbm_resync_queue_len(self._h))

    def resync_errors_len(self) -> int:
        return int(lib.gbm_resync_errors_len(self._h))

    RESYNC_STATS = ("taken", "ok", "errors", "skipped", "rebuilt", "deleted", "offloaded", "device_calls")

    def resync_run(self, max_blocks: int = 0, check: bool = True) -> dict:
        """One pass of the resync queue over everything that is due."""
        st = (ctypes.c_uint64 * 8)()
        rc = lib.gbm_resync_run(self._h, max_blocks, st)
        if check:
            _check(rc, "resync_run")
        d = dict(zip(self.RESYNC_STATS, [int(x) for x in st]))
        d["rc"] = rc
        return d

    def set_put_spot_check(self, every_n: int) -> None:
        """Every Nth put trip one device-computed shard checksum is re-computed on the host before anything is sent (0 = never)."""
        _check(lib.gbm_set_put_spot_check(self._h, every_n), "set_put_spot_check")

    def get_tranquility(self) -> tuple[int, int]:
        out = (ctypes.c_uint32 * 2)()
        _check(lib.gbm_get_tranquility(self._h, out), "get_tranquility")
        return int(out[0]), int(out[1])

    def resync_worker_start(self) -> None:
        _check(lib.gbm_resync_worker_start(self._h), "resync_worker_start")

    def resync_worker_stop(self) -> None:
        _check(lib.gbm_resync_worker_stop(self._h), "resync_worker_stop")

    def set_resync_workers(self, n: int) -> None:
        """`resync-worker-count` (src/block/resync.rs:136-152): 1..MAX_RESYNC_WORKERS."""
        _check(lib.gbm_set_resync_workers(self._h, n), "set_resync_workers")

    @property
    def resync_workers(self) -> int:
        return int(lib.gbm_get_resync_workers(self._h))

    def resync_config_persist(self, path: str) -> None:
        """ResyncPersistedConfig (resync.rs:58-71): worker count + tranquility, loaded from / saved to `path`."""
        _check(lib.gbm_resync_config_persist(self._h, path.encode()), "resync_config_persist")

    def list_resync_errors(self) -> list[dict]:
        """BlockManager::list_resync_errors (`garage block list-errors`)."""
        n = ctypes.c_size_t()
        _check(lib.gbm_list_resync_errors(self._h, None, 0, ctypes.byref(n)), "list_resync_errors")
        arr = (ResyncErrorInfo * max(1, n.value))()
        _check(lib.gbm_list_resync_errors(self._h, arr, n.value, ctypes.byref(n)), "list_resync_errors")
        return [{"hash": bytes(e.hash), "refcount": int(e.refcount), "error_count": int(e.error_count), "last_try_ms": int(e.last_try_ms),
                 "next_try_ms": int(e.next_try_ms)} for e in arr[:min(n.value, len(arr))]]

    def resync_clear_backoff(self, hash_: bytes) -> None:
        """BlockResyncManager::clear_backoff (`garage block retry-now`)."""
        _check(lib.gbm_resync_clear_backoff(self._h, hash_), "resync_clear_backoff")

    def put_to_resync(self, hash_: bytes, delay_ms: int = 0) -> None:
        _check(lib.gbm_put_to_resync(self._h, hash_, delay_ms), "put_to_resync")

    def block_rc(self, hash_: bytes) -> tuple[int, str, int]:
        out = (ctypes.c_uint64 * 3)()
        _check(lib.gbm_block_rc(self._h, hash_, out), "block_rc")
        return int(out[0]), ("Absent", "Present", "Deletable")[int(out[1])], int(out[2])

    def set_timing(self, gc_delay_ms: int = -1, resync_retry_delay_ms: int = -1, incref_check_delay_ms: int = -1) -> None:
        _check(lib.gbm_set_timing(self._h, gc_delay_ms, resync_retry_delay_ms, incref_check_delay_ms), "set_timing")

    def clock_advance(self, ms: int) -> None:
        _check(lib.gbm_clock_advance(self._h, ms), "clock_advance")

    VERIFY_MODES = {"off": 0, "always": 1, "rebuilt": 2, "rebuilt-only": 2}

    def set_verify_block_hash(self, mode) -> None:
        """The requester's end-to-end block hash: "off", "rebuilt" (only blocks that went through a decode; the default over
        BLAKE2b-tree shard checksums), "always" (the default over MLH64 shard checksums, header version 3); True / False mean
        "always" / "off".  Shard checksums are checked in every mode."""
        if isinstance(mode, str):
            mode = self.VERIFY_MODES[mode]
        _check(lib.gbm_set_verify_block_hash(self._h, int(mode)), "set_verify_block_hash")

    @property
    def verify_block_hash(self) -> str:
        return {0: "off", 1: "always", 2: "rebuilt"}[int(lib.gbm_get_verify_block_hash(self._h))]

    # ---- request_order (rpc_helper.rs:621-660) applied to the holders of a block's shards
    def node_set_zone(self, node: int, zone: int) -> None:
        _check(lib.gbm_node_set_zone(self._h, node, zone), "node_set_zone")

    def node_set_ping(self, node: int, ping_us: int) -> None:
        _check(lib.gbm_node_set_ping(self._h, node, ping_us), "node_set_ping")

    def set_self_node(self, node: int, zone: int = 0) -> None:
        """who is asking: one of the storage nodes (or -1) and its zone"""
        _check(lib.gbm_set_self_node(self._h, node, zone), "set_self_node")

    def block_read_order(self, hash_: bytes) -> list[tuple[int, int, int]]:
        """[(node, shard index, layout version)] in the order a read of this block asks its holders"""
        cnt = ctypes.c_size_t()
        cap = 64 * self.n
        nodes, shards, vers = (ctypes.c_int * cap)(), (ctypes.c_int * cap)(), (ctypes.c_int * cap)()
        _check(lib.gbm_block_read_order(self._h, hash_, cap, nodes, shards, vers, ctypes.byref(cnt)), "block_read_order")
        return [(nodes[i], shards[i], vers[i]) for i in range(min(cap, cnt.value))]

    def set_migrate_on_read(self, enabled: bool) -> None:
        """Reads also REWRITE shards of another header version in this manager's (default off: only scrub and resync do)."""
        _check(lib.gbm_set_migrate_on_read(self._h, int(bool(enabled))), "set_migrate_on_read")

    @property
    def shards_migrated(self) -> int:
        return int(lib.gbm_shards_migrated(self._h))

    def set_threads(self, n: int) -> None:
        _check(lib.gbm_set_threads(self._h, n), "set_threads")

    def layout_update(self) -> int:
        return int(lib.gbm_layout_update(self._h))

    def layout_trim(self) -> None:
        _check(lib.gbm_layout_trim(self._h), "layout_trim")

    def node_shard_header(self, node: int, hash_: bytes, idx: int) -> bytes:
        out = ctypes.create_string_buffer(64)
        _check(lib.gbm_node_shard_header(self._h, node, hash_, idx, out), "node_shard_header")
        return out.raw

    def node_order_violations(self, node: int) -> int:
        return int(lib.gbm_node_order_violations(self._h, node))

    def set_host_block_hash_max(self, nblocks: int) -> None:
        """Gets of up to this many blocks verify the block hash on the host pool (0 = always on the device)."""
        _check(lib.gbm_set_host_block_hash_max(self._h, nblocks), "set_host_block_hash_max")

    def set_read_hedge(self, hedge_us: int) -> None:
        """0 = ask the first k holders and go further only on failure; >0 = ask the next holders too when some
        have not answered after hedge_us."""
        _check(lib.gbm_set_read_hedge(self._h, hedge_us), "set_read_hedge")

    @property
    def hedged_reads(self) -> int:
        return int(lib.gbm_hedged_reads(self._h))

    def node_set_latency(self, node: int, latency_us: int) -> None:
        _check(lib.gbm_node_set_latency(self._h, node, latency_us), "node_set_latency")

    def scrub(self, hashes: Sequence[bytes]) -> list[bytes]:
        n = len(hashes)
        bad = (ctypes.c_uint8 * n)()
        _check(lib.gbm_scrub(self._h, n, b"".join(hashes), bad), "scrub")
        return [h for h, b in zip(hashes, bad) if b]

    def scrub_all(self, batch_blocks: int = 0) -> dict:
        """ScrubWorker over everything stored: {scrubbed, corruptions, device_calls, located}."""
        st = (ctypes.c_uint64 * 4)()
        _check(lib.gbm_scrub_all(self._h, batch_blocks, st), "scrub_all")
        return dict(zip(("scrubbed", "corruptions", "device_calls", "located"), [int(x) for x in st]))

    def scrub_state(self) -> tuple[int, int]:
        out = (ctypes.c_uint64 * 2)()
        _check(lib.gbm_scrub_state(self._h, out), "scrub_state")
        return int(out[0]), int(out[1])

    def set_tranquility(self, scrub: int = -1, resync: int = -1) -> None:
        """scrub-tranquility / resync-tranquility (src/block/repair.rs:26-27, resync.rs:46); -1 leaves a value as it is."""
        _check(lib.gbm_set_tranquility(self._h, scrub, resync), "set_tranquility")

    def scrub_worker_start(self, persist_path: str | None = None, batch_blocks: int = 0, checkpoint_interval_ms: int = 0) -> None:
        """The continuously running ScrubWorker (src/block/repair.rs:156-500); its state file survives a restart."""
        _check(lib.gbm_scrub_worker_start(self._h, persist_path.encode() if persist_path else None, batch_blocks, checkpoint_interval_ms),
               "scrub_worker_start")

    def scrub_worker_stop(self) -> None:
        _check(lib.gbm_scrub_worker_stop(self._h), "scrub_worker_stop")

    def scrub_worker_command(self, cmd: int, pause_ms: int = 0) -> None:
        """ScrubWorkerCommand::{Start, Pause(d), Resume, Cancel}; a command that does not fit the state raises."""
        _check(lib.gbm_scrub_worker_command(self._h, cmd, pause_ms), "scrub_worker_command")

    def scrub_worker_status(self) -> dict:
        st = ScrubStatus()
        _check(lib.gbm_scrub_worker_status(self._h, ctypes.byref(st)), "scrub_worker_status")
        return {name: getattr(st, name) for name, _ in ScrubStatus._fields_}

    def block_metrics(self, batcher=None) -> dict:
        """BlockManagerMetrics (src/block/metrics.rs) as a dict; histograms as {count, sum_s, bucket: [cumulative...]}."""
        x = BlockMetrics()
        _check(lib.gbm_block_metrics_get(self._h, batcher._h if batcher is not None else None, ctypes.byref(x)), "block_metrics_get")
        out = {}
        for name, ty in BlockMetrics._fields_:
            v = getattr(x, name)
            out[name] = {"count": int(v.count), "sum_s": float(v.sum_s), "bucket": [int(c) for c in v.bucket]} if ty is Histogram else int(v)
        return out

    def metrics_prometheus(self, batcher=None) -> str:
        """The same as Prometheus text exposition (the admin API's /metrics)."""
        bh = batcher._h if batcher is not None else None
        need = ctypes.c_size_t()
        rc = lib.gbm_metrics_prometheus(self._h, bh, None, 0, ctypes.byref(need))
        assert rc == GBM_E_BUFFER_TOO_SMALL, rc
        buf = ctypes.create_string_buffer(need.value + 4096)   # (the counters may have grown a digit since)
        _check(lib.gbm_metrics_prometheus(self._h, bh, buf, len(buf), ctypes.byref(need)), "metrics_prometheus")
        return buf.value.decode()

    def repair_all(self) -> int:
        n = ctypes.c_size_t()
        _check(lib.gbm_repair_all(self._h, ctypes.byref(n)), "repair_all")
        return int(n.value)

    # fault injection -------------------------------------------------------------
    def node_set_down(self, node: int, down: bool) -> None:
        _check(lib.gbm_node_set_down(self._h, node, int(down)), "node_set_down")

    def node_has_shard(self, node: int, hash_: bytes, idx: int) -> bool:
        return bool(lib.gbm_node_has_shard(self._h, node, hash_, idx))

    def node_delete_shard(self, node: int, hash_: bytes, idx: int) -> None:
        _check(lib.gbm_node_delete_shard(self._h, node, hash_, idx), "node_delete_shard")

    def node_corrupt_shard(self, node: int, hash_: bytes, idx: int, offset: int, mask: int = 1, fix_checksum: bool = False):
        _check(lib.gbm_node_corrupt_shard(self._h, node, hash_, idx, offset, mask, int(fix_checksum)), "node_corrupt_shard")

    def gpu_hashed(self) -> int:
        return int(lib.gbm_gpu_hashed(self._h))

    @property
    def metrics(self) -> dict:
        out = (ctypes.c_uint64 * 6)()
        _check(lib.gbm_metrics(self._h, out), "gbm_metrics")
        return dict(zip(self.METRICS, [int(x) for x in out]))

    # several devices -------------------------------------------------------------
    @property
    def device_count(self) -> int:
        return int(lib.gbm_device_count(self._h))

    def device_of_hash(self, hash_: bytes) -> int:
        return int(lib.gbm_device_of_hash(self._h, hash_))

    def device_metrics(self, dev: int) -> dict:
        out = (ctypes.c_uint64 * 6)()
        _check(lib.gbm_device_metrics(self._h, dev, out), "gbm_device_metrics")
        return dict(zip(self.METRICS, [int(x) for x in out]))


class Batcher:
    """gbm_batcher: thread-safe put_block() calls coalesced into device batches."""

    def __init__(self, manager: NativeBlockManager, max_blocks: int = 64, max_wait_us: int = 200,
                 ram_buffer_max: Optional[int] = None):
        self.manager = manager  # keep alive
        h = ctypes.c_void_p()
        _check(lib.gbm_batcher_create(manager._h, max_blocks, max_wait_us, ctypes.byref(h)), "gbm_batcher_create")
        self._h = h
        if ram_buffer_max is not None:  # Config.block_ram_buffer_max (default 256 MiB)
            _check(lib.gbm_batcher_set_ram_buffer_max(self._h, ram_buffer_max), "gbm_batcher_set_ram_buffer_max")

    def close(self):
        if getattr(self, "_h", None):
            lib.gbm_batcher_destroy(self._h)
            self._h = None

    __del__ = close

    def put_block(self, hash_: bytes, data: bytes, prevent_compression: bool = False, order_tag=None) -> None:
        """Blocks until the batch containing this block is stored (ctypes drops the GIL)."""
        _check(lib.gbm_batcher_put_block(self._h, hash_, data, len(data), int(bool(prevent_compression)),
                                         NativeBlockManager._tag(order_tag)), "batcher.put_block")

    def submit(self, hash_: bytes, data: bytes, prevent_compression: bool = False, order_tag=None):
        """rpc_put_block as a future: queues the block and returns a ticket at once; `wait(ticket)` blocks until the
        block's batch has been fanned out.  `hash_` and `data` must stay alive until then (the ticket keeps them)."""
        tk = ctypes.c_void_p()
        _check(lib.gbm_batcher_submit(self._h, hash_, data, len(data), int(bool(prevent_compression)),
                                      NativeBlockManager._tag(order_tag), ctypes.byref(tk)), "batcher.submit")
        return (tk, hash_, data)

    def wait(self, ticket) -> None:
        _check(lib.gbm_batcher_wait(ticket[0]), "batcher.wait")

    def stats(self) -> dict:
        out = (ctypes.c_uint64 * 3)()
        _check(lib.gbm_batcher_stats(self._h, out), "gbm_batcher_stats")
        return {"batches": int(out[0]), "blocks": int(out[1]), "max_batch": int(out[2])}

    def get_block(self, hash_: bytes, max_len: int) -> bytes:
        """The read side: blocks until the batch of concurrent gets containing this one has been fetched and checked."""
        buf = ctypes.create_string_buffer(max(max_len, 1))
        n = ctypes.c_size_t()
        _check(lib.gbm_batcher_get_block(self._h, hash_, buf, max_len, ctypes.byref(n)), "batcher.get_block")
        return buf.raw[: n.value]

    def get_stats(self) -> dict:
        out = (ctypes.c_uint64 * 3)()
        _check(lib.gbm_batcher_get_stats(self._h, out), "gbm_batcher_get_stats")
        return {"batches": int(out[0]), "blocks": int(out[1]), "max_batch": int(out[2])}

    def device_stats(self, dev: int) -> dict:
        """One device's queue of a multi-device manager's batcher: {"put": {...}, "get": {...}}."""
        p, g = (ctypes.c_uint64 * 3)(), (ctypes.c_uint64 * 3)()
        _check(lib.gbm_batcher_device_stats(self._h, dev, p, g), "gbm_batcher_device_stats")
        f = lambda o: {"batches": int(o[0]), "blocks": int(o[1]), "max_batch": int(o[2])}  # noqa: E731
        return {"put": f(p), "get": f(g)}
