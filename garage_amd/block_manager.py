"""Host-side mirror of Garage's block store surface with erasure-coded fan-out
(SURVEY.md section 8, rows f1-f3).

Same names, argument meaning and error behaviour as the reference so tests read
like Garage's:

* ``DataBlockHeader`` / ``DataBlock``            src/block/block.rs:12-97
* ``BlockManager.rpc_put_block``                 src/block/manager.rs:366-408
* ``BlockManager.rpc_get_block`` / ``rpc_get_raw_block``   :243-363
* ``block_incref`` / ``block_decref``            :452-500 (rc semantics of src/block/rc.rs)
* ``resync_block``                               src/block/resync.rs:354-503
* scrub                                          src/block/repair.rs:438-490

What changes with EC: instead of sending the SAME bytes to ``replication_factor``
nodes (``try_write_many_sets``, src/rpc/rpc_helper.rs:432-538), shard j of the
block goes to ``who[j]``; reads gather any k shards and reconstruct when a data
shard is missing.  Every shard byte is produced by the codec passed in
(``garage_amd.ReedSolomon`` = the GPU library); this module only moves buffers.

Nodes are in-process objects (the way the reference tests multi-node logic with
several NetApp instances on loopback, src/net/test.rs:15-118); the network and
the metadata tables are out of scope.

FROZEN (round 3): the product mirror is the C++ one (garage_amd/csrc/bm_*.cpp, include/garage_block.h); this module
stays for the scenarios both mirrors share in the tests and for on-disk interop checks, and takes no new features.
"""
from __future__ import annotations

import enum
import os
import struct
import threading
from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np

from .codec import shard_len, shardsum
from .partition import block_hash

INLINE_THRESHOLD = 3072  # src/block/manager.rs:46

from . import zstd_ffi  # Garage compresses with zstd (src/block/block.rs:99-106)


# ------------------------------------------------------------------ errors
class Error(Exception):
    """garage_util::error::Error (src/util/error.rs:14-78), the variants this path uses."""


class CorruptData(Error):
    def __init__(self, hash_: bytes):
        super().__init__(f"Corrupt data: does not match hash {hash_.hex()[:16]}")
        self.hash = hash_


class MissingBlock(Error):
    def __init__(self, hash_: bytes):
        super().__init__(f"Missing block {hash_.hex()[:16]}: no node returned a valid block")
        self.hash = hash_


class Quorum(Error):
    def __init__(self, quorum: int, ok: int, total: int, errors: Sequence[str]):
        super().__init__(f"Could not reach quorum of {quorum}. {ok} of {total} request succeeded, others returned errors: {list(errors)}")
        self.quorum, self.ok, self.total = quorum, ok, total


# --------------------------------------------------------------- DataBlock
class DataBlockHeader(enum.Enum):
    Plain = 0
    Compressed = 1

    def is_compressed(self) -> bool:
        return self is DataBlockHeader.Compressed


@dataclass
class DataBlock:
    """A possibly compressed block of data (src/block/block.rs:24-25)."""
    header: DataBlockHeader
    elem: bytes

    @classmethod
    def from_parts(cls, header: DataBlockHeader, elem: bytes) -> "DataBlock":
        return cls(header, elem)

    @classmethod
    def plain(cls, elem: bytes) -> "DataBlock":
        return cls(DataBlockHeader.Plain, elem)

    @classmethod
    def compressed(cls, elem: bytes) -> "DataBlock":
        return cls(DataBlockHeader.Compressed, elem)

    def into_parts(self):
        return self.header, self.elem

    def as_parts_ref(self):
        return self.header, self.elem

    def verify(self, hash_: bytes) -> None:
        """Plain: blake2sum == hash; Compressed: the zstd frame (with checksum) decodes."""
        if self.header is DataBlockHeader.Plain:
            if block_hash(self.elem) != hash_:
                raise CorruptData(hash_)
        else:
            try:
                zstd_decode(self.elem)
            except Exception:
                raise CorruptData(hash_) from None

    @classmethod
    def from_buffer(cls, data: bytes, level: Optional[int]) -> "DataBlock":
        """zstd at `level` (Garage's default config is Some(1)), Plain on None or on
        any encoder error (src/block/block.rs:85-96)."""
        if level is not None:
            try:
                return cls.compressed(zstd_encode(data, level))
            except Exception:
                pass
        return cls.plain(data)


def zstd_encode(data: bytes, level: int) -> bytes:
    """One frame with the content checksum on, like the reference's zstd_encode."""
    return zstd_ffi.zstd_encode(data, level)


def zstd_decode(data: bytes) -> bytes:
    return zstd_ffi.zstd_decode(data)


# ------------------------------------------------------------ shard format
@dataclass
class ShardHeader:
    """64-byte header in front of every stored / transmitted shard (row f2).  A
    block file is self-verifying against its name (blake2 of the content,
    src/block/block.rs:69-77); a shard is not, so it carries its own checksum."""
    k: int
    m: int
    idx: int
    compressed: bool
    orig_len: int      # length of the (possibly compressed) block payload
    shard_len: int
    checksum: bytes    # shardsum of the shard payload (32 bytes, BLAKE2b tree mode)

    MAGIC = b"GECS"
    VERSION = 2        # version 2: checksum = shardsum (BLAKE2b tree mode, include/garage_ec.h); 1 was plain blake2sum
    SIZE = 64
    _FMT = "<4sBBBBB3xQII32s"

    def pack(self) -> bytes:
        b = struct.pack(self._FMT, self.MAGIC, self.VERSION, self.k, self.m, self.idx, int(self.compressed),
                        self.orig_len, self.shard_len, 0, self.checksum)
        return b.ljust(self.SIZE, b"\0")

    @classmethod
    def unpack(cls, raw: bytes) -> "ShardHeader":
        if len(raw) < cls.SIZE:
            raise ValueError("short shard header")
        magic, ver, k, m, idx, comp, orig_len, slen, _, csum = struct.unpack(cls._FMT, raw[: struct.calcsize(cls._FMT)])
        if magic != cls.MAGIC or ver != cls.VERSION:
            raise ValueError("bad shard magic/version")
        return cls(k, m, idx, bool(comp), orig_len, slen, csum)


# ------------------------------------------------------------------- nodes
class ShardStore:
    """One storage node's local shard files.  `down` simulates an unreachable node."""

    def __init__(self):
        self.down = False

    def put(self, hash_: bytes, idx: int, raw: bytes) -> None:
        raise NotImplementedError

    def get(self, hash_: bytes, idx: int) -> Optional[bytes]:
        raise NotImplementedError

    def delete(self, hash_: bytes, idx: int) -> None:
        raise NotImplementedError

    def _check(self):
        if self.down:
            raise ConnectionError("node unreachable")


class MemoryShardStore(ShardStore):
    def __init__(self):
        super().__init__()
        self.files: dict[tuple[bytes, int], bytes] = {}

    def put(self, hash_, idx, raw):
        self._check()
        self.files[(hash_, idx)] = bytes(raw)

    def get(self, hash_, idx):
        self._check()
        return self.files.get((hash_, idx))

    def delete(self, hash_, idx):
        self._check()
        self.files.pop((hash_, idx), None)


class DirShardStore(ShardStore):
    """Garage's on-disk naming: ``<root>/<h[0]>/<h[1]>/<hex(hash)>`` (block_dir_from,
    src/block/layout.rs:286-291) with a ``.s<idx>`` suffix; tmp-file + rename like
    write_block_inner (src/block/manager.rs:720-805)."""

    def __init__(self, root: str, fsync: bool = False):
        super().__init__()
        self.root, self.fsync = root, fsync

    def _path(self, hash_: bytes, idx: int) -> str:
        hx = hash_.hex()
        return os.path.join(self.root, hx[0:2], hx[2:4], f"{hx}.s{idx}")

    def put(self, hash_, idx, raw):
        self._check()
        path = self._path(hash_, idx)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        tmp = f"{path}.tmp{os.getpid()}_{threading.get_ident()}"
        with open(tmp, "wb") as f:
            f.write(raw)
            if self.fsync:
                f.flush()
                os.fsync(f.fileno())
        os.replace(tmp, path)

    def get(self, hash_, idx):
        self._check()
        try:
            with open(self._path(hash_, idx), "rb") as f:
                return f.read()
        except FileNotFoundError:
            return None

    def delete(self, hash_, idx):
        self._check()
        try:
            os.remove(self._path(hash_, idx))
        except FileNotFoundError:
            pass

    def mark_corrupted(self, hash_, idx):
        """rename to *.corrupted like src/block/manager.rs:807-819"""
        p = self._path(hash_, idx)
        if os.path.exists(p):
            os.replace(p, p + ".corrupted")


# ------------------------------------------------------------ BlockManager
class BlockManager:
    """EC counterpart of garage_block::BlockManager for one cluster of in-process nodes."""

    def __init__(self, codec, stores: Sequence[ShardStore], compression_level: Optional[int] = None,
                 write_quorum: Optional[int] = None):
        self.codec = codec
        self.k, self.m = codec.k, codec.m
        self.n = self.k + self.m
        if len(stores) < self.n:
            # nodes_of asserts n == replication_factor (src/rpc/layout/version.rs:118)
            raise Error(f"RS({self.k},{self.m}) needs at least {self.n} storage nodes, got {len(stores)}")
        self.stores = list(stores)
        self.compression_level = compression_level
        # writes must land on >= k shards to be readable at all; default margin = half the parity
        self.write_quorum = write_quorum if write_quorum is not None else self.k + (self.m + 1) // 2
        self.rc: dict[bytes, int] = {}
        # RcEntry::Deletable{at_time} (src/block/rc.rs:122-240): a block whose count is zero may only be deleted
        # once BLOCK_GC_DELAY (src/block/manager.rs:51) has passed; no entry at all = Absent = deletable
        self.deletable_at: dict[bytes, int] = {}
        self.gc_delay_ms = 600_000
        self._clock_skew_ms = 0
        self.resync_queue: list[bytes] = []
        self.metrics = {"bytes_written": 0, "bytes_read": 0, "corruption_counter": 0, "ec_reconstructs": 0}

    def now_ms(self) -> int:
        import time

        return int(time.time() * 1000) + self._clock_skew_ms

    def clock_advance(self, ms: int) -> None:
        self._clock_skew_ms += ms

    # -- placement: partition = top byte of the hash (src/rpc/layout/version.rs:101-104)
    def storage_nodes_of(self, hash_: bytes) -> list[int]:
        start = (hash_[0] * 31 + hash_[1]) % len(self.stores)
        return [(start + j) % len(self.stores) for j in range(self.n)]

    # -- write path ---------------------------------------------------------
    def rpc_put_block(self, hash_: bytes, data: bytes, prevent_compression: bool = False, order_tag=None) -> None:
        """Send block to nodes that should have it (one shard each)."""
        self.rpc_put_blocks([(hash_, data)], prevent_compression)

    def rpc_put_blocks(self, items: Sequence[tuple[bytes, bytes]], prevent_compression: bool = False) -> None:
        """Batched form: the coalescing queue in front of the FFI -- all blocks go
        to the device in ONE encode call."""
        level = None if prevent_compression else self.compression_level
        blocks = [DataBlock.from_buffer(data, level) for _, data in items]
        # Shard geometry is a pure function of the block: S = shard_len(k, payload length),
        # never the batch maximum -- a later put of the same block must produce compatible
        # shards.  Blocks of equal S share ONE device call.
        by_s: dict[int, list[int]] = {}
        for i, blk in enumerate(blocks):
            by_s.setdefault(shard_len(self.k, len(blk.elem)), []).append(i)
        failure = None
        for S, ids in by_s.items():
            parities = self.codec.encode_blocks([blocks[i].elem for i in ids], S)
            for i, par in zip(ids, parities):
                hash_, blk = items[i][0], blocks[i]
                padded = np.zeros(self.k * S, dtype=np.uint8)
                padded[: len(blk.elem)] = np.frombuffer(blk.elem, dtype=np.uint8)
                shards = [padded[j * S:(j + 1) * S] for j in range(self.k)] + [np.asarray(par[r]) for r in range(self.m)]
                who = self.storage_nodes_of(hash_)
                ok, errors = 0, []
                for j, node in enumerate(who):
                    payload = shards[j].tobytes()
                    hdr = ShardHeader(self.k, self.m, j, blk.header.is_compressed(), len(blk.elem), S, shardsum(payload))
                    try:
                        self.stores[node].put(hash_, j, hdr.pack() + payload)
                        ok += 1
                        self.metrics["bytes_written"] += len(payload)
                    except Exception as e:  # node down: stragglers are retried by resync
                        errors.append(f"node {node}: {e}")
                if ok < self.write_quorum:
                    failure = Quorum(self.write_quorum, ok, self.n, errors)
                    continue
                # a block nobody references yet (PutObject runs the put and the incref concurrently,
                # src/api/s3/put.rs:545-581) is protected for BLOCK_GC_DELAY like one whose count just
                # dropped to zero: resync must never delete what a put has just acknowledged
                if self.rc.get(hash_, 0) == 0:
                    self.deletable_at[hash_] = max(self.deletable_at.get(hash_, 0), self.now_ms() + self.gc_delay_ms)
                if ok < self.n:
                    self.resync_queue.append(hash_)  # stragglers are REBUILT by resync while the block is needed
        if failure is not None:
            raise failure

    # -- read path ------------------------------------------------------------
    def _gather(self, hash_: bytes, want: int):
        """Fetch shards in node order until `want` mutually consistent ones are in hand.
        A shard whose checksum does not match is treated as missing and queued for
        resync.  Shards are grouped by geometry (compressed, orig_len, shard_len): a
        block can have leftovers of another geometry on disk (e.g. a later put with
        compression enabled that failed its quorum half-way); the largest consistent
        group wins, like find_block chooses between <hash> and <hash>.zst
        (src/block/manager.rs:627-662)."""
        who = self.storage_nodes_of(hash_)
        groups: dict[tuple, tuple[ShardHeader, dict[int, np.ndarray]]] = {}
        for j, node in enumerate(who):
            if groups and max(len(g[1]) for g in groups.values()) >= want:
                break
            try:
                raw = self.stores[node].get(hash_, j)
            except Exception:
                continue
            if raw is None:
                continue
            try:
                hdr = ShardHeader.unpack(raw)
                payload = raw[ShardHeader.SIZE:]
                if hdr.idx != j or hdr.k != self.k or hdr.m != self.m or len(payload) != hdr.shard_len \
                        or shardsum(payload) != hdr.checksum:
                    raise ValueError("shard checksum/geometry mismatch")
            except ValueError:
                self.metrics["corruption_counter"] += 1
                self.resync_queue.append(hash_)
                store = self.stores[node]
                if hasattr(store, "mark_corrupted"):
                    store.mark_corrupted(hash_, j)
                continue
            key = (hdr.compressed, hdr.orig_len, hdr.shard_len)
            groups.setdefault(key, (hdr, {}))[1][j] = np.frombuffer(payload, dtype=np.uint8)
            self.metrics["bytes_read"] += len(payload)
        if not groups:
            return {}, None
        if len(groups) > 1:
            self.resync_queue.append(hash_)
        meta, got = max(groups.values(), key=lambda g: len(g[1]))
        return got, meta

    def rpc_get_raw_block(self, hash_: bytes, order_tag=None) -> DataBlock:
        got, meta = self._gather(hash_, self.k)
        if meta is None or len(got) < self.k:
            raise MissingBlock(hash_)
        if any(j not in got for j in range(self.k)):
            row = [got.get(j) for j in range(self.n)]
            row = self.codec.reconstruct_data([row])[0]
            self.metrics["ec_reconstructs"] += 1
            data = [row[j] for j in range(self.k)]
        else:
            data = [got[j] for j in range(self.k)]
        payload = np.concatenate(data)[: meta.orig_len].tobytes()
        header = DataBlockHeader.Compressed if meta.compressed else DataBlockHeader.Plain
        return DataBlock.from_parts(header, payload)

    def rpc_get_block(self, hash_: bytes, order_tag=None) -> bytes:
        """rpc_get_block_streaming collected into bytes: decompress if needed and,
        for plain blocks, check the content against its name."""
        blk = self.rpc_get_raw_block(hash_, order_tag)
        blk.verify(hash_)
        return zstd_decode(blk.elem) if blk.header.is_compressed() else blk.elem

    # -- refcounts (src/block/rc.rs) --------------------------------------------
    def block_incref(self, hash_: bytes) -> None:
        self.rc[hash_] = self.rc.get(hash_, 0) + 1
        if self.rc[hash_] == 1:
            self.deletable_at.pop(hash_, None)
            self.resync_queue.append(hash_)  # presence check later (manager.rs:452-475)

    def block_decref(self, hash_: bytes) -> None:
        if self.rc.get(hash_, 0) == 0:
            return  # Deletable / Absent stay what they are (RcEntry::decrement)
        self.rc[hash_] -= 1
        if self.rc[hash_] == 0:
            self.deletable_at[hash_] = self.now_ms() + self.gc_delay_ms
            self.resync_queue.append(hash_)  # (the reference queues it BLOCK_GC_DELAY + 10 s later, manager.rs:478-500)

    def _is_deletable(self, hash_: bytes) -> bool:
        if self.rc.get(hash_, 0) > 0:
            return False
        at = self.deletable_at.get(hash_)
        return at is None or self.now_ms() > at

    # -- repair ---------------------------------------------------------------
    def resync_block(self, hash_: bytes) -> int:
        """rc > 0: every node gets back the shard it should hold (gather k, rebuild
        all, rewrite the missing/corrupt ones).  rc == 0: delete.  Returns the
        number of shards rewritten or deleted."""
        who = self.storage_nodes_of(hash_)
        if self._is_deletable(hash_):
            nd = 0
            for j, node in enumerate(who):
                try:
                    if self.stores[node].get(hash_, j) is not None:
                        self.stores[node].delete(hash_, j)
                        nd += 1
                except Exception:
                    pass
            self.deletable_at.pop(hash_, None)  # clear_deleted_block_rc
            self.rc.pop(hash_, None)
            return nd
        got, meta = self._gather(hash_, self.n)
        if meta is None or len(got) < self.k:
            raise MissingBlock(hash_)
        if len(got) == self.n:
            return 0
        row = self.codec.reconstruct([[got.get(j) for j in range(self.n)]])[0]
        self.metrics["ec_reconstructs"] += 1
        fixed = 0
        for j, node in enumerate(who):
            if j in got:
                continue
            payload = np.asarray(row[j]).tobytes()
            hdr = ShardHeader(self.k, self.m, j, meta.compressed, meta.orig_len, meta.shard_len, shardsum(payload))
            try:
                self.stores[node].put(hash_, j, hdr.pack() + payload)
                fixed += 1
            except Exception:
                self.resync_queue.append(hash_)
        return fixed

    def resync_all(self) -> int:
        todo, self.resync_queue = list(dict.fromkeys(self.resync_queue)), []
        total = 0
        for h in todo:
            total += self.resync_block(h)
            if self.rc.get(h, 0) == 0 and h in self.deletable_at:
                self.resync_queue.append(h)  # still inside its GC delay: look again later
        return total

    def scrub(self, hashes: Sequence[bytes]) -> list[bytes]:
        """Batch-verify stripes on the device (ReedSolomon::verify): returns the
        hashes whose shards are inconsistent or unreadable."""
        full, idx, bad = [], [], []
        for h in hashes:
            got, meta = self._gather(h, self.n)
            if len(got) == self.n:
                full.append(np.stack([got[j] for j in range(self.n)]))
                idx.append(h)
            else:
                bad.append(h)
        by_len: dict[int, list[int]] = {}
        for i, st in enumerate(full):
            by_len.setdefault(st.shape[1], []).append(i)
        for ids in by_len.values():
            ok = self.codec.verify(np.stack([full[i] for i in ids]))
            bad += [idx[i] for i, o in zip(ids, ok) if not o]
        return bad
