/*
 * garage_block.h -- C API of libgarage_block.so: the host-side (C++) mirror of
 * garage_block::BlockManager with erasure-coded shard fan-out, built on top of
 * libgarage_ec's C ABI (include/garage_ec.h).  SURVEY.md section 8 rows f1-f3.
 *
 * It mirrors, by name and behaviour:
 *   BlockManager::rpc_put_block          src/block/manager.rs:366-408
 *   BlockManager::rpc_get_block(_streaming) / rpc_get_raw_block   :243-363
 *   BlockManager::block_incref/decref    :452-500  (rc: src/block/rc.rs)
 *   BlockResyncManager::resync_block     src/block/resync.rs:354-503
 *   ScrubWorker verify                   src/block/repair.rs:438-490
 *   blake2sum (blake2b-512 truncated to 32 bytes)   src/util/data.rs:130-138
 * Storage nodes are in-process objects (memory- or directory-backed) -- the way
 * the reference tests multi-node logic on loopback (src/net/test.rs:15-118);
 * the network and the metadata tables are out of scope.  zstd (DataBlock::
 * from_buffer, src/block/block.rs:85-106) goes through the system's libzstd.so.1,
 * resolved at run time.
 */
#ifndef GARAGE_BLOCK_H
#define GARAGE_BLOCK_H

#include <stddef.h>
#include <stdint.h>

#include "garage_ec.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gbm_manager gbm_manager;

enum {
	GBM_OK = 0,
	GBM_E_MISSING_BLOCK = -1, /* Error::MissingBlock  (src/util/error.rs:74-77) */
	GBM_E_CORRUPT_DATA = -2,  /* Error::CorruptData   (:70-72) */
	GBM_E_QUORUM = -3,        /* Error::Quorum        (:55-62) */
	GBM_E_INVALID_ARG = -4,
	GBM_E_EC = -5,            /* libgarage_ec returned an error; see gbm_last_error() */
	GBM_E_IO = -6,
	GBM_E_BUFFER_TOO_SMALL = -7
};

#define GBM_INLINE_THRESHOLD 3072 /* src/block/manager.rs:46 */
#define GBM_SHARD_HEADER_SIZE 64

const char *gbm_last_error(void);

/* Garage's content hash: blake2b-512 truncated to 32 bytes (NOT blake2b-256). */
void gbm_blake2sum(const uint8_t *data, size_t len, uint8_t out[32]);

/* node_dirs == NULL: in-memory nodes; otherwise nnodes directory roots using
 * Garage's naming <root>/<h0>/<h1>/<hex>.s<idx> (src/block/layout.rs:286-291).
 * write_quorum <= 0: k + ceil(m/2).  nnodes must be >= k+m
 * (src/rpc/layout/version.rs:118).  The codec is borrowed, not owned. */
int gbm_create(const gec_codec *codec, int nnodes, const char *const *node_dirs,
	       int write_quorum, gbm_manager **out);
void gbm_destroy(gbm_manager *m);

/* Config.compression_level (src/util/config.rs:52-58): enabled=0 is "none";
 * Garage's default is level 1.  Blocks are compressed (one zstd frame, content
 * checksum on) before they are cut into shards; on any encoder error the block
 * is stored Plain (src/block/block.rs:88-93). */
int gbm_set_compression_level(gbm_manager *m, int enabled, int level);

/* Config.data_fsync (src/util/config.rs:22-24; off by default): directory-backed nodes
 * fsync the shard file before the rename and its directory after it
 * (write_block_inner, src/block/manager.rs:775-800).  No effect on in-memory nodes. */
int gbm_set_data_fsync(gbm_manager *m, int enabled);

/* nodes_out[k+m]: node index that stores shard j of this hash. */
int gbm_storage_nodes_of(const gbm_manager *m, const uint8_t hash[32], int *nodes_out);

/* Send block to nodes that should have it: shard j to nodes_of(hash)[j]. */
int gbm_rpc_put_block(gbm_manager *m, const uint8_t hash[32], const uint8_t *data, size_t len);
/* Coalesced form: ONE device encode for all n blocks (hashes = n*32 bytes). */
int gbm_rpc_put_blocks(gbm_manager *m, size_t n, const uint8_t *hashes,
		       const uint8_t *const *data, const size_t *len);

/* Gather >= k shards, reconstruct if a data shard is missing, check the
 * content against its name.  *len_out = block length (also on
 * GBM_E_BUFFER_TOO_SMALL). */
int gbm_rpc_get_block(gbm_manager *m, const uint8_t hash[32], uint8_t *out, size_t cap, size_t *len_out);
/* Batched: ONE device reconstruct for all blocks that need it.  out[i] has
 * cap[i] bytes; rc[i] receives the per-block result. */
int gbm_rpc_get_blocks(gbm_manager *m, size_t n, const uint8_t *hashes, uint8_t *const *out,
		       const size_t *cap, size_t *len_out, int *rc);

/* Coalescing queue in front of the device: Garage keeps <= 3 block puts in flight
 * per PutObject (PUT_BLOCKS_MAX_PARALLEL, src/api/s3/put.rs:42,486-511) across many
 * concurrent requests.  gbm_batcher_put_block is thread-safe and blocks its caller
 * (like `rpc_put_block(..).await`) until the batch that contains the block has been
 * encoded and fanned out; one worker thread turns everything queued within
 * max_wait_us (or max_blocks) into ONE device call.  Returns that block's own
 * result (GBM_OK / GBM_E_QUORUM / a device error). */
typedef struct gbm_batcher gbm_batcher;
int gbm_batcher_create(gbm_manager *m, size_t max_blocks, unsigned max_wait_us, gbm_batcher **out);
void gbm_batcher_destroy(gbm_batcher *b);
int gbm_batcher_put_block(gbm_batcher *b, const uint8_t hash[32], const uint8_t *data, size_t len);
/* Config.block_ram_buffer_max (src/util/config.rs:74-76,276-278; default 256 MiB): bytes of blocks that
 * may be on their way to the storage nodes at once.  gbm_batcher_put_block takes len/1024 permits before it
 * queues the block and returns them when its batch has been fanned out (buffer_kb_semaphore,
 * src/block/manager.rs:380-384); callers beyond the budget wait. */
int gbm_batcher_set_ram_buffer_max(gbm_batcher *b, size_t bytes);
/* out = { device batches issued, blocks put, largest batch } */
int gbm_batcher_stats(gbm_batcher *b, uint64_t out[3]);

int gbm_block_incref(gbm_manager *m, const uint8_t hash[32]);
int gbm_block_decref(gbm_manager *m, const uint8_t hash[32]);

/* rc > 0: rewrite every missing/corrupt shard; rc == 0: delete all shards.
 * *changed = shards rewritten or deleted. */
int gbm_resync_block(gbm_manager *m, const uint8_t hash[32], int *changed);
int gbm_resync_all(gbm_manager *m, int *changed);
size_t gbm_resync_queue_len(const gbm_manager *m);

/* Batch verify on the device: bad_out[i] = 1 if block i is inconsistent or
 * not fully readable. */
int gbm_scrub(gbm_manager *m, size_t n, const uint8_t *hashes, uint8_t *bad_out);

/* Fault injection / inspection for tests. */
int gbm_node_set_down(gbm_manager *m, int node, int down);
int gbm_node_has_shard(gbm_manager *m, int node, const uint8_t hash[32], int idx);
int gbm_node_delete_shard(gbm_manager *m, int node, const uint8_t hash[32], int idx);
/* XOR `mask` into payload byte `offset` of the stored shard; fix_checksum != 0
 * re-stamps the header checksum (silent corruption only scrub can find). */
int gbm_node_corrupt_shard(gbm_manager *m, int node, const uint8_t hash[32], int idx,
			   size_t offset, uint8_t mask, int fix_checksum);

/* out[0..5] = bytes_written, bytes_read, corruption_counter, ec_reconstructs,
 * blocks_put, blocks_get */
int gbm_metrics(const gbm_manager *m, uint64_t out[6]);
/* number of messages (shards / blocks) whose blake2sum was computed on the GPU */
uint64_t gbm_gpu_hashed(const gbm_manager *m);

#ifdef __cplusplus
}
#endif
#endif
