/*
 * garage_ec.h -- C ABI of libgarage_ec.so, the MI355X (gfx950) Reed-Solomon
 * engine for Garage's object-block write/read path.
 *
 * Garage has NO plugin/FFI interface at this spot: the
 * only codec hook is the hard-wired zstd call in
 *   src/block/manager.rs:375-378  (DataBlock::from_buffer inside rpc_put_block)
 * so every entry point below cites the reference code it slots in next to /
 * replaces, and -- for the arithmetic -- the `reed-solomon-erasure::galois_8`
 * API [EXT, crate not vendored] whose results it must reproduce bit-exactly.
 * A Rust caller uses it exactly like zstd is used today: a blocking call inside
 * tokio::task::spawn_blocking (src/block/block.rs:85-96).  INTEGRATION.md has
 * the `extern "C"` binding a Garage maintainer would add.
 *
 * Conventions
 *  - plain pointers and sizes only; no exceptions/aborts cross the boundary;
 *  - return 0 (GEC_OK) or a negative code; gec_last_error() gives thread-local
 *    detail (e.g. hipGetErrorString);
 *  - a codec is immutable after creation and may be shared by any number of
 *    threads (Rust: Send + Sync); every entry point is blocking w.r.t. host
 *    buffers; *_dev entry points are asynchronous on the given HIP stream;
 *  - a codec has a backend: GEC_BACKEND_HIP computes every shard byte on the
 *    GPU with hand-written gfx950 kernels and is the product; GEC_BACKEND_CPU is the library's own
 *    host-core data path (AVX-512 + GFNI / AVX2 / scalar, chosen at run time) behind the same
 *    host-pointer entry points -- what a node runs on when it has no GPU, or has lost it, so that it
 *    can still read and repair its own erasure-coded blocks (BASELINE config 1's "CPU path via
 *    BlockManager").  Both produce identical bytes.  The *_dev and gec_group_* entry points need a
 *    HIP codec.
 *
 * Shard geometry: a block of L bytes is cut into k
 * data shards of S = round_up(ceil(L/k), 64) bytes, shard i = block bytes
 * [i*S, (i+1)*S) zero-extended, so data shards are zero-copy slices of the
 * (padded) block buffer and every shard starts on a 64-byte line.
 *
 * Environment: every switch is optional, read once per process, and none changes results.  The one
 * table (name, default, meaning) lives in garage_amd/csrc/ec_env.cpp; gec_env_table() returns it as
 * text and INTEGRATION.md section 6 prints it.
 */
#ifndef GARAGE_EC_H
#define GARAGE_EC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GEC_VERSION 0x00040000u /* major.minor.patch = 0.4.0 (0.4: shard checksum kinds, default MLH64; peer-pointer striped decode; per-block erasure patterns) */
#define GEC_MAX_SHARDS 256      /* GF(2^8): k + m <= 256 [EXT] */

typedef struct gec_codec gec_codec; /* opaque */

/* Error codes: -1..-10 mirror reed_solomon_erasure::Error [EXT]; the Rust shim
 * maps them next to Error::CorruptData / Error::MissingBlock
 * (src/util/error.rs:70-77). */
enum {
	GEC_OK = 0,
	GEC_E_TOO_FEW_SHARDS = -1,
	GEC_E_TOO_MANY_SHARDS = -2,
	GEC_E_TOO_FEW_DATA = -3,
	GEC_E_TOO_MANY_DATA = -4,
	GEC_E_TOO_FEW_PARITY = -5,
	GEC_E_TOO_MANY_PARITY = -6,
	GEC_E_INCORRECT_SHARD_SIZE = -7,
	GEC_E_TOO_FEW_PRESENT = -8,
	GEC_E_EMPTY_SHARD = -9,
	GEC_E_INVALID_INDEX = -10,
	GEC_E_DEVICE = -100,     /* HIP failure / no GPU; see gec_last_error() */
	GEC_E_NOMEM = -101,
	GEC_E_INVALID_ARG = -102 /* NULL pointer, misaligned device buffer, ... */
};

/* ------------------------------------------------------------- library */
uint32_t gec_version(void);
/* Number of usable HIP devices (0 => only GEC_BACKEND_CPU codecs can be created). */
int gec_device_count(void);
/* Which of a node's `ndev` devices owns a block: hash[4] % ndev on Garage's 32-byte block hash (blake2sum,
 * src/util/data.rs:130-138).  Garage already places by hash bits at three levels -- cluster partition = bytes 0-1
 * (partition_of, src/rpc/layout/version.rs:101-104), drive = bytes 2-3 (src/block/layout.rs:278-284), mutation
 * lock = bytes 0-1 (src/block/manager.rs:679-689) -- so byte 4 makes the device independent of node, drive and
 * lock stripe.  Blocks are independent units: no data-path collective.  The ONE definition of the rule:
 * libgarage_block's multi-device manager (gbm_create_multi), garage_amd/partition.py and bench.py all call it.
 * ndev < 1 or a NULL hash: -1. */
int gec_device_of_hash(const uint8_t hash[32], int ndev);
/* A blocking host-pointer call keeps the host link busy only for part of its time: the kernels that read the caller's
 * buffers and write results into them come first, checksum kernels over what is by then in HBM and a few small copies
 * follow.  A scheduler that lets one call per device onto the link at a time (libgarage_block's coalescing queue does) can
 * let the next one start at that point instead of at the return.  gec_thread_link_release(fn, arg) arms the CALLING
 * THREAD's next gec_encode_hash_batch / gec_decode_verify_batch: fn(arg) is called exactly once, on that thread, when the
 * call's bulk transfers over the link have finished -- at the latest right before the call returns (errors, calls without
 * such a phase, CPU codecs).  fn = NULL disarms.  Results are unaffected. */
typedef void (*gec_link_release_fn)(void *arg);
void gec_thread_link_release(gec_link_release_fn fn, void *arg);
/* Every GEC_* environment switch: "NAME<tab>default<tab>meaning" lines (static storage). */
const char *gec_env_table(void);
/* The kernel a GEC_BACKEND_CPU codec runs on this host: "avx512+gfni", "avx2" or "scalar" (static storage). */
const char *gec_cpu_isa(void);
const char *gec_strerror(int code);
/* Thread-local detail string of the last failing call on this thread. */
const char *gec_last_error(void);

/* ------------------------------------------------- host logic (no GPU) */
/* S = round_up(ceil(block_len/k), 64); 0 if k <= 0.  Decides how
 * rpc_put_block's `data: Bytes` (src/block/manager.rs:366-371) is sliced. */
size_t gec_shard_len(int k, size_t block_len);
/* (k+m) x k systematic encoding matrix, row-major.
 * == ReedSolomon::new(k,m) internal matrix [EXT core.rs build_matrix]. */
int gec_build_matrix(int k, int m, uint8_t *out_n_by_k);

/* Coding-matrix families.  VANDERMONDE is the reed-solomon-erasure / Backblaze
 * systematic matrix and the only one that is bit-compatible with the crate (the
 * default everywhere).  CAUCHY is the extra mode the project brief names:
 * parity row r, column c = 1 / ((k + r) XOR c) over GF(2^8)/0x11D -- every square
 * sub-matrix of a Cauchy matrix is invertible, so [I; C] is MDS by construction
 * (the same formula as klauspost/reedsolomon's WithCauchyMatrix).  Shards encoded
 * with one family cannot be decoded with the other. */
enum { GEC_MATRIX_VANDERMONDE = 0, GEC_MATRIX_CAUCHY = 1 };
int gec_build_matrix_ex(int k, int m, int matrix, uint8_t *out_n_by_k);
/* present[k+m] (0/1).  valid_out[k] = first k present shard indices,
 * out_k_by_k = inverse of those rows [EXT core.rs get_data_decode_matrix]. */
int gec_build_decode_matrix(int k, int m, const uint8_t *present,
			    int32_t *valid_out, uint8_t *out_k_by_k);

/* --------------------------------------------------------------- codec */
/* Who moves the bytes.  HIP: `device` names the GPU; without a usable one creation fails with GEC_E_DEVICE.
 * CPU: the host cores (`device` is ignored) -- the same host-pointer entry points, the same bytes; the *_dev
 * and gec_group_* entry points answer GEC_E_DEVICE.  AUTO: HIP when gec_device_count() > 0, else CPU -- what
 * a Garage node uses so that losing its GPU degrades its throughput, not its ability to read and repair
 * (the codec call sits in spawn_blocking either way, src/block/block.rs:85-96). */
enum { GEC_BACKEND_CPU = 0, GEC_BACKEND_HIP = 1, GEC_BACKEND_AUTO = 2 };
/* == ReedSolomon::new(data_shards, parity_shards) [EXT]; belongs in
 * BlockManager::new (src/block/manager.rs:122-192) next to the
 * compression_level / data_fsync fields, built from new Config keys.
 * Argument errors are reported before any device is touched.  (SURVEY.md Appendix B's signature.) */
int gec_codec_create(int k, int m, int backend, int device, gec_codec **out);
/* Same with an explicit matrix family (gec_codec_create == GEC_MATRIX_VANDERMONDE). */
int gec_codec_create_ex(int k, int m, int backend, int device, int matrix, gec_codec **out);

/* Which SHARD CHECKSUM a codec's *_hash_* / gec_shardsum_* entry points produce (32 bytes per shard either way; both are
 * defined further down, next to gec_shardsum_batch):
 *   GEC_SHARDSUM_MLH64         (3, the default) a multilinear hash over 32-bit words in 4 KiB leaves under a BLAKE2b root --
 *                              accumulated by the RS kernels from the registers that hold the bytes anyway, and checked by a
 *                              host core at memory speed;
 *   GEC_SHARDSUM_BLAKE2B_TREE  (2) BLAKE2b in tree mode, what rounds 2-4 wrote into shard headers (version 2): kept so that
 *                              stores written then stay readable; a second pass over the stripe at 15 lane-ops per byte.
 * The numbers are the shard-header versions of libgarage_block (include/garage_block.h).  A codec produces ONE kind; a store
 * with shards of both kinds keeps a sibling codec for the other (gec_codec_with_shardsum; gec_codec_background keeps its
 * parent's kind). */
enum { GEC_SHARDSUM_DEFAULT = 0, GEC_SHARDSUM_BLAKE2B_TREE = 2, GEC_SHARDSUM_MLH64 = 3 };
int gec_codec_create_ex2(int k, int m, int backend, int device, int matrix, int shardsum, gec_codec **out);
/* a sibling of `c` -- same code, backend, device and class -- that produces the other kind */
int gec_codec_with_shardsum(const gec_codec *c, int shardsum, gec_codec **out);
int gec_codec_shardsum(const gec_codec *c); /* GEC_SHARDSUM_BLAKE2B_TREE or GEC_SHARDSUM_MLH64 */
/* One shard's checksum of the given kind on the CALLING host core, no codec and no device involved: what a node runs over a
 * shard it is about to serve, and what a requester runs over the k shards of a healthy get instead of sending them to the
 * device (MLH64: AVX-512 / AVX2 / scalar at tens of GB/s per core; BLAKE2b tree: ~1 GB/s). */
int gec_shardsum_host(int shardsum, const uint8_t *data, size_t len, uint8_t out[32]);

/* Foreground and background work.  Garage keeps repair off the request path with a bounded worker pool and a
 * Tranquilizer (src/block/resync.rs:43-46,513-599, src/util/tranquilizer.rs:38-69); on a device the same split
 * needs the codec's cooperation.  gec_codec_background returns a sibling of `c` -- same code, same backend and
 * device, its own staging slots and copy threads -- whose work is classed BACKGROUND:
 *   - its streams run at the device's lowest priority and on a subset of the CUs (GEC_BG_CUS), so a long
 *     checksum kernel of a scrub never holds the whole chip when a PutObject's encode arrives;
 *   - its host-pointer trips go in small chunks (GEC_BG_CHUNK_MB) and, before each chunk, wait (up to
 *     GEC_BG_YIELD_US) for the foreground calls in flight on the same device to finish: a foreground call
 *     finds the link, the copy threads and the CUs busy with at most one background chunk;
 *   - its link kernels -- like every codec's -- are launched with no more workgroups than its CUs hold at once
 *     (each walks a list of tiles): a launch with more keeps its queue's dispatcher busy until the last workgroup
 *     is placed, and kernels of foreground streams that share that dispatcher would wait for as long
 *     (tools/dispatch_probe);
 *   - its link kernels give way on the device: foreground link kernels count their running workgroups in a word of
 *     device memory, and a background link kernel's workgroups sleep before every tile while that count is non-zero
 *     (at most GEC_BG_LINK_WAIT_US per workgroup and launch): the link is the one resource the classes cannot be
 *     given halves of;
 *   - it keeps to two staging-copy threads.
 * libgarage_block runs gbm_scrub_all / gbm_resync_run on such a sibling.  Results are identical. */
enum { GEC_CLASS_FOREGROUND = 0, GEC_CLASS_BACKGROUND = 1 };
int gec_codec_background(const gec_codec *c, gec_codec **out);
int gec_codec_class(const gec_codec *c);   /* GEC_CLASS_* */
int gec_codec_backend(const gec_codec *c); /* GEC_BACKEND_CPU or GEC_BACKEND_HIP (AUTO is resolved at creation) */
/* times a background chunk found foreground work in flight on `device` and waited (process-wide counter) */
uint64_t gec_qos_yields(int device);
/* Whether this process's staging slots run on CU-masked streams (hipExtStreamCreateWithCUMask): -1 = no host-pointer
 * path that wants them has run yet, 0 = the runtime / partition mode refused them (everything shares the CUs; a
 * background codec then only has its low stream priority, its small chunks and its yields), 1 = masks for the link
 * kernels, 2 = masks and the foreground / background partition (GEC_BG_CUS > 0). */
int gec_cu_masks_active(void);
/* Must not run concurrently with any other call on the same codec, and only after
 * work enqueued by *_dev calls on caller streams has completed. */
void gec_codec_destroy(gec_codec *c);
int gec_codec_k(const gec_codec *c);
int gec_codec_m(const gec_codec *c);
int gec_codec_device(const gec_codec *c); /* -1 for a CPU codec */
/* m x k parity rows of this codec (test introspection). */
int gec_parity_matrix(const gec_codec *c, uint8_t *out_m_by_k);
/* Number of decode matrices currently cached / total inversions performed
 * (the crate keeps an LRU of decode matrices per erasure pattern [EXT]). */
int gec_codec_cache_stats(const gec_codec *c, uint64_t *cached,
			  uint64_t *inversions);

/* ------------------------------------------------ host-pointer entry points
 * What the Rust shim calls.  Caller owns every buffer for the duration of the
 * call; the library keeps no pointer after return. */

/* Pinned (page-locked, DMA-able) host memory.  The host-pointer entry points below accept ANY
 * memory; when every buffer of a gec_encode_batch / gec_reconstruct_batch call lies inside a range
 * obtained here (and is 16-byte aligned) the kernel reads the shards out of the caller's memory and
 * writes its results into it over PCIe -- no host copy, no staging in device memory, one launch
 * (50 GiB/s of payload on a Gen5 x16 link, 39 us for one 1 MiB block) --
 * while ordinary pageable buffers are first copied through the library's own pinned staging slots.  The Rust
 * shim would draw the block buffers PutObject fills (src/api/s3/put.rs:440-456) and the parity
 * buffers from a pool allocated with gec_host_alloc, or register its existing arena once.
 * Process-wide, thread-safe, usable with every device's codec. */
/* (On a host without a usable device gec_host_alloc hands out page-aligned ordinary memory, so that callers
 * that draw their buffers from it keep working over a GEC_BACKEND_CPU codec; gec_host_is_pinned says 0.) */
void *gec_host_alloc(size_t bytes);   /* NULL on failure (gec_last_error) */
void gec_host_free(void *p);
int gec_host_register(void *p, size_t bytes);   /* pin caller-owned memory (hipHostRegister) */
int gec_host_unregister(void *p);
int gec_host_is_pinned(const void *p, size_t bytes); /* 1 if [p, p+bytes) lies inside one such range */

/* NUMA placement of a codec's host side.  No reference counterpart (Garage has no device; the nearest it has is one set of
 * resources per use: the data-dir shards of /root/reference/src/block/manager.rs:679-689, the RAM buffer permits of :156 and
 * :380-385).  The deployable paths are host-fed: a device's DMA engines and link kernels read pinned host memory that the
 * codec's copy threads (and the caller's) fill, and on a two-socket node half the GPUs hang off each socket.  A HIP codec
 * therefore resolves its device's memory node once, at creation (PCI address -> /sys/bus/pci/devices/<bdf>/numa_node), runs its
 * copy threads on that node's CPUs and binds its pinned staging slots to it; GEC_NUMA=0 switches all of it off.
 *   gec_codec_numa_node   that node; -1 = a CPU codec, a box with one node, an unknown topology, or GEC_NUMA=0
 *   gec_codec_numa_cpus   the node's CPUs (*count = how many; at most cap are written)
 *   gec_host_alloc_near   gec_host_alloc with the pages bound to the codec's node whatever device the calling thread has
 *                         current (free with gec_host_free).  What a lane's shard buffers should come from.
 *   gec_numa_bind_thread  restricts the CALLING thread to the codec's node's CPUs (a host that dedicates threads to a lane --
 *                         libgarage_block's pool, batcher workers -- calls it once per thread); 1 = bound, 0 = nothing done
 *   gec_numa_node_of      the node the page at p is on right now (move_pages in query mode); -1 = not resident / not allowed */
int gec_codec_numa_node(const gec_codec *c);
int gec_codec_numa_cpus(const gec_codec *c, size_t cap, int *cpus, size_t *count);
void *gec_host_alloc_near(const gec_codec *c, size_t bytes);
int gec_numa_bind_thread(const gec_codec *c);
int gec_numa_node_of(const void *p);

/* == ReedSolomon::encode_sep(&data, &mut parity) [EXT], batched.
 * Replaces the "clone the same Bytes to rf nodes" fan-out payload of
 * rpc_put_block (src/block/manager.rs:387-405; src/rpc/rpc_helper.rs:493):
 * blocks[b] is block b's payload (block_len[b] bytes); data shard i is bytes
 * [i*S,(i+1)*S) zero-extended; parity[b] receives m consecutive shards of S
 * bytes.  S must be a multiple of 64 and >= gec_shard_len(k, block_len[b]). */
int gec_encode_batch(const gec_codec *c, size_t nblocks,
		     const uint8_t *const *blocks, const size_t *block_len,
		     size_t S, uint8_t *const *parity);

/* == ReedSolomon::verify(&shards) [EXT], batched; the scrub-worker check that
 * replaces DataBlock::verify's blake2 compare for shards
 * (src/block/block.rs:69-83, src/block/repair.rs:450-458).
 * shards[b*(k+m) + j] = shard j of block b (S bytes); ok[b] = 1 iff consistent. */
int gec_verify_batch(const gec_codec *c, size_t nblocks,
		     const uint8_t *const *shards, size_t S, uint8_t *ok);

/* The scrub worker's whole check in ONE trip (src/block/repair.rs:450-458 reads a block and compares its hash
 * with its name; a stripe has k+m shards to check and a code to check them against): ok[b] as gec_verify_batch,
 * and shard_sums[32*(b*(k+m) + j)] = the shard checksum (gec_shardsum_batch) of every shard, for the caller to
 * compare with the shard headers.  With all shards in pinned memory every byte crosses the link once: the
 * compare form of the pointer-table kernel reads the shards, leaves verdicts in pinned memory and the shards in
 * device memory, where the checksums are computed chunk by chunk beside the next chunk's transfer. */
int gec_verify_hash_batch(const gec_codec *c, size_t nblocks,
			  const uint8_t *const *shards, size_t S, uint8_t *ok,
			  uint8_t *shard_sums);

/* == ReedSolomon::reconstruct / reconstruct_data [EXT], batched; sits where
 * rpc_get_raw_block_internal returns the first whole block it finds
 * (src/block/manager.rs:292-334) and where resync_block fetches an absent
 * block (src/block/resync.rs:485-499).
 * shards[b*n + j] == NULL => shard j of block b is missing and is written to
 * out[b*n + j] (S bytes) -- unless that entry is NULL too, which means "not wanted":
 * only the rows that are asked for are computed (resync rebuilds just the shards
 * that are absent on reachable nodes).  out entries of present shards are ignored;
 * with data_only != 0 missing parity shards are never rebuilt.
 * Fewer than k present shards in any block => GEC_E_TOO_FEW_PRESENT. */
int gec_reconstruct_batch(const gec_codec *c, size_t nblocks,
			  const uint8_t *const *shards, uint8_t *const *out,
			  size_t S, int data_only);

/* gec_reconstruct_batch plus, from the same trip, the shard checksums (gec_shardsum_batch) of the k shards that
 * were READ for every block (in_sums[32*(b*n + j)] for the first k present shards j; other entries untouched) and
 * of the shards that were WRITTEN (out_sums[32*(b*n + j)]).  A block of which nothing is wanted (every out entry NULL, or
 * only entries of shards that are present) is not read at all: its in_sums entries are untouched too -- a caller must not
 * compare them with anything.  The resync worker's rebuild
 * (src/block/resync.rs:485-499, "fetching absent but needed block") in one pass over the link: the caller
 * compares in_sums with the shard headers it read and stamps out_sums into the headers it writes.  Pinned
 * buffers: one pointer-table kernel per erasure pattern that also mirrors what it reads and writes into device
 * memory, where the checksums are computed beside the next chunk's transfer. */
int gec_reconstruct_hash_batch(const gec_codec *c, size_t nblocks,
			       const uint8_t *const *shards, uint8_t *const *out,
			       size_t S, int data_only, uint8_t *in_sums, uint8_t *out_sums);

/* --------------------------------------------- device-resident entry points
 * Used by bench.py (no PCIe in the timed region) and by callers that already
 * hold blocks in HBM.  All device pointers must be 16-byte aligned, strides
 * multiples of 16, S a multiple of 64; `hip_stream` is a hipStream_t (NULL =
 * default stream).  Asynchronous: returns after enqueueing. */

/* data shard i of block b at d_data + b*data_stride + i*S; parity r of block b
 * at d_parity + b*parity_stride + r*S. */
int gec_encode_batch_dev(const gec_codec *c, size_t nblocks, const void *d_data,
			 size_t data_stride, size_t S, void *d_parity,
			 size_t parity_stride, void *hip_stream);

/* stripe layout: shard j (0..k+m) of block b at d_stripes + b*stride + j*S.
 * d_bad[b] (uint32) is set to 0 if parity is consistent, else non-zero. */
int gec_verify_batch_dev(const gec_codec *c, size_t nblocks,
			 const void *d_stripes, size_t stride, size_t S,
			 uint32_t *d_bad, void *hip_stream);

/* One erasure pattern for the whole batch: present[k+m] (host, 0/1).  Missing
 * shards are rebuilt in place inside each stripe.  Host-side decode-matrix
 * inversion happens here (cached per pattern). */
int gec_reconstruct_batch_dev(const gec_codec *c, size_t nblocks,
			      void *d_stripes, size_t stride, size_t S,
			      const uint8_t *present, int data_only,
			      void *hip_stream);

/* A pattern PER BLOCK: present[b*(k+m) + j] (host, 0/1) -- ReedSolomon::reconstruct is per call [EXT], and a device-resident
 * batch gathered from a degraded cluster holds a mix.  The blocks are grouped by pattern on the host (one cached decode
 * matrix per distinct pattern) and rebuilt by ONE launch in which every workgroup expands the coefficient table of the
 * block it is at.  Every block must have >= k shards present (GEC_E_TOO_FEW_PRESENT names the first that has not, before
 * anything is enqueued).  With a GEC_BACKEND_CPU codec the same call works on HOST memory at d_stripes. */
int gec_reconstruct_batch_dev_ex(const gec_codec *c, size_t nblocks, void *d_stripes, size_t stride, size_t S,
				 const uint8_t *present, int data_only, void *hip_stream);

/* Same, restricted to bytes [byte_off, byte_off+byte_len) of every shard
 * (both multiples of 16): after the all-gather of a striped object each GPU
 * rebuilds its 1/n byte-range of every missing shard. */
int gec_reconstruct_range_dev(const gec_codec *c, size_t nblocks,
			      void *d_stripes, size_t stride, size_t S,
			      const uint8_t *present, int data_only,
			      size_t byte_off, size_t byte_len,
			      void *hip_stream);

/* Most general device form: shard j of block b lives at
 *   d_base + b*block_stride + shard_off[j]        (shard_off: host array, k+m entries)
 * so the shards of one stripe need not be adjacent.  This is what decodes the
 * output of the all-gather of a striped object in place: with the gathered
 * buffer laid out [rank][object][slot][S], shard j = slot j/N of rank j%N has
 * shard_off[j] = (j%N)*nobjects*slots*S + (j/N)*S and block_stride = slots*S.
 * All offsets/strides multiples of 16; rebuilt shards are written to their own
 * (shard_off) positions.
 * With a GEC_BACKEND_CPU codec the same call (and gec_reconstruct_range_dev / _batch_dev) works on HOST memory at
 * d_base -- the strided layout is the call's shape, not a property of device memory; hip_stream is ignored and the
 * call returns when the shards are rebuilt.  (Encode, verify and the checksums of a CPU codec go through the
 * host-pointer entry points.) */
int gec_reconstruct_scattered_dev(const gec_codec *c, size_t nblocks,
				  void *d_base, size_t block_stride,
				  const size_t *shard_off, size_t S,
				  const uint8_t *present, int data_only,
				  size_t byte_off, size_t byte_len,
				  void *hip_stream);

/* ------------------------------------------- multi-GPU decode of a striped object
 * SURVEY.md section 8b ("gec_group_create/..._allgather_decode") and 8e, BASELINE
 * config 5.  One process (or thread) per GPU; rank r of N owns shard j of every
 * object iff j % N == r, in slot j / N of its slot buffer
 *     d_local_slots  [nobjects][slots][S],   slots = ceil((k+m) / N)
 * (slots whose shard is erased, or padding slots, may hold anything).  Decode is
 * the path's one real exchange step:
 *   1. ONE all-gather of the slot buffers into d_gathered [N][nobjects][slots][S]
 *      (ncclAllGather, ncclUint8 -- RCCL over xGMI);
 *   2. every rank rebuilds ITS 1/N byte range [16*floor(c*r/N), 16*floor(c*(r+1)/N)),
 *      c = S/16, of every missing shard in place in d_gathered
 *      (gec_reconstruct_scattered_dev: no permute copy);
 *   3. complete != 0: a second all-gather of just the rebuilt ranges, so that
 *      every rank ends up with every missing shard in full.
 * Whole blocks never come through here: they are independent units, partitioned
 * over GPUs by hash with no collective (the way Garage partitions,
 * src/rpc/layout/version.rs:101-104).
 *
 * RCCL is resolved at run time (librccl.so.1, or $GEC_RCCL_LIB), so the library
 * loads on hosts without it; the caller carries rank 0's unique id to the other
 * ranks out of band (Garage: its own RPC; the tests: torch.distributed / a file).
 * A group serves one call at a time; everything is enqueued on `hip_stream`. */
typedef struct gec_group gec_group;
#define GEC_GROUP_ID_BYTES 128 /* sizeof(ncclUniqueId) */

int gec_group_unique_id(uint8_t id[GEC_GROUP_ID_BYTES]);
/* Collective: all nranks ranks call it with the same id (ncclCommInitRank on the
 * codec's device).  The codec is borrowed and must outlive the group. */
int gec_group_create(const gec_codec *c, int rank, int nranks,
		     const uint8_t id[GEC_GROUP_ID_BYTES], gec_group **out);

/* Bring-your-own transport instead of RCCL (another fabric, or N logical ranks on
 * one device in the tests).  all_gather must deliver, for every rank q, rank q's
 * `bytes` at d_recv + q*bytes on every rank, ordered with respect to `hip_stream`
 * (it may enqueue asynchronously on it or block); 0 = success.
 * Over a GEC_BACKEND_CPU codec such a group runs the same steps on HOST buffers (d_local_slots, d_gathered,
 * d_rebuilt and the transport's pointers are host memory, hip_stream is NULL / ignored, the calls block): a rank
 * that has lost its GPU stays in the group, and ranks of different backends interoperate as long as the transport
 * moves bytes between their buffers.  (gec_group_create, RCCL, needs a HIP codec.) */
typedef int (*gec_allgather_fn)(void *ctx, const void *d_send, void *d_recv,
				size_t bytes, void *hip_stream);
int gec_group_create_with_transport(const gec_codec *c, int rank, int nranks,
				    gec_allgather_fn all_gather, void *ctx,
				    gec_group **out);
void gec_group_destroy(gec_group *g);
int gec_group_rank(const gec_group *g);
int gec_group_size(const gec_group *g);
size_t gec_group_slots(const gec_group *g);

/* present[k+m] (host, 0/1) is the erasure pattern, identical on all ranks.
 * d_gathered: caller-allocated, N*nobjects*slots*S bytes, 16-byte aligned; on
 * return (stream order) shard j of object b is at
 *     d_gathered + ((j % N)*nobjects + b)*slots*S + (j / N)*S.
 * With complete == 0 only this rank's byte range of each missing shard is valid. */
int gec_group_allgather_decode(gec_group *g, size_t nobjects,
			       const void *d_local_slots, size_t S,
			       const uint8_t *present, int data_only, int complete,
			       void *d_gathered, void *hip_stream);

/* All-to-all variant of the exchange.  After the all-gather every rank holds EVERY survivor slot (7/8 of them
 * arrive over xGMI), although it only reads its own 1/N byte range of the k shards the decode uses.  Here each
 * rank sends every peer just that peer's range of its own valid shards (grouped ncclSend/ncclRecv over the xGMI
 * full mesh, or the caller's gec_alltoall_fn), rebuilds its range of the missing shards, and -- with
 * complete != 0 -- the rebuilt ranges are all-gathered as before.  Bytes received per rank:
 * k*S*nobjects*(N-1)/N^2 instead of slots*S*nobjects*(N-1): 11x less at BASELINE config 5 (N = 8, RS(20,8)).
 * The result is NOT the gathered stripe buffer (survivors stay where they are): d_rebuilt receives the missing
 * shards only, [nmiss][nobjects][S] in ascending shard-index order of the missing (data_only: missing data) shards,
 * complete on every rank (complete != 0) or valid in this rank's byte range only.
 * all-gather stays the default exchange of the project brief (gec_group_allgather_decode). */
typedef int (*gec_alltoall_fn)(void *ctx, const void *d_send, void *d_recv,
			       size_t bytes_per_peer, void *hip_stream);
/* like gec_group_create_with_transport, with an all-to-all as well: rank r's `bytes_per_peer` at
 * d_send + q*bytes_per_peer must arrive at rank q's d_recv + r*bytes_per_peer (all_to_all may be NULL). */
int gec_group_create_with_transport2(const gec_codec *c, int rank, int nranks,
				     gec_allgather_fn all_gather, gec_alltoall_fn all_to_all,
				     void *ctx, gec_group **out);
int gec_group_alltoall_decode(gec_group *g, size_t nobjects,
			      const void *d_local_slots, size_t S,
			      const uint8_t *present, int data_only, int complete,
			      void *d_rebuilt, void *hip_stream);
/* bytes this rank received from other ranks during the last *_decode call on the group (every exchange; for the peer form:
 * bytes its kernel read out of other ranks' memory, plus the rebuilt ranges) */
uint64_t gec_group_bytes_exchanged(const gec_group *g);

/* PEER-POINTER variant: no exchange buffer at all.  On MI355X every GPU addresses every other GPU's memory over the xGMI
 * full mesh, and the decode kernel already takes its inputs as pointer tables -- so each rank builds ONE table whose inputs
 * are ITS byte range of the k shards the decode reads, WHEREVER they live (slot v/N of rank v%N's slot buffer), and rebuilds
 * its range of every missing shard in one launch: no pack, no unpack, no RCCL staging, the same 1/N of the all-gather's
 * bytes per link as the all-to-all form.  d_peer_slots[q] = rank q's slot buffer [nobjects][slots][S] as an address THIS
 * process can use on the codec's device ([rank] = its own): a plain pointer when the ranks are threads of one process (peer
 * access is enabled here on first use), or what gec_ipc_open returned for the handle rank q made with gec_ipc_export
 * (hipIpcGetMemHandle / hipIpcOpenMemHandle) when they are processes.  The group's transport carries only two 16-byte
 * barriers (every slot buffer is final before a peer reads it; nobody's changes while a peer still does) and -- with
 * complete != 0 -- the rebuilt ranges, exactly as in the all-to-all form, whose result layout this call shares:
 * d_rebuilt [nmiss][nobjects][S].  Over a GEC_BACKEND_CPU codec the same call runs on host pointers. */
#define GEC_IPC_HANDLE_BYTES 72 /* hipIpcMemHandle_t of the allocation + the pointer's offset inside it */
int gec_ipc_export(const void *d_ptr, uint8_t handle[GEC_IPC_HANDLE_BYTES]);
int gec_ipc_open(const uint8_t handle[GEC_IPC_HANDLE_BYTES], int device, void **d_ptr);
int gec_ipc_close(void *d_ptr);
int gec_group_peer_decode(gec_group *g, size_t nobjects, const void *const *d_peer_slots, size_t S,
			  const uint8_t *present, int data_only, int complete,
			  void *d_rebuilt, void *hip_stream);

/* ------------------------------------------------------- blake2sum on the GPU
 * SURVEY.md section 8 row f4.  Garage's content hash `blake2sum` = blake2b-512
 * truncated to 32 bytes (src/util/data.rs:130-138); today it is a CPU pass in
 * spawn_blocking (src/api/s3/put.rs:440-456, src/block/block.rs:69-77).  Shards
 * are not self-verifying by name like blocks are, so every shard carries such a
 * checksum; computing it where the shards already are removes the CPU pass
 * (~1 GiB/s/core) that otherwise dominates put/get once encode runs at TB/s. */

/* n messages of `len` bytes, message i at d_base + i*stride (stride multiple of
 * 16, base 16-byte aligned); d_out receives 32 bytes per message.  Async. */
int gec_blake2sum_batch_dev(const gec_codec *c, size_t n, const void *d_base,
			    size_t stride, size_t len, void *d_out,
			    void *hip_stream);

/* Host buffers of arbitrary lengths; out = n*32 bytes.  Blocking. */
int gec_blake2sum_batch(const gec_codec *c, size_t n,
			const uint8_t *const *msgs, const size_t *lens,
			uint8_t *out);

/* SHARD CHECKSUMS ("shardsum").  Shards are this project's own storage format (Garage has none).
 *
 * GEC_SHARDSUM_MLH64 (version 3, the default; garage_amd/csrc/mlh64.hpp has the rationale, oracle/mlh64.py an independent
 * restatement).  All integers little-endian:
 *   K[i]  = (uint32)(splitmix64(0x6761726167654d4c + (i+1) * 0x9E3779B97F4A7C15) >> 32) | 1,            i < 1024
 *   s_l   = SUM_{i<1024} K[i] * u32(shard[4096 l + 4 i ..+4])  mod 2^64      (leaf l; the shard zero-extended)
 *   sum   = blake2b-512("GECSUM3\0" || u64(len) || u64(s_0) || ... || u64(s_{ceil(len/4096)-1}))[0..32]
 * Why: a leaf sum is a sum of per-word terms, so the RS kernels -- which already hold 16 bytes of every shard per lane --
 * add their terms with four multiply-accumulates and the checksum costs no second pass over the stripe (the encode with
 * all 14 checksums of RS(10,4): one kernel + one tiny root kernel, instead of 5.4x the encode's time); a host core checks
 * a shard at memory speed, so a healthy get does not cross the link at all.  Every corruption inside one 32-bit word is
 * detected, random corruption of a leaf with probability 1 - 2^-64; it is NOT a MAC (public keys) -- integrity against
 * an adversary is the job of the block's own name, Garage's blake2sum, exactly as the reference lets zstd's frame checksum
 * stand in for a compressed block's verify (src/block/block.rs:69-83).
 *
 * GEC_SHARDSUM_BLAKE2B_TREE (version 2): BLAKE2b in its standard TREE mode (BLAKE2 specification section 2.10, parameter block of RFC 7693
 * section 2.5): leaves of GEC_SHARDSUM_LEAF bytes, unlimited fanout, depth 2, 64-byte inner digests, the root's
 * 64-byte digest truncated to 32 bytes like blake2sum.  Python's hashlib reproduces it:
 *   leaf i = blake2b(shard[i*4096:(i+1)*4096], digest_size=64, fanout=0, depth=2, leaf_size=4096, node_offset=i,
 *                    node_depth=0, inner_size=64, last_node=(i == nleaves-1)).digest()
 *   sum    = blake2b(b"".join(leaves), digest_size=64, fanout=0, depth=2, leaf_size=4096, node_offset=0,
 *                    node_depth=1, inner_size=64, last_node=True).digest()[:32]
 * Why a tree: one BLAKE2b message is one serial dependency chain, and on MI355X a lone wave issues one VALU
 * instruction per ~5 cycles -- the 14336 shards of a 1024-block batch take 1.4 ms as plain BLAKE2b however the
 * kernel is written (profiles/r02_valu_probe.txt), 5x the RS encode beside them.  26 independent leaves per
 * shard fill the machine.  Block NAMES remain plain blake2sum: they are Garage's (src/util/data.rs:130-138). */
#define GEC_SHARDSUM_LEAF 4096
/* n shards of `len` bytes, shard i at d_base + i*stride; d_out receives 32 bytes each (the codec's kind).  Async. */
int gec_shardsum_batch_dev(const gec_codec *c, size_t n, const void *d_base,
			   size_t stride, size_t len, void *d_out,
			   void *hip_stream);
/* Host buffers of arbitrary lengths (16-byte aligned pinned buffers are read in place); out = n*32 bytes. */
int gec_shardsum_batch(const gec_codec *c, size_t n,
		       const uint8_t *const *msgs, const size_t *lens,
		       uint8_t *out);

/* gec_encode_batch + the shardsum (of the codec's kind) of all k+m shards of every block, computed on
 * the device while the stripe is resident: shard_sums[(b*(k+m) + j)*32 ..] is the
 * checksum of shard j of block b (data shards as zero-extended to S bytes).
 * Kind 3 (MLH64): the kernel that reads the data shards and writes the parity accumulates the leaf sums of all of them from
 * its registers -- pinned buffers: ONE link kernel + one root kernel, nothing mirrored in HBM, no second pass. */
int gec_encode_hash_batch(const gec_codec *c, size_t nblocks,
			  const uint8_t *const *blocks, const size_t *block_len,
			  size_t S, uint8_t *const *parity, uint8_t *shard_sums);

/* The read path in ONE trip to the device -- the twin of gec_encode_hash_batch, for what
 * rpc_get_raw_block_internal does after gathering (src/block/manager.rs:292-334) plus the verify that
 * read_block_from does on the serving node (:577-609):
 *   - shards[b*n + j] (S bytes each, NULL = not in hand; >= k per block, else GEC_E_TOO_FEW_PRESENT): the first
 *     k present shards of every block are uploaded (the crate's rule);
 *   - shard_sums[(b*n + j)*32] receives the shardsum of every shard that was uploaded (entries of the others
 *     are left untouched) -- the caller compares them with the checksums in the shard headers;
 *   - missing DATA shards are rebuilt into rebuilt[b*n + j] (S bytes; must be non-NULL exactly for those);
 *   - block_sums (may be NULL): blake2sum of the first block_len[b] bytes of block b's data area, i.e. of the
 *     block itself (DataBlock::verify's content-against-name check, src/block/block.rs:69-77).
 * Every checksum kernel runs ONCE over the whole batch (a BLAKE2b chain costs the same however many messages
 * run beside it), the shard checksums beside the decode on their own stream.  Buffers inside pinned ranges
 * (gec_host_alloc / gec_host_register) are read and written by copy kernels directly, in k stages (data slot s of
 * every block), and the block checksums of blocks that need no decode advance behind the stages: their chains -- the
 * longest thing in the call, ~13 ms per MiB of block -- start while the blocks are still arriving.  Other buffers are
 * staged through pinned pieces.  (One block's chain on a host core takes ~1 ms per MiB: for a handful of blocks pass
 * block_sums = NULL and hash on the host, as libgarage_block does below 128 blocks.) */
int gec_decode_verify_batch(const gec_codec *c, size_t nblocks,
			    const uint8_t *const *shards, size_t S,
			    const size_t *block_len, uint8_t *const *rebuilt,
			    uint8_t *shard_sums, uint8_t *block_sums);

/* Device-resident form: stripes already in HBM (shard j of block b at d_stripes + b*stride + j*S), parity
 * written in place, d_sums (16-byte aligned device memory, nblocks*(k+m)*32 bytes) receives the checksums.
 * Asynchronous: everything is ordered behind / ahead of the work on `hip_stream`.  Kind 3: the encode kernel's SUM form +
 * the root kernel (BASELINE config 2: 0.29 ms against 0.25 for the encode alone).  Kind 2: the checksums of the k data shards
 * run on a second stream beside the RS kernel (they do not depend on it), the m parity checksums follow it (1.3 ms). */
int gec_encode_hash_batch_dev(const gec_codec *c, size_t nblocks, void *d_stripes,
			      size_t stride, size_t S, void *d_sums,
			      void *hip_stream);

/* Introspection for tests (no device needed): what the host would decide for ONE launch of the
 * default kernel with k input shards and rows_left output rows still to produce -- how many rows
 * the launch takes, the table-entry width (4/8/16 bytes), loads per batch, workgroup size and
 * dynamic LDS bytes.  Lets a CPU-only test sweep every (k, rows) for the invariants a launch
 * failure would otherwise only reveal on a GPU: LDS <= 64 KiB, >= 1 row of progress, the
 * 16-row form only where its coefficients fit the argument block. */
int gec_launch_geometry(int k, int rows_left, int *rows, int *entry_bytes,
			int *loads_per_batch, int *threads, size_t *lds_bytes);

/* Kernel selection for A/B measurements and tests, process-wide.  0 = default (nibble product tables in LDS; every
 * other choice by size), 1 = the RS kernel with log/antilog tables in LDS (the literal north_star formulation, kept as
 * the measured baseline: bench.py --variant 1).  Test routes -- they pick between kernels the default already uses at
 * different sizes, so that one input reaches both: 2 / 3 = the BLAKE2b kernels with one lane / four lanes per message,
 * 4 = small pinned trips of a checksum-v2 codec take the streaming kernels instead of the one-launch kernel, 5 = gec_group_peer_decode
 * always through pointer tables (by default only when the peers' buffers are more than 64 GiB apart in this process's address space). */
int gec_set_kernel_variant(int variant);
int gec_get_kernel_variant(void);

#ifdef __cplusplus
}
#endif
#endif /* GARAGE_EC_H */
