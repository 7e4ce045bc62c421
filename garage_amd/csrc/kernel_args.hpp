// kernel_args.hpp -- kernel argument blocks and launch limits of libgarage_ec's device code.
//
// Plain structs and constants only (no HIP types): kernels.hpp / blake2b.hpp (device code, compiled once, in
// ec_hip_launch.hip) and the host-only translation units that fill these blocks in (ec_hip_*.cpp) both include
// this file.  What each field means to the kernel is documented where the kernel is (kernels.hpp, blake2b.hpp).
#pragma once

#include <stdint.h>

namespace gec {

constexpr uint32_t SHARDSUM_LEAF_BYTES = 4096;  // == SHARDSUM_LEAF below (needed before it is declared)
constexpr int KMAX = 256;    // input shards per launch (k + m <= 256 => k <= 255)
constexpr int RMAX = 8;      // output rows per launch with 4- and 8-byte table entries
constexpr int RMAX16 = 16;   // ... with 16-byte entries (MW = 4): k <= K16MAX only
constexpr int K16MAX = 120;  // 16 coefficient bytes per input shard must fit ApplyArgs.coef, the tables 64 KiB of LDS
constexpr int BLOCK = 256;   // threads per workgroup of the baseline kernel
constexpr int MODE_STORE = 0, MODE_COMPARE = 2;  // write the rows / compare them with what is stored
// compare, with the stored rows requested up front behind the data loads (instead of at the end of the tile, where
// their latency is exposed: verify is a pure read stream).  Only where every one of the 4 row slots is a real row
// (rows == 4, 4-byte table entries): RS(10,4) verify 250 -> 232 us = 75 -> 81 % of peak; with fewer rows the
// index-clamped duplicate loads cost more than they hide (RS(3,1): -6 %), with 8-byte entries the 32 extra VGPRs
// cost occupancy (RS(20,8): -7 %).  tools/verify_bench.py.
constexpr int MODE_COMPARE_PF = 3;

struct ApplyArgs {
	const uint8_t *in;   // input stripes base
	uint8_t *out;        // output base (may alias `in`: rows never overlap inputs)
	uint32_t *bad;       // MODE_COMPARE: bad[b] |= 1 on mismatch
	uint64_t in_stride;  // bytes between consecutive blocks
	uint64_t out_stride;
	uint32_t col0;       // first 16-byte column of every shard to process
	uint32_t cols;       // number of 16-byte columns to process
	uint32_t nblocks;
	uint32_t tiles_per_block;  // baseline kernel only
	uint32_t total_cols;       // nblocks * cols (< 2^32: the host splits larger batches by blocks)
	uint32_t k;          // inputs  (<= KMAX)
	uint32_t rows;       // outputs (<= RMAX)
	uint32_t in_off[KMAX];   // shard offsets inside a block, in 16-byte units
	uint32_t out_off[RMAX16];
	// coef[t][r] = mat[r][t]: one 8-byte row per input shard.  The 16-row kernel (MW = 4) reads the
	// same bytes as a flat [k][16] array (k <= K16MAX).
	uint8_t coef[KMAX][RMAX];
	// SUM forms only (shard checksum v3, mlh64_dev.hpp): leaf sums of block b go to
	//   lsum[((b * sum_slots_total + sum_slot0 + slot) * sum_nleaf_max) + leaf],   slot = input t (sum_inputs) then row r
	uint64_t *lsum;
	uint32_t sum_nleaf_max, sum_slots_total, sum_slot0, sum_inputs;
	// PAT form only: per-block coefficient sets.  Block b uses entry pat[b] of pat_tab (pat_stride bytes apart, 16-byte
	// aligned): [in_off kp x u32][out_off RMAX x u32][rows, 3 x pad u32][coef k x RMAX bytes], kp = k rounded up to 4
	const uint8_t *pat_tab;
	const uint16_t *pat;
	uint32_t pat_stride;
};

// exp[512] | log[256], filled by the host from gec::Field (768 bytes).
struct LogExp {
	uint8_t exp[512];
	uint8_t log[256];
};

constexpr int PTR_KMAX = 128;  // coef[PTR_KMAX][RMAX] keeps the kernel argument block at 1 KiB

// Workgroups (of 256 lanes = 4 waves, one per SIMD) of a tile-walking kernel that are resident on one CU at once:
// promised by the kernels' __launch_bounds__(256, RESIDENT_WGS) (second argument = waves per SIMD the register
// allocation must leave room for) and used by the host to size grids that FIT (ec_hip_launch.hip, resident_grid).
constexpr int RESIDENT_WGS = 6;

enum : uint32_t { LINK_NONE = 0, LINK_SIGNAL = 1, LINK_YIELD = 2 };

struct PtrApplyArgs {
	const uint8_t *const *in;  // [nblocks][k]: 16-byte aligned shard pointers
	const uint32_t *in_valid;  // [nblocks][k]: bytes of the shard that exist (<= 16*cols)
	uint8_t *const *out;       // [nblocks][rows]
	uint32_t cols;             // 16-byte columns per shard
	uint32_t k, rows;
	uint32_t tiles_x, tiles_total;  // 256-column tiles per shard; tiles_x * (blocks of this launch)
	// The link is the one thing the two classes cannot be given halves of.  link_busy counts the workgroups of
	// FOREGROUND link kernels that are running on the device (LINK_SIGNAL: +1 on entry, -1 on exit); a BACKGROUND link
	// kernel (LINK_YIELD) looks at it before every tile and sleeps while it is non-zero -- for at most
	// link_wait_ticks (10 ns each) per workgroup and launch, so that it always finishes.
	uint32_t *link_busy;
	uint32_t link_role, link_wait_ticks;
	// Per-block coefficient sets (one decode launch for several erasure patterns): pat != NULL -- block b uses set pat[b],
	// set p = coef_tab + p * k * RMAX bytes laid out [t][r] like `coef` (which is then unused); a workgroup rebuilds its
	// tables when the pattern changes between two of its tiles (the host lays blocks out pattern by pattern); a NULL
	// out[b][r] means block b's pattern has no row r
	const uint8_t *coef_tab;
	const uint16_t *pat;
	// pace_ticks > 0 (10 ns each): a workgroup starts its i-th tile no earlier than i * pace_ticks after its first -- a
	// BACKGROUND codec's rows on their way into host memory (resync's rebuilt shards) must not fill the fabric's queues
	// towards the link with posted writes: every load of the request path waits behind them (GEC_BG_HOME_RATE_GBPS)
	uint32_t pace_ticks;
	// MIRROR: everything the kernel reads and computes is also laid down in device memory, dense --
	// mirror + b*mirror_stride + t*16*cols for input shard t (first row group only: mirror_inputs),
	// ... + mirror_row0 + r*16*cols for output row r -- so that the shard checksums can be computed from
	// HBM while the bytes cross the link only once (gec_encode_hash_batch on pinned memory)
	uint8_t *mirror;
	uint64_t mirror_stride, mirror_row0;
	uint32_t mirror_inputs;
	// COMPARE: the rows are compared with what out[b][r] holds instead of being stored; bad[b] = 1 on a mismatch
	// (bad may itself be pinned host memory: the flags need no copy back)
	uint32_t *bad;
	// SUM forms only (shard checksum v3): like ApplyArgs -- leaf sums of block b go to
	//   lsum[((b * sum_slots_total + sum_slot0 + slot) * sum_nleaf_max) + leaf],  slot = input t (sum_inputs) then row r
	uint64_t *lsum;
	uint32_t sum_nleaf_max, sum_slots_total, sum_slot0, sum_inputs;
	uint8_t coef[PTR_KMAX][RMAX];
};

// gf_ptrs_hash (fused.hpp): ONE launch for a small trip -- the product over pointer tables like gf_apply_ptrs, and the
// tree-mode shard checksums of what it read (and, for a put, wrote) from the same tile while it is in LDS.
constexpr uint32_t FUSED_LEAF_PITCH = SHARDSUM_LEAF_BYTES + 32;  // LDS bytes between the leaves of a tile (bank spread)
constexpr int FUSED_MAX_LEAVES = 64;                             // leaves per tile: four waves x sixteen quads

struct FusedArgs {
	const uint8_t *const *in;  // [nblocks][k]
	const uint32_t *in_valid;  // [nblocks][k]
	uint8_t *const *out;       // [nblocks][rows]; a NULL entry: this block does not want that row
	uint32_t cols, k, rows;
	uint32_t tiles_x, tiles_total;  // 256-column tiles per shard (= leaves per shard); tiles_x * blocks
	uint32_t hash_rows;        // 1: the output rows are hashed too (a put: k + rows checksums per block), 0: the inputs only
	// per-block coefficient sets (one decode launch for several erasure patterns): block b uses set pat[b] (NULL: set 0);
	// set p = coef_tab + p * k * RMAX bytes, [t][r] like PtrApplyArgs::coef
	const uint8_t *coef_tab;
	const uint16_t *pat;
	uint8_t *leafdig;          // device scratch: [nblocks][nh][tiles_x][64], nh = k + (hash_rows ? rows : 0)
	uint32_t *done;            // device: [nblocks] tiles finished; all zero before the launch, zero again after it
	uint8_t *sums;             // [nblocks][nh][32] (may be pinned host memory)
	uint32_t *link_busy;
	uint32_t link_role, link_wait_ticks;
};

struct CopyEntry {
	const uint8_t *src;
	uint8_t *dst;
	uint64_t bytes;
};

struct RangeArgs {
	uint8_t *gathered;     // [rank][object][slot][S]
	uint8_t *packed;       // pack: send buffer, unpack: receive buffer
	uint64_t obj_stride;   // slots*S, bytes between consecutive objects of one rank
	uint32_t nobj, nmiss;
	uint32_t cols;         // S / 16
	uint32_t max_cols;
	uint32_t world, rank;
	uint64_t shard_off[KMAX];  // missing shard i of object 0, byte offset in `gathered`
};

struct A2aArgs {
	const uint8_t *local;  // [obj][slots][S]
	uint8_t *send;         // [peer][nvs_max][obj][max_cols]
	uint64_t obj_stride;   // slots*S
	uint32_t nobj, nvs, nvs_max;
	uint32_t cols, max_cols, world;
	uint32_t slot_of[KMAX];  // local slot of valid shard vs
};

struct RebuiltArgs {
	const uint8_t *packed;  // [rank][nmiss][obj][max_cols] (or just this rank's [nmiss][obj][max_cols] with world_in == 1)
	uint8_t *rebuilt;       // [nmiss][obj][S]
	uint32_t nobj, nmiss, cols, max_cols, world;
	uint32_t first_rank, nranks_in;  // ranks whose ranges `packed` holds: first_rank .. first_rank + nranks_in - 1
};

struct Blake2Args {
	const uint8_t *base;
	const uint64_t *off;   // per-message byte offset from base (NULL: computed, below)
	const uint64_t *len;   // per-message length (NULL: uniform_len)
	uint64_t stride;       // message i at base + i*stride, or with group != 0:
	uint64_t uniform_len;  //   base + (i / group)*group_stride + (i % group)*stride
	uint8_t *out;          // 32 bytes per message, at out + 32*i, or with group != 0:
	uint32_t n;            //   out + 32*((i / group)*out_group + i % group)
	uint32_t group;        // messages per group (e.g. the m parity shards of one stripe); 0 = flat
	uint64_t group_stride;
	uint32_t out_group;
	// Segmented hashing (blake2b_batch_quad only): this launch compresses blocks [seg_begin_blk, seg_end_blk) of
	// every message, picking the chaining value up from `state` (8 words per message) when it does not start at
	// block 0 and leaving it there when the message goes on beyond seg_end_blk; a message whose last block falls
	// inside the range is finished (digest written) by this launch, messages that ended earlier are skipped.
	// The whole-message form is seg_begin_blk = 0, seg_end_blk = ~0.  This is how a block's checksum chain --
	// serial, ~14 ms per MiB whatever runs beside it -- starts while the rest of the block is still on the link.
	uint64_t *state = nullptr;
	uint64_t seg_begin_blk = 0, seg_end_blk = ~0ull;
};

constexpr uint32_t SHARDSUM_LEAF = SHARDSUM_LEAF_BYTES;
constexpr uint64_t SHARDSUM_P0 = 64ull /*digest*/ | (0ull << 8) /*key*/ | (0ull << 16) /*fanout: unlimited*/ | (2ull << 24) /*depth*/ |
				 ((uint64_t)SHARDSUM_LEAF << 32);
constexpr uint64_t SHARDSUM_P2_LEAF = 0ull /*node_depth*/ | (64ull << 8) /*inner_length*/;
constexpr uint64_t SHARDSUM_P2_ROOT = 1ull | (64ull << 8);

}  // namespace gec
