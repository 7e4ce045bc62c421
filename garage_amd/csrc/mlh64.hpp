// mlh64.hpp -- the arithmetic of shard checksum v3 ("MLH64 tree"), shared by the device kernels (mlh64_dev.hpp), the host
// implementation below and -- as a specification -- oracle/mlh64.py (an independent restatement in Python integers).
//
// Why a second shard checksum.  v2 is BLAKE2b in tree mode: 15 lane-ops per byte, which makes the checksum kernel the one
// furthest below its roofline (0.21 of HBM peak) and 5x longer than the RS encode it sits beside; on a host core it runs at
// ~1 GB/s, so even a healthy get had to cross the link to have its shards checked.  Shards are THIS project's format
// (Garage has none), and the reference itself lets a non-cryptographic check stand in for a block's integrity where it
// is cheaper (DataBlock::verify on a compressed block only lets zstd's frame checksum speak, src/block/block.rs:69-83).
// The block's NAME stays Garage's blake2sum; only the per-shard check changes.
//
// Definition (all integers little-endian):
//   words   w_j = bytes [4j, 4j+4) of the shard as a u32, the shard zero-extended to a multiple of 4 bytes;
//   keys    K[i], i < 1024: K[i] = (splitmix64(MLH_SEED + (i+1) * 0x9E3779B97F4A7C15) >> 32) | 1     (public constants)
//   leaf l  covers bytes [4096 l, 4096 (l+1));  s_l = SUM_{i<1024} K[i] * w_{1024 l + i}  mod 2^64
//           (a 32x32->64 multiply-accumulate per word: the multilinear hash family of Carter-Wegman / Lemire-Kaser,
//            with per-position keys that repeat every leaf; the position of the leaf is bound by the root)
//   root    checksum = blake2b-512( "GECSUM3\0" || u64(len) || u64(s_0) || ... || u64(s_{nleaf-1}) )[0..32],
//           nleaf = ceil(len / 4096)  (0 leaves for the empty shard)
// Properties that matter here:
//   * order-free inside a leaf: s_l is a sum of per-word terms, so 256 lanes each add their own 16 bytes and the partial
//     sums combine with plain 64-bit adds -- the kernels that already hold the bytes in registers (gf_apply_nibble,
//     gf_apply_ptrs) accumulate it for 4 v_mad_u64_u32 per 16 bytes, with no extra HBM or link traffic;
//   * every corruption confined to one 32-bit word changes s_l (K[i] * delta != 0 mod 2^64 for 0 < |delta| < 2^32);
//     a corruption of several words goes unnoticed only if SUM K[i] * delta_i = 0 mod 2^64 -- about 2^-64 of all patterns
//     (2^-32 is the floor for patterns confined to two words); truncation, extension and leaf swaps are caught by the root;
//   * zero bytes contribute nothing: the checksum of a shard zero-extended to S bytes needs only the bytes that exist;
//   * NOT a MAC: the keys are public, an adversary can forge.  Integrity against an adversary is the block hash's job.
#pragma once

#include <stddef.h>
#include <stdint.h>

namespace mlh {

constexpr uint32_t LEAF_BYTES = 4096;
constexpr uint32_t LEAF_WORDS = LEAF_BYTES / 4;
constexpr uint64_t SEED = 0x6761726167654d4cULL;  // "garageML"
constexpr uint32_t ROOT_HEADER_BYTES = 16;        // "GECSUM3\0" + u64 length
constexpr uint64_t ROOT_MAGIC = 0x00334d5553434547ULL;  // "GECSUM3\0" as a little-endian u64

#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)
#define MLH_HD __host__ __device__
#else
#define MLH_HD
#endif

MLH_HD constexpr uint64_t splitmix64(uint64_t x)
{
	x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
	x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
	return x ^ (x >> 31);
}

MLH_HD constexpr uint32_t key(uint32_t i) { return (uint32_t)(splitmix64(SEED + (uint64_t)(i + 1) * 0x9E3779B97F4A7C15ULL) >> 32) | 1u; }

struct KeyTable {
	uint32_t k[LEAF_WORDS];
};

constexpr KeyTable make_keys()
{
	KeyTable t{};
	for (uint32_t i = 0; i < LEAF_WORDS; ++i)
		t.k[i] = key(i);
	return t;
}

inline size_t nleaf(size_t len) { return (len + LEAF_BYTES - 1) / LEAF_BYTES; }
// bytes of the root message of a shard of `len` bytes
inline size_t root_msg_bytes(size_t len) { return ROOT_HEADER_BYTES + 8 * nleaf(len); }

}  // namespace mlh
