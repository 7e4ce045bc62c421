// mlh64_dev.hpp -- shard checksum v3 on gfx950 (definition and rationale: mlh64.hpp).
//
// Three producers of LEAF SUMS (8 bytes per 4 KiB leaf, lsum[shard * nleaf_max + leaf]) and one consumer:
//   * the RS kernels themselves (kernels.hpp: gf_apply_nibble<.., SUM>, gf_apply_ptrs<.., SUM>) -- every lane already holds
//     16 bytes of each input shard (d[j]) and of each output row (P[r]) in registers; four v_mad_u64_u32 turn them into the
//     lane's term of the leaf sum.  No extra HBM or link traffic: the checksum costs VALU slots only
//     (4 half-rate ops per 16 bytes = 0.5 full-rate op per byte, tools/csum_probe);
//   * mlh_leaves: a plain streaming kernel for shards that no RS kernel is touching (gec_shardsum_batch[_dev], the shards
//     a read uploads);
//   * mlh_roots: one lane per shard, BLAKE2b over "GECSUM3\0" || len || leaf sums (two compressions for a 104 KiB shard).
//
// Cross-lane reduction (WaveSums).  A lane's term must meet the other 255 terms of its leaf.  DPP butterflies over 14-28
// 64-bit values cost more VALU than the multiplies; LDS atomics serialize.  So: every wave owns an LDS region
// [64 lanes][CAP slots] of 64-bit terms (lane-major, 8 bytes of padding per lane: conflict-free writes), a lane drops its
// term for slot s there as soon as it has it (one ds_write_b64, nothing kept in registers), and when the region is full --
// or at the end of the tile -- the wave sums it up: lane (s, q) adds the terms of lanes 8q..8q+7 for slot s (8 ds_read_b64,
// conflict-free), two DPP steps add the four q of a quad, and the wave's two partial sums go to wsum[slot][2 wave + half].  One
// barrier at the very end of the tile, then (slot, leaf) threads add the eight partial sums of a leaf and store 8 bytes.  Wave-private regions need no barrier
// while the tile is being computed (LDS operations of one wave execute in order).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "blake2b.hpp"
#include "mlh64.hpp"

namespace gec {

__device__ const mlh::KeyTable MLH_KEYS = mlh::make_keys();

typedef uint32_t mlh_u32x4 __attribute__((ext_vector_type(4)));

// the four keys of 16-byte column `col` (0..255) of a leaf
__device__ __forceinline__ mlh_u32x4 mlh_keys_of(uint32_t col)
{
	return reinterpret_cast<const mlh_u32x4 *>(MLH_KEYS.k)[col & 255u];
}

// a lane's term: K[4c..4c+3] . (the column's four words), mod 2^64 -- four v_mad_u64_u32
__device__ __forceinline__ uint64_t mlh_col(const mlh_u32x4 d, const mlh_u32x4 k)
{
	uint64_t s = (uint64_t)k.x * d.x;
	s += (uint64_t)k.y * d.y;
	s += (uint64_t)k.z * d.z;
	s += (uint64_t)k.w * d.w;
	return s;
}

constexpr uint32_t mlh_region_bytes(int cap) { return 64u * (8u * cap + 8u); }

// LDS a workgroup of `waves` waves needs for nsl slots: the wave regions + wsum[nsl][2 * waves]
constexpr uint32_t mlh_lds_bytes(int cap, int waves, int nsl) { return waves * mlh_region_bytes(cap) + (uint32_t)nsl * waves * 16u; }

// x + (x of the lane DPP control `CTRL` names), both halves of the 64-bit value: two v_mov_b32_dpp and one 64-bit add
template <int CTRL>
__device__ __forceinline__ uint64_t mlh_add_dpp(uint64_t x)
{
	const uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
	const uint32_t plo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)lo, CTRL, 0xf, 0xf, true);
	const uint32_t phi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hi, CTRL, 0xf, 0xf, true);
	return x + (((uint64_t)phi << 32) | plo);
}

template <int CAP>
struct WaveSums {
	static constexpr uint32_t PITCH = 8u * CAP + 8u;  // bytes per lane: CAP terms + 8 bytes so that a slot's 64 writes spread over all banks
	uint8_t *region;   // this wave's [64 lanes][CAP slots]
	uint64_t *wsum;    // [nsl][2 * waves]: a wave leaves two partial sums per slot (lanes 0-31 / 32-63 of the summing pass)
	uint32_t lane, wave, waves;
	uint32_t pend = 0, base = 0;  // slots waiting in the region / slots already summed up (wave-uniform)

	__device__ __forceinline__ void init(uint8_t *lds_area, uint32_t tid, uint32_t nwaves)
	{
		lane = tid & 63u;
		wave = tid >> 6;
		waves = nwaves;
		region = lds_area + wave * mlh_region_bytes(CAP);
		wsum = reinterpret_cast<uint64_t *>(lds_area + nwaves * mlh_region_bytes(CAP));
	}
	// this lane's term for the next slot (slots are pushed in ascending order by every lane of the wave)
	__device__ __forceinline__ void put(uint64_t v)
	{
		*reinterpret_cast<uint64_t *>(region + lane * PITCH + 8u * pend) = v;
		++pend;
	}
	__device__ __forceinline__ bool full(uint32_t more) const { return pend + more > (uint32_t)CAP; }
	// Sums the region up: wsum[base + s][2 * wave + half] for s < pend.  Lane l of a pass takes slot s = (l >> 2) & 7 and the
	// terms of lanes 8q .. 8q + 7, q = (l & 3) + 4 * (l >> 5): within either half of the wave the 32 reads of one step fall on
	// 32 different bank pairs (lane pitch 34 dwords: bank = 16 q + 34 i + 2 s mod 64, q < 4), so the transposition costs no
	// bank conflicts; the four q of a quad then meet through two DPP steps -- no ds_bpermute, no barrier.
	__device__ __forceinline__ void flush()
	{
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		const uint32_t half = lane >> 5, q = (lane & 3u) + 4u * half;
		for (uint32_t s0 = 0; s0 < pend; s0 += 8) {  // (wave-uniform trip count)
			const uint32_t s = s0 + ((lane >> 2) & 7u);
			const uint32_t sc = s < pend ? s : pend - 1;  // (a lane without a slot shadows the last one: no divergence, its sum is not stored)
			const uint8_t *p = region + (q * 8u) * PITCH + 8u * sc;
			uint64_t acc = 0;
#pragma unroll
			for (int i = 0; i < 8; ++i)
				acc += *reinterpret_cast<const uint64_t *>(p + i * PITCH);
			acc = mlh_add_dpp<0xB1>(acc);  // quad_perm [1,0,3,2]
			acc = mlh_add_dpp<0x4E>(acc);  // quad_perm [2,3,0,1]
			if ((lane & 3u) == 0 && s < pend)
				wsum[(base + s) * (2 * waves) + 2 * wave + half] = acc;
		}
		base += pend;
		pend = 0;
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
	}
	// After the workgroup's barrier: thread t < nsl * leaves adds the 8 partial sums of (slot, leaf-of-the-tile) -- four waves, two
	// halves each -- and hands the sum to store(slot, leaf_in_tile, value).  leaves = waves / 4.
	template <class Store>
	__device__ __forceinline__ void combine(uint32_t tid, uint32_t nthr, uint32_t nsl, Store store) const
	{
		const uint32_t leaves = waves >> 2;
		for (uint32_t t = tid; t < nsl * leaves; t += nthr) {
			const uint32_t slot = t / leaves, g = t - slot * leaves;
			const uint64_t *w = wsum + slot * (2 * waves) + 8 * g;
			store(slot, g, ((w[0] + w[1]) + (w[2] + w[3])) + ((w[4] + w[5]) + (w[6] + w[7])));
		}
	}
};

// ---------------------------------------------------------------------------
// mlh_leaves: leaf sums of n shards that sit in memory (device, or pinned host memory read over the link).
// One wave per leaf, four 16-byte columns per lane (the whole leaf is requested before the first multiply), addressed
// like the BLAKE2b kernels (Blake2Args: base + stride / group / per-message offsets and lengths; bytes past a
// shard's length read as zero).  lsum[shard * nleaf_max + leaf].
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mlh_leaves(const Blake2Args a, uint32_t nleaf_max, uint64_t *__restrict__ lsum)
{
	const uint32_t lane = threadIdx.x & 63u;
	const uint64_t total = (uint64_t)a.n * nleaf_max;
	const uint64_t wl = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);  // (shard, leaf) of this wave
	if (wl >= total)
		return;
	const uint32_t s = (uint32_t)(wl / nleaf_max), l = (uint32_t)(wl - (uint64_t)s * nleaf_max);
	const uint64_t slen = a.len ? a.len[s] : a.uniform_len;
	const uint32_t nleaf = (uint32_t)((slen + mlh::LEAF_BYTES - 1) / mlh::LEAF_BYTES);
	if (l >= nleaf)
		return;
	const uint8_t *p = b2_msg_ptr(a, s) + (uint64_t)l * mlh::LEAF_BYTES;
	const uint64_t left = slen - (uint64_t)l * mlh::LEAF_BYTES;  // bytes of the shard from the start of this leaf
	const bool al16 = ((uintptr_t)p & 15u) == 0;
	mlh_u32x4 d[4];
#pragma unroll
	for (int i = 0; i < 4; ++i) {
		const uint32_t off = (uint32_t)(i * 64 + lane) * 16u;
		if (al16 && (uint64_t)off + 16 <= left) {
			d[i] = __builtin_nontemporal_load(reinterpret_cast<const mlh_u32x4 *>(p + off));
		} else {
			uint32_t w[4] = {0, 0, 0, 0};
			for (uint32_t b = 0; b < 16 && (uint64_t)off + b < left; ++b)  // the ragged end, or an unaligned buffer
				w[b >> 2] |= (uint32_t)p[off + b] << (8 * (b & 3));
			d[i] = mlh_u32x4{w[0], w[1], w[2], w[3]};
		}
	}
	uint64_t acc = 0;
#pragma unroll
	for (int i = 0; i < 4; ++i)
		acc += mlh_col(d[i], mlh_keys_of(i * 64 + lane));
	acc += __shfl_xor(acc, 1);
	acc += __shfl_xor(acc, 2);
	acc += __shfl_xor(acc, 4);
	acc += __shfl_xor(acc, 8);
	acc += __shfl_xor(acc, 16);
	acc += __shfl_xor(acc, 32);
	if (lane == 0)
		lsum[wl] = acc;
}

// ---------------------------------------------------------------------------
// mlh_roots: one lane per shard.  Message = MAGIC || len || s_0 .. s_{nleaf-1}; placement of the 32-byte result like the
// BLAKE2b kernels (b2_out_ptr).  slot_map != NULL: shard i's leaf sums are at lsum[slot_map[i] * nleaf_max] (the RS kernels
// number their sums (block, slot); the caller wants (block, shard index)), otherwise at lsum[i * nleaf_max].
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void mlh_roots(const Blake2Args a, uint32_t nleaf_max, const uint64_t *__restrict__ lsum,
						const uint32_t *__restrict__ slot_map)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= a.n)
		return;
	const uint64_t slen = a.len ? a.len[i] : a.uniform_len;
	const uint32_t nleaf = (uint32_t)((slen + mlh::LEAF_BYTES - 1) / mlh::LEAF_BYTES);
	const uint64_t *sums = lsum + (uint64_t)(slot_map ? slot_map[i] : i) * nleaf_max;
	const uint64_t len = mlh::ROOT_HEADER_BYTES + 8ull * nleaf;
	uint64_t h[8] = {0x6a09e667f3bcc908ULL ^ 0x01010040ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL, 0xa54ff53a5f1d36f1ULL,
			 0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
	uint64_t m[16];
	// word w of the message: 0 = magic, 1 = length, 2 + j = leaf sum j
	auto word = [&](uint64_t w) -> uint64_t { return w == 0 ? mlh::ROOT_MAGIC : w == 1 ? slen : (w - 2 < nleaf ? sums[w - 2] : 0); };
	uint64_t done = 0;
	while (len - done > 128) {
#pragma unroll
		for (int j = 0; j < 16; ++j)
			m[j] = word(done / 8 + j);
		done += 128;
		b2_compress<0>(h, m, done, false);
	}
#pragma unroll
	for (int j = 0; j < 16; ++j)
		m[j] = done + 8 * j < len ? word(done / 8 + j) : 0;
	b2_compress<0>(h, m, len, true);
	u64x2 *o = reinterpret_cast<u64x2 *>(b2_out_ptr(a, i));
	o[0] = u64x2{h[0], h[1]};
	o[1] = u64x2{h[2], h[3]};
}

// ---------------------------------------------------------------------------
// mlh_roots_quad: FOUR lanes per shard (blake2b.hpp's quad layout: lane q owns column q of the 4x4 state).  The root of a
// handful of shards -- one put's 14, a PutObject's 42 -- is one lone wave's dependency chain however idle the chip is:
// two compressions of ~2700 dependent VALU instructions with one lane per shard (9.6 us), a quarter of that with the G
// functions of a step in four lanes.  The message (16 + 8 * nleaf bytes) is staged in LDS whole, zero-padded to whole
// blocks, at a pitch of 128 * nblk_max + 16 bytes per quad (the 16: the quads' reads of the same word fall into different
// banks); the last quad's look-ahead reads run into the 144 spare bytes the launch adds.  Same results as mlh_roots, which
// stays for shards too long for LDS (the launch picks: mlh_roots_quad_fits).
// ---------------------------------------------------------------------------
constexpr uint32_t MLH_ROOTQ_MAX_BLOCKS = 24;  // 16 quads x (24 x 128 + 16) = 48.25 KiB of LDS; shards up to 1.49 MiB
__host__ __device__ constexpr uint32_t mlh_rootq_blocks(uint32_t nleaf) { return (2 + nleaf + 15) / 16; }  // >= 1: an empty shard's message is the 16-byte header
__host__ __device__ constexpr uint32_t mlh_rootq_pitch(uint32_t nleaf_max) { return 128u * mlh_rootq_blocks(nleaf_max) + 16u; }
__host__ __device__ constexpr uint32_t mlh_rootq_lds_bytes(uint32_t nleaf_max) { return 16u * mlh_rootq_pitch(nleaf_max) + 144u; }
__host__ __device__ constexpr bool mlh_roots_quad_fits(uint32_t nleaf_max) { return mlh_rootq_blocks(nleaf_max) <= MLH_ROOTQ_MAX_BLOCKS; }

__global__ __launch_bounds__(64) void mlh_roots_quad(const Blake2Args a, uint32_t nleaf_max, const uint64_t *__restrict__ lsum,
						     const uint32_t *__restrict__ slot_map)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t mlh_rootq_lds[];
	typedef __attribute__((address_space(3))) uint64_t lds_u64_w;
	const uint32_t tid = threadIdx.x, quad = tid >> 2, q = tid & 3;
	const uint32_t i = blockIdx.x * 16 + quad;
	const bool live = i < a.n;
	const uint32_t ii = live ? i : a.n - 1;  // dead quads shadow the last shard (no stores)
	const uint64_t slen = a.len ? a.len[ii] : a.uniform_len;
	const uint32_t nleaf = (uint32_t)((slen + mlh::LEAF_BYTES - 1) / mlh::LEAF_BYTES);
	const uint64_t *sums = lsum + (uint64_t)(slot_map ? slot_map[ii] : ii) * nleaf_max;
	const uint32_t nblk = mlh_rootq_blocks(nleaf), words = 16 * nblk;
	const uint32_t msg = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)mlh_rootq_lds + quad * mlh_rootq_pitch(nleaf_max);
	// word w of the message: 0 = magic, 1 = length, 2 + j = leaf sum j, zero beyond; lane q stages words q, q + 4, ...
	for (uint32_t w = q; w < words; w += 4) {
		const uint64_t v = w == 0 ? mlh::ROOT_MAGIC : w == 1 ? slen : (w - 2 < nleaf ? sums[w - 2] : 0);
		*reinterpret_cast<lds_u64_w *>(msg + 8 * w) = v;
	}
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
	uint64_t ha = q == 0 ? (0x6a09e667f3bcc908ULL ^ 0x01010040ULL) : q == 1 ? 0xbb67ae8584caa73bULL : q == 2 ? 0x3c6ef372fe94f82bULL : 0xa54ff53a5f1d36f1ULL;
	uint64_t hb = q == 0 ? 0x510e527fade682d1ULL : q == 1 ? 0x9b05688c2b3e6c1fULL : q == 2 ? 0x1f83d9abfb41bd6bULL : 0x5be0cd19137e2179ULL;
	b2q_hash_lds(ha, hb, msg, nblk, 0, mlh::ROOT_HEADER_BYTES + 8ull * nleaf, true, false, q);
	if (live)
		reinterpret_cast<uint64_t *>(b2_out_ptr(a, i))[q] = ha;  // h[0..3]: the first 32 bytes of the digest, 8 per lane
}

}  // namespace gec
