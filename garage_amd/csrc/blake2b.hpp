// blake2b.hpp -- batched BLAKE2b-512 on gfx950, truncated to Garage's 32-byte
// `blake2sum` (src/util/data.rs:130-138: blake2b-512, first 32 bytes; NOT blake2b-256).
// once RS encode runs at TB/s the CPU hash pass
// (~1 GiB/s/core) is the pipeline bottleneck, so shard checksums are computed
// where the shards already are.
//
// BLAKE2b is a serial chain per message (RFC 7693), so the parallelism is ACROSS
// messages: one lane per message, all state in VGPRs (v[16], m[16], h[8] as
// 64-bit pairs), 12 rounds x 8 G per 128-byte block = ~2600 VALU ops per block.
// That makes it VALU-bound (~20 ops/byte): the loads (each lane streams its own
// message, 128 contiguous bytes per round) are <10% of the issue slots.  A batch of
// 1024 RS(10,4) stripes = 14336 shard messages = 224 waves.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernel_args.hpp"

namespace gec {


__device__ __forceinline__ const uint8_t *b2_msg_ptr(const Blake2Args &a, uint32_t i)
{
	if (a.off)
		return a.base + a.off[i];
	if (a.group)
		return a.base + (uint64_t)(i / a.group) * a.group_stride + (uint64_t)(i % a.group) * a.stride;
	return a.base + (uint64_t)i * a.stride;
}

__device__ __forceinline__ uint8_t *b2_out_ptr(const Blake2Args &a, uint32_t i)
{
	if (a.group)
		return a.out + 32ull * ((uint64_t)(i / a.group) * a.out_group + i % a.group);
	return a.out + 32ull * i;
}

// rotr64 as two v_alignbit_b32 on the 32-bit halves (hipcc's generic lowering is a
// 64-bit shift + shift + or); n = 32 is a free register swap.
typedef uint32_t b2_u32x2 __attribute__((ext_vector_type(2)));

// {lo, hi} -> u64 as a register pair.  NOT `(hi << 32) | lo`: LLVM turns that `or` of disjoint bits into
// an add and then folds it into the following 64-bit add as TWO v_lshl_add_u64 (c + lo + (hi << 32)),
// i.e. one more quarter-rate instruction on the dependency chain after every rotate.
__device__ __forceinline__ uint64_t b2_mk64(uint32_t lo, uint32_t hi)
{
	const b2_u32x2 v = {lo, hi};
	return __builtin_bit_cast(uint64_t, v);
}

template <int N>
__device__ __forceinline__ uint64_t b2_rotr(uint64_t x)
{
	const b2_u32x2 h = __builtin_bit_cast(b2_u32x2, x);
	const uint32_t lo = h.x, hi = h.y;
	if (N == 32)
		return b2_mk64(hi, lo);
	if (N < 32)
		return b2_mk64(__builtin_amdgcn_alignbit(hi, lo, N), __builtin_amdgcn_alignbit(lo, hi, N));
	// N > 32: rotate by 32 (swap) then by N - 32
	return b2_mk64(__builtin_amdgcn_alignbit(lo, hi, N - 32), __builtin_amdgcn_alignbit(hi, lo, N - 32));
}

// ADD32 selects how a 64-bit add is spelled (GEC_B2_ADD=0|1 is the A/B switch, tools/shardsum_bench.py):
//   0  v_lshl_add_u64, hipcc's own choice and the default;
//   1  v_add_co_u32 + v_addc_co_u32: 1.4x SLOWER, alone on a SIMD and with six waves per SIMD alike.
// (A third spelling -- carry out of bit 31 computed with v_bitop3_b32 and folded in with v_add3_u32, four
// full-rate instructions -- was tried in round 2 and measured 1.5x slower than v_lshl_add_u64 as well:
// profiles/r02_shardsum.txt.  The 64-bit add is not the bottleneck it looks like on paper.)
template <int ADD32>
__device__ __forceinline__ uint64_t b2_add(uint64_t a, uint64_t b)
{
	if (ADD32 == 0)
		return a + b;
	uint32_t lo, hi;
	const uint32_t carry = __builtin_uadd_overflow((uint32_t)a, (uint32_t)b, &lo);
	hi = (uint32_t)(a >> 32) + (uint32_t)(b >> 32) + carry;
	return ((uint64_t)hi << 32) | lo;
}

#define GEC_B2_G(a, b, c, d, x, y)       \
	a = b2_add<ADD32>(b2_add<ADD32>(a, b), (x));   \
	d = b2_rotr<32>(d ^ a);          \
	c = b2_add<ADD32>(c, d);               \
	b = b2_rotr<24>(b ^ c);          \
	a = b2_add<ADD32>(b2_add<ADD32>(a, b), (y));   \
	d = b2_rotr<16>(d ^ a);          \
	c = b2_add<ADD32>(c, d);               \
	b = b2_rotr<63>(b ^ c);

#define GEC_B2_ROUND(s0, s1, s2, s3, s4, s5, s6, s7, s8, s9, s10, s11, s12, s13, s14, s15) \
	GEC_B2_G(v0, v4, v8, v12, m[s0], m[s1]);                                           \
	GEC_B2_G(v1, v5, v9, v13, m[s2], m[s3]);                                           \
	GEC_B2_G(v2, v6, v10, v14, m[s4], m[s5]);                                          \
	GEC_B2_G(v3, v7, v11, v15, m[s6], m[s7]);                                          \
	GEC_B2_G(v0, v5, v10, v15, m[s8], m[s9]);                                          \
	GEC_B2_G(v1, v6, v11, v12, m[s10], m[s11]);                                        \
	GEC_B2_G(v2, v7, v8, v13, m[s12], m[s13]);                                         \
	GEC_B2_G(v3, v4, v9, v14, m[s14], m[s15]);

template <int ADD32>
__device__ __forceinline__ void b2_compress(uint64_t (&h)[8], const uint64_t (&m)[16], uint64_t t, bool last, bool last_node = false)
{
	const uint64_t IV0 = 0x6a09e667f3bcc908ULL, IV1 = 0xbb67ae8584caa73bULL, IV2 = 0x3c6ef372fe94f82bULL,
		       IV3 = 0xa54ff53a5f1d36f1ULL, IV4 = 0x510e527fade682d1ULL, IV5 = 0x9b05688c2b3e6c1fULL,
		       IV6 = 0x1f83d9abfb41bd6bULL, IV7 = 0x5be0cd19137e2179ULL;
	uint64_t v0 = h[0], v1 = h[1], v2 = h[2], v3 = h[3], v4 = h[4], v5 = h[5], v6 = h[6], v7 = h[7];
	uint64_t v8 = IV0, v9 = IV1, v10 = IV2, v11 = IV3, v12 = IV4 ^ t, v13 = IV5, v14 = last ? ~IV6 : IV6,
		 v15 = (last && last_node) ? ~IV7 : IV7;  // f1: "last node" of a tree level (RFC 7693 / BLAKE2 spec 2.10)
	// sigma permutations are compile-time, so m[] stays in registers
	GEC_B2_ROUND(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15)
	GEC_B2_ROUND(14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3)
	GEC_B2_ROUND(11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4)
	GEC_B2_ROUND(7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8)
	GEC_B2_ROUND(9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13)
	GEC_B2_ROUND(2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9)
	GEC_B2_ROUND(12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11)
	GEC_B2_ROUND(13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10)
	GEC_B2_ROUND(6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5)
	GEC_B2_ROUND(10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0)
	GEC_B2_ROUND(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15)
	GEC_B2_ROUND(14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3)
	h[0] ^= v0 ^ v8;
	h[1] ^= v1 ^ v9;
	h[2] ^= v2 ^ v10;
	h[3] ^= v3 ^ v11;
	h[4] ^= v4 ^ v12;
	h[5] ^= v5 ^ v13;
	h[6] ^= v6 ^ v14;
	h[7] ^= v7 ^ v15;
}

typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));

// Messages must start on 16-byte boundaries (shards do: 64-byte geometry; the host
// API stages each message into a 16-byte aligned slot).
template <int ADD32>
__global__ __launch_bounds__(64) void blake2b_batch(const Blake2Args a)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= a.n)
		return;
	const uint8_t *p = b2_msg_ptr(a, i);
	const uint64_t len = a.len ? a.len[i] : a.uniform_len;
	uint64_t h[8] = {0x6a09e667f3bcc908ULL ^ 0x01010040ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL,
			 0xa54ff53a5f1d36f1ULL, 0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL,
			 0x5be0cd19137e2179ULL};
	uint64_t m[16];
	uint64_t done = 0;
	// full blocks, all but the last one
	while (len - done > 128) {
		const u64x2 *q = reinterpret_cast<const u64x2 *>(p + done);
#pragma unroll
		for (int j = 0; j < 8; ++j) {
			u64x2 w = __builtin_nontemporal_load(q + j);
			m[2 * j] = w.x;
			m[2 * j + 1] = w.y;
		}
		done += 128;
		b2_compress<ADD32>(h, m, done, false);
	}
	// last block: 0..128 bytes, zero-padded; never reads past p + len
	const uint64_t rem = len - done;
#pragma unroll
	for (int j = 0; j < 16; ++j) {
		uint64_t w = 0;
		if ((uint64_t)(8 * j + 8) <= rem) {
			w = *reinterpret_cast<const uint64_t *>(p + done + 8 * j);
		} else if ((uint64_t)(8 * j) < rem) {
			for (uint64_t b = 0; b < rem - 8 * j; ++b)
				w |= (uint64_t)p[done + 8 * j + b] << (8 * b);
		}
		m[j] = w;
	}
	b2_compress<ADD32>(h, m, len, true);
	u64x2 *o = reinterpret_cast<u64x2 *>(b2_out_ptr(a, i));
	u64x2 lo = {h[0], h[1]}, hi = {h[2], h[3]};
	o[0] = lo;
	o[1] = hi;
}


// ---------------------------------------------------------------------------
// Quad variant: FOUR lanes per message (the classic SIMD BLAKE2 layout).  Lane q of
// a quad owns column q of the 4x4 state (a,b,c,d = v[q], v[4+q], v[8+q], v[12+q]);
// the four G of a column step run in the four lanes, the diagonal step first rotates
// b/c/d by 1/2/3 lanes inside the quad with DPP quad_perm moves.  The 128-byte
// message block is staged in LDS (each lane loads 32 bytes) and every lane gathers
// the two words its G needs from there: the sigma schedule becomes a packed
// per-round constant of four 7-bit byte offsets, one v_bfe_u32 away.
//
// With 14336 shard messages (one 1024-stripe batch) this is 896 single-wave workgroups on
// 1024 SIMDs: every wave has a SIMD to itself, so a launch lasts as long as ONE message's
// dependency chain -- 820 blocks x 24 G steps.  Everything that is not on that chain is
// therefore moved off it:
//  * the two message words of a step are gathered from LDS ONE STEP AHEAD (they do not
//    depend on the state), the first step's words of block i+1 during the last step of
//    block i: two LDS slots per message, block i+1 is written to its slot before block i
//    is compressed, block i+2 is already on its way from HBM;
//  * a = a + b + x is evaluated as (a + x) + b: a has been ready for three steps, only b is
//    fresh, so one 64-bit add instead of two sits on the chain (same for + y);
//  * the waves raise their priority: when the hash runs beside the RS kernel (encode+hash),
//    its chain must not queue behind the other kernel's VALU bursts on the same SIMD.
// Peak rate with very many messages is lower than the one-lane kernel's (DPP + LDS overhead),
// so the host picks by batch size.
// ---------------------------------------------------------------------------
constexpr int B2Q_SLOT = 144;  // LDS bytes per message block: 128 + pad (bank spread, 16-B aligned)

template <int CTRL>
__device__ __forceinline__ uint64_t b2_quad_perm(uint64_t x)
{
	const int lo = __builtin_amdgcn_mov_dpp((int)(uint32_t)x, CTRL, 0xf, 0xf, true);
	const int hi = __builtin_amdgcn_mov_dpp((int)(uint32_t)(x >> 32), CTRL, 0xf, 0xf, true);
	return ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo;
}

// byte offsets (8 * word index, 7 bits each) of sigma[r][i0 + 2q], q = 0..3
constexpr uint32_t b2q_pack(const uint8_t (&s)[16], int i0)
{
	return (uint32_t)(s[i0] * 8) | ((uint32_t)(s[i0 + 2] * 8) << 7) | ((uint32_t)(s[i0 + 4] * 8) << 14) |
	       ((uint32_t)(s[i0 + 6] * 8) << 21);
}

struct B2QSchedule {
	uint32_t w[12][4];  // [round][col x, col y, diag x, diag y]
};

constexpr B2QSchedule b2q_schedule()
{
	constexpr uint8_t SIG[12][16] = {
		{0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
		{11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
		{9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
		{12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
		{6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0},
		{0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3}};
	B2QSchedule t{};
	for (int r = 0; r < 12; ++r) {
		t.w[r][0] = b2q_pack(SIG[r], 0);
		t.w[r][1] = b2q_pack(SIG[r], 1);
		t.w[r][2] = b2q_pack(SIG[r], 8);
		t.w[r][3] = b2q_pack(SIG[r], 9);
	}
	return t;
}

typedef __attribute__((address_space(3))) const uint64_t lds_u64_t;

// one G with the message words folded in off the dependency chain: (a + x) and, later, (a + y) only
// need `a`, which is ready long before the freshly rotated `b`
// b2_pin: an empty asm the optimiser cannot see through -- without it LLVM re-associates (a + x) + b
// back to (b + x) + a and both adds wait for the freshly rotated b again.
__device__ __forceinline__ uint64_t b2_pin(uint64_t v)
{
#ifdef GEC_B2Q_PIN  // A/B: since a lone wave is issue-bound, the s_nop the hazard recognizer puts behind every inline asm
		    // (its result could be a dst_sel write) costs more than the longer dependency chain: 1449 vs 1404 ns
	asm("" : "+v"(v));
#endif
	return v;
}

// One step (the four G of a column or diagonal step, one per lane).  Comes in with ax = a + x already
// formed (x = this step's first message word), y = the second word, and nx = the NEXT step's first word;
// leaves ax = a + nx, formed as soon as the new a exists, i.e. in the shadow of the d -> c -> b tail.
// On the dependency chain per step: 4 adds, 4 xors, 3 rotates (rot 32 is a register swap).
#define GEC_B2Q_STEP(ax, b, c, d, y, nx)  \
	{                                     \
		uint64_t a_ = ax + b;             \
		d = b2_rotr<32>(d ^ a_);          \
		c = c + d;                        \
		b = b2_rotr<24>(b ^ c);           \
		a_ = b2_pin(a_ + (y)) + b;        \
		ax = b2_pin(a_ + (nx));           \
		d = b2_rotr<16>(d ^ a_);          \
		c = c + d;                        \
		b = b2_rotr<63>(b ^ c);           \
	}

// Compresses the block staged in the current slot.  ax/y come in holding (a + x) and y of round 0's column
// step (x, y gathered by the caller or by the previous call) and leave holding those of the NEXT
// block's, whose words are read from the other slot (garbage after the last block: never used).  The state
// words a live inside ax: h_a is recovered at the end as ax - (next x).
// wa[r][i]: LDS byte address, in slot 0, of the message word lane q needs at round r (column x, column y,
// diagonal x, diagonal y) -- computed once per kernel; the two slots are a compile-time distance apart, so which
// slot a read goes to is an immediate offset on the ds_read (ODD = the current block sits in slot 1): no address
// arithmetic is left in the compression loop.
constexpr uint32_t B2Q_SLOT1 = 16 * B2Q_SLOT;

// (b2q_compress_off: the same with the two offsets spelled out -- the fused small-trip kernel, fused.hpp, hashes messages that
// lie in LDS whole, block i at base + 128 * i: CUR = 0 / 128, NXT = CUR + 128)
template <uint32_t CUR, uint32_t NXT>
__device__ __forceinline__ void b2q_compress_off(uint64_t &ha, uint64_t &hb, const uint32_t (&wa)[10][4], uint32_t q, uint64_t t,
						 bool last, uint64_t &x, uint64_t &y, bool last_node = false)
{
	const uint64_t IVq = q == 0 ? 0x6a09e667f3bcc908ULL : q == 1 ? 0xbb67ae8584caa73bULL
			   : q == 2 ? 0x3c6ef372fe94f82bULL : 0xa54ff53a5f1d36f1ULL;
	const uint64_t IVq4 = q == 0 ? 0x510e527fade682d1ULL : q == 1 ? 0x9b05688c2b3e6c1fULL
			    : q == 2 ? 0x1f83d9abfb41bd6bULL : 0x5be0cd19137e2179ULL;
	uint64_t b = hb, c = IVq, d = IVq4;
	uint64_t ax = ha + x;
	if (q == 0)
		d ^= t;
	if (q == 2 && last)
		d = ~d;
	if (q == 3 && last && last_node)  // tree mode: f1 on the final block of the last node of a level
		d = ~d;
#define GEC_B2Q_WORD(off, r, i) (*reinterpret_cast<lds_u64_t *>(wa[(r) % 10][i] + (off)))
	// The four message words of a round are gathered together, one whole round before they are used (a lone wave pays
	// an issue slot for every s_waitcnt: LDS returns in order, so ONE wait per round -- "all but the four reads just
	// issued" -- covers the round's four words, where gathering them pairwise a step ahead cost a wait per use).
	// w[0..3] = (column x, column y, diagonal x, diagonal y); x and y arrive holding round 0's column words.
	uint64_t cx = x, cy = y, dx = GEC_B2Q_WORD(CUR, 0, 2), dy = GEC_B2Q_WORD(CUR, 0, 3);
#pragma unroll
	for (int r = 0; r < 12; ++r) {
		// next round's words (or, in the last round, the next block's round-0 column words: the diagonal ones are
		// read by the next call)
		uint64_t ncx, ncy, ndx = 0, ndy = 0;
		if (r < 11) {
			ncx = GEC_B2Q_WORD(CUR, r + 1, 0);
			ncy = GEC_B2Q_WORD(CUR, r + 1, 1);
			ndx = GEC_B2Q_WORD(CUR, r + 1, 2);
			ndy = GEC_B2Q_WORD(CUR, r + 1, 3);
		} else {
			ncx = GEC_B2Q_WORD(NXT, 0, 0);
			ncy = GEC_B2Q_WORD(NXT, 0, 1);
		}
		__builtin_amdgcn_sched_barrier(0);  // keep the gathers up here: hipcc otherwise sinks them next to their use
		GEC_B2Q_STEP(ax, b, c, d, cy, dx)
		b = b2_quad_perm<0x39>(b);
		c = b2_quad_perm<0x4E>(c);
		d = b2_quad_perm<0x93>(d);
		GEC_B2Q_STEP(ax, b, c, d, dy, ncx)
		b = b2_quad_perm<0x93>(b);
		c = b2_quad_perm<0x4E>(c);
		d = b2_quad_perm<0x39>(d);
		cx = ncx;
		cy = ncy;
		dx = ndx;
		dy = ndy;
	}
	x = cx;
	y = cy;
#undef GEC_B2Q_WORD
	const uint64_t a = ax - x;  // undo the look-ahead add of the (not yet started) next step
	ha ^= a ^ c;
	hb ^= b ^ d;
}

template <bool ODD>
__device__ __forceinline__ void b2q_compress(uint64_t &ha, uint64_t &hb, const uint32_t (&wa)[10][4], uint32_t q, uint64_t t,
					     bool last, uint64_t &x, uint64_t &y, bool last_node = false)
{
	b2q_compress_off<(ODD ? B2Q_SLOT1 : 0u), (ODD ? 0u : B2Q_SLOT1)>(ha, hb, wa, q, t, last, x, y, last_node);
}

// `nblk` blocks of a message that lies in LDS at byte address `msg`, four lanes per message (lane q owns column q of the
// 4x4 state), with blake2b.hpp's compression: the message words of a round gathered a round ahead, the first words of the
// NEXT block during the last round of the current one, (a + x) formed off the dependency chain.  wa[r][i] = LDS address of
// the word this lane needs at round r of block 0; two blocks per trip (immediate offsets 0 / 128 / 256 on the ds_reads),
// then all forty addresses move on by 256.  t_before = bytes of the message hashed before these blocks; `finishes`: the
// message (total_len bytes) ends with them.  (The look-ahead reads up to 128 bytes past the last block: LDS, harmless.)
__device__ __forceinline__ void b2q_hash_lds(uint64_t &ha, uint64_t &hb, uint32_t msg, uint32_t nblk, uint64_t t_before,
					     uint64_t total_len, bool finishes, bool last_node, uint32_t q)
{
	constexpr B2QSchedule SCH = b2q_schedule();
	const uint32_t q7 = q * 7;
	uint32_t wa[10][4];
#pragma unroll
	for (int r = 0; r < 10; ++r)
#pragma unroll
		for (int w = 0; w < 4; ++w) {
			wa[r][w] = msg + __builtin_amdgcn_ubfe(SCH.w[r][w], q7, 7);
			asm("" : "+v"(wa[r][w]));  // opaque: otherwise the sums are re-formed inside the loop
		}
	uint64_t x = *reinterpret_cast<lds_u64_t *>(wa[0][0]), y = *reinterpret_cast<lds_u64_t *>(wa[0][1]);
	uint32_t i = 0;
	// pairs with nothing to decide: neither block is the message's last
	const uint32_t plain = finishes ? (nblk ? nblk - 1 : 0) : nblk;  // blocks that are certainly not final
	for (; i + 2 <= plain; i += 2) {
		b2q_compress_off<0, 128>(ha, hb, wa, q, t_before + 128ull * (i + 1), false, x, y);
		b2q_compress_off<128, 256>(ha, hb, wa, q, t_before + 128ull * (i + 2), false, x, y);
#pragma unroll
		for (int r = 0; r < 10; ++r)
#pragma unroll
			for (int w = 0; w < 4; ++w)
				wa[r][w] += 256;
	}
	for (; i < nblk; ++i) {  // the last one or two
		const bool last = finishes && i + 1 == nblk;
		b2q_compress_off<0, 128>(ha, hb, wa, q, last ? total_len : t_before + 128ull * (i + 1), last, x, y, last_node);
#pragma unroll
		for (int r = 0; r < 10; ++r)
#pragma unroll
			for (int w = 0; w < 4; ++w)
				wa[r][w] += 128;
	}
}

// Lane q's 32-byte quarter of the 128-byte block that starts at byte `off` of a `len`-byte message,
// zero beyond the end; never reads past p + len.  Full quarters (all but a message's tail) are two
// 16-byte streaming loads.
__device__ __forceinline__ void b2q_fetch(const uint8_t *p, uint64_t off, uint64_t len, uint32_t q, u64x2 &w0, u64x2 &w1)
{
	const uint64_t o = off + 32 * q;
	if (o + 32 <= len) {
		const u64x2 *g = reinterpret_cast<const u64x2 *>(p + o);
		w0 = __builtin_nontemporal_load(g);
		w1 = __builtin_nontemporal_load(g + 1);
		return;
	}
	uint64_t w[4];
#pragma unroll
	for (int j = 0; j < 4; ++j) {
		const uint64_t oj = o + 8 * j;
		uint64_t v = 0;
		if (oj + 8 <= len) {
			v = *reinterpret_cast<const uint64_t *>(p + oj);
		} else if (oj < len) {
			for (uint64_t b = 0; b < len - oj; ++b)
				v |= (uint64_t)p[oj + b] << (8 * b);
		}
		w[j] = v;
	}
	w0 = u64x2{w[0], w[1]};
	w1 = u64x2{w[2], w[3]};
}

// One block of the chain: stage block blk+1 (requested one step ago) into the other slot, request block blk+2,
// compress block blk out of the current slot.  ODD: the current slot is slot 1.
struct B2QLane {
	const uint8_t *pq;   // message + 32*q: this lane's quarter of block 0
	uint64_t len, nblk, b1;
	uint32_t q, stage;   // stage = LDS address of this lane's quarter in slot 0
	u64x2 w0, w1;
	uint64_t ha, hb, x, y;
	bool last_node = false;  // tree-mode leaves only (blake2b_batch_quad<true>)
};

template <bool ODD>
__device__ __forceinline__ void b2q_block(B2QLane &L, const uint32_t (&wa)[10][4], uint64_t blk)
{
	typedef __attribute__((address_space(3))) u64x2 lds_u64x2_w;
	if (blk + 1 < L.b1) {  // block blk+1 has arrived: stage it
		lds_u64x2_w *s = reinterpret_cast<lds_u64x2_w *>(L.stage + (ODD ? 0 : B2Q_SLOT1));
		s[0] = L.w0;
		s[1] = L.w1;
	}
	if (blk + 2 < L.b1) {  // block blk+2: on its way from HBM while block blk is compressed
		if (blk + 3 < L.nblk) {  // not the message's last block: whole, no bounds to check
			const u64x2 *g = reinterpret_cast<const u64x2 *>(L.pq + (blk + 2) * 128);
			L.w0 = __builtin_nontemporal_load(g);
			L.w1 = __builtin_nontemporal_load(g + 1);
		} else {
			b2q_fetch(L.pq - 32 * L.q, (blk + 2) * 128, L.len, L.q, L.w0, L.w1);
		}
	}
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
	const bool last = blk + 1 == L.nblk;
	b2q_compress<ODD>(L.ha, L.hb, wa, L.q, last ? L.len : (blk + 1) * 128, last, L.x, L.y, L.last_node);
	__builtin_amdgcn_wave_barrier();
}

template <bool ODD>
__device__ __forceinline__ void b2q_block_fast(B2QLane &L, const uint32_t (&wa)[10][4], uint64_t blk)
{
	typedef __attribute__((address_space(3))) u64x2 lds_u64x2_w;
	lds_u64x2_w *s = reinterpret_cast<lds_u64x2_w *>(L.stage + (ODD ? 0 : B2Q_SLOT1));
	s[0] = L.w0;
	s[1] = L.w1;
	const u64x2 *g = reinterpret_cast<const u64x2 *>(L.pq + (blk + 2) * 128);
	L.w0 = __builtin_nontemporal_load(g);
	L.w1 = __builtin_nontemporal_load(g + 1);
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
	b2q_compress<ODD>(L.ha, L.hb, wa, L.q, (blk + 1) * 128, false, L.x, L.y);
	__builtin_amdgcn_wave_barrier();
}

// LEAF: the messages are the 4 KiB leaves of the shard checksums' tree (below): message i is leaf i % nleaf_max of shard
// i / nleaf_max, hashed with the leaf parameter block, and all 64 bytes of its digest go to leafdig[i].  For batches too
// small to give the one-lane-per-leaf kernel a wave per SIMD -- a PutObject's three blocks are 1092 leaves, 17 waves, 139 us
// of one wave's issue latency however idle the chip is -- four lanes per leaf finish in a third of the time.
// TREE = B2Q_ROOT: the messages are the shards' roots -- message s is the nleaf(s) * 64 bytes of leaf digests of shard s, hashed
// with the root parameter block, its first 32 bytes stored like a plain digest (shardsum_roots with four lanes per shard).
enum : int { B2Q_PLAIN = 0, B2Q_LEAF = 1, B2Q_ROOT = 2 };
template <int TREE = B2Q_PLAIN>
__global__ __launch_bounds__(64) void blake2b_batch_quad(const Blake2Args a, uint32_t nleaf_max = 0, uint8_t *__restrict__ leafdig = nullptr)
{
	constexpr bool LEAF = TREE == B2Q_LEAF, ROOT = TREE == B2Q_ROOT;
	__shared__ __attribute__((aligned(16))) uint8_t lds[2 * 16 * B2Q_SLOT];
	__builtin_amdgcn_s_setprio(3);
	const uint32_t lane = threadIdx.x;
	const uint32_t q = lane & 3;
	const uint32_t i = blockIdx.x * 16 + (lane >> 2);
	bool live;
	uint32_t ii;
	const uint8_t *p;
	uint64_t len;
	uint32_t leaf = 0;
	bool last_node = false;
	if (LEAF) {
		const uint64_t total = (uint64_t)a.n * nleaf_max;
		const uint64_t iq = (uint64_t)i < total ? i : total - 1;  // dead quads shadow the last leaf slot (no stores)
		const uint32_t s = (uint32_t)(iq / nleaf_max);
		leaf = (uint32_t)(iq % nleaf_max);
		const uint64_t slen = a.len ? a.len[s] : a.uniform_len;
		const uint32_t nleaf = slen ? (uint32_t)((slen + SHARDSUM_LEAF - 1) / SHARDSUM_LEAF) : 1;
		live = (uint64_t)i < total && leaf < nleaf;
		if (leaf >= nleaf)
			leaf = nleaf - 1;  // a slot beyond this shard's leaves: shadow its last leaf
		ii = s;
		p = b2_msg_ptr(a, s) + (uint64_t)leaf * SHARDSUM_LEAF;
		const uint64_t lo = (uint64_t)leaf * SHARDSUM_LEAF;
		len = slen > lo ? (slen - lo < SHARDSUM_LEAF ? slen - lo : SHARDSUM_LEAF) : 0;
		last_node = leaf + 1 == nleaf;
	} else if (ROOT) {
		live = i < a.n;
		ii = live ? i : a.n - 1;
		const uint64_t slen = a.len ? a.len[ii] : a.uniform_len;
		const uint32_t nleaf = slen ? (uint32_t)((slen + SHARDSUM_LEAF - 1) / SHARDSUM_LEAF) : 1;
		p = leafdig + (uint64_t)ii * nleaf_max * 64;
		len = (uint64_t)nleaf * 64;
		last_node = true;
	} else {
		live = i < a.n;
		ii = live ? i : a.n - 1;  // dead quads shadow the last message (no stores)
		p = b2_msg_ptr(a, ii);
		len = a.len ? a.len[ii] : a.uniform_len;
	}
	const uint32_t slot0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)(lds + (lane >> 2) * B2Q_SLOT);
	const uint32_t q7 = q * 7;
	B2QLane L;
	if (LEAF)  // the leaf parameter block: like shardsum_leaves
		L.ha = (q == 0 ? 0x6a09e667f3bcc908ULL ^ SHARDSUM_P0 : q == 1 ? 0xbb67ae8584caa73bULL ^ (uint64_t)leaf /*node_offset*/
			: q == 2 ? 0x3c6ef372fe94f82bULL ^ SHARDSUM_P2_LEAF : 0xa54ff53a5f1d36f1ULL);
	else if (ROOT)
		L.ha = (q == 0 ? 0x6a09e667f3bcc908ULL ^ SHARDSUM_P0 : q == 1 ? 0xbb67ae8584caa73bULL
			: q == 2 ? 0x3c6ef372fe94f82bULL ^ SHARDSUM_P2_ROOT : 0xa54ff53a5f1d36f1ULL);
	else
		L.ha = (q == 0 ? 0x6a09e667f3bcc908ULL ^ 0x01010040ULL : q == 1 ? 0xbb67ae8584caa73bULL
			: q == 2 ? 0x3c6ef372fe94f82bULL : 0xa54ff53a5f1d36f1ULL);
	L.hb = (q == 0 ? 0x510e527fade682d1ULL : q == 1 ? 0x9b05688c2b3e6c1fULL
		: q == 2 ? 0x1f83d9abfb41bd6bULL : 0x5be0cd19137e2179ULL);
	L.last_node = (LEAF || ROOT) && last_node;
	typedef __attribute__((address_space(3))) u64x2 lds_u64x2_w;
	const uint64_t nblk = len ? (len + 127) / 128 : 1;  // the empty message still has one (all-zero, final) block
	const uint64_t b0 = TREE ? 0 : a.seg_begin_blk, b1 = TREE ? nblk : (nblk < a.seg_end_blk ? nblk : a.seg_end_blk);
	if (!TREE && b0 > 0 && b0 < b1) {  // resume
		L.ha = a.state[8ull * ii + q];
		L.hb = a.state[8ull * ii + 4 + q];
	}
	L.pq = p + 32 * q;
	L.len = len;
	L.nblk = nblk;
	L.b1 = b1;
	L.q = q;
	L.stage = slot0 + 32 * q;
	// the message-word addresses of this lane for every round (rounds 10 and 11 repeat 0 and 1)
	constexpr B2QSchedule SCH = b2q_schedule();
	uint32_t wa[10][4];
#pragma unroll
	for (int r = 0; r < 10; ++r)
#pragma unroll
		for (int w = 0; w < 4; ++w) {
			wa[r][w] = slot0 + __builtin_amdgcn_ubfe(SCH.w[r][w], q7, 7);
			asm("" : "+v"(wa[r][w]));  // opaque: otherwise the sums are re-formed inside the loop (19 v_add_u32 per block)
		}
	b2q_fetch(p, b0 * 128, len, q, L.w0, L.w1);
	{
		lds_u64x2_w *s = reinterpret_cast<lds_u64x2_w *>(L.stage);
		s[0] = L.w0;
		s[1] = L.w1;
	}
	if (b0 + 1 < b1)
		b2q_fetch(p, (b0 + 1) * 128, len, q, L.w0, L.w1);
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
	L.x = *reinterpret_cast<lds_u64_t *>(wa[0][0]);
	L.y = *reinterpret_cast<lds_u64_t *>(wa[0][1]);
	// two blocks per trip: which slot is current is then a compile-time fact of each half.  First the blocks with
	// nothing to decide -- two whole successors inside this segment, not the message's last block: stage, request,
	// compress, no per-block conditions (a lone wave pays an issue slot for every compare, select and branch) --
	// then the general form for the last few.
	uint64_t blk = b0;
	{
		const uint64_t lim_a = b1 >= 2 ? b1 - 2 : 0, lim_b = nblk >= 3 ? nblk - 3 : 0;
		const uint64_t fast_until = lim_a < lim_b ? lim_a : lim_b;  // blocks below it are "fast"
		for (; blk + 1 < fast_until; blk += 2) {
			b2q_block_fast<false>(L, wa, blk);
			b2q_block_fast<true>(L, wa, blk + 1);
		}
	}
	for (; blk < b1; blk += 2) {
		b2q_block<false>(L, wa, blk);
		if (blk + 1 < b1)
			b2q_block<true>(L, wa, blk + 1);
	}
	if (!live || b0 >= b1)
		return;
	if (LEAF) {
		uint64_t *o = reinterpret_cast<uint64_t *>(leafdig + 64ull * i);  // i = shard * nleaf_max + leaf
		o[q] = L.ha;
		o[4 + q] = L.hb;
	} else if (b1 == nblk) {
		reinterpret_cast<uint64_t *>(b2_out_ptr(a, i))[q] = L.ha;  // h[0..3] = first 32 bytes
	} else {
		a.state[8ull * i + q] = L.ha;
		a.state[8ull * i + 4 + q] = L.hb;
	}
}

// ---------------------------------------------------------------------------
// Shard checksums: BLAKE2b in TREE mode (BLAKE2 specification section 2.10; the parameter block of RFC 7693
// section 2.5), two levels: leaves of GEC_SHARDSUM_LEAF = 4096 bytes, unlimited fanout, 64-byte inner digests,
// the root's 64-byte digest truncated to 32 bytes like Garage's blake2sum.  Python's hashlib reproduces it
// (blake2b(..., fanout=0, depth=2, leaf_size=4096, node_offset=i, node_depth=0|1, inner_size=64, last_node=..)).
//
// Why not plain BLAKE2b over the shard: a BLAKE2b message is one serial chain, and a lone wave issues one VALU
// instruction per ~5 cycles (profiles/r02_valu_probe.txt) -- 14336 shards of 105 KB are 820-block chains that take
// 1.4 ms however the kernel is written, with three quarters of the chip's issue slots idle.  The shard checksum is
// this project's own format (Garage has no shards), so it is free to use the standard's own answer to that:
// 26 independent 32-block leaves per shard = 373k messages, enough waves per SIMD to fill the VALUs, and a
// 13-block root per shard.  Block names stay plain blake2sum (gec_blake2sum_batch): they are Garage's.
// ---------------------------------------------------------------------------

// One lane per LEAF: lane i hashes leaf (i % nleaf) of shard (i / nleaf); shards addressed like messages of
// Blake2Args (flat / grouped / offset table, uniform or per-shard lengths).  leafdig: [shard][leaf][64].
template <int ADD32>
__global__ __launch_bounds__(64) void shardsum_leaves(const Blake2Args a, uint32_t nleaf_max, uint8_t *__restrict__ leafdig)
{
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const uint32_t s = (uint32_t)(i / nleaf_max), l = (uint32_t)(i % nleaf_max);
	if (s >= a.n)
		return;
	const uint64_t slen = a.len ? a.len[s] : a.uniform_len;
	const uint32_t nleaf = slen ? (uint32_t)((slen + SHARDSUM_LEAF - 1) / SHARDSUM_LEAF) : 1;
	if (l >= nleaf)
		return;
	const uint8_t *p = b2_msg_ptr(a, s) + (uint64_t)l * SHARDSUM_LEAF;
	const uint64_t len = slen > (uint64_t)l * SHARDSUM_LEAF ? (slen - (uint64_t)l * SHARDSUM_LEAF < SHARDSUM_LEAF ? slen - (uint64_t)l * SHARDSUM_LEAF : SHARDSUM_LEAF) : 0;
	uint64_t h[8] = {0x6a09e667f3bcc908ULL ^ SHARDSUM_P0, 0xbb67ae8584caa73bULL ^ (uint64_t)l /*node_offset*/,
			 0x3c6ef372fe94f82bULL ^ SHARDSUM_P2_LEAF, 0xa54ff53a5f1d36f1ULL, 0x510e527fade682d1ULL,
			 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
	const bool last_node = l + 1 == nleaf;
	uint64_t m[16];
	uint64_t done = 0;
	while (len - done > 128) {
		const u64x2 *q = reinterpret_cast<const u64x2 *>(p + done);
#pragma unroll
		for (int j = 0; j < 8; ++j) {
			u64x2 w = __builtin_nontemporal_load(q + j);
			m[2 * j] = w.x;
			m[2 * j + 1] = w.y;
		}
		done += 128;
		b2_compress<ADD32>(h, m, done, false);
	}
	const uint64_t rem = len - done;
#pragma unroll
	for (int j = 0; j < 16; ++j) {
		uint64_t w = 0;
		if ((uint64_t)(8 * j + 8) <= rem) {
			w = *reinterpret_cast<const uint64_t *>(p + done + 8 * j);
		} else if ((uint64_t)(8 * j) < rem) {
			for (uint64_t b = 0; b < rem - 8 * j; ++b)
				w |= (uint64_t)p[done + 8 * j + b] << (8 * b);
		}
		m[j] = w;
	}
	b2_compress<ADD32>(h, m, len, true, last_node);
	u64x2 *o = reinterpret_cast<u64x2 *>(leafdig + ((uint64_t)s * nleaf_max + l) * 64);
	o[0] = u64x2{h[0], h[1]};
	o[1] = u64x2{h[2], h[3]};
	o[2] = u64x2{h[4], h[5]};
	o[3] = u64x2{h[6], h[7]};
}

// One lane per shard: the root over that shard's leaf digests (nleaf * 64 bytes, always whole blocks of 128 bytes
// except possibly the last 64).  Output placement like b2_out_ptr.
__global__ __launch_bounds__(64) void shardsum_roots(const Blake2Args a, uint32_t nleaf_max, const uint8_t *__restrict__ leafdig)
{
	const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= a.n)
		return;
	const uint64_t slen = a.len ? a.len[s] : a.uniform_len;
	const uint32_t nleaf = slen ? (uint32_t)((slen + SHARDSUM_LEAF - 1) / SHARDSUM_LEAF) : 1;
	const uint64_t len = (uint64_t)nleaf * 64;
	const uint8_t *p = leafdig + (uint64_t)s * nleaf_max * 64;
	uint64_t h[8] = {0x6a09e667f3bcc908ULL ^ SHARDSUM_P0, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL ^ SHARDSUM_P2_ROOT,
			 0xa54ff53a5f1d36f1ULL, 0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL,
			 0x5be0cd19137e2179ULL};
	uint64_t m[16];
	uint64_t done = 0;
	while (len - done > 128) {
		const u64x2 *q = reinterpret_cast<const u64x2 *>(p + done);
#pragma unroll
		for (int j = 0; j < 8; ++j) {
			u64x2 w = q[j];
			m[2 * j] = w.x;
			m[2 * j + 1] = w.y;
		}
		done += 128;
		b2_compress<0>(h, m, done, false);
	}
	const uint64_t rem = len - done;  // 64 or 128
#pragma unroll
	for (int j = 0; j < 16; ++j)
		m[j] = (uint64_t)(8 * j) < rem ? *reinterpret_cast<const uint64_t *>(p + done + 8 * j) : 0;
	b2_compress<0>(h, m, len, true, true);
	u64x2 *o = reinterpret_cast<u64x2 *>(b2_out_ptr(a, s));
	o[0] = u64x2{h[0], h[1]};
	o[1] = u64x2{h[2], h[3]};
}

}  // namespace gec
