// blake2b_mb.hpp -- BLAKE2b-512 of EIGHT messages at a time on one host core (AVX-512: one 64-bit lane per message).
//
// A BLAKE2b message is one serial chain, 3.3 cycles per byte on a core whatever its vector width.  The checksums of this
// project always come in batches, though -- the 14 shards of a stripe, the 26 leaves of a shard's tree, the blocks of a
// GetObject -- and eight independent chains in the eight lanes of a zmm register cost barely more than one: the CPU
// backend's encode + checksums path (1.4 bytes hashed per payload byte against a GF multiply at 16 GiB/s per core) is
// a hashing path, and so is a small get's end-to-end check.  Same answers as blake2b_host.hpp, which stays the
// reference form and the fallback on cores without AVX-512 (tests/test_cpu_backend.py compares both with hashlib).
//
// Messages of one group advance in lock step; a lane whose message has ended keeps its state (masked), the last block
// of every message goes through a zero-padded copy.  Jobs of similar length should sit next to each other (the callers'
// batches do: shards of one size, 4 KiB leaves); a ragged group costs as much as its longest message.
#pragma once

#include "blake2b_host.hpp"

#include <atomic>

#if defined(__x86_64__)
#include <immintrin.h>
#define B2MB_X86 1
#endif

namespace b2host {

struct Job {
	const uint8_t *p = nullptr;  // may be NULL when len == 0
	size_t len = 0;
	uint64_t x0 = 0x01010000ULL ^ 64, x1 = 0, x2 = 0;  // parameter block words 0..2 (XORed into IV[0..2]); default: plain blake2b-512
	bool last_node = false;                            // tree mode: f1 on the final block
	uint8_t *out = nullptr;
	uint32_t outlen = 32;  // 32 (truncated, like blake2sum) or 64
	// A message that lies in several buffers (a block in its k data shards): pieces[t] holds bytes [t * piece_len,
	// (t + 1) * piece_len) of it, `p` is unused.  piece_len need not be a multiple of 128.
	const uint8_t *const *pieces = nullptr;
	size_t piece_len = 0;

	// copies bytes [off, off + n) of the message to dst
	void read(uint64_t off, size_t n, uint8_t *dst) const
	{
		if (!pieces) {
			std::memcpy(dst, p + off, n);
			return;
		}
		while (n) {
			const size_t t = off / piece_len, within = off % piece_len, take = std::min(n, piece_len - within);
			std::memcpy(dst, pieces[t] + within, take);
			dst += take;
			off += take;
			n -= take;
		}
	}
};

inline void run_job_scalar(const Job &j)
{
	uint8_t full[64];
	if (j.pieces) {
		State st(j.x0, j.x1, j.x2);
		for (size_t off = 0; off < j.len; off += j.piece_len)
			st.update(j.pieces[off / j.piece_len], std::min(j.piece_len, j.len - off));
		st.final(full, j.last_node);
	} else {
		blake2b_params(j.len ? j.p : nullptr, j.len, j.x0, j.x1, j.x2, j.last_node, full);
	}
	std::memcpy(j.out, full, j.outlen);
}

#ifdef B2MB_X86
// 0 = one message at a time everywhere (GEC_CPU_ISA = scalar / avx2, applied by each library when it is loaded:
// A/B, and how the tests reach the fallback on an AVX-512 box)
inline std::atomic<int> &mb_mode()
{
	static std::atomic<int> m{1};
	return m;
}
inline bool mb_available()
{
	static const bool hw = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512vl");
	return hw && mb_mode().load(std::memory_order_relaxed) != 0;
}

namespace mbimpl {

// m[j] lane i = word j of row i, for rows given as 8 pointers to 128-byte blocks
__attribute__((target("avx512f"))) inline void load_transposed(const uint8_t *const p[8], __m512i m[16])
{
	for (int half = 0; half < 2; ++half) {
		__m512i r[8], t[8], u[8];
		for (int i = 0; i < 8; ++i)
			r[i] = _mm512_loadu_si512(reinterpret_cast<const void *>(p[i] + 64 * half));
		for (int i = 0; i < 4; ++i) {
			t[2 * i] = _mm512_unpacklo_epi64(r[2 * i], r[2 * i + 1]);      // words 0,2,4,6 of rows 2i, 2i+1 interleaved
			t[2 * i + 1] = _mm512_unpackhi_epi64(r[2 * i], r[2 * i + 1]);  // words 1,3,5,7
		}
		// 128-bit pieces: t[2i] = [ (w0) (w2) (w4) (w6) ] of the row pair i, t[2i+1] = [ (w1) (w3) (w5) (w7) ]
		for (int odd = 0; odd < 2; ++odd) {
			u[4 * odd + 0] = _mm512_shuffle_i64x2(t[0 + odd], t[2 + odd], 0x88);  // pieces 0,2 of pairs 0,1: w(0|1), w(4|5)
			u[4 * odd + 1] = _mm512_shuffle_i64x2(t[0 + odd], t[2 + odd], 0xDD);  // pieces 1,3: w(2|3), w(6|7)
			u[4 * odd + 2] = _mm512_shuffle_i64x2(t[4 + odd], t[6 + odd], 0x88);  // the same of pairs 2,3
			u[4 * odd + 3] = _mm512_shuffle_i64x2(t[4 + odd], t[6 + odd], 0xDD);
		}
		__m512i *o = m + 8 * half;
		o[0] = _mm512_shuffle_i64x2(u[0], u[2], 0x88);
		o[4] = _mm512_shuffle_i64x2(u[0], u[2], 0xDD);
		o[2] = _mm512_shuffle_i64x2(u[1], u[3], 0x88);
		o[6] = _mm512_shuffle_i64x2(u[1], u[3], 0xDD);
		o[1] = _mm512_shuffle_i64x2(u[4], u[6], 0x88);
		o[5] = _mm512_shuffle_i64x2(u[4], u[6], 0xDD);
		o[3] = _mm512_shuffle_i64x2(u[5], u[7], 0x88);
		o[7] = _mm512_shuffle_i64x2(u[5], u[7], 0xDD);
	}
}

#define B2MB_G(a, b, c, d, x, y)                                               \
	v[a] = _mm512_add_epi64(_mm512_add_epi64(v[a], v[b]), (x));            \
	v[d] = _mm512_ror_epi64(_mm512_xor_si512(v[d], v[a]), 32);             \
	v[c] = _mm512_add_epi64(v[c], v[d]);                                   \
	v[b] = _mm512_ror_epi64(_mm512_xor_si512(v[b], v[c]), 24);             \
	v[a] = _mm512_add_epi64(_mm512_add_epi64(v[a], v[b]), (y));            \
	v[d] = _mm512_ror_epi64(_mm512_xor_si512(v[d], v[a]), 16);             \
	v[c] = _mm512_add_epi64(v[c], v[d]);                                   \
	v[b] = _mm512_ror_epi64(_mm512_xor_si512(v[b], v[c]), 63);
#define B2MB_ROUND(s0, s1, s2, s3, s4, s5, s6, s7, s8, s9, s10, s11, s12, s13, s14, s15) \
	B2MB_G(0, 4, 8, 12, m[s0], m[s1])                                                \
	B2MB_G(1, 5, 9, 13, m[s2], m[s3])                                                \
	B2MB_G(2, 6, 10, 14, m[s4], m[s5])                                               \
	B2MB_G(3, 7, 11, 15, m[s6], m[s7])                                               \
	B2MB_G(0, 5, 10, 15, m[s8], m[s9])                                               \
	B2MB_G(1, 6, 11, 12, m[s10], m[s11])                                             \
	B2MB_G(2, 7, 8, 13, m[s12], m[s13])                                              \
	B2MB_G(3, 4, 9, 14, m[s14], m[s15])

// h (8 words x 8 lanes) <- compress(h, m, t, f0, f1) in the lanes of `active`
__attribute__((target("avx512f"))) inline void compress8(__m512i h[8], const __m512i m[16], __m512i t, __m512i f0, __m512i f1, __mmask8 active)
{
	const uint64_t *IV = iv();
	__m512i v[16];
	for (int i = 0; i < 8; ++i) {
		v[i] = h[i];
		v[i + 8] = _mm512_set1_epi64((long long)IV[i]);
	}
	v[12] = _mm512_xor_si512(v[12], t);
	v[14] = _mm512_xor_si512(v[14], f0);
	v[15] = _mm512_xor_si512(v[15], f1);
	B2MB_ROUND(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15)
	B2MB_ROUND(14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3)
	B2MB_ROUND(11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4)
	B2MB_ROUND(7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8)
	B2MB_ROUND(9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13)
	B2MB_ROUND(2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9)
	B2MB_ROUND(12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11)
	B2MB_ROUND(13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10)
	B2MB_ROUND(6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5)
	B2MB_ROUND(10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0)
	B2MB_ROUND(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15)
	B2MB_ROUND(14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3)
	for (int i = 0; i < 8; ++i)
		h[i] = _mm512_mask_xor_epi64(h[i], active, h[i], _mm512_xor_si512(v[i], v[i + 8]));
}
#undef B2MB_ROUND
#undef B2MB_G

// up to eight jobs in lock step
__attribute__((target("avx512f"))) inline void run_group(const Job *jobs, int n)
{
	alignas(64) static const uint8_t kZero[128] = {};
	alignas(64) uint8_t pad[8][128];
	alignas(64) uint64_t hh[8][8];  // [word][lane]
	uint64_t nblk[8], len[8];
	const uint8_t *base[8];
	const uint64_t *IV = iv();
	uint64_t longest = 0;
	for (int i = 0; i < 8; ++i) {
		const bool live = i < n;
		len[i] = live ? jobs[i].len : 0;
		base[i] = live && jobs[i].len && !jobs[i].pieces ? jobs[i].p : kZero;
		nblk[i] = live ? (len[i] ? (len[i] + 127) / 128 : 1) : 0;
		longest = std::max(longest, nblk[i]);
		for (int w = 0; w < 8; ++w)
			hh[w][i] = IV[w];
		if (live) {
			hh[0][i] ^= jobs[i].x0;
			hh[1][i] ^= jobs[i].x1;
			hh[2][i] ^= jobs[i].x2;
		}
	}
	__m512i h[8];
	for (int w = 0; w < 8; ++w)
		h[w] = _mm512_load_si512(reinterpret_cast<const void *>(hh[w]));
	// blocks that are whole and not the last one in EVERY lane: nothing to decide per lane
	uint64_t common = 0;
	if (n == 8) {
		common = ~0ull;
		for (int i = 0; i < 8; ++i)
			common = std::min(common, jobs[i].pieces ? 0 : nblk[i] - 1);
	}
	const __m512i zero = _mm512_setzero_si512();
	for (uint64_t b = 0; b < common; ++b) {
		const uint8_t *p[8];
		for (int i = 0; i < 8; ++i)
			p[i] = base[i] + 128 * b;
		__m512i m[16];
		load_transposed(p, m);
		compress8(h, m, _mm512_set1_epi64((long long)(128 * (b + 1))), zero, zero, (__mmask8)0xFF);
	}
	for (uint64_t b = common; b < longest; ++b) {
		const uint8_t *p[8];
		alignas(64) uint64_t t[8], f0[8], f1[8];
		__mmask8 active = 0;
		for (int i = 0; i < 8; ++i) {
			t[i] = f0[i] = f1[i] = 0;
			if (b >= nblk[i]) {
				p[i] = kZero;
				continue;
			}
			active |= (__mmask8)(1u << i);
			if (b + 1 < nblk[i]) {
				t[i] = 128 * (b + 1);
				if (!jobs[i].pieces) {
					p[i] = base[i] + 128 * b;
				} else {  // in place when the block lies inside one piece, stitched when it straddles two
					const uint64_t off = 128 * b, within = off % jobs[i].piece_len;
					if (within + 128 <= jobs[i].piece_len) {
						p[i] = jobs[i].pieces[off / jobs[i].piece_len] + within;
					} else {
						jobs[i].read(off, 128, pad[i]);
						p[i] = pad[i];
					}
				}
			} else {  // the message's last block: zero-padded copy, total length, final flags
				const uint64_t done = 128 * b, rem = len[i] - done;
				std::memset(pad[i], 0, 128);
				if (rem)
					jobs[i].read(done, rem, pad[i]);
				p[i] = pad[i];
				t[i] = len[i];
				f0[i] = ~0ull;
				f1[i] = jobs[i].last_node ? ~0ull : 0;
			}
		}
		__m512i m[16];
		load_transposed(p, m);
		compress8(h, m, _mm512_load_si512(reinterpret_cast<const void *>(t)), _mm512_load_si512(reinterpret_cast<const void *>(f0)),
			  _mm512_load_si512(reinterpret_cast<const void *>(f1)), active);
	}
	for (int w = 0; w < 8; ++w)
		_mm512_store_si512(reinterpret_cast<void *>(hh[w]), h[w]);
	for (int i = 0; i < n; ++i) {
		uint64_t d[8];
		for (int w = 0; w < 8; ++w)
			d[w] = hh[w][i];
		std::memcpy(jobs[i].out, d, jobs[i].outlen);
	}
}

}  // namespace mbimpl
#else
inline std::atomic<int> &mb_mode()
{
	static std::atomic<int> m{0};
	return m;
}
inline bool mb_available() { return false; }
#endif

// Every job of the list (any number, any lengths): eight at a time where the core can, one at a time elsewhere.
inline void run_jobs(const Job *jobs, size_t n)
{
#ifdef B2MB_X86
	if (mb_available() && n > 1) {
		for (size_t i = 0; i < n; i += 8)
			mbimpl::run_group(jobs + i, (int)std::min<size_t>(8, n - i));
		return;
	}
#endif
	for (size_t i = 0; i < n; ++i)
		run_job_scalar(jobs[i]);
}

// blake2sum of n messages -> out[32 * i]
inline void blake2sum_many(const uint8_t *const *msgs, const size_t *lens, size_t n, uint8_t *out)
{
	std::vector<Job> jobs(n);
	for (size_t i = 0; i < n; ++i) {
		jobs[i].p = msgs[i];
		jobs[i].len = lens[i];
		jobs[i].out = out + 32 * i;
	}
	run_jobs(jobs.data(), n);
}

// shardsum (tree mode, see blake2b_host.hpp) of n shards -> out[32 * i]; `dst` lets the results go to scattered places instead
inline void shardsum_many(const uint8_t *const *msgs, const size_t *lens, size_t n, uint8_t *out, uint8_t *const *dst = nullptr)
{
	const uint64_t P0 = 64ull | (2ull << 24) | ((uint64_t)kShardsumLeaf << 32);
	constexpr size_t kShardsPerRound = 16;  // leaf digests of a round stay in L1 / L2 for the roots
	// scratch per thread, grown once: sixty-four pool threads allocating and freeing these per task met in malloc
	thread_local std::vector<Job> jobs, roots;
	thread_local std::vector<uint8_t> dig;
	thread_local std::vector<size_t> first;
	for (size_t s0 = 0; s0 < n; s0 += kShardsPerRound) {
		const size_t ns = std::min(kShardsPerRound, n - s0);
		size_t nleaf_total = 0;
		first.assign(ns + 1, 0);
		for (size_t s = 0; s < ns; ++s) {
			const size_t len = lens[s0 + s];
			nleaf_total += len ? (len + kShardsumLeaf - 1) / kShardsumLeaf : 1;
			first[s + 1] = nleaf_total;
		}
		dig.resize(nleaf_total * 64);
		jobs.assign(nleaf_total, Job());
		for (size_t s = 0; s < ns; ++s) {
			const size_t len = lens[s0 + s], nleaf = first[s + 1] - first[s];
			for (size_t l = 0; l < nleaf; ++l) {
				Job &j = jobs[first[s] + l];
				const size_t lo = l * kShardsumLeaf;
				j.len = len > lo ? std::min<size_t>(kShardsumLeaf, len - lo) : 0;
				j.p = j.len ? msgs[s0 + s] + lo : nullptr;
				j.x0 = P0;
				j.x1 = l;  // node_offset
				j.x2 = 64ull << 8;
				j.last_node = l + 1 == nleaf;
				j.out = dig.data() + 64 * (first[s] + l);
				j.outlen = 64;
			}
		}
		run_jobs(jobs.data(), jobs.size());
		roots.assign(ns, Job());
		for (size_t s = 0; s < ns; ++s) {
			Job &j = roots[s];
			j.p = dig.data() + 64 * first[s];
			j.len = 64 * (first[s + 1] - first[s]);
			j.x0 = P0;
			j.x1 = 0;
			j.x2 = 1ull | (64ull << 8);
			j.last_node = true;
			j.out = dst ? dst[s0 + s] : out + 32 * (s0 + s);
			j.outlen = 32;
		}
		run_jobs(roots.data(), ns);
	}
}

// The same tree in pieces, for a caller that spreads ONE shard's check over several threads (the streaming get's first
// shard: the first byte waits for nothing else): the 64-byte digests of leaves [leaf_lo, leaf_hi) into dig[64 * leaf],
// and the root over all nleaf of them.
inline size_t shardsum_nleaf(size_t len) { return len ? (len + kShardsumLeaf - 1) / kShardsumLeaf : 1; }

inline void shardsum_leaf_range(const uint8_t *msg, size_t len, size_t leaf_lo, size_t leaf_hi, uint8_t *dig)
{
	const uint64_t P0 = 64ull | (2ull << 24) | ((uint64_t)kShardsumLeaf << 32);
	const size_t nleaf = shardsum_nleaf(len);
	thread_local std::vector<Job> jobs;
	jobs.assign(leaf_hi - leaf_lo, Job());
	for (size_t l = leaf_lo; l < leaf_hi; ++l) {
		Job &j = jobs[l - leaf_lo];
		const size_t lo = l * kShardsumLeaf;
		j.len = len > lo ? std::min<size_t>(kShardsumLeaf, len - lo) : 0;
		j.p = j.len ? msg + lo : nullptr;
		j.x0 = P0;
		j.x1 = l;
		j.x2 = 64ull << 8;
		j.last_node = l + 1 == nleaf;
		j.out = dig + 64 * l;
		j.outlen = 64;
	}
	run_jobs(jobs.data(), jobs.size());
}

inline void shardsum_root(const uint8_t *dig, size_t nleaf, uint8_t out[32])
{
	Job j;
	j.p = dig;
	j.len = 64 * nleaf;
	j.x0 = 64ull | (2ull << 24) | ((uint64_t)kShardsumLeaf << 32);
	j.x1 = 0;
	j.x2 = 1ull | (64ull << 8);
	j.last_node = true;
	j.out = out;
	j.outlen = 32;
	run_jobs(&j, 1);
}

}  // namespace b2host
