// mlh64_host.hpp -- shard checksum v3 on a host core (definition: mlh64.hpp).  Header-only like blake2b_host.hpp, shared by
// libgarage_ec (CPU backend, gec_shardsum3) and libgarage_block (a healthy get checks its shards here, at memory speed,
// instead of crossing the link).  AVX-512 / AVX2 / scalar, picked at run time; every path is the same integer arithmetic.
#pragma once

#include <atomic>
#include <cstring>
#include <vector>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include "blake2b_host.hpp"
#include "mlh64.hpp"

namespace mlh {

inline const uint32_t *keys()
{
	static const KeyTable t = make_keys();
	return t.k;
}

// s = SUM K[i] * w_i over `nwords` whole words starting at key index 0
inline uint64_t leaf_scalar(const uint8_t *p, size_t nwords, const uint32_t *K)
{
	uint64_t s = 0;
	for (size_t i = 0; i < nwords; ++i) {
		uint32_t w;
		std::memcpy(&w, p + 4 * i, 4);
		s += (uint64_t)K[i] * w;
	}
	return s;
}

#if defined(__x86_64__)
__attribute__((target("avx2"))) inline uint64_t leaf_avx2(const uint8_t *p, size_t nwords, const uint32_t *K)
{
	__m256i a0 = _mm256_setzero_si256(), a1 = _mm256_setzero_si256();
	size_t i = 0;
	for (; i + 8 <= nwords; i += 8) {
		const __m256i v = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(p + 4 * i));
		const __m256i k = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(K + i));
		a0 = _mm256_add_epi64(a0, _mm256_mul_epu32(v, k));                                            // words 0,2,4,6
		a1 = _mm256_add_epi64(a1, _mm256_mul_epu32(_mm256_srli_epi64(v, 32), _mm256_srli_epi64(k, 32)));  // words 1,3,5,7
	}
	a0 = _mm256_add_epi64(a0, a1);
	uint64_t lane[4];
	_mm256_storeu_si256(reinterpret_cast<__m256i *>(lane), a0);
	return lane[0] + lane[1] + lane[2] + lane[3] + leaf_scalar(p + 4 * i, nwords - i, K + i);
}

// ahead != 0: while a line is summed, the line `ahead` bytes further on is requested (into L2) -- the caller is about to stream there
// (ec_cpu.cpp: the next chunk of the same shard), and a core that only sums what is already in its cache leaves the memory idle
template <bool PF>
__attribute__((target("avx512f"))) inline uint64_t leaf_avx512_t(const uint8_t *p, size_t nwords, const uint32_t *K, size_t ahead)
{
	__m512i a0 = _mm512_setzero_si512(), a1 = _mm512_setzero_si512();
	size_t i = 0;
	for (; i + 16 <= nwords; i += 16) {
		if (PF)
			_mm_prefetch(reinterpret_cast<const char *>(p + 4 * i + ahead), _MM_HINT_T1);
		const __m512i v = _mm512_loadu_si512(p + 4 * i);
		const __m512i k = _mm512_loadu_si512(K + i);
		a0 = _mm512_add_epi64(a0, _mm512_mul_epu32(v, k));
		a1 = _mm512_add_epi64(a1, _mm512_mul_epu32(_mm512_srli_epi64(v, 32), _mm512_srli_epi64(k, 32)));
	}
	uint64_t lane[8];  // (not _mm512_reduce_add_epi64: gcc spells it with SIGNED 64-bit adds, which must not wrap)
	_mm512_storeu_si512(lane, _mm512_add_epi64(a0, a1));
	return lane[0] + lane[1] + lane[2] + lane[3] + lane[4] + lane[5] + lane[6] + lane[7] + leaf_scalar(p + 4 * i, nwords - i, K + i);
}

__attribute__((target("avx512f"))) inline uint64_t leaf_avx512(const uint8_t *p, size_t nwords, const uint32_t *K)
{
	__m512i a0 = _mm512_setzero_si512(), a1 = _mm512_setzero_si512();
	size_t i = 0;
	for (; i + 16 <= nwords; i += 16) {
		const __m512i v = _mm512_loadu_si512(p + 4 * i);
		const __m512i k = _mm512_loadu_si512(K + i);
		a0 = _mm512_add_epi64(a0, _mm512_mul_epu32(v, k));
		a1 = _mm512_add_epi64(a1, _mm512_mul_epu32(_mm512_srli_epi64(v, 32), _mm512_srli_epi64(k, 32)));
	}
	uint64_t lane[8];  // (not _mm512_reduce_add_epi64: gcc spells it with SIGNED 64-bit adds, which must not wrap)
	_mm512_storeu_si512(lane, _mm512_add_epi64(a0, a1));
	return lane[0] + lane[1] + lane[2] + lane[3] + lane[4] + lane[5] + lane[6] + lane[7] + leaf_scalar(p + 4 * i, nwords - i, K + i);
}
#endif

// 0 = scalar, 1 = AVX2, 2 = AVX-512: the best this host runs, capped by isa_cap() (libgarage_ec lowers it from GEC_CPU_ISA =
// scalar / avx2 when it is loaded, so that the tests reach every path on one box; results never differ)
inline std::atomic<int> &isa_cap()
{
	static std::atomic<int> v{2};
	return v;
}

inline int isa()
{
	static const int hw = [] {
		int best = 0;
#if defined(__x86_64__)
		if (__builtin_cpu_supports("avx2"))
			best = 1;
		if (__builtin_cpu_supports("avx512f"))
			best = 2;
#endif
		return best;
	}();
	const int cap = isa_cap().load(std::memory_order_relaxed);
	return hw < cap ? hw : cap;
}

// the leaf sums of a shard of `len` bytes: out[l], l < nleaf(len).  prefetch_ahead (AVX-512 form only): see leaf_avx512_t.
inline void leaf_sums(const uint8_t *data, size_t len, uint64_t *out, int force_isa = -1, size_t prefetch_ahead = 0)
{
	const uint32_t *K = keys();
	const int how = force_isa >= 0 ? force_isa : isa();
	const size_t nl = nleaf(len);
	for (size_t l = 0; l < nl; ++l) {
		const uint8_t *p = data + l * LEAF_BYTES;
		const size_t n = len - l * LEAF_BYTES < LEAF_BYTES ? len - l * LEAF_BYTES : LEAF_BYTES;
		const size_t nw = n / 4;
		uint64_t s;
#if defined(__x86_64__)
		if (how == 2 && prefetch_ahead)
			s = leaf_avx512_t<true>(p, nw, K, prefetch_ahead);
		else if (how == 2)
			s = leaf_avx512(p, nw, K);
		else if (how == 1)
			s = leaf_avx2(p, nw, K);
		else
#endif
			s = leaf_scalar(p, nw, K);
		if (n & 3) {  // the last, partial word, zero-extended
			uint32_t w = 0;
			std::memcpy(&w, p + 4 * nw, n & 3);
			s += (uint64_t)K[nw] * w;
		}
		out[l] = s;
	}
}

// root of `nl` leaf sums of a shard of `len` bytes
inline void root(uint64_t len, const uint64_t *sums, size_t nl, uint8_t out[32])
{
	b2host::State st;
	const uint64_t hdr[2] = {ROOT_MAGIC, len};
	st.update(reinterpret_cast<const uint8_t *>(hdr), sizeof(hdr));
	st.update(reinterpret_cast<const uint8_t *>(sums), 8 * nl);
	uint8_t full[64];
	st.final(full);
	std::memcpy(out, full, 32);
}

inline void shardsum3(const uint8_t *data, size_t len, uint8_t out[32], int force_isa = -1)
{
	uint64_t stack_sums[64];  // up to 256 KiB shards without an allocation
	std::vector<uint64_t> heap;
	const size_t nl = nleaf(len);
	uint64_t *sums = stack_sums;
	if (nl > 64) {
		heap.resize(nl);
		sums = heap.data();
	}
	leaf_sums(data, len, sums, force_isa);
	root(len, sums, nl, out);
}

// the leaves [leaf_lo, leaf_hi) only (a streaming get checks a shard as it arrives, bm_stream.cpp)
inline void leaf_range(const uint8_t *data, size_t len, size_t leaf_lo, size_t leaf_hi, uint64_t *out)
{
	const size_t nl = nleaf(len);
	if (leaf_hi > nl)
		leaf_hi = nl;
	if (leaf_lo >= leaf_hi)
		return;
	const size_t lo = leaf_lo * LEAF_BYTES, hi = leaf_hi * LEAF_BYTES < len ? leaf_hi * LEAF_BYTES : len;
	leaf_sums(data + lo, hi - lo, out + leaf_lo);
}

}  // namespace mlh
