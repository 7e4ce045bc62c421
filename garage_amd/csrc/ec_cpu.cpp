// ec_cpu.cpp -- GEC_BACKEND_CPU: libgarage_ec's own data path on the host cores, behind the same host-pointer entry
// points as the HIP backend and with identical results.
//
// Why it exists (SURVEY.md Appendix B's `backend` argument, BASELINE config 1): a Garage node that has no GPU, or has
// lost it, must still be able to read and repair its erasure-coded blocks; and the insertion point of the codec is a
// blocking call on a tokio blocking-pool thread either way (src/block/block.rs:85-96), so the caller does not care
// which backend answers.  It is not the product's fast path and it is never used behind a HIP codec's back: a HIP codec
// computes Reed-Solomon on the device or fails (round 5 removed the one opt-in detour there was).  Nothing here includes, links or calls anything under
// oracle/ -- the oracle checks this backend exactly like it checks the kernels (tests/test_cpu_backend.py).
//
// The arithmetic: out[r] = XOR_t coef[r][t] * in[t] over GF(2^8)/0x11D, the same product the kernels compute
// [EXT reed-solomon-erasure core.rs code_some_slices].  Three kernels, picked once per process (GEC_CPU_ISA):
//   avx512+gfni  one VGF2P8AFFINEQB per (64 data bytes, coefficient): multiplication by a constant c is linear over
//                GF(2), i.e. an 8x8 bit matrix per coefficient.  (GF2P8MULB itself is useless here: it is hard-wired
//                to the AES polynomial 0x11B; the affine form takes ANY matrix, so 0x11D is free.)
//   avx2         split-nibble: two VPSHUFB lookups in 16-entry product tables per (32 data bytes, coefficient) -- the
//                CPU twin of the kernels' LDS nibble tables, and what the crate's simd-accel feature does [EXT];
//   scalar       one lookup in a 256-entry product row per (byte, coefficient) -- the crate's default MUL_TABLE form.
// Up to 8 output rows are accumulated in registers per pass over the inputs (each input byte is loaded once per
// pass); a call is cut into (block, 16 KiB column chunk) work items spread over the codec's threads, so even one
// 1 MiB block keeps several cores busy.
#include "ec_internal.hpp"

#if defined(__x86_64__) || defined(_M_X64)
#define GEC_CPU_X86 1
#include <immintrin.h>
#endif

#include <algorithm>
#include <map>

#include "blake2b_mb.hpp"
#include "mlh64_host.hpp"
#include "ec_env.hpp"

namespace gecimpl {
namespace {

enum Isa { ISA_SCALAR = 0, ISA_AVX2 = 1, ISA_GFNI512 = 2 };

Isa detect_isa()
{
#ifdef GEC_CPU_X86
	__builtin_cpu_init();
	const bool avx2 = __builtin_cpu_supports("avx2");
	const bool gfni512 = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("gfni");
#else  // any other host: the scalar kernel (the vector kernels below are x86 intrinsics and are not even compiled)
	const bool avx2 = false, gfni512 = false;
#endif
	const std::string &want = env().cpu_isa;
	if (want == "scalar")
		return ISA_SCALAR;
	if (want == "avx2")
		return avx2 ? ISA_AVX2 : ISA_SCALAR;
	if (want == "gfni")
		return gfni512 ? ISA_GFNI512 : (avx2 ? ISA_AVX2 : ISA_SCALAR);
	return gfni512 ? ISA_GFNI512 : (avx2 ? ISA_AVX2 : ISA_SCALAR);
}

Isa isa()
{
	static const Isa v = detect_isa();
	return v;
}

constexpr int kRowsPerPass = 8;
constexpr size_t kChunk = 16384;  // bytes of every shard per work item

// 256 x 256 product table of the scalar kernel (64 KiB, built on first use)
const uint8_t (*mul_table())[256]
{
	static const std::vector<uint8_t> tab = [] {
		std::vector<uint8_t> t(65536);
		const gec::Field &f = gec::field();
		for (int a = 0; a < 256; ++a)
			for (int b = 0; b < 256; ++b)
				t[(size_t)a * 256 + b] = f.mul((uint8_t)a, (uint8_t)b);
		return t;
	}();
	return reinterpret_cast<const uint8_t(*)[256]>(tab.data());
}

// The 8x8 bit matrix of "multiply by c", in VGF2P8AFFINEQB's layout: result bit i of every byte is the parity of
// (matrix byte [7 - i] AND source byte), so matrix byte 7-i holds row i: bit j set iff bit i of c * 2^j is set.
uint64_t gfni_matrix(uint8_t c)
{
	const gec::Field &f = gec::field();
	uint8_t row[8] = {0, 0, 0, 0, 0, 0, 0, 0};
	for (int j = 0; j < 8; ++j) {
		const uint8_t p = f.mul(c, (uint8_t)(1u << j));
		for (int i = 0; i < 8; ++i)
			if ((p >> i) & 1)
				row[i] |= (uint8_t)(1u << j);
	}
	uint64_t m = 0;
	for (int i = 0; i < 8; ++i)
		m |= (uint64_t)row[i] << (8 * (7 - i));
	return m;
}

// A coefficient matrix (rows x k) expanded for the active kernel, in passes of up to 8 rows.
struct Program {
	int k = 0, rows = 0;
	std::vector<uint8_t> coef;      // rows x k (scalar kernel, and the source of the other two)
	std::vector<uint64_t> affine;   // [pass][t][r in pass]: GFNI bit matrices
	std::vector<uint8_t> nibbles;   // [pass][t][r in pass][lo 16 | hi 16]: split-nibble tables

	Program(const uint8_t *c, int nrows, int kk) : k(kk), rows(nrows), coef(c, c + (size_t)nrows * kk)
	{
		const gec::Field &f = gec::field();
		const int npass = (rows + kRowsPerPass - 1) / kRowsPerPass;
		if (isa() == ISA_GFNI512) {
			affine.resize((size_t)npass * k * kRowsPerPass, 0);
			for (int r = 0; r < rows; ++r)
				for (int t = 0; t < k; ++t)
					affine[((size_t)(r / kRowsPerPass) * k + t) * kRowsPerPass + r % kRowsPerPass] = gfni_matrix(coef[(size_t)r * k + t]);
		} else if (isa() == ISA_AVX2) {
			nibbles.resize((size_t)npass * k * kRowsPerPass * 32, 0);
			for (int r = 0; r < rows; ++r)
				for (int t = 0; t < k; ++t) {
					uint8_t *e = &nibbles[(((size_t)(r / kRowsPerPass) * k + t) * kRowsPerPass + r % kRowsPerPass) * 32];
					const uint8_t cc = coef[(size_t)r * k + t];
					for (int x = 0; x < 16; ++x) {
						e[x] = f.mul(cc, (uint8_t)x);
						e[16 + x] = f.mul(cc, (uint8_t)(x << 4));
					}
				}
		}
	}
};

// ---- kernels: out[r][0..len) = XOR_t coef[r][t] * in[t][0..len) for the R rows of one pass.
// `active` lists the inputs that have bytes in this range (a zero input contributes nothing).

#ifdef GEC_CPU_X86
template <int R>
__attribute__((target("avx512f,avx512bw,gfni"))) void pass_gfni(const uint8_t *const *in, const int *active, int nactive,
								 const uint64_t *mat /* [t][8] of this pass */, uint8_t *const *out, size_t len)
{
	size_t pos = 0;
	for (; pos + 128 <= len; pos += 128) {  // two vectors per trip: every broadcast matrix is used twice
		__m512i a0[R], a1[R];
		for (int r = 0; r < R; ++r)
			a0[r] = a1[r] = _mm512_setzero_si512();
		for (int q = 0; q < nactive; ++q) {
			const int t = active[q];
			const __m512i x0 = _mm512_loadu_si512(in[t] + pos), x1 = _mm512_loadu_si512(in[t] + pos + 64);
			for (int r = 0; r < R; ++r) {
				const __m512i m = _mm512_set1_epi64((long long)mat[(size_t)t * kRowsPerPass + r]);
				a0[r] = _mm512_xor_si512(a0[r], _mm512_gf2p8affine_epi64_epi8(x0, m, 0));
				a1[r] = _mm512_xor_si512(a1[r], _mm512_gf2p8affine_epi64_epi8(x1, m, 0));
			}
		}
		for (int r = 0; r < R; ++r) {
			_mm512_storeu_si512(out[r] + pos, a0[r]);
			_mm512_storeu_si512(out[r] + pos + 64, a1[r]);
		}
	}
	for (; pos < len; pos += 64) {  // last vectors, the final one masked
		const size_t left = len - pos;
		const __mmask64 mk = left >= 64 ? ~0ull : ((1ull << left) - 1);
		__m512i a[R];
		for (int r = 0; r < R; ++r)
			a[r] = _mm512_setzero_si512();
		for (int q = 0; q < nactive; ++q) {
			const int t = active[q];
			const __m512i x = _mm512_maskz_loadu_epi8(mk, in[t] + pos);
			for (int r = 0; r < R; ++r)
				a[r] = _mm512_xor_si512(a[r], _mm512_gf2p8affine_epi64_epi8(x, _mm512_set1_epi64((long long)mat[(size_t)t * kRowsPerPass + r]), 0));
		}
		for (int r = 0; r < R; ++r)
			_mm512_mask_storeu_epi8(out[r] + pos, mk, a[r]);
	}
}

void scalar_range(const uint8_t *const *in, const int *active, int nactive, const uint8_t *coef, int k, int row0, int R,
		  uint8_t *const *out, size_t pos, size_t len)
{
	const uint8_t(*MUL)[256] = mul_table();
	for (int r = 0; r < R; ++r) {
		uint8_t *o = out[r];
		std::memset(o + pos, 0, len - pos);
		for (int q = 0; q < nactive; ++q) {
			const int t = active[q];
			const uint8_t *row = MUL[coef[(size_t)(row0 + r) * k + t]];
			const uint8_t *src = in[t];
			for (size_t i = pos; i < len; ++i)
				o[i] ^= row[src[i]];
		}
	}
}

template <int R>
__attribute__((target("avx2"))) void pass_avx2(const uint8_t *const *in, const int *active, int nactive,
						const uint8_t *tab /* [t][8][32] of this pass */, uint8_t *const *out, size_t len,
						const uint8_t *coef, int k, int row0)
{
	const __m256i low = _mm256_set1_epi8(0x0f);
	size_t pos = 0;
	for (; pos + 32 <= len; pos += 32) {
		__m256i a[R];
		for (int r = 0; r < R; ++r)
			a[r] = _mm256_setzero_si256();
		for (int q = 0; q < nactive; ++q) {
			const int t = active[q];
			const __m256i x = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(in[t] + pos));
			const __m256i xl = _mm256_and_si256(x, low), xh = _mm256_and_si256(_mm256_srli_epi64(x, 4), low);
			for (int r = 0; r < R; ++r) {
				const uint8_t *e = tab + ((size_t)t * kRowsPerPass + r) * 32;
				const __m256i lo = _mm256_broadcastsi128_si256(_mm_loadu_si128(reinterpret_cast<const __m128i *>(e)));
				const __m256i hi = _mm256_broadcastsi128_si256(_mm_loadu_si128(reinterpret_cast<const __m128i *>(e + 16)));
				a[r] = _mm256_xor_si256(a[r], _mm256_xor_si256(_mm256_shuffle_epi8(lo, xl), _mm256_shuffle_epi8(hi, xh)));
			}
		}
		for (int r = 0; r < R; ++r)
			_mm256_storeu_si256(reinterpret_cast<__m256i *>(out[r] + pos), a[r]);
	}
	if (pos < len)
		scalar_range(in, active, nactive, coef, k, row0, R, out, pos, len);
}
#endif  // GEC_CPU_X86

template <int R>
void run_pass(const Program &p, int pass, const uint8_t *const *in, const int *active, int nactive, uint8_t *const *out, size_t len)
{
	switch (isa()) {
#ifdef GEC_CPU_X86
	case ISA_GFNI512:
		pass_gfni<R>(in, active, nactive, p.affine.data() + (size_t)pass * p.k * kRowsPerPass, out, len);
		break;
	case ISA_AVX2:
		pass_avx2<R>(in, active, nactive, p.nibbles.data() + (size_t)pass * p.k * kRowsPerPass * 32, out, len, p.coef.data(), p.k,
			     pass * kRowsPerPass);
		break;
#endif
	default:
		scalar_range(in, active, nactive, p.coef.data(), p.k, pass * kRowsPerPass, R, out, 0, len);
	}
}

// out[r][0..len) for all rows of the program; in[t] are pointers to the first byte of this column range,
// avail[t] = how many of its `len` bytes exist (the rest reads as zero: a block's last data shard is short).
void apply_range(const Program &p, const uint8_t *const *in, const size_t *avail, uint8_t *const *out, size_t len)
{
	const int k = p.k;
	// one input may straddle the end of the data: it is padded into a private buffer (at most one per block).
	// All scratch is per thread and only ever grows: a work item is ~10 us of arithmetic, and five heap allocations per
	// item put every thread of the pool on the allocator's locks.
	thread_local std::vector<uint8_t> pad;
	thread_local std::vector<const uint8_t *> src;
	thread_local std::vector<int> active;
	src.assign(in, in + k);
	active.clear();
	size_t npad = 0;
	for (int t = 0; t < k; ++t)
		if (avail[t] > 0 && avail[t] < len)
			++npad;
	if (pad.size() < npad * len)
		pad.resize(npad * len);
	npad = 0;
	for (int t = 0; t < k; ++t) {
		if (avail[t] == 0)
			continue;
		if (avail[t] < len) {
			uint8_t *q = pad.data() + npad++ * len;
			std::memcpy(q, in[t], avail[t]);
			std::memset(q + avail[t], 0, len - avail[t]);
			src[t] = q;
		}
		active.push_back(t);
	}
	const int npass = (p.rows + kRowsPerPass - 1) / kRowsPerPass;
	for (int pass = 0; pass < npass; ++pass) {
		const int R = std::min(kRowsPerPass, p.rows - pass * kRowsPerPass);
		uint8_t *const *o = out + (size_t)pass * kRowsPerPass;
		const int na = (int)active.size();
		switch (R) {
		case 1: run_pass<1>(p, pass, src.data(), active.data(), na, o, len); break;
		case 2: run_pass<2>(p, pass, src.data(), active.data(), na, o, len); break;
		case 3: run_pass<3>(p, pass, src.data(), active.data(), na, o, len); break;
		case 4: run_pass<4>(p, pass, src.data(), active.data(), na, o, len); break;
		case 5: run_pass<5>(p, pass, src.data(), active.data(), na, o, len); break;
		case 6: run_pass<6>(p, pass, src.data(), active.data(), na, o, len); break;
		case 7: run_pass<7>(p, pass, src.data(), active.data(), na, o, len); break;
		default: run_pass<8>(p, pass, src.data(), active.data(), na, o, len); break;
		}
	}
}

struct CpuBackend : Backend {
	const gec_codec *c = nullptr;
	std::unique_ptr<ForkJoinPool> pool;

	// One unit of the product: block `b` of a bucket, all rows, every column chunk.
	struct Job {
		const Program *prog;
		const uint8_t *const *in;  // k shard pointers (whole shards)
		const size_t *valid;       // k: bytes of each shard that exist (<= S)
		uint8_t *const *out;       // rows output pointers (whole shards)
	};

	// Leaf sums of checksum v3 left by the encode's own pass (the host analogue of the kernels' SUM form): entry
	// [(job * nshard + slot) * nleaf + leaf]; slots 0..k-1 are a job's inputs, k.. its output rows.  kChunk is a whole number
	// of leaves, so a (job, chunk) item owns the leaves it touches; bytes a short input does not have contribute nothing
	// ("zero bytes contribute nothing", mlh64.hpp), their leaves keep the 0 the array was filled with.
	struct LeafOut {
		uint64_t *sums = nullptr;
		size_t nleaf = 0, nshard = 0;
	};
	static_assert(kChunk % mlh::LEAF_BYTES == 0, "a chunk must cover whole checksum leaves");

	// runs every (job, chunk) on the pool
	void run_jobs(const std::vector<Job> &jobs, size_t S, const LeafOut *lo = nullptr) const
	{
		const size_t nch = (S + kChunk - 1) / kChunk;
		pool->parallel_for(jobs.size() * nch, [&](size_t item) {
			const size_t ji = item / nch;
			const Job &j = jobs[ji];
			const size_t off = (item % nch) * kChunk, len = std::min(kChunk, S - off);
			const int k = j.prog->k, rows = j.prog->rows;
			thread_local std::vector<const uint8_t *> in;
			thread_local std::vector<size_t> avail;
			thread_local std::vector<uint8_t *> out;
			in.resize(k);
			avail.resize(k);
			out.resize(rows);
			for (int t = 0; t < k; ++t) {
				in[t] = j.in[t] + off;
				avail[t] = j.valid[t] > off ? std::min(len, j.valid[t] - off) : 0;
			}
			for (int r = 0; r < rows; ++r)
				out[r] = j.out[r] + off;
			apply_range(*j.prog, in.data(), avail.data(), out.data(), len);
			if (lo) {  // while the chunk is in this core's cache
				uint64_t *base = lo->sums + ji * lo->nshard * lo->nleaf + off / mlh::LEAF_BYTES;
				// (while the chunk's inputs are summed out of this core's cache, the same shards' NEXT chunk -- the pool's next
				// item on this block -- is requested from memory: a core that only sums leaves the memory system idle for a third
				// of the item on a host that is bound by it)
				for (int t = 0; t < k; ++t)
					if (avail[t])
						mlh::leaf_sums(in[t], avail[t], base + (size_t)t * lo->nleaf, -1, j.valid[t] > off + 2 * kChunk ? kChunk : 0);
				for (int r = 0; r < rows; ++r)
					mlh::leaf_sums(out[r], len, base + (size_t)(k + r) * lo->nleaf);
			}
		});
	}

	// roots of checksum v3 from leaf sums laid out as LeafOut says: every shard is S bytes long (short ones zero-extended)
	void roots_from_leaves(const LeafOut &lo, size_t nshards_total, size_t S, uint8_t *shard_sums)
	{
		constexpr size_t kPer = 64;
		pool->parallel_for((nshards_total + kPer - 1) / kPer, [&](size_t g) {
			for (size_t q = g * kPer; q < std::min(nshards_total, (g + 1) * kPer); ++q)
				mlh::root(S, lo.sums + q * lo.nleaf, lo.nleaf, shard_sums + 32 * q);
		});
	}

	// shard checksums of many S-byte shards, eight chains at a time per core (blake2b_mb.hpp), kShardTask shards per pool task.
	// `have` < S: only that many bytes exist at p, the shard is those bytes zero-extended to S -- v3 sums the bytes that exist and
	// binds S in the root; v2 (a hash of every leaf's bytes) extends the shard in a buffer of the pool thread that hashes it
	struct ShardRef {
		const uint8_t *p;
		uint8_t *dst;
		size_t have = SIZE_MAX;
	};
	void shardsums(const std::vector<ShardRef> &list, size_t S)
	{
		constexpr size_t kShardTaskMax = 16;
		// (one shard per task where a core hashes one chain at a time: the scalar fallback keeps the pool busy)
		const size_t kShardTask = b2host::mb_available() || c->sumkind == GEC_SHARDSUM_MLH64 ? kShardTaskMax : 1;
		const size_t nl = mlh::nleaf(S);
		pool->parallel_for((list.size() + kShardTask - 1) / kShardTask, [&](size_t g) {
			const size_t i0 = g * kShardTask, cnt = std::min(kShardTask, list.size() - i0);
			if (c->sumkind == GEC_SHARDSUM_MLH64) {  // checksum v3: memory speed on one core (mlh64_host.hpp)
				thread_local std::vector<uint64_t> sums;
				for (size_t i = 0; i < cnt; ++i) {
					const ShardRef &r = list[i0 + i];
					if (r.have >= S) {
						mlh::shardsum3(r.p, S, r.dst);
						continue;
					}
					sums.assign(nl, 0);
					if (r.have)
						mlh::leaf_sums(r.p, r.have, sums.data());
					mlh::root(S, sums.data(), nl, r.dst);
				}
				return;
			}
			thread_local std::vector<uint8_t> padbuf;
			size_t npad = 0;
			for (size_t i = 0; i < cnt; ++i)
				npad += list[i0 + i].have < S ? 1 : 0;
			if (padbuf.size() < npad * S)
				padbuf.resize(npad * S);
			npad = 0;
			const uint8_t *ptr[kShardTaskMax];
			uint8_t *dst[kShardTaskMax];
			size_t len[kShardTaskMax];
			for (size_t i = 0; i < cnt; ++i) {
				const ShardRef &r = list[i0 + i];
				ptr[i] = r.p;
				if (r.have < S) {
					uint8_t *q = padbuf.data() + npad++ * S;
					if (r.have)
						std::memcpy(q, r.p, r.have);
					std::memset(q + r.have, 0, S - r.have);
					ptr[i] = q;
				}
				dst[i] = r.dst;
				len[i] = S;
			}
			b2host::shardsum_many(ptr, len, cnt, nullptr, dst);
		});
	}

	int encode_batch(size_t nblocks, const uint8_t *const *blocks, const size_t *block_len, size_t S, uint8_t *const *parity,
			 uint8_t *shard_sums) override
	{
		const size_t k = c->k, m = c->m, n = k + m;
		const Program prog(c->enc.row((int)k), (int)m, (int)k);
		std::vector<const uint8_t *> in(nblocks * k);
		std::vector<size_t> valid(nblocks * k);
		std::vector<uint8_t *> out(nblocks * m);
		std::vector<Job> jobs(nblocks);
		for (size_t b = 0; b < nblocks; ++b) {
			for (size_t t = 0; t < k; ++t) {
				in[b * k + t] = blocks[b] + t * S;  // never dereferenced beyond valid[]
				valid[b * k + t] = block_len[b] > t * S ? std::min(S, block_len[b] - t * S) : 0;
			}
			for (size_t r = 0; r < m; ++r)
				out[b * m + r] = parity[b] + r * S;
			jobs[b] = Job{&prog, &in[b * k], &valid[b * k], &out[b * m]};
		}
		if (shard_sums && c->sumkind == GEC_SHARDSUM_MLH64) {
			// checksum v3: the leaf sums of all k + m shards come out of the encode's own pass, then one small root per shard
			std::vector<uint64_t> leaves(nblocks * n * mlh::nleaf(S), 0);
			const LeafOut lo{leaves.data(), mlh::nleaf(S), n};
			run_jobs(jobs, S, &lo);
			roots_from_leaves(lo, nblocks * n, S, shard_sums);
			return GEC_OK;
		}
		run_jobs(jobs, S);
		if (!shard_sums)
			return GEC_OK;
		// the checksum of every shard, data shards as zero-extended to S bytes (extended by the pool task that hashes them)
		std::vector<ShardRef> list(nblocks * n);
		for (size_t q = 0; q < nblocks * n; ++q) {
			const size_t b = q / n, j = q % n;
			uint8_t *dst = shard_sums + 32 * q;
			if (j >= k)
				list[q] = {parity[b] + (j - k) * S, dst, S};
			else
				list[q] = {blocks[b] + j * S, dst, valid[b * k + j]};
		}
		shardsums(list, S);
		return GEC_OK;
	}

	int verify_batch(size_t nblocks, const uint8_t *const *shards, size_t S, uint8_t *ok) override
	{
		const size_t k = c->k, m = c->m, n = k + m;
		const Program prog(c->enc.row((int)k), (int)m, (int)k);
		const size_t nch = (S + kChunk - 1) / kChunk;
		std::fill(ok, ok + nblocks, (uint8_t)1);
		pool->parallel_for(nblocks * nch, [&](size_t item) {
			const size_t b = item / nch, off = (item % nch) * kChunk, len = std::min(kChunk, S - off);
			thread_local std::vector<uint8_t> tmp;
			thread_local std::vector<const uint8_t *> in;
			thread_local std::vector<size_t> avail;
			thread_local std::vector<uint8_t *> out;
			if (tmp.size() < m * kChunk)
				tmp.resize(m * kChunk);
			in.resize(k);
			avail.assign(k, len);
			out.resize(m);
			for (size_t t = 0; t < k; ++t)
				in[t] = shards[b * n + t] + off;
			for (size_t r = 0; r < m; ++r)
				out[r] = tmp.data() + r * kChunk;
			apply_range(prog, in.data(), avail.data(), out.data(), len);
			for (size_t r = 0; r < m; ++r)
				if (std::memcmp(out[r], shards[b * n + k + r] + off, len) != 0) {
					__atomic_store_n(&ok[b], (uint8_t)0, __ATOMIC_RELAXED);
					break;
				}
		});
		return GEC_OK;
	}

	int verify_hash_batch(size_t nblocks, const uint8_t *const *shards, size_t S, uint8_t *ok, uint8_t *shard_sums) override
	{
		int rc = verify_batch(nblocks, shards, S, ok);
		if (rc)
			return rc;
		const size_t n = (size_t)c->k + c->m;
		std::vector<ShardRef> list(nblocks * n);
		for (size_t q = 0; q < nblocks * n; ++q)
			list[q] = {shards[q], shard_sums + 32 * q};
		shardsums(list, S);
		return GEC_OK;
	}

	// The strided ("device-style") reconstruct on HOST memory: shard j of block b at base + b*block_stride + shard_off[j]
	// (contiguous stripes when shard_off is NULL), only bytes [byte_off, byte_off + byte_len) of every shard touched,
	// missing shards rebuilt in place.  This is what a rank of a striped-object group that runs on the host cores calls
	// for its byte range of the gathered buffer (ec_hip_group.cpp, garage_amd/striped.py on CPU tensors); the stream
	// argument means nothing here.
	int reconstruct_dev(size_t nblocks, void *d_base, size_t block_stride, const size_t *shard_off, size_t S, const uint8_t *present,
			    int data_only, size_t byte_off, size_t byte_len, void *) override
	{
		const size_t k = c->k, n = (size_t)c->k + c->m;
		if (byte_len == 0)
			return GEC_OK;
		std::vector<const uint8_t *> sp(nblocks * n, nullptr);
		std::vector<uint8_t *> op(nblocks * n, nullptr);
		uint8_t *base = static_cast<uint8_t *>(d_base);
		for (size_t b = 0; b < nblocks; ++b)
			for (size_t j = 0; j < n; ++j) {
				uint8_t *p = base + b * block_stride + (shard_off ? shard_off[j] : j * S) + byte_off;
				if (present[j])
					sp[b * n + j] = p;
				else if (!(data_only && j >= k))
					op[b * n + j] = p;
			}
		return reconstruct_batch(nblocks, sp.data(), op.data(), byte_len, data_only, nullptr, nullptr);
	}

	// ... with a pattern per block: the pointer form takes any mix of patterns in one call
	int reconstruct_dev_ex(size_t nblocks, void *d_stripes, size_t stride, size_t S, const uint8_t *present, int data_only, void *) override
	{
		const size_t k = c->k, n = (size_t)c->k + c->m;
		std::vector<const uint8_t *> sp(nblocks * n, nullptr);
		std::vector<uint8_t *> op(nblocks * n, nullptr);
		uint8_t *base = static_cast<uint8_t *>(d_stripes);
		for (size_t b = 0; b < nblocks; ++b)
			for (size_t j = 0; j < n; ++j) {
				uint8_t *p = base + b * stride + j * S;
				if (present[b * n + j])
					sp[b * n + j] = p;
				else if (!(data_only && j >= k))
					op[b * n + j] = p;
			}
		return reconstruct_batch(nblocks, sp.data(), op.data(), S, data_only, nullptr, nullptr);
	}

	int reconstruct_batch(size_t nblocks, const uint8_t *const *shards, uint8_t *const *out, size_t S, int data_only, uint8_t *in_sums,
			      uint8_t *out_sums) override
	{
		const size_t k = c->k, n = (size_t)c->k + c->m;
		// bucket blocks by erasure pattern AND by which of the missing shards the caller wants back (out entry
		// non-NULL; with data_only parity is never wanted): one decode plan and one program per bucket, only the
		// wanted rows computed.  key[j]: 1 present, 0 missing + wanted, 2 missing + unwanted
		std::map<std::string, std::vector<size_t>> buckets;
		for (size_t b = 0; b < nblocks; ++b) {
			std::string key(n, 0);
			size_t nwanted = 0;
			for (size_t j = 0; j < n; ++j) {
				if (shards[b * n + j])
					key[j] = 1;
				else if ((data_only && j >= k) || !out[b * n + j])
					key[j] = 2;
				else
					++nwanted;
			}
			if (nwanted)
				buckets[key].push_back(b);
		}
		struct Work {
			std::shared_ptr<const Plan> plan;
			std::vector<int> wanted;  // shard indices, ascending
			std::unique_ptr<Program> prog;
			const std::vector<size_t> *ids;
		};
		std::vector<Work> work;
		size_t njobs = 0;
		for (auto &kv : buckets) {
			std::string pres(kv.first);
			for (auto &ch : pres)
				ch = ch == 1 ? 1 : 0;
			Work w;
			int rc = get_plan(c, reinterpret_cast<const uint8_t *>(pres.data()), false, w.plan);
			if (rc)
				return rc;
			std::vector<uint8_t> rows;
			for (size_t r = 0; r < w.plan->missing.size(); ++r)
				if (kv.first[w.plan->missing[r]] == 0) {
					w.wanted.push_back(w.plan->missing[r]);
					rows.insert(rows.end(), w.plan->rows.row((int)r), w.plan->rows.row((int)r) + k);
				}
			if (w.wanted.empty())
				continue;
			w.prog.reset(new Program(rows.data(), (int)w.wanted.size(), (int)k));
			w.ids = &kv.second;
			njobs += kv.second.size();
			work.push_back(std::move(w));
		}
		std::vector<const uint8_t *> in;
		std::vector<uint8_t *> outp;
		std::vector<Job> jobs;
		in.reserve(njobs * k);
		jobs.reserve(njobs);
		size_t nout = 0;
		for (const Work &w : work)
			nout += w.ids->size() * w.wanted.size();
		outp.reserve(nout);
		const std::vector<size_t> full(k, S);
		for (const Work &w : work)
			for (size_t b : *w.ids) {
				const size_t i0 = in.size(), o0 = outp.size();
				for (size_t t = 0; t < k; ++t)
					in.push_back(shards[b * n + w.plan->valid[t]]);
				for (int j : w.wanted)
					outp.push_back(out[b * n + j]);
				jobs.push_back(Job{w.prog.get(), in.data() + i0, full.data(), outp.data() + o0});
			}
		run_jobs(jobs, S);
		if (!in_sums)
			return GEC_OK;
		// the checksums of the k shards that were read and of the shards that were written
		std::vector<ShardRef> sums;
		for (const Work &w : work)
			for (size_t b : *w.ids) {
				for (size_t t = 0; t < k; ++t)
					sums.push_back({shards[b * n + w.plan->valid[t]], in_sums + 32 * (b * n + w.plan->valid[t])});
				for (int j : w.wanted)
					sums.push_back({out[b * n + j], out_sums + 32 * (b * n + j)});
			}
		shardsums(sums, S);
		return GEC_OK;
	}

	int decode_verify_batch(size_t nblocks, const uint8_t *const *shards, size_t S, const size_t *block_len, uint8_t *const *rebuilt,
				uint8_t *shard_sums, uint8_t *block_sums) override
	{
		const size_t k = c->k, n = (size_t)c->k + c->m;
		// the first k present shards of every block are "read" (the crate's rule); missing data shards are rebuilt
		// through gec_reconstruct_batch's machinery with only those k marked present
		std::vector<const uint8_t *> used(nblocks * n, nullptr);
		std::vector<uint8_t *> want(nblocks * n, nullptr);
		bool any_missing = false;
		for (size_t b = 0; b < nblocks; ++b) {
			size_t seen = 0;
			for (size_t j = 0; j < n && seen < k; ++j)
				if (shards[b * n + j]) {
					used[b * n + j] = shards[b * n + j];
					++seen;
				}
			for (size_t j = 0; j < k; ++j)
				if (!shards[b * n + j]) {
					want[b * n + j] = rebuilt[b * n + j];
					any_missing = true;
				}
		}
		if (any_missing) {
			int rc = reconstruct_batch(nblocks, used.data(), want.data(), S, /*data_only=*/1, nullptr, nullptr);
			if (rc)
				return rc;
		}
		// checksums: every shard that was read, and (optionally) the block itself from its k data shards
		std::vector<ShardRef> list;
		list.reserve(nblocks * n);
		for (size_t b = 0; b < nblocks; ++b)
			for (size_t j = 0; j < n; ++j)
				if (used[b * n + j])
					list.push_back({used[b * n + j], shard_sums + 32 * (b * n + j)});
		shardsums(list, S);
		if (block_sums) {
			// a block is one chain over its k data shards (read or rebuilt): eight blocks at a time per core
			std::vector<const uint8_t *> piece(nblocks * k);
			for (size_t b = 0; b < nblocks; ++b)
				for (size_t t = 0; t < k; ++t)
					piece[b * k + t] = shards[b * n + t] ? shards[b * n + t] : rebuilt[b * n + t];
			const size_t per = b2host::mb_available() ? 8 : 1;
			pool->parallel_for((nblocks + per - 1) / per, [&](size_t g) {
				const size_t b0 = g * per, cnt = std::min<size_t>(per, nblocks - b0);
				b2host::Job jobs[8];
				for (size_t i = 0; i < cnt; ++i) {
					jobs[i].pieces = &piece[(b0 + i) * k];
					jobs[i].piece_len = S;
					jobs[i].len = block_len[b0 + i];
					jobs[i].out = block_sums + 32 * (b0 + i);
				}
				b2host::run_jobs(jobs, cnt);
			});
		}
		return GEC_OK;
	}

	int hash_batch(size_t nmsg, const uint8_t *const *msgs, const size_t *lens, uint8_t *out, bool tree) override
	{
		// eight chains at a time per core: tasks of 8 (plain) / 16 (tree: the leaves are the chains) messages
		// (one message per task without such lanes: GEC_CPU_ISA=scalar / avx2, or a host without AVX-512)
		const size_t per = !b2host::mb_available() ? 1 : tree ? 16 : 8;
		pool->parallel_for((nmsg + per - 1) / per, [&](size_t g) {
			const size_t i0 = g * per, cnt = std::min(per, nmsg - i0);
			if (tree && c->sumkind == GEC_SHARDSUM_MLH64)
				for (size_t i = i0; i < i0 + cnt; ++i)
					mlh::shardsum3(msgs[i], lens[i], out + 32 * i);
			else if (tree)
				b2host::shardsum_many(msgs + i0, lens + i0, cnt, out + 32 * i0);
			else
				b2host::blake2sum_many(msgs + i0, lens + i0, cnt, out + 32 * i0);
		});
		return GEC_OK;
	}
};

}  // namespace

int make_cpu_backend(gec_codec *c, std::unique_ptr<Backend> &out)
{
	std::unique_ptr<CpuBackend> be(new (std::nothrow) CpuBackend());
	if (!be)
		return fail(GEC_E_NOMEM, "alloc backend");
	be->c = c;
	// a background codec keeps to a quarter of the threads: repair must not take the cores the request path needs
	const int threads = c->qos_class == GEC_CLASS_BACKGROUND ? std::max(1, env().cpu_threads / 4) : env().cpu_threads;
	be->pool.reset(new ForkJoinPool((unsigned)std::max(0, threads - 1)));  // the calling thread works too
	(void)mul_table();
	out = std::move(be);
	return GEC_OK;
}

}  // namespace gecimpl

extern "C" const char *gec_cpu_isa(void)
{
	switch (gecimpl::isa()) {
	case gecimpl::ISA_GFNI512: return "avx512+gfni";
	case gecimpl::ISA_AVX2: return "avx2";
	default: return "scalar";
	}
}
