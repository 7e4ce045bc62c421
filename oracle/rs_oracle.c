/*
 * rs_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 * See rs_oracle.h for provenance and the "parity unpinned" statement.
 *
 * Each function names the part of reed-solomon-erasure (galois_8) [EXT] it
 * restates and the SURVEY.md Appendix A paragraph that specifies it.
 */
#include "rs_oracle.h"

#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#if defined(__x86_64__)
#include <immintrin.h>
#endif

/* ------------------------------------------------------------------ field */
/* [EXT] galois_8 build.rs: LOG_TABLE / EXP_TABLE / MUL_TABLE generated from
 * the "generating polynomial 29" (0x11D), generator 2.  Appendix A.1. */
static uint8_t EXP_T[512];
static uint8_t LOG_T[256];
static uint8_t MUL_T[256][256];
static int tables_ready;

static void init_tables(void)
{
	if (tables_ready)
		return;
	unsigned x = 1;
	for (int i = 0; i < 255; i++) {
		EXP_T[i] = (uint8_t)x;
		LOG_T[x] = (uint8_t)i;
		x <<= 1;
		if (x & 0x100)
			x ^= 0x11D;
	}
	for (int i = 255; i < 512; i++)
		EXP_T[i] = EXP_T[i - 255];
	LOG_T[0] = 0;
	for (int a = 0; a < 256; a++)
		for (int b = 0; b < 256; b++)
			MUL_T[a][b] = (a == 0 || b == 0)
					      ? 0
					      : EXP_T[LOG_T[a] + LOG_T[b]];
	tables_ready = 1;
}

__attribute__((constructor)) static void ctor(void) { init_tables(); }

const uint8_t *rso_exp_table(void) { init_tables(); return EXP_T; }
const uint8_t *rso_log_table(void) { init_tables(); return LOG_T; }

uint8_t rso_gf_mul(uint8_t a, uint8_t b)
{
	init_tables();
	return MUL_T[a][b];
}

uint8_t rso_gf_div(uint8_t a, uint8_t b)
{
	init_tables();
	if (a == 0)
		return 0;
	int d = (int)LOG_T[a] - (int)LOG_T[b];
	if (d < 0)
		d += 255;
	return EXP_T[d];
}

/* [EXT] galois_8::exp(a, n): 1 if n==0; 0 if a==0; EXP[(LOG[a]*n) mod 255] */
uint8_t rso_gf_exp(uint8_t a, unsigned n)
{
	init_tables();
	if (n == 0)
		return 1;
	if (a == 0)
		return 0;
	return EXP_T[((unsigned)LOG_T[a] * n) % 255];
}

/* --------------------------------------------------------------- matrices */
/* [EXT] matrix.rs Matrix::vandermonde: m[r][c] = exp(r, c).  Appendix A.2. */
void rso_vandermonde(int rows, int cols, uint8_t *out)
{
	for (int r = 0; r < rows; r++)
		for (int c = 0; c < cols; c++)
			out[r * cols + c] = rso_gf_exp((uint8_t)r, (unsigned)c);
}

/* [EXT] matrix.rs Matrix::invert -> augment with identity, gaussian_elim
 * (pivot search below, scale row, clear below, then clear above). */
int rso_invert(int n, const uint8_t *in, uint8_t *out)
{
	int w = 2 * n;
	uint8_t *a = (uint8_t *)calloc((size_t)n * w, 1);
	if (!a)
		return RSO_SINGULAR;
	for (int r = 0; r < n; r++) {
		memcpy(a + r * w, in + r * n, (size_t)n);
		a[r * w + n + r] = 1;
	}
	for (int r = 0; r < n; r++) {
		if (a[r * w + r] == 0) {
			int rb;
			for (rb = r + 1; rb < n; rb++)
				if (a[rb * w + r] != 0)
					break;
			if (rb == n) {
				free(a);
				return RSO_SINGULAR;
			}
			for (int c = 0; c < w; c++) {
				uint8_t t = a[r * w + c];
				a[r * w + c] = a[rb * w + c];
				a[rb * w + c] = t;
			}
		}
		uint8_t piv = a[r * w + r];
		if (piv != 1) {
			uint8_t s = rso_gf_div(1, piv);
			for (int c = 0; c < w; c++)
				a[r * w + c] = rso_gf_mul(a[r * w + c], s);
		}
		for (int rb = r + 1; rb < n; rb++) {
			uint8_t f = a[rb * w + r];
			if (f)
				for (int c = 0; c < w; c++)
					a[rb * w + c] ^= rso_gf_mul(f, a[r * w + c]);
		}
	}
	for (int d = 0; d < n; d++)
		for (int ra = 0; ra < d; ra++) {
			uint8_t f = a[ra * w + d];
			if (f)
				for (int c = 0; c < w; c++)
					a[ra * w + c] ^= rso_gf_mul(f, a[d * w + c]);
		}
	for (int r = 0; r < n; r++)
		memcpy(out + r * n, a + r * w + n, (size_t)n);
	free(a);
	return RSO_OK;
}

static int check_km(int k, int m)
{
	if (k <= 0)
		return RSO_TOO_FEW_DATA;
	if (m <= 0)
		return RSO_TOO_FEW_PARITY;
	if (k + m > 256)
		return RSO_TOO_MANY_SHARDS;
	return RSO_OK;
}

/* [EXT] core.rs ReedSolomon::build_matrix: vandermonde(n,k) * invert(top). */
int rso_build_matrix(int k, int m, uint8_t *out)
{
	int rc = check_km(k, m);
	if (rc)
		return rc;
	int n = k + m;
	uint8_t *v = (uint8_t *)malloc((size_t)n * k);
	uint8_t *ti = (uint8_t *)malloc((size_t)k * k);
	if (!v || !ti) {
		free(v);
		free(ti);
		return RSO_SINGULAR;
	}
	rso_vandermonde(n, k, v);
	rc = rso_invert(k, v, ti); /* top k rows of v are the first k*k bytes */
	if (rc == RSO_OK)
		for (int r = 0; r < n; r++)
			for (int c = 0; c < k; c++) {
				uint8_t acc = 0;
				for (int t = 0; t < k; t++)
					acc ^= rso_gf_mul(v[r * k + t], ti[t * k + c]);
				out[r * k + c] = acc;
			}
	free(v);
	free(ti);
	return rc;
}

/* [EXT] core.rs get_data_decode_matrix: rows `valid` of M, inverted, where
 * `valid` = first k present shard indices in ascending order.  A.3. */
int rso_decode_matrix(int k, int m, const uint8_t *present, int *valid,
		      uint8_t *decode)
{
	int rc = check_km(k, m);
	if (rc)
		return rc;
	int n = k + m, nv = 0;
	for (int j = 0; j < n && nv < k; j++)
		if (present[j])
			valid[nv++] = j;
	if (nv < k)
		return RSO_TOO_FEW_PRESENT;
	uint8_t *M = (uint8_t *)malloc((size_t)n * k);
	uint8_t *sub = (uint8_t *)malloc((size_t)k * k);
	if (!M || !sub) {
		free(M);
		free(sub);
		return RSO_SINGULAR;
	}
	rc = rso_build_matrix(k, m, M);
	if (rc == RSO_OK) {
		for (int t = 0; t < k; t++)
			memcpy(sub + t * k, M + valid[t] * k, (size_t)k);
		rc = rso_invert(k, sub, decode);
	}
	free(M);
	free(sub);
	return rc;
}

/* -------------------------------------------------------------- inner ops */
/* [EXT] galois_8::mul_slice / mul_slice_xor, pure-Rust path: one MUL_TABLE
 * row lookup per byte. */
static void mul_slice_scalar(uint8_t c, const uint8_t *in, uint8_t *out,
			     size_t n, int xor_into)
{
	const uint8_t *row = MUL_T[c];
	if (xor_into)
		for (size_t i = 0; i < n; i++)
			out[i] ^= row[in[i]];
	else
		for (size_t i = 0; i < n; i++)
			out[i] = row[in[i]];
}

int rso_has_avx2(void)
{
#if defined(__x86_64__)
	return __builtin_cpu_supports("avx2");
#else
	return 0;
#endif
}

#if defined(__x86_64__)
/* [EXT] `simd-accel` feature: split-nibble pshufb (two 16-entry tables per
 * coefficient).  Bit-identical to the scalar path. */
__attribute__((target("avx2"))) static void
mul_slice_avx2(uint8_t c, const uint8_t *in, uint8_t *out, size_t n,
	       int xor_into)
{
	uint8_t lo[16], hi[16];
	for (int i = 0; i < 16; i++) {
		lo[i] = MUL_T[c][i];
		hi[i] = MUL_T[c][i << 4];
	}
	__m128i lo128 = _mm_loadu_si128((const __m128i *)lo);
	__m128i hi128 = _mm_loadu_si128((const __m128i *)hi);
	__m256i tlo = _mm256_broadcastsi128_si256(lo128);
	__m256i thi = _mm256_broadcastsi128_si256(hi128);
	__m256i mask = _mm256_set1_epi8(0x0f);
	size_t i = 0;
	for (; i + 32 <= n; i += 32) {
		__m256i x = _mm256_loadu_si256((const __m256i *)(in + i));
		__m256i l = _mm256_and_si256(x, mask);
		__m256i h = _mm256_and_si256(_mm256_srli_epi64(x, 4), mask);
		__m256i p = _mm256_xor_si256(_mm256_shuffle_epi8(tlo, l),
					     _mm256_shuffle_epi8(thi, h));
		if (xor_into)
			p = _mm256_xor_si256(
				p, _mm256_loadu_si256((const __m256i *)(out + i)));
		_mm256_storeu_si256((__m256i *)(out + i), p);
	}
	if (i < n)
		mul_slice_scalar(c, in + i, out + i, n - i, xor_into);
}
#endif

static void mul_slice(uint8_t c, const uint8_t *in, uint8_t *out, size_t n,
		      int xor_into, int variant)
{
#if defined(__x86_64__)
	if (variant == RSO_AVX2 && rso_has_avx2()) {
		mul_slice_avx2(c, in, out, n, xor_into);
		return;
	}
#endif
	(void)variant;
	mul_slice_scalar(c, in, out, n, xor_into);
}

/* [EXT] core.rs code_some_slices: for each input i, for each output r:
 * first input overwrites (mul_slice), later ones accumulate (mul_slice_xor).
 * rows[r] points at k coefficients.  Walks the shard in 32 KiB strips so the
 * strip stays in L1/L2 across the r loop (throughput detail, not bytes). */
static void code_some(int k, int nout, const uint8_t *const *rows,
		      const uint8_t *const *in, uint8_t *const *out, size_t S,
		      int variant)
{
	const size_t STRIP = 32768;
	for (size_t off = 0; off < S; off += STRIP) {
		size_t len = S - off < STRIP ? S - off : STRIP;
		for (int i = 0; i < k; i++)
			for (int r = 0; r < nout; r++)
				mul_slice(rows[r][i], in[i] + off, out[r] + off,
					  len, i != 0, variant);
	}
}

/* ------------------------------------------------------------- operations */
/* [EXT] core.rs ReedSolomon::encode_sep.  Appendix A.3 "encode". */
int rso_encode(int k, int m, size_t S, const uint8_t *const *data,
	       uint8_t *const *parity, int variant)
{
	int rc = check_km(k, m);
	if (rc)
		return rc;
	if (S == 0)
		return RSO_EMPTY_SHARD;
	int n = k + m;
	uint8_t *M = (uint8_t *)malloc((size_t)n * k);
	const uint8_t **rows = (const uint8_t **)malloc(sizeof(*rows) * m);
	if (!M || !rows) {
		free(M);
		free(rows);
		return RSO_SINGULAR;
	}
	rc = rso_build_matrix(k, m, M);
	if (rc == RSO_OK) {
		for (int r = 0; r < m; r++)
			rows[r] = M + (size_t)(k + r) * k;
		code_some(k, m, rows, data, parity, S, variant);
	}
	free(M);
	free(rows);
	return rc;
}

/* [EXT] core.rs ReedSolomon::verify: recompute parity into a buffer, compare. */
int rso_verify(int k, int m, size_t S, const uint8_t *const *shards, int *ok)
{
	int rc = check_km(k, m);
	if (rc)
		return rc;
	if (S == 0)
		return RSO_EMPTY_SHARD;
	uint8_t *buf = (uint8_t *)malloc((size_t)m * S);
	uint8_t **pp = (uint8_t **)malloc(sizeof(*pp) * m);
	if (!buf || !pp) {
		free(buf);
		free(pp);
		return RSO_SINGULAR;
	}
	for (int r = 0; r < m; r++)
		pp[r] = buf + (size_t)r * S;
	rc = rso_encode(k, m, S, shards, pp, RSO_SCALAR);
	if (rc == RSO_OK) {
		*ok = 1;
		for (int r = 0; r < m; r++)
			if (memcmp(pp[r], shards[k + r], S) != 0)
				*ok = 0;
	}
	free(buf);
	free(pp);
	return rc;
}

/* [EXT] core.rs reconstruct_internal.  Appendix A.3 "reconstruct". */
static int reconstruct_v(int k, int m, size_t S, uint8_t *const *shards,
			 const uint8_t *present, int data_only, int variant)
{
	int rc = check_km(k, m);
	if (rc)
		return rc;
	if (S == 0)
		return RSO_EMPTY_SHARD;
	int n = k + m, npresent = 0, ndata_present = 0;
	for (int j = 0; j < n; j++)
		if (present[j]) {
			npresent++;
			if (j < k)
				ndata_present++;
		}
	if (npresent == n)
		return RSO_OK;
	if (npresent < k)
		return RSO_TOO_FEW_PRESENT;

	uint8_t *M = (uint8_t *)malloc((size_t)n * k);
	uint8_t *D = (uint8_t *)malloc((size_t)k * k);
	int *valid = (int *)malloc(sizeof(int) * k);
	const uint8_t **rows = (const uint8_t **)malloc(sizeof(*rows) * n);
	const uint8_t **in = (const uint8_t **)malloc(sizeof(*in) * k);
	uint8_t **out = (uint8_t **)malloc(sizeof(*out) * n);
	rc = RSO_SINGULAR;
	if (!M || !D || !valid || !rows || !in || !out)
		goto done;
	rc = rso_build_matrix(k, m, M);
	if (rc)
		goto done;

	if (ndata_present < k) {
		rc = rso_decode_matrix(k, m, present, valid, D);
		if (rc)
			goto done;
		int nout = 0;
		for (int t = 0; t < k; t++)
			in[t] = shards[valid[t]];
		for (int j = 0; j < k; j++)
			if (!present[j]) {
				rows[nout] = D + (size_t)j * k;
				out[nout++] = shards[j];
			}
		code_some(k, nout, rows, in, out, S, variant);
	}
	if (!data_only) {
		int nout = 0;
		for (int i = 0; i < k; i++)
			in[i] = shards[i];
		for (int j = k; j < n; j++)
			if (!present[j]) {
				rows[nout] = M + (size_t)j * k;
				out[nout++] = shards[j];
			}
		if (nout)
			code_some(k, nout, rows, in, out, S, variant);
	}
	rc = RSO_OK;
done:
	free(M);
	free(D);
	free(valid);
	free(rows);
	free(in);
	free(out);
	return rc;
}

int rso_reconstruct(int k, int m, size_t S, uint8_t *const *shards,
		    const uint8_t *present, int data_only)
{
	return reconstruct_v(k, m, S, shards, present, data_only, RSO_SCALAR);
}

/* ------------------------------------------------ batched (bench baseline) */
int rso_max_threads(void)
{
#ifdef _OPENMP
	return omp_get_max_threads();
#else
	return 1;
#endif
}

int rso_encode_batch(int k, int m, size_t S, size_t nblocks,
		     const uint8_t *data, size_t data_stride, uint8_t *parity,
		     size_t parity_stride, int variant, int threads)
{
	int rc = check_km(k, m);
	if (rc)
		return rc;
	if (S == 0)
		return RSO_EMPTY_SHARD;
	int n = k + m;
	uint8_t *M = (uint8_t *)malloc((size_t)n * k);
	if (!M)
		return RSO_SINGULAR;
	rc = rso_build_matrix(k, m, M);
	if (rc) {
		free(M);
		return rc;
	}
#ifdef _OPENMP
	if (threads <= 0)
		threads = omp_get_max_threads();
#pragma omp parallel for schedule(static) num_threads(threads)
#endif
	for (long b = 0; b < (long)nblocks; b++) {
		const uint8_t *in[256];
		uint8_t *out[256];
		const uint8_t *rows[256];
		for (int i = 0; i < k; i++)
			in[i] = data + (size_t)b * data_stride + (size_t)i * S;
		for (int r = 0; r < m; r++) {
			out[r] = parity + (size_t)b * parity_stride + (size_t)r * S;
			rows[r] = M + (size_t)(k + r) * k;
		}
		code_some(k, m, rows, in, out, S, variant);
	}
	free(M);
	return RSO_OK;
}

int rso_reconstruct_batch(int k, int m, size_t S, size_t nblocks,
			  uint8_t *stripes, size_t stride,
			  const uint8_t *present, int data_only, int threads)
{
	int rc = check_km(k, m);
	if (rc)
		return rc;
	int n = k + m;
	int err = RSO_OK;
#ifdef _OPENMP
	if (threads <= 0)
		threads = omp_get_max_threads();
#pragma omp parallel for schedule(static) num_threads(threads)
#endif
	for (long b = 0; b < (long)nblocks; b++) {
		uint8_t *sh[256];
		for (int j = 0; j < n; j++)
			sh[j] = stripes + (size_t)b * stride + (size_t)j * S;
		int r = reconstruct_v(k, m, S, sh, present, data_only,
				      rso_has_avx2() ? RSO_AVX2 : RSO_SCALAR);
		if (r != RSO_OK)
			err = r;
	}
	return err;
}

static double now_s(void)
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return ts.tv_sec + ts.tv_nsec * 1e-9;
}

static int cmp_double(const void *a, const void *b)
{
	double x = *(const double *)a, y = *(const double *)b;
	return x < y ? -1 : x > y;
}

/* the contiguous share of thread t of T under schedule(static) without a chunk size */
static void range_of(long n, int T, int t, long *lo, long *hi)
{
	const long q = n / T, r = n % T;
	*lo = t * q + (t < r ? t : r);
	*hi = *lo + q + (t < r ? 1 : 0);
}

double rso_bench_encode(int k, int m, size_t S, size_t nblocks, int reps,
			int variant, int threads, uint64_t seed, uint8_t *checksum)
{
	if (check_km(k, m) || S == 0 || nblocks == 0 || reps <= 0)
		return -1.0;
	int n = k + m;
	size_t stride = (size_t)n * S;
	uint8_t *buf = (uint8_t *)malloc(stride * nblocks);
	uint8_t *M = (uint8_t *)malloc((size_t)n * k);
	double *t = (double *)malloc(sizeof(double) * reps);
	long *cursor = (long *)malloc(sizeof(long) * 8 * 1024); /* one cache line per thread */
	if (!buf || !M || !t || !cursor || threads > 1024 || rso_build_matrix(k, m, M)) {
		free(buf);
		free(M);
		free(t);
		free(cursor);
		return -1.0;
	}
#ifdef _OPENMP
	if (threads <= 0)
		threads = omp_get_max_threads();
#else
	threads = 1;
#endif
	/* first touch by the thread that will encode the block (schedule(static) = the contiguous ranges of range_of) */
#pragma omp parallel for schedule(static) num_threads(threads)
	for (long b = 0; b < (long)nblocks; b++) {
		uint64_t *p = (uint64_t *)(buf + (size_t)b * stride);
		uint64_t x = seed + (uint64_t)b * 0xD1B54A32D192ED03ull;
		for (size_t i = 0; i < (size_t)k * S / 8; i++) {
			uint64_t z = (x += 0x9E3779B97F4A7C15ull);
			z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
			z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
			p[i] = z ^ (z >> 31);
		}
		memset(buf + (size_t)b * stride + (size_t)k * S, 0, (size_t)m * S);
	}
	for (int r = -1; r < reps; r++) { /* r == -1: warm-up */
		double t0 = now_s();
		/* Every thread encodes the blocks it first-touched (its static range), then takes what is left of the others'
		 * ranges, nearest first: a team member that loses its CPU for a while -- a vCPU the hypervisor hands to someone
		 * else, any other runnable thread -- no longer holds the whole team at the closing barrier for the length of
		 * its share (profiles/r03_cpu_baseline.txt: 128 threads 22 instead of 350 GiB/s on one box). */
#pragma omp parallel num_threads(threads)
		{
#ifdef _OPENMP
			const int tid = omp_get_thread_num(), T = omp_get_num_threads();
#else
			const int tid = 0, T = 1;
#endif
			long lo, hi;
			range_of((long)nblocks, T, tid, &lo, &hi);
			cursor[(size_t)tid * 8] = lo;
#pragma omp barrier
			for (int v = 0; v < T; v++) {
				const int vic = (tid + v) % T;
				range_of((long)nblocks, T, vic, &lo, &hi);
				for (;;) {
					const long b = __atomic_fetch_add(&cursor[(size_t)vic * 8], 1, __ATOMIC_RELAXED);
					if (b >= hi)
						break;
					const uint8_t *in[256];
					uint8_t *out[256];
					const uint8_t *rows[256];
					uint8_t *base = buf + (size_t)b * stride;
					for (int i = 0; i < k; i++)
						in[i] = base + (size_t)i * S;
					for (int j = 0; j < m; j++) {
						out[j] = base + (size_t)(k + j) * S;
						rows[j] = M + (size_t)(k + j) * k;
					}
					code_some(k, m, rows, in, out, S, variant);
				}
			}
		}
		if (r >= 0)
			t[r] = now_s() - t0;
	}
	uint8_t acc = 0;
	for (size_t b = 0; b < nblocks; b++)
		for (size_t i = 0; i < (size_t)m * S; i += 4099)
			acc ^= buf[b * stride + (size_t)k * S + i];
	if (checksum)
		*checksum = acc;
	qsort(t, reps, sizeof(double), cmp_double);
	double med = t[reps / 2];
	free(buf);
	free(M);
	free(t);
	free(cursor);
	return med;
}
