/*
 * rs_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the GF(2^8) Reed-Solomon algorithm of the third-party
 * crate `reed-solomon-erasure` (module `galois_8`), which is the arithmetic
 * BASELINE.json names as "the reference reed-solomon-erasure CPU path".
 *
 * PARITY UNPINNED by /root/reference: Garage has no erasure coding
 * (doc/book/design/goals.md:27) and the crate is neither vendored nor pinned
 * in Cargo.lock, so no version number can be quoted.
 * The oracle is instead pinned to the upstream crate's / Backblaze
 * JavaReedSolomon's published known-answer vectors as listed in SURVEY.md
 * Appendix A (tests/test_oracle_kat.py checks every one of them), and to the
 * worked example printed in Backblaze's article on JavaReedSolomon (4 + 2:
 * coding rows 1b 1c 12 14 / 1c 1b 14 12, "ABCD EFGH IJKL MNOP" ->
 * 51 52 53 49 / 55 56 57 25).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * link or call this.  The product library (libgarage_ec.so) never does.
 *
 * Would-be call sites in the reference (where the crate would be invoked):
 *   write: src/block/manager.rs:375-405 (between DataBlock::from_buffer and
 *          RpcHelper::try_write_many_sets)
 *   read:  src/block/manager.rs:292-334 (after gathering >=k shard streams)
 */
#ifndef RS_ORACLE_H
#define RS_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Error codes mirror reed_solomon_erasure::Error [EXT]. */
enum {
	RSO_OK = 0,
	RSO_TOO_FEW_SHARDS = -1,
	RSO_TOO_MANY_SHARDS = -2,
	RSO_TOO_FEW_DATA = -3,
	RSO_TOO_MANY_DATA = -4,
	RSO_TOO_FEW_PARITY = -5,
	RSO_TOO_MANY_PARITY = -6,
	RSO_INCORRECT_SHARD_SIZE = -7,
	RSO_TOO_FEW_PRESENT = -8,
	RSO_EMPTY_SHARD = -9,
	RSO_INVALID_INDEX = -10,
	RSO_SINGULAR = -50
};

enum { RSO_SCALAR = 0, RSO_AVX2 = 1 };

/* --- field (Appendix A.1: poly 0x11D, generator 2) --- */
uint8_t rso_gf_mul(uint8_t a, uint8_t b);
uint8_t rso_gf_div(uint8_t a, uint8_t b);           /* b != 0 */
uint8_t rso_gf_exp(uint8_t a, unsigned n);
const uint8_t *rso_exp_table(void);                 /* 512 entries (doubled) */
const uint8_t *rso_log_table(void);                 /* 256 entries, [0] unused */

/* --- matrices (Appendix A.2), row-major uint8 --- */
/* out[rows*cols] = r^c over GF(2^8) */
void rso_vandermonde(int rows, int cols, uint8_t *out);
/* Gauss-Jordan inverse of an n x n matrix; returns RSO_SINGULAR if singular */
int rso_invert(int n, const uint8_t *in, uint8_t *out);
/* out[(k+m)*k]: systematic encoding matrix; top k rows identity */
int rso_build_matrix(int k, int m, uint8_t *out);
/* valid[k], decode[k*k]: crate's selection = first k present shard indices */
int rso_decode_matrix(int k, int m, const uint8_t *present /*k+m*/,
		      int *valid, uint8_t *decode);

/* --- operations (Appendix A.3) --- */
int rso_has_avx2(void);
/* parity[r][0..S) = XOR_i M[k+r][i] * data[i][0..S) */
int rso_encode(int k, int m, size_t S, const uint8_t *const *data,
	       uint8_t *const *parity, int variant);
/* ok = 1 iff recomputed parity == stored parity */
int rso_verify(int k, int m, size_t S, const uint8_t *const *shards, int *ok);
/* shards[j]==present[j]?valid data:scratch of S bytes to be filled.
 * Missing data shards are rebuilt from the first k present shards, then
 * missing parity shards are re-encoded from the complete data (crate order).
 * data_only != 0 skips missing parity. */
int rso_reconstruct(int k, int m, size_t S, uint8_t *const *shards,
		    const uint8_t *present, int data_only);

/* Batched strided forms used by the bench's cpu_baseline leg: block b's data
 * shard i is at data + b*data_stride + i*S; parity r at parity +
 * b*parity_stride + r*S.  OpenMP over blocks, `threads` <= 0 -> omp default. */
int rso_encode_batch(int k, int m, size_t S, size_t nblocks,
		     const uint8_t *data, size_t data_stride, uint8_t *parity,
		     size_t parity_stride, int variant, int threads);
/* stripes: all k+m shards of block b at stripes + b*stride + j*S */
int rso_reconstruct_batch(int k, int m, size_t S, size_t nblocks,
			  uint8_t *stripes, size_t stride,
			  const uint8_t *present, int data_only, int threads);
int rso_max_threads(void);
/* Self-contained timing loop for bench.py's cpu_baseline leg: allocates
 * nblocks stripes, fills the data shards with a SplitMix64 stream inside the
 * same static OpenMP schedule that encodes them (NUMA first-touch), runs `reps`
 * encodes and returns seconds per rep (median of reps); *checksum gets the XOR
 * of all parity bytes of the last rep so the work cannot be elided. */
double rso_bench_encode(int k, int m, size_t S, size_t nblocks, int reps,
			int variant, int threads, uint64_t seed, uint8_t *checksum);

#ifdef __cplusplus
}
#endif
#endif
