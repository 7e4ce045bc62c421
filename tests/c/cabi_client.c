/* A plain-C client of libgarage_ec.so -- what a non-Python host (the Rust shim's
 * moral equivalent) does: only include/garage_ec.h, plain pointers and sizes.
 *
 *   cabi_client            host-logic checks only (runs without a GPU)
 *   cabi_client gpu        + encode / verify / reconstruct through the host API,
 *                            checked against the RS(3,1) parity == XOR identity and
 *                            an encode -> erase -> reconstruct round trip; blake2sum
 *                            on the device against the RFC 7693 vector; also runs
 *                            4 threads on one shared codec.
 * exit code 0 = all good. */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "garage_ec.h"

#define CHECK(cond)                                                            \
	do {                                                                   \
		if (!(cond)) {                                                 \
			fprintf(stderr, "FAIL %s:%d: %s (last error: %s)\n",   \
				__FILE__, __LINE__, #cond, gec_last_error());  \
			exit(1);                                               \
		}                                                              \
	} while (0)

static void fill(uint8_t *p, size_t n, unsigned seed)
{
	unsigned long long z = seed * 0x9E3779B97F4A7C15ull + 1;
	for (size_t i = 0; i < n; i++) {
		z ^= z << 13;
		z ^= z >> 7;
		z ^= z << 17;
		p[i] = (uint8_t)(z >> 24);
	}
}

struct job {
	const gec_codec *c;
	int id;
	int ok;
};

static void *worker(void *arg)
{
	struct job *j = (struct job *)arg;
	const int k = gec_codec_k(j->c), m = gec_codec_m(j->c), n = k + m;
	const size_t L = 200000 + 1000 * j->id, S = gec_shard_len(k, L);
	enum { NB = 6 };
	uint8_t *blk[NB], *par[NB];
	size_t len[NB];
	for (int b = 0; b < NB; b++) {
		blk[b] = (uint8_t *)calloc(k * S, 1);
		par[b] = (uint8_t *)malloc(m * S);
		fill(blk[b], L, 100 * j->id + b);
		len[b] = L;
	}
	for (int rep = 0; rep < 5; rep++) {
		if (gec_encode_batch(j->c, NB, (const uint8_t *const *)blk, len, S, par) != GEC_OK)
			return NULL;
		/* drop data shard (id % k) and parity shard 0 of every block, rebuild */
		const uint8_t *sh[NB * 64];
		uint8_t *out[NB * 64];
		uint8_t *tmp = (uint8_t *)malloc((size_t)NB * 2 * S);
		int lost = j->id % k;
		for (int b = 0; b < NB; b++)
			for (int s = 0; s < n; s++) {
				sh[b * n + s] = s < k ? blk[b] + s * S : par[b] + (s - k) * S;
				out[b * n + s] = NULL;
				if (s == lost) {
					sh[b * n + s] = NULL;
					out[b * n + s] = tmp + (size_t)(2 * b) * S;
				} else if (s == k) {
					sh[b * n + s] = NULL;
					out[b * n + s] = tmp + (size_t)(2 * b + 1) * S;
				}
			}
		if (gec_reconstruct_batch(j->c, NB, sh, out, S, 0) != GEC_OK)
			return NULL;
		for (int b = 0; b < NB; b++)
			if (memcmp(tmp + (size_t)(2 * b) * S, blk[b] + lost * S, S) ||
			    memcmp(tmp + (size_t)(2 * b + 1) * S, par[b], S))
				return NULL;
		free(tmp);
	}
	/* round-2 entry points from the same threads: pinned buffers (copy kernels instead of staging), encode +
	 * shard checksums, the read path in one trip (checksums of the shards read, rebuilt data shards, the block's
	 * own blake2sum) -- every result cross-checked against the other entry points */
	{
		uint8_t *pblk[NB], *ppar[NB], sums[NB * 64 * 32], sums2[NB * 64 * 32], bsum[NB * 32], bsum2[NB * 32];
		for (int b = 0; b < NB; b++) {
			pblk[b] = (uint8_t *)gec_host_alloc(k * S);
			ppar[b] = (uint8_t *)gec_host_alloc(m * S);
			if (!pblk[b] || !ppar[b] || (gec_device_count() > 0 && !gec_host_is_pinned(pblk[b], k * S)))
				return NULL;
			memcpy(pblk[b], blk[b], k * S);
		}
		for (int rep = 0; rep < 3; rep++) {
			if (gec_encode_hash_batch(j->c, NB, (const uint8_t *const *)pblk, len, S, ppar, sums) != GEC_OK)
				return NULL;
			const uint8_t *sh[NB * 64];
			uint8_t *out[NB * 64];
			const uint8_t *flat[NB * 64];
			size_t flen[NB * 64];
			uint8_t *tmp = (uint8_t *)gec_host_alloc((size_t)NB * S);
			const int lost = (j->id + rep) % k;
			for (int b = 0; b < NB; b++) {
				if (memcmp(ppar[b], par[b], m * S))
					return NULL; /* pinned path == staged path */
				for (int s = 0; s < n; s++) {
					sh[b * n + s] = s < k ? pblk[b] + s * S : ppar[b] + (s - k) * S;
					flat[b * n + s] = sh[b * n + s];
					flen[b * n + s] = S;
					out[b * n + s] = NULL;
				}
				sh[b * n + lost] = NULL;
				out[b * n + lost] = tmp + (size_t)b * S;
			}
			memset(sums2, 0, sizeof sums2);
			if (gec_decode_verify_batch(j->c, NB, sh, S, len, out, sums2, bsum) != GEC_OK)
				return NULL;
			const uint8_t *bp[NB];
			for (int b = 0; b < NB; b++)
				bp[b] = pblk[b];
			if (gec_blake2sum_batch(j->c, NB, bp, len, bsum2) != GEC_OK || memcmp(bsum, bsum2, NB * 32))
				return NULL;
			uint8_t direct[NB * 64 * 32];
			if (gec_shardsum_batch(j->c, (size_t)NB * n, flat, flen, direct) != GEC_OK || memcmp(direct, sums, (size_t)NB * n * 32))
				return NULL;
			for (int b = 0; b < NB; b++) {
				if (memcmp(tmp + (size_t)b * S, pblk[b] + lost * S, S))
					return NULL;
				/* the k shards that were read: every data shard but `lost`, plus parity shard k */
				for (int s = 0; s <= k; s++)
					if (s != lost && memcmp(sums2 + (b * n + s) * 32, sums + (b * n + s) * 32, 32))
						return NULL;
			}
			gec_host_free(tmp);
		}
		for (int b = 0; b < NB; b++) {
			gec_host_free(pblk[b]);
			gec_host_free(ppar[b]);
		}
	}
	j->ok = 1;
	return NULL;
}

int main(int argc, char **argv)
{
	/* mode: none = host logic only; "gpu" = every section on a GEC_BACKEND_HIP codec; "cpu" = the same sections on a
	 * GEC_BACKEND_CPU codec (the library's own host-core data path: runs on a box without a GPU) */
	const int backend = (argc > 1 && strcmp(argv[1], "cpu") == 0) ? GEC_BACKEND_CPU : GEC_BACKEND_HIP;
	/* ---- host logic, no GPU needed ---- */
	CHECK(gec_version() == GEC_VERSION);
	CHECK(gec_shard_len(10, 1 << 20) == 104896 && gec_shard_len(3, 65536) == 21888);
	uint8_t mat[14 * 10];
	CHECK(gec_build_matrix(10, 4, mat) == GEC_OK);
	static const uint8_t row0[10] = {129, 150, 175, 184, 210, 196, 254, 232, 3, 2};
	CHECK(memcmp(mat + 10 * 10, row0, 10) == 0); /* SURVEY Appendix A.4.4 */
	CHECK(gec_build_matrix(0, 4, mat) == GEC_E_TOO_FEW_DATA);
	CHECK(gec_build_matrix(250, 7, mat) == GEC_E_TOO_MANY_SHARDS);
	gec_codec *c = NULL;
	CHECK(gec_codec_create(10, 0, GEC_BACKEND_HIP, 0, &c) == GEC_E_TOO_FEW_PARITY && c == NULL);
	CHECK(gec_codec_create(10, 4, 7, 0, &c) == GEC_E_INVALID_ARG && c == NULL); /* unknown backend */
	if (gec_device_count() == 0) {
		/* no GPU: a HIP codec is refused, AUTO falls back to the host cores */
		CHECK(gec_codec_create(10, 4, GEC_BACKEND_HIP, 0, &c) == GEC_E_DEVICE && c == NULL);
		CHECK(strstr(gec_last_error(), "GEC_BACKEND_CPU") != NULL);
		CHECK(gec_codec_create(10, 4, GEC_BACKEND_AUTO, 0, &c) == GEC_OK && gec_codec_backend(c) == GEC_BACKEND_CPU);
		CHECK(gec_codec_device(c) == -1);
		uint8_t dummy[64];
		CHECK(gec_encode_batch_dev(c, 1, dummy, 640, 64, dummy, 256, NULL) == GEC_E_INVALID_ARG ||
		      gec_encode_batch_dev(c, 1, dummy, 640, 64, dummy, 256, NULL) == GEC_E_DEVICE); /* no device entry points */
		gec_codec_destroy(c);
		c = NULL;
		if (backend == GEC_BACKEND_HIP) {
			printf("cabi_client: host-logic checks OK (no GPU: HIP codec refused, AUTO -> CPU)\n");
			return argc > 1 ? 2 : 0;
		}
	}
	if (argc < 2)
		return 0;
	printf("cabi_client: backend %s (cpu kernel: %s)\n", backend == GEC_BACKEND_CPU ? "cpu" : "hip", gec_cpu_isa());

	/* ---- RS(3,1): parity must be the XOR of the three data shards ---- */
	CHECK(gec_codec_create(3, 1, backend, 0, &c) == GEC_OK);
	const size_t L = 65536, S = gec_shard_len(3, L);
	uint8_t *blk = (uint8_t *)calloc(3 * S, 1), *par = (uint8_t *)malloc(S);
	fill(blk, L, 7);
	const uint8_t *blocks[1] = {blk};
	uint8_t *parity[1] = {par};
	CHECK(gec_encode_batch(c, 1, blocks, &L, S, parity) == GEC_OK);
	for (size_t i = 0; i < S; i++)
		CHECK(par[i] == (uint8_t)(blk[i] ^ blk[S + i] ^ blk[2 * S + i]));
	const uint8_t *sh[4] = {blk, blk + S, blk + 2 * S, par};
	uint8_t ok = 0;
	CHECK(gec_verify_batch(c, 1, sh, S, &ok) == GEC_OK && ok == 1);
	par[5] ^= 1;
	CHECK(gec_verify_batch(c, 1, sh, S, &ok) == GEC_OK && ok == 0);
	par[5] ^= 1;
	const uint8_t *sh2[4] = {blk, NULL, NULL, par};
	uint8_t *outp[4] = {NULL, NULL, NULL, NULL};
	CHECK(gec_reconstruct_batch(c, 1, sh2, outp, S, 0) == GEC_E_TOO_FEW_PRESENT);
	{
		/* scrub in one trip and rebuild in one trip, from ordinary and from pinned memory: the checksums they
		 * return are gec_shardsum_batch's */
		uint8_t want[4 * 32], sums[4 * 32], ins[4 * 32], outs[4 * 32];
		const size_t lens4[4] = {S, S, S, S};
		CHECK(gec_shardsum_batch(c, 4, sh, lens4, want) == GEC_OK);
		for (int pinned = 0; pinned < 2; pinned++) {
			uint8_t *arena = pinned ? (uint8_t *)gec_host_alloc(5 * S) : (uint8_t *)malloc(5 * S);
			CHECK(arena != NULL);
			memcpy(arena, blk, 3 * S);
			memcpy(arena + 3 * S, par, S);
			const uint8_t *shp[4] = {arena, arena + S, arena + 2 * S, arena + 3 * S};
			memset(sums, 0, sizeof sums);
			CHECK(gec_verify_hash_batch(c, 1, shp, S, &ok, sums) == GEC_OK && ok == 1 && memcmp(sums, want, sizeof want) == 0);
			arena[S + 9] ^= 4;
			CHECK(gec_verify_hash_batch(c, 1, shp, S, &ok, sums) == GEC_OK && ok == 0);
			CHECK(memcmp(sums, want, 32) == 0 && memcmp(sums + 32, want + 32, 32) != 0);
			/* shard 1 lost: rebuilt into the spare slot, checksums of shards 0, 2, 3 read and of shard 1 written */
			const uint8_t *sh3[4] = {arena, NULL, arena + 2 * S, arena + 3 * S};
			uint8_t *out3[4] = {NULL, arena + 4 * S, NULL, NULL};
			memset(ins, 0, sizeof ins);
			memset(outs, 0, sizeof outs);
			CHECK(gec_reconstruct_hash_batch(c, 1, sh3, out3, S, 0, ins, outs) == GEC_OK);
			CHECK(memcmp(arena + 4 * S, blk + S, S) == 0);
			CHECK(memcmp(ins, want, 32) == 0 && memcmp(ins + 64, want + 64, 64) == 0 && memcmp(outs + 32, want + 32, 32) == 0);
			for (int i = 0; i < 32; i++)
				CHECK(ins[32 + i] == 0 && outs[i] == 0);
			if (pinned)
				gec_host_free(arena);
			else
				free(arena);
		}
	}
	gec_codec_destroy(c);

	/* ---- blake2sum on the device: RFC 7693 "abc" (blake2b-512, first 32 bytes = Garage's blake2sum),
	 *      the empty message, and encode + shard checksums in one call ---- */
	CHECK(gec_codec_create(10, 4, backend, 0, &c) == GEC_OK);
	{
		static const uint8_t abc_want[32] = {0xba, 0x80, 0xa5, 0x3f, 0x98, 0x1c, 0x4d, 0x0d, 0x6a, 0x27, 0x97,
						     0xb6, 0x9f, 0x12, 0xf6, 0xe9, 0x4c, 0x21, 0x2f, 0x14, 0x68, 0x5a,
						     0xc4, 0xb7, 0x4b, 0x12, 0xbb, 0x6f, 0xdb, 0xff, 0xa2, 0xd1};
		static const uint8_t empty_want[8] = {0x78, 0x6a, 0x02, 0xf7, 0x42, 0x01, 0x59, 0x03};
		const uint8_t *msgs[2] = {(const uint8_t *)"abc", (const uint8_t *)""};
		const size_t lens[2] = {3, 0};
		uint8_t sums[64];
		CHECK(gec_blake2sum_batch(c, 2, msgs, lens, sums) == GEC_OK);
		CHECK(memcmp(sums, abc_want, 32) == 0 && memcmp(sums + 32, empty_want, 8) == 0);

		const size_t L2 = 300000, S2 = gec_shard_len(10, L2);
		uint8_t *b2 = (uint8_t *)calloc(10 * S2, 1), *p2 = (uint8_t *)malloc(4 * S2), *p3 = (uint8_t *)malloc(4 * S2);
		fill(b2, L2, 99);
		const uint8_t *bl[1] = {b2};
		uint8_t *pa[1] = {p2}, *pb[1] = {p3};
		uint8_t shard_sums[14 * 32], direct[14 * 32];
		CHECK(gec_encode_hash_batch(c, 1, bl, &L2, S2, pa, shard_sums) == GEC_OK);
		CHECK(gec_encode_batch(c, 1, bl, &L2, S2, pb) == GEC_OK && memcmp(p2, p3, 4 * S2) == 0);
		const uint8_t *sp[14];
		size_t sl[14];
		for (int j = 0; j < 14; j++) {
			sp[j] = j < 10 ? b2 + j * S2 : p2 + (j - 10) * S2;
			sl[j] = S2;
		}
		CHECK(gec_shardsum_batch(c, 14, sp, sl, direct) == GEC_OK);   /* shard checksums are BLAKE2b tree mode */
		CHECK(memcmp(shard_sums, direct, sizeof direct) == 0);
		free(b2);
		free(p2);
		free(p3);
	}
	gec_codec_destroy(c);

	/* ---- 4 threads sharing one RS(10,4) codec (Send + Sync on the Rust side) ---- */
	CHECK(gec_codec_create(10, 4, backend, 0, &c) == GEC_OK);
	pthread_t th[4];
	struct job jobs[4];
	for (int i = 0; i < 4; i++) {
		jobs[i].c = c;
		jobs[i].id = i;
		jobs[i].ok = 0;
		pthread_create(&th[i], NULL, worker, &jobs[i]);
	}
	for (int i = 0; i < 4; i++) {
		pthread_join(th[i], NULL);
		CHECK(jobs[i].ok);
	}
	gec_codec_destroy(c);
	printf("cabi_client: %s checks OK\n", backend == GEC_BACKEND_CPU ? "CPU" : "GPU");
	return 0;
}
