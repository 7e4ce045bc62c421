/* Small trips through libgarage_block on one device, natively (no Python between the caller and the library):
 *   - one 1 MiB put / a PutObject's three through the batcher (closed loop);
 *   - one 1 MiB get (gbm_rpc_get_block), healthy and with a data shard to rebuild, in the three end-to-end hash modes;
 *   - streaming get of a 1 MiB and a 4 MiB block: time to the first chunk and to the last one;
 *   - R concurrent readers through gbm_batcher_get_block in the three modes (GiB/s, latency);
 * RS(10,4), 16 in-memory nodes.  usage: small_trip_bench [readers=48] [rounds=20] */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "garage_block.h"
#include "garage_ec.h"

#define L1 (1u << 20)
static double now_ms(void)
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return ts.tv_sec * 1e3 + ts.tv_nsec / 1e6;
}
static int cmpd(const void *a, const void *b) { return *(const double *)a < *(const double *)b ? -1 : 1; }
static double median(double *v, int n)
{
	qsort(v, n, sizeof *v, cmpd);
	return v[n / 2];
}
static void fill(uint8_t *p, size_t n, uint64_t seed)
{
	uint64_t x = seed * 0x9E3779B97F4A7C15ull + 12345;
	for (size_t i = 0; i + 8 <= n; i += 8) {
		x ^= x << 13;
		x ^= x >> 7;
		x ^= x << 17;
		memcpy(p + i, &x, 8);
	}
}
#define CHECK(c)                                                                            \
	do {                                                                                \
		if (!(c)) {                                                                 \
			fprintf(stderr, "FAIL %s:%d %s (%s)\n", __FILE__, __LINE__, #c, gbm_last_error()); \
			exit(1);                                                            \
		}                                                                           \
	} while (0)

static gbm_manager *mg;
static gbm_batcher *bt;
static const char *mode_name[3] = {"off", "rebuilt-only", "always"};
static const int mode_val[3] = {GBM_VERIFY_OFF, GBM_VERIFY_REBUILT, GBM_VERIFY_ALWAYS};

struct sink {
	double t0, first, last;
	size_t bytes;
};
static int sink_fn(void *ctx, const uint8_t *chunk, size_t len)
{
	struct sink *s = ctx;
	(void)chunk;
	const double t = now_ms();
	if (s->bytes == 0)
		s->first = t - s->t0;
	s->last = t - s->t0;
	s->bytes += len;
	return 0;
}

/* readers */
static int R, ROUNDS, NOBJ;
static uint8_t *rhashes;
static double *rlat;
static void *reader(void *arg)
{
	const int t = (int)(size_t)arg;
	uint8_t *buf = malloc(L1);
	for (int j = 0; j < ROUNDS; j++) {
		const int i = (t * ROUNDS + j) % NOBJ;
		size_t len = 0;
		const double t0 = now_ms();
		CHECK(gbm_batcher_get_block(bt, rhashes + 32 * i, buf, L1, &len) == GBM_OK && len == L1);
		rlat[t * ROUNDS + j] = now_ms() - t0;
	}
	free(buf);
	return NULL;
}

int main(int argc, char **argv)
{
	R = argc > 1 ? atoi(argv[1]) : 48;
	ROUNDS = argc > 2 ? atoi(argv[2]) : 20;
	gec_codec *c;
	CHECK(gec_codec_create(10, 4, GEC_BACKEND_AUTO, 0, &c) == GEC_OK);
	CHECK(gbm_create(c, 16, NULL, 0, &mg) == GBM_OK);
	CHECK(gbm_batcher_create(mg, 128, 300, &bt) == GBM_OK);
	printf("small_trip_bench: backend %s, RS(10,4), 16 memory nodes\n", gec_codec_backend(c) == GEC_BACKEND_CPU ? "cpu" : "hip");

	/* ---- puts through the batcher: 1 caller, then a PutObject's 3 */
	enum { NP = 60 };
	static uint8_t *pb[NP];
	static uint8_t ph[NP][32];
	for (int i = 0; i < NP; i++) {
		pb[i] = malloc(L1);
		fill(pb[i], L1, 100 + i);
		gbm_blake2sum(pb[i], L1, ph[i]);
	}
	double lat[NP];
	for (int rep = 0; rep < 3; rep++) {  // (the first pass also grows the pinned-buffer pool: ~0.3 ms per new buffer)
		for (int i = 0; i < NP; i++) {
			const double t0 = now_ms();
			CHECK(gbm_batcher_put_block(bt, ph[i], pb[i], L1, 0, NULL) == GBM_OK);
			lat[i] = now_ms() - t0;
		}
		const double first = lat[0];
		printf("put, 1 caller through the batcher, pass %d: median %.3f ms (first %.3f)\n", rep, median(lat + 5, NP - 5), first);
	}
	{
		double l3[NP / 3];
		for (int i = 0; i + 3 <= NP; i += 3) {
			gbm_put_ticket *tk[3];
			const double t0 = now_ms();
			for (int j = 0; j < 3; j++)
				CHECK(gbm_batcher_submit(bt, ph[i + j], pb[i + j], L1, 0, NULL, &tk[j]) == GBM_OK);
			for (int j = 0; j < 3; j++)
				CHECK(gbm_batcher_wait(tk[j]) == GBM_OK);
			l3[i / 3] = now_ms() - t0;
		}
		printf("put, a PutObject's three in flight:       median %.3f ms for the three\n", median(l3, NP / 3));
	}

	if (argc > 3 && strcmp(argv[3], "puts") == 0)  /* an A/B of the put trip: the first section only */
		return 0;

	/* ---- single gets */
	uint8_t *out = malloc(4u << 20);
	int who[14];
	CHECK(gbm_storage_nodes_of(mg, ph[7], who) == GBM_OK);
	for (int deg = 0; deg < 2; deg++) {
		if (deg)
			CHECK(gbm_node_delete_shard(mg, who[2], ph[7], 2) == GBM_OK);
		for (int mi = 0; mi < 3; mi++) {
			CHECK(gbm_set_verify_block_hash(mg, mode_val[mi]) == GBM_OK);
			double l[40];
			for (int i = 0; i < 40; i++) {
				size_t len = 0;
				const double t0 = now_ms();
				CHECK(gbm_rpc_get_block(mg, ph[7], NULL, out, L1, &len) == GBM_OK && len == L1);
				l[i] = now_ms() - t0;
			}
			CHECK(memcmp(out, pb[7], L1) == 0);
			printf("get, one 1 MiB block, %-8s mode %-12s: median %.3f ms\n", deg ? "degraded" : "healthy", mode_name[mi], median(l + 5, 35));
		}
	}
	/* ---- streaming gets: first chunk vs last chunk */
	{
		uint8_t *big = malloc(4u << 20);
		uint8_t hb[32];
		fill(big, 4u << 20, 999);
		gbm_blake2sum(big, 4u << 20, hb);
		CHECK(gbm_rpc_put_block(mg, hb, big, 4u << 20, 0, NULL) == GBM_OK);
		const uint8_t *hs[2] = {ph[9], hb};
		const size_t sz[2] = {L1, 4u << 20};
		for (int w = 0; w < 2; w++)
			for (int mi = 0; mi < 3; mi++) {
				CHECK(gbm_set_verify_block_hash(mg, mode_val[mi]) == GBM_OK);
				double f[30], l[30], e[30];
				for (int i = 0; i < 30; i++) {
					struct sink s = {now_ms(), 0, 0, 0};
					CHECK(gbm_rpc_get_block_streaming(mg, hs[w], NULL, 65536, sink_fn, &s) == GBM_OK && s.bytes == sz[w]);
					f[i] = s.first;
					l[i] = s.last;
					e[i] = now_ms() - s.t0;
				}
				printf("streaming get, %zu MiB block, mode %-12s: first chunk %.3f ms, last chunk %.3f ms, call returns %.3f ms (medians)\n",
				       sz[w] >> 20, mode_name[mi], median(f + 5, 25), median(l + 5, 25), median(e + 5, 25));
			}
		/* ---- ranged gets (body_from_blocks_range): only the data shards the range touches are read */
		CHECK(gbm_set_verify_block_hash(mg, GBM_VERIFY_OFF) == GBM_OK);
		const size_t rb[3][2] = {{300000, 364000}, {100000, 600000}, {0, 4u << 20}};
		for (int q = 0; q < 3; q++) {
			double e[30];
			uint64_t m0[6], m1[6];
			CHECK(gbm_metrics(mg, m0) == GBM_OK);
			for (int i = 0; i < 30; i++) {
				struct sink s = {now_ms(), 0, 0, 0};
				CHECK(gbm_rpc_get_block_range_streaming(mg, hb, NULL, 4u << 20, rb[q][0], rb[q][1], 65536, sink_fn, &s) == GBM_OK &&
				      s.bytes == rb[q][1] - rb[q][0]);
				e[i] = now_ms() - s.t0;
			}
			CHECK(gbm_metrics(mg, m1) == GBM_OK);
			printf("ranged get, %7zu bytes of a 4 MiB block: median %.3f ms, %.0f KiB of shards read per call\n", rb[q][1] - rb[q][0], median(e + 5, 25),
			       (double)(m1[1] - m0[1]) / 30 / 1024);
		}
		free(big);
	}
	/* ---- R readers through the batcher's read side */
	NOBJ = 512;
	rhashes = malloc(32 * (size_t)NOBJ);
	{
		uint8_t **blk = malloc(sizeof(*blk) * NOBJ);
		const uint8_t **ptr = malloc(sizeof(*ptr) * NOBJ);
		size_t *len = malloc(sizeof(*len) * NOBJ);
		for (int i = 0; i < NOBJ; i++) {
			blk[i] = malloc(L1);
			fill(blk[i], L1, 5000 + i);
			gbm_blake2sum(blk[i], L1, rhashes + 32 * i);
			ptr[i] = blk[i];
			len[i] = L1;
		}
		const double t0 = now_ms();
		CHECK(gbm_rpc_put_blocks(mg, NOBJ, rhashes, ptr, len, NULL, NULL) == GBM_OK);
		printf("bulk put of %d x 1 MiB: %.2f GiB/s\n", NOBJ, NOBJ / 1024.0 / ((now_ms() - t0) / 1e3));
		for (int i = 0; i < NOBJ; i++)
			free(blk[i]);
	}
	rlat = malloc(sizeof(double) * R * ROUNDS);
	for (int mi = 0; mi < 3; mi++) {
		CHECK(gbm_set_verify_block_hash(mg, mode_val[mi]) == GBM_OK);
		uint64_t s0[3], s1[3];
		gbm_batcher_get_stats(bt, s0);
		pthread_t th[512];
		const double t0 = now_ms();
		for (int t = 0; t < R; t++)
			pthread_create(&th[t], NULL, reader, (void *)(size_t)t);
		for (int t = 0; t < R; t++)
			pthread_join(th[t], NULL);
		const double secs = (now_ms() - t0) / 1e3;
		gbm_batcher_get_stats(bt, s1);
		qsort(rlat, R * ROUNDS, sizeof(double), cmpd);
		printf("%d readers x %d gets through the batcher, mode %-12s: %.2f GiB/s, median %.2f ms, p99 %.2f ms, %llu batches (largest %llu)\n", R,
		       ROUNDS, mode_name[mi], R * ROUNDS / 1024.0 / secs, rlat[R * ROUNDS / 2], rlat[(int)(R * ROUNDS * 0.99)],
		       (unsigned long long)(s1[0] - s0[0]), (unsigned long long)s1[2]);
	}
	/* ---- bulk get, healthy and with 4 of 16 nodes down, mode off */
	{
		uint8_t **outs = malloc(sizeof(*outs) * NOBJ);
		size_t *cap = malloc(sizeof(*cap) * NOBJ), *len = malloc(sizeof(*len) * NOBJ);
		int *rcs = malloc(sizeof(int) * NOBJ);
		uint8_t *arena = gec_host_alloc((size_t)NOBJ * L1);
		for (int i = 0; i < NOBJ; i++) {
			outs[i] = arena + (size_t)i * L1;
			cap[i] = L1;
		}
		for (int deg = 0; deg < 2; deg++) {
			if (deg)
				for (int nd = 0; nd < 4; nd++)
					gbm_node_set_down(mg, 3 + 4 * nd, 1);
			for (int mi = 0; mi < 3; mi++) {
				CHECK(gbm_set_verify_block_hash(mg, mode_val[mi]) == GBM_OK);
				double best = 1e9;
				for (int rep = 0; rep < 4; rep++) {
					const double t0 = now_ms();
					CHECK(gbm_rpc_get_blocks(mg, NOBJ, rhashes, NULL, outs, cap, len, rcs) == GBM_OK);
					const double ms = now_ms() - t0;
					if (ms < best)
						best = ms;
					for (int i = 0; i < NOBJ; i++)
						CHECK(rcs[i] == GBM_OK);
				}
				printf("bulk get of %d x 1 MiB, %s, mode %-12s: %.2f ms = %.2f GiB/s\n", NOBJ, deg ? "4 of 16 nodes down" : "healthy", mode_name[mi],
				       best, NOBJ / 1024.0 / (best / 1e3));
			}
		}
		gec_host_free(arena);
	}
	gbm_batcher_destroy(bt);
	gbm_destroy(mg);
	gec_codec_destroy(c);
	return 0;
}
