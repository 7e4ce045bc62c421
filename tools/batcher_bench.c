/* Native load on the coalescing batcher (no Python between the callers and the library): T caller threads, each
 * putting P blocks of 1 MiB one after the other through gbm_batcher_put_block -- 16 PutObject requests x
 * PUT_BLOCKS_MAX_PARALLEL = 3 callers (src/api/s3/put.rs:42) is T = 48.  RS(10,4), 16 in-memory nodes.
 * usage: batcher_bench [threads] [puts per thread] [max_blocks] [max_wait_us]     (needs a GPU) */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "garage_block.h"
#include "garage_ec.h"

#define L (1u << 20)

static gbm_batcher *bt;
static uint8_t **blocks;
static uint8_t *hashes;
static int P;
static double *lat_ms;

static double now_ms(void)
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return ts.tv_sec * 1e3 + ts.tv_nsec / 1e6;
}

static void *caller(void *arg)
{
	const int t = (int)(size_t)arg;
	for (int j = 0; j < P; j++) {
		const int i = t * P + j;
		const double t0 = now_ms();
		if (gbm_batcher_put_block(bt, hashes + 32 * i, blocks[i], L, 0, NULL) != GBM_OK) {
			fprintf(stderr, "put failed: %s\n", gbm_last_error());
			exit(1);
		}
		lat_ms[i] = now_ms() - t0;
	}
	return NULL;
}

static int cmp(const void *a, const void *b) { return *(const double *)a < *(const double *)b ? -1 : 1; }

int main(int argc, char **argv)
{
	const int T = argc > 1 ? atoi(argv[1]) : 48;
	P = argc > 2 ? atoi(argv[2]) : 20;
	const size_t max_blocks = argc > 3 ? (size_t)atoi(argv[3]) : 128;
	const unsigned wait_us = argc > 4 ? (unsigned)atoi(argv[4]) : 300;
	const int N = T * P;
	gec_codec *c;
	gbm_manager *m;
	if (gec_codec_create(10, 4, GEC_BACKEND_AUTO, 0, &c) != GEC_OK || gbm_create(c, 16, NULL, 0, &m) != GBM_OK) {
		fprintf(stderr, "setup failed: %s / %s\n", gec_last_error(), gbm_last_error());
		return 2;
	}
	blocks = malloc(sizeof(*blocks) * N);
	hashes = malloc(32 * (size_t)N);
	lat_ms = malloc(sizeof(double) * N);
	unsigned long long x = 88172645463325252ull;
	for (int i = 0; i < N; i++) {
		blocks[i] = malloc(L);
		for (size_t o = 0; o < L; o += 8) {
			x ^= x << 13, x ^= x >> 7, x ^= x << 17;
			memcpy(blocks[i] + o, &x, 8);
		}
		gbm_blake2sum(blocks[i], L, hashes + 32 * i);
	}
	/* warm: sizes the pinned buffer pool and the device staging */
	const uint8_t *wd[64];
	size_t wl[64];
	for (int r = 0; r < 3; r++) {
		for (int i = 0; i < 64 && i < N; i++)
			wd[i] = blocks[i], wl[i] = L;
		gbm_rpc_put_blocks(m, N < 64 ? N : 64, hashes, wd, wl, NULL, NULL);
	}
	for (int rep = 0; rep < 3; rep++) {
		if (gbm_batcher_create(m, max_blocks, wait_us, &bt) != GBM_OK)
			return 2;
		pthread_t *th = malloc(sizeof(*th) * T);
		const double t0 = now_ms();
		for (int t = 0; t < T; t++)
			pthread_create(&th[t], NULL, caller, (void *)(size_t)t);
		for (int t = 0; t < T; t++)
			pthread_join(th[t], NULL);
		const double dt = now_ms() - t0;
		uint64_t st[3];
		gbm_batcher_stats(bt, st);
		qsort(lat_ms, N, sizeof(double), cmp);
		printf("%d callers x %d puts of 1 MiB (batch <= %zu, linger %u us): %.1f ms = %.2f GiB/s; %llu batches, largest %llu; "
		       "put latency median %.2f ms, p99 %.2f ms\n",
		       T, P, max_blocks, wait_us, dt, N / 1024.0 / (dt / 1e3), (unsigned long long)st[0], (unsigned long long)st[2],
		       lat_ms[N / 2], lat_ms[(int)(N * 0.99)]);
		gbm_batcher_destroy(bt);
		free(th);
	}
	gbm_destroy(m);
	gec_codec_destroy(c);
	return 0;
}
