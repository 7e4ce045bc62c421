/* The multi-device manager under native load: ONE process, gbm_create_multi over N codecs (device d each, or all on
 * device 0 with dry = 1: a one-GPU box), one coalescing queue per device, C callers x N closed-loop callers putting
 * 1 MiB blocks through gbm_batcher_put_block, then as many readers through gbm_batcher_get_block.  Blocks are routed by
 * gec_device_of_hash (hash[4] % N); every byte that comes back is compared.  Prints ONE JSON object.
 * usage: multi_bench [ndev=2] [callers_per_device=48] [puts_per_caller=20] [dry=0] */
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
static int P, NDISTINCT;
static volatile int failed;

static double now_s(void)
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return ts.tv_sec + ts.tv_nsec / 1e9;
}
static void *putter(void *arg)
{
	const int t = (int)(size_t)arg;
	for (int j = 0; j < P; j++) {
		const int i = (t * P + j) % NDISTINCT;
		if (gbm_batcher_put_block(bt, hashes + 32 * i, blocks[i], L, 0, NULL) != GBM_OK) {
			fprintf(stderr, "put failed: %s\n", gbm_last_error());
			failed = 1;
			return NULL;
		}
	}
	return NULL;
}
static void *getter(void *arg)
{
	const int t = (int)(size_t)arg;
	uint8_t *buf = malloc(L);
	for (int j = 0; j < P; j++) {
		const int i = (t * P + j) % NDISTINCT;
		size_t len = 0;
		if (gbm_batcher_get_block(bt, hashes + 32 * i, buf, L, &len) != GBM_OK || len != L || memcmp(buf, blocks[i], L) != 0) {
			fprintf(stderr, "get failed or wrong bytes: %s\n", gbm_last_error());
			failed = 1;
			break;
		}
	}
	free(buf);
	return NULL;
}

int main(int argc, char **argv)
{
	const int ndev = argc > 1 ? atoi(argv[1]) : 2;
	const int cpd = argc > 2 ? atoi(argv[2]) : 48;
	P = argc > 3 ? atoi(argv[3]) : 20;
	const int dry = argc > 4 ? atoi(argv[4]) : 0;
	const int T = ndev * cpd;
	if (ndev < 1 || ndev > 64 || T > 4096)
		return 2;
	gec_codec *codecs[64];
	for (int d = 0; d < ndev; d++)
		if (gec_codec_create(10, 4, GEC_BACKEND_AUTO, dry ? 0 : d, &codecs[d]) != GEC_OK) {
			fprintf(stderr, "codec %d: %s\n", d, gec_last_error());
			return 2;
		}
	gbm_manager *m;
	if (gbm_create_multi((const gec_codec *const *)codecs, ndev, 16, NULL, 0, &m) != GBM_OK || gbm_batcher_create(m, 128, 300, &bt) != GBM_OK) {
		fprintf(stderr, "setup: %s\n", gbm_last_error());
		return 2;
	}
	NDISTINCT = T * P < 2048 ? T * P : 2048;
	blocks = malloc(sizeof(*blocks) * NDISTINCT);
	hashes = malloc(32 * (size_t)NDISTINCT);
	for (int i = 0; i < NDISTINCT; i++) {
		blocks[i] = malloc(L);
		uint64_t x = 0x9E3779B97F4A7C15ull * (i + 1);
		for (size_t o = 0; o < L; o += 8) {
			x ^= x << 13;
			x ^= x >> 7;
			x ^= x << 17;
			memcpy(blocks[i] + o, &x, 8);
		}
		gbm_blake2sum(blocks[i], L, hashes + 32 * i);
	}
	pthread_t *th = malloc(sizeof(pthread_t) * T);
	double put_s = 0, get_s = 0;
	uint64_t p0[64][3], p1[64][3], g0[64][3], g1[64][3];
	for (int rep = 0; rep < 3 && !failed; rep++) {  // (the first pass sizes the pinned pools: the last one counts)
		for (int d = 0; d < ndev; d++)
			gbm_batcher_device_stats(bt, d, p0[d], g0[d]);
		double t0 = now_s();
		for (int t = 0; t < T; t++)
			pthread_create(&th[t], NULL, putter, (void *)(size_t)t);
		for (int t = 0; t < T; t++)
			pthread_join(th[t], NULL);
		put_s = now_s() - t0;
		t0 = now_s();
		for (int t = 0; t < T; t++)
			pthread_create(&th[t], NULL, getter, (void *)(size_t)t);
		for (int t = 0; t < T; t++)
			pthread_join(th[t], NULL);
		get_s = now_s() - t0;
		for (int d = 0; d < ndev; d++)
			gbm_batcher_device_stats(bt, d, p1[d], g1[d]);
	}
	/* routing: every device's queue took exactly the blocks gec_device_of_hash gives it */
	int routed_ok = 1;
	uint64_t want[64] = {0};
	for (int t = 0; t < T; t++)
		for (int j = 0; j < P; j++)
			want[gec_device_of_hash(hashes + 32 * ((t * P + j) % NDISTINCT), ndev)]++;
	const double gib = (double)T * P / 1024.0;
	printf("{\"what\": \"libgarage_block over %d devices%s: gbm_create_multi, one coalescing queue per device, %d closed-loop native callers "
	       "(%d per device) x %d puts of 1 MiB through gbm_batcher_put_block, then as many gets through gbm_batcher_get_block; RS(10,4), "
	       "16 in-memory nodes; payload GiB/s, host memory to host memory\", \"n_devices\": %d, \"callers\": %d, \"puts_per_caller\": %d, ",
	       ndev, dry ? " (DRY RUN: every codec on device 0)" : "", T, cpd, P, ndev, T, P);
	printf("\"put_GiBps\": %.2f, \"get_GiBps\": %.2f, \"per_device\": [", gib / put_s, gib / get_s);
	for (int d = 0; d < ndev; d++) {
		const uint64_t pb = p1[d][1] - p0[d][1], gb = g1[d][1] - g0[d][1];
		if (pb != want[d] || gb != want[d])
			routed_ok = 0;
		uint64_t met[6];
		gbm_device_metrics(m, d, met);
		printf("%s{\"device\": %d, \"blocks_put\": %llu, \"put_GiBps\": %.2f, \"put_batches\": %llu, \"blocks_get\": %llu, \"get_GiBps\": %.2f, "
		       "\"get_batches\": %llu}",
		       d ? ", " : "", gec_codec_device(gbm_device_codec(m, d)), (unsigned long long)pb, pb / 1024.0 / put_s,
		       (unsigned long long)(p1[d][0] - p0[d][0]), (unsigned long long)gb, gb / 1024.0 / get_s, (unsigned long long)(g1[d][0] - g0[d][0]));
	}
	printf("], \"routing_follows_gec_device_of_hash\": %s, \"every_byte_compared\": %s}\n", routed_ok ? "true" : "false", failed ? "false" : "true");
	gbm_batcher_destroy(bt);
	gbm_destroy(m);
	for (int d = 0; d < ndev; d++)
		gec_codec_destroy(codecs[d]);
	return failed || !routed_ok;
}
