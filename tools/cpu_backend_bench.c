/* cpu_backend_bench -- the library's own CPU backend (GEC_BACKEND_CPU) through the C ABI: RS(10,4) encode of 1 MiB blocks
 * from one contiguous caller buffer, rate and process CPU time / wall time (how many threads really worked).  The thread
 * count is GEC_CPU_THREADS (read once per process): run it once per count.
 * usage: cpu_backend_bench [nblocks=512] [first_touch_parallel=0] [hash=0]
 * hash = 1: gec_encode_hash_batch (the parity AND the 14 shard checksums of every stripe) instead of gec_encode_batch */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/resource.h>
#include <time.h>

#include "garage_ec.h"

static double now(void)
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return ts.tv_sec + ts.tv_nsec * 1e-9;
}
static double cpu(void)
{
	struct rusage r;
	getrusage(RUSAGE_SELF, &r);
	return r.ru_utime.tv_sec + r.ru_utime.tv_usec * 1e-6 + r.ru_stime.tv_sec + r.ru_stime.tv_usec * 1e-6;
}

int main(int argc, char **argv)
{
	const int nb = argc > 1 ? atoi(argv[1]) : 512;
	const int hash = argc > 3 ? atoi(argv[3]) : 0;
	const size_t L = 1 << 20, S = gec_shard_len(10, L);
	gec_codec *c;
	if (gec_codec_create(10, 4, GEC_BACKEND_CPU, 0, &c)) {
		puts(gec_last_error());
		return 1;
	}
	uint8_t *big = malloc((size_t)nb * 10 * S), *out = malloc((size_t)nb * 4 * S);
	for (size_t i = 0; i < (size_t)nb * 10 * S; i += 8) {
		unsigned long long x = i * 0x9E3779B97F4A7C15ull;
		memcpy(big + i, &x, 8);
	}
	memset(out, 1, (size_t)nb * 4 * S);
	const uint8_t **bp = malloc(sizeof(void *) * nb);
	uint8_t **pp = malloc(sizeof(void *) * nb);
	size_t *len = malloc(sizeof(size_t) * nb);
	for (int b = 0; b < nb; b++) {
		bp[b] = big + (size_t)b * 10 * S;
		pp[b] = out + (size_t)b * 4 * S;
		len[b] = L;
	}
	uint8_t *sums = malloc((size_t)nb * 14 * 32);
	double best = 1e9, ratio = 0;
	for (int r = 0; r < 8; r++) {
		const double t0 = now(), c0 = cpu();
		if ((hash ? gec_encode_hash_batch(c, nb, bp, len, S, pp, sums) : gec_encode_batch(c, nb, bp, len, S, pp)) != GEC_OK)
			return 2;
		const double dt = now() - t0, dc = cpu() - c0;
		if (getenv("VERBOSE")) printf("  rep %d: %.2f GiB/s cpu/wall %.2f\n", r, nb * (double)L / dt / (1 << 30), dc / dt);
		if (dt < best) {
			best = dt;
			ratio = dc / dt;
		}
	}
	const char *thr = getenv("GEC_CPU_THREADS");
	printf("GEC_CPU_THREADS=%-4s %-12s %4d blocks%s: %7.2f GiB/s  (cpu/wall %.1f)\n", thr ? thr : "dflt", gec_cpu_isa(), nb, hash ? " + 14 checksums each" : "",
	       nb * (double)L / best / (1 << 30), ratio);
	gec_codec_destroy(c);
	return 0;
}
