// dispatch_probe -- does a kernel with more workgroups than its CU partition can hold delay kernels of OTHER streams
// that are confined to other CUs?
//
// Background (profiles/r03_qos.txt): the background class keeps its kernels on CUs of its own
// (hipExtStreamCreateWithCUMask), yet in some process runs a foreground checksum kernel took as long as the background
// link kernel that happened to run beside it (~630 us instead of 140 us), and in other runs of the same binary it did not.
// The link kernel was launched as ~800 workgroups onto 8 CUs: for its whole duration the dispatcher holds workgroups
// that do not fit yet.  This probe measures what that costs the other streams:
//   stream A: CU mask [32,40), runs `long` = a kernel of total duration ~1 ms, either as MANY short workgroups
//             (40 waves of workgroups per CU slot) or as a RESIDENT grid (one workgroup per slot, each spinning 1 ms)
//   streams B0..B7: CU mask [64,128) each; while A runs, a tiny kernel is launched on Bi and its completion timed
// usage: dispatch_probe [reps=15]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CHECK(x)                                                                                     \
	do {                                                                                         \
		hipError_t e_ = (x);                                                                 \
		if (e_ != hipSuccess) {                                                              \
			fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
			exit(1);                                                                     \
		}                                                                                    \
	} while (0)

__global__ __launch_bounds__(256) void spin(unsigned long long ticks, unsigned *sink)
{
	const unsigned long long t0 = wall_clock64();  // 100 MHz
	while (wall_clock64() - t0 < ticks)
		__builtin_amdgcn_s_sleep(8);
	if (sink && threadIdx.x == 0 && blockIdx.x == 0xffffffffu)
		*sink = 1;
}

static hipStream_t masked(int num_cu, int lo, int hi)
{
	const int words = (num_cu + 31) / 32;
	std::vector<uint32_t> m(words, 0);
	for (int i = lo; i < hi; ++i)
		m[i / 32] |= 1u << (i % 32);
	hipStream_t s;
	CHECK(hipExtStreamCreateWithCUMask(&s, words, m.data()));
	return s;
}

using Clock = std::chrono::steady_clock;
static double us_since(Clock::time_point t) { return std::chrono::duration<double, std::micro>(Clock::now() - t).count(); }

int main(int argc, char **argv)
{
	const int reps = argc > 1 ? atoi(argv[1]) : 15;
	hipDeviceProp_t p;
	CHECK(hipGetDeviceProperties(&p, 0));
	const int ncu = p.multiProcessorCount;
	int occ = 0;
	CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, spin, 256, 0));
	const int slots = 8 * occ;  // workgroups resident on A's 8 CUs
	printf("device: %d CUs, spin kernel: %d workgroups of 256 resident per CU -> %d slots on A's 8 CUs\n", ncu, occ, slots);
	hipStream_t A = masked(ncu, 32, 40);
	const int NB = 8;
	hipStream_t B[NB];
	for (int i = 0; i < NB; ++i)
		B[i] = masked(ncu, 64, 128);
	hipStream_t plain;
	CHECK(hipStreamCreateWithFlags(&plain, hipStreamNonBlocking));
	// warm
	hipLaunchKernelGGL(spin, dim3(slots), dim3(256), 0, A, 100ull, nullptr);
	for (int i = 0; i < NB; ++i)
		hipLaunchKernelGGL(spin, dim3(4), dim3(256), 0, B[i], 100ull, nullptr);
	hipLaunchKernelGGL(spin, dim3(4), dim3(256), 0, plain, 100ull, nullptr);
	CHECK(hipDeviceSynchronize());
	const char *names[3] = {"A idle", "A = 40 x slots short workgroups (25 us each)", "A = resident grid (slots workgroups, 1 ms each)"};
	for (int mode = 0; mode < 3; ++mode) {
		printf("%s\n", names[mode]);
		for (int i = 0; i <= NB; ++i) {
			hipStream_t b = i < NB ? B[i] : plain;
			std::vector<double> lat, adur;
			for (int r = 0; r < reps; ++r) {
				const auto ta = Clock::now();
				if (mode == 1)
					hipLaunchKernelGGL(spin, dim3(40 * slots), dim3(256), 0, A, 2500ull, nullptr);
				else if (mode == 2)
					hipLaunchKernelGGL(spin, dim3(slots), dim3(256), 0, A, 100000ull, nullptr);
				if (mode)  // let A get going
					while (us_since(ta) < 150) {
					}
				const auto tb = Clock::now();
				hipLaunchKernelGGL(spin, dim3(4), dim3(256), 0, b, 500ull, nullptr);  // 5 us of work
				CHECK(hipStreamSynchronize(b));
				lat.push_back(us_since(tb));
				CHECK(hipStreamSynchronize(A));
				adur.push_back(us_since(ta));
			}
			std::sort(lat.begin(), lat.end());
			std::sort(adur.begin(), adur.end());
			printf("  %-8s tiny kernel done after median %8.1f us  (min %7.1f, max %8.1f);  A's launch-to-done median %8.1f us\n",
			       i < NB ? (std::string("B") + std::to_string(i)).c_str() : "unmasked", lat[lat.size() / 2], lat.front(), lat.back(),
			       adur[adur.size() / 2]);
		}
	}
	return 0;
}
