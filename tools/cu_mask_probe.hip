// cu_mask_probe -- which physical CUs a hipExtStreamCreateWithCUMask bit selects on this device: for a few masks, a
// kernel of many workgroups records (XCC_ID, SE_ID, CU_ID) of the CU it ran on; the distinct ones are printed.
// What the staging slots' partition relies on (ec_hip_staging.cpp): bit i = XCD i % 8, so a contiguous bit range is
// spread evenly over the XCDs and bits {x, x+8, x+16, ...} are the CUs of XCD x.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>
#include <vector>

#define CHECK(x)                                                                                     \
	do {                                                                                         \
		hipError_t e_ = (x);                                                                 \
		if (e_ != hipSuccess) {                                                              \
			fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
			exit(1);                                                                     \
		}                                                                                    \
	} while (0)

__global__ void where(uint32_t *out)
{
	uint32_t xcc, hw;
	asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
	asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
	const unsigned long long t0 = wall_clock64();
	while (wall_clock64() - t0 < 2000)  // 20 us: keeps the CU occupied so that the grid spreads over the whole mask
		__builtin_amdgcn_s_sleep(8);
	if (threadIdx.x == 0) {
		out[2 * blockIdx.x] = xcc;
		out[2 * blockIdx.x + 1] = hw;
	}
}

static std::set<uint32_t> run(const char *name, int ncu, const std::vector<int> &bits)
{
	const int words = (ncu + 31) / 32;
	std::vector<uint32_t> m(words, 0);
	for (int i : bits)
		m[i / 32] |= 1u << (i % 32);
	hipStream_t s;
	CHECK(hipExtStreamCreateWithCUMask(&s, words, m.data()));
	const int nwg = 4096;
	uint32_t *d;
	CHECK(hipMalloc(&d, nwg * 8));
	hipLaunchKernelGGL(where, dim3(nwg), dim3(64), 0, s, d);
	CHECK(hipStreamSynchronize(s));
	std::vector<uint32_t> h(2 * nwg);
	CHECK(hipMemcpy(h.data(), d, nwg * 8, hipMemcpyDeviceToHost));
	std::map<uint32_t, std::set<uint32_t>> per_xcc;
	std::set<uint32_t> all;
	for (int i = 0; i < nwg; ++i) {
		const uint32_t xcc = h[2 * i] & 0xf, hw = h[2 * i + 1];
		const uint32_t cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 0x1, se = (hw >> 13) & 0x7;
		per_xcc[xcc].insert(se * 100 + sh * 50 + cu);
		all.insert(xcc * 1000 + se * 100 + sh * 50 + cu);
	}
	printf("%-34s %3zu bits ->", name, bits.size());
	size_t total = 0;
	for (auto &kv : per_xcc) {
		printf("  xcc%u:%zu", kv.first, kv.second.size());
		total += kv.second.size();
	}
	printf("   (%zu distinct CUs)\n", total);
	CHECK(hipFree(d));
	CHECK(hipStreamDestroy(s));
	return all;
}

int main()
{
	hipDeviceProp_t p;
	CHECK(hipGetDeviceProperties(&p, 0));
	const int ncu = p.multiProcessorCount;
	printf("device: %d CUs\n", ncu);
	auto range = [](int lo, int hi) {
		std::vector<int> v;
		for (int i = lo; i < hi; ++i)
			v.push_back(i);
		return v;
	};
	auto stride8 = [&](int x, int j0, int j1) {
		std::vector<int> v;
		for (int j = j0; j < j1; ++j)
			v.push_back(x + 8 * j);
		return v;
	};
	// the staging slots' partition (ec_hip_staging.cpp, defaults): pairwise disjoint physical CUs, together the whole chip
	std::vector<std::set<uint32_t>> part;
	part.push_back(run("bits [0,16)", ncu, range(0, 16)));
	part.push_back(run("bits [16,32)", ncu, range(16, 32)));
	part.push_back(run("bits [32,40)", ncu, range(32, 40)));
	part.push_back(run("bits [40,192)", ncu, range(40, 192)));
	part.push_back(run("bits [192,256)", ncu, range(192, ncu)));
	std::set<uint32_t> uni;
	size_t sum = 0;
	for (auto &s : part) {
		uni.insert(s.begin(), s.end());
		sum += s.size();
	}
	printf("partition: %zu CUs in the five ranges, %zu distinct -> %s\n", sum, uni.size(),
	       sum == uni.size() && (int)sum == ncu ? "DISJOINT, covers the device" : "OVERLAP or gap");
	run("bits 7+8j, j in [0,32)", ncu, stride8(7, 0, 32));
	run("bits 7+8j, j in [5,21)", ncu, stride8(7, 5, 21));
	run("bits 0+8j, j in [0,32)", ncu, stride8(0, 0, 32));
	run("bit 0", ncu, range(0, 1));
	run("bit 1", ncu, range(1, 2));
	run("bit 8", ncu, range(8, 9));
	return 0;
}
