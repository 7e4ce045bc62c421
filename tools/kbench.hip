// kbench.hip -- within-process A/B bench of the gf_apply kernel variants
// (guide rule: perf deltas come from interleaved rounds in ONE process).
// Standalone: build with `make -C tools`, run on the GPU box.
//
//   kbench [k m L nblocks rounds]
//
// Prints one line per variant: median / min kernel time (HIP events), algorithmic
// GB/s ((k+m)*S*nblocks per launch) and whether the parity equals variant 0's.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../garage_amd/csrc/gf256.hpp"
#include "../garage_amd/csrc/kernels.hpp"

#define CK(x)                                                                          \
	do {                                                                           \
		hipError_t e_ = (x);                                                   \
		if (e_ != hipSuccess) {                                                \
			fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));        \
			exit(1);                                                       \
		}                                                                      \
	} while (0)

__global__ void fill_random(uint32_t *p, size_t n, uint32_t seed)
{
	size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
	size_t stride = (size_t)gridDim.x * blockDim.x;
	for (; i < n; i += stride) {
		uint64_t z = (i + 1) * 0x9E3779B97F4A7C15ull + seed;
		z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
		z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
		p[i] = (uint32_t)(z ^ (z >> 31));
	}
}

// plain streaming copy of known size: HBM ceiling + FETCH_SIZE/WRITE_SIZE calibration
__global__ void copy16(const gec::u32x4 *__restrict__ src, gec::u32x4 *__restrict__ dst, size_t n)
{
	size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
	size_t stride = (size_t)gridDim.x * blockDim.x;
	for (; i < n; i += stride)
		__builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}

// 4 independent 16-byte loads in flight per lane, block-contiguous
__global__ void copy16x4(const gec::u32x4 *__restrict__ src, gec::u32x4 *__restrict__ dst, size_t n)
{
	size_t base = (size_t)blockIdx.x * blockDim.x * 4 + threadIdx.x;
	if (base + 3 * (size_t)blockDim.x < n) {
		gec::u32x4 v0 = __builtin_nontemporal_load(src + base);
		gec::u32x4 v1 = __builtin_nontemporal_load(src + base + blockDim.x);
		gec::u32x4 v2 = __builtin_nontemporal_load(src + base + 2 * blockDim.x);
		gec::u32x4 v3 = __builtin_nontemporal_load(src + base + 3 * blockDim.x);
		__builtin_nontemporal_store(v0, dst + base);
		__builtin_nontemporal_store(v1, dst + base + blockDim.x);
		__builtin_nontemporal_store(v2, dst + base + 2 * blockDim.x);
		__builtin_nontemporal_store(v3, dst + base + 3 * blockDim.x);
	}
}

// copy16x4 with the XCD-aware block order of the RS kernel (each XCD a contiguous range)
__global__ void copy16x4_xcd(const gec::u32x4 *__restrict__ src, gec::u32x4 *__restrict__ dst, size_t n)
{
	const unsigned chunk = gridDim.x >> 3;
	const unsigned bid = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
	size_t base = (size_t)bid * blockDim.x * 4 + threadIdx.x;
	if (base + 3 * (size_t)blockDim.x < n) {
		gec::u32x4 v0 = __builtin_nontemporal_load(src + base);
		gec::u32x4 v1 = __builtin_nontemporal_load(src + base + blockDim.x);
		gec::u32x4 v2 = __builtin_nontemporal_load(src + base + 2 * blockDim.x);
		gec::u32x4 v3 = __builtin_nontemporal_load(src + base + 3 * blockDim.x);
		__builtin_nontemporal_store(v0, dst + base);
		__builtin_nontemporal_store(v1, dst + base + blockDim.x);
		__builtin_nontemporal_store(v2, dst + base + 2 * blockDim.x);
		__builtin_nontemporal_store(v3, dst + base + 3 * blockDim.x);
	}
}

// read-only stream (10 reads : 0 writes) -- upper bound for the read side
__global__ void read16x4(const gec::u32x4 *__restrict__ src, gec::u32x4 *__restrict__ dst, size_t n)
{
	size_t base = (size_t)blockIdx.x * blockDim.x * 4 + threadIdx.x;
	if (base + 3 * (size_t)blockDim.x < n) {
		gec::u32x4 v0 = __builtin_nontemporal_load(src + base);
		gec::u32x4 v1 = __builtin_nontemporal_load(src + base + blockDim.x);
		gec::u32x4 v2 = __builtin_nontemporal_load(src + base + 2 * blockDim.x);
		gec::u32x4 v3 = __builtin_nontemporal_load(src + base + 3 * blockDim.x);
		gec::u32x4 x = v0 ^ v1 ^ v2 ^ v3;
		if ((x.x ^ x.y ^ x.z ^ x.w) == 0x12345678u)  // practically never: keeps the loads live
			dst[base] = x;
	}
}

// The RS kernel's ACCESS PATTERN with the arithmetic removed: same flattened XCD-ordered tiles,
// K nt loads per lane (one per input shard), R nt stores (one per output shard), outputs = XOR of
// the inputs rotated per row.  No LDS, no prologue: what HBM gives this 10-read : 4-write
// pattern -- the ceiling the real kernel is measured against.
// ROT: 0 = every wave opens the shards in order 0..K-1; 1 = wave w starts at shard 3w; 2 = tile t starts
// at shard t % K (same set of loads, different issue order).  LNT / SNT: nt loads / stores.
template <int K, int R, int TPB, int ROT = 0, bool LNT = true, bool SNT = true>
__global__ __launch_bounds__(TPB) void stream_pattern(const gec::ApplyArgs a)
{
	const uint32_t chunk = gridDim.x >> 3;
	const uint32_t tile_id = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
	if ((uint64_t)tile_id * TPB >= a.total_cols)
		return;
	uint32_t gcol = tile_id * TPB + threadIdx.x;
	const bool live = gcol < a.total_cols;
	if (!live)
		gcol = tile_id * TPB;
	const uint32_t bb = gcol / a.cols, col = gcol - bb * a.cols;
	const gec::u32x4 *src = reinterpret_cast<const gec::u32x4 *>(a.in + (uint64_t)bb * a.in_stride) + col;
	gec::u32x4 *dst = reinterpret_cast<gec::u32x4 *>(a.out + (uint64_t)bb * a.out_stride) + col;
	uint32_t rot = 0;
	if (ROT == 1)
		rot = __builtin_amdgcn_readfirstlane((threadIdx.x >> 6) * 3) % K;
	if (ROT == 2)
		rot = tile_id % K;
	gec::u32x4 d[K];
#pragma unroll
	for (int j = 0; j < K; ++j) {
		uint32_t t = j + rot;
		t = t >= K ? t - K : t;
		const gec::u32x4 *p = src + a.in_off[ROT ? t : j];
		d[j] = LNT ? __builtin_nontemporal_load(p) : *p;
	}
	gec::u32x4 x = d[0];
#pragma unroll
	for (int j = 1; j < K; ++j)
		x ^= d[j];
	if (!live)
		return;
#pragma unroll
	for (int r = 0; r < R; ++r) {
		gec::u32x4 v = {x.x + r, x.y, x.z, x.w};
		if (SNT)
			__builtin_nontemporal_store(v, dst + a.out_off[r]);
		else
			dst[a.out_off[r]] = v;
	}
}

// k-split pattern (the north_star's "k split across waves + XOR reduction" applied to
// the 8-row shape): the tile is still TPB = 256 columns, but wave w of the 4 opens only shards t = w, w+4, ...
// over ALL 256 columns (4 columns per lane: 4 KiB contiguous per shard per wave instead of 1 KiB), and writes
// only rows r = w, w+4, ... -- 5 input + 2 output streams per wave instead of 20 + 8.  REDUCE = true adds
// what the real kernel would need: every wave's partial vector goes through LDS and the row owner XORs the four.
template <int K, int R, bool REDUCE>
__global__ __launch_bounds__(256) void stream_pattern_ksplit(const gec::ApplyArgs a)
{
	__shared__ gec::u32x4 part[REDUCE ? 4 * 256 : 1];
	const uint32_t chunk = gridDim.x >> 3;
	const uint32_t tile_id = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
	if ((uint64_t)tile_id * 256 >= a.total_cols)
		return;
	const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const gec::u32x4 *src[4];
	gec::u32x4 *dst[4];
	bool live[4];
#pragma unroll
	for (int c = 0; c < 4; ++c) {
		uint32_t gcol = tile_id * 256 + c * 64 + lane;
		live[c] = gcol < a.total_cols;
		if (!live[c])
			gcol = tile_id * 256;
		const uint32_t bb = gcol / a.cols, col = gcol - bb * a.cols;
		src[c] = reinterpret_cast<const gec::u32x4 *>(a.in + (uint64_t)bb * a.in_stride) + col;
		dst[c] = reinterpret_cast<gec::u32x4 *>(a.out + (uint64_t)bb * a.out_stride) + col;
	}
	constexpr int KW = (K + 3) / 4;
	gec::u32x4 d[KW][4];
#pragma unroll
	for (int j = 0; j < KW; ++j) {
		const uint32_t t = __builtin_amdgcn_readfirstlane(wave + 4 * j < (uint32_t)K ? wave + 4 * j : wave);
#pragma unroll
		for (int c = 0; c < 4; ++c)
			d[j][c] = __builtin_nontemporal_load(src[c] + a.in_off[t]);
	}
	gec::u32x4 x[4];
#pragma unroll
	for (int c = 0; c < 4; ++c) {
		x[c] = d[0][c];
#pragma unroll
		for (int j = 1; j < KW; ++j)
			x[c] ^= d[j][c];
	}
	if (REDUCE) {
		// partial of wave w for column (c*64 + lane): 16 KiB of LDS per workgroup
#pragma unroll
		for (int c = 0; c < 4; ++c)
			part[wave * 256 + c * 64 + lane] = x[c];
		__syncthreads();
#pragma unroll
		for (int c = 0; c < 4; ++c)
			x[c] = part[c * 64 + lane] ^ part[256 + c * 64 + lane] ^ part[512 + c * 64 + lane] ^ part[768 + c * 64 + lane];
	}
#pragma unroll
	for (int r = 0; r < (R + 3) / 4; ++r) {
		const uint32_t row = __builtin_amdgcn_readfirstlane(wave + 4 * r);
		if (row >= (uint32_t)R)
			break;
#pragma unroll
		for (int c = 0; c < 4; ++c) {
			if (!live[c])
				continue;
			gec::u32x4 v = {x[c].x + row, x[c].y, x[c].z, x[c].w};
			__builtin_nontemporal_store(v, dst[c] + a.out_off[row]);
		}
	}
}

// Same pattern with explicit cache-policy bits on the loads (LP) and stores (SP), via inline asm:
// 0 = none, 1 = nt, 2 = sc1, 3 = sc0 sc1, 4 = sc0 sc1 nt, 5 = sc1 nt, 6 = sc0, 7 = sc0 nt
#define KB_POLICY(P) ((P) == 0 ? "" : (P) == 1 ? " nt" : (P) == 2 ? " sc1" : (P) == 3 ? " sc0 sc1" : (P) == 4 ? " sc0 sc1 nt" : (P) == 5 ? " sc1 nt" : (P) == 6 ? " sc0" : " sc0 nt")
template <int P>
__device__ __forceinline__ gec::u32x4 asm_load(const gec::u32x4 *p)
{
	gec::u32x4 v;
	if (P == 0) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
	if (P == 1) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
	if (P == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
	if (P == 3) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
	if (P == 4) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt" : "=v"(v) : "v"(p) : "memory");
	if (P == 5) asm volatile("global_load_dwordx4 %0, %1, off sc1 nt" : "=v"(v) : "v"(p) : "memory");
	if (P == 6) asm volatile("global_load_dwordx4 %0, %1, off sc0" : "=v"(v) : "v"(p) : "memory");
	if (P == 7) asm volatile("global_load_dwordx4 %0, %1, off sc0 nt" : "=v"(v) : "v"(p) : "memory");
	return v;
}
template <int P>
__device__ __forceinline__ void asm_store(gec::u32x4 v, gec::u32x4 *p)
{
	if (P == 0) asm volatile("global_store_dwordx4 %0, %1, off" : : "v"(p), "v"(v) : "memory");
	if (P == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" : : "v"(p), "v"(v) : "memory");
	if (P == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
	if (P == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(p), "v"(v) : "memory");
	if (P == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" : : "v"(p), "v"(v) : "memory");
	if (P == 5) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" : : "v"(p), "v"(v) : "memory");
	if (P == 6) asm volatile("global_store_dwordx4 %0, %1, off sc0" : : "v"(p), "v"(v) : "memory");
	if (P == 7) asm volatile("global_store_dwordx4 %0, %1, off sc0 nt" : : "v"(p), "v"(v) : "memory");
}

template <int K, int R, int TPB, int LP, int SP>
__global__ __launch_bounds__(TPB) void stream_policy(const gec::ApplyArgs a)
{
	const uint32_t chunk = gridDim.x >> 3;
	const uint32_t tile_id = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
	if ((uint64_t)tile_id * TPB >= a.total_cols)
		return;
	uint32_t gcol = tile_id * TPB + threadIdx.x;
	const bool live = gcol < a.total_cols;
	if (!live)
		gcol = tile_id * TPB;
	const uint32_t bb = gcol / a.cols, col = gcol - bb * a.cols;
	const gec::u32x4 *src = reinterpret_cast<const gec::u32x4 *>(a.in + (uint64_t)bb * a.in_stride) + col;
	gec::u32x4 *dst = reinterpret_cast<gec::u32x4 *>(a.out + (uint64_t)bb * a.out_stride) + col;
	gec::u32x4 d[K];
#pragma unroll
	for (int j = 0; j < K; ++j)
		d[j] = asm_load<LP>(src + a.in_off[j]);
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	gec::u32x4 x = d[0];
#pragma unroll
	for (int j = 1; j < K; ++j)
		x ^= d[j];
	if (!live)
		return;
#pragma unroll
	for (int r = 0; r < R; ++r) {
		gec::u32x4 v = {x.x + r, x.y, x.z, x.w};
		asm_store<SP>(v, dst + a.out_off[r]);
	}
}

__global__ void diff_count(const uint32_t *a, const uint32_t *b, size_t n, unsigned long long *out)
{
	size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
	size_t stride = (size_t)gridDim.x * blockDim.x;
	unsigned long long c = 0;
	for (; i < n; i += stride)
		c += a[i] != b[i];
	if (c)
		atomicAdd(out, c);
}

typedef void (*kern_t)(const gec::ApplyArgs, const gec::LogExp *);

struct Variant {
	std::string name;
	kern_t fn;
	int threads, cpt, wg_per_cu;  // wg_per_cu <= 0: one tile per workgroup
	bool nibble;
	bool check;
};

int main(int argc, char **argv)
{
	int k = argc > 1 ? atoi(argv[1]) : 10;
	int m = argc > 2 ? atoi(argv[2]) : 4;
	size_t L = argc > 3 ? strtoull(argv[3], 0, 0) : (1u << 20);
	size_t nb = argc > 4 ? strtoull(argv[4], 0, 0) : 1024;
	int rounds = argc > 5 ? atoi(argv[5]) : 7;
	size_t S = ((L + k - 1) / k + 63) / 64 * 64;
	int n = k + m;
	hipDeviceProp_t prop;
	CK(hipGetDeviceProperties(&prop, 0));
	int cus = prop.multiProcessorCount;
	printf("# %s CUs=%d  RS(%d,%d) L=%zu S=%zu nblocks=%zu  algorithmic bytes/launch=%zu\n", prop.name, cus, k, m, L,
	       S, nb, (size_t)n * S * nb);

	size_t bytes = nb * n * S;
	uint8_t *d_st, *d_ref;
	CK(hipMalloc((void **)&d_st, bytes));
	CK(hipMalloc((void **)&d_ref, nb * (size_t)m * S));
	fill_random<<<4096, 256>>>((uint32_t *)d_st, bytes / 4, 12345u);
	CK(hipDeviceSynchronize());

	gec::Matrix enc;
	gec::build_encoding_matrix(k, m, enc);
	gec::LogExp le, *d_le;
	memcpy(le.exp, gec::field().exp.data(), 512);
	memcpy(le.log, gec::field().log.data(), 256);
	CK(hipMalloc((void **)&d_le, sizeof(le)));
	CK(hipMemcpy(d_le, &le, sizeof(le), hipMemcpyHostToDevice));
	unsigned long long *d_cnt;
	CK(hipMalloc((void **)&d_cnt, 8));

	using namespace gec;
	std::vector<Variant> vs;
#define NIBW(MW, KC, CPT, NT, THREADS, MINW, WG, TAG) \
	vs.push_back({TAG, gf_apply_nibble_w<MW, MODE_STORE, KC, CPT, NT, THREADS, MINW>, THREADS, CPT, WG, true, true})
#define NIB(MW, KC, CPT, NT, THREADS, MINW, WG, TAG) \
	vs.push_back({TAG, gf_apply_nibble<MW, MODE_STORE, KC, CPT, NT, THREADS>, THREADS, CPT, WG, true, true})
	const int MW = m <= 4 ? 1 : m <= 8 ? 2 : 4;
	if (MW == 1) {
		NIB(1, 10, 1, true, 256, 0, 0, "nib kc10 cpt1 nt  t256 (default)");
		NIBW(1, 10, 1, true, 256, 5, 0, "nib kc10 cpt1 nt  t256 w5");
		NIBW(1, 10, 1, true, 256, 6, 0, "nib kc10 cpt1 nt  t256 w6");
		NIBW(1, 10, 1, true, 256, 4, 0, "nib kc10 cpt1 nt  t256 w4");
		NIBW(1, 10, 1, true, 256, 3, 0, "nib kc10 cpt1 nt  t256 w3");
		NIB(1, 5, 1, true, 256, 0, 0, "nib kc5  cpt1 nt  t256");
		NIB(1, 4, 1, true, 256, 0, 0, "nib kc4  cpt1 nt  t256");
		NIB(1, 12, 1, true, 256, 0, 0, "nib kc12 cpt1 nt  t256");
		NIB(1, 16, 1, true, 256, 0, 0, "nib kc16 cpt1 nt  t256");
		NIB(1, 20, 1, true, 256, 0, 0, "nib kc20 cpt1 nt  t256");
		NIB(1, 6, 1, true, 256, 0, 0, "nib kc6  cpt1 nt  t256");
		NIB(1, 10, 1, true, 320, 0, 0, "nib kc10 cpt1 nt  t320");
		NIB(1, 10, 1, true, 384, 0, 0, "nib kc10 cpt1 nt  t384");
		NIB(1, 10, 1, true, 192, 0, 0, "nib kc10 cpt1 nt  t192");
		NIB(1, 10, 1, true, 448, 0, 0, "nib kc10 cpt1 nt  t448");
		NIB(1, 10, 1, true, 512, 0, 0, "nib kc10 cpt1 nt  t512");
	} else if (MW == 4) {
		NIB(4, 4, 1, true, 512, 0, 0, "nib16 kc4 cpt1 nt t512 (default)");
		NIB(4, 3, 1, true, 256, 0, 0, "nib16 kc3 cpt1 nt t256");
		NIB(4, 2, 1, true, 256, 0, 0, "nib16 kc2 cpt1 nt t256");
		NIB(4, 4, 1, true, 192, 0, 0, "nib16 kc4 cpt1 nt t192");
		NIB(4, 4, 1, true, 320, 0, 0, "nib16 kc4 cpt1 nt t320");
		NIB(4, 4, 1, true, 384, 0, 0, "nib16 kc4 cpt1 nt t384");
		NIB(4, 4, 1, true, 256, 0, 0, "nib16 kc4 cpt1 nt t256");
	} else {
		NIB(2, 5, 1, true, 512, 0, 0, "nib8 kc5  cpt1 nt t512 (default)");
		NIB(2, 5, 1, true, 256, 0, 0, "nib8 kc5  cpt1 nt t256");
		NIB(2, 4, 1, true, 256, 0, 0, "nib8 kc4  cpt1 nt t256");
		NIB(2, 5, 1, true, 320, 0, 0, "nib8 kc5  cpt1 nt t320");
		NIBW(2, 5, 1, true, 256, 6, 0, "nib8 kc5  cpt1 nt t256 w6");
		NIBW(2, 5, 1, true, 256, 8, 0, "nib8 kc5  cpt1 nt t256 w8");
		NIB(2, 5, 1, true, 512, 0, 0, "nib8 kc5  cpt1 nt t512 (again)");
		NIB(2, 5, 1, true, 384, 0, 0, "nib8 kc5  cpt1 nt t384");
		NIB(2, 5, 1, true, 448, 0, 0, "nib8 kc5  cpt1 nt t448");
		NIB(2, 5, 1, true, 640, 0, 0, "nib8 kc5  cpt1 nt t640");
		NIB(2, 6, 1, true, 512, 0, 0, "nib8 kc6  cpt1 nt t512");
		NIB(2, 10, 1, true, 256, 0, 0, "nib8 kc10 cpt1 nt t256");
		NIBW(2, 10, 1, true, 256, 4, 0, "nib8 kc10 cpt1 nt t256 w4");
		NIBW(2, 5, 1, true, 512, 6, 0, "nib8 kc5  cpt1 nt t512 w6");
		NIBW(2, 5, 1, true, 512, 5, 0, "nib8 kc5  cpt1 nt t512 w5");
	}
	if (MW != 4)  // the baseline kernel takes at most 8 rows (and the 8-byte coefficient layout)
		vs.push_back({"logexp baseline (north_star literal)", gf_apply_logexp<MODE_STORE>, 256, 1, 8, false, true});

	ApplyArgs a;
	memset(&a, 0, sizeof(a));
	a.in = d_st;
	a.out = d_st;
	a.in_stride = a.out_stride = (uint64_t)n * S;
	a.col0 = 0;
	a.cols = (uint32_t)(S / 16);
	a.nblocks = (uint32_t)nb;
	a.k = k;
	a.rows = std::min(m, MW == 4 ? RMAX16 : RMAX);
	const int CR = MW == 4 ? RMAX16 : RMAX;
	for (int t = 0; t < k; ++t)
		a.in_off[t] = (uint32_t)((size_t)t * S / 16);
	for (int r = 0; r < (int)a.rows; ++r) {
		a.out_off[r] = (uint32_t)((size_t)(k + r) * S / 16);
		for (int t = 0; t < k; ++t)
			(&a.coef[0][0])[t * CR + r] = enc.row(k + r)[t];
	}

	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0));
	CK(hipEventCreate(&e1));
	std::vector<std::vector<float>> times(vs.size());
	std::vector<long long> diffs(vs.size(), -1);
	// KBENCH_WG_PER_CU=n pads the dynamic LDS request so that at most n workgroups fit a CU's
	// 160 KiB: an occupancy knob that does not touch the code under test
	size_t lds = (size_t)k * 32 * 4 * MW + 768 + (size_t)k * CR;
	if (getenv("KBENCH_WG_PER_CU")) {
		const size_t cap = (160u << 10) / std::max(1, atoi(getenv("KBENCH_WG_PER_CU")));
		lds = std::max(lds, std::min<size_t>(cap - 512, 64u << 10));
		printf("# LDS request padded to %zu B per workgroup (<= %s workgroups per CU)\n", lds, getenv("KBENCH_WG_PER_CU"));
	}
	const double algo = (double)(k + a.rows) * S * nb;

	auto launch = [&](const Variant &v) {
		ApplyArgs aa = a;
		aa.total_cols = aa.nblocks * a.cols;
		aa.tiles_per_block = (a.cols + v.threads * v.cpt - 1) / (v.threads * v.cpt);
		uint64_t ntiles = ((uint64_t)aa.total_cols + v.threads * v.cpt - 1) / (v.threads * v.cpt);
		unsigned grid = (v.wg_per_cu > 0 && !v.nibble) ? (unsigned)std::min<uint64_t>(ntiles, (uint64_t)cus * v.wg_per_cu) : (unsigned)ntiles;
		if (v.nibble)
			grid = (grid + 7) / 8 * 8;  // XCD-aware tile order needs a multiple of 8
		if (!v.nibble) {
			aa.tiles_per_block = (a.cols + BLOCK - 1) / BLOCK;
			grid = (unsigned)std::min<uint64_t>((uint64_t)aa.nblocks * aa.tiles_per_block, (uint64_t)cus * 8);
		}
		hipLaunchKernelGGL(v.fn, dim3(grid), dim3(v.nibble ? v.threads : BLOCK), v.nibble ? lds : 0, 0, aa, d_le);
	};

	// correctness: every variant's parity vs variant 0's
	launch(vs[0]);
	CK(hipDeviceSynchronize());
	for (size_t b = 0; b < nb; ++b)  // gather parity rows into d_ref
		CK(hipMemcpyAsync(d_ref + b * (size_t)m * S, d_st + b * (size_t)n * S + (size_t)k * S, (size_t)a.rows * S, hipMemcpyDeviceToDevice, 0));
	CK(hipDeviceSynchronize());
	for (size_t i = 0; i < vs.size(); ++i) {
		if (!vs[i].check)
			continue;
		CK(hipMemsetAsync(d_cnt, 0, 8, 0));
		for (size_t b = 0; b < nb; b += 97)  // poison a sample of parity rows first
			CK(hipMemsetAsync(d_st + b * (size_t)n * S + (size_t)k * S, 0xCD, (size_t)a.rows * S, 0));
		launch(vs[i]);
		for (size_t b = 0; b < nb; b += 97)
			diff_count<<<64, 256>>>((const uint32_t *)(d_st + b * (size_t)n * S + (size_t)k * S), (const uint32_t *)(d_ref + b * (size_t)m * S), (size_t)a.rows * S / 4, d_cnt);
		unsigned long long c = 0;
		CK(hipMemcpy(&c, d_cnt, 8, hipMemcpyDeviceToHost));
		diffs[i] = (long long)c;
	}

	const int sustained = getenv("KBENCH_SUSTAINED") ? atoi(getenv("KBENCH_SUSTAINED")) : 0;
	if (sustained > 0) {
		// steady-state (power-managed) rate: N untimed launches to let the DVFS
		// controller settle on this kernel's power draw, then N timed ones
		printf("# sustained mode: %d warm + %d timed back-to-back launches per variant\n", sustained, sustained);
		// KBENCH_FIRST_ONLY=1: only the default variant, then the ceilings (for power sampling)
		for (size_t i = 0; i < (getenv("KBENCH_FIRST_ONLY") ? 1 : vs.size()); ++i) {
			for (int q = 0; q < sustained; ++q)
				launch(vs[i]);
			CK(hipEventRecord(e0, 0));
			for (int q = 0; q < sustained; ++q)
				launch(vs[i]);
			CK(hipEventRecord(e1, 0));
			CK(hipEventSynchronize(e1));
			float ms;
			CK(hipEventElapsedTime(&ms, e0, e1));
			float per = ms / sustained;
			printf("%-52s sustained %8.1f us  %7.1f GB/s  %5.1f%% of 8TB/s  payload %7.1f GiB/s  %s\n", vs[i].name.c_str(), per * 1e3,
			       algo / (per * 1e-3) / 1e9, 100.0 * algo / (per * 1e-3) / 8e12, (double)nb * L / (per * 1e-3) / (1ull << 30),
			       diffs[i] < 0 ? "-" : (diffs[i] == 0 ? "parity==v0" : "PARITY MISMATCH"));
		}
		size_t nvec = bytes / 2 / 16;
		unsigned grid = (unsigned)((nvec + 1023) / 1024);
		for (int q = 0; q < sustained; ++q)
			copy16x4<<<grid, 256>>>((const u32x4 *)d_st, (u32x4 *)(d_st + bytes / 2), nvec);
		CK(hipEventRecord(e0, 0));
		for (int q = 0; q < sustained; ++q)
			copy16x4<<<grid, 256>>>((const u32x4 *)d_st, (u32x4 *)(d_st + bytes / 2), nvec);
		CK(hipEventRecord(e1, 0));
		CK(hipEventSynchronize(e1));
		float ms;
		CK(hipEventElapsedTime(&ms, e0, e1));
		printf("%-52s sustained %8.1f us  %7.1f GB/s\n", "copy16x4 nt (HBM copy ceiling)", ms / sustained * 1e3, 2.0 * nvec * 16 / (ms / sustained * 1e-3) / 1e9);
		// the RS kernel's own access pattern without the arithmetic (k, m of the headline shapes only).
		// NOTE: overwrites the parity area with junk; runs last.
		if ((k == 10 && m == 4) || (k == 20 && m == 8)) {
			ApplyArgs aa = a;
			aa.total_cols = aa.nblocks * a.cols;
			auto run = [&](int tpb, auto kern, const char *tag) {
				unsigned g = (unsigned)(((aa.total_cols + tpb - 1) / tpb + 7) / 8 * 8);
				for (int q = 0; q < sustained; ++q)
					hipLaunchKernelGGL(kern, dim3(g), dim3(tpb), 0, 0, aa);
				CK(hipEventRecord(e0, 0));
				for (int q = 0; q < sustained; ++q)
					hipLaunchKernelGGL(kern, dim3(g), dim3(tpb), 0, 0, aa);
				CK(hipEventRecord(e1, 0));
				CK(hipEventSynchronize(e1));
				CK(hipEventElapsedTime(&ms, e0, e1));
				printf("%-52s sustained %8.1f us  %7.1f GB/s  %5.1f%% of 8TB/s\n", tag, ms / sustained * 1e3, algo / (ms / sustained * 1e-3) / 1e9,
				       algo / (ms / sustained * 1e-3) / 8e12 * 100);
			};
			if (k == 10) {
				run(256, stream_pattern<10, 4, 256>, "access pattern only 10r:4w t256 (pattern ceiling)");
				run(512, stream_pattern<10, 4, 512>, "access pattern only 10r:4w t512");
				run(128, stream_pattern<10, 4, 128>, "access pattern only 10r:4w t128");
				run(256, stream_pattern<10, 4, 256, 1>, "pattern t256, shard order rotated per wave");
				run(256, stream_pattern<10, 4, 256, 2>, "pattern t256, shard order rotated per tile");
				run(256, stream_pattern<10, 4, 256, 0, false, true>, "pattern t256, plain loads, nt stores");
				run(256, stream_pattern<10, 4, 256, 0, true, false>, "pattern t256, nt loads, plain stores");
				run(256, stream_pattern<10, 4, 256, 0, false, false>, "pattern t256, plain loads, plain stores");
				run(256, stream_pattern<10, 4, 256>, "access pattern only 10r:4w t256 (again)");
				run(256, stream_pattern_ksplit<10, 4, false>, "k-split pattern 10r:4w t256: 3r+1w streams per wave, 4 KiB runs");
				run(256, stream_pattern_ksplit<10, 4, true>, "k-split pattern 10r:4w t256 + LDS XOR reduction of partials");
				if (getenv("KBENCH_POLICY")) {
					run(256, stream_policy<10, 4, 256, 1, 1>, "policy: loads nt          stores nt   (= the kernel)");
					run(256, stream_policy<10, 4, 256, 2, 1>, "policy: loads sc1         stores nt");
					run(256, stream_policy<10, 4, 256, 3, 1>, "policy: loads sc0 sc1     stores nt");
					run(256, stream_policy<10, 4, 256, 4, 1>, "policy: loads sc0 sc1 nt  stores nt");
					run(256, stream_policy<10, 4, 256, 5, 1>, "policy: loads sc1 nt      stores nt");
					run(256, stream_policy<10, 4, 256, 6, 1>, "policy: loads sc0         stores nt");
					run(256, stream_policy<10, 4, 256, 7, 1>, "policy: loads sc0 nt      stores nt");
					run(256, stream_policy<10, 4, 256, 1, 2>, "policy: loads nt          stores sc1");
					run(256, stream_policy<10, 4, 256, 1, 3>, "policy: loads nt          stores sc0 sc1");
					run(256, stream_policy<10, 4, 256, 1, 4>, "policy: loads nt          stores sc0 sc1 nt");
					run(256, stream_policy<10, 4, 256, 1, 5>, "policy: loads nt          stores sc1 nt");
					run(256, stream_policy<10, 4, 256, 1, 7>, "policy: loads nt          stores sc0 nt");
					run(256, stream_policy<10, 4, 256, 1, 1>, "policy: loads nt          stores nt   (again)");
				}
			} else {
				run(512, stream_pattern<20, 8, 512>, "access pattern only 20r:8w t512 (pattern ceiling)");
				run(256, stream_pattern<20, 8, 256>, "access pattern only 20r:8w t256");
				run(256, stream_pattern_ksplit<20, 8, false>, "k-split pattern 20r:8w t256: 5r+2w streams per wave, 4 KiB runs");
				run(256, stream_pattern_ksplit<20, 8, true>, "k-split pattern 20r:8w t256 + LDS XOR reduction of partials");
				run(512, stream_pattern<20, 8, 512>, "access pattern only 20r:8w t512 (again)");
				run(256, stream_pattern_ksplit<20, 8, false>, "k-split pattern 20r:8w t256 (again)");
			}
		}
		return 0;
	}
	for (int r = 0; r < rounds; ++r)
		for (size_t i = 0; i < vs.size(); ++i) {
			if (r == 0)
				launch(vs[i]);  // warm
			const int reps = vs[i].nibble ? 5 : 2;
			CK(hipEventRecord(e0, 0));
			for (int q = 0; q < reps; ++q)
				launch(vs[i]);
			CK(hipEventRecord(e1, 0));
			CK(hipEventSynchronize(e1));
			float ms;
			CK(hipEventElapsedTime(&ms, e0, e1));
			times[i].push_back(ms / reps);
		}
	// known-size copy for the HBM ceiling / PMC calibration
	{
		size_t nvec = bytes / 2 / 16;
		std::vector<float> t;
		for (int r = 0; r < rounds + 1; ++r) {
			CK(hipEventRecord(e0, 0));
			copy16<<<cus * 8, 256>>>((const u32x4 *)d_st, (u32x4 *)(d_st + bytes / 2), nvec);
			CK(hipEventRecord(e1, 0));
			CK(hipEventSynchronize(e1));
			float ms;
			CK(hipEventElapsedTime(&ms, e0, e1));
			if (r)
				t.push_back(ms);
		}
		std::sort(t.begin(), t.end());
		printf("%-52s med %8.1f us  min %8.1f us  %7.1f GB/s (read %zu + write %zu bytes)\n", "copy16 nt grid-stride (known bytes)",
		       t[t.size() / 2] * 1e3, t[0] * 1e3, 2.0 * nvec * 16 / (t[t.size() / 2] * 1e-3) / 1e9, nvec * 16, nvec * 16);
		for (int which = 0; which < 3; ++which) {
			t.clear();
			unsigned grid = (unsigned)((nvec + 1023) / 1024);
			if (which == 2)
				grid = (grid + 7) / 8 * 8;
			for (int r = 0; r < rounds + 1; ++r) {
				CK(hipEventRecord(e0, 0));
				if (which == 0)
					copy16x4<<<grid, 256>>>((const u32x4 *)d_st, (u32x4 *)(d_st + bytes / 2), nvec);
				else if (which == 1)
					read16x4<<<grid, 256>>>((const u32x4 *)d_st, (u32x4 *)(d_st + bytes / 2), nvec);
				else
					copy16x4_xcd<<<grid, 256>>>((const u32x4 *)d_st, (u32x4 *)(d_st + bytes / 2), nvec);
				CK(hipEventRecord(e1, 0));
				CK(hipEventSynchronize(e1));
				float ms;
				CK(hipEventElapsedTime(&ms, e0, e1));
				if (r)
					t.push_back(ms);
			}
			std::sort(t.begin(), t.end());
			double bytes_moved = which == 1 ? 1.0 * nvec * 16 : 2.0 * nvec * 16;
			printf("%-52s med %8.1f us  min %8.1f us  %7.1f GB/s\n", which == 0 ? "copy16x4 nt one-tile/wg (HBM copy ceiling)" : which == 1 ? "read16x4 nt one-tile/wg (HBM read ceiling)" : "copy16x4 nt, XCD-contiguous block order",
			       t[t.size() / 2] * 1e3, t[0] * 1e3, bytes_moved / (t[t.size() / 2] * 1e-3) / 1e9);
		}
	}
	for (size_t i = 0; i < vs.size(); ++i) {
		auto t = times[i];
		std::sort(t.begin(), t.end());
		float med = t[t.size() / 2], mn = t[0];
		printf("%-52s med %8.1f us  min %8.1f us  %7.1f GB/s  %5.1f%% of 8TB/s  payload %7.1f GiB/s  %s\n", vs[i].name.c_str(),
		       med * 1e3, mn * 1e3, algo / (med * 1e-3) / 1e9, 100.0 * algo / (med * 1e-3) / 8e12,
		       (double)nb * L / (med * 1e-3) / (1ull << 30),
		       diffs[i] < 0 ? "-" : (diffs[i] == 0 ? "parity==v0" : "PARITY MISMATCH"));
	}
	return 0;
}
