// pcie_probe.hip -- what the host link gives, to size the host-pointer API against
// (DESIGN.md "PCIe-inclusive rate"): DMA copies (one big, many 1 MiB pieces, both directions at
// once) and ZERO-COPY kernels that read / write pinned host memory directly.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/pcie_probe tools/pcie_probe.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                        \
	do {                                                                         \
		hipError_t e_ = (x);                                                 \
		if (e_ != hipSuccess) {                                              \
			fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));      \
			exit(1);                                                     \
		}                                                                    \
	} while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// each workgroup streams TILE 16-byte vectors; UNROLL loads in flight per lane
template <int UNROLL>
__global__ __launch_bounds__(256) void zc_copy(const u32x4 *__restrict__ src, u32x4 *__restrict__ dst, size_t nvec)
{
	const size_t per_wg = 256 * UNROLL;
	size_t base = blockIdx.x * per_wg + threadIdx.x;
	u32x4 v[UNROLL];
#pragma unroll
	for (int j = 0; j < UNROLL; ++j)
		v[j] = base + j * 256 < nvec ? __builtin_nontemporal_load(src + base + j * 256) : u32x4{0, 0, 0, 0};
#pragma unroll
	for (int j = 0; j < UNROLL; ++j)
		if (base + j * 256 < nvec)
			__builtin_nontemporal_store(v[j], dst + base + j * 256);
}

// read `in` (10 units) and write `out` (4 units) at once: the RS(10,4) traffic shape
template <int UNROLL>
__global__ __launch_bounds__(256) void zc_rs_shape(const u32x4 *__restrict__ in, u32x4 *__restrict__ out, size_t ncol, size_t S16)
{
	const size_t col = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (col >= ncol)
		return;
	const size_t b = col / S16, c = col % S16;
	u32x4 acc = {0, 0, 0, 0};
	u32x4 d[10];
#pragma unroll
	for (int t = 0; t < 10; ++t)
		d[t] = __builtin_nontemporal_load(in + (b * 10 + t) * S16 + c);
#pragma unroll
	for (int t = 0; t < 10; ++t)
		acc ^= d[t];
#pragma unroll
	for (int r = 0; r < 4; ++r) {
		u32x4 v = {acc.x + r, acc.y, acc.z, acc.w};
		__builtin_nontemporal_store(v, out + (b * 4 + r) * S16 + c);
	}
}

static double ms_of(hipEvent_t a, hipEvent_t b)
{
	float ms;
	CK(hipEventElapsedTime(&ms, a, b));
	return ms;
}

int main(int argc, char **argv)
{
	const size_t MiB = 1 << 20;
	const size_t bytes = (argc > 1 ? strtoull(argv[1], 0, 0) : 1024) * MiB;
	uint8_t *h_in, *h_out, *d_a, *d_b;
	CK(hipHostMalloc((void **)&h_in, bytes, hipHostMallocPortable));
	CK(hipHostMalloc((void **)&h_out, bytes, hipHostMallocPortable));
	CK(hipMalloc((void **)&d_a, bytes));
	CK(hipMalloc((void **)&d_b, bytes));
	for (size_t i = 0; i < bytes; i += 4096)
		h_in[i] = (uint8_t)i, h_out[i] = 1;
	hipStream_t s0, s1;
	CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
	CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
	hipEvent_t e0, e1, f0, f1;
	CK(hipEventCreate(&e0));
	CK(hipEventCreate(&e1));
	CK(hipEventCreate(&f0));
	CK(hipEventCreate(&f1));
	auto report = [&](const char *what, double ms, double gbytes) { printf("%-64s %8.2f ms  %7.2f GB/s\n", what, ms, gbytes / ms); };
	const double GB = bytes / 1e6;  // so that GB/ms = GB/s
	for (int rep = 0; rep < 2; ++rep) {
		CK(hipEventRecord(e0, s0));
		CK(hipMemcpyAsync(d_a, h_in, bytes, hipMemcpyHostToDevice, s0));
		CK(hipEventRecord(e1, s0));
		CK(hipStreamSynchronize(s0));
		if (rep)
			report("H2D one hipMemcpyAsync", ms_of(e0, e1), GB);
		CK(hipEventRecord(e0, s0));
		CK(hipMemcpyAsync(h_out, d_a, bytes, hipMemcpyDeviceToHost, s0));
		CK(hipEventRecord(e1, s0));
		CK(hipStreamSynchronize(s0));
		if (rep)
			report("D2H one hipMemcpyAsync", ms_of(e0, e1), GB);
		CK(hipEventRecord(e0, s0));
		for (size_t o = 0; o < bytes; o += MiB)
			CK(hipMemcpyAsync(d_a + o, h_in + o, MiB, hipMemcpyHostToDevice, s0));
		CK(hipEventRecord(e1, s0));
		CK(hipStreamSynchronize(s0));
		if (rep)
			report("H2D in 1 MiB pieces, one stream", ms_of(e0, e1), GB);
		CK(hipEventRecord(e0, s0));
		CK(hipEventRecord(f0, s1));
		for (size_t o = 0, i = 0; o < bytes; o += MiB, ++i)
			CK(hipMemcpyAsync(d_a + o, h_in + o, MiB, hipMemcpyHostToDevice, (i & 1) ? s1 : s0));
		CK(hipEventRecord(e1, s0));
		CK(hipEventRecord(f1, s1));
		CK(hipStreamSynchronize(s0));
		CK(hipStreamSynchronize(s1));
		if (rep)
			report("H2D in 1 MiB pieces, two streams", std::max(ms_of(e0, e1), ms_of(f0, f1)), GB);
		// both directions at once: 1.0 in, 0.4 out (the RS(10,4) ratio)
		CK(hipEventRecord(e0, s0));
		CK(hipEventRecord(f0, s1));
		CK(hipMemcpyAsync(d_a, h_in, bytes, hipMemcpyHostToDevice, s0));
		CK(hipMemcpyAsync(h_out, d_b, bytes * 2 / 5, hipMemcpyDeviceToHost, s1));
		CK(hipEventRecord(e1, s0));
		CK(hipEventRecord(f1, s1));
		CK(hipStreamSynchronize(s0));
		CK(hipStreamSynchronize(s1));
		if (rep) {
			report("duplex: H2D 1.0 (rate of the H2D leg)", ms_of(e0, e1), GB);
			report("duplex: D2H 0.4 (rate of the D2H leg)", ms_of(f0, f1), GB * 0.4);
		}
		// zero-copy kernels
		const size_t nvec = bytes / 16;
		auto zc = [&](const char *what, const void *src, void *dst, int unroll) {
			CK(hipEventRecord(e0, s0));
			if (unroll == 4)
				zc_copy<4><<<(unsigned)((nvec + 1023) / 1024), 256, 0, s0>>>((const u32x4 *)src, (u32x4 *)dst, nvec);
			else if (unroll == 8)
				zc_copy<8><<<(unsigned)((nvec + 2047) / 2048), 256, 0, s0>>>((const u32x4 *)src, (u32x4 *)dst, nvec);
			else
				zc_copy<1><<<(unsigned)((nvec + 255) / 256), 256, 0, s0>>>((const u32x4 *)src, (u32x4 *)dst, nvec);
			CK(hipEventRecord(e1, s0));
			CK(hipStreamSynchronize(s0));
			if (rep)
				report(what, ms_of(e0, e1), GB);
		};
		zc("zero-copy kernel: read host -> HBM, 1 load/lane", h_in, d_a, 1);
		zc("zero-copy kernel: read host -> HBM, 4 loads/lane", h_in, d_a, 4);
		zc("zero-copy kernel: read host -> HBM, 8 loads/lane", h_in, d_a, 8);
		zc("zero-copy kernel: HBM -> write host, 4/lane", d_a, h_out, 4);
		zc("zero-copy kernel: read host -> write host, 4/lane (duplex)", h_in, h_out, 4);
		{
			const size_t S16 = 104896 / 16, nb = bytes / (10 * 104896);
			const size_t ncol = nb * S16;
			CK(hipEventRecord(e0, s0));
			zc_rs_shape<1><<<(unsigned)((ncol + 255) / 256), 256, 0, s0>>>((const u32x4 *)h_in, (u32x4 *)h_out, ncol, S16);
			CK(hipEventRecord(e1, s0));
			CK(hipStreamSynchronize(s0));
			if (rep)
				report("zero-copy kernel, RS(10,4) shape: 10 host streams in, 4 out (payload rate)", ms_of(e0, e1), nb * 10 * 104896 / 1e6);
		}
	}
	return 0;
}
