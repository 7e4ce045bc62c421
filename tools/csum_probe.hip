// csum_probe.hip -- THROUGHPUT (not latency) of the multiply-class VALU instructions a lane-combinable shard checksum could be
// built from, with the chip full (4 waves per SIMD), relative to v_xor_b32.  Decides the arithmetic of shardsum v3.
//   build: hipcc --offload-arch=gfx950 -O3 -o tools/csum_probe tools/csum_probe.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                   \
	do {                                                                    \
		hipError_t e_ = (x);                                            \
		if (e_ != hipSuccess) {                                         \
			fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); \
			exit(1);                                                \
		}                                                               \
	} while (0)

constexpr int ITER = 4000;
#define REP4(x) x x x x

template <int KIND>
__global__ __launch_bounds__(256) void probe(uint64_t *out, uint32_t seed)
{
	uint32_t x0 = seed + threadIdx.x, x1 = x0 * 3, x2 = x0 * 5, x3 = x0 * 7, x4 = x0 * 11, x5 = x0 * 13, x6 = x0 * 17, x7 = x0 * 19;
	uint64_t a0 = x0, a1 = x1, a2 = x2, a3 = x3;
	uint32_t y = seed | 1, z = seed * 0x9E3779B9u | 1;
	for (int i = 0; i < ITER; ++i) {
		if (KIND == 0) {
			REP4(asm volatile("v_xor_b32 %0, %0, %8\n v_xor_b32 %1, %1, %8\n v_xor_b32 %2, %2, %8\n v_xor_b32 %3, %3, %8\n v_xor_b32 %4, %4, %8\n v_xor_b32 %5, %5, %8\n v_xor_b32 %6, %6, %8\n v_xor_b32 %7, %7, %8"
					  : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(y));)
		} else if (KIND == 1) {  // v_mad_u64_u32: 32x32 + 64
			REP4(asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %6, %1\n v_mad_u64_u32 %2, vcc, %5, %6, %2\n v_mad_u64_u32 %3, vcc, %6, %6, %3\n"
					  "v_mad_u64_u32 %0, vcc, %5, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %4, %1\n v_mad_u64_u32 %2, vcc, %6, %4, %2\n v_mad_u64_u32 %3, vcc, %5, %4, %3"
					  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x0), "v"(y), "v"(z) : "vcc");)
		} else if (KIND == 2) {  // v_mul_lo_u32
			REP4(asm volatile("v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8"
					  : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(y));)
		} else if (KIND == 3) {  // v_dot2_u32_u16
			REP4(asm volatile("v_dot2_u32_u16 %0, %8, %9, %0\n v_dot2_u32_u16 %1, %8, %9, %1\n v_dot2_u32_u16 %2, %8, %9, %2\n v_dot2_u32_u16 %3, %8, %9, %3\n v_dot2_u32_u16 %4, %8, %9, %4\n v_dot2_u32_u16 %5, %8, %9, %5\n v_dot2_u32_u16 %6, %8, %9, %6\n v_dot2_u32_u16 %7, %8, %9, %7"
					  : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(y), "v"(z));)
		} else if (KIND == 4) {  // v_dot4_u32_u8
			REP4(asm volatile("v_dot4_u32_u8 %0, %8, %9, %0\n v_dot4_u32_u8 %1, %8, %9, %1\n v_dot4_u32_u8 %2, %8, %9, %2\n v_dot4_u32_u8 %3, %8, %9, %3\n v_dot4_u32_u8 %4, %8, %9, %4\n v_dot4_u32_u8 %5, %8, %9, %5\n v_dot4_u32_u8 %6, %8, %9, %6\n v_dot4_u32_u8 %7, %8, %9, %7"
					  : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(y), "v"(z));)
		} else if (KIND == 5) {  // v_pk_mad_u16
			REP4(asm volatile("v_pk_mad_u16 %0, %8, %9, %0\n v_pk_mad_u16 %1, %8, %9, %1\n v_pk_mad_u16 %2, %8, %9, %2\n v_pk_mad_u16 %3, %8, %9, %3\n v_pk_mad_u16 %4, %8, %9, %4\n v_pk_mad_u16 %5, %8, %9, %5\n v_pk_mad_u16 %6, %8, %9, %6\n v_pk_mad_u16 %7, %8, %9, %7"
					  : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(y), "v"(z));)
		} else if (KIND == 6) {  // v_mad_u32_u24
			REP4(asm volatile("v_mad_u32_u24 %0, %8, %9, %0\n v_mad_u32_u24 %1, %8, %9, %1\n v_mad_u32_u24 %2, %8, %9, %2\n v_mad_u32_u24 %3, %8, %9, %3\n v_mad_u32_u24 %4, %8, %9, %4\n v_mad_u32_u24 %5, %8, %9, %5\n v_mad_u32_u24 %6, %8, %9, %6\n v_mad_u32_u24 %7, %8, %9, %7"
					  : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(y), "v"(z));)
		} else if (KIND == 7) {  // v_mul_hi_u32
			REP4(asm volatile("v_mul_hi_u32 %0, %0, %8\n v_mul_hi_u32 %1, %1, %8\n v_mul_hi_u32 %2, %2, %8\n v_mul_hi_u32 %3, %3, %8\n v_mul_hi_u32 %4, %4, %8\n v_mul_hi_u32 %5, %5, %8\n v_mul_hi_u32 %6, %6, %8\n v_mul_hi_u32 %7, %7, %8"
					  : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(y));)
		} else if (KIND == 8) {  // v_lshl_add_u64
			REP4(asm volatile("v_lshl_add_u64 %0, %0, 0, %4\n v_lshl_add_u64 %1, %1, 0, %4\n v_lshl_add_u64 %2, %2, 0, %4\n v_lshl_add_u64 %3, %3, 0, %4\n"
					  "v_lshl_add_u64 %0, %0, 0, %4\n v_lshl_add_u64 %1, %1, 0, %4\n v_lshl_add_u64 %2, %2, 0, %4\n v_lshl_add_u64 %3, %3, 0, %4"
					  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a0 | 1));)
		} else if (KIND == 9) {  // v_add_u32 dpp row_shr (cross-lane add)
			REP4(asm volatile("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n"
					  "v_add_u32_dpp %4, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %5, %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %6, %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %7, %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf"
					  : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));)
		} else if (KIND == 10) {  // v_mul_u32_u24
			REP4(asm volatile("v_mul_u32_u24 %0, %0, %8\n v_mul_u32_u24 %1, %1, %8\n v_mul_u32_u24 %2, %2, %8\n v_mul_u32_u24 %3, %3, %8\n v_mul_u32_u24 %4, %4, %8\n v_mul_u32_u24 %5, %5, %8\n v_mul_u32_u24 %6, %6, %8\n v_mul_u32_u24 %7, %7, %8"
					  : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(y));)
		} else if (KIND == 11) {  // v_pk_mul_lo_u16
			REP4(asm volatile("v_pk_mul_lo_u16 %0, %0, %8\n v_pk_mul_lo_u16 %1, %1, %8\n v_pk_mul_lo_u16 %2, %2, %8\n v_pk_mul_lo_u16 %3, %3, %8\n v_pk_mul_lo_u16 %4, %4, %8\n v_pk_mul_lo_u16 %5, %5, %8\n v_pk_mul_lo_u16 %6, %6, %8\n v_pk_mul_lo_u16 %7, %7, %8"
					  : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(y));)
		}
	}
	out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7;
}

template <int KIND>
void run(const char *name, uint64_t *d_out)
{
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0));
	CK(hipEventCreate(&e1));
	const int grid = 256 * 4;  // 4 workgroups of 4 waves per CU: 4 waves per SIMD
	probe<KIND><<<grid, 256>>>(d_out, 12345);
	CK(hipDeviceSynchronize());
	CK(hipEventRecord(e0));
	probe<KIND><<<grid, 256>>>(d_out, 12345);
	CK(hipEventRecord(e1));
	CK(hipEventSynchronize(e1));
	float ms;
	CK(hipEventElapsedTime(&ms, e0, e1));
	const double insts = (double)grid * 4 /*waves*/ * ITER * 32;
	printf("%-28s %8.3f ms   %8.2f G wave-instr/s   (%.3f per SIMD per ns)\n", name, ms, insts / ms / 1e6, insts / ms / 1e6 / 1024);
}

int main()
{
	uint64_t *d_out;
	CK(hipMalloc((void **)&d_out, 1024 * 256 * 8));
	printf("# 4 waves per SIMD, 32 independent instructions per loop turn; v_xor_b32 is the full-rate unit\n");
	run<0>("v_xor_b32", d_out);
	run<1>("v_mad_u64_u32", d_out);
	run<2>("v_mul_lo_u32", d_out);
	run<7>("v_mul_hi_u32", d_out);
	run<3>("v_dot2_u32_u16", d_out);
	run<4>("v_dot4_u32_u8", d_out);
	run<5>("v_pk_mad_u16", d_out);
	run<11>("v_pk_mul_lo_u16", d_out);
	run<6>("v_mad_u32_u24", d_out);
	run<10>("v_mul_u32_u24", d_out);
	run<8>("v_lshl_add_u64", d_out);
	run<9>("v_add_u32_dpp row_shr:1", d_out);
	return 0;
}
