// valu_probe.hip -- issue cost and dependent latency (shader cycles per wave64 instruction, one wave per
// SIMD) of the handful of VALU instructions the blake2 chain is made of.  Decides how the 64-bit adds of
// blake2b.hpp are spelled.   build: hipcc --offload-arch=gfx950 -O3 -o tools/valu_probe tools/valu_probe.hip
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

constexpr int ITER = 2000;

#define REP8(x) x x x x x x x x

// KIND: which instruction; DEP: true = every instruction consumes the previous result
template <int KIND, bool DEP>
__global__ __launch_bounds__(64) void probe(uint64_t *out, uint64_t seed, uint32_t active_lanes)
{
	if (threadIdx.x >= active_lanes)  // the rest of the wave runs with a partial EXEC mask
		return;
	uint64_t a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, b = seed | 1;
	uint32_t x0 = (uint32_t)a0, x1 = (uint32_t)a1, x2 = (uint32_t)a2, x3 = (uint32_t)a3, y = (uint32_t)b, sel = 0x02010003;
	const uint64_t t0 = __builtin_amdgcn_s_memtime();
	for (int i = 0; i < ITER; ++i) {
		if (KIND == 0) {  // v_lshl_add_u64
			if (DEP) {
				REP8(asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(a0) : "v"(b));)
			} else {
				REP8(asm volatile("v_lshl_add_u64 %0, %0, 0, %4\n v_lshl_add_u64 %1, %1, 0, %4\n v_lshl_add_u64 %2, %2, 0, %4\n v_lshl_add_u64 %3, %3, 0, %4"
						  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));)
			}
		} else if (KIND == 1) {  // v_add_co_u32 + v_addc_co_u32 through VCC (one 64-bit add = 2 instructions)
			if (DEP) {
				REP8(asm volatile("v_add_co_u32 %0, vcc, %0, %2\n v_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(x0), "+v"(x1) : "v"(y), "v"(sel) : "vcc");)
			} else {
				REP8(asm volatile("v_add_co_u32 %0, vcc, %0, %4\n v_addc_co_u32 %1, vcc, %1, %5, vcc\n v_add_co_u32 %2, vcc, %2, %4\n v_addc_co_u32 %3, vcc, %3, %5, vcc"
						  : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(y), "v"(sel) : "vcc");)
			}
		} else if (KIND == 2) {  // v_xor_b32
			if (DEP) {
				REP8(asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x0) : "v"(y));)
			} else {
				REP8(asm volatile("v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n v_xor_b32 %2, %2, %4\n v_xor_b32 %3, %3, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(y));)
			}
		} else if (KIND == 3) {  // v_alignbit_b32
			if (DEP) {
				REP8(asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(x0) : "v"(y));)
			} else {
				REP8(asm volatile("v_alignbit_b32 %0, %0, %4, 7\n v_alignbit_b32 %1, %1, %4, 7\n v_alignbit_b32 %2, %2, %4, 7\n v_alignbit_b32 %3, %3, %4, 7" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(y));)
			}
		} else if (KIND == 4) {  // v_perm_b32
			if (DEP) {
				REP8(asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x0) : "v"(y), "v"(sel));)
			} else {
				REP8(asm volatile("v_perm_b32 %0, %0, %4, %5\n v_perm_b32 %1, %1, %4, %5\n v_perm_b32 %2, %2, %4, %5\n v_perm_b32 %3, %3, %4, %5" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(y), "v"(sel));)
			}
		} else if (KIND == 5) {  // v_mov_b32_dpp quad_perm
			if (DEP) {
				REP8(asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf" : "+v"(x0));)
			} else {
				REP8(asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %1 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %2 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %3 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));)
			}
		} else if (KIND == 6) {  // v_pk_mov_b32 (64-bit half swap)
			if (DEP) {
				REP8(asm volatile("v_pk_mov_b32 %0, %0, %0 op_sel:[1,0]" : "+v"(a0));)
			} else {
				REP8(asm volatile("v_pk_mov_b32 %0, %0, %0 op_sel:[1,0]\n v_pk_mov_b32 %1, %1, %1 op_sel:[1,0]\n v_pk_mov_b32 %2, %2, %2 op_sel:[1,0]\n v_pk_mov_b32 %3, %3, %3 op_sel:[1,0]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
			}
		} else if (KIND == 7) {  // v_add_co_u32 / v_addc_co_u32 with an SGPR pair other than VCC as the carry
			if (DEP) {
				REP8(asm volatile("v_add_co_u32 %0, s[20:21], %0, %2\n v_addc_co_u32 %1, s[20:21], %1, %3, s[20:21]" : "+v"(x0), "+v"(x1) : "v"(y), "v"(sel) : "s20", "s21");)
			} else {
				REP8(asm volatile("v_add_co_u32 %0, s[20:21], %0, %4\n v_addc_co_u32 %1, s[20:21], %1, %5, s[20:21]\n v_add_co_u32 %2, s[22:23], %2, %4\n v_addc_co_u32 %3, s[22:23], %3, %5, s[22:23]"
						  : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(y), "v"(sel) : "s20", "s21", "s22", "s23");)
			}
		} else if (KIND == 8) {  // v_add3_u32
			if (DEP) {
				REP8(asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x0) : "v"(y), "v"(sel));)
			} else {
				REP8(asm volatile("v_add3_u32 %0, %0, %4, %5\n v_add3_u32 %1, %1, %4, %5\n v_add3_u32 %2, %2, %4, %5\n v_add3_u32 %3, %3, %4, %5" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(y), "v"(sel));)
			}
		}
	}
	const uint64_t t1 = __builtin_amdgcn_s_memtime();
	if (threadIdx.x == 0)
		out[blockIdx.x * 2] = t1 - t0;
	out[blockIdx.x * 2 + 1] = a0 ^ a1 ^ a2 ^ a3 ^ x0 ^ x1 ^ x2 ^ x3;  // keep everything alive
}

template <int KIND>
void run(const char *name, int per_iter_dep, int per_iter_ind, uint64_t *d_out, uint32_t active = 64)
{
	uint64_t h[2];
	double res[2];
	for (int dep = 0; dep < 2; ++dep) {
		for (int rep = 0; rep < 2; ++rep) {
			if (dep)
				probe<KIND, true><<<1024, 64>>>(d_out, 12345, active);
			else
				probe<KIND, false><<<1024, 64>>>(d_out, 12345, active);
			CK(hipDeviceSynchronize());
		}
		CK(hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost));
		res[dep] = (double)h[0] / ITER / (dep ? per_iter_dep : per_iter_ind);
	}
	printf("%-46s independent %6.2f   dependent %6.2f   (s_memtime ticks per wave64 instruction)\n", name, res[0], res[1]);
}

int main()
{
	uint64_t *d_out;
	CK(hipMalloc((void **)&d_out, 1024 * 16));
	printf("# one wave per SIMD (1024 x 64 threads); s_memtime ticks at 100 MHz x ? -- compare rows, v_xor_b32 is the full-rate unit\n");
	run<2>("v_xor_b32", 8, 32, d_out);
	run<0>("v_lshl_add_u64", 8, 32, d_out);
	run<1>("v_add_co_u32 + v_addc_co_u32 via VCC (per instr)", 16, 32, d_out);
	run<7>("v_add_co/addc via SGPR pair (per instr)", 16, 32, d_out);
	run<8>("v_add3_u32", 8, 32, d_out);
	run<3>("v_alignbit_b32", 8, 32, d_out);
	run<4>("v_perm_b32", 8, 32, d_out);
	run<5>("v_mov_b32_dpp quad_perm", 8, 32, d_out);
	run<6>("v_pk_mov_b32", 8, 32, d_out);
	// does the VALU skip the quarter-wave passes whose lanes are all inactive?  (the quad kernel keeps 16 messages per
	// wave busy in 64 lanes; 4 messages in 16 lanes would be 4x faster per message if it did)
	run<2>("v_xor_b32, lanes 0-31 active", 8, 32, d_out, 32);
	run<2>("v_xor_b32, lanes 0-15 active", 8, 32, d_out, 16);
	run<2>("v_xor_b32, lanes 0-3 active", 8, 32, d_out, 4);
	run<0>("v_lshl_add_u64, lanes 0-15 active", 8, 32, d_out, 16);
	run<5>("v_mov_b32_dpp quad_perm, lanes 0-15 active", 8, 32, d_out, 16);
	return 0;
}
