// kernels.hpp -- gfx950 (CDNA4) device code of libgarage_ec.
//
// One operation covers encode, reconstruct and verify:
//
//     out[r][b] = XOR_{t<k} mul(mat[r][t], in[t][b])      r < rows, b < S
//
// over GF(2^8)/0x11D, i.e. what reed-solomon-erasure's code_some_slices does
// one MUL_TABLE lookup at a time [EXT core.rs; SURVEY.md Appendix A.3].
//
// MI355X mapping (DESIGN.md "Kernels" has the full derivation):
//  * HBM-bound byte streaming, no MFMA: (k+rows) bytes of traffic per k payload
//    bytes.  Each lane owns 16-byte columns: one global_load_dwordx4 per input
//    shard (a wave covers 1 KiB contiguous per shard, 64-B aligned by the shard
//    geometry), one global_store_dwordx4 per output shard.
//  * GF multiply = two LDS lookups per data byte in *wide nibble product
//    tables*: for input shard t, T_lo[t][x&15] and T_hi[t][x>>4] are 4-byte
//    (rows<=4), 8-byte (rows<=8) or 16-byte (rows<=16) entries holding the
//    products for ALL output rows at once, so one ds_read feeds every parity
//    accumulator.  A 16-entry table occupies 16 (resp. 32, 64) distinct LDS
//    banks, and lanes that hit the same entry broadcast, so the lookups are
//    bank-conflict-free for any data.
//  * The tables are expanded per workgroup from the k x rows coefficient
//    matrix with log/antilog LUTs that are themselves pinned in LDS.
//  * Accumulators stay in VGPRs in "row-interleaved" form (byte r of a dword =
//    output row r); a v_perm_b32 4x4 byte transpose per 4 columns turns them
//    into per-shard dwords just before the store.
//  * One tile of the flattened (block, column) space per workgroup, handed out so
//    that each XCD walks a contiguous range; the prologue (table expansion) runs
//    under the latency of the tile's data loads, which are issued first.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernel_args.hpp"  // argument structs and limits shared with the host-only translation units
#include "mlh64_dev.hpp"    // shard checksum v3: the lane terms the SUM forms of the kernels below accumulate

namespace gec {


typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));


__device__ __forceinline__ void transpose4x4(uint32_t a0, uint32_t a1, uint32_t a2,
					     uint32_t a3, uint32_t &p0, uint32_t &p1,
					     uint32_t &p2, uint32_t &p3)
{
	// a_j = (row0,row1,row2,row3) bytes at column j  ->  p_r = row r, columns 0..3
	uint32_t t0 = __builtin_amdgcn_perm(a1, a0, 0x05010400u);
	uint32_t t1 = __builtin_amdgcn_perm(a1, a0, 0x07030602u);
	uint32_t t2 = __builtin_amdgcn_perm(a3, a2, 0x05010400u);
	uint32_t t3 = __builtin_amdgcn_perm(a3, a2, 0x07030602u);
	p0 = __builtin_amdgcn_perm(t2, t0, 0x05040100u);
	p1 = __builtin_amdgcn_perm(t2, t0, 0x07060302u);
	p2 = __builtin_amdgcn_perm(t3, t1, 0x05040100u);
	p3 = __builtin_amdgcn_perm(t3, t1, 0x07060302u);
}


typedef __attribute__((address_space(3))) const uint32_t lds_u32_t;
typedef __attribute__((address_space(3))) const u32x2 lds_u32x2_t;
typedef __attribute__((address_space(3))) const u32x4 lds_u32x4_t;

// base + byte P of `packed` in ONE VALU op: SDWA selects the byte, the table base
// rides in as the scalar operand.  (hipcc emits shift+and+add for the same C.)
template <int P>
__device__ __forceinline__ uint32_t add_byte(uint32_t sbase, uint32_t packed)
{
	uint32_t r;
	if (P == 0)
		asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "s"(sbase), "v"(packed));
	else if (P == 1)
		asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "s"(sbase), "v"(packed));
	else if (P == 2)
		asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "s"(sbase), "v"(packed));
	else
		asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "s"(sbase), "v"(packed));
	return r;
}

// a ^ b ^ c in ONE VALU op: gfx950's three-input boolean (truth table 0x96 = odd parity).
// hipcc lowers a plain `a ^ b ^ c` to two v_xor_b32.
__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c)
{
#ifdef GEC_NO_XOR3  // A/B switch for tools/kbench
	return a ^ b ^ c;
#else
	return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
#endif
}

// acc ^= T_lo[lo nibble of byte P] ^ T_hi[hi nibble of byte P]
template <int MW, int P>
__device__ __forceinline__ void lut_acc(uint32_t tb, uint32_t lo, uint32_t hi, uint32_t (&acc)[MW])
{
	const uint32_t al = add_byte<P>(tb, lo);
	const uint32_t ah = add_byte<P>(tb, hi);
	if constexpr (MW == 1) {
		acc[0] = xor3(acc[0], *reinterpret_cast<lds_u32_t *>(al), *reinterpret_cast<lds_u32_t *>(ah + 64));
	} else if constexpr (MW == 2) {
		const u32x2 vl = *reinterpret_cast<lds_u32x2_t *>(al);
		const u32x2 vh = *reinterpret_cast<lds_u32x2_t *>(ah + 128);
		acc[0] = xor3(acc[0], vl.x, vh.x);
		acc[1] = xor3(acc[1], vl.y, vh.y);
	} else {  // 16-byte entries: rows 0-3 | 4-7 | 8-11 | 12-15, one ds_read_b128 per nibble
		const u32x4 vl = *reinterpret_cast<lds_u32x4_t *>(al);
		const u32x4 vh = *reinterpret_cast<lds_u32x4_t *>(ah + 256);
		acc[0] = xor3(acc[0], vl.x, vh.x);
		acc[1] = xor3(acc[1], vl.y, vh.y);
		acc[2] = xor3(acc[2], vl.z, vh.z);
		acc[3] = xor3(acc[3], vl.w, vh.w);
	}
}

// ---------------------------------------------------------------------------
// Default kernel: wide nibble product tables in LDS.
//   MW   = dwords per table entry (1: rows<=4, 2: rows<=8, 4: rows<=16 -- reads the data ONCE for
//          codes with 9..16 parity rows instead of once per 8-row group)
//   MODE = store / compare-with-existing
//   KC   = input shards loaded per batch (loads in flight per lane = KC*CPT)
//   CPT  = 16-byte columns per thread per tile (strided by blockDim for coalescing)
//   NT   = non-temporal (streaming) global loads/stores
//   TPB  = threads per workgroup (launch bound; the register budget follows it)
// gf_apply_nibble uses hipcc's default register heuristics; gf_apply_nibble_w pins
// the minimum waves per SIMD (MINW) the register allocator must leave room for.
// ---------------------------------------------------------------------------
template <bool NT>
__device__ __forceinline__ u32x4 ld16(const u32x4 *p)
{
	return NT ? __builtin_nontemporal_load(p) : *p;
}
template <bool NT>
__device__ __forceinline__ void st16(u32x4 v, u32x4 *p)
{
	if (NT)
		__builtin_nontemporal_store(v, p);
	else
		*p = v;
}

// SUM: the kernel also leaves the MLH64 leaf sums (shard checksum v3, mlh64.hpp) of every shard it reads and of every row it
// writes (compare modes: of every stored row it checks) in a.lsum -- from the registers that hold the bytes anyway.  Tiles
// are then cut per block (tile = TPB columns of ONE block, i.e. TPB/256 whole leaves of each of its shards; the last tile of
// a block is ragged) so that a leaf's 256 terms meet inside one workgroup.
// PAT: one coefficient set PER BLOCK (a device-resident batch whose blocks lost different shards, decoded in one launch):
// block b uses entry a.pat[b] of a.pat_tab -- [in_off k x u32][out_off RMAX x u32][rows u32][pad][coef k x RMAX bytes] --
// instead of the launch's own in_off / out_off / coef.  Tiles are cut per block, as for SUM (a workgroup builds ONE table).
template <int MW, int MODE, int KC, int CPT, bool NT, int TPB, bool SUM = false, bool PAT = false>
__device__ __forceinline__ void gf_apply_nibble_body(const ApplyArgs &a, const LogExp *__restrict__ le)
{
	constexpr int ENT = 4 * MW;            // bytes per table entry
	constexpr int TBL = 32 * ENT;          // bytes per input shard (lo 16 | hi 16)
	constexpr uint32_t nthr = TPB;
	static_assert(TPB >= 192, "the log/antilog image is fetched one dword per thread");
	static_assert(MW == 1 || MW == 2 || MW == 4, "table entries are 4, 8 or 16 bytes");
	constexpr int CR = MW == 4 ? RMAX16 : RMAX;  // coefficient bytes per input shard in a.coef / lcoef
	// single dynamic LDS object (no static __shared__ in front of it, so the base
	// stays 16-byte aligned): [tables k*TBL][exp 512][log 256][coef k*CR]
	extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
#ifndef GEC_NO_SETPRIO
	// When the shard-checksum kernels run beside this one (gec_encode_hash_batch_dev forks the data shards'
	// checksums onto a second stream) they are pure VALU work with many waves per SIMD: at equal priority this
	// kernel's few VALU bursts queue behind them and its HBM streams stall (0.25 ms alone -> 0.92 ms beside the
	// hash, profiles/r02_shardsum_kernel_stats.txt).  Raised priority lets the HBM-bound kernel through; alone
	// on the chip it changes nothing.
	__builtin_amdgcn_s_setprio(3);
#endif
	const uint32_t tid = threadIdx.x;
	const uint32_t k = a.k;
	uint32_t rows = a.rows;
	uint8_t *lexp = lds + k * TBL;
	uint8_t *llog = lexp + 512;
	uint8_t *lcoef = llog + 256;
	const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)lds;

	// One tile = TPB*CPT consecutive columns of the FLATTENED (block, column) space per
	// workgroup, dispatched by the hardware: on MI355X this beats a persistent grid-stride
	// loop because fast CUs/XCDs simply pull more tiles (profiles/r01_kbench_*.txt).
	// Flattening means a tile may straddle blocks: no ragged last tile per block, and
	// batches of small blocks (fewer columns per shard than lanes per workgroup) still
	// fill every lane.  Everything up to the first barrier is straight-line code with
	// unconditional (index-clamped) loads so that hipcc can count them: the waits below
	// are vmcnt(N), not vmcnt(0).
	// XCD-aware tile order: workgroup b runs on XCD b % 8 (observed dispatch order, used
	// for speed only -- any placement is correct), so give each XCD a CONTIGUOUS range of
	// tiles instead of every 8th one.  There is no data reuse to keep in an L2 here, but it
	// measures +5.5 % (264 -> 250 us on RS(10,4) x1024, same-box A/B) -- specific to this
	// kernel's 14 streams per workgroup: a plain copy does not benefit (tools/kbench).  The host rounds the grid up
	// to a multiple of 8; surplus workgroups exit here, before any barrier.
	const uint32_t chunk = gridDim.x >> 3;
	const uint32_t tile_id = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
	bool live[CPT];
	uint32_t bb[CPT];
	const u32x4 *srcp[CPT];
	u32x4 *dstp[CPT];
	uint32_t sum_tile = 0;  // SUM: this tile's index inside its block (its first leaf = sum_tile * TPB / 256)
	if constexpr (SUM || PAT) {
		static_assert(CPT == 1 && TPB % 256 == 0, "a leaf is 256 columns: one column per lane, whole leaves per workgroup");
		if (tile_id >= a.nblocks * a.tiles_per_block)
			return;
		bb[0] = tile_id / a.tiles_per_block;
		sum_tile = tile_id - bb[0] * a.tiles_per_block;
		uint32_t col = sum_tile * nthr + tid;
		live[0] = col < a.cols;
		if (!live[0])
			col = sum_tile * nthr;  // lanes past the end of the shard shadow the tile's first column (loads only)
		srcp[0] = reinterpret_cast<const u32x4 *>(a.in + (uint64_t)bb[0] * a.in_stride) + a.col0 + col;
		dstp[0] = reinterpret_cast<u32x4 *>(a.out + (uint64_t)bb[0] * a.out_stride) + a.col0 + col;
	} else {
		if ((uint64_t)tile_id * (nthr * CPT) >= a.total_cols)
			return;
		const uint32_t g0 = tile_id * (nthr * CPT);
#pragma unroll
		for (int c = 0; c < CPT; ++c) {
			uint32_t gcol = g0 + tid + c * nthr;
			live[c] = gcol < a.total_cols;
			if (!live[c])
				gcol = g0;  // lanes past the end shadow the tile's first column (loads only)
			bb[c] = gcol / a.cols;
			const uint32_t col = gcol - bb[c] * a.cols;
			srcp[c] = reinterpret_cast<const u32x4 *>(a.in + (uint64_t)bb[c] * a.in_stride) + a.col0 + col;
			dstp[c] = reinterpret_cast<u32x4 *>(a.out + (uint64_t)bb[c] * a.out_stride) + a.col0 + col;
		}
	}

	// -- prologue 0: log/antilog image (768 B) and coefficient rows (k*CR B) are
	//    requested FIRST, so their wait does not drain the data loads behind them
	//    (index-clamped and stored unconditionally below: a `tid <` branch would let
	//    hipcc sink the load into it, behind the data loads, where it waits vmcnt(0))
	// where this tile's offsets and coefficients come from: the launch's argument block, or the block's pattern entry
	const uint32_t *pin = a.in_off, *pout = a.out_off;
	const uint32_t *pcoef = reinterpret_cast<const uint32_t *>(&a.coef[0][0]);
	if constexpr (PAT) {
		static_assert(MW <= 2, "pattern entries carry RMAX rows");
		const uint8_t *pe = a.pat_tab + (uint64_t)a.pat[bb[0]] * a.pat_stride;
		pin = reinterpret_cast<const uint32_t *>(pe);
		pout = pin + ((k + 3) & ~3u);
		rows = pout[RMAX];
		pcoef = pout + RMAX + 4;  // (16-byte aligned: the entry is, and kp + RMAX + 4 words are a multiple of four)
	}
	const uint32_t le_idx = tid < 192 ? tid : 191;
	const uint32_t le_word = reinterpret_cast<const uint32_t *>(le)[le_idx];
	const uint32_t ncw = k * (CR / 4);  // coefficient dwords
	const uint32_t coef_idx = tid < ncw ? tid : ncw - 1;
	const uint32_t coef_word = pcoef[coef_idx];
	// SUM: this lane's four checksum keys (its column's place in its leaf), zero for a lane past the end of the shard
	mlh_u32x4 k4 = {0, 0, 0, 0};
	constexpr int SUMCAP = TPB > 256 ? 8 : 16;  // slots a wave's LDS region holds between two flushes (>= KC, see the host)
	WaveSums<SUMCAP> ws;
	if constexpr (SUM) {
		static_assert(KC <= SUMCAP, "a batch of loads must fit the wave's region");
		k4 = mlh_keys_of(tid);
		if (!live[0])
			k4 = mlh_u32x4{0, 0, 0, 0};
		ws.init(lds + ((k * TBL + 768 + k * CR + 15) & ~15u), tid, TPB / 64);
	}

	// -- first batch of data loads goes out now; HBM latency covers the table expansion
	u32x4 d[KC][CPT];
#pragma unroll
	for (int j = 0; j < KC; ++j) {
		const uint32_t off = pin[(uint32_t)j < k ? j : k - 1];
#pragma unroll
		for (int c = 0; c < CPT; ++c)
			d[j][c] = ld16<NT>(srcp[c] + off);
	}

	// -- compare mode: the stored rows are requested NOW, behind the data loads, instead of at the end of the tile
	//    where their latency would be exposed (verify is a pure read stream: nothing else is left to hide it)
	constexpr int NOLD = MODE == MODE_COMPARE_PF ? 4 * MW : 1;
	u32x4 oldv[CPT][NOLD];
	if (MODE == MODE_COMPARE_PF) {
#pragma unroll
		for (int r = 0; r < NOLD; ++r) {
			const uint32_t ooff = pout[(uint32_t)r < rows ? r : rows - 1];
#pragma unroll
			for (int c = 0; c < CPT; ++c)
				oldv[c][r] = ld16<NT>(reinterpret_cast<const u32x4 *>(dstp[c]) + ooff);
		}
	}

	// -- prologue 1: pin log/antilog + coefficients in LDS
	reinterpret_cast<uint32_t *>(lexp)[le_idx] = le_word;
	reinterpret_cast<uint32_t *>(lcoef)[coef_idx] = coef_word;
#pragma unroll 1
	for (uint32_t i = tid + nthr; i < ncw; i += nthr)  // k > TPB/2 only
		reinterpret_cast<uint32_t *>(lcoef)[i] = pcoef[i];
	__syncthreads();
	// -- prologue 2: expand coef[k][rows] into wide nibble product tables
	for (uint32_t idx = tid; idx < k * 32; idx += nthr) {
		const uint32_t t = idx >> 5, e = idx & 31;
		const uint32_t x = e < 16 ? e : (e - 16) << 4;
		uint32_t w[MW] = {};
		if (x) {
			const uint32_t lx = llog[x];
#pragma unroll
			for (int r = 0; r < 4 * MW; ++r) {
				const uint32_t c = lcoef[t * CR + r];  // rows beyond `rows` are 0
				const uint32_t p = c ? lexp[llog[c] + lx] : 0;
				w[r >> 2] |= p << (8 * (r & 3));
			}
		}
		uint32_t *tdst = reinterpret_cast<uint32_t *>(lds + t * TBL + e * ENT);
#pragma unroll
		for (int h = 0; h < MW; ++h)
			tdst[h] = w[h];
	}
	__syncthreads();

	// acc[c][w][j][h]: column c, dword w of the column, byte position j, row half h
	uint32_t acc[CPT][4][4][MW];
#pragma unroll
	for (int c = 0; c < CPT; ++c)
#pragma unroll
		for (int w = 0; w < 4; ++w)
#pragma unroll
			for (int j = 0; j < 4; ++j)
#pragma unroll
				for (int h = 0; h < MW; ++h)
					acc[c][w][j][h] = 0;

	for (uint32_t t0 = 0; t0 < k; t0 += KC) {
		if (t0 > 0) {
#pragma unroll
			for (int j = 0; j < KC; ++j) {
				const uint32_t off = pin[t0 + j < k ? t0 + j : k - 1];
#pragma unroll
				for (int c = 0; c < CPT; ++c)
					d[j][c] = ld16<NT>(srcp[c] + off);
			}
		}
		if constexpr (SUM) {
			if (a.sum_inputs && ws.full(KC))
				ws.flush();
		}
#pragma unroll
		for (int j = 0; j < KC; ++j) {
			if (t0 + j >= k)
				break;
			if constexpr (SUM) {
				if (a.sum_inputs)
					ws.put(mlh_col(d[j][0], k4));
			}
			// absolute LDS byte address of this shard's lo table (wave-uniform -> SGPR)
			const uint32_t tb = __builtin_amdgcn_readfirstlane(lds_base + (t0 + j) * TBL);
#pragma unroll
			for (int c = 0; c < CPT; ++c) {
				const uint32_t xs[4] = {d[j][c].x, d[j][c].y, d[j][c].z, d[j][c].w};
#pragma unroll
				for (int w = 0; w < 4; ++w) {
					const uint32_t x = xs[w];
					// entry byte offsets of the lo / hi nibbles of all 4 bytes at once
					const uint32_t lo = (MW == 1) ? ((x << 2) & 0x3C3C3C3Cu) : (MW == 2) ? ((x << 3) & 0x78787878u) : ((x << 4) & 0xF0F0F0F0u);
					const uint32_t hi = (MW == 1) ? ((x >> 2) & 0x3C3C3C3Cu) : (MW == 2) ? ((x >> 1) & 0x78787878u) : (x & 0xF0F0F0F0u);
					lut_acc<MW, 0>(tb, lo, hi, acc[c][w][0]);
					lut_acc<MW, 1>(tb, lo, hi, acc[c][w][1]);
					lut_acc<MW, 2>(tb, lo, hi, acc[c][w][2]);
					lut_acc<MW, 3>(tb, lo, hi, acc[c][w][3]);
				}
			}
		}
	}

#pragma unroll
	for (int c = 0; c < CPT; ++c) {
		uint32_t diff = 0;
		// row-interleaved accumulators -> per-shard dwords
		uint32_t P[4 * MW][4];
#pragma unroll
		for (int h = 0; h < MW; ++h)
#pragma unroll
			for (int w = 0; w < 4; ++w)
				transpose4x4(acc[c][w][0][h], acc[c][w][1][h], acc[c][w][2][h], acc[c][w][3][h],
					     P[4 * h + 0][w], P[4 * h + 1][w], P[4 * h + 2][w], P[4 * h + 3][w]);
		if (!SUM && !live[c])
			continue;
#pragma unroll
		for (int r = 0; r < 4 * MW; ++r) {
			if (r >= (int)rows)  // `continue`, not `break`: with 16 rows hipcc keeps a `break` loop rolled and spills P[] to scratch
				continue;
			u32x4 v = {P[r][0], P[r][1], P[r][2], P[r][3]};
			u32x4 *o = dstp[c] + pout[r];
			if constexpr (SUM) {  // (rows go into the wave's region in groups of at most eight)
				if ((r & 7) == 0 && ws.full(rows - r < 8 ? rows - r : 8))
					ws.flush();
			}
			if (MODE == MODE_COMPARE || MODE == MODE_COMPARE_PF) {
				u32x4 old;
				if constexpr (MODE == MODE_COMPARE_PF)
					old = oldv[c][r < NOLD ? r : 0];
				else
					old = ld16<NT>(o);  // (a lane past the end reads the tile's first column: valid memory, term zeroed by k4)
				diff |= (v.x ^ old.x) | (v.y ^ old.y) | (v.z ^ old.z) | (v.w ^ old.w);
				if constexpr (SUM)
					ws.put(mlh_col(old, k4));  // the STORED row: what the shard's header vouches for
			} else {
				if (!SUM || live[c])
					st16<NT>(v, o);
				if constexpr (SUM)
					ws.put(mlh_col(v, k4));
			}
		}
		if ((MODE == MODE_COMPARE || MODE == MODE_COMPARE_PF) && diff && live[c])
			a.bad[bb[c]] = 1u;
	}
	if constexpr (SUM) {
		// the waves' totals meet: one barrier at the very end of the tile, then 8 bytes per (slot, leaf)
		ws.flush();
		__syncthreads();
		const uint32_t nsl = (a.sum_inputs ? k : 0u) + rows;
		const uint32_t leaf0 = sum_tile * (TPB / 256);
		uint64_t *dst = a.lsum + ((uint64_t)bb[0] * a.sum_slots_total + a.sum_slot0) * a.sum_nleaf_max;
		const uint32_t nleaf = (a.cols + 255) >> 8;
		ws.combine(tid, nthr, nsl, [&](uint32_t slot, uint32_t g, uint64_t v) {
			if (leaf0 + g < nleaf)
				dst[(uint64_t)slot * a.sum_nleaf_max + leaf0 + g] = v;
		});
	}
}

template <int MW, int MODE, int KC, int CPT, bool NT, int TPB>
__global__ __launch_bounds__(TPB) void gf_apply_nibble(const ApplyArgs a, const LogExp *__restrict__ le)
{
	gf_apply_nibble_body<MW, MODE, KC, CPT, NT, TPB>(a, le);
}

// the same kernel with a coefficient set per block (see PAT above)
template <int MW, int KC, bool NT, int TPB>
__global__ __launch_bounds__(TPB) void gf_apply_nibble_pat(const ApplyArgs a, const LogExp *__restrict__ le)
{
	gf_apply_nibble_body<MW, MODE_STORE, KC, 1, NT, TPB, false, true>(a, le);
}

// the same kernel leaving the shard checksums' leaf sums behind (see SUM above)
template <int MW, int MODE, int KC, bool NT, int TPB>
__global__ __launch_bounds__(TPB) void gf_apply_nibble_sum(const ApplyArgs a, const LogExp *__restrict__ le)
{
	gf_apply_nibble_body<MW, MODE, KC, 1, NT, TPB, true>(a, le);
}

template <int MW, int MODE, int KC, int CPT, bool NT, int TPB, int MINW>
__global__ __launch_bounds__(TPB, MINW) void gf_apply_nibble_w(const ApplyArgs a, const LogExp *__restrict__ le)
{
	gf_apply_nibble_body<MW, MODE, KC, CPT, NT, TPB>(a, le);
}

// ---------------------------------------------------------------------------
// gf_apply_ptrs: the same product for shards that stay in the CALLER's memory -- pinned host buffers the device
// addresses over PCIe (gec_host_alloc / gec_host_register).  Shards are named by pointer tables instead of
// base + stride, each input shard carries the number of its bytes that exist (a block's last data shard is
// short; the rest of it reads as zero), and every output row has its own pointer.  One launch replaces
// copy-in kernel + apply + copy-out kernel and nothing is staged in HBM; the link, not the kernel, is the bound
// (tools/pcie_probe: 53 GB/s for this access shape on a Gen5 x16 link), so the kernel is the plain one-column-
// per-lane form of gf_apply_nibble: same LDS tables, same lookups, KC loads in flight per lane.
// Tile = (block, 256 columns of its shards), see the kernel.  The tables themselves live in pinned host memory too:
// they are wave-uniform, i.e. a few scalar loads per workgroup.
// ---------------------------------------------------------------------------

__device__ __forceinline__ u32x4 ld16_valid(const uint8_t *shard, uint32_t col, uint32_t valid)
{
	const uint32_t off = col << 4;
	if (off + 16 <= valid)
		return __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(shard) + col);
	uint32_t w[4] = {0, 0, 0, 0};
	for (uint32_t i = off; i < valid; ++i)  // the one column per block that straddles the end of the data
		w[(i - off) >> 2] |= (uint32_t)shard[i] << (8 * (i & 3));
	return u32x4{w[0], w[1], w[2], w[3]};
}

// see PtrApplyArgs::link_busy
__device__ __forceinline__ void link_enter(uint32_t *busy, uint32_t role)
{
	if (role == LINK_SIGNAL && threadIdx.x == 0)
		__hip_atomic_fetch_add(busy, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void link_leave(uint32_t *busy, uint32_t role)
{
	if (role == LINK_SIGNAL && threadIdx.x == 0)
		__hip_atomic_fetch_sub(busy, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// returns what is left of the wait budget
__device__ __forceinline__ uint32_t link_yield(const uint32_t *busy, uint32_t role, uint32_t budget_ticks)
{
	if (role != LINK_YIELD || budget_ticks == 0)
		return budget_ticks;
	const uint64_t t0 = wall_clock64();
	uint64_t waited = 0;
	while (__hip_atomic_load(busy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 && waited < budget_ticks) {
		__builtin_amdgcn_s_sleep(64);
		waited = wall_clock64() - t0;
	}
	return waited >= budget_ticks ? 0u : budget_ticks - (uint32_t)waited;
}

// SUM: the kernel also leaves the MLH64 leaf sums (shard checksum v3, mlh64_dev.hpp) of the shards it reads (a.sum_inputs) and
// of the rows it writes / checks in a.lsum -- a tile is 256 columns of one block's shards, i.e. exactly one leaf of each --
// so a put on pinned memory needs no mirror in HBM and no second pass: link kernel + one tiny root kernel.
template <int MW, int KC, bool MIRROR, bool COMPARE = false, bool SUM = false>
__global__ __launch_bounds__(256, RESIDENT_WGS) void gf_apply_ptrs(const PtrApplyArgs a, const LogExp *__restrict__ le)
{
	constexpr int ENT = 4 * MW, TBL = 32 * ENT;
	static_assert(MW == 1 || MW == 2, "rows go out in groups of at most 8");
	static_assert(!SUM || KC <= 16, "a batch of loads must fit the wave's region");
	extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
	const uint32_t tid = threadIdx.x, k = a.k, rows = a.rows;
	uint8_t *lexp = lds + k * TBL, *llog = lexp + 512, *lcoef = llog + 256;
	WaveSums<16> ws;
	mlh_u32x4 kbase = {0, 0, 0, 0};
	if constexpr (SUM) {
		ws.init(lds + ((k * TBL + 768 + k * RMAX + 15) & ~15u), tid, 4);
		kbase = mlh_keys_of(tid);
	}
	const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)lds;
	// The grid is 1-D and may be SHORTER than the tile list (a.tiles_total = tiles_x * nblocks; tile = block * tiles_x
	// + 256-column tile of the shard): a workgroup then walks tiles blockIdx.x, + gridDim.x, ...  The host sizes the
	// grid to what the stream's CU partition holds at once (ec_hip_launch.hip, resident_grid): a launch with more
	// workgroups than fit keeps its queue's dispatcher busy until the last one is placed, and kernels of OTHER streams
	// that share that dispatcher wait for as long (tools/dispatch_probe, profiles/r03_qos.txt).
	link_enter(a.link_busy, a.link_role);
	uint32_t wait_left = link_yield(a.link_busy, a.link_role, a.link_wait_ticks);
	uint32_t tile = blockIdx.x;
	uint32_t b = tile / a.tiles_x;
	uint32_t col_raw = (tile - b * a.tiles_x) * 256 + tid;
	bool live = col_raw < a.cols;
	uint32_t col = live ? col_raw : 0;  // dead lanes shadow column 0 (loads only)
	const uint8_t *const *inp = a.in + (size_t)b * k;
	const uint32_t *valid = a.in_valid + (size_t)b * k;

	// first batch of shard loads goes out before the tables are built: PCIe latency hides behind the expansion
	u32x4 d[KC];
#pragma unroll
	for (int j = 0; j < KC; ++j) {
		const uint32_t t = (uint32_t)j < k ? j : k - 1;
		d[j] = ld16_valid(inp[t], col, valid[t]);
	}
	if (tid < 192)
		reinterpret_cast<uint32_t *>(lexp)[tid] = reinterpret_cast<const uint32_t *>(le)[tid];
	// coefficient set -> nibble product tables (the launch's one set, or the set of the block the workgroup is at)
	auto build_tables = [&](const uint32_t *cs) {
		for (uint32_t i = tid; i < k * (RMAX / 4); i += 256)
			reinterpret_cast<uint32_t *>(lcoef)[i] = cs[i];
		__syncthreads();
		for (uint32_t idx = tid; idx < k * 32; idx += 256) {
			const uint32_t t = idx >> 5, e = idx & 31;
			const uint32_t x = e < 16 ? e : (e - 16) << 4;
			uint32_t w[MW] = {};
			if (x) {
				const uint32_t lx = llog[x];
#pragma unroll
				for (int r = 0; r < 4 * MW; ++r) {
					const uint32_t c = lcoef[t * RMAX + r];  // rows beyond `rows` are 0
					const uint32_t p = c ? lexp[llog[c] + lx] : 0;
					w[r >> 2] |= p << (8 * (r & 3));
				}
			}
			uint32_t *tdst = reinterpret_cast<uint32_t *>(lds + t * TBL + e * ENT);
#pragma unroll
			for (int h = 0; h < MW; ++h)
				tdst[h] = w[h];
		}
		__syncthreads();
	};
	uint32_t cur_pat = a.pat ? a.pat[b] : 0;
	__syncthreads();  // lexp / llog are in place
	build_tables(a.pat ? reinterpret_cast<const uint32_t *>(a.coef_tab + (size_t)cur_pat * k * RMAX) : reinterpret_cast<const uint32_t *>(&a.coef[0][0]));

	const uint64_t pace_t0 = a.pace_ticks ? wall_clock64() : 0;
	uint32_t turn = 0;
	for (;;) {  // one tile per turn; no barrier in here unless the block's coefficient set differs from the last tile's
		if (a.pat) {
			const uint32_t p = a.pat[b];
			if (p != cur_pat) {  // (workgroup-uniform)
				__syncthreads();
				build_tables(reinterpret_cast<const uint32_t *>(a.coef_tab + (size_t)p * k * RMAX));
				cur_pat = p;
			}
		}
		u32x4 *mir = MIRROR ? reinterpret_cast<u32x4 *>(a.mirror + (size_t)b * a.mirror_stride) + col : nullptr;
		const bool mir_in = MIRROR && live && a.mirror_inputs;
		const mlh_u32x4 k4 = live ? kbase : mlh_u32x4{0, 0, 0, 0};  // (SUM: a lane past the end of the shard adds nothing)
		uint32_t acc[4][4][MW];
#pragma unroll
		for (int w = 0; w < 4; ++w)
#pragma unroll
			for (int j = 0; j < 4; ++j)
#pragma unroll
				for (int h = 0; h < MW; ++h)
					acc[w][j][h] = 0;
		for (uint32_t t0 = 0; t0 < k; t0 += KC) {
			if (t0 > 0) {
#pragma unroll
				for (int j = 0; j < KC; ++j) {
					const uint32_t t = t0 + j < k ? t0 + j : k - 1;
					d[j] = ld16_valid(inp[t], col, valid[t]);
				}
			}
			if constexpr (SUM) {
				if (a.sum_inputs && ws.full(KC))
					ws.flush();
			}
#pragma unroll
			for (int j = 0; j < KC; ++j) {
				if (t0 + j >= k)
					break;
				if (mir_in)
					mir[(size_t)(t0 + j) * a.cols] = d[j];
				if constexpr (SUM) {
					if (a.sum_inputs)
						ws.put(mlh_col(d[j], k4));
				}
				const uint32_t tb = __builtin_amdgcn_readfirstlane(lds_base + (t0 + j) * TBL);
				const uint32_t xs[4] = {d[j].x, d[j].y, d[j].z, d[j].w};
#pragma unroll
				for (int w = 0; w < 4; ++w) {
					const uint32_t x = xs[w];
					const uint32_t lo = (MW == 1) ? ((x << 2) & 0x3C3C3C3Cu) : ((x << 3) & 0x78787878u);
					const uint32_t hi = (MW == 1) ? ((x >> 2) & 0x3C3C3C3Cu) : ((x >> 1) & 0x78787878u);
					lut_acc<MW, 0>(tb, lo, hi, acc[w][0]);
					lut_acc<MW, 1>(tb, lo, hi, acc[w][1]);
					lut_acc<MW, 2>(tb, lo, hi, acc[w][2]);
					lut_acc<MW, 3>(tb, lo, hi, acc[w][3]);
				}
			}
		}
		// the next tile's first loads go out before this tile's rows are stored
		const uint32_t next = tile + gridDim.x;
		const bool more = next < a.tiles_total;
		const uint32_t nb = more ? next / a.tiles_x : b;
		const uint32_t ncol_raw = more ? (next - nb * a.tiles_x) * 256 + tid : col_raw;
		const bool nlive = ncol_raw < a.cols;
		const uint32_t ncol = nlive ? ncol_raw : 0;
		const uint8_t *const *ninp = a.in + (size_t)nb * k;
		const uint32_t *nvalid = a.in_valid + (size_t)nb * k;
		if (more)
			wait_left = link_yield(a.link_busy, a.link_role, wait_left);
		if (more && a.pace_ticks) {  // (before the next tile's loads: a paced kernel's reads keep the same beat as its writes)
			++turn;
			while (wall_clock64() - pace_t0 < (uint64_t)turn * a.pace_ticks)
				__builtin_amdgcn_s_sleep(16);
		}
		if (more) {
#pragma unroll
			for (int j = 0; j < KC; ++j) {
				const uint32_t t = (uint32_t)j < k ? j : k - 1;
				d[j] = ld16_valid(ninp[t], ncol, nvalid[t]);
			}
		}
		if (live || SUM) {
			uint32_t P[4 * MW][4];
#pragma unroll
			for (int h = 0; h < MW; ++h)
#pragma unroll
				for (int w = 0; w < 4; ++w)
					transpose4x4(acc[w][0][h], acc[w][1][h], acc[w][2][h], acc[w][3][h], P[4 * h + 0][w], P[4 * h + 1][w],
						     P[4 * h + 2][w], P[4 * h + 3][w]);
			uint8_t *const *outp = a.out + (size_t)b * rows;
			if constexpr (SUM) {
				if (ws.full(rows))
					ws.flush();
			}
			if (COMPARE) {
				uint32_t diff = 0;
#pragma unroll
				for (int r = 0; r < 4 * MW; ++r) {
					if (r >= (int)rows)
						continue;
					const u32x4 old = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(outp[r]) + col);  // (col = 0 for a lane past the end)
					diff |= (P[r][0] ^ old.x) | (P[r][1] ^ old.y) | (P[r][2] ^ old.z) | (P[r][3] ^ old.w);
					if (MIRROR && live)  // the STORED row: what the shard checksum is computed over
						reinterpret_cast<u32x4 *>(a.mirror + (size_t)b * a.mirror_stride + a.mirror_row0)[(size_t)r * a.cols + col] = old;
					if constexpr (SUM)
						ws.put(mlh_col(old, k4));
				}
				if (diff && live)
					a.bad[b] = 1u;
			} else {
#pragma unroll
				for (int r = 0; r < 4 * MW; ++r) {
					if (r >= (int)rows)
						continue;
					const u32x4 v = {P[r][0], P[r][1], P[r][2], P[r][3]};
					if constexpr (SUM)
						ws.put(mlh_col(v, k4));  // (a row nobody wants still takes its slot: the host ignores it)
					if (!outp[r] || !live)
						continue;
					__builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(outp[r]) + col);
					if (MIRROR)
						reinterpret_cast<u32x4 *>(a.mirror + (size_t)b * a.mirror_stride + a.mirror_row0)[(size_t)r * a.cols + col] = v;
				}
			}
		}
		if constexpr (SUM) {
			// the four waves' totals of this tile's leaf meet; two barriers per tile (the sums must be read before the next
			// tile's flush overwrites them) -- this kernel is bound by the link, not by them
			ws.flush();
			__syncthreads();
			const uint32_t nsl = (a.sum_inputs ? k : 0u) + rows;
			const uint32_t leaf = tile - b * a.tiles_x;
			uint64_t *dst = a.lsum + ((uint64_t)b * a.sum_slots_total + a.sum_slot0) * a.sum_nleaf_max + leaf;
			ws.combine(tid, 256, nsl, [&](uint32_t slot, uint32_t, uint64_t v) { dst[(uint64_t)slot * a.sum_nleaf_max] = v; });
			__syncthreads();
			ws.base = 0;
		}
		if (!more)
			break;
		tile = next;
		b = nb;
		col_raw = ncol_raw;
		live = nlive;
		col = ncol;
		inp = ninp;
		valid = nvalid;
	}
	link_leave(a.link_busy, a.link_role);
}

// Clears the per-block mismatch flags ahead of a MODE_COMPARE launch.  A kernel rather
// than hipMemsetAsync: inside a captured hipGraph (ROCm 7.0/7.2) the memset node did not
// order the compare kernel behind the preceding encode kernel, so verify could race with
// the encode that produces its input (tests/test_gpu_parity.py::test_device_api_is_
// hipgraph_capturable caught it); kernel -> kernel edges are reliable.
__global__ void clear_flags(uint32_t *p, uint32_t n)
{
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n)
		p[i] = 0;
}

// ---------------------------------------------------------------------------
// copy_table: a batch of independent byte-range copies in ONE launch -- entry e moves `bytes` bytes from src to
// dst; either side may be PINNED HOST memory, which the kernel reads / writes directly over PCIe.  This is how
// caller buffers registered with gec_host_alloc / gec_host_register reach the device stripes and how parity
// gets back: tools/pcie_probe measures 55 GB/s for such a kernel, against 36-47 GB/s for the same pieces as
// individual hipMemcpyAsync calls (per-copy engine overhead) and 57 GB/s for one huge DMA.
// src and dst of every entry are 16-byte aligned (the host checks); the last <16 bytes go byte-wise.
// Tile = (entry, 16 KiB tile of the entry); the tile list covers the largest entry for every entry.
// ---------------------------------------------------------------------------

__global__ __launch_bounds__(256, RESIDENT_WGS) void copy_table(const CopyEntry *__restrict__ tab, uint32_t tiles_x, uint32_t tiles_total,
								  uint32_t pace_ticks, uint32_t *link_busy, uint32_t link_role,
								  uint32_t link_wait_ticks)
{
	link_enter(link_busy, link_role);
	uint32_t wait_left = link_wait_ticks;
	// 1-D grid, possibly shorter than the tile list (see gf_apply_ptrs): tile = entry * tiles_x + 16 KiB tile of the entry.
	// pace_ticks > 0: a workgroup starts its i-th tile no earlier than i * pace_ticks (10 ns each) after its first -- a
	// copy that has the whole trip to finish in (rebuilt shards going home beside the checksum chains) must not fill the
	// fabric's queues towards the link: loads of every other kernel wait behind them (profiles/r03_get_degraded.txt).
	const uint64_t t0 = pace_ticks ? wall_clock64() : 0;
	uint32_t turn = 0;
	for (uint32_t tile = blockIdx.x; tile < tiles_total; tile += gridDim.x, ++turn) {
		if (pace_ticks)
			while (wall_clock64() - t0 < (uint64_t)turn * pace_ticks)
				__builtin_amdgcn_s_sleep(16);
		wait_left = link_yield(link_busy, link_role, wait_left);
		const uint32_t ent = tile / tiles_x, tx = tile - ent * tiles_x;
		const CopyEntry e = tab[ent];
		const uint64_t nvec = e.bytes >> 4;
		const uint64_t base = (uint64_t)tx * 1024 + threadIdx.x;
		if ((uint64_t)tx * 1024 >= nvec + 1)  // whole tile past the end (tail handled by the tile that holds nvec)
			continue;
		const u32x4 *s = reinterpret_cast<const u32x4 *>(e.src);
		u32x4 *d = reinterpret_cast<u32x4 *>(e.dst);
		u32x4 v[4];
#pragma unroll
		for (int j = 0; j < 4; ++j)
			if (base + j * 256 < nvec)
				v[j] = __builtin_nontemporal_load(s + base + j * 256);
#pragma unroll
		for (int j = 0; j < 4; ++j)
			if (base + j * 256 < nvec)
				__builtin_nontemporal_store(v[j], d + base + j * 256);
		// ragged tail: one thread of the tile that contains vector index nvec
		const uint64_t tail = e.bytes & 15;
		if (tail && nvec / 1024 == tx && threadIdx.x == 0)
			for (uint64_t q = 0; q < tail; ++q)
				e.dst[(nvec << 4) + q] = e.src[(nvec << 4) + q];
	}
	link_leave(link_busy, link_role);
}

// ---------------------------------------------------------------------------
// Striped decode, step 3 (gec_group_allgather_decode): after every rank rebuilt its byte
// range of each missing shard inside the gathered buffer, the ranges are exchanged with a
// second all-gather.  range_pack gathers THIS rank's ranges into a dense send buffer
// [nmiss][nobj][max_cols], range_unpack scatters the other ranks' ranges from the receive
// buffer [world][nmiss][nobj][max_cols] back into the shards.  Rank r owns columns
// [cols*r/world, cols*(r+1)/world) of every shard (16-byte columns); ranges differ in
// length by at most one column, max_cols is the longest.  Pure 16-byte copies, HBM-bound,
// a few percent of the decode's traffic.
// ---------------------------------------------------------------------------

__device__ __forceinline__ uint32_t range_lo(uint32_t cols, uint32_t r, uint32_t world)
{
	return (uint32_t)((uint64_t)cols * r / world);
}

__global__ __launch_bounds__(256) void range_pack(const RangeArgs a)
{
	const uint32_t lo = range_lo(a.cols, a.rank, a.world), n = range_lo(a.cols, a.rank + 1, a.world) - lo;
	const uint64_t total = (uint64_t)a.nmiss * a.nobj * n;
	for (uint64_t q = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; q < total; q += (uint64_t)gridDim.x * blockDim.x) {
		const uint32_t c = (uint32_t)(q % n);
		const uint64_t io = q / n;  // i*nobj + obj
		const uint32_t obj = (uint32_t)(io % a.nobj), i = (uint32_t)(io / a.nobj);
		const u32x4 *src = reinterpret_cast<const u32x4 *>(a.gathered + a.shard_off[i] + obj * a.obj_stride) + lo + c;
		reinterpret_cast<u32x4 *>(a.packed)[io * a.max_cols + c] = *src;
	}
}

__global__ __launch_bounds__(256) void range_unpack(const RangeArgs a)
{
	const uint64_t per_rank = (uint64_t)a.nmiss * a.nobj * a.max_cols;
	const uint64_t total = per_rank * a.world;
	for (uint64_t q = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; q < total; q += (uint64_t)gridDim.x * blockDim.x) {
		const uint32_t r = (uint32_t)(q / per_rank);
		if (r == a.rank)
			continue;
		const uint64_t w = q % per_rank;
		const uint32_t c = (uint32_t)(w % a.max_cols);
		const uint32_t lo = range_lo(a.cols, r, a.world), n = range_lo(a.cols, r + 1, a.world) - lo;
		if (c >= n)
			continue;
		const uint64_t io = w / a.max_cols;
		const uint32_t obj = (uint32_t)(io % a.nobj), i = (uint32_t)(io / a.nobj);
		u32x4 *dst = reinterpret_cast<u32x4 *>(a.gathered + a.shard_off[i] + obj * a.obj_stride) + lo + c;
		*dst = reinterpret_cast<const u32x4 *>(a.packed)[q];
	}
}

// ---------------------------------------------------------------------------
// Striped decode, all-to-all exchange (gec_group_alltoall_decode): rank r only needs ITS byte range of the k
// shards the decode reads, so instead of gathering every survivor slot everywhere each rank packs, for every
// peer, that peer's range of its own valid shards -- 1/N of the all-gather's traffic per link.
//   a2a_pack:   send[peer][vs][obj][max_cols]  <-  local[obj][slot(vs)][range of peer]      (vs = index in `slots_used`)
//   rebuilt_unpack: d_rebuilt[i][obj][S]       <-  recv[rank][i][obj][max_cols]              (i = missing shard)
// Pure 16-byte copies.
// ---------------------------------------------------------------------------

__global__ __launch_bounds__(256) void a2a_pack(const A2aArgs a)
{
	const uint64_t per_peer = (uint64_t)a.nvs * a.nobj * a.max_cols;
	const uint64_t total = per_peer * a.world;
	for (uint64_t q = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; q < total; q += (uint64_t)gridDim.x * blockDim.x) {
		const uint32_t peer = (uint32_t)(q / per_peer);
		const uint64_t w = q % per_peer;
		const uint32_t c = (uint32_t)(w % a.max_cols);
		const uint32_t lo = range_lo(a.cols, peer, a.world), n = range_lo(a.cols, peer + 1, a.world) - lo;
		if (c >= n)
			continue;
		const uint64_t vo = w / a.max_cols;  // vs*nobj + obj
		const uint32_t obj = (uint32_t)(vo % a.nobj), vs = (uint32_t)(vo / a.nobj);
		const u32x4 *src = reinterpret_cast<const u32x4 *>(a.local + obj * a.obj_stride) + (uint64_t)a.slot_of[vs] * a.cols + lo + c;
		reinterpret_cast<u32x4 *>(a.send)[((uint64_t)peer * a.nvs_max * a.nobj + vo) * a.max_cols + c] = *src;
	}
}


__global__ __launch_bounds__(256) void rebuilt_unpack(const RebuiltArgs a)
{
	const uint64_t per_rank = (uint64_t)a.nmiss * a.nobj * a.max_cols;
	const uint64_t total = per_rank * a.nranks_in;
	for (uint64_t q = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; q < total; q += (uint64_t)gridDim.x * blockDim.x) {
		const uint32_t r = a.first_rank + (uint32_t)(q / per_rank);
		const uint64_t w = q % per_rank;
		const uint32_t c = (uint32_t)(w % a.max_cols);
		const uint32_t lo = range_lo(a.cols, r, a.world), n = range_lo(a.cols, r + 1, a.world) - lo;
		if (c >= n)
			continue;
		const uint64_t io = w / a.max_cols;  // i*nobj + obj
		reinterpret_cast<u32x4 *>(a.rebuilt)[io * a.cols + lo + c] = reinterpret_cast<const u32x4 *>(a.packed)[q];
	}
}

// ---------------------------------------------------------------------------
// Baseline kernel (variant 1): the literal north_star formulation -- per-byte
// log/antilog lookups in LDS, one GF multiply per (byte, row).  Kept only as the
// measured "before" of DESIGN.md; same results, ~an order of magnitude more LDS
// traffic (conflict-prone byte gathers) than the nibble product tables.
// ---------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(BLOCK) void gf_apply_logexp(const ApplyArgs a, const LogExp *__restrict__ le)
{
	__shared__ __attribute__((aligned(16))) uint8_t lds[768 + RMAX * KMAX];
	uint8_t *lexp = lds;
	uint8_t *llog = lds + 512;
	uint8_t *lcoef = lds + 768;  // log of coefficient, 0xff marks a zero coefficient
	const uint32_t tid = threadIdx.x;
	const uint32_t k = a.k, rows = a.rows;
	for (uint32_t i = tid; i < 768 / 4; i += BLOCK)
		reinterpret_cast<uint32_t *>(lds)[i] = reinterpret_cast<const uint32_t *>(le)[i];
	__syncthreads();
	for (uint32_t i = tid; i < RMAX * KMAX; i += BLOCK) {
		uint32_t r = i / KMAX, t = i % KMAX;
		uint8_t c = (r < rows && t < k) ? a.coef[t][r] : 0;
		lcoef[i] = c ? llog[c] : 0xff;
	}
	__syncthreads();

	const uint32_t ntiles = a.nblocks * a.tiles_per_block;
	for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
		const uint32_t b = tile / a.tiles_per_block;
		const uint32_t col = (tile - b * a.tiles_per_block) * BLOCK + tid;
		if (col >= a.cols)
			continue;
		const u32x4 *src = reinterpret_cast<const u32x4 *>(a.in + (uint64_t)b * a.in_stride) + a.col0 + col;
		u32x4 *dst = reinterpret_cast<u32x4 *>(a.out + (uint64_t)b * a.out_stride) + a.col0 + col;
		uint32_t P[RMAX][4];
#pragma unroll
		for (int r = 0; r < RMAX; ++r)
#pragma unroll
			for (int w = 0; w < 4; ++w)
				P[r][w] = 0;
		for (uint32_t t = 0; t < k; ++t) {
			const u32x4 d = src[a.in_off[t]];
			const uint32_t xs[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
			for (int w = 0; w < 4; ++w)
#pragma unroll
				for (int p = 0; p < 4; ++p) {
					const uint32_t x = (xs[w] >> (8 * p)) & 0xffu;
					const uint32_t lx = llog[x];
#pragma unroll
					for (int r = 0; r < RMAX; ++r) {
						if (r >= (int)rows)
							break;
						const uint32_t lc = lcoef[r * KMAX + t];
						const uint32_t prod = (x && lc != 0xff) ? lexp[lx + lc] : 0;
						P[r][w] ^= prod << (8 * p);
					}
				}
		}
		uint32_t diff = 0;
#pragma unroll
		for (int r = 0; r < RMAX; ++r) {
			if (r >= (int)rows)
				break;
			u32x4 v = {P[r][0], P[r][1], P[r][2], P[r][3]};
			u32x4 *o = dst + a.out_off[r];
			if (MODE == MODE_STORE) {
				*o = v;
			} else {
				u32x4 old = *o;
				diff |= (v.x ^ old.x) | (v.y ^ old.y) | (v.z ^ old.z) | (v.w ^ old.w);
			}
		}
		if (MODE == MODE_COMPARE && diff)
			a.bad[b] = 1u;
	}
}

}  // namespace gec
