// fused.hpp -- gf_ptrs_hash: ONE launch for a small trip (round 4).
//
// A PutObject's block through gec_encode_hash_batch used to be a link kernel (parity, mirrored into HBM), a leaf
// kernel and a root kernel on two streams with events between them; a degraded GetObject's block an upload kernel,
// leaf + root kernels, one decode launch per erasure pattern and a copy home.  Every launch pays ~10 us by itself and
// ~36 us more when its queue shares a dispatcher pipe with a running kernel (profiles/r03_qos.txt), which is what was
// left of the small-trip latency.  Here the workgroup that has a tile of a block in hand does everything for it:
//
//   tile = (block b, 256 columns of 16 bytes of its shards) = exactly one GEC_SHARDSUM_LEAF (4 KiB) of every shard.
//   1. the product out[r] = XOR_t coef[r][t] * in[t] over pointer tables, as gf_apply_ptrs does it (nibble product tables
//      in LDS, inputs read from -- rows written to -- pinned caller memory over the link), with the coefficient set
//      chosen PER BLOCK (pat[b]): one launch serves a batch whose blocks lost different shards;
//   2. the k leaves it has just read (and, for a put, the rows it has just computed) are laid down in LDS on the way, and
//      hashed from there with BLAKE2b's leaf parameters, four lanes per leaf (the DPP quad layout of blake2b.hpp, the
//      message words gathered straight out of the resident leaf), the leaves spread over the workgroup's four waves;
//   3. the 64-byte leaf digests go to device memory; the workgroup that finishes a block's last tile (one atomic counter
//      per block) hashes the block's roots from them and writes the 32-byte shard checksums where the host wants them
//      (pinned memory: no copy to launch behind the kernel).
//
// Bound: one lone wave's issue latency -- 32 compressions x ~1.4 us per leaf, 13 per root of a 1 MiB block's shards --
// i.e. ~65 us for ANY batch that fits the chip once, against three to five launches before.  The link and HBM are idle
// by comparison; big batches keep the streaming paths (ec_hip_host.cpp picks by leaf count).
#pragma once

#include "blake2b.hpp"
#include "kernels.hpp"

namespace gec {

// (b2q_hash_lds -- a message that lies in LDS whole, four lanes per message -- lives in blake2b.hpp: mlh_roots_quad uses it too)

template <int MW, int KC>
__global__ __launch_bounds__(256) void gf_ptrs_hash(const FusedArgs a, const LogExp *__restrict__ le)
{
	constexpr int ENT = 4 * MW, TBL = 32 * ENT;
	static_assert(MW == 1 || MW == 2, "rows <= 8");
	extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
	typedef __attribute__((address_space(3))) u32x4 lds_u32x4_w;
	const uint32_t tid = threadIdx.x, k = a.k, rows = a.rows;
	const uint32_t nh = k + (a.hash_rows ? rows : 0);
	uint8_t *lexp = lds + k * TBL, *llog = lexp + 512, *lcoef = llog + 256;
	const uint32_t leaves_off = (k * TBL + 768 + k * RMAX + 15) & ~15u;
	uint32_t *flag = reinterpret_cast<uint32_t *>(lds + leaves_off + nh * FUSED_LEAF_PITCH);
	const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)lds;
	const uint32_t leaves_base = lds_base + leaves_off;
	// which leaf / root of a tile this lane helps to hash: the leaves are dealt round the four waves, four lanes each
	const uint32_t wave = tid >> 6, quad = (tid & 63) >> 2, q = tid & 3;
	const uint32_t hj = quad * 4 + wave;
	const bool hashes = hj < nh;
	const uint32_t my_msg = leaves_base + hj * FUSED_LEAF_PITCH;

	if (tid < 192)
		reinterpret_cast<uint32_t *>(lexp)[tid] = reinterpret_cast<const uint32_t *>(le)[tid];
	uint32_t cur_pat = 0xffffffffu;
	for (uint32_t tile = blockIdx.x; tile < a.tiles_total; tile += gridDim.x) {
		// the workgroup counts as a foreground link kernel only while it is ON the link (a tile's loads and row stores: a
		// fifth of its time): background link kernels give way to that, not to the hashing that follows
		link_enter(a.link_busy, a.link_role);
		const uint32_t b = tile / a.tiles_x, tx = tile - b * a.tiles_x;
		const uint32_t pat = a.pat ? a.pat[b] : 0;
		const uint32_t col_raw = tx * 256 + tid;
		const bool live = col_raw < a.cols;
		const uint32_t col = live ? col_raw : 0;  // dead lanes shadow column 0 (loads only; they lay down zeros)
		const uint8_t *const *inp = a.in + (size_t)b * k;
		const uint32_t *valid = a.in_valid + (size_t)b * k;
		// the tile's first loads go out before anything else: the link's latency hides behind the table expansion
		u32x4 d[KC];
#pragma unroll
		for (int j = 0; j < KC; ++j) {
			const uint32_t t = (uint32_t)j < k ? j : k - 1;
			d[j] = ld16_valid(inp[t], col, valid[t]);
		}
		if (pat != cur_pat) {  // (workgroup-uniform) this block's coefficient set -> nibble product tables
			__syncthreads();
			const uint32_t *cs = reinterpret_cast<const uint32_t *>(a.coef_tab + (size_t)pat * k * RMAX);
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
						const uint32_t c = lcoef[t * RMAX + r];  // rows this set does not have are 0
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
			cur_pat = pat;
		}
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
#pragma unroll
			for (int j = 0; j < KC; ++j) {
				if (t0 + j >= k)
					break;
				// the leaf of input shard t0 + j, as the checksum sees it: zero beyond the shard's end
				const u32x4 keep = live ? d[j] : u32x4{0, 0, 0, 0};
				*reinterpret_cast<lds_u32x4_w *>(leaves_base + (t0 + j) * FUSED_LEAF_PITCH + tid * 16) = keep;
				if (rows == 0)
					continue;
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
		if (rows) {
			uint32_t P[4 * MW][4];
#pragma unroll
			for (int h = 0; h < MW; ++h)
#pragma unroll
				for (int w = 0; w < 4; ++w)
					transpose4x4(acc[w][0][h], acc[w][1][h], acc[w][2][h], acc[w][3][h], P[4 * h + 0][w], P[4 * h + 1][w],
						     P[4 * h + 2][w], P[4 * h + 3][w]);
			uint8_t *const *outp = a.out + (size_t)b * rows;
#pragma unroll
			for (int r = 0; r < 4 * MW; ++r) {
				if (r >= (int)rows)
					continue;
				const u32x4 v = {P[r][0], P[r][1], P[r][2], P[r][3]};
				uint8_t *o = outp[r];
				if (live && o)
					__builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(o) + col);
				if (a.hash_rows)
					*reinterpret_cast<lds_u32x4_w *>(leaves_base + (k + r) * FUSED_LEAF_PITCH + tid * 16) = live ? v : u32x4{0, 0, 0, 0};
			}
		}
		__syncthreads();
		link_leave(a.link_busy, a.link_role);
		// ---- the tile's leaves, out of LDS
		if (hashes) {
			const uint32_t left = (a.cols - tx * 256) * 16;
			const uint32_t len = left < SHARDSUM_LEAF ? left : SHARDSUM_LEAF;
			uint64_t ha = q == 0 ? 0x6a09e667f3bcc908ULL ^ SHARDSUM_P0 : q == 1 ? 0xbb67ae8584caa73bULL ^ (uint64_t)tx /* node_offset */
				    : q == 2 ? 0x3c6ef372fe94f82bULL ^ SHARDSUM_P2_LEAF : 0xa54ff53a5f1d36f1ULL;
			uint64_t hb = q == 0 ? 0x510e527fade682d1ULL : q == 1 ? 0x9b05688c2b3e6c1fULL
				    : q == 2 ? 0x1f83d9abfb41bd6bULL : 0x5be0cd19137e2179ULL;
			b2q_hash_lds(ha, hb, my_msg, (len + 127) / 128, 0, len, true, tx + 1 == a.tiles_x, q);
			uint64_t *o = reinterpret_cast<uint64_t *>(a.leafdig + (((size_t)b * nh + hj) * a.tiles_x + tx) * 64);
			o[q] = ha;
			o[4 + q] = hb;
			__threadfence();  // the digests are out before this workgroup counts itself done
		}
		__syncthreads();
		if (tid == 0) {
			const uint32_t before = __hip_atomic_fetch_add(a.done + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			flag[0] = before + 1 == a.tiles_x ? 1u : 0u;
		}
		__syncthreads();
		if (flag[0]) {  // (workgroup-uniform) this was the block's last tile: its shards' roots
			__threadfence();  // the other workgroups' digests
			const uint64_t total = (uint64_t)a.tiles_x * 64;
			uint64_t ha = q == 0 ? 0x6a09e667f3bcc908ULL ^ SHARDSUM_P0 : q == 1 ? 0xbb67ae8584caa73bULL
				    : q == 2 ? 0x3c6ef372fe94f82bULL ^ SHARDSUM_P2_ROOT : 0xa54ff53a5f1d36f1ULL;
			uint64_t hb = q == 0 ? 0x510e527fade682d1ULL : q == 1 ? 0x9b05688c2b3e6c1fULL
				    : q == 2 ? 0x1f83d9abfb41bd6bULL : 0x5be0cd19137e2179ULL;
			for (uint64_t off = 0; off < total; off += SHARDSUM_LEAF) {  // 64 digests per piece
				const uint32_t bytes = total - off < SHARDSUM_LEAF ? (uint32_t)(total - off) : SHARDSUM_LEAF;
				for (uint32_t j = 0; j < nh; ++j) {
					u32x4 v = {0, 0, 0, 0};
					if (tid * 16 < bytes)
						v = *reinterpret_cast<const u32x4 *>(a.leafdig + ((size_t)b * nh + j) * a.tiles_x * 64 + off + tid * 16);
					*reinterpret_cast<lds_u32x4_w *>(leaves_base + j * FUSED_LEAF_PITCH + tid * 16) = v;
				}
				__syncthreads();
				if (hashes)
					b2q_hash_lds(ha, hb, my_msg, (bytes + 127) / 128, off, total, off + SHARDSUM_LEAF >= total, true, q);
				__syncthreads();
			}
			if (hashes)
				reinterpret_cast<uint64_t *>(a.sums + ((size_t)b * nh + hj) * 32)[q] = ha;  // h[0..3]: the first 32 bytes
			if (tid == 0)
				a.done[b] = 0;  // the counters are all zero again when the launch is over
		}
		__syncthreads();  // flag and the leaves belong to the next tile from here on
	}
}

}  // namespace gec
