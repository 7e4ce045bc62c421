// ec_hip_launch.hip -- the one translation unit of libgarage_ec with device code: launch geometry and every kernel
// launch.  Everything else in the HIP backend reaches the kernels through the launch_* / blake2_dev functions
// declared in ec_hip.hpp.
#include "ec_hip.hpp"

#include <algorithm>
#include <map>
#include <mutex>
#include <set>

#include "kernels.hpp"
#include "blake2b.hpp"
#include "fused.hpp"
#include "mlh64_dev.hpp"

namespace gecimpl {

namespace {

std::atomic<int> g_variant{0};
// gec_set_kernel_variant(2) / (3): the BLAKE2 kernels with one lane / four lanes per message whatever the batch size (0 = by size)
static inline int test_route_blake2()
{
	const int v = g_variant.load(std::memory_order_relaxed);
	return v == 2 ? 1 : v == 3 ? 2 : 0;
}

// Launch geometry of the default kernel, tuned on MI355X with tools/kbench
// (profiles/r01_kbench_*.txt): one tile per workgroup, 1 column per thread.
//   4-byte table entries (rows <= 4): 256 threads, up to 10 shards loaded per batch
//     (RS(10,4): all 10 loads go out before the table expansion; 72-74% of 8 TB/s)
//   8-byte table entries (rows <= 8): 512 threads, up to 6 per batch (register budget)
constexpr int kCPT = 1;
constexpr int kThreadsMW1 = 256;
constexpr int kThreadsMW2 = 512;
constexpr int kThreadsMW4 = 512;  // 16-byte entries: 64 accumulator VGPRs per lane, 4 shards per batch (512 beats 256 by 2-5 %)

// Test hook: GEC_MAX_COLS_PER_LAUNCH caps the columns one launch may cover, so the
// multi-launch split (normally only beyond 2^32 columns = 64 GiB per shard slot) can be
// exercised on small inputs.
uint64_t launch_cols_limit() { return env().max_cols_per_launch; }

// Loads per batch (tools/kbench sweeps, profiles/r01_kbench_kc_sweep.txt): k itself when small;
// with 4-byte entries one batch of 10 / 12 / 16 for k <= 16, beyond that batches of 10 whenever
// the duplicate (index-clamped, cache-hit) loads of the last batch stay within a quarter of k --
// fewer, larger batches win even with some waste; otherwise the candidate that wastes the fewest
// (ties: the larger).
int choose_kc(int k, int mw)
{
	if (k <= 6)
		return k;
	if (mw == 2 && k <= 10)  // one batch, in the 256-thread geometry (threads_for): +3-4 % on RS(8,8) / RS(10,8)
		return 10;
	if (mw == 1) {
		// up to 16 shards: ONE batch (all loads in flight before the table expansion) beats two
		// by 2-3 % even at 150 VGPRs / 3 waves per SIMD; 20 in one batch is too many (-10 %)
		if (k <= 10)
			return 10;
		if (k <= 12)
			return 12;
		if (k <= 16)
			return 16;
		const int w10 = (k + 9) / 10 * 10 - k;
		if (w10 * 4 <= k)
			return 10;
	}
	int best = 0, waste = 1 << 30;
	for (int kc : {6, 5, 4}) {
		int w = (k + kc - 1) / kc * kc - k;
		if (w < waste) {
			waste = w;
			best = kc;
		}
	}
	return best;
}

// Workgroup size that goes with (table width, loads per batch).
int threads_for(int mw, int kc)
{
	if (mw == 1)
		return kThreadsMW1;
	if (mw == 2)
		return kc == 10 ? 256 : kThreadsMW2;  // 10 loads in flight per lane need the 256-thread register budget
	return kThreadsMW4;
}

// Everything the host decides about one launch of the default kernel, in one place (also what
// gec_launch_geometry reports, so the invariants -- LDS within 64 KiB, coefficients within the
// argument block -- are testable without a GPU).
struct Geometry {
	int rows;     // output rows this launch takes (<= rows_left)
	int mw;       // dwords per table entry: 1, 2 or 4
	int kc;       // loads per batch
	int threads;  // workgroup size
	size_t lds;   // dynamic LDS bytes: tables + log/antilog image + coefficient rows
};

Geometry pick_geometry(int k, int rows_left, bool rows16_allowed)
{
	Geometry g;
	g.rows = std::min(gec::RMAX, rows_left);
	// More than 8 rows left: 16-byte table entries take up to 16 of them in ONE pass over the
	// data (instead of one pass per 8 rows), as long as k*16 coefficient bytes fit the argument
	// block and k*512 bytes of tables fit 64 KiB of LDS.
	if (rows_left > gec::RMAX && k <= gec::K16MAX && rows16_allowed)
		g.rows = std::min(gec::RMAX16, rows_left);
	// 8-byte table entries need k*256 bytes of LDS; beyond the 64 KiB a workgroup gets without
	// opting in (k > ~245) fall back to groups of 4 rows (4-byte entries)
	if (g.rows > 4 && g.rows <= gec::RMAX && (size_t)k * 256 + 768 + (size_t)k * gec::RMAX > 65536)
		g.rows = 4;
	g.mw = g.rows <= 4 ? 1 : g.rows <= gec::RMAX ? 2 : 4;
	g.kc = g.mw == 4 ? std::min(k, 4) : choose_kc(k, g.mw);  // 16-byte entries: 64 accumulator VGPRs, 4 shards in flight
	g.threads = threads_for(g.mw, g.kc);
	const int cr = g.mw == 4 ? gec::RMAX16 : gec::RMAX;
	g.lds = (size_t)k * 32 * 4 * g.mw + 768 + (size_t)k * cr;
	return g;
}

template <int MW, int MODE, int KC, int TPB, bool SUM>
void launch_one(const gec::ApplyArgs &a, const gec::LogExp *le, unsigned grid, size_t lds, hipStream_t s)
{
	if constexpr (SUM)
		hipLaunchKernelGGL((gec::gf_apply_nibble_sum<MW, MODE, KC, true, TPB>), dim3(grid), dim3(TPB), lds, s, a, le);
	else
		hipLaunchKernelGGL((gec::gf_apply_nibble<MW, MODE, KC, kCPT, true, TPB>), dim3(grid), dim3(TPB), lds, s, a, le);
}

// SUM: the form that also leaves the shard checksums' leaf sums (mlh64_dev.hpp); 4- and 8-byte table entries only
template <int MW, int MODE, int TPB, bool SUM = false>
void launch_nibble(const gec::ApplyArgs &a, const gec::LogExp *le, int kc, unsigned grid, size_t lds, hipStream_t s)
{
	if constexpr (MW == 2) {
		if (kc == 10) {
			launch_one<MW, MODE, 10, 256, SUM>(a, le, grid, lds, s);
			return;
		}
	}
	if constexpr (MW == 1) {
		if (kc == 10) {
			launch_one<MW, MODE, 10, TPB, SUM>(a, le, grid, lds, s);
			return;
		}
		if (kc == 12) {
			launch_one<MW, MODE, 12, TPB, SUM>(a, le, grid, lds, s);
			return;
		}
		if (kc == 16) {
			launch_one<MW, MODE, 16, TPB, SUM>(a, le, grid, lds, s);
			return;
		}
	}
#define GEC_CASE(KC)                                                \
	case KC:                                                    \
		launch_one<MW, MODE, KC, TPB, SUM>(a, le, grid, lds, s); \
		break;
	if constexpr (MW == 4) {
		switch (kc) {
			GEC_CASE(1)
			GEC_CASE(2)
			GEC_CASE(3)
			GEC_CASE(4)
		}
	} else {
		switch (kc) {
			GEC_CASE(1)
			GEC_CASE(2)
			GEC_CASE(3)
			GEC_CASE(4)
			GEC_CASE(5)
			GEC_CASE(6)
		}
	}
#undef GEC_CASE
}

// LDS the SUM form adds: the waves' term regions + wsum (mlh64_dev.hpp), behind the tables at a 16-byte boundary
size_t sum_lds_bytes(const Geometry &g, int k, int nsl)
{
	const int cap = g.threads > 256 ? 8 : 16;
	return ((g.lds + 15) & ~(size_t)15) + gec::mlh_lds_bytes(cap, g.threads / 64, nsl) - g.lds;
}


// The roots of `a.n` shards from their leaf sums: four lanes per shard while a shard's message fits the workgroup's LDS (every
// geometry of the BASELINE configs: 26 and 52 leaves), one lane per shard beyond.
int launch_mlh_roots(const gec::Blake2Args &a, uint32_t nleaf_max, const uint64_t *lsum, const uint32_t *slot_map, hipStream_t stream)
{
	if (gec::mlh_roots_quad_fits(nleaf_max))
		hipLaunchKernelGGL(gec::mlh_roots_quad, dim3((a.n + 15) / 16), dim3(64), gec::mlh_rootq_lds_bytes(nleaf_max), stream, a, nleaf_max, lsum, slot_map);
	else
		hipLaunchKernelGGL(gec::mlh_roots, dim3((a.n + 63) / 64), dim3(64), 0, stream, a, nleaf_max, lsum, slot_map);
	HIP_TRY(hipGetLastError());
	return GEC_OK;
}
}  // namespace

// out[r] = XOR_t coef[r][t] * in[t] for r < nout: shard t of block b is read at
// in + b*in_stride + in_base_off[t], row r written at out + b*out_stride +
// out_base_off[r]; only bytes [byte_off, byte_off+byte_len) of every shard are
// touched.  Rows go out in groups of RMAX per launch.
int launch_apply(const gec_codec *c, const uint8_t *in, size_t in_stride, uint8_t *out, size_t out_stride,
		 uint32_t *bad, size_t byte_off, size_t byte_len, size_t nblocks, const size_t *in_base_off,
		 const size_t *out_base_off, int nout, const uint8_t *coef /* nout x k */, int mode,
		 hipStream_t stream, const SumOut *sum)
{
	const int k = c->k;
	const HipBackend &hb = hip_of(c);
	if (nblocks == 0 || nout == 0 || byte_len == 0)
		return GEC_OK;
	if (nblocks > 0xffffffffull || (byte_len / 16) > 0x7fffffffull)
		return fail(GEC_E_INVALID_ARG, "batch too large for one call");
	gec::ApplyArgs a;
	std::memset(&a, 0, sizeof(a));
	a.in = in;
	a.out = out;
	a.bad = bad;
	a.in_stride = in_stride;
	a.out_stride = out_stride;
	a.col0 = (uint32_t)(byte_off / 16);
	a.cols = (uint32_t)(byte_len / 16);
	a.nblocks = (uint32_t)nblocks;
	a.k = (uint32_t)k;
	for (int t = 0; t < k; ++t) {
		if (in_base_off[t] / 16 > 0xffffffffull)
			return fail(GEC_E_INVALID_ARG, "stripe too large");
		a.in_off[t] = (uint32_t)(in_base_off[t] / 16);
	}
	const int variant = sum ? 0 : (g_variant.load(std::memory_order_relaxed) == 1 ? 1 : 0);  // (2..4 are routes of other kernels)
	if (sum) {
		// leaves are 256 columns of a WHOLE shard: the sums of a byte range would not be the shard's
		if (byte_off != 0 || byte_len % 16 || (size_t)sum->nleaf_max < (byte_len + gec::SHARDSUM_LEAF - 1) / gec::SHARDSUM_LEAF)
			return fail(GEC_E_INVALID_ARG, "checksummed launch: whole shards only");
		a.lsum = sum->lsum;
		a.sum_nleaf_max = sum->nleaf_max;
		a.sum_slots_total = sum->slots_total;
	}
	int rows = 0;
	for (int r0 = 0; r0 < nout; r0 += rows) {
		// (the SUM form exists for 4- and 8-byte table entries: more than 8 rows go out in groups of 8)
		const Geometry geo = pick_geometry(k, nout - r0, variant == 0 && !sum);
		rows = variant == 1 ? std::min(gec::RMAX, nout - r0) : geo.rows;  // the baseline kernel takes up to 8 rows
		a.rows = (uint32_t)rows;
		const int mw = variant == 1 ? 2 : geo.mw;
		const int cr = mw == 4 ? gec::RMAX16 : gec::RMAX;  // coefficient bytes per input shard
		uint8_t *flat = &a.coef[0][0];
		for (int r = 0; r < cr; ++r) {
			if (r < rows) {
				if (out_base_off[r0 + r] / 16 > 0xffffffffull)  // same limit as the input offsets: 64 GiB per stripe
					return fail(GEC_E_INVALID_ARG, "stripe too large");
				a.out_off[r] = (uint32_t)(out_base_off[r0 + r] / 16);
			}
			for (int t = 0; t < k; ++t)
				flat[(size_t)t * cr + r] = r < rows ? coef[(size_t)(r0 + r) * k + t] : 0;
		}
		if (variant == 1) {
			// measured baseline: persistent grid-stride log/antilog kernel, one tile = 256
			// columns of one block
			a.tiles_per_block = (a.cols + gec::BLOCK - 1) / gec::BLOCK;
			const uint64_t ntiles = (uint64_t)a.nblocks * a.tiles_per_block;
			if (ntiles > 0xffffffffull)
				return fail(GEC_E_INVALID_ARG, "batch too large for the baseline kernel");
			const unsigned grid = (unsigned)std::min<uint64_t>(ntiles, (uint64_t)hb.num_cu * 8);
			if (mode == gec::MODE_STORE)
				hipLaunchKernelGGL((gec::gf_apply_logexp<gec::MODE_STORE>), dim3(grid), dim3(gec::BLOCK), 0, stream, a, hb.d_logexp);
			else
				hipLaunchKernelGGL((gec::gf_apply_logexp<gec::MODE_COMPARE>), dim3(grid), dim3(gec::BLOCK), 0, stream, a, hb.d_logexp);
			HIP_TRY(hipGetLastError());
			continue;
		}
		// The (block, column) space is flattened: a launch covers a range of whole blocks
		// whose columns fit 32 bits and whose tiles fit HIP's grid limit (grid*block < 2^32).
		const int threads = geo.threads;
		const uint64_t tile_cols = (uint64_t)threads * kCPT;
		uint64_t max_cols = std::min<uint64_t>(0xfffff000ull, (0xffffffffull / threads - 8) * tile_cols);
		if (launch_cols_limit())
			max_cols = std::min<uint64_t>(max_cols, launch_cols_limit());
		if (a.cols > max_cols)
			return fail(GEC_E_INVALID_ARG, "shard too large for one launch");
		const uint64_t blocks_per_launch = std::max<uint64_t>(1, max_cols / a.cols);
		size_t lds = geo.lds;
		gec::ApplyArgs la = a;
		if (sum) {
			// inputs are summed by the first row group only; rows by the group that produces (or checks) them
			la.sum_inputs = (r0 == 0 && sum->inputs) ? 1u : 0u;
			la.sum_slot0 = sum->slot0 + (la.sum_inputs ? 0u : (sum->inputs ? (uint32_t)k : 0u) + (uint32_t)r0);
			la.tiles_per_block = (uint32_t)((a.cols + tile_cols - 1) / tile_cols);
			lds += sum_lds_bytes(geo, k, (la.sum_inputs ? k : 0) + rows);
		}
		for (uint64_t b0 = 0; b0 < nblocks; b0 += blocks_per_launch) {
			const uint64_t nb = std::min<uint64_t>(blocks_per_launch, nblocks - b0);
			la.in = in + b0 * in_stride;
			la.out = out + b0 * out_stride;
			la.bad = bad ? bad + b0 : nullptr;
			la.nblocks = (uint32_t)nb;
			la.total_cols = (uint32_t)(nb * a.cols);
			// multiple of 8: the kernel hands each XCD a contiguous range of tiles
			unsigned grid = (unsigned)(((la.total_cols + tile_cols - 1) / tile_cols + 7) / 8 * 8);
			if (sum) {
				la.lsum = sum->lsum + b0 * sum->slots_total * sum->nleaf_max;
				grid = (unsigned)((nb * la.tiles_per_block + 7) / 8 * 8);  // tiles are cut per block (blocks_per_launch keeps this in range: a ragged tile per block at most)
				if (mw == 1 && mode == gec::MODE_STORE)
					launch_nibble<1, gec::MODE_STORE, kThreadsMW1, true>(la, hb.d_logexp, geo.kc, grid, lds, stream);
				else if (mw == 1)
					launch_nibble<1, gec::MODE_COMPARE, kThreadsMW1, true>(la, hb.d_logexp, geo.kc, grid, lds, stream);
				else if (mode == gec::MODE_STORE)
					launch_nibble<2, gec::MODE_STORE, kThreadsMW2, true>(la, hb.d_logexp, geo.kc, grid, lds, stream);
				else
					launch_nibble<2, gec::MODE_COMPARE, kThreadsMW2, true>(la, hb.d_logexp, geo.kc, grid, lds, stream);
				HIP_TRY(hipGetLastError());
				continue;
			}
			if (mw == 1 && mode == gec::MODE_STORE)
				launch_nibble<1, gec::MODE_STORE, kThreadsMW1>(la, hb.d_logexp, geo.kc, grid, lds, stream);
			else if (mw == 1 && rows == 4)  // all four row slots real: stored rows prefetched behind the data loads
				launch_nibble<1, gec::MODE_COMPARE_PF, kThreadsMW1>(la, hb.d_logexp, geo.kc, grid, lds, stream);
			else if (mw == 1)
				launch_nibble<1, gec::MODE_COMPARE, kThreadsMW1>(la, hb.d_logexp, geo.kc, grid, lds, stream);
			else if (mw == 2 && mode == gec::MODE_STORE)
				launch_nibble<2, gec::MODE_STORE, kThreadsMW2>(la, hb.d_logexp, geo.kc, grid, lds, stream);
			else if (mw == 2)
				launch_nibble<2, gec::MODE_COMPARE, kThreadsMW2>(la, hb.d_logexp, geo.kc, grid, lds, stream);
			else if (mode == gec::MODE_STORE)
				launch_nibble<4, gec::MODE_STORE, kThreadsMW4>(la, hb.d_logexp, geo.kc, grid, lds, stream);
			else
				launch_nibble<4, gec::MODE_COMPARE, kThreadsMW4>(la, hb.d_logexp, geo.kc, grid, lds, stream);
			HIP_TRY(hipGetLastError());
		}
	}
	return GEC_OK;
}

// blake2sum of n messages.  group != 0: message i lives at d_base + (i / group)*group_stride + (i % group)*stride and
// its checksum goes to d_out + 32*((i / group)*out_group + i % group) -- e.g. only the data (or only the parity)
// shards of every stripe.  tree: the shard checksum (BLAKE2b tree mode, blake2b.hpp) instead of the plain hash;
// max_len = the longest message (sizes the leaf grid).
int blake2_dev(const gec_codec *c, size_t n, const uint8_t *d_base, const uint64_t *d_off, const uint64_t *d_len, size_t stride,
	       size_t len, uint8_t *d_out, hipStream_t stream, uint32_t group, size_t group_stride, uint32_t out_group, bool tree,
	       size_t max_len, uint64_t *d_state, uint64_t seg_begin_blk, uint64_t seg_end_blk)
{
	if (n == 0)
		return GEC_OK;
	if (n > 0xffffffffull)
		return fail(GEC_E_INVALID_ARG, "too many messages for one call");
	gec::Blake2Args a;
	a.state = d_state;
	a.seg_begin_blk = seg_begin_blk;
	a.seg_end_blk = seg_end_blk;
	a.base = d_base;
	a.off = d_off;
	a.len = d_len;
	a.stride = stride;
	a.uniform_len = len;
	a.out = d_out;
	a.n = (uint32_t)n;
	a.group = group;
	a.group_stride = group_stride;
	a.out_group = out_group;
	if (tree && c->sumkind == GEC_SHARDSUM_MLH64) {
		// shard checksum v3: a streaming pass for the leaf sums (8 bytes per 4 KiB), then one lane per shard for the roots
		const size_t longest = d_len ? max_len : len;
		const uint32_t nleaf_max = (uint32_t)std::max<size_t>(1, (longest + mlh::LEAF_BYTES - 1) / mlh::LEAF_BYTES);
		const uint64_t leaves = (uint64_t)n * nleaf_max;
		if ((leaves + 3) / 4 > 0x7fffffffull)
			return fail(GEC_E_INVALID_ARG, "too many leaves for one call");
		uint8_t *scratch = nullptr;
		int rc = leaf_scratch(c, stream, leaves * 8, &scratch);
		if (rc)
			return rc;
		hipLaunchKernelGGL(gec::mlh_leaves, dim3((unsigned)((leaves + 3) / 4)), dim3(256), 0, stream, a, nleaf_max, reinterpret_cast<uint64_t *>(scratch));
		HIP_TRY(hipGetLastError());
		return launch_mlh_roots(a, nleaf_max, reinterpret_cast<const uint64_t *>(scratch), nullptr, stream);
	}
	if (tree) {
		const size_t longest = d_len ? max_len : len;
		const uint32_t nleaf = (uint32_t)std::max<size_t>(1, (longest + gec::SHARDSUM_LEAF - 1) / gec::SHARDSUM_LEAF);
		const uint64_t lanes = (uint64_t)n * nleaf;
		if ((lanes + 63) / 64 > 0x7fffffffull)
			return fail(GEC_E_INVALID_ARG, "too many leaves for one call");
		uint8_t *scratch = nullptr;
		int rc = leaf_scratch(c, stream, lanes * 64, &scratch);
		if (rc)
			return rc;
		const dim3 lgrid((unsigned)((lanes + 63) / 64));
		// few leaves (a PutObject's blocks, a GetObject's): four lanes per leaf, like the plain hash below
		const int forced_leaf = test_route_blake2();
		const bool quad_tree = forced_leaf ? forced_leaf == 2 : lanes < 40000;
		if (quad_tree && (lanes + 15) / 16 > 0x7fffffffull)  // (the quad kernel indexes its messages in 32 bits)
			return fail(GEC_E_INVALID_ARG, "too many leaves for one call of the four-lane kernel");
		if (quad_tree)
			hipLaunchKernelGGL(gec::blake2b_batch_quad<gec::B2Q_LEAF>, dim3((unsigned)((lanes + 15) / 16)), dim3(64), 0, stream, a, nleaf, scratch);
		else
			hipLaunchKernelGGL(gec::shardsum_leaves<0>, lgrid, dim3(64), 0, stream, a, nleaf, scratch);
		HIP_TRY(hipGetLastError());
		if (quad_tree)
			hipLaunchKernelGGL(gec::blake2b_batch_quad<gec::B2Q_ROOT>, dim3((unsigned)((n + 15) / 16)), dim3(64), 0, stream, a, nleaf, scratch);
		else
			hipLaunchKernelGGL(gec::shardsum_roots, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, stream, a, nleaf, scratch);
		HIP_TRY(hipGetLastError());
		return GEC_OK;
	}
	// one lane per message is the faster kernel once there are enough messages to put a
	// wave on every SIMD (1024 SIMDs x 64 lanes); below that the quad kernel (4 lanes per
	// message, ~4x shorter chain) wins.  (gec_set_kernel_variant(2 | 3) forces one: the tests reach both on one input.)
	const int forced = test_route_blake2();
	const bool quad = d_state ? true : forced ? forced == 2 : n < 40000;  // segments: the quad kernel only
	if (quad)
		hipLaunchKernelGGL(gec::blake2b_batch_quad<gec::B2Q_PLAIN>, dim3((unsigned)((n + 15) / 16)), dim3(64), 0, stream, a, 0u, static_cast<uint8_t *>(nullptr));
	else
		hipLaunchKernelGGL(gec::blake2b_batch<0>, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, stream, a);
	HIP_TRY(hipGetLastError());
	return GEC_OK;
}

// One launch, a coefficient set per block (gec_reconstruct_batch_dev_ex): block b is rebuilt with plans[pat_of_block[b]] -- its
// valid shards in, its missing shards out, in place in the stripe (shard j of block b at d_base + b*stride + j*S).
int launch_apply_pat(const gec_codec *c, uint8_t *d_base, size_t stride, size_t S, size_t nblocks,
		     const std::vector<std::shared_ptr<const Plan>> &plans, const std::vector<uint16_t> &pat_of_block, hipStream_t stream)
{
	const int k = c->k;
	const HipBackend &hb = hip_of(c);
	if (nblocks == 0 || plans.empty())
		return GEC_OK;
	size_t max_rows = 0;
	for (const auto &pl : plans)
		max_rows = std::max(max_rows, pl->missing.size());
	if (max_rows == 0)
		return GEC_OK;
	if (max_rows > (size_t)gec::RMAX || plans.size() > 0xffff || nblocks > 0xffffffffull || S / 16 > 0x7fffffffull)
		return fail(GEC_E_INVALID_ARG, "per-block erasure patterns: at most 8 missing shards per block, 65535 patterns");
	const uint32_t kp = ((uint32_t)k + 3) & ~3u;
	const uint32_t ent = (4 * kp + 4 * gec::RMAX + 16 + (uint32_t)k * gec::RMAX + 15) & ~15u;
	const size_t pat_bytes = (nblocks * 2 + 15) & ~(size_t)15;
	std::vector<uint8_t> host(pat_bytes + plans.size() * ent, 0);
	std::memcpy(host.data(), pat_of_block.data(), nblocks * 2);
	for (size_t p = 0; p < plans.size(); ++p) {
		uint8_t *e = host.data() + pat_bytes + p * ent;
		uint32_t *in_off = reinterpret_cast<uint32_t *>(e), *out_off = in_off + kp;
		const Plan &pl = *plans[p];
		for (int t = 0; t < k; ++t)
			in_off[t] = (uint32_t)((size_t)pl.valid[t] * S / 16);
		for (size_t r = 0; r < pl.missing.size(); ++r)
			out_off[r] = (uint32_t)((size_t)pl.missing[r] * S / 16);
		out_off[gec::RMAX] = (uint32_t)pl.missing.size();
		uint8_t *coef = reinterpret_cast<uint8_t *>(out_off + gec::RMAX + 4);
		for (int t = 0; t < k; ++t)
			for (size_t r = 0; r < pl.missing.size(); ++r)
				coef[(size_t)t * gec::RMAX + r] = pl.rows.v[r * (size_t)k + t];
	}
	uint8_t *d_tab = nullptr;
	int rc = leaf_scratch(c, stream, host.size(), &d_tab);
	if (rc)
		return rc;
	HIP_TRY(hipMemcpyAsync(d_tab, host.data(), host.size(), hipMemcpyHostToDevice, stream));  // (pageable source: staged before the call returns)
	gec::ApplyArgs a;
	std::memset(&a, 0, sizeof(a));
	a.in = d_base;
	a.out = d_base;
	a.in_stride = a.out_stride = stride;
	a.cols = (uint32_t)(S / 16);
	a.k = (uint32_t)k;
	a.rows = (uint32_t)max_rows;
	a.pat = reinterpret_cast<const uint16_t *>(d_tab);
	a.pat_tab = d_tab + pat_bytes;
	a.pat_stride = ent;
	const Geometry geo = pick_geometry(k, (int)max_rows, false);
	const uint64_t tile_cols = (uint64_t)geo.threads;
	a.tiles_per_block = (uint32_t)((a.cols + tile_cols - 1) / tile_cols);
	const uint64_t blocks_per_launch = std::max<uint64_t>(1, std::min<uint64_t>(0xfffff000ull / a.cols, (0xffffffffull / geo.threads - 8) / a.tiles_per_block));
	for (uint64_t b0 = 0; b0 < nblocks; b0 += blocks_per_launch) {
		const uint64_t nb = std::min<uint64_t>(blocks_per_launch, nblocks - b0);
		gec::ApplyArgs la = a;
		la.in = la.out = d_base + b0 * stride;
		la.pat = a.pat + b0;
		la.nblocks = (uint32_t)nb;
		la.total_cols = (uint32_t)(nb * a.cols);
		const unsigned grid = (unsigned)((nb * a.tiles_per_block + 7) / 8 * 8);
		const int kc = geo.kc;
#define GEC_PAT(MW, KC, TPB) hipLaunchKernelGGL((gec::gf_apply_nibble_pat<MW, KC, true, TPB>), dim3(grid), dim3(TPB), geo.lds, stream, la, hb.d_logexp)
		if (geo.mw == 1) {
			switch (kc) {
			case 1: GEC_PAT(1, 1, 256); break;
			case 2: GEC_PAT(1, 2, 256); break;
			case 3: GEC_PAT(1, 3, 256); break;
			case 4: GEC_PAT(1, 4, 256); break;
			case 5: GEC_PAT(1, 5, 256); break;
			case 6: GEC_PAT(1, 6, 256); break;
			case 10: GEC_PAT(1, 10, 256); break;
			case 12: GEC_PAT(1, 12, 256); break;
			default: GEC_PAT(1, 16, 256); break;
			}
		} else if (kc == 10) {
			GEC_PAT(2, 10, 256);
		} else {
			switch (kc) {
			case 1: GEC_PAT(2, 1, 512); break;
			case 2: GEC_PAT(2, 2, 512); break;
			case 3: GEC_PAT(2, 3, 512); break;
			case 4: GEC_PAT(2, 4, 512); break;
			case 5: GEC_PAT(2, 5, 512); break;
			default: GEC_PAT(2, 6, 512); break;
			}
		}
#undef GEC_PAT
		HIP_TRY(hipGetLastError());
	}
	return GEC_OK;
}

int mlh_roots_dev(const gec_codec *c, size_t n, const uint64_t *lsum, uint32_t nleaf_max, const uint64_t *d_len, size_t len,
		  uint8_t *d_out, hipStream_t stream, const uint32_t *slot_map, uint32_t group, uint32_t out_group)
{
	if (n == 0)
		return GEC_OK;
	if (n > 0xffffffffull)
		return fail(GEC_E_INVALID_ARG, "too many shards for one call");
	gec::Blake2Args a;
	a.base = nullptr;
	a.off = nullptr;
	a.len = d_len;
	a.stride = 0;
	a.uniform_len = len;
	a.out = d_out;
	a.n = (uint32_t)n;
	a.group = group;
	a.group_stride = 0;
	a.out_group = out_group;
	return launch_mlh_roots(a, nleaf_max, lsum, slot_map, stream);
}

// Grid of a tile-walking kernel (gf_apply_ptrs, copy_table) on a CU-masked stream: no more workgroups than the
// stream's CUs hold at once; each walks tiles blockIdx.x, + gridDim.x, ...  A launch with more workgroups than fit
// occupies its queue's dispatcher until the last one is placed -- and kernels of other streams served by the same
// dispatcher wait for as long, whatever CUs THEY are confined to.  That is how a scrub's 600 us link kernel (800
// workgroups onto 8 CUs) made a PutObject's 140 us checksum kernel take 660 us in some process runs and not in others
// (which queues share a dispatcher is decided when they are created): tools/dispatch_probe, profiles/r03_qos.txt.
namespace {
unsigned resident_grid(const Staging &st, hipStream_t stream, uint32_t tiles, size_t lds_bytes = 0)
{
	// Every codec: a background codec's kernels must not stand in a PutObject's way, and the request path's own
	// kernels are meant to overlap too -- the read path's upload stages beside its checksum segments: with one
	// workgroup per tile, a get whose upload queue happened to share a dispatcher with its chain queue ran every
	// segment BEHIND the next stage's upload (17.7 instead of 15.8 ms for the same 512 blocks, process by process).
	// gec::RESIDENT_WGS workgroups per CU is what the kernels' __launch_bounds__ guarantees room for (the occupancy
	// query of the runtime does not count scalar registers and promised 8 for a kernel that fits 7 times: the
	// workgroups that did not fit started when the others were done, and held the dispatcher until then)
	// ... unless the launch's dynamic LDS allows fewer: wide codes (k up to PTR_KMAX: ~33 KiB of tables per workgroup)
	// fit 160 KiB of LDS only four times, and a grid sized for six would bring the hold-up back for exactly those
	uint64_t per_cu = (uint64_t)gec::RESIDENT_WGS;
	if (lds_bytes)
		per_cu = std::max<uint64_t>(1, std::min<uint64_t>(per_cu, (160u << 10) / lds_bytes));
	const uint64_t fit = (uint64_t)std::max(st.cus_of(stream), 1) * per_cu;
	return (unsigned)std::min<uint64_t>(tiles, fit);
}
}  // namespace

// out[b][r] = XOR_t coef[r][t] * in[b][t] over shards that stay in the caller's pinned memory (gf_apply_ptrs):
// in[b*k + t] / valid[b*k + t] name the k input shards of block b and how many of their S bytes exist,
// out[b*nout + r] the output rows.  The tables are written into the staging slot's pinned table area, which the
// kernel reads directly.  k <= PTR_KMAX.
int launch_apply_ptrs(const gec_codec *c, Staging &st, size_t nblocks, const uint8_t *const *in, const uint32_t *valid,
		      uint8_t *const *out, int nout, size_t S, const uint8_t *coef /* nout x k */, hipStream_t stream, uint8_t *d_mirror,
		      uint32_t *bad, size_t npat, const uint16_t *pat, const SumOut *sum)
{
	const size_t k = c->k;
	const HipBackend &hb = hip_of(c);
	if (nblocks == 0 || nout == 0)
		return GEC_OK;
	if (k > (size_t)gec::PTR_KMAX || S / 16 > 0xffffffffull)
		return fail(GEC_E_INVALID_ARG, "shape not supported by the pointer-table kernel");
	if (pat && (nout > gec::RMAX || npat == 0))
		return fail(GEC_E_INVALID_ARG, "per-block coefficient sets: one row group only");
	if (sum && (d_mirror || (size_t)sum->nleaf_max < (S + gec::SHARDSUM_LEAF - 1) / gec::SHARDSUM_LEAF))
		return fail(GEC_E_INVALID_ARG, "checksummed pointer-table launch: no mirror, leaf pitch >= the shard's leaves");
	const size_t in_bytes = nblocks * k * 8, valid_bytes = (nblocks * k * 4 + 7) / 8 * 8, out_bytes = nblocks * (size_t)nout * 8;
	const size_t coef_bytes = pat ? (npat * k * gec::RMAX + 7) / 8 * 8 : 0, pat_bytes = pat ? (nblocks * 2 + 7) / 8 * 8 : 0;
	const size_t need = (st.tab_used * sizeof(gec::CopyEntry) + in_bytes + valid_bytes + out_bytes + coef_bytes + pat_bytes) / sizeof(gec::CopyEntry) + 2;
	if (need > st.tab_cap)
		return fail(GEC_E_INVALID_ARG, "pointer table overflow");
	uint8_t *base = reinterpret_cast<uint8_t *>(st.h_tab + st.tab_used);
	const uint8_t **t_in = reinterpret_cast<const uint8_t **>(base);
	uint32_t *t_valid = reinterpret_cast<uint32_t *>(base + in_bytes);
	uint8_t **t_out = reinterpret_cast<uint8_t **>(base + in_bytes + valid_bytes);
	st.tab_used = need;
	std::memcpy(t_in, in, in_bytes);
	std::memcpy(t_valid, valid, nblocks * k * 4);
	gec::PtrApplyArgs a;
	std::memset(&a, 0, sizeof(a));
	if (pat) {
		uint8_t *t_coef = base + in_bytes + valid_bytes + out_bytes;
		uint16_t *t_pat = reinterpret_cast<uint16_t *>(t_coef + coef_bytes);
		std::memset(t_coef, 0, coef_bytes);
		for (size_t p = 0; p < npat; ++p)
			for (size_t t = 0; t < k; ++t)
				for (int r = 0; r < nout; ++r)
					t_coef[(p * k + t) * gec::RMAX + r] = coef[(p * (size_t)nout + r) * k + t];
		std::memcpy(t_pat, pat, nblocks * 2);
		a.coef_tab = t_coef;
		a.pat = t_pat;
	}
	a.cols = (uint32_t)(S / 16);
	a.k = (uint32_t)k;
	const unsigned gx = (a.cols + 255) / 256;
	if ((uint64_t)gx * nblocks > 0xffffffffull)
		return fail(GEC_E_INVALID_ARG, "too many tiles for one launch");
	a.tiles_x = gx;
	a.tiles_total = (uint32_t)(gx * nblocks);
	a.in = t_in;
	a.in_valid = t_valid;
	a.bad = bad;
	link_role_of(st, 0, &a.link_busy, &a.link_role, &a.link_wait_ticks);
	int rows = 0;
	size_t out_done = 0;  // entries of t_out consumed by earlier row groups
	for (int r0 = 0; r0 < nout; r0 += rows) {
		rows = std::min(gec::RMAX, nout - r0);
		a.rows = (uint32_t)rows;
		for (int r = 0; r < gec::RMAX && !pat; ++r)
			for (size_t t = 0; t < k; ++t)
				a.coef[t][r] = r < rows ? coef[(size_t)(r0 + r) * k + t] : 0;
		uint8_t **grp = t_out + out_done;  // [nblocks][rows] for this group
		for (size_t b = 0; b < nblocks; ++b)
			for (int r = 0; r < rows; ++r)
				grp[b * rows + r] = out[b * nout + r0 + r];
		out_done += nblocks * rows;
		const int mw = rows <= 4 ? 1 : 2;
		size_t lds = k * 32 * 4 * mw + 768 + k * gec::RMAX;
		if (sum) {
			a.lsum = sum->lsum;
			a.sum_nleaf_max = sum->nleaf_max;
			a.sum_slots_total = sum->slots_total;
			a.sum_inputs = (r0 == 0 && sum->inputs) ? 1u : 0u;
			a.sum_slot0 = sum->slot0 + (a.sum_inputs ? 0u : (sum->inputs ? (uint32_t)k : 0u) + (uint32_t)r0);
			lds = ((lds + 15) & ~(size_t)15) + gec::mlh_lds_bytes(16, 4, (int)((a.sum_inputs ? k : 0) + rows));
		}
		a.mirror_stride = (k + (size_t)nout) * S;
		a.mirror_row0 = (k + (size_t)r0) * S;
		a.mirror_inputs = r0 == 0;
		a.out = grp;
		a.mirror = d_mirror;
		using Kern = void (*)(const gec::PtrApplyArgs, const gec::LogExp *);
		Kern kern;
		if (sum && bad)
			kern = mw == 1 ? (Kern)gec::gf_apply_ptrs<1, 5, false, true, true> : (Kern)gec::gf_apply_ptrs<2, 5, false, true, true>;
		else if (sum)
			kern = mw == 1 ? (Kern)gec::gf_apply_ptrs<1, 5, false, false, true> : (Kern)gec::gf_apply_ptrs<2, 5, false, false, true>;
		else if (bad && d_mirror)
			kern = mw == 1 ? (Kern)gec::gf_apply_ptrs<1, 5, true, true> : (Kern)gec::gf_apply_ptrs<2, 5, true, true>;
		else if (bad)
			kern = mw == 1 ? (Kern)gec::gf_apply_ptrs<1, 5, false, true> : (Kern)gec::gf_apply_ptrs<2, 5, false, true>;
		else if (d_mirror)
			kern = mw == 1 ? (Kern)gec::gf_apply_ptrs<1, 5, true> : (Kern)gec::gf_apply_ptrs<2, 5, true>;
		else
			kern = mw == 1 ? (Kern)gec::gf_apply_ptrs<1, 5, false> : (Kern)gec::gf_apply_ptrs<2, 5, false>;
		const unsigned grid = resident_grid(st, stream, a.tiles_total, lds);
		// a background codec's rows into HOST memory are paced (GEC_BG_HOME_RATE_GBPS): `grid` resident workgroups, each
		// writing rows * 4 KiB per tile, start a tile every grid * rows * 4096 / rate nanoseconds
		a.pace_ticks = 0;
		if (st.qos.background && !bad && env().bg_home_rate_gbps > 0 && nblocks && out[0] && pinned().contains(out[0], S))
			a.pace_ticks = (uint32_t)std::min<uint64_t>((uint64_t)grid * rows * 4096ull / env().bg_home_rate_gbps / 10, 1u << 24);
		hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, a, hb.d_logexp);
		HIP_TRY(hipGetLastError());
	}
	return GEC_OK;
}

// The pointer-table kernel over DEVICE-resident tables (gec_group_peer_decode: the inputs are byte ranges of shards that live in
// OTHER devices' memory, read over xGMI): in[b*k + t] / out[b*nout + r] are host arrays of device-addressable pointers, every
// input has `cols` 16-byte columns; they are laid down in d_scratch (>= ptrs_dev_scratch_bytes) behind whatever `stream` holds.
size_t ptrs_dev_scratch_bytes(size_t nblocks, size_t k, int nout) { return nblocks * k * 8 + ((nblocks * k * 4 + 15) & ~(size_t)15) + nblocks * (size_t)nout * 8; }

// `in` == NULL: d_scratch already holds the tables of an earlier call with the same nblocks / nout / pointers (the caller keys
// them: a steady-state peer decode is then the launch alone -- no host table, no upload).
int launch_apply_ptrs_dev(const gec_codec *c, uint8_t *d_scratch, size_t nblocks, const uint8_t *const *in, uint8_t *const *out, int nout,
			  uint32_t cols, const uint8_t *coef /* nout x k */, hipStream_t stream)
{
	const size_t k = c->k;
	const HipBackend &hb = hip_of(c);
	if (nblocks == 0 || nout == 0 || cols == 0)
		return GEC_OK;
	if (k > (size_t)gec::PTR_KMAX)
		return fail(GEC_E_INVALID_ARG, "shape not supported by the pointer-table kernel (k > 128)");
	const unsigned gx = (cols + 255) / 256;
	if ((uint64_t)gx * nblocks > 0xffffffffull)
		return fail(GEC_E_INVALID_ARG, "too many tiles for one launch");
	const size_t in_bytes = nblocks * k * 8, valid_bytes = (nblocks * k * 4 + 15) & ~(size_t)15, out_bytes = nblocks * (size_t)nout * 8;
	size_t done = 0;
	if (in) {
		std::vector<uint8_t> host(in_bytes + valid_bytes + out_bytes);
		std::memcpy(host.data(), in, in_bytes);
		uint32_t *v = reinterpret_cast<uint32_t *>(host.data() + in_bytes);
		for (size_t i = 0; i < nblocks * k; ++i)
			v[i] = cols * 16u;
		// outputs: row groups of at most RMAX rows, each group's table [nblocks][rows] contiguous
		uint8_t **o = reinterpret_cast<uint8_t **>(host.data() + in_bytes + valid_bytes);
		for (int r0 = 0; r0 < nout; r0 += gec::RMAX) {
			const int rows = std::min(gec::RMAX, nout - r0);
			for (size_t b = 0; b < nblocks; ++b)
				for (int r = 0; r < rows; ++r)
					o[done + b * rows + r] = out[b * nout + r0 + r];
			done += nblocks * rows;
		}
		HIP_TRY(hipMemcpyAsync(d_scratch, host.data(), host.size(), hipMemcpyHostToDevice, stream));  // (pageable source: staged before the call returns)
	}
	gec::PtrApplyArgs a;
	std::memset(&a, 0, sizeof(a));
	a.in = reinterpret_cast<const uint8_t *const *>(d_scratch);
	a.in_valid = reinterpret_cast<const uint32_t *>(d_scratch + in_bytes);
	a.cols = cols;
	a.k = (uint32_t)k;
	a.tiles_x = gx;
	a.tiles_total = (uint32_t)(gx * nblocks);
	a.link_role = gec::LINK_NONE;
	done = 0;
	for (int r0 = 0; r0 < nout; r0 += gec::RMAX) {
		const int rows = std::min(gec::RMAX, nout - r0);
		a.rows = (uint32_t)rows;
		for (int r = 0; r < gec::RMAX; ++r)
			for (size_t t = 0; t < k; ++t)
				a.coef[t][r] = r < rows ? coef[(size_t)(r0 + r) * k + t] : 0;
		a.out = reinterpret_cast<uint8_t *const *>(d_scratch + in_bytes + valid_bytes) + done;
		done += nblocks * rows;
		const int mw = rows <= 4 ? 1 : 2;
		const size_t lds = k * 32 * 4 * mw + 768 + k * gec::RMAX;
		const uint64_t per_cu = std::max<uint64_t>(1, std::min<uint64_t>(gec::RESIDENT_WGS, (160u << 10) / lds));
		// (a grid of what is resident at once: 2x, 4x and one workgroup per tile were 3 - 7 % slower at world 1, profiles/r06_striped.txt)
		const unsigned grid = (unsigned)std::min<uint64_t>(a.tiles_total, (uint64_t)hb.num_cu * per_cu);
		using Kern = void (*)(const gec::PtrApplyArgs, const gec::LogExp *);
		const Kern kern = mw == 1 ? (Kern)gec::gf_apply_ptrs<1, 5, false> : (Kern)gec::gf_apply_ptrs<2, 5, false>;
		hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, a, hb.d_logexp);
		HIP_TRY(hipGetLastError());
	}
	return GEC_OK;
}

// ---- the fused small-trip kernel (fused.hpp)
namespace {
size_t fused_lds_bytes(size_t k, size_t nh, int mw)
{
	const size_t leaves_off = (k * 32 * 4 * mw + 768 + k * gec::RMAX + 15) & ~(size_t)15;
	return leaves_off + nh * gec::FUSED_LEAF_PITCH + 16;
}
// LDS one workgroup may ask for on this device (gfx950: 160 KiB per CU, all of it available to one workgroup)
size_t max_lds_per_workgroup(int device)
{
	static std::mutex mu;
	static std::map<int, size_t> seen;
	std::lock_guard<std::mutex> g(mu);
	auto it = seen.find(device);
	if (it != seen.end())
		return it->second;
	int v = 0;
	if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, device) != hipSuccess || v <= 0)
		v = 64 << 10;
	(void)hipGetLastError();
	return seen[device] = (size_t)v;
}
}  // namespace

bool fused_fits(const gec_codec *c, size_t nblocks, size_t S, int nout, bool hash_rows)
{
	const size_t k = c->k, nh = k + (hash_rows ? (size_t)nout : 0);
	if (c->sumkind != GEC_SHARDSUM_BLAKE2B_TREE)  // (the one-launch kernel hashes BLAKE2b leaves out of LDS: checksum v2 only)
		return false;
	if (g_variant.load(std::memory_order_relaxed) == 4 || nblocks == 0 || k > (size_t)gec::PTR_KMAX || nout > gec::RMAX || nh > (size_t)gec::FUSED_MAX_LEAVES)
		return false;
	const size_t tiles_x = (S / 16 + 255) / 256;
	if (S % 16 || tiles_x * nblocks > 0xffffffffull)
		return false;
	// small: the trip's leaves (what the one-lane-per-leaf kernels need tens of thousands of to fill the chip)
	if (tiles_x * nblocks * nh >= (hash_rows ? env().fused_max_leaves : env().fused_get_max_leaves))
		return false;
	return fused_lds_bytes(k, nh, nout <= 4 ? 1 : 2) <= max_lds_per_workgroup(c->device);
}

int launch_fused(const gec_codec *c, Staging &st, size_t nblocks, const uint8_t *const *in, const uint32_t *valid, uint8_t *const *out,
		 int nout, size_t S, const uint8_t *coef_sets, size_t npat, const uint16_t *pat, bool hash_rows, uint8_t *sums,
		 hipStream_t stream)
{
	const size_t k = c->k, nh = k + (hash_rows ? (size_t)nout : 0);
	const HipBackend &hb = hip_of(c);
	if (!fused_fits(c, nblocks, S, nout, hash_rows))
		return fail(GEC_E_INVALID_ARG, "fused: shape does not fit the one-launch kernel");
	if (npat == 0)
		npat = 1;
	// the tables ride in the slot's pinned table area, which the kernel reads directly: [in][valid][out][coef sets][pat]
	const size_t in_bytes = nblocks * k * 8, valid_bytes = (nblocks * k * 4 + 7) / 8 * 8, out_bytes = nblocks * (size_t)nout * 8;
	const size_t coef_bytes = (npat * k * gec::RMAX + 7) / 8 * 8, pat_bytes = pat ? (nblocks * 2 + 7) / 8 * 8 : 0;
	const size_t need = (st.tab_used * sizeof(gec::CopyEntry) + in_bytes + valid_bytes + out_bytes + coef_bytes + pat_bytes) / sizeof(gec::CopyEntry) + 2;
	if (need > st.tab_cap)
		return fail(GEC_E_INVALID_ARG, "pointer table overflow");
	uint8_t *base = reinterpret_cast<uint8_t *>(st.h_tab + st.tab_used);
	st.tab_used = need;
	const uint8_t **t_in = reinterpret_cast<const uint8_t **>(base);
	uint32_t *t_valid = reinterpret_cast<uint32_t *>(base + in_bytes);
	uint8_t **t_out = reinterpret_cast<uint8_t **>(base + in_bytes + valid_bytes);
	uint8_t *t_coef = base + in_bytes + valid_bytes + out_bytes;
	uint16_t *t_pat = reinterpret_cast<uint16_t *>(t_coef + coef_bytes);
	std::memcpy(t_in, in, in_bytes);
	std::memcpy(t_valid, valid, nblocks * k * 4);
	if (nout)
		std::memcpy(t_out, out, out_bytes);
	std::memset(t_coef, 0, coef_bytes);
	for (size_t p = 0; p < npat && nout; ++p)
		for (size_t t = 0; t < k; ++t)
			for (int r = 0; r < nout; ++r)
				t_coef[(p * k + t) * gec::RMAX + r] = coef_sets[(p * (size_t)nout + r) * k + t];
	if (pat)
		std::memcpy(t_pat, pat, nblocks * 2);
	gec::FusedArgs a;
	std::memset(&a, 0, sizeof(a));
	a.in = t_in;
	a.in_valid = t_valid;
	a.out = nout ? t_out : nullptr;
	a.cols = (uint32_t)(S / 16);
	a.k = (uint32_t)k;
	a.rows = (uint32_t)nout;
	a.tiles_x = (a.cols + 255) / 256;
	a.tiles_total = (uint32_t)(a.tiles_x * nblocks);
	a.hash_rows = hash_rows ? 1u : 0u;
	a.coef_tab = t_coef;
	a.pat = pat ? t_pat : nullptr;
	a.sums = sums;
	int rc = leaf_scratch(c, stream, nblocks * nh * a.tiles_x * 64, &a.leafdig);
	if (!rc)
		rc = done_counters(c, stream, nblocks, &a.done);
	if (rc)
		return rc;
	link_role_of(st, 0, &a.link_busy, &a.link_role, &a.link_wait_ticks);
	const int mw = nout <= 4 ? 1 : 2;
	const size_t lds = fused_lds_bytes(k, nh, mw);
	using Kern = void (*)(const gec::FusedArgs, const gec::LogExp *);
	// k <= 10 with 4-byte entries: all of a tile's loads in ONE batch (a small trip is a few link round trips long)
	const Kern kern = mw == 1 ? (k <= 10 ? (Kern)gec::gf_ptrs_hash<1, 10> : (Kern)gec::gf_ptrs_hash<1, 5>) : (Kern)gec::gf_ptrs_hash<2, 5>;
	if (lds > (48u << 10)) {  // beyond the default limit: opt in, once per kernel and device
		static std::mutex mu;
		static std::set<std::pair<const void *, int>> opted;
		std::lock_guard<std::mutex> g(mu);
		if (opted.insert({reinterpret_cast<const void *>(kern), c->device}).second)
			HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
						    (int)max_lds_per_workgroup(c->device)));
	}
	// a grid that fits: what LDS lets a CU hold (the dispatcher hold-up of resident_grid applies here too)
	const size_t per_cu = std::max<size_t>(1, std::min<size_t>(2, (160u << 10) / lds));
	const uint64_t fit = (uint64_t)std::max(st.cus_of(stream), 1) * per_cu;
	const unsigned grid = (unsigned)std::min<uint64_t>(a.tiles_total, fit);
	hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, a, hb.d_logexp);
	HIP_TRY(hipGetLastError());
	return GEC_OK;
}

int launch_copy_table(Staging &st, const std::vector<gec::CopyEntry> &ents, hipStream_t stream, unsigned max_wgs, unsigned pace_ns)
{
	if (ents.empty())
		return GEC_OK;
	if (st.tab_used + ents.size() > st.tab_cap)
		return fail(GEC_E_INVALID_ARG, "copy table overflow");
	gec::CopyEntry *tab = st.h_tab + st.tab_used;
	uint64_t maxb = 0;
	for (size_t i = 0; i < ents.size(); ++i) {
		tab[i] = ents[i];
		maxb = std::max<uint64_t>(maxb, ents[i].bytes);
	}
	st.tab_used += ents.size();
	const uint64_t gx = (maxb >> 4) / 1024 + 1;
	if (gx * ents.size() > 0xffffffffull)
		return fail(GEC_E_INVALID_ARG, "too many tiles for one launch");
	const uint32_t total = (uint32_t)(gx * ents.size());
	unsigned grid = resident_grid(st, stream, total);
	if (max_wgs > 0)
		grid = std::min(grid, max_wgs);
	unsigned pace = max_wgs > 0 ? pace_ns : 0;
	// a background codec's copies INTO host memory keep to GEC_BG_HOME_RATE_GBPS like its link kernels' rows do
	if (pace == 0 && st.qos.background && env().bg_home_rate_gbps > 0 && pinned().contains(ents[0].dst, 1))
		pace = (unsigned)std::min<uint64_t>((uint64_t)grid * 16384ull / env().bg_home_rate_gbps, 10000000ull);
	uint32_t *busy = nullptr, role = 0, wait_ticks = 0;
	link_role_of(st, pace, &busy, &role, &wait_ticks);
	hipLaunchKernelGGL(gec::copy_table, dim3(grid), dim3(256), 0, stream, tab, (uint32_t)gx, total, pace / 10, busy, role, wait_ticks);
	HIP_TRY(hipGetLastError());
	return GEC_OK;
}

int launch_clear_flags(uint32_t *d_bad, size_t n, hipStream_t stream)
{
	hipLaunchKernelGGL(gec::clear_flags, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, d_bad, (uint32_t)n);
	HIP_TRY(hipGetLastError());
	return GEC_OK;
}

namespace {
unsigned copy_grid(size_t items) { return (unsigned)std::max<size_t>(1, std::min<size_t>((items + 255) / 256, 1u << 16)); }
}

int launch_range_pack(const gec::RangeArgs &a, size_t items, hipStream_t stream)
{
	hipLaunchKernelGGL(gec::range_pack, dim3(copy_grid(items)), dim3(256), 0, stream, a);
	HIP_TRY(hipGetLastError());
	return GEC_OK;
}

int launch_range_unpack(const gec::RangeArgs &a, size_t items, hipStream_t stream)
{
	hipLaunchKernelGGL(gec::range_unpack, dim3(copy_grid(items)), dim3(256), 0, stream, a);
	HIP_TRY(hipGetLastError());
	return GEC_OK;
}

int launch_a2a_pack(const gec::A2aArgs &a, size_t items, hipStream_t stream)
{
	hipLaunchKernelGGL(gec::a2a_pack, dim3(copy_grid(items)), dim3(256), 0, stream, a);
	HIP_TRY(hipGetLastError());
	return GEC_OK;
}

int launch_rebuilt_unpack(const gec::RebuiltArgs &a, size_t items, hipStream_t stream)
{
	hipLaunchKernelGGL(gec::rebuilt_unpack, dim3(copy_grid(items)), dim3(256), 0, stream, a);
	HIP_TRY(hipGetLastError());
	return GEC_OK;
}

}  // namespace gecimpl

using namespace gecimpl;

extern "C" {

int gec_launch_geometry(int k, int rows_left, int *rows, int *entry_bytes, int *loads_per_batch, int *threads, size_t *lds_bytes)
{
	if (k < 1 || k > GEC_MAX_SHARDS - 1 || rows_left < 1)
		return fail(GEC_E_INVALID_ARG, "need 1 <= k <= 255 and rows_left >= 1");
	const Geometry g = pick_geometry(k, rows_left, true);
	if (rows)
		*rows = g.rows;
	if (entry_bytes)
		*entry_bytes = 4 * g.mw;
	if (loads_per_batch)
		*loads_per_batch = g.kc;
	if (threads)
		*threads = g.threads;
	if (lds_bytes)
		*lds_bytes = g.lds;
	return GEC_OK;
}

int gec_set_kernel_variant(int variant)
{
	if (variant < 0 || variant > 5)
		return fail(GEC_E_INVALID_ARG, "unknown kernel variant");
	g_variant.store(variant);
	return GEC_OK;
}

int gec_get_kernel_variant(void) { return g_variant.load(); }

}  // extern "C"
