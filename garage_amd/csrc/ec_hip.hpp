// ec_hip.hpp -- what the translation units of the HIP backend share (see ec_internal.hpp for the file map).
// Host-side declarations only: the device code lives in kernels.hpp / blake2b.hpp and is compiled exactly once,
// in ec_hip_launch.hip, which exports the launch_* / blake2_dev wrappers declared at the bottom of this file.
#pragma once

#include <exception>

#include "ec_internal.hpp"

#include <hip/hip_runtime_api.h>  // host API only: the device code is in ec_hip_launch.hip

#include <map>

#include "ec_env.hpp"
#include "kernel_args.hpp"
#include "numa.hpp"

namespace gecimpl {

#define HIP_TRY(expr)                                                                  \
	do {                                                                           \
		hipError_t e_ = (expr);                                                \
		if (e_ != hipSuccess)                                                  \
			return ::gecimpl::fail(e_ == hipErrorOutOfMemory ? GEC_E_NOMEM : GEC_E_DEVICE, \
				    std::string(#expr) + ": " + hipGetErrorString(e_)); \
	} while (0)

// Restores the calling thread's current device on scope exit (torch and other
// callers keep their own notion of "current device").
struct DeviceGuard {
	int prev = -1;
	bool ok = false;
	explicit DeviceGuard(int dev)
	{
		if (hipGetDevice(&prev) != hipSuccess)
			prev = -1;
		ok = (prev == dev) || hipSetDevice(dev) == hipSuccess;
	}
	~DeviceGuard()
	{
		if (prev >= 0)
			(void)hipSetDevice(prev);
	}
};

// Pinned host ranges the caller told us about (gec_host_alloc / gec_host_register): blocks and
// output buffers that lie inside one go over PCIe by DMA straight from / to the caller's memory,
// without the pageable -> pinned staging copy (which costs a third of the PCIe-inclusive rate).
class PinnedRanges {
public:
	// dev_delta: what to add to a host address inside the range to get the address the GPU must use
	// (0 for hipHostMalloc; hipHostRegister may map the pages at a different device address).
	// plain: ordinary heap memory handed out by gec_host_alloc on a host without a device -- tracked so that
	// gec_host_free can release it, never treated as device-addressable.
	void add(const void *p, size_t n, bool owned, intptr_t dev_delta = 0, bool plain = false);
	// returns true and whether the library allocated it / whether it is plain heap memory
	bool remove(const void *p, bool &owned, bool &plain);
	bool contains(const void *p, size_t n, intptr_t *dev_delta = nullptr) const;
	// the address a kernel uses for host address p (p must lie in a registered range)
	template <class T>
	T *dev(T *p) const
	{
		intptr_t d = 0;
		contains(p, 1, &d);
		return reinterpret_cast<T *>(reinterpret_cast<intptr_t>(p) + d);
	}

private:
	struct R {
		size_t len;
		bool owned, plain;
		intptr_t dev_delta;
	};
	mutable std::mutex mu_;
	std::map<uintptr_t, R> ranges_;
};

PinnedRanges &pinned();

// Quality-of-service class of the work a codec enqueues (gec_codec_background, include/garage_ec.h): what the
// staging slots of a background codec do differently is decided here, once.
struct QosPolicy {
	bool background = false;
	int stream_priority = 0;   // hipStreamCreateWithPriority value (background: the device's lowest)
	int compute_cus = 0;       // > 0: the codec's streams are confined to this many CUs (of num_cu)
	int compute_cus_plan = 0;  // GEC_BG_CUS as every codec of the process sees it: the CUs set aside for the background class
	int num_cu = 0;
	int device = 0;
	int numa_node = -1;        // the memory node the codec keeps its host side on (-1: unknown / GEC_NUMA=0: nothing is placed)
};

// hipHostMalloc with the pages on `node` (the codec's: numa.hpp).  The runtime puts pinned memory near the calling thread's CURRENT
// device unless hipHostMallocNumaUser is given, and then follows the thread's memory policy: so the placement is said out
// loud -- a lane of device 5 that allocates from a thread whose current device is 0 gets its pages near device 5 all the same.
// node < 0 (or a kernel that refuses the policy call): the runtime's own choice.
inline hipError_t host_malloc_on_node(void **p, size_t bytes, unsigned flags, int node)
{
	if (node < 0)
		return hipHostMalloc(p, bytes, flags);
	gecnuma::ScopedBind bind(node);
	return hipHostMalloc(p, bytes, bind.ok() ? (flags | hipHostMallocNumaUser) : flags);
}

// Staging resources for the host-pointer entry points (one per in-flight call).
struct Staging {
	hipStream_t stream = nullptr;
	// fork/join partner of `stream`: the blake2 of the data shards runs here, beside the RS kernel
	hipStream_t stream2 = nullptr;
	hipEvent_t ev_fork = nullptr, ev_join = nullptr;
	hipEvent_t ev_in = nullptr, ev_out = nullptr;  // "this slot's copy-in / copy-out kernel is done" (PipeChain)
	uint8_t *h_buf = nullptr, *d_buf = nullptr;
	size_t cap = 0;
	uint32_t *d_bad = nullptr, *h_bad = nullptr;
	size_t bad_cap = 0;
	// copy tables of the zero-copy path (pinned, read by the copy_table kernel straight from host memory)
	gec::CopyEntry *h_tab = nullptr;
	size_t tab_cap = 0, tab_used = 0;
	// whole-batch device buffer of the read path (gec_decode_verify_batch): grow-only
	uint8_t *d_big = nullptr;
	size_t big_cap = 0;
	// the read path's upload stages: "stage s is on the device" events, and a third stream so that the shard
	// checksums do not queue behind the block checksums' serial chains
	static constexpr int kMaxSeg = 16;
	hipStream_t stream3 = nullptr;
	hipEvent_t ev_seg[kMaxSeg] = {};
	// CU-masked pair for the staged upload: the copy kernels' host reads sit in the memory pipeline of the CUs they
	// run on for microseconds each, and a checksum chain sharing such a CU crawls (7x slower, measured); so the
	// upload gets a few CUs of its own (the link needs very little in flight) and the chains the rest.
	hipStream_t stream_up = nullptr, stream_chain = nullptr;
	// ... and a few more for kernels that WRITE host memory while uploads are still running (the rebuilt shards of
	// the read path on their way home): sixteen CUs read host memory at the link's rate but cannot also write it
	hipStream_t stream_down = nullptr;
	hipEvent_t ev_dec[kMaxSeg] = {};  // "the decodes of stage s are done"
	int cus_up = 0, cus_chain = 0, cus_down = 0;  // CUs of the three masked streams (0 = that stream has no mask)
	bool seg_split = false;  // the masks were made for "a background class exists on this device"
	QosPolicy qos;  // set by the lease from the codec's class before anything is created

	int ensure_segments(int num_cu);
	int ensure_big(size_t bytes);
	int ensure_tab(size_t entries);
	int ensure(size_t bytes, size_t nbad);
	void release();
	// how many CUs kernels launched on `s` (a stream of this slot) may run on
	int cus_of(hipStream_t s) const;

private:
	int make_stream(hipStream_t *s);
};

// Per-device count of foreground host-pointer calls in flight: a background codec's chunk loop looks at it
// before it queues its next chunk (QosGate::yield_to_foreground) so that a PutObject's encode finds the link,
// the copy threads and the CUs free within one background chunk.
class QosGate {
public:
	static QosGate &of(int device);
	void enter() { fg_.fetch_add(1, std::memory_order_acq_rel); }
	void leave();
	// blocks while foreground calls are in flight on this device, at most max_wait_us; returns the time waited
	uint64_t yield_to_foreground(unsigned max_wait_us);
	uint64_t yields() const { return yields_.load(); }

private:
	std::atomic<int> fg_{0};
	std::atomic<uint64_t> yields_{0};
	std::mutex mu_;
	std::condition_variable cv_;
};

// The HIP backend of one codec: device state + the Backend entry points.
struct HipBackend : Backend {
	const gec_codec *c = nullptr;
	int device = 0;
	int num_cu = 256;
	// where the host side of this codec lives (resolved once, at creation: the device's PCI address -> sysfs): the node its copy
	// threads run on and its pinned memory is bound to.  -1 / empty: unknown, one-node box, or GEC_NUMA=0.
	int numa_node_ = -1, numa_node_of_device = -1;
	std::vector<int> numa_cpus_;
	gec::LogExp *d_logexp = nullptr;
	QosPolicy qos;

	mutable std::mutex pool_mu;
	mutable std::vector<Staging> pool;
	// host-pointer calls in flight on this codec (GEC_MAX_CALLS): every call owns one to three staging slots, every slot
	// three or four device queues, and past a few dozen queues the device spends its time switching between them
	mutable std::mutex calls_mu;
	mutable std::condition_variable calls_cv;
	mutable unsigned calls_in_flight = 0;

	// leaf-digest scratch of the tree-mode shard checksums, one per stream that ever hashed (work on one
	// stream is ordered, so reuse on the same stream is safe; grow-only)
	struct LeafScratch {
		uint8_t *p = nullptr;
		size_t cap = 0;
	};
	mutable std::mutex leaf_mu;
	mutable std::map<hipStream_t, LeafScratch> leaf_scratch;
	// per-block "tiles finished" counters of the fused small-trip kernel (fused.hpp), one array per stream like the leaf
	// scratch: zeroed when allocated, left all zero by every launch
	mutable std::map<hipStream_t, LeafScratch> done_counters;

	// One copy pool per codec = per device: a process that drives several GPUs (one codec each)
	// must not funnel all their staging copies through one set of threads.  Created on the
	// first host-pointer call; device-API-only users never start the threads.
	mutable std::once_flag copy_once;
	mutable std::unique_ptr<ForkJoinPool> copy_threads;
	ForkJoinPool &copy_pool() const;

	~HipBackend() override;
	int numa_node() const override { return numa_node_; }
	const std::vector<int> *numa_cpus() const override { return numa_cpus_.empty() ? nullptr : &numa_cpus_; }
	void *host_alloc(size_t bytes) const override;

	int encode_batch(size_t nblocks, const uint8_t *const *blocks, const size_t *block_len, size_t S, uint8_t *const *parity,
			 uint8_t *shard_sums) override;
	int verify_batch(size_t nblocks, const uint8_t *const *shards, size_t S, uint8_t *ok) override;
	int verify_hash_batch(size_t nblocks, const uint8_t *const *shards, size_t S, uint8_t *ok, uint8_t *shard_sums) override;
	int reconstruct_batch(size_t nblocks, const uint8_t *const *shards, uint8_t *const *out, size_t S, int data_only, uint8_t *in_sums,
			      uint8_t *out_sums) override;
	int decode_verify_batch(size_t nblocks, const uint8_t *const *shards, size_t S, const size_t *block_len, uint8_t *const *rebuilt,
				uint8_t *shard_sums, uint8_t *block_sums) override;
	int hash_batch(size_t n, const uint8_t *const *msgs, const size_t *lens, uint8_t *out, bool tree) override;

	int encode_batch_dev(size_t nblocks, const void *d_data, size_t data_stride, size_t S, void *d_parity, size_t parity_stride,
			     void *hip_stream) override;
	int verify_batch_dev(size_t nblocks, const void *d_stripes, size_t stride, size_t S, uint32_t *d_bad, void *hip_stream) override;
	int reconstruct_dev(size_t nblocks, void *d_base, size_t block_stride, const size_t *shard_off, size_t S, const uint8_t *present,
			    int data_only, size_t byte_off, size_t byte_len, void *hip_stream) override;
	int reconstruct_dev_ex(size_t nblocks, void *d_stripes, size_t stride, size_t S, const uint8_t *present, int data_only,
			       void *hip_stream) override;
	int hash_batch_dev(size_t n, const void *d_base, size_t stride, size_t len, void *d_out, void *hip_stream, bool tree) override;
	int encode_hash_batch_dev(size_t nblocks, void *d_stripes, size_t stride, size_t S, void *d_sums, void *hip_stream) override;
};

inline HipBackend &hip_of(const gec_codec *c) { return *static_cast<HipBackend *>(c->be.get()); }

// A staging slot on loan from the codec's pool for the duration of one call.
struct StagingLease {
	const gec_codec *c;
	Staging st;
	int unwinding_at_entry = 0;  // std::uncaught_exceptions() when the slot was taken (see the destructor)
	explicit StagingLease(const gec_codec *cc);
	~StagingLease();
};

// a BACKGROUND-class codec was created on / removed from `device` (the CU partition between the classes follows)
void background_codec_born(int device);
void background_codec_gone(int device);

// RAII: a foreground host-pointer call is in flight on this codec's device (no-op for a background codec)
// What every host-pointer entry point of the HIP backend holds for its duration: a call permit of its codec (at most
// GEC_MAX_CALLS at a time; a call made from inside another one of the same thread rides on the outer permit) and, for a
// foreground codec, a count in the device's QosGate.
struct ForegroundScope {
	QosGate *gate = nullptr;
	const HipBackend *permit_of = nullptr;
	explicit ForegroundScope(const gec_codec *c);
	~ForegroundScope();
};
// a background codec calls this between chunks; a foreground codec's call returns at once
void background_yield(const gec_codec *c);

constexpr size_t kChunkBytes = 16ull << 20;  // staging chunk: small enough to overlap, big enough to fill the GPU

// blocks per staging chunk of about `target` bytes (callers pass kChunkBytes or a multiple):
// staging memory stays bounded whatever the batch size.
inline size_t chunk_blocks(size_t bytes_per_block, size_t nblocks, size_t target)
{
	size_t n = std::max<size_t>(1, target / std::max<size_t>(bytes_per_block, 1));
	return std::min(n, nblocks);
}

// chunk target of the one-trip pinned paths (link kernel + mirror + checksums): 128 MiB for a foreground codec,
// GEC_BG_CHUNK_MB for a background one (a foreground call then waits for at most that much background traffic)
size_t trip_chunk_bytes(const gec_codec *c);
// chunk size when the caller's memory is pinned end to end (no host staging to bound): GEC_PINNED_CHUNK_MB
size_t pinned_chunk_bytes(const gec_codec *c);

// Host-pointer calls run their chunks through three staging slots, each with its own stream: while
// chunk i is on the PCIe bus / in the kernel, the host drains chunk i-2 and fills chunk i+1.  fill/drain
// run on the calling thread (+ copy pool), enqueue only queues asynchronous work on st.stream.
// Copy KERNELS (pinned callers) additionally chain through PipeChain so that the copies of one direction
// run one after the other: left alone, the slots phase-lock -- all copy-ins at once, then all copy-outs --
// and the link idles in one direction at a time.
struct PipeChain {
	hipEvent_t last_in = nullptr, last_out = nullptr;
	// call before / after launching a copy on `stream`; `mine` = the slot's event for that direction
	int before(hipEvent_t last, hipStream_t stream)
	{
		if (last)
			HIP_TRY(hipStreamWaitEvent(stream, last, 0));
		return GEC_OK;
	}
	int after_in(Staging &st)
	{
		HIP_TRY(hipEventRecord(st.ev_in, st.stream));
		last_in = st.ev_in;
		return GEC_OK;
	}
	int after_out(Staging &st)
	{
		HIP_TRY(hipEventRecord(st.ev_out, st.stream));
		last_out = st.ev_out;
		return GEC_OK;
	}
};

constexpr size_t kSlots = 3;

template <class Fill, class Enqueue, class Drain>
int run_pipeline(const gec_codec *c, size_t nchunks, size_t slot_bytes, size_t nbad, Fill fill, Enqueue enqueue, Drain drain)
{
	DeviceGuard g(c->device);
	if (!g.ok)
		return fail(GEC_E_DEVICE, "hipSetDevice failed");
	StagingLease lease0(c), lease1(c), lease2(c);
	Staging *slot[kSlots] = {&lease0.st, &lease1.st, &lease2.st};
	for (size_t i = 0; i < std::min<size_t>(nchunks, kSlots); ++i) {
		int rc = slot[i]->ensure(slot_bytes, nbad);
		if (rc)
			return rc;
	}
	int rc = GEC_OK;
	auto finish = [&](size_t ci) {
		Staging &st = *slot[ci % kSlots];
		hipError_t e = hipStreamSynchronize(st.stream);
		if (e != hipSuccess)
			rc = fail(GEC_E_DEVICE, std::string("hipStreamSynchronize: ") + hipGetErrorString(e));
		else
			drain(ci, st);
	};
	size_t drained = 0;
	for (size_t ci = 0; ci < nchunks && rc == GEC_OK; ++ci) {
		if (ci >= kSlots) {  // the slot is still busy with chunk ci - kSlots
			finish(ci - kSlots);
			++drained;
			if (rc != GEC_OK)
				break;
		}
		background_yield(c);
		Staging &st = *slot[ci % kSlots];
		fill(ci, st);
		rc = enqueue(ci, st);
	}
	for (; drained < nchunks && rc == GEC_OK; ++drained)
		finish(drained);
	if (rc != GEC_OK)  // leave no work in flight on pooled buffers
		for (Staging *st : slot)
			if (st->stream)
				(void)hipStreamSynchronize(st->stream);
	return rc;
}

// ------------------------------------------------------------------ launches (ec_hip_launch.hip)
// out[r] = XOR_t coef[r][t] * in[t] for r < nout: shard t of block b is read at
// in + b*in_stride + in_base_off[t], row r written at out + b*out_stride +
// out_base_off[r]; only bytes [byte_off, byte_off+byte_len) of every shard are
// touched.  Rows go out in groups of RMAX per launch.
// sum != NULL (shard checksum v3, whole shards only: byte_off = 0, byte_len = S): the launch also leaves the MLH64 leaf sums
// of what it reads and writes (compare mode: checks) in device memory,
//   lsum[((b * slots_total + slot0 + slot) * nleaf_max) + leaf],  slot = input t (when `inputs`), then (k if inputs) + row r
// for mlh_roots_dev to turn into 32-byte checksums.
struct SumOut {
	uint64_t *lsum;
	uint32_t nleaf_max, slots_total, slot0;
	bool inputs;
};
int launch_apply(const gec_codec *c, const uint8_t *in, size_t in_stride, uint8_t *out, size_t out_stride, uint32_t *bad,
		 size_t byte_off, size_t byte_len, size_t nblocks, const size_t *in_base_off, const size_t *out_base_off, int nout,
		 const uint8_t *coef /* nout x k */, int mode, hipStream_t stream, const SumOut *sum = nullptr);
// The pointer-table kernel over tables in DEVICE memory (inputs may live on peer devices): see ec_hip_launch.hip.
size_t ptrs_dev_scratch_bytes(size_t nblocks, size_t k, int nout);
int launch_apply_ptrs_dev(const gec_codec *c, uint8_t *d_scratch, size_t nblocks, const uint8_t *const *in, uint8_t *const *out, int nout,
			  uint32_t cols, const uint8_t *coef /* nout x k */, hipStream_t stream);
// One launch, a decode plan per block: block b is rebuilt in place with plans[pat_of_block[b]] (<= 8 missing shards each).
int launch_apply_pat(const gec_codec *c, uint8_t *d_base, size_t stride, size_t S, size_t nblocks,
		     const std::vector<std::shared_ptr<const Plan>> &plans, const std::vector<uint16_t> &pat_of_block, hipStream_t stream);
// The roots of n shards' leaf sums: shard i's sums at lsum[(slot_map ? slot_map[i] : i) * nleaf_max] (slot_map: device-
// addressable, may be NULL), its length d_len[i] (NULL: len), its 32-byte checksum placed like blake2_dev places results.
int mlh_roots_dev(const gec_codec *c, size_t n, const uint64_t *lsum, uint32_t nleaf_max, const uint64_t *d_len, size_t len,
		  uint8_t *d_out, hipStream_t stream, const uint32_t *slot_map = nullptr, uint32_t group = 0, uint32_t out_group = 0);

// out[b][r] = XOR_t coef[r][t] * in[b][t] over shards that stay in the caller's pinned memory (gf_apply_ptrs):
// in[b*k + t] / valid[b*k + t] name the k input shards of block b and how many of their S bytes exist,
// out[b*nout + r] the output rows.  The tables are written into the staging slot's pinned table area, which the
// kernel reads directly.  k <= PTR_KMAX.
int launch_apply_ptrs(const gec_codec *c, Staging &st, size_t nblocks, const uint8_t *const *in, const uint32_t *valid,
		      uint8_t *const *out, int nout, size_t S, const uint8_t *coef /* nout x k */, hipStream_t stream,
		      uint8_t *d_mirror = nullptr /* [nblocks][k + nout][S]: inputs and outputs also laid down in HBM */,
		      uint32_t *bad = nullptr /* compare with what out[] holds instead of storing: bad[b] = 1 on mismatch */,
		      size_t npat = 0, const uint16_t *pat = nullptr /* per-block coefficient sets: coef = [npat][nout][k], block b uses
		      set pat[b], a NULL out entry = that block's set has no such row; nout <= RMAX; shards may be device memory */,
		      const SumOut *sum = nullptr /* checksum v3: leaf sums of what is read and written (compare: checked), no mirror */);

// Appends the entries to the slot's table and launches ONE copy_table kernel over them.  Entries must have
// 16-byte aligned src and dst.
// max_wgs > 0: a grid of at most that many workgroups walking the tiles; pace_ns > 0: each starts a tile every pace_ns
int launch_copy_table(Staging &st, const std::vector<gec::CopyEntry> &ents, hipStream_t stream, unsigned max_wgs = 0, unsigned pace_ns = 0);
// what a link kernel launched from this slot does with the device's link_busy count
void link_role_of(const Staging &st, unsigned pace_ns, uint32_t **busy, uint32_t *role, uint32_t *wait_ticks);

int launch_clear_flags(uint32_t *d_bad, size_t n, hipStream_t stream);
// the device's count of running foreground link workgroups (kernel_args.hpp, PtrApplyArgs::link_busy); NULL if it could not be allocated
uint32_t *link_busy_counter(int device);

// blake2sum of n messages.  group != 0: message i lives at d_base + (i / group)*group_stride + (i % group)*stride and
// its checksum goes to d_out + 32*((i / group)*out_group + i % group) -- e.g. only the data (or only the parity)
// shards of every stripe.  tree: the shard checksum (BLAKE2b tree mode, blake2b.hpp) instead of the plain hash;
// max_len = the longest message (sizes the leaf grid).  d_state / seg_*: segmented chains (Blake2Args).
int blake2_dev(const gec_codec *c, size_t n, const uint8_t *d_base, const uint64_t *d_off, const uint64_t *d_len, size_t stride,
	       size_t len, uint8_t *d_out, hipStream_t stream, uint32_t group = 0, size_t group_stride = 0, uint32_t out_group = 0,
	       bool tree = false, size_t max_len = 0, uint64_t *d_state = nullptr, uint64_t seg_begin_blk = 0,
	       uint64_t seg_end_blk = ~0ull);

// the copy kernels of the striped decode's two exchanges
int launch_range_pack(const gec::RangeArgs &a, size_t items, hipStream_t stream);
int launch_range_unpack(const gec::RangeArgs &a, size_t items, hipStream_t stream);
int launch_a2a_pack(const gec::A2aArgs &a, size_t items, hipStream_t stream);
int launch_rebuilt_unpack(const gec::RebuiltArgs &a, size_t items, hipStream_t stream);

// ------------------------------------------------------------------ device-side compositions (ec_hip_backend.cpp)
int encode_dev(const gec_codec *c, size_t nblocks, const uint8_t *d_data, size_t data_stride, size_t S, uint8_t *d_parity,
	       size_t parity_stride, hipStream_t stream);
int verify_dev(const gec_codec *c, size_t nblocks, const uint8_t *d_stripes, size_t stride, size_t S, uint32_t *d_bad,
	       hipStream_t stream);
int reconstruct_dev(const gec_codec *c, size_t nblocks, uint8_t *d_base, size_t stride, const size_t *shard_off, const uint8_t *present,
		    bool data_only, size_t byte_off, size_t byte_len, hipStream_t stream);
// Encode + the shard checksum of all k+m shards of every stripe (d_stripes: shard j of block b at b*stride + j*S),
// everything enqueued behind whatever `stream` already holds.  `aux` provides the partner stream and the fork/join events.
int encode_hash_dev(const gec_codec *c, size_t nblocks, uint8_t *d_stripes, size_t stride, size_t S, uint8_t *d_sums, hipStream_t stream,
		    Staging &aux);
int leaf_scratch(const gec_codec *c, hipStream_t stream, size_t bytes, uint8_t **out);
int done_counters(const gec_codec *c, hipStream_t stream, size_t nblocks, uint32_t **out);

// ONE launch for a small trip (fused.hpp): out[b][r] = XOR_t coef_set(pat[b])[r][t] * in[b][t] over pointer tables, AND the
// tree-mode checksum of every input shard (hash_rows: and of every output row) of every block.
//   in / valid [nblocks][k], out [nblocks][nout] (entries may be NULL: row not wanted), coef_sets [npat][nout][k],
//   pat [nblocks] (NULL: every block uses set 0), sums [nblocks][k + (hash_rows ? nout : 0)][32]: device-addressable
//   (the slot's pinned area).  nout may be 0 (checksums only).  Returns GEC_E_INVALID_ARG with "fused: ..." when the
//   shape does not fit (caller falls back to the streaming paths): see fused_fits.
bool fused_fits(const gec_codec *c, size_t nblocks, size_t S, int nout, bool hash_rows);
int launch_fused(const gec_codec *c, Staging &st, size_t nblocks, const uint8_t *const *in, const uint32_t *valid, uint8_t *const *out,
		 int nout, size_t S, const uint8_t *coef_sets, size_t npat, const uint16_t *pat, bool hash_rows, uint8_t *sums,
		 hipStream_t stream);

// contiguous stripe: shard j at j*S
inline std::vector<size_t> stripe_offsets(const gec_codec *c, size_t S)
{
	std::vector<size_t> off((size_t)c->k + c->m);
	for (size_t j = 0; j < off.size(); ++j)
		off[j] = j * S;
	return off;
}

}  // namespace gecimpl
