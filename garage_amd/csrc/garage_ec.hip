// garage_ec.hip -- host side + C ABI of libgarage_ec.so (see include/garage_ec.h).
//
// Host responsibilities (all tiny): coding matrices, erasure-pattern -> decode
// plan (LRU-cached like the crate's decode-matrix cache [EXT core.rs]), shard
// geometry, argument checking, H2D/D2H staging for the host-pointer entry
// points.  Every shard byte is produced by the kernels in kernels.hpp; there is
// no CPU data path.
#include "../../include/garage_ec.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // types only: RCCL itself is resolved with dlopen (gec_group_*)

#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <list>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "gf256.hpp"
#include "kernels.hpp"
#include "blake2b.hpp"

namespace {

thread_local std::string g_last_error;
std::atomic<int> g_variant{0};
// A/B switch (GEC_ROWS16=0): codes with 9..16 output rows as two 8-row passes instead of one 16-row pass
std::atomic<int> g_rows16{[] { const char *e = getenv("GEC_ROWS16"); return e ? atoi(e) : 1; }()};

int fail(int code, const std::string &detail)
{
	g_last_error = detail;
	return code;
}

#define HIP_TRY(expr)                                                                  \
	do {                                                                           \
		hipError_t e_ = (expr);                                                \
		if (e_ != hipSuccess)                                                  \
			return fail(e_ == hipErrorOutOfMemory ? GEC_E_NOMEM : GEC_E_DEVICE, \
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

int check_km(int k, int m)
{
	// same order of checks as ReedSolomon::new [EXT]
	if (k <= 0)
		return fail(GEC_E_TOO_FEW_DATA, "data shards must be >= 1");
	if (m <= 0)
		return fail(GEC_E_TOO_FEW_PARITY, "parity shards must be >= 1");
	if (k + m > GEC_MAX_SHARDS)
		return fail(GEC_E_TOO_MANY_SHARDS, "k + m must be <= 256 in GF(2^8)");
	return GEC_OK;
}

// What to compute for one erasure pattern: out[missing[r]] = rows[r] . in[valid[*]]
struct Plan {
	std::vector<int> valid;    // k input shard indices
	std::vector<int> missing;  // output shard indices
	gec::Matrix rows;          // missing.size() x k
};

// Tiny fork-join pool for the host-side staging copies (pageable user memory <->
// pinned buffers): one memcpy thread tops out near 10 GB/s, well below PCIe Gen5.
class CopyPool {
public:
	explicit CopyPool(unsigned n)
	{
		for (unsigned i = 0; i < n; ++i)
			workers_.emplace_back([this] { run(); });
	}
	~CopyPool()
	{
		{
			std::lock_guard<std::mutex> g(mu_);
			stop_ = true;
		}
		cv_.notify_all();
		for (auto &t : workers_)
			t.join();
	}
	// fn(i) for i in [0, n), spread over the workers and the calling thread
	void parallel_for(size_t n, const std::function<void(size_t)> &fn)
	{
		if (n == 0)
			return;
		if (workers_.empty() || n == 1) {
			for (size_t i = 0; i < n; ++i)
				fn(i);
			return;
		}
		std::unique_lock<std::mutex> call_lock(call_mu_);  // one parallel_for at a time
		{
			std::lock_guard<std::mutex> g(mu_);
			fn_ = &fn;
			n_ = n;
			next_ = 0;
			pending_ = n;
			++epoch_;
		}
		cv_.notify_all();
		work();
		std::unique_lock<std::mutex> g(mu_);
		done_cv_.wait(g, [this] { return pending_ == 0; });
		fn_ = nullptr;
	}

private:
	void work()
	{
		for (;;) {
			size_t i;
			const std::function<void(size_t)> *fn;
			{
				std::lock_guard<std::mutex> g(mu_);
				if (!fn_ || next_ >= n_)
					return;
				i = next_++;
				fn = fn_;
			}
			(*fn)(i);
			std::lock_guard<std::mutex> g(mu_);
			if (--pending_ == 0)
				done_cv_.notify_all();
		}
	}
	void run()
	{
		uint64_t seen = 0;
		for (;;) {
			{
				std::unique_lock<std::mutex> g(mu_);
				cv_.wait(g, [&] { return stop_ || epoch_ != seen; });
				if (stop_)
					return;
				seen = epoch_;
			}
			work();
		}
	}
	std::vector<std::thread> workers_;
	std::mutex mu_, call_mu_;
	std::condition_variable cv_, done_cv_;
	const std::function<void(size_t)> *fn_ = nullptr;
	size_t n_ = 0, next_ = 0, pending_ = 0;
	uint64_t epoch_ = 0;
	bool stop_ = false;
};

// Worker count: GEC_COPY_THREADS (0 = copy on the calling thread only), default 7 or fewer on small hosts.
unsigned copy_pool_threads()
{
	const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
	if (const char *e = getenv("GEC_COPY_THREADS"))
		return (unsigned)std::min<unsigned long>(strtoul(e, nullptr, 0), 64ul);
	return std::min(7u, hw - 1);
}

// Pinned host ranges the caller told us about (gec_host_alloc / gec_host_register): blocks and
// output buffers that lie inside one go over PCIe by DMA straight from / to the caller's memory,
// without the pageable -> pinned staging copy (which costs a third of the PCIe-inclusive rate).
class PinnedRanges {
public:
	// dev_delta: what to add to a host address inside the range to get the address the GPU must use
	// (0 for hipHostMalloc; hipHostRegister may map the pages at a different device address)
	void add(const void *p, size_t n, bool owned, intptr_t dev_delta = 0)
	{
		std::lock_guard<std::mutex> g(mu_);
		ranges_[reinterpret_cast<uintptr_t>(p)] = {n, owned, dev_delta};
	}
	// returns true and whether the library allocated it
	bool remove(const void *p, bool &owned)
	{
		std::lock_guard<std::mutex> g(mu_);
		auto it = ranges_.find(reinterpret_cast<uintptr_t>(p));
		if (it == ranges_.end())
			return false;
		owned = it->second.owned;
		ranges_.erase(it);
		return true;
	}
	bool contains(const void *p, size_t n, intptr_t *dev_delta = nullptr) const
	{
		if (!p)
			return false;
		const uintptr_t a = reinterpret_cast<uintptr_t>(p);
		std::lock_guard<std::mutex> g(mu_);
		if (ranges_.empty())
			return false;
		auto it = ranges_.upper_bound(a);
		if (it == ranges_.begin())
			return false;
		--it;
		if (!(a >= it->first && a + n <= it->first + it->second.len))
			return false;
		if (dev_delta)
			*dev_delta = it->second.dev_delta;
		return true;
	}
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
		bool owned;
		intptr_t dev_delta;
	};
	mutable std::mutex mu_;
	std::map<uintptr_t, R> ranges_;
};

PinnedRanges &pinned()
{
	static PinnedRanges r;
	return r;
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

	int ensure_segments(int num_cu)
	{
		if (stream3)
			return GEC_OK;
		for (int i = 0; i < kMaxSeg; ++i)
			HIP_TRY(hipEventCreateWithFlags(&ev_seg[i], hipEventDisableTiming));
		static const int up_cus = [] {
			const char *e = std::getenv("GEC_UPLOAD_CUS");  // 0 = no CU masks (A/B)
			return e ? std::atoi(e) : 16;
		}();
		if (up_cus > 0 && up_cus < num_cu) {
			const int words = (num_cu + 31) / 32;
			std::vector<uint32_t> up(words, 0), rest(words, 0);
			for (int i = 0; i < num_cu; ++i)
				(i < up_cus ? up : rest)[i / 32] |= 1u << (i % 32);
			// (a runtime or partition mode without CU masks is not an error: the paths then share all CUs, slower)
			if (hipExtStreamCreateWithCUMask(&stream_up, (uint32_t)words, up.data()) != hipSuccess)
				stream_up = nullptr;
			if (!stream_up || hipExtStreamCreateWithCUMask(&stream_chain, (uint32_t)words, rest.data()) != hipSuccess) {
				if (stream_up)
					(void)hipStreamDestroy(stream_up);
				stream_up = stream_chain = nullptr;
				(void)hipGetLastError();
			}
		}
		HIP_TRY(hipStreamCreateWithFlags(&stream3, hipStreamNonBlocking));
		return GEC_OK;
	}

	int ensure_big(size_t bytes)
	{
		if (bytes <= big_cap)
			return GEC_OK;
		if (d_big)
			(void)hipFree(d_big);
		d_big = nullptr;
		big_cap = 0;
		HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d_big), bytes));
		big_cap = bytes;
		return GEC_OK;
	}

	int ensure_tab(size_t entries)
	{
		tab_used = 0;
		if (entries <= tab_cap)
			return GEC_OK;
		if (h_tab)
			(void)hipHostFree(h_tab);
		h_tab = nullptr;
		tab_cap = 0;
		HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&h_tab), entries * sizeof(gec::CopyEntry), hipHostMallocDefault));
		tab_cap = entries;
		return GEC_OK;
	}

	int ensure(size_t bytes, size_t nbad)
	{
		if (!stream)
			HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
		if (!stream2) {
			HIP_TRY(hipStreamCreateWithFlags(&stream2, hipStreamNonBlocking));
			HIP_TRY(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
			HIP_TRY(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
			HIP_TRY(hipEventCreateWithFlags(&ev_in, hipEventDisableTiming));
			HIP_TRY(hipEventCreateWithFlags(&ev_out, hipEventDisableTiming));
		}
		if (bytes > cap) {
			if (h_buf)
				(void)hipHostFree(h_buf);
			if (d_buf)
				(void)hipFree(d_buf);
			h_buf = d_buf = nullptr;
			cap = 0;
			HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&h_buf), bytes, hipHostMallocDefault));
			HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d_buf), bytes));
			cap = bytes;
		}
		if (nbad > bad_cap) {
			if (h_bad)
				(void)hipHostFree(h_bad);
			if (d_bad)
				(void)hipFree(d_bad);
			h_bad = d_bad = nullptr;
			bad_cap = 0;
			HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&h_bad), nbad * sizeof(uint32_t), hipHostMallocDefault));
			HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d_bad), nbad * sizeof(uint32_t)));
			bad_cap = nbad;
		}
		return GEC_OK;
	}
	void release()
	{
		if (h_buf)
			(void)hipHostFree(h_buf);
		if (d_buf)
			(void)hipFree(d_buf);
		if (h_bad)
			(void)hipHostFree(h_bad);
		if (d_bad)
			(void)hipFree(d_bad);
		if (h_tab)
			(void)hipHostFree(h_tab);
		if (d_big)
			(void)hipFree(d_big);
		if (stream)
			(void)hipStreamDestroy(stream);
		if (stream2)
			(void)hipStreamDestroy(stream2);
		if (ev_fork)
			(void)hipEventDestroy(ev_fork);
		if (ev_join)
			(void)hipEventDestroy(ev_join);
		if (ev_in)
			(void)hipEventDestroy(ev_in);
		if (ev_out)
			(void)hipEventDestroy(ev_out);
		for (hipEvent_t e : ev_seg)
			if (e)
				(void)hipEventDestroy(e);
		if (stream3)
			(void)hipStreamDestroy(stream3);
		if (stream_up)
			(void)hipStreamDestroy(stream_up);
		if (stream_chain)
			(void)hipStreamDestroy(stream_chain);
		*this = Staging();
	}
};

}  // namespace

struct gec_codec {
	int k = 0, m = 0, device = 0;
	int num_cu = 256;
	gec::Matrix enc;               // (k+m) x k
	gec::LogExp *d_logexp = nullptr;

	// decode-plan LRU, keyed by present bitmap + data_only
	static constexpr size_t kCacheCap = 254;
	mutable std::mutex cache_mu;
	mutable std::list<std::string> lru;
	mutable std::unordered_map<std::string, std::pair<std::shared_ptr<const Plan>, std::list<std::string>::iterator>> cache;
	mutable uint64_t inversions = 0;

	mutable std::mutex pool_mu;
	mutable std::vector<Staging> pool;

	// One copy pool per codec = per device: a process that drives several GPUs (one codec each)
	// must not funnel all their staging copies through one set of threads.  Created on the
	// first host-pointer call; device-API-only users never start the threads.
	// leaf-digest scratch of the tree-mode shard checksums, one per stream that ever hashed (work on one
	// stream is ordered, so reuse on the same stream is safe; grow-only)
	struct LeafScratch {
		uint8_t *p = nullptr;
		size_t cap = 0;
	};
	mutable std::mutex leaf_mu;
	mutable std::map<hipStream_t, LeafScratch> leaf_scratch;

	mutable std::once_flag copy_once;
	mutable std::unique_ptr<CopyPool> copy_threads;
	CopyPool &copy_pool() const
	{
		std::call_once(copy_once, [this] { copy_threads.reset(new CopyPool(copy_pool_threads())); });
		return *copy_threads;
	}
};

// RCCL entry points, resolved on first use (the library must load on hosts without RCCL,
// and inside a PyTorch process it must bind to the RCCL torch already loaded).
namespace {
struct Rccl {
	void *handle = nullptr;
	decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
	decltype(&ncclCommInitRank) CommInitRank = nullptr;
	decltype(&ncclCommDestroy) CommDestroy = nullptr;
	decltype(&ncclAllGather) AllGather = nullptr;
	decltype(&ncclSend) Send = nullptr;
	decltype(&ncclRecv) Recv = nullptr;
	decltype(&ncclGroupStart) GroupStart = nullptr;
	decltype(&ncclGroupEnd) GroupEnd = nullptr;
	decltype(&ncclGetErrorString) GetErrorString = nullptr;
	std::string error;
};

const Rccl &rccl()
{
	static const Rccl r = [] {
		Rccl x;
		std::vector<std::string> names;
		if (const char *e = getenv("GEC_RCCL_LIB"))
			names.push_back(e);
		names.insert(names.end(), {"librccl.so.1", "librccl.so"});
		for (const std::string &n : names) {
			x.handle = dlopen(n.c_str(), RTLD_NOW | RTLD_GLOBAL);
			if (x.handle)
				break;
			const char *de = dlerror();
			x.error += n + ": " + (de ? de : "?") + "; ";
		}
		if (!x.handle)
			return x;
		x.GetUniqueId = reinterpret_cast<decltype(x.GetUniqueId)>(dlsym(x.handle, "ncclGetUniqueId"));
		x.CommInitRank = reinterpret_cast<decltype(x.CommInitRank)>(dlsym(x.handle, "ncclCommInitRank"));
		x.CommDestroy = reinterpret_cast<decltype(x.CommDestroy)>(dlsym(x.handle, "ncclCommDestroy"));
		x.AllGather = reinterpret_cast<decltype(x.AllGather)>(dlsym(x.handle, "ncclAllGather"));
		x.GetErrorString = reinterpret_cast<decltype(x.GetErrorString)>(dlsym(x.handle, "ncclGetErrorString"));
		x.Send = reinterpret_cast<decltype(x.Send)>(dlsym(x.handle, "ncclSend"));
		x.Recv = reinterpret_cast<decltype(x.Recv)>(dlsym(x.handle, "ncclRecv"));
		x.GroupStart = reinterpret_cast<decltype(x.GroupStart)>(dlsym(x.handle, "ncclGroupStart"));
		x.GroupEnd = reinterpret_cast<decltype(x.GroupEnd)>(dlsym(x.handle, "ncclGroupEnd"));
		if (!x.GetUniqueId || !x.CommInitRank || !x.CommDestroy || !x.AllGather || !x.GetErrorString || !x.Send || !x.Recv ||
		    !x.GroupStart || !x.GroupEnd) {
			x.error = "RCCL library lacks a required symbol";
			x.handle = nullptr;
		}
		return x;
	}();
	return r;
}
}  // namespace

struct gec_group {
	const gec_codec *c = nullptr;
	int rank = 0, nranks = 1;
	gec_allgather_fn all_gather = nullptr;
	gec_alltoall_fn all_to_all = nullptr;
	void *ctx = nullptr;
	ncclComm_t comm = nullptr;  // RCCL transport only
	// all-to-all exchange scratch
	uint8_t *d_a2a_send = nullptr, *d_a2a_recv = nullptr;
	size_t a2a_cap = 0;
	uint64_t bytes_exchanged = 0;  // bytes this rank RECEIVED from other ranks in the last decode call
	// scratch for the exchange of the rebuilt ranges (step 3)
	uint8_t *d_send = nullptr, *d_recv = nullptr;
	size_t send_cap = 0, recv_cap = 0;
};

namespace {

int rccl_all_gather(void *ctx, const void *d_send, void *d_recv, size_t bytes, void *hip_stream)
{
	gec_group *g = static_cast<gec_group *>(ctx);
	ncclResult_t r = rccl().AllGather(d_send, d_recv, bytes, ncclUint8, g->comm, static_cast<hipStream_t>(hip_stream));
	if (r != ncclSuccess)
		return fail(GEC_E_DEVICE, std::string("ncclAllGather: ") + rccl().GetErrorString(r));
	return GEC_OK;
}

// all-to-all over RCCL: one grouped ncclSend/ncclRecv pair per peer (xGMI is a full mesh: every pair has its own link)
int rccl_all_to_all(void *ctx, const void *d_send, void *d_recv, size_t bytes, void *hip_stream)
{
	gec_group *g = static_cast<gec_group *>(ctx);
	const Rccl &R = rccl();
	hipStream_t s = static_cast<hipStream_t>(hip_stream);
	ncclResult_t r = R.GroupStart();
	for (int q = 0; q < g->nranks && r == ncclSuccess; ++q) {
		r = R.Send(static_cast<const uint8_t *>(d_send) + (size_t)q * bytes, bytes, ncclUint8, q, g->comm, s);
		if (r == ncclSuccess)
			r = R.Recv(static_cast<uint8_t *>(d_recv) + (size_t)q * bytes, bytes, ncclUint8, q, g->comm, s);
	}
	ncclResult_t e = R.GroupEnd();
	if (r == ncclSuccess)
		r = e;
	if (r != ncclSuccess)
		return fail(GEC_E_DEVICE, std::string("ncclSend/ncclRecv: ") + R.GetErrorString(r));
	return GEC_OK;
}

int get_plan(const gec_codec *c, const uint8_t *present, bool data_only, std::shared_ptr<const Plan> &out)
{
	const int k = c->k, n = c->k + c->m;
	std::string key(reinterpret_cast<const char *>(present), n);
	for (auto &ch : key)
		ch = ch ? 1 : 0;
	key.push_back(data_only ? 1 : 0);
	{
		std::lock_guard<std::mutex> g(c->cache_mu);
		auto it = c->cache.find(key);
		if (it != c->cache.end()) {
			c->lru.splice(c->lru.begin(), c->lru, it->second.second);
			out = it->second.first;
			return GEC_OK;
		}
	}
	auto plan = std::make_shared<Plan>();
	for (int j = 0; j < n && (int)plan->valid.size() < k; ++j)
		if (present[j])
			plan->valid.push_back(j);
	if ((int)plan->valid.size() < k)
		return fail(GEC_E_TOO_FEW_PRESENT, "fewer than k shards present");
	for (int j = 0; j < n; ++j)
		if (!present[j] && !(data_only && j >= k))
			plan->missing.push_back(j);
	if (!plan->missing.empty()) {
		gec::Matrix sub(k, k), dec;
		for (int t = 0; t < k; ++t)
			std::memcpy(&sub.at(t, 0), c->enc.row(plan->valid[t]), k);
		if (!gec::invert(sub, dec))
			return fail(GEC_E_INVALID_ARG, "decode sub-matrix singular (cannot happen for an MDS code)");
		// missing data j: row j of dec.  missing parity p: enc[p] * dec, which
		// equals re-encoding p from the completed data (crate order) because GF
		// arithmetic is exact.
		plan->rows = gec::Matrix((int)plan->missing.size(), k);
		for (size_t r = 0; r < plan->missing.size(); ++r) {
			int j = plan->missing[r];
			if (j < k) {
				std::memcpy(&plan->rows.at((int)r, 0), dec.row(j), k);
			} else {
				gec::Matrix prow(1, k);
				std::memcpy(&prow.at(0, 0), c->enc.row(j), k);
				gec::Matrix comp = gec::matmul(prow, dec);
				std::memcpy(&plan->rows.at((int)r, 0), comp.row(0), k);
			}
		}
	}
	{
		std::lock_guard<std::mutex> g(c->cache_mu);
		++c->inversions;
		if (c->cache.find(key) == c->cache.end()) {
			c->lru.push_front(key);
			c->cache[key] = {plan, c->lru.begin()};
			if (c->cache.size() > gec_codec::kCacheCap) {
				c->cache.erase(c->lru.back());
				c->lru.pop_back();
			}
		}
	}
	out = plan;
	return GEC_OK;
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
// exercised on small inputs.  Read once.
uint64_t launch_cols_limit()
{
	static const uint64_t v = [] {
		const char *e = getenv("GEC_MAX_COLS_PER_LAUNCH");
		return e ? strtoull(e, nullptr, 0) : 0ull;
	}();
	return v;
}

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

template <int MW, int MODE, int TPB>
void launch_nibble(const gec::ApplyArgs &a, const gec::LogExp *le, int kc, unsigned grid, size_t lds, hipStream_t s)
{
	if constexpr (MW == 2) {
		if (kc == 10) {
			hipLaunchKernelGGL((gec::gf_apply_nibble<MW, MODE, 10, kCPT, true, 256>), dim3(grid), dim3(256), lds, s, a, le);
			return;
		}
	}
	if constexpr (MW == 1) {
		if (kc == 10) {
			hipLaunchKernelGGL((gec::gf_apply_nibble<MW, MODE, 10, kCPT, true, TPB>), dim3(grid), dim3(TPB), lds, s, a, le);
			return;
		}
		if (kc == 12) {
			hipLaunchKernelGGL((gec::gf_apply_nibble<MW, MODE, 12, kCPT, true, TPB>), dim3(grid), dim3(TPB), lds, s, a, le);
			return;
		}
		if (kc == 16) {
			hipLaunchKernelGGL((gec::gf_apply_nibble<MW, MODE, 16, kCPT, true, TPB>), dim3(grid), dim3(TPB), lds, s, a, le);
			return;
		}
	}
#define GEC_CASE(KC)                                                                                              \
	case KC:                                                                                                  \
		hipLaunchKernelGGL((gec::gf_apply_nibble<MW, MODE, KC, kCPT, true, TPB>), dim3(grid), dim3(TPB), lds, s, \
				   a, le);                                                                        \
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

// out[r] = XOR_t coef[r][t] * in[t] for r < nout: shard t of block b is read at
// in + b*in_stride + in_base_off[t], row r written at out + b*out_stride +
// out_base_off[r]; only bytes [byte_off, byte_off+byte_len) of every shard are
// touched.  Rows go out in groups of RMAX per launch.
int launch_apply(const gec_codec *c, const uint8_t *in, size_t in_stride, uint8_t *out, size_t out_stride,
		 uint32_t *bad, size_t byte_off, size_t byte_len, size_t nblocks, const size_t *in_base_off,
		 const size_t *out_base_off, int nout, const uint8_t *coef /* nout x k */, int mode,
		 hipStream_t stream)
{
	const int k = c->k;
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
	const int variant = g_variant.load(std::memory_order_relaxed);
	int rows = 0;
	for (int r0 = 0; r0 < nout; r0 += rows) {
		const Geometry geo = pick_geometry(k, nout - r0, variant == 0 && g_rows16.load(std::memory_order_relaxed) != 0);
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
			const unsigned grid = (unsigned)std::min<uint64_t>(ntiles, (uint64_t)c->num_cu * 8);
			if (mode == gec::MODE_STORE)
				hipLaunchKernelGGL((gec::gf_apply_logexp<gec::MODE_STORE>), dim3(grid), dim3(gec::BLOCK), 0, stream, a, c->d_logexp);
			else
				hipLaunchKernelGGL((gec::gf_apply_logexp<gec::MODE_COMPARE>), dim3(grid), dim3(gec::BLOCK), 0, stream, a, c->d_logexp);
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
		const size_t lds = geo.lds;
		gec::ApplyArgs la = a;
		for (uint64_t b0 = 0; b0 < nblocks; b0 += blocks_per_launch) {
			const uint64_t nb = std::min<uint64_t>(blocks_per_launch, nblocks - b0);
			la.in = in + b0 * in_stride;
			la.out = out + b0 * out_stride;
			la.bad = bad ? bad + b0 : nullptr;
			la.nblocks = (uint32_t)nb;
			la.total_cols = (uint32_t)(nb * a.cols);
			// multiple of 8: the kernel hands each XCD a contiguous range of tiles
			const unsigned grid = (unsigned)(((la.total_cols + tile_cols - 1) / tile_cols + 7) / 8 * 8);
			if (mw == 1 && mode == gec::MODE_STORE)
				launch_nibble<1, gec::MODE_STORE, kThreadsMW1>(la, c->d_logexp, geo.kc, grid, lds, stream);
			else if (mw == 1 && rows == 4)  // all four row slots real: stored rows prefetched behind the data loads
				launch_nibble<1, gec::MODE_COMPARE_PF, kThreadsMW1>(la, c->d_logexp, geo.kc, grid, lds, stream);
			else if (mw == 1)
				launch_nibble<1, gec::MODE_COMPARE, kThreadsMW1>(la, c->d_logexp, geo.kc, grid, lds, stream);
			else if (mw == 2 && mode == gec::MODE_STORE)
				launch_nibble<2, gec::MODE_STORE, kThreadsMW2>(la, c->d_logexp, geo.kc, grid, lds, stream);
			else if (mw == 2)
				launch_nibble<2, gec::MODE_COMPARE, kThreadsMW2>(la, c->d_logexp, geo.kc, grid, lds, stream);
			else if (mode == gec::MODE_STORE)
				launch_nibble<4, gec::MODE_STORE, kThreadsMW4>(la, c->d_logexp, geo.kc, grid, lds, stream);
			else
				launch_nibble<4, gec::MODE_COMPARE, kThreadsMW4>(la, c->d_logexp, geo.kc, grid, lds, stream);
			HIP_TRY(hipGetLastError());
		}
	}
	return GEC_OK;
}

int check_dev_layout(const void *p, size_t stride, size_t S, size_t need)
{
	if (S == 0)
		return fail(GEC_E_EMPTY_SHARD, "shard length is 0");
	if (S % 64 != 0)
		return fail(GEC_E_INCORRECT_SHARD_SIZE, "S must be a multiple of 64");
	if (!p)
		return fail(GEC_E_INVALID_ARG, "NULL device pointer");
	if (reinterpret_cast<uintptr_t>(p) % 16 != 0 || stride % 16 != 0)
		return fail(GEC_E_INVALID_ARG, "device pointer/stride must be 16-byte aligned");
	if (stride < need)
		return fail(GEC_E_INCORRECT_SHARD_SIZE, "stride smaller than the shards it must hold");
	return GEC_OK;
}

int encode_dev(const gec_codec *c, size_t nblocks, const uint8_t *d_data, size_t data_stride, size_t S,
	       uint8_t *d_parity, size_t parity_stride, hipStream_t stream)
{
	const int k = c->k, m = c->m;
	std::vector<size_t> in_off(k), out_off(m);
	for (int t = 0; t < k; ++t)
		in_off[t] = (size_t)t * S;
	for (int r = 0; r < m; ++r)
		out_off[r] = (size_t)r * S;
	return launch_apply(c, d_data, data_stride, d_parity, parity_stride, nullptr, 0, S, nblocks, in_off.data(),
			    out_off.data(), m, c->enc.row(k), gec::MODE_STORE, stream);
}

int verify_dev(const gec_codec *c, size_t nblocks, const uint8_t *d_stripes, size_t stride, size_t S,
	       uint32_t *d_bad, hipStream_t stream)
{
	const int k = c->k, m = c->m;
	std::vector<size_t> in_off(k), out_off(m);
	for (int t = 0; t < k; ++t)
		in_off[t] = (size_t)t * S;
	for (int r = 0; r < m; ++r)
		out_off[r] = (size_t)(k + r) * S;
	hipLaunchKernelGGL(gec::clear_flags, dim3((unsigned)((nblocks + 255) / 256)), dim3(256), 0, stream, d_bad, (uint32_t)nblocks);
	HIP_TRY(hipGetLastError());
	return launch_apply(c, d_stripes, stride, const_cast<uint8_t *>(d_stripes), stride, d_bad, 0, S, nblocks,
			    in_off.data(), out_off.data(), m, c->enc.row(k), gec::MODE_COMPARE, stream);
}

int leaf_scratch(const gec_codec *c, hipStream_t stream, size_t bytes, uint8_t **out)
{
	std::lock_guard<std::mutex> g(c->leaf_mu);
	gec_codec::LeafScratch &ls = c->leaf_scratch[stream];
	if (bytes > ls.cap) {
		if (ls.p) {
			HIP_TRY(hipStreamSynchronize(stream));  // earlier launches on this stream may still read the old one
			(void)hipFree(ls.p);
			ls.p = nullptr;
			ls.cap = 0;
		}
		const size_t want = std::max<size_t>(bytes + bytes / 4, 1 << 20);
		HIP_TRY(hipMalloc(reinterpret_cast<void **>(&ls.p), want));
		ls.cap = want;
	}
	*out = ls.p;
	return GEC_OK;
}

// blake2sum of n messages.  group != 0: message i lives at d_base + (i / group)*group_stride + (i % group)*stride and
// its checksum goes to d_out + 32*((i / group)*out_group + i % group) -- e.g. only the data (or only the parity)
// shards of every stripe.  tree: the shard checksum (BLAKE2b tree mode, blake2b.hpp) instead of the plain hash;
// max_len = the longest message (sizes the leaf grid).
int blake2_dev(const gec_codec *c, size_t n, const uint8_t *d_base, const uint64_t *d_off, const uint64_t *d_len, size_t stride,
	       size_t len, uint8_t *d_out, hipStream_t stream, uint32_t group = 0, size_t group_stride = 0,
	       uint32_t out_group = 0, bool tree = false, size_t max_len = 0, uint64_t *d_state = nullptr,
	       uint64_t seg_begin_blk = 0, uint64_t seg_end_blk = ~0ull)
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
		static const int addmode = [] { const char *e = getenv("GEC_B2_ADD"); return e ? atoi(e) : 0; }();  // A/B, see blake2b.hpp
		const dim3 lgrid((unsigned)((lanes + 63) / 64));
		if (addmode == 0)
			hipLaunchKernelGGL(gec::shardsum_leaves<0>, lgrid, dim3(64), 0, stream, a, nleaf, scratch);
		else
			hipLaunchKernelGGL(gec::shardsum_leaves<1>, lgrid, dim3(64), 0, stream, a, nleaf, scratch);
		HIP_TRY(hipGetLastError());
		hipLaunchKernelGGL(gec::shardsum_roots, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, stream, a, nleaf, scratch);
		HIP_TRY(hipGetLastError());
		return GEC_OK;
	}
	// one lane per message is the faster kernel once there are enough messages to put a
	// wave on every SIMD (1024 SIMDs x 64 lanes); below that the quad kernel (4 lanes per
	// message, ~4x shorter chain) wins.  GEC_BLAKE2_KERNEL=lane|quad forces one (A/B).
	static const int forced = [] {
		const char *e = getenv("GEC_BLAKE2_KERNEL");
		return !e ? 0 : (e[0] == 'l' ? 1 : (e[0] == 'q' ? 2 : 0));
	}();
	const bool quad = d_state ? true : forced ? forced == 2 : n < 40000;  // segments: the quad kernel only
	if (quad)
		hipLaunchKernelGGL(gec::blake2b_batch_quad, dim3((unsigned)((n + 15) / 16)), dim3(64), 0, stream, a);
	else
		{
		static const int addmode = [] { const char *e = getenv("GEC_B2_ADD"); return e ? atoi(e) : 0; }();
		if (addmode == 0)
			hipLaunchKernelGGL(gec::blake2b_batch<0>, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, stream, a);
		else
			hipLaunchKernelGGL(gec::blake2b_batch<1>, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, stream, a);
	}
	HIP_TRY(hipGetLastError());
	return GEC_OK;
}

int reconstruct_dev(const gec_codec *c, size_t nblocks, uint8_t *d_base, size_t stride, const size_t *shard_off,
		    const uint8_t *present, bool data_only, size_t byte_off, size_t byte_len, hipStream_t stream)
{
	std::shared_ptr<const Plan> plan;
	int rc = get_plan(c, present, data_only, plan);
	if (rc)
		return rc;
	if (plan->missing.empty())
		return GEC_OK;
	const int k = c->k;
	std::vector<size_t> in_off(k), out_off(plan->missing.size());
	for (int t = 0; t < k; ++t)
		in_off[t] = shard_off[plan->valid[t]];
	for (size_t r = 0; r < plan->missing.size(); ++r)
		out_off[r] = shard_off[plan->missing[r]];
	return launch_apply(c, d_base, stride, d_base, stride, nullptr, byte_off, byte_len, nblocks, in_off.data(),
			    out_off.data(), (int)plan->missing.size(), plan->rows.v.data(), gec::MODE_STORE, stream);
}

// Appends `n` entries to the slot's table and launches ONE copy_table kernel over them.  Entries must have
// 16-byte aligned src and dst (callers check with copyable()).
// GEC_ZERO_COPY=0: caller memory that is pinned still goes through the HBM staging pipeline (A/B switch)
bool zero_copy_enabled()
{
	static const bool on = [] {
		const char *e = std::getenv("GEC_ZERO_COPY");
		return !(e && e[0] == '0');
	}();
	return on;
}

// out[b][r] = XOR_t coef[r][t] * in[b][t] over shards that stay in the caller's pinned memory (gf_apply_ptrs):
// in[b*k + t] / valid[b*k + t] name the k input shards of block b and how many of their S bytes exist,
// out[b*nout + r] the output rows.  The tables are written into the staging slot's pinned table area, which the
// kernel reads directly.  k <= PTR_KMAX.
int launch_apply_ptrs(const gec_codec *c, Staging &st, size_t nblocks, const uint8_t *const *in, const uint32_t *valid,
		      uint8_t *const *out, int nout, size_t S, const uint8_t *coef /* nout x k */, hipStream_t stream,
		      uint8_t *d_mirror = nullptr /* [nblocks][k + nout][S]: inputs and outputs also laid down in HBM */,
		      uint32_t *bad = nullptr /* compare with what out[] holds instead of storing: bad[b] = 1 on mismatch */)
{
	const size_t k = c->k;
	if (nblocks == 0 || nout == 0)
		return GEC_OK;
	if (k > (size_t)gec::PTR_KMAX || S / 16 > 0xffffffffull)
		return fail(GEC_E_INVALID_ARG, "shape not supported by the pointer-table kernel");
	const size_t in_bytes = nblocks * k * 8, valid_bytes = (nblocks * k * 4 + 7) / 8 * 8, out_bytes = nblocks * (size_t)nout * 8;
	const size_t need = (st.tab_used * sizeof(gec::CopyEntry) + in_bytes + valid_bytes + out_bytes) / sizeof(gec::CopyEntry) + 2;
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
	a.cols = (uint32_t)(S / 16);
	a.k = (uint32_t)k;
	const unsigned gx = (a.cols + 255) / 256;
	int rows = 0;
	size_t out_done = 0;  // entries of t_out consumed by earlier row groups
	for (int r0 = 0; r0 < nout; r0 += rows) {
		rows = std::min(gec::RMAX, nout - r0);
		a.rows = (uint32_t)rows;
		for (int r = 0; r < gec::RMAX; ++r)
			for (size_t t = 0; t < k; ++t)
				a.coef[t][r] = r < rows ? coef[(size_t)(r0 + r) * k + t] : 0;
		uint8_t **grp = t_out + out_done;  // [nblocks][rows] for this group
		for (size_t b = 0; b < nblocks; ++b)
			for (int r = 0; r < rows; ++r)
				grp[b * rows + r] = out[b * nout + r0 + r];
		out_done += nblocks * rows;
		const int mw = rows <= 4 ? 1 : 2;
		const size_t lds = k * 32 * 4 * mw + 768 + k * gec::RMAX;
		a.mirror_stride = (k + (size_t)nout) * S;
		a.mirror_row0 = (k + (size_t)r0) * S;
		a.mirror_inputs = r0 == 0;
		for (size_t b0 = 0; b0 < nblocks; b0 += 65535) {
			const unsigned gy = (unsigned)std::min<size_t>(65535, nblocks - b0);
			a.in = t_in + b0 * k;
			a.in_valid = t_valid + b0 * k;
			a.out = grp + b0 * rows;
			a.mirror = d_mirror ? d_mirror + b0 * a.mirror_stride : nullptr;
			a.bad = bad ? bad + b0 : nullptr;
			if (bad && d_mirror && mw == 1)
				hipLaunchKernelGGL((gec::gf_apply_ptrs<1, 5, true, true>), dim3(gx, gy), dim3(256), lds, stream, a, c->d_logexp);
			else if (bad && d_mirror)
				hipLaunchKernelGGL((gec::gf_apply_ptrs<2, 5, true, true>), dim3(gx, gy), dim3(256), lds, stream, a, c->d_logexp);
			else if (bad && mw == 1)
				hipLaunchKernelGGL((gec::gf_apply_ptrs<1, 5, false, true>), dim3(gx, gy), dim3(256), lds, stream, a, c->d_logexp);
			else if (bad)
				hipLaunchKernelGGL((gec::gf_apply_ptrs<2, 5, false, true>), dim3(gx, gy), dim3(256), lds, stream, a, c->d_logexp);
			else if (mw == 1 && d_mirror)
				hipLaunchKernelGGL((gec::gf_apply_ptrs<1, 5, true>), dim3(gx, gy), dim3(256), lds, stream, a, c->d_logexp);
			else if (mw == 1)
				hipLaunchKernelGGL((gec::gf_apply_ptrs<1, 5, false>), dim3(gx, gy), dim3(256), lds, stream, a, c->d_logexp);
			else if (d_mirror)
				hipLaunchKernelGGL((gec::gf_apply_ptrs<2, 5, true>), dim3(gx, gy), dim3(256), lds, stream, a, c->d_logexp);
			else
				hipLaunchKernelGGL((gec::gf_apply_ptrs<2, 5, false>), dim3(gx, gy), dim3(256), lds, stream, a, c->d_logexp);
			HIP_TRY(hipGetLastError());
		}
	}
	return GEC_OK;
}

int launch_copy_table(Staging &st, const std::vector<gec::CopyEntry> &ents, hipStream_t stream)
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
	const unsigned gx = (unsigned)((maxb >> 4) / 1024 + 1);
	// grid.y <= 65535: split long tables
	for (size_t e0 = 0; e0 < ents.size(); e0 += 65535) {
		const unsigned gy = (unsigned)std::min<size_t>(65535, ents.size() - e0);
		hipLaunchKernelGGL(gec::copy_table, dim3(gx, gy), dim3(256), 0, stream, tab + e0);
		HIP_TRY(hipGetLastError());
	}
	return GEC_OK;
}

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// Encode + the blake2sum of all k+m shards of every stripe (d_stripes: shard j of block b at b*stride + j*S),
// everything enqueued behind whatever `stream` already holds.  The checksums of the k data shards do not depend
// on the encode, so they are computed on a second stream BESIDE the RS kernel (HBM-bound, it leaves the VALUs
// mostly idle; the hash is a pure VALU dependency chain); only the m parity checksums follow the encode.
// `aux` provides the partner stream and the fork/join events.
int encode_hash_dev(const gec_codec *c, size_t nblocks, uint8_t *d_stripes, size_t stride, size_t S, uint8_t *d_sums,
		    hipStream_t stream, Staging &aux)
{
	const size_t k = c->k, m = c->m, n = k + m;
	static const bool fork = [] { const char *e = getenv("GEC_HASH_FORK"); return !e || atoi(e) != 0; }();  // A/B switch
	if (fork) {
		HIP_TRY(hipEventRecord(aux.ev_fork, stream));
		HIP_TRY(hipStreamWaitEvent(aux.stream2, aux.ev_fork, 0));
	}
	int rc = blake2_dev(c, nblocks * k, d_stripes, nullptr, nullptr, S, S, d_sums, fork ? aux.stream2 : stream, (uint32_t)k, stride, (uint32_t)n, true);
	if (rc)
		return rc;
	if (fork)
		HIP_TRY(hipEventRecord(aux.ev_join, aux.stream2));
	rc = encode_dev(c, nblocks, d_stripes, stride, S, d_stripes + k * S, stride, stream);
	if (rc)
		return rc;
	rc = blake2_dev(c, nblocks * m, d_stripes + k * S, nullptr, nullptr, S, S, d_sums + 32 * k, stream, (uint32_t)m, stride, (uint32_t)n, true);
	if (rc)
		return rc;
	if (fork)
		HIP_TRY(hipStreamWaitEvent(stream, aux.ev_join, 0));
	return GEC_OK;
}

// contiguous stripe: shard j at j*S
std::vector<size_t> stripe_offsets(const gec_codec *c, size_t S)
{
	std::vector<size_t> off((size_t)c->k + c->m);
	for (size_t j = 0; j < off.size(); ++j)
		off[j] = j * S;
	return off;
}

struct StagingLease {
	const gec_codec *c;
	Staging st;
	explicit StagingLease(const gec_codec *cc) : c(cc)
	{
		std::lock_guard<std::mutex> g(c->pool_mu);
		if (!c->pool.empty()) {
			st = c->pool.back();
			c->pool.pop_back();
		}
	}
	~StagingLease()
	{
		std::lock_guard<std::mutex> g(c->pool_mu);
		c->pool.push_back(st);
	}
};

// blocks per staging chunk of about `target` bytes (callers pass kChunkBytes or a multiple):
// staging memory stays bounded whatever the batch size.
size_t chunk_blocks(size_t bytes_per_block, size_t nblocks, size_t target)
{
	size_t n = std::max<size_t>(1, target / std::max<size_t>(bytes_per_block, 1));
	return std::min(n, nblocks);
}

constexpr size_t kChunkBytes = 16ull << 20;  // staging chunk: small enough to overlap, big enough to fill the GPU

// chunk size when the caller's memory is pinned end to end (no host staging to bound): GEC_PINNED_CHUNK_MB, default 128
size_t pinned_chunk_bytes()
{
	static const size_t v = [] {
		const char *e = getenv("GEC_PINNED_CHUNK_MB");
		const size_t mb = e ? strtoull(e, nullptr, 0) : 128;  // sweep on MI355X: 16..64 MiB 33-37 GiB/s, 128 MiB 45, 256 MiB 39
		return std::max<size_t>(mb, 1) << 20;
	}();
	return v;
}

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
int run_pipeline(const gec_codec *c, size_t nchunks, size_t slot_bytes, size_t nbad, Fill fill, Enqueue enqueue,
		 Drain drain)
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

}  // namespace

// ---------------------------------------------------------------------------
extern "C" {

uint32_t gec_version(void) { return GEC_VERSION; }

int gec_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess)
		return 0;
	return n;
}

const char *gec_strerror(int code)
{
	switch (code) {
	case GEC_OK: return "ok";
	case GEC_E_TOO_FEW_SHARDS: return "too few shards";
	case GEC_E_TOO_MANY_SHARDS: return "too many shards";
	case GEC_E_TOO_FEW_DATA: return "too few data shards";
	case GEC_E_TOO_MANY_DATA: return "too many data shards";
	case GEC_E_TOO_FEW_PARITY: return "too few parity shards";
	case GEC_E_TOO_MANY_PARITY: return "too many parity shards";
	case GEC_E_INCORRECT_SHARD_SIZE: return "incorrect shard size";
	case GEC_E_TOO_FEW_PRESENT: return "too few shards present";
	case GEC_E_EMPTY_SHARD: return "empty shard";
	case GEC_E_INVALID_INDEX: return "invalid index";
	case GEC_E_DEVICE: return "device (HIP) error";
	case GEC_E_NOMEM: return "out of memory";
	case GEC_E_INVALID_ARG: return "invalid argument";
	default: return "unknown error";
	}
}

const char *gec_last_error(void) { return g_last_error.c_str(); }

size_t gec_shard_len(int k, size_t block_len)
{
	if (k <= 0)
		return 0;
	size_t per = (std::max<size_t>(block_len, 1) + (size_t)k - 1) / (size_t)k;
	return (per + 63) / 64 * 64;
}

static int build_matrix_kind(int k, int m, int matrix, gec::Matrix &enc)
{
	if (matrix == GEC_MATRIX_VANDERMONDE) {
		if (!gec::build_encoding_matrix(k, m, enc))
			return fail(GEC_E_INVALID_ARG, "vandermonde top block singular");
	} else if (matrix == GEC_MATRIX_CAUCHY) {
		gec::build_cauchy_matrix(k, m, enc);
	} else {
		return fail(GEC_E_INVALID_ARG, "unknown matrix family");
	}
	return GEC_OK;
}

int gec_build_matrix_ex(int k, int m, int matrix, uint8_t *out)
{
	int rc = check_km(k, m);
	if (rc)
		return rc;
	if (!out)
		return fail(GEC_E_INVALID_ARG, "NULL output");
	gec::Matrix enc;
	rc = build_matrix_kind(k, m, matrix, enc);
	if (rc)
		return rc;
	std::memcpy(out, enc.v.data(), enc.v.size());
	return GEC_OK;
}

int gec_build_matrix(int k, int m, uint8_t *out) { return gec_build_matrix_ex(k, m, GEC_MATRIX_VANDERMONDE, out); }

int gec_build_decode_matrix(int k, int m, const uint8_t *present, int32_t *valid_out, uint8_t *out)
{
	int rc = check_km(k, m);
	if (rc)
		return rc;
	if (!present || !valid_out || !out)
		return fail(GEC_E_INVALID_ARG, "NULL argument");
	gec::Matrix enc;
	if (!gec::build_encoding_matrix(k, m, enc))
		return fail(GEC_E_INVALID_ARG, "vandermonde top block singular");
	int nv = 0;
	for (int j = 0; j < k + m && nv < k; ++j)
		if (present[j])
			valid_out[nv++] = j;
	if (nv < k)
		return fail(GEC_E_TOO_FEW_PRESENT, "fewer than k shards present");
	gec::Matrix sub(k, k), dec;
	for (int t = 0; t < k; ++t)
		std::memcpy(&sub.at(t, 0), enc.row(valid_out[t]), k);
	if (!gec::invert(sub, dec))
		return fail(GEC_E_INVALID_ARG, "decode sub-matrix singular");
	std::memcpy(out, dec.v.data(), dec.v.size());
	return GEC_OK;
}

int gec_codec_create(int k, int m, int device, gec_codec **out)
{
	return gec_codec_create_ex(k, m, device, GEC_MATRIX_VANDERMONDE, out);
}

int gec_codec_create_ex(int k, int m, int device, int matrix, gec_codec **out)
{
	if (!out)
		return fail(GEC_E_INVALID_ARG, "NULL out");
	*out = nullptr;
	int rc = check_km(k, m);
	if (rc)
		return rc;
	if (matrix != GEC_MATRIX_VANDERMONDE && matrix != GEC_MATRIX_CAUCHY)
		return fail(GEC_E_INVALID_ARG, "unknown matrix family");
	int ndev = 0;
	hipError_t e = hipGetDeviceCount(&ndev);
	if (e != hipSuccess || ndev <= 0)
		return fail(GEC_E_DEVICE, std::string("no HIP device available (libgarage_ec has no CPU fallback): ") +
						  (e != hipSuccess ? hipGetErrorString(e) : "device count is 0"));
	if (device < 0 || device >= ndev)
		return fail(GEC_E_INVALID_ARG, "device index out of range");
	std::unique_ptr<gec_codec> c(new (std::nothrow) gec_codec());
	if (!c)
		return fail(GEC_E_NOMEM, "alloc codec");
	c->k = k;
	c->m = m;
	c->device = device;
	rc = build_matrix_kind(k, m, matrix, c->enc);
	if (rc)
		return rc;
	DeviceGuard g(device);
	if (!g.ok)
		return fail(GEC_E_DEVICE, "hipSetDevice failed");
	hipDeviceProp_t prop;
	HIP_TRY(hipGetDeviceProperties(&prop, device));
	c->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
	gec::LogExp le;
	const gec::Field &f = gec::field();
	std::memcpy(le.exp, f.exp.data(), 512);
	std::memcpy(le.log, f.log.data(), 256);
	HIP_TRY(hipMalloc(reinterpret_cast<void **>(&c->d_logexp), sizeof(le)));
	HIP_TRY(hipMemcpy(c->d_logexp, &le, sizeof(le), hipMemcpyHostToDevice));
	*out = c.release();
	return GEC_OK;
}

void gec_codec_destroy(gec_codec *c)
{
	if (!c)
		return;
	{
		DeviceGuard g(c->device);
		for (auto &s : c->pool)
			s.release();
		for (auto &kv : c->leaf_scratch)
			if (kv.second.p)
				(void)hipFree(kv.second.p);
		if (c->d_logexp)
			(void)hipFree(c->d_logexp);
	}
	delete c;
}

int gec_codec_k(const gec_codec *c) { return c ? c->k : 0; }
int gec_codec_m(const gec_codec *c) { return c ? c->m : 0; }
int gec_codec_device(const gec_codec *c) { return c ? c->device : -1; }

int gec_parity_matrix(const gec_codec *c, uint8_t *out)
{
	if (!c || !out)
		return fail(GEC_E_INVALID_ARG, "NULL argument");
	std::memcpy(out, c->enc.row(c->k), (size_t)c->m * c->k);
	return GEC_OK;
}

int gec_codec_cache_stats(const gec_codec *c, uint64_t *cached, uint64_t *inversions)
{
	if (!c)
		return fail(GEC_E_INVALID_ARG, "NULL codec");
	std::lock_guard<std::mutex> g(c->cache_mu);
	if (cached)
		*cached = c->cache.size();
	if (inversions)
		*inversions = c->inversions;
	return GEC_OK;
}

int gec_launch_geometry(int k, int rows_left, int *rows, int *entry_bytes, int *loads_per_batch, int *threads,
			size_t *lds_bytes)
{
	if (k < 1 || k > GEC_MAX_SHARDS - 1 || rows_left < 1)
		return fail(GEC_E_INVALID_ARG, "need 1 <= k <= 255 and rows_left >= 1");
	const Geometry g = pick_geometry(k, rows_left, g_rows16.load(std::memory_order_relaxed) != 0);
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
	if (variant < 0 || variant > 1)
		return fail(GEC_E_INVALID_ARG, "unknown kernel variant");
	g_variant.store(variant);
	return GEC_OK;
}

int gec_get_kernel_variant(void) { return g_variant.load(); }

// ------------------------------------------------------------ device API
int gec_encode_batch_dev(const gec_codec *c, size_t nblocks, const void *d_data, size_t data_stride, size_t S,
			 void *d_parity, size_t parity_stride, void *hip_stream)
{
	if (!c)
		return fail(GEC_E_INVALID_ARG, "NULL codec");
	if (nblocks == 0)
		return GEC_OK;
	int rc = check_dev_layout(d_data, data_stride, S, (size_t)c->k * S);
	if (rc)
		return rc;
	rc = check_dev_layout(d_parity, parity_stride, S, (size_t)c->m * S);
	if (rc)
		return rc;
	DeviceGuard g(c->device);
	if (!g.ok)
		return fail(GEC_E_DEVICE, "hipSetDevice failed");
	return encode_dev(c, nblocks, static_cast<const uint8_t *>(d_data), data_stride, S,
			  static_cast<uint8_t *>(d_parity), parity_stride, static_cast<hipStream_t>(hip_stream));
}

int gec_verify_batch_dev(const gec_codec *c, size_t nblocks, const void *d_stripes, size_t stride, size_t S,
			 uint32_t *d_bad, void *hip_stream)
{
	if (!c || !d_bad)
		return fail(GEC_E_INVALID_ARG, "NULL argument");
	if (nblocks == 0)
		return GEC_OK;
	int rc = check_dev_layout(d_stripes, stride, S, (size_t)(c->k + c->m) * S);
	if (rc)
		return rc;
	DeviceGuard g(c->device);
	if (!g.ok)
		return fail(GEC_E_DEVICE, "hipSetDevice failed");
	return verify_dev(c, nblocks, static_cast<const uint8_t *>(d_stripes), stride, S, d_bad,
			  static_cast<hipStream_t>(hip_stream));
}

int gec_reconstruct_range_dev(const gec_codec *c, size_t nblocks, void *d_stripes, size_t stride, size_t S,
			      const uint8_t *present, int data_only, size_t byte_off, size_t byte_len,
			      void *hip_stream)
{
	if (!c || !present)
		return fail(GEC_E_INVALID_ARG, "NULL argument");
	if (nblocks == 0)
		return GEC_OK;
	int rc = check_dev_layout(d_stripes, stride, S, (size_t)(c->k + c->m) * S);
	if (rc)
		return rc;
	if (byte_off % 16 || byte_len % 16 || byte_off > S || byte_len > S - byte_off)
		return fail(GEC_E_INVALID_ARG, "byte range must be 16-byte aligned and inside the shard");
	DeviceGuard g(c->device);
	if (!g.ok)
		return fail(GEC_E_DEVICE, "hipSetDevice failed");
	return reconstruct_dev(c, nblocks, static_cast<uint8_t *>(d_stripes), stride, stripe_offsets(c, S).data(), present,
			       data_only != 0, byte_off, byte_len, static_cast<hipStream_t>(hip_stream));
}

int gec_reconstruct_scattered_dev(const gec_codec *c, size_t nblocks, void *d_base, size_t block_stride,
				  const size_t *shard_off, size_t S, const uint8_t *present, int data_only,
				  size_t byte_off, size_t byte_len, void *hip_stream)
{
	if (!c || !present || !shard_off)
		return fail(GEC_E_INVALID_ARG, "NULL argument");
	if (nblocks == 0)
		return GEC_OK;
	int rc = check_dev_layout(d_base, block_stride, S, S);
	if (rc)
		return rc;
	for (int j = 0; j < c->k + c->m; ++j)
		if (shard_off[j] % 16)
			return fail(GEC_E_INVALID_ARG, "shard offsets must be multiples of 16");
	if (byte_off % 16 || byte_len % 16 || byte_off > S || byte_len > S - byte_off)
		return fail(GEC_E_INVALID_ARG, "byte range must be 16-byte aligned and inside the shard");
	DeviceGuard g(c->device);
	if (!g.ok)
		return fail(GEC_E_DEVICE, "hipSetDevice failed");
	return reconstruct_dev(c, nblocks, static_cast<uint8_t *>(d_base), block_stride, shard_off, present,
			       data_only != 0, byte_off, byte_len, static_cast<hipStream_t>(hip_stream));
}

int gec_reconstruct_batch_dev(const gec_codec *c, size_t nblocks, void *d_stripes, size_t stride, size_t S,
			      const uint8_t *present, int data_only, void *hip_stream)
{
	return gec_reconstruct_range_dev(c, nblocks, d_stripes, stride, S, present, data_only, 0, S, hip_stream);
}

// ------------------------------------------------- striped objects over several GPUs
int gec_group_unique_id(uint8_t id[GEC_GROUP_ID_BYTES])
{
	static_assert(sizeof(ncclUniqueId) == GEC_GROUP_ID_BYTES, "GEC_GROUP_ID_BYTES must equal sizeof(ncclUniqueId)");
	if (!id)
		return fail(GEC_E_INVALID_ARG, "NULL id");
	const Rccl &R = rccl();
	if (!R.handle)
		return fail(GEC_E_DEVICE, "RCCL is not available: " + R.error);
	ncclUniqueId u;
	ncclResult_t r = R.GetUniqueId(&u);
	if (r != ncclSuccess)
		return fail(GEC_E_DEVICE, std::string("ncclGetUniqueId: ") + R.GetErrorString(r));
	std::memcpy(id, &u, sizeof(u));
	return GEC_OK;
}

static int group_new(const gec_codec *c, int rank, int nranks, gec_group **out, std::unique_ptr<gec_group> &g)
{
	if (!out)
		return fail(GEC_E_INVALID_ARG, "NULL out");
	*out = nullptr;
	if (!c)
		return fail(GEC_E_INVALID_ARG, "NULL codec");
	if (nranks < 1 || rank < 0 || rank >= nranks)
		return fail(GEC_E_INVALID_ARG, "need 0 <= rank < nranks");
	g.reset(new (std::nothrow) gec_group());
	if (!g)
		return fail(GEC_E_NOMEM, "alloc group");
	g->c = c;
	g->rank = rank;
	g->nranks = nranks;
	return GEC_OK;
}

int gec_group_create(const gec_codec *c, int rank, int nranks, const uint8_t id[GEC_GROUP_ID_BYTES], gec_group **out)
{
	std::unique_ptr<gec_group> g;
	int rc = group_new(c, rank, nranks, out, g);
	if (rc)
		return rc;
	if (!id)
		return fail(GEC_E_INVALID_ARG, "NULL id");
	const Rccl &R = rccl();
	if (!R.handle)
		return fail(GEC_E_DEVICE, "RCCL is not available: " + R.error);
	DeviceGuard dg(c->device);
	if (!dg.ok)
		return fail(GEC_E_DEVICE, "hipSetDevice failed");
	ncclUniqueId u;
	std::memcpy(&u, id, sizeof(u));
	ncclResult_t r = R.CommInitRank(&g->comm, nranks, u, rank);
	if (r != ncclSuccess)
		return fail(GEC_E_DEVICE, std::string("ncclCommInitRank: ") + R.GetErrorString(r));
	g->all_gather = rccl_all_gather;
	g->all_to_all = rccl_all_to_all;
	g->ctx = g.get();
	*out = g.release();
	return GEC_OK;
}

int gec_group_create_with_transport2(const gec_codec *c, int rank, int nranks, gec_allgather_fn all_gather,
				     gec_alltoall_fn all_to_all, void *ctx, gec_group **out)
{
	std::unique_ptr<gec_group> g;
	int rc = group_new(c, rank, nranks, out, g);
	if (rc)
		return rc;
	if (!all_gather)
		return fail(GEC_E_INVALID_ARG, "NULL all_gather");
	g->all_gather = all_gather;
	g->all_to_all = all_to_all;
	g->ctx = ctx;
	*out = g.release();
	return GEC_OK;
}

int gec_group_create_with_transport(const gec_codec *c, int rank, int nranks, gec_allgather_fn all_gather, void *ctx,
				    gec_group **out)
{
	return gec_group_create_with_transport2(c, rank, nranks, all_gather, nullptr, ctx, out);
}

void gec_group_destroy(gec_group *g)
{
	if (!g)
		return;
	{
		DeviceGuard dg(g->c->device);
		if (g->d_send)
			(void)hipFree(g->d_send);
		if (g->d_recv)
			(void)hipFree(g->d_recv);
		if (g->d_a2a_send)
			(void)hipFree(g->d_a2a_send);
		if (g->d_a2a_recv)
			(void)hipFree(g->d_a2a_recv);
		if (g->comm)
			(void)rccl().CommDestroy(g->comm);
	}
	delete g;
}

int gec_group_rank(const gec_group *g) { return g ? g->rank : -1; }
int gec_group_size(const gec_group *g) { return g ? g->nranks : 0; }
size_t gec_group_slots(const gec_group *g)
{
	return g ? ((size_t)(g->c->k + g->c->m) + g->nranks - 1) / g->nranks : 0;
}

int gec_group_allgather_decode(gec_group *g, size_t nobjects, const void *d_local_slots, size_t S,
			       const uint8_t *present, int data_only, int complete, void *d_gathered, void *hip_stream)
{
	if (!g || !present)
		return fail(GEC_E_INVALID_ARG, "NULL argument");
	if (nobjects == 0)
		return GEC_OK;
	const gec_codec *c = g->c;
	const size_t n = (size_t)c->k + c->m, N = (size_t)g->nranks, slots = gec_group_slots(g);
	int rc = check_dev_layout(d_local_slots, slots * S, S, slots * S);
	if (rc)
		return rc;
	rc = check_dev_layout(d_gathered, slots * S, S, slots * S);
	if (rc)
		return rc;
	hipStream_t stream = static_cast<hipStream_t>(hip_stream);
	DeviceGuard dg(c->device);
	if (!dg.ok)
		return fail(GEC_E_DEVICE, "hipSetDevice failed");
	std::shared_ptr<const Plan> plan;  // before the exchange: a bad pattern fails on every rank alike, no rank hangs
	rc = get_plan(c, present, data_only != 0, plan);
	if (rc)
		return rc;
	// (1) the exchange step: every rank's slot buffer to everybody
	const size_t per_rank = nobjects * slots * S;
	g->bytes_exchanged = per_rank * (N - 1);
	rc = g->all_gather(g->ctx, d_local_slots, d_gathered, per_rank, hip_stream);
	if (rc)
		return rc > 0 ? fail(GEC_E_DEVICE, "all_gather transport failed") : rc;
	if (plan->missing.empty())
		return GEC_OK;
	// (2) my byte range of every missing shard, in place in the gathered buffer
	std::vector<size_t> shard_off(n);
	for (size_t j = 0; j < n; ++j)
		shard_off[j] = (j % N) * per_rank + (j / N) * S;
	const size_t cols = S / 16;
	auto range_lo = [&](size_t r) { return cols * r / N; };
	const size_t lo = range_lo(g->rank), my_cols = range_lo(g->rank + 1) - lo;
	if (my_cols) {
		rc = reconstruct_dev(c, nobjects, static_cast<uint8_t *>(d_gathered), slots * S, shard_off.data(), present,
				     data_only != 0, lo * 16, my_cols * 16, stream);
		if (rc)
			return rc;
	}
	if (!complete || N == 1)
		return GEC_OK;
	// (3) exchange the rebuilt ranges (ranges differ by at most one column: pad to the longest)
	size_t max_cols = 0;
	for (size_t r = 0; r < N; ++r)
		max_cols = std::max(max_cols, range_lo(r + 1) - range_lo(r));
	const size_t nmiss = plan->missing.size();
	const size_t send_bytes = nmiss * nobjects * max_cols * 16;
	if (nobjects > 0xffffffffull || cols > 0xffffffffull)
		return fail(GEC_E_INVALID_ARG, "batch too large for one call");
	if (send_bytes > g->send_cap || send_bytes * N > g->recv_cap) {
		HIP_TRY(hipStreamSynchronize(stream));  // earlier calls may still use the old buffers
		if (g->d_send)
			(void)hipFree(g->d_send);
		if (g->d_recv)
			(void)hipFree(g->d_recv);
		g->d_send = g->d_recv = nullptr;
		g->send_cap = g->recv_cap = 0;
		HIP_TRY(hipMalloc(reinterpret_cast<void **>(&g->d_send), send_bytes));
		HIP_TRY(hipMalloc(reinterpret_cast<void **>(&g->d_recv), send_bytes * N));
		HIP_TRY(hipMemsetAsync(g->d_send, 0, send_bytes, stream));  // pad columns: defined bytes on the wire
		g->send_cap = send_bytes;
		g->recv_cap = send_bytes * N;
	}
	gec::RangeArgs ra;
	std::memset(&ra, 0, sizeof(ra));
	ra.gathered = static_cast<uint8_t *>(d_gathered);
	ra.obj_stride = slots * S;
	ra.nobj = (uint32_t)nobjects;
	ra.nmiss = (uint32_t)nmiss;
	ra.cols = (uint32_t)cols;
	ra.max_cols = (uint32_t)max_cols;
	ra.world = (uint32_t)N;
	ra.rank = (uint32_t)g->rank;
	for (size_t i = 0; i < nmiss; ++i)
		ra.shard_off[i] = shard_off[plan->missing[i]];
	auto grid_for = [](size_t items) { return (unsigned)std::min<size_t>((items + 255) / 256, 1u << 16); };
	if (my_cols) {
		ra.packed = g->d_send;
		hipLaunchKernelGGL(gec::range_pack, dim3(grid_for(nmiss * nobjects * my_cols)), dim3(256), 0, stream, ra);
		HIP_TRY(hipGetLastError());
	}
	g->bytes_exchanged += send_bytes * (N - 1);
	rc = g->all_gather(g->ctx, g->d_send, g->d_recv, send_bytes, hip_stream);
	if (rc)
		return rc > 0 ? fail(GEC_E_DEVICE, "all_gather transport failed") : rc;
	ra.packed = g->d_recv;
	hipLaunchKernelGGL(gec::range_unpack, dim3(grid_for(send_bytes / 16 * N)), dim3(256), 0, stream, ra);
	HIP_TRY(hipGetLastError());
	return GEC_OK;
}

// ------------------------------------------------------------ pinned host memory
void *gec_host_alloc(size_t bytes)
{
	void *p = nullptr;
	// portable: usable by every device's DMA engines (one process may drive several codecs)
	hipError_t e = hipHostMalloc(&p, std::max<size_t>(bytes, 1), hipHostMallocPortable);
	if (e != hipSuccess) {
		fail(e == hipErrorOutOfMemory ? GEC_E_NOMEM : GEC_E_DEVICE, std::string("hipHostMalloc: ") + hipGetErrorString(e));
		return nullptr;
	}
	pinned().add(p, std::max<size_t>(bytes, 1), true);
	return p;
}

void gec_host_free(void *p)
{
	bool owned = false;
	if (p && pinned().remove(p, owned) && owned)
		(void)hipHostFree(p);
}

int gec_host_register(void *p, size_t bytes)
{
	if (!p || bytes == 0)
		return fail(GEC_E_INVALID_ARG, "NULL / empty range");
	HIP_TRY(hipHostRegister(p, bytes, hipHostRegisterPortable | hipHostRegisterMapped));
	void *dptr = p;
	if (hipHostGetDevicePointer(&dptr, p, 0) != hipSuccess || !dptr)
		dptr = p;
	pinned().add(p, bytes, false, reinterpret_cast<intptr_t>(dptr) - reinterpret_cast<intptr_t>(p));
	return GEC_OK;
}

int gec_host_unregister(void *p)
{
	bool owned = false;
	if (!p || !pinned().remove(p, owned))
		return fail(GEC_E_INVALID_ARG, "not a registered range");
	if (owned) {  // it came from gec_host_alloc: treat like gec_host_free
		(void)hipHostFree(p);
		return GEC_OK;
	}
	HIP_TRY(hipHostUnregister(p));
	return GEC_OK;
}

int gec_host_is_pinned(const void *p, size_t bytes) { return pinned().contains(p, bytes) ? 1 : 0; }

uint64_t gec_group_bytes_exchanged(const gec_group *g) { return g ? g->bytes_exchanged : 0; }

int gec_group_alltoall_decode(gec_group *g, size_t nobjects, const void *d_local_slots, size_t S, const uint8_t *present,
			      int data_only, int complete, void *d_rebuilt, void *hip_stream)
{
	if (!g || !present || !d_rebuilt)
		return fail(GEC_E_INVALID_ARG, "NULL argument");
	if (!g->all_to_all)
		return fail(GEC_E_INVALID_ARG, "this group's transport has no all-to-all");
	if (nobjects == 0)
		return GEC_OK;
	const gec_codec *c = g->c;
	const size_t k = c->k, N = (size_t)g->nranks, slots = gec_group_slots(g);
	int rc = check_dev_layout(d_local_slots, slots * S, S, slots * S);
	if (rc)
		return rc;
	if (reinterpret_cast<uintptr_t>(d_rebuilt) % 16)
		return fail(GEC_E_INVALID_ARG, "d_rebuilt must be 16-byte aligned");
	if (nobjects > 0xffffffffull || S / 16 > 0xffffffffull)
		return fail(GEC_E_INVALID_ARG, "batch too large for one call");
	hipStream_t stream = static_cast<hipStream_t>(hip_stream);
	DeviceGuard dg(c->device);
	if (!dg.ok)
		return fail(GEC_E_DEVICE, "hipSetDevice failed");
	std::shared_ptr<const Plan> plan;  // before the exchange: a bad pattern fails on every rank alike, no rank hangs
	rc = get_plan(c, present, data_only != 0, plan);
	if (rc)
		return rc;
	g->bytes_exchanged = 0;
	const size_t nmiss = plan->missing.size();
	if (nmiss == 0)
		return GEC_OK;
	// which of the k shards the decode reads live on which rank: shard v on rank v % N, local slot v / N;
	// vs index = position among that rank's valid shards
	std::vector<std::vector<int>> valid_of(N);
	for (size_t t = 0; t < k; ++t)
		valid_of[plan->valid[t] % N].push_back(plan->valid[t]);
	size_t nvs_max = 0;
	for (auto &v : valid_of)
		nvs_max = std::max(nvs_max, v.size());
	const size_t cols = S / 16;
	auto range_lo = [&](size_t r) { return cols * r / N; };
	size_t max_cols = 0;
	for (size_t r = 0; r < N; ++r)
		max_cols = std::max(max_cols, range_lo(r + 1) - range_lo(r));
	const size_t my_cols = range_lo(g->rank + 1) - range_lo(g->rank);
	const size_t per_peer = nvs_max * nobjects * max_cols * 16;
	const size_t packed_bytes = nmiss * nobjects * max_cols * 16;
	// scratch: [send N*per_peer][recv N*per_peer]; the rebuilt ranges reuse the group's d_send / d_recv
	if (N * per_peer > g->a2a_cap || packed_bytes > g->send_cap || packed_bytes * N > g->recv_cap) {
		HIP_TRY(hipStreamSynchronize(stream));
		for (uint8_t **p : {&g->d_a2a_send, &g->d_a2a_recv, &g->d_send, &g->d_recv})
			if (*p) {
				(void)hipFree(*p);
				*p = nullptr;
			}
		g->a2a_cap = g->send_cap = g->recv_cap = 0;
		HIP_TRY(hipMalloc(reinterpret_cast<void **>(&g->d_a2a_send), N * per_peer));
		HIP_TRY(hipMalloc(reinterpret_cast<void **>(&g->d_a2a_recv), N * per_peer));
		HIP_TRY(hipMalloc(reinterpret_cast<void **>(&g->d_send), packed_bytes));
		HIP_TRY(hipMalloc(reinterpret_cast<void **>(&g->d_recv), packed_bytes * N));
		HIP_TRY(hipMemsetAsync(g->d_a2a_send, 0, N * per_peer, stream));  // pad columns / unused vs slots: defined bytes on the wire
		HIP_TRY(hipMemsetAsync(g->d_send, 0, packed_bytes, stream));
		g->a2a_cap = N * per_peer;
		g->send_cap = packed_bytes;
		g->recv_cap = packed_bytes * N;
	}
	auto grid_for = [](size_t items) { return (unsigned)std::max<size_t>(1, std::min<size_t>((items + 255) / 256, 1u << 16)); };
	// (1) pack: for every peer, that peer's byte range of my valid shards
	const std::vector<int> &mine = valid_of[g->rank];
	if (!mine.empty()) {
		gec::A2aArgs pa;
		std::memset(&pa, 0, sizeof(pa));
		pa.local = static_cast<const uint8_t *>(d_local_slots);
		pa.send = g->d_a2a_send;
		pa.obj_stride = slots * S;
		pa.nobj = (uint32_t)nobjects;
		pa.nvs = (uint32_t)mine.size();
		pa.nvs_max = (uint32_t)nvs_max;
		pa.cols = (uint32_t)cols;
		pa.max_cols = (uint32_t)max_cols;
		pa.world = (uint32_t)N;
		for (size_t i = 0; i < mine.size(); ++i)
			pa.slot_of[i] = (uint32_t)(mine[i] / N);
		hipLaunchKernelGGL(gec::a2a_pack, dim3(grid_for(N * mine.size() * nobjects * max_cols)), dim3(256), 0, stream, pa);
		HIP_TRY(hipGetLastError());
	}
	// (2) the exchange step: 1/N of the all-gather's bytes
	g->bytes_exchanged = per_peer * (N - 1);
	rc = g->all_to_all(g->ctx, g->d_a2a_send, g->d_a2a_recv, per_peer, hip_stream);
	if (rc)
		return rc > 0 ? fail(GEC_E_DEVICE, "all_to_all transport failed") : rc;
	// (3) my byte range of every missing shard, from the received ranges: input shard valid[t] sits at
	//     recv[(owner*nvs_max + vs)*nobj + obj][max_cols]; output i at d_send[(i*nobj + obj)][max_cols]
	if (my_cols) {
		std::vector<size_t> in_off(k), out_off(nmiss);
		for (size_t t = 0; t < k; ++t) {
			const int v = plan->valid[t];
			const size_t owner = v % N;
			const size_t vs = std::find(valid_of[owner].begin(), valid_of[owner].end(), v) - valid_of[owner].begin();
			in_off[t] = (owner * nvs_max + vs) * nobjects * max_cols * 16;
		}
		for (size_t i = 0; i < nmiss; ++i)
			out_off[i] = i * nobjects * max_cols * 16;
		rc = launch_apply(c, g->d_a2a_recv, max_cols * 16, g->d_send, max_cols * 16, nullptr, 0, my_cols * 16, nobjects,
				  in_off.data(), out_off.data(), (int)nmiss, plan->rows.v.data(), gec::MODE_STORE, stream);
		if (rc)
			return rc;
	}
	// (4) the rebuilt ranges: mine only, or everybody's after a (small) all-gather
	gec::RebuiltArgs ua;
	std::memset(&ua, 0, sizeof(ua));
	ua.rebuilt = static_cast<uint8_t *>(d_rebuilt);
	ua.nobj = (uint32_t)nobjects;
	ua.nmiss = (uint32_t)nmiss;
	ua.cols = (uint32_t)cols;
	ua.max_cols = (uint32_t)max_cols;
	ua.world = (uint32_t)N;
	if (complete && N > 1) {
		g->bytes_exchanged += packed_bytes * (N - 1);
		rc = g->all_gather(g->ctx, g->d_send, g->d_recv, packed_bytes, hip_stream);
		if (rc)
			return rc > 0 ? fail(GEC_E_DEVICE, "all_gather transport failed") : rc;
		ua.packed = g->d_recv;
		ua.first_rank = 0;
		ua.nranks_in = (uint32_t)N;
	} else {
		ua.packed = g->d_send;
		ua.first_rank = (uint32_t)g->rank;
		ua.nranks_in = 1;
	}
	hipLaunchKernelGGL(gec::rebuilt_unpack, dim3(grid_for((size_t)ua.nranks_in * nmiss * nobjects * max_cols)), dim3(256), 0, stream, ua);
	HIP_TRY(hipGetLastError());
	return GEC_OK;
}

// -------------------------------------------------------- host-pointer API
static int encode_batch_impl(const gec_codec *c, size_t nblocks, const uint8_t *const *blocks, const size_t *block_len,
			     size_t S, uint8_t *const *parity, uint8_t *shard_sums)
{
	if (!c)
		return fail(GEC_E_INVALID_ARG, "NULL codec");
	if (nblocks == 0)
		return GEC_OK;
	if (!blocks || !block_len || !parity)
		return fail(GEC_E_INVALID_ARG, "NULL argument");
	if (S == 0)
		return fail(GEC_E_EMPTY_SHARD, "shard length is 0");
	if (S % 64)
		return fail(GEC_E_INCORRECT_SHARD_SIZE, "S must be a multiple of 64");
	const size_t k = c->k, m = c->m, n = k + m;
	for (size_t b = 0; b < nblocks; ++b) {
		if (!blocks[b] || !parity[b])
			return fail(GEC_E_INVALID_ARG, "NULL block/parity pointer");
		if (block_len[b] > k * S)
			return fail(GEC_E_INCORRECT_SHARD_SIZE, "block longer than k*S");
	}
	const size_t stripe = n * S;
	// the hash kernel's serial chain costs ~3.5 ms per launch whatever the batch, so
	// chunks are 8x larger when checksums are requested
	// caller memory that is pinned end to end needs no host staging, so the only reasons to chunk are the size of
	// the device buffer and the overlap of copy-in / kernels / copy-out between the two slots: 128 MiB chunks
	bool all_pinned = true;
	for (size_t b = 0; b < nblocks && all_pinned; ++b)
		all_pinned = aligned16(blocks[b]) && aligned16(parity[b]) && pinned().contains(blocks[b], block_len[b]) &&
			     pinned().contains(parity[b], m * S);
	if (all_pinned && k <= (size_t)gec::PTR_KMAX && zero_copy_enabled()) {
		// every buffer is device-addressable: ONE kernel reads the data shards and writes the parity in place
		// over the link; nothing is staged in HBM, no host copy.  With checksums requested the same kernel also
		// lays everything it reads and computes down in HBM (the bytes still cross the link once), chunk by
		// chunk on two streams, and each chunk's shard checksums are computed from there while the next chunk
		// is on the link.
		DeviceGuard dg(c->device);
		if (!dg.ok)
			return fail(GEC_E_DEVICE, "hipSetDevice failed");
		StagingLease lease(c);
		Staging &st = lease.st;
		const size_t zch = shard_sums ? chunk_blocks(stripe, nblocks, 8 * kChunkBytes) : nblocks;
		const size_t nz = (nblocks + zch - 1) / zch;
		int rc = st.ensure(shard_sums ? nblocks * n * 32 + 64 : 64, 0);
		if (!rc)
			rc = st.ensure_tab((nblocks * (k * 12 + m * 8) + 64 * nz) / sizeof(gec::CopyEntry) + 4 * nz + 4);
		if (!rc && shard_sums)
			rc = st.ensure_big(2 * (zch * stripe + zch * n * 32));
		if (!rc && shard_sums)
			rc = st.ensure_segments(c->num_cu);
		if (rc)
			return rc;
		// with checksums: the link kernel on a few CUs of its own, the checksum kernels on the rest (a kernel whose
		// loads share a CU with microsecond-long host reads crawls, see Staging::stream_up); chunk ci+2 reuses the
		// mirror of chunk ci, so its link kernel waits for that chunk's checksums
		hipStream_t up = shard_sums && st.stream_up ? st.stream_up : st.stream;
		hipStream_t chain = shard_sums && st.stream_chain ? st.stream_chain : st.stream2;
		std::vector<const uint8_t *> in(zch * k);
		std::vector<uint32_t> valid(zch * k);
		std::vector<uint8_t *> out(zch * m);
		for (size_t ci = 0; ci < nz && !rc; ++ci) {
			const size_t b0 = ci * zch, nb = std::min(zch, nblocks - b0);
			for (size_t i = 0; i < nb; ++i) {
				const uint8_t *p = pinned().dev(blocks[b0 + i]);
				uint8_t *q = pinned().dev(parity[b0 + i]);
				const size_t len = block_len[b0 + i];
				for (size_t t = 0; t < k; ++t) {
					in[i * k + t] = p + t * S;
					valid[i * k + t] = (uint32_t)(len > t * S ? std::min(S, len - t * S) : 0);
				}
				for (size_t r = 0; r < m; ++r)
					out[i * m + r] = q + r * S;
			}
			uint8_t *mir = shard_sums ? st.d_big + (ci & 1) * (zch * stripe + zch * n * 32) : nullptr;
			if (shard_sums && ci >= 2 && hipStreamWaitEvent(up, st.ev_seg[2 + (ci & 1)], 0) != hipSuccess)
				rc = fail(GEC_E_DEVICE, "hipStreamWaitEvent");
			if (!rc)
				rc = launch_apply_ptrs(c, st, nb, in.data(), valid.data(), out.data(), (int)m, S, c->enc.row(k), up, mir);
			if (rc || !shard_sums)
				continue;
			uint8_t *d_sums = mir + zch * stripe;
			hipError_t e = hipEventRecord(st.ev_seg[ci & 1], up);
			if (e == hipSuccess)
				e = hipStreamWaitEvent(chain, st.ev_seg[ci & 1], 0);
			if (e != hipSuccess) {
				rc = fail(GEC_E_DEVICE, "chunk event");
				continue;
			}
			rc = blake2_dev(c, nb * n, mir, nullptr, nullptr, S, S, d_sums, chain, 0, 0, 0, true);
			if (!rc && hipMemcpyAsync(st.h_buf + b0 * n * 32, d_sums, nb * n * 32, hipMemcpyDeviceToHost, chain) != hipSuccess)
				rc = fail(GEC_E_DEVICE, "hipMemcpyAsync (shard sums)");
			if (!rc && hipEventRecord(st.ev_seg[2 + (ci & 1)], chain) != hipSuccess)
				rc = fail(GEC_E_DEVICE, "hipEventRecord");
		}
		const hipError_t e1 = hipStreamSynchronize(up), e2 = shard_sums ? hipStreamSynchronize(chain) : hipSuccess;
		if (rc)
			return rc;
		HIP_TRY(e1);
		HIP_TRY(e2);
		if (shard_sums)
			std::memcpy(shard_sums, st.h_buf, nblocks * n * 32);
		return GEC_OK;
	}
	const size_t ch = chunk_blocks(stripe, nblocks, shard_sums ? 8 * kChunkBytes : all_pinned ? pinned_chunk_bytes() : kChunkBytes);
	const size_t sums_off = ch * stripe;  // checksum area behind the stripes of a slot
	const size_t nchunks = (nblocks + ch - 1) / ch;
	CopyPool &pool = c->copy_pool();
	// per chunk: are all its blocks / all its parity buffers in pinned memory the caller registered?
	// Then the DMA engines read / write the caller's memory directly and the staging copy is skipped.
	std::vector<uint8_t> in_pinned(nchunks, 1), out_pinned(nchunks, 1);
	std::vector<size_t> min_len(nchunks, k * S);
	PipeChain chain;
	for (size_t b = 0; b < nblocks; ++b) {
		const size_t ci = b / ch;
		if (in_pinned[ci] && !(aligned16(blocks[b]) && pinned().contains(blocks[b], block_len[b])))
			in_pinned[ci] = 0;
		if (out_pinned[ci] && !(aligned16(parity[b]) && pinned().contains(parity[b], m * S)))
			out_pinned[ci] = 0;
		min_len[ci] = std::min(min_len[ci], block_len[b]);
	}
	return run_pipeline(
		c, nchunks, ch * stripe + (shard_sums ? ch * n * 32 : 0), 0,
		[&](size_t ci, Staging &st) {  // host: user blocks -> pinned, zero-padded to k*S
			(void)st.ensure_tab(2 * ch);  // a failure shows up as "copy table overflow" when the table is used
			if (in_pinned[ci])
				return;
			const size_t b0 = ci * ch, nb = std::min(ch, nblocks - b0);
			pool.parallel_for(nb, [&](size_t i) {
				uint8_t *dst = st.h_buf + i * stripe;
				const size_t len = block_len[b0 + i];
				std::memcpy(dst, blocks[b0 + i], len);
				std::memset(dst + len, 0, k * S - len);
			});
		},
		[&](size_t ci, Staging &st) -> int {  // device: only data shards go H2D, only parity (+sums) comes back
			const size_t b0 = ci * ch, nb = std::min(ch, nblocks - b0);
			if (in_pinned[ci]) {
				// zero padding behind the shortest block of the chunk first (stream order), then
				// every block straight from the caller's memory
				if (min_len[ci] < k * S)
					HIP_TRY(hipMemset2DAsync(st.d_buf + min_len[ci], stripe, 0, k * S - min_len[ci], nb, st.stream));
				// ONE copy_table launch: the kernel reads every block straight from the caller's pinned memory
				std::vector<gec::CopyEntry> ents;
				ents.reserve(nb);
				for (size_t i = 0; i < nb; ++i)
					if (block_len[b0 + i])
						ents.push_back({pinned().dev(blocks[b0 + i]), st.d_buf + i * stripe, block_len[b0 + i]});
				int rct = chain.before(chain.last_in, st.stream);
				if (!rct)
					rct = launch_copy_table(st, ents, st.stream);
				if (!rct)
					rct = chain.after_in(st);
				if (rct)
					return rct;
			} else {
				HIP_TRY(hipMemcpy2DAsync(st.d_buf, stripe, st.h_buf, stripe, k * S, nb, hipMemcpyHostToDevice, st.stream));
			}
			int rc = shard_sums ? encode_hash_dev(c, nb, st.d_buf, stripe, S, st.d_buf + sums_off, st.stream, st)
					    : encode_dev(c, nb, st.d_buf, stripe, S, st.d_buf + k * S, stripe, st.stream);
			if (rc)
				return rc;
			if (out_pinned[ci]) {
				std::vector<gec::CopyEntry> ents;
				ents.reserve(nb);
				for (size_t i = 0; i < nb; ++i)
					ents.push_back({st.d_buf + i * stripe + k * S, pinned().dev(parity[b0 + i]), m * S});
				int rct = chain.before(chain.last_out, st.stream);
				if (!rct)
					rct = launch_copy_table(st, ents, st.stream);
				if (!rct)
					rct = chain.after_out(st);
				if (rct)
					return rct;
			} else {
				HIP_TRY(hipMemcpy2DAsync(st.h_buf + k * S, stripe, st.d_buf + k * S, stripe, m * S, nb, hipMemcpyDeviceToHost, st.stream));
			}
			if (shard_sums)
				HIP_TRY(hipMemcpyAsync(st.h_buf + sums_off, st.d_buf + sums_off, nb * n * 32, hipMemcpyDeviceToHost, st.stream));
			return GEC_OK;
		},
		[&](size_t ci, Staging &st) {  // host: parity (+sums) -> user buffers
			const size_t b0 = ci * ch, nb = std::min(ch, nblocks - b0);
			if (!out_pinned[ci])
				pool.parallel_for(nb, [&](size_t i) { std::memcpy(parity[b0 + i], st.h_buf + i * stripe + k * S, m * S); });
			if (shard_sums)
				std::memcpy(shard_sums + b0 * n * 32, st.h_buf + sums_off, nb * n * 32);
		});
}

int gec_encode_batch(const gec_codec *c, size_t nblocks, const uint8_t *const *blocks, const size_t *block_len,
		     size_t S, uint8_t *const *parity)
{
	return encode_batch_impl(c, nblocks, blocks, block_len, S, parity, nullptr);
}

int gec_encode_hash_batch(const gec_codec *c, size_t nblocks, const uint8_t *const *blocks, const size_t *block_len,
			  size_t S, uint8_t *const *parity, uint8_t *shard_sums)
{
	if (!shard_sums)
		return fail(GEC_E_INVALID_ARG, "NULL shard_sums");
	return encode_batch_impl(c, nblocks, blocks, block_len, S, parity, shard_sums);
}

int gec_encode_hash_batch_dev(const gec_codec *c, size_t nblocks, void *d_stripes, size_t stride, size_t S, void *d_sums,
			      void *hip_stream)
{
	if (!c || !d_sums)
		return fail(GEC_E_INVALID_ARG, "NULL argument");
	if (nblocks == 0)
		return GEC_OK;
	int rc = check_dev_layout(d_stripes, stride, S, (size_t)(c->k + c->m) * S);
	if (rc)
		return rc;
	if (reinterpret_cast<uintptr_t>(d_sums) % 16)
		return fail(GEC_E_INVALID_ARG, "d_sums must be 16-byte aligned");
	DeviceGuard g(c->device);
	if (!g.ok)
		return fail(GEC_E_DEVICE, "hipSetDevice failed");
	StagingLease aux(c);  // only its partner stream and events are used; enqueued work may outlive the lease
	rc = aux.st.ensure(0, 0);
	if (rc)
		return rc;
	return encode_hash_dev(c, nblocks, static_cast<uint8_t *>(d_stripes), stride, S, static_cast<uint8_t *>(d_sums),
			       static_cast<hipStream_t>(hip_stream), aux.st);
}

static int hash_batch_dev_impl(const gec_codec *c, size_t n, const void *d_base, size_t stride, size_t len, void *d_out,
			       void *hip_stream, int tree)
{
	if (!c || (n && (!d_base || !d_out)))
		return fail(GEC_E_INVALID_ARG, "NULL argument");
	if (reinterpret_cast<uintptr_t>(d_base) % 16 || stride % 16 || reinterpret_cast<uintptr_t>(d_out) % 16)
		return fail(GEC_E_INVALID_ARG, "device pointers/stride must be 16-byte aligned");
	if (n > 1 && stride < len)
		return fail(GEC_E_INVALID_ARG, "stride smaller than the message length");
	DeviceGuard g(c->device);
	if (!g.ok)
		return fail(GEC_E_DEVICE, "hipSetDevice failed");
	return blake2_dev(c, n, static_cast<const uint8_t *>(d_base), nullptr, nullptr, stride, len,
			  static_cast<uint8_t *>(d_out), static_cast<hipStream_t>(hip_stream), 0, 0, 0, tree != 0, len);
}

int gec_blake2sum_batch_dev(const gec_codec *c, size_t n, const void *d_base, size_t stride, size_t len, void *d_out,
			    void *hip_stream)
{
	return hash_batch_dev_impl(c, n, d_base, stride, len, d_out, hip_stream, 0);
}

int gec_shardsum_batch_dev(const gec_codec *c, size_t n, const void *d_base, size_t stride, size_t len, void *d_out,
			   void *hip_stream)
{
	return hash_batch_dev_impl(c, n, d_base, stride, len, d_out, hip_stream, 1);
}

static int hash_batch_impl(const gec_codec *c, size_t n, const uint8_t *const *msgs, const size_t *lens, uint8_t *out, bool tree)
{
	if (!c)
		return fail(GEC_E_INVALID_ARG, "NULL codec");
	if (n == 0)
		return GEC_OK;
	if (!msgs || !lens || !out)
		return fail(GEC_E_INVALID_ARG, "NULL argument");
	for (size_t i = 0; i < n; ++i)
		if (!msgs[i] && lens[i])
			return fail(GEC_E_INVALID_ARG, "NULL message pointer");
	size_t longest = 0;
	bool all_pinned = true;
	for (size_t i = 0; i < n; ++i) {
		longest = std::max(longest, lens[i]);
		if (all_pinned && lens[i] && !(aligned16(msgs[i]) && pinned().contains(msgs[i], lens[i])))
			all_pinned = false;
	}
	// (a) every message in pinned, 16-byte aligned caller memory: NO copy at all -- ONE launch whose lanes
	//     stream their messages straight from host memory over PCIe; only the (offset, length) table and the
	//     32-byte results go through a staging slot.
	// (b) long messages (a BLAKE2b chain costs ~4000 cycles per 128-byte block however many messages run
	//     beside it: 14 ms per MiB): everything is first moved into ONE device buffer through two pinned
	//     staging pieces, then hashed by ONE launch -- chunked launches would pay the chain once per chunk.
	if (all_pinned && tree && zero_copy_enabled() && n > 1) {
		// Shard checksums of pinned messages: lanes that each stream a 4 KiB leaf out of host memory read the link in
		// 16-byte pieces (22 GiB/s); a copy kernel moves the same bytes coalesced at the link's rate, so the messages
		// go to HBM chunk by chunk (copy_table on the upload stream's CUs) and are hashed there beside the next
		// chunk's transfer.
		DeviceGuard dg(c->device);
		if (!dg.ok)
			return fail(GEC_E_DEVICE, "hipSetDevice failed");
		StagingLease lease(c);
		Staging &st = lease.st;
		const size_t kChunk = 8 * kChunkBytes;
		const size_t buf_bytes = std::max(kChunk, (longest + 15) / 16 * 16);
		int rc = st.ensure(n * 48 + 64, 0);  // [off][len][out]
		if (!rc)
			rc = st.ensure_tab(n + 8);
		if (!rc)
			rc = st.ensure_big(2 * buf_bytes);
		if (!rc)
			rc = st.ensure_segments(c->num_cu);
		if (rc)
			return rc;
		hipStream_t up = st.stream_up ? st.stream_up : st.stream;
		hipStream_t chain = st.stream_chain ? st.stream_chain : st.stream2;
		uint64_t *h_off = reinterpret_cast<uint64_t *>(st.h_buf), *h_len = h_off + n;
		uint8_t *h_out = st.h_buf + n * 16;
		size_t ci = 0;
		for (size_t i = 0; i < n && !rc; ++ci) {
			uint8_t *buf = st.d_big + (ci & 1) * buf_bytes;
			std::vector<gec::CopyEntry> ents;
			size_t j = i, bytes = 0, chunk_longest = 0;
			while (j < n && (j == i || bytes + (lens[j] + 15) / 16 * 16 <= buf_bytes)) {
				h_off[j] = bytes;
				h_len[j] = lens[j];
				if (lens[j])
					ents.push_back({pinned().dev(msgs[j]), buf + bytes, lens[j]});
				chunk_longest = std::max(chunk_longest, lens[j]);
				bytes += (lens[j] + 15) / 16 * 16;
				++j;
			}
			if (ci >= 2 && hipStreamWaitEvent(up, st.ev_seg[2 + (ci & 1)], 0) != hipSuccess)
				rc = fail(GEC_E_DEVICE, "hipStreamWaitEvent");
			if (!rc)
				rc = launch_copy_table(st, ents, up);
			hipError_t e = rc ? hipSuccess : hipEventRecord(st.ev_seg[ci & 1], up);
			if (!rc && e == hipSuccess)
				e = hipStreamWaitEvent(chain, st.ev_seg[ci & 1], 0);
			if (!rc && e != hipSuccess)
				rc = fail(GEC_E_DEVICE, "chunk event");
			if (!rc)
				rc = blake2_dev(c, j - i, buf, h_off + i, h_len + i, 0, 0, h_out + 32 * i, chain, 0, 0, 0, true, chunk_longest);
			if (!rc && hipEventRecord(st.ev_seg[2 + (ci & 1)], chain) != hipSuccess)
				rc = fail(GEC_E_DEVICE, "hipEventRecord");
			i = j;
		}
		const hipError_t e1 = hipStreamSynchronize(up), e2 = hipStreamSynchronize(chain);
		if (rc)
			return rc;
		HIP_TRY(e1);
		HIP_TRY(e2);
		std::memcpy(out, h_out, n * 32);
		return GEC_OK;
	}
	if (all_pinned || longest >= (256u << 10) || tree) {
		DeviceGuard dg(c->device);
		if (!dg.ok)
			return fail(GEC_E_DEVICE, "hipSetDevice failed");
		constexpr size_t kPiece = 32ull << 20;
		std::vector<uint64_t> off(n);
		size_t dev_bytes = 0;
		for (size_t i = 0; i < n; ++i) {
			off[i] = dev_bytes;
			dev_bytes += (lens[i] + 15) / 16 * 16;
		}
		if (!all_pinned && dev_bytes > (8ull << 30)) {
			// more than a device buffer should hold at once: halves (each still one launch)
			const size_t h = n / 2;
			int rc = hash_batch_impl(c, h, msgs, lens, out, tree);
			return rc ? rc : hash_batch_impl(c, n - h, msgs + h, lens + h, out + 32 * h, tree);
		}
		StagingLease l0(c);
		const size_t meta = n * 16, res = n * 32;
		const size_t meta_off = all_pinned ? 0 : 2 * kPiece;  // h_buf of slot 0: [piece A][piece B][off][len][out]
		int rc = l0.st.ensure(meta_off + meta + res + 64, 0);
		if (rc)
			return rc;
		Staging &st = l0.st;
		uint8_t *d_msgs = nullptr;
		if (!all_pinned)
			HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d_msgs), std::max<size_t>(dev_bytes, 16)));
		uint64_t *h_off = reinterpret_cast<uint64_t *>(st.h_buf + meta_off);
		uint64_t *h_len = h_off + n;
		uint8_t *h_out = st.h_buf + meta_off + meta;
		for (size_t i = 0; i < n; ++i) {
			h_off[i] = all_pinned ? reinterpret_cast<uint64_t>(pinned().dev(msgs[i])) : off[i];
			h_len[i] = lens[i];
		}
		auto cleanup = [&](int code) {
			if (d_msgs) {
				(void)hipStreamSynchronize(st.stream);
				(void)hipFree(d_msgs);
			}
			return code;
		};
		if (!all_pinned) {
			// two staging pieces, filled by the copy pool while the other one is on the bus
			CopyPool &pool = c->copy_pool();
			hipEvent_t done[2] = {st.ev_fork, st.ev_join};
			bool used[2] = {false, false};
			size_t piece = 0;
			for (size_t i = 0; i < n;) {
				// messages (or parts of a long one) that fit the piece
				uint8_t *hp = st.h_buf + (piece & 1) * kPiece;
				if (used[piece & 1]) {
					hipError_t e = hipEventSynchronize(done[piece & 1]);
					if (e != hipSuccess)
						return cleanup(fail(GEC_E_DEVICE, std::string("hipEventSynchronize: ") + hipGetErrorString(e)));
				}
				const size_t d0 = off[i];
				size_t j = i, bytes = 0;
				while (j < n && bytes + (lens[j] + 15) / 16 * 16 <= kPiece) {
					bytes += (lens[j] + 15) / 16 * 16;
					++j;
				}
				if (j == i) {  // one message longer than a piece: stream it through in piece-sized parts
					for (size_t o = 0; o < lens[i]; o += kPiece) {
						hp = st.h_buf + (piece & 1) * kPiece;
						if (used[piece & 1] && hipEventSynchronize(done[piece & 1]) != hipSuccess)
							return cleanup(fail(GEC_E_DEVICE, "hipEventSynchronize failed"));
						const size_t nbytes = std::min(kPiece, lens[i] - o);
						const size_t parts = (nbytes + (1 << 20) - 1) >> 20;
						pool.parallel_for(parts, [&](size_t q) {
							const size_t a = q << 20, b = std::min(nbytes, a + (1 << 20));
							std::memcpy(hp + a, msgs[i] + o + a, b - a);
						});
						hipError_t e = hipMemcpyAsync(d_msgs + off[i] + o, hp, nbytes, hipMemcpyHostToDevice, st.stream);
						if (e == hipSuccess)
							e = hipEventRecord(done[piece & 1], st.stream);
						if (e != hipSuccess)
							return cleanup(fail(GEC_E_DEVICE, std::string("H2D: ") + hipGetErrorString(e)));
						used[piece & 1] = true;
						++piece;
					}
					++i;
					continue;
				}
				pool.parallel_for(j - i, [&](size_t q) { std::memcpy(hp + (off[i + q] - d0), msgs[i + q], lens[i + q]); });
				hipError_t e = hipMemcpyAsync(d_msgs + d0, hp, bytes, hipMemcpyHostToDevice, st.stream);
				if (e == hipSuccess)
					e = hipEventRecord(done[piece & 1], st.stream);
				if (e != hipSuccess)
					return cleanup(fail(GEC_E_DEVICE, std::string("H2D: ") + hipGetErrorString(e)));
				used[piece & 1] = true;
				++piece;
				i = j;
			}
		}
		// the (offset, length) table and the results live in pinned host memory the kernel reads / writes directly
		rc = blake2_dev(c, n, all_pinned ? nullptr : d_msgs, h_off, h_len, 0, 0, h_out, st.stream, 0, 0, 0, tree, longest);
		if (rc)
			return cleanup(rc);
		hipError_t e = hipStreamSynchronize(st.stream);
		if (e != hipSuccess)
			return cleanup(fail(GEC_E_DEVICE, std::string("hipStreamSynchronize: ") + hipGetErrorString(e)));
		std::memcpy(out, h_out, res);
		return cleanup(GEC_OK);
	}
	// (c) many short messages in pageable memory: greedy chunks of <= 4*kChunkBytes of (16-byte aligned) message
	//     slots through the two-slot pipeline (hashing of chunk i overlaps the upload of chunk i+1)
	struct Chunk {
		size_t first, count, bytes;
	};
	std::vector<Chunk> chunks;
	std::vector<uint64_t> slot_off(n);
	size_t max_bytes = 0, max_count = 0;
	for (size_t i = 0; i < n;) {
		Chunk ck{i, 0, 0};
		while (i < n && (ck.count == 0 || ck.bytes + lens[i] <= 4 * kChunkBytes)) {
			slot_off[i] = ck.bytes;
			ck.bytes += (lens[i] + 15) / 16 * 16;
			++ck.count;
			++i;
		}
		chunks.push_back(ck);
		max_bytes = std::max(max_bytes, ck.bytes);
		max_count = std::max(max_count, ck.count);
	}
	// slot layout: [messages][off u64 x count][len u64 x count][out 32 x count]
	const size_t meta_off = (max_bytes + 63) / 64 * 64;
	const size_t out_off = meta_off + 16 * max_count;
	CopyPool &pool = c->copy_pool();
	return run_pipeline(
		c, chunks.size(), out_off + 32 * max_count, 0,
		[&](size_t ci, Staging &st) {
			const Chunk &ck = chunks[ci];
			uint64_t *offs = reinterpret_cast<uint64_t *>(st.h_buf + meta_off);
			uint64_t *ls = offs + ck.count;
			pool.parallel_for(ck.count, [&](size_t i) {
				std::memcpy(st.h_buf + slot_off[ck.first + i], msgs[ck.first + i], lens[ck.first + i]);
				offs[i] = slot_off[ck.first + i];
				ls[i] = lens[ck.first + i];
			});
		},
		[&](size_t ci, Staging &st) -> int {
			const Chunk &ck = chunks[ci];
			HIP_TRY(hipMemcpyAsync(st.d_buf, st.h_buf, ck.bytes, hipMemcpyHostToDevice, st.stream));
			HIP_TRY(hipMemcpyAsync(st.d_buf + meta_off, st.h_buf + meta_off, 16 * ck.count, hipMemcpyHostToDevice, st.stream));
			const uint64_t *d_off = reinterpret_cast<const uint64_t *>(st.d_buf + meta_off);
			int rc = blake2_dev(c, ck.count, st.d_buf, d_off, d_off + ck.count, 0, 0, st.d_buf + out_off, st.stream);
			if (rc)
				return rc;
			HIP_TRY(hipMemcpyAsync(st.h_buf + out_off, st.d_buf + out_off, 32 * ck.count, hipMemcpyDeviceToHost, st.stream));
			return GEC_OK;
		},
		[&](size_t ci, Staging &st) {
			const Chunk &ck = chunks[ci];
			std::memcpy(out + 32 * ck.first, st.h_buf + out_off, 32 * ck.count);
		});
}

int gec_blake2sum_batch(const gec_codec *c, size_t n, const uint8_t *const *msgs, const size_t *lens, uint8_t *out)
{
	return hash_batch_impl(c, n, msgs, lens, out, false);
}

int gec_shardsum_batch(const gec_codec *c, size_t n, const uint8_t *const *msgs, const size_t *lens, uint8_t *out)
{
	return hash_batch_impl(c, n, msgs, lens, out, true);
}

// ---------------------------------------------------------------------------------------------------
// The read path in ONE trip (twin of gec_encode_hash_batch): upload the k shards each block is read from,
// checksum every uploaded shard, rebuild missing data shards, checksum the assembled block.
int gec_decode_verify_batch(const gec_codec *c, size_t nblocks, const uint8_t *const *shards, size_t S,
			    const size_t *block_len, uint8_t *const *rebuilt, uint8_t *shard_sums, uint8_t *block_sums)
{
	if (!c)
		return fail(GEC_E_INVALID_ARG, "NULL codec");
	if (nblocks == 0)
		return GEC_OK;
	if (!shards || !shard_sums || (block_sums && !block_len))
		return fail(GEC_E_INVALID_ARG, "NULL argument");
	if (S == 0)
		return fail(GEC_E_EMPTY_SHARD, "shard length is 0");
	if (S % 64)
		return fail(GEC_E_INCORRECT_SHARD_SIZE, "S must be a multiple of 64");
	const size_t k = c->k, n = c->k + c->m;
	// more than one device buffer should hold: halves
	const size_t kMaxBytes = 6ull << 30;
	if (nblocks > 1 && nblocks * n * S > kMaxBytes) {
		const size_t h = nblocks / 2;
		int rc = gec_decode_verify_batch(c, h, shards, S, block_len, rebuilt, shard_sums, block_sums);
		if (rc)
			return rc;
		return gec_decode_verify_batch(c, nblocks - h, shards + h * n, S, block_len ? block_len + h : nullptr,
					       rebuilt ? rebuilt + h * n : nullptr, shard_sums + h * n * 32,
					       block_sums ? block_sums + h * 32 : nullptr);
	}
	// -- per block: which shards are read (the crate's rule: the first k present), which data shards are rebuilt;
	//    blocks are laid out on the device bucket by bucket (one erasure pattern each), a block's stripe holding
	//    its k data slots followed by one slot per parity shard it is decoded from
	struct Bucket {
		std::shared_ptr<const Plan> plan;
		std::vector<size_t> ids;
		size_t base = 0, stripe = 0, npar = 0;
	};
	std::map<std::string, Bucket> buckets;
	for (size_t b = 0; b < nblocks; ++b) {
		std::string key(n, 0);
		size_t np = 0;
		for (size_t j = 0; j < n; ++j) {
			key[j] = shards[b * n + j] ? 1 : 0;
			np += key[j];
		}
		if (np < k)
			return fail(GEC_E_TOO_FEW_PRESENT, "fewer than k shards present");
		if (block_len && block_len[b] > k * S)
			return fail(GEC_E_INCORRECT_SHARD_SIZE, "block longer than k*S");
		buckets[key].ids.push_back(b);
	}
	size_t dev_bytes = 0, nup = 0, nreb = 0;
	for (auto &kv : buckets) {
		Bucket &bk = kv.second;
		int rc = get_plan(c, reinterpret_cast<const uint8_t *>(kv.first.data()), true, bk.plan);
		if (rc)
			return rc;
		bk.npar = bk.plan->missing.size();  // as many parity inputs as data shards to rebuild
		bk.stripe = (k + bk.npar) * S;
		bk.base = dev_bytes;
		dev_bytes += bk.ids.size() * bk.stripe;
		nup += bk.ids.size() * k;
		nreb += bk.ids.size() * bk.npar;
		for (size_t b : bk.ids)
			for (int j : bk.plan->missing)
				if (!rebuilt || !rebuilt[b * n + j])
					return fail(GEC_E_INVALID_ARG, "NULL output for a missing data shard");
	}
	DeviceGuard dg(c->device);
	if (!dg.ok)
		return fail(GEC_E_DEVICE, "hipSetDevice failed");
	bool all_pinned = true;
	for (size_t b = 0; b < nblocks && all_pinned; ++b)
		for (size_t j = 0; j < n && all_pinned; ++j)
			if (shards[b * n + j])
				all_pinned = aligned16(shards[b * n + j]) && pinned().contains(shards[b * n + j], S);
	StagingLease lease(c);
	Staging &st = lease.st;
	constexpr size_t kPiece = 32ull << 20;
	// host staging: [piece A][piece B] (pageable shards only) [shard off | shard len | block off | block len][shard sums][block sums][rebuilt]
	const size_t tab_off = all_pinned ? 0 : 2 * kPiece;
	const size_t tab_bytes = (nup * 2 + nblocks * 2) * 8;
	const size_t ssum_off = tab_off + tab_bytes, bsum_off = ssum_off + nup * 32;
	const size_t reb_off = (bsum_off + nblocks * 32 + 63) / 64 * 64;
	int rc = st.ensure(reb_off + nreb * S + 64, 0);
	if (rc)
		return rc;
	const size_t state_off = (dev_bytes + 63) / 64 * 64;  // chaining values of the segmented block checksums
	rc = st.ensure_big(state_off + nblocks * 64 + 64);
	if (rc)
		return rc;
	rc = st.ensure_tab(nup + nreb);
	if (rc)
		return rc;
	rc = st.ensure_segments(c->num_cu);
	if (rc)
		return rc;
	uint64_t *h_soff = reinterpret_cast<uint64_t *>(st.h_buf + tab_off), *h_slen = h_soff + nup;
	uint64_t *h_boff = h_slen + nup, *h_blen = h_boff + nblocks;
	// -- block table: the blocks that need no decode first -- their checksum chains start while the upload is
	//    still running (below); the others are hashed after their decode
	size_t bi = 0, nh = 0, longest = 0;
	std::vector<size_t> block_order(nblocks);
	for (int pass = 0; pass < 2; ++pass) {
		for (auto &kv : buckets) {
			if ((kv.second.npar == 0) != (pass == 0))
				continue;
			for (size_t i = 0; i < kv.second.ids.size(); ++i) {
				h_boff[bi] = kv.second.base + i * kv.second.stripe;
				h_blen[bi] = block_len ? block_len[kv.second.ids[i]] : 0;
				longest = std::max<size_t>(longest, h_blen[bi]);
				block_order[bi++] = kv.second.ids[i];
			}
		}
		if (pass == 0)
			nh = bi;
	}
	// Upload stages: stage s carries data slots [k*s/nseg, k*(s+1)/nseg) of the blocks that need no decode (stage 0
	// also everything of the blocks that do).  One stage unless every shard is pinned (the staged path uploads
	// dense device ranges block by block) and the chains are long enough to be worth hiding: a BLAKE2b chain runs
	// at ~14 ms per MiB, the link moves the MiB of 512 such blocks in 10 ms.
	static const int seg_max = [] {
		const char *e = std::getenv("GEC_VERIFY_SEGMENTS");  // A/B: 1 = upload everything, then hash
		const int v = e ? std::atoi(e) : Staging::kMaxSeg;
		return std::max(1, std::min(v, (int)Staging::kMaxSeg));
	}();
	const size_t nseg = (all_pinned && block_sums && nh > 0 && longest >= (256u << 10)) ? std::min<size_t>(k, (size_t)seg_max) : 1;
	// -- upload list, in device order
	struct Up {
		const uint8_t *src;
		size_t dst;  // byte offset in d_big
		size_t idx;  // (b*n + j): where the shard's checksum goes
		size_t stage;
	};
	std::vector<Up> ups;
	ups.reserve(nup);
	for (auto &kv : buckets) {
		Bucket &bk = kv.second;
		for (size_t i = 0; i < bk.ids.size(); ++i) {
			const size_t b = bk.ids[i];
			size_t q = 0;  // parity input slot
			for (size_t t = 0; t < k; ++t) {
				const int j = bk.plan->valid[t];
				const size_t slot = (size_t)j < k ? (size_t)j : k + q++;
				const uint8_t *p = shards[b * n + j];
				// stage of data slot j: the s with k*s/nseg <= j < k*(s+1)/nseg
				size_t stage = 0;
				if (bk.npar == 0 && nseg > 1)
					while (k * (stage + 1) / nseg <= slot)
						++stage;
				ups.push_back({p, bk.base + i * bk.stripe + slot * S, b * n + j, stage});
			}
		}
	}
	std::sort(ups.begin(), ups.end(), [](const Up &a, const Up &b) { return a.dst < b.dst; });
	for (size_t i = 0; i < ups.size(); ++i) {
		h_soff[i] = ups[i].dst;
		h_slen[i] = S;
	}
	auto hip_fail = [&](hipError_t e, const char *what) { return fail(GEC_E_DEVICE, std::string(what) + ": " + hipGetErrorString(e)); };
	hipEvent_t ev_up = nullptr, ev_sh = nullptr;
	HIP_TRY(hipEventCreateWithFlags(&ev_up, hipEventDisableTiming));
	HIP_TRY(hipEventCreateWithFlags(&ev_sh, hipEventDisableTiming));
	// staged upload: its own (CU-masked) stream pair when there is one
	hipStream_t up_stream = nseg > 1 && st.stream_up ? st.stream_up : st.stream;
	hipStream_t chain_stream = nseg > 1 && st.stream_chain ? st.stream_chain : st.stream2;
	auto finish = [&](int code) {
		(void)hipStreamSynchronize(up_stream);
		(void)hipStreamSynchronize(chain_stream);
		(void)hipStreamSynchronize(st.stream);
		(void)hipStreamSynchronize(st.stream2);
		(void)hipStreamSynchronize(st.stream3);
		(void)hipEventDestroy(ev_up);
		(void)hipEventDestroy(ev_sh);
		return code;
	};
	bool healthy_hashed = false;
	if (all_pinned) {
		std::vector<std::vector<gec::CopyEntry>> ents(nseg);
		for (size_t i = 0; i < ups.size();) {  // merge neighbours (the data shards of a block are slices of one buffer)
			size_t run = 1;
			while (i + run < ups.size() && ups[i + run].stage == ups[i].stage && ups[i + run].src == ups[i].src + run * S &&
			       ups[i + run].dst == ups[i].dst + run * S)
				++run;
			ents[ups[i].stage].push_back({pinned().dev(ups[i].src), st.d_big + ups[i].dst, run * S});
			i += run;
		}
		for (size_t sg = 0; sg < nseg; ++sg) {
			rc = launch_copy_table(st, ents[sg], up_stream);
			if (rc)
				return finish(rc);
			if (nseg == 1)
				break;
			// the chains of the no-decode blocks advance over what has arrived: whole 128-byte blocks below the
			// end of this stage's last slot (the final stage finishes every message)
			hipError_t es = hipEventRecord(st.ev_seg[sg], up_stream);
			if (es == hipSuccess)
				es = hipStreamWaitEvent(chain_stream, st.ev_seg[sg], 0);
			if (es != hipSuccess)
				return finish(hip_fail(es, "stage event"));
			const uint64_t blk0 = (k * sg / nseg) * S / 128;
			const uint64_t blk1 = sg + 1 == nseg ? ~0ull : (k * (sg + 1) / nseg) * S / 128;
			rc = blake2_dev(c, nh, st.d_big, h_boff, h_blen, 0, 0, st.h_buf + bsum_off, chain_stream, 0, 0, 0, false, 0,
					reinterpret_cast<uint64_t *>(st.d_big + state_off), blk0, blk1);
			if (rc)
				return finish(rc);
		}
		healthy_hashed = nseg > 1;
	} else {
		// pageable shards: the pieces are images of dense device ranges, filled by the copy pool while the other is on the bus
		CopyPool &pool = c->copy_pool();
		hipEvent_t done[2] = {st.ev_fork, st.ev_join};
		bool used[2] = {false, false};
		size_t piece = 0;
		for (size_t i = 0; i < ups.size();) {
			uint8_t *hp = st.h_buf + (piece & 1) * kPiece;
			if (used[piece & 1]) {
				hipError_t e = hipEventSynchronize(done[piece & 1]);
				if (e != hipSuccess)
					return finish(hip_fail(e, "hipEventSynchronize"));
			}
			const size_t d0 = ups[i].dst;
			size_t j = i;
			while (j < ups.size() && ups[j].dst + S - d0 <= kPiece)
				++j;
			if (j == i)
				return finish(fail(GEC_E_INVALID_ARG, "shard larger than a staging piece"));
			const size_t bytes = ups[j - 1].dst + S - d0;
			pool.parallel_for(j - i, [&](size_t q) { std::memcpy(hp + (ups[i + q].dst - d0), ups[i + q].src, S); });
			hipError_t e = hipMemcpyAsync(st.d_big + d0, hp, bytes, hipMemcpyHostToDevice, st.stream);
			if (e == hipSuccess)
				e = hipEventRecord(done[piece & 1], st.stream);
			if (e != hipSuccess)
				return finish(hip_fail(e, "H2D"));
			used[piece & 1] = true;
			++piece;
			i = j;
		}
	}
	// -- everything is on the device.  Shard checksums on their own stream (they depend on nothing else); on the
	//    main stream: decode per bucket, then the checksums of the blocks not hashed yet, then the rebuilt shards go home.
	hipError_t e = hipEventRecord(ev_up, up_stream);
	if (e == hipSuccess)
		e = hipStreamWaitEvent(st.stream3, ev_up, 0);
	if (e == hipSuccess && up_stream != st.stream)
		e = hipStreamWaitEvent(st.stream, ev_up, 0);
	if (e != hipSuccess)
		return finish(hip_fail(e, "fork"));
	rc = blake2_dev(c, nup, st.d_big, h_soff, h_slen, 0, 0, st.h_buf + ssum_off, st.stream3, 0, 0, 0, true, S);
	if (rc)
		return finish(rc);
	e = hipEventRecord(ev_sh, st.stream3);
	if (e != hipSuccess)
		return finish(hip_fail(e, "hipEventRecord"));
	std::vector<gec::CopyEntry> outs;
	size_t rq = 0;
	for (auto &kv : buckets) {
		Bucket &bk = kv.second;
		if (bk.npar == 0)
			continue;
		std::vector<size_t> in_off(k), out_off(bk.npar);
		size_t q = 0;
		for (size_t t = 0; t < k; ++t) {
			const int j = bk.plan->valid[t];
			in_off[t] = ((size_t)j < k ? (size_t)j : k + q++) * S;
		}
		for (size_t r = 0; r < bk.npar; ++r)
			out_off[r] = (size_t)bk.plan->missing[r] * S;  // rebuilt in place, in the block's data area
		rc = launch_apply(c, st.d_big + bk.base, bk.stripe, st.d_big + bk.base, bk.stripe, nullptr, 0, S, bk.ids.size(),
				  in_off.data(), out_off.data(), (int)bk.npar, bk.plan->rows.v.data(), gec::MODE_STORE, st.stream);
		if (rc)
			return finish(rc);
		for (size_t i = 0; i < bk.ids.size(); ++i)
			for (size_t r = 0; r < bk.npar; ++r) {
				uint8_t *dst = rebuilt[bk.ids[i] * n + bk.plan->missing[r]];
				const bool direct = aligned16(dst) && pinned().contains(dst, S);
				outs.push_back({st.d_big + bk.base + i * bk.stripe + out_off[r], direct ? pinned().dev(dst) : st.h_buf + reb_off + rq * S, S});
				++rq;
			}
	}
	if (block_sums) {
		const size_t first = healthy_hashed ? nh : 0;  // [0, nh) went through the segments above
		rc = blake2_dev(c, nblocks - first, st.d_big, h_boff + first, h_blen + first, 0, 0, st.h_buf + bsum_off + 32 * first, st.stream);
		if (rc)
			return finish(rc);
	}
	rc = launch_copy_table(st, outs, st.stream);
	if (rc)
		return finish(rc);
	if (healthy_hashed) {
		e = hipEventRecord(st.ev_join, chain_stream);
		if (e == hipSuccess)
			e = hipStreamWaitEvent(st.stream, st.ev_join, 0);
		if (e != hipSuccess)
			return finish(hip_fail(e, "join"));
	}
	e = hipStreamWaitEvent(st.stream, ev_sh, 0);
	if (e == hipSuccess)
		e = hipStreamSynchronize(st.stream);
	if (e != hipSuccess)
		return finish(hip_fail(e, "hipStreamSynchronize"));
	// -- results: checksums to where the caller indexes them, rebuilt shards that could not be written directly
	for (size_t i = 0; i < ups.size(); ++i)
		std::memcpy(shard_sums + 32 * ups[i].idx, st.h_buf + ssum_off + 32 * i, 32);
	if (block_sums)
		for (size_t i = 0; i < nblocks; ++i)
			std::memcpy(block_sums + 32 * block_order[i], st.h_buf + bsum_off + 32 * i, 32);
	rq = 0;
	for (auto &kv : buckets) {
		Bucket &bk = kv.second;
		for (size_t i = 0; i < bk.ids.size(); ++i)
			for (size_t r = 0; r < bk.npar; ++r, ++rq) {
				uint8_t *dst = rebuilt[bk.ids[i] * n + bk.plan->missing[r]];
				if (!(aligned16(dst) && pinned().contains(dst, S)))
					std::memcpy(dst, st.h_buf + reb_off + rq * S, S);
			}
	}
	return finish(GEC_OK);
}

int gec_verify_batch(const gec_codec *c, size_t nblocks, const uint8_t *const *shards, size_t S, uint8_t *ok)
{
	if (!c)
		return fail(GEC_E_INVALID_ARG, "NULL codec");
	if (nblocks == 0)
		return GEC_OK;
	if (!shards || !ok)
		return fail(GEC_E_INVALID_ARG, "NULL argument");
	if (S == 0)
		return fail(GEC_E_EMPTY_SHARD, "shard length is 0");
	if (S % 64)
		return fail(GEC_E_INCORRECT_SHARD_SIZE, "S must be a multiple of 64");
	const size_t n = c->k + c->m;
	for (size_t i = 0; i < nblocks * n; ++i)
		if (!shards[i])
			return fail(GEC_E_TOO_FEW_SHARDS, "verify needs all k+m shards");
	const size_t stripe = n * S;
	bool all_pinned = (size_t)c->k <= (size_t)gec::PTR_KMAX && zero_copy_enabled();
	for (size_t i = 0; i < nblocks * n && all_pinned; ++i)
		all_pinned = aligned16(shards[i]) && pinned().contains(shards[i], S);
	if (all_pinned) {
		// scrub of shards that sit in pinned memory: one kernel reads all k+m shards over the link and leaves the
		// per-block verdicts in pinned memory; nothing is staged
		const size_t k = c->k, m = c->m;
		DeviceGuard dg(c->device);
		if (!dg.ok)
			return fail(GEC_E_DEVICE, "hipSetDevice failed");
		StagingLease lease(c);
		Staging &st = lease.st;
		int rc = st.ensure(64, nblocks);
		if (!rc)
			rc = st.ensure_tab((nblocks * (k * 12 + m * 8) + 64) / sizeof(gec::CopyEntry) + 8);
		if (rc)
			return rc;
		std::vector<const uint8_t *> in(nblocks * k);
		std::vector<uint32_t> valid(nblocks * k, (uint32_t)S);
		std::vector<uint8_t *> par(nblocks * m);
		for (size_t b = 0; b < nblocks; ++b) {
			for (size_t t = 0; t < k; ++t)
				in[b * k + t] = pinned().dev(shards[b * n + t]);
			for (size_t r = 0; r < m; ++r)
				par[b * m + r] = const_cast<uint8_t *>(pinned().dev(shards[b * n + k + r]));
		}
		std::memset(st.h_bad, 0, nblocks * sizeof(uint32_t));
		rc = launch_apply_ptrs(c, st, nblocks, in.data(), valid.data(), par.data(), (int)m, S, c->enc.row((int)k), st.stream, nullptr, st.h_bad);
		const hipError_t e = hipStreamSynchronize(st.stream);
		if (rc)
			return rc;
		HIP_TRY(e);
		for (size_t b = 0; b < nblocks; ++b)
			ok[b] = st.h_bad[b] ? 0 : 1;
		return GEC_OK;
	}
	const size_t ch = chunk_blocks(stripe, nblocks, kChunkBytes);
	CopyPool &pool = c->copy_pool();
	return run_pipeline(
		c, (nblocks + ch - 1) / ch, ch * stripe, ch,
		[&](size_t ci, Staging &st) {
			const size_t b0 = ci * ch, nb = std::min(ch, nblocks - b0);
			pool.parallel_for(nb * n, [&](size_t q) {
				std::memcpy(st.h_buf + (q / n) * stripe + (q % n) * S, shards[(b0 + q / n) * n + q % n], S);
			});
		},
		[&](size_t ci, Staging &st) -> int {
			const size_t nb = std::min(ch, nblocks - ci * ch);
			HIP_TRY(hipMemcpyAsync(st.d_buf, st.h_buf, nb * stripe, hipMemcpyHostToDevice, st.stream));
			int rc = verify_dev(c, nb, st.d_buf, stripe, S, st.d_bad, st.stream);
			if (rc)
				return rc;
			HIP_TRY(hipMemcpyAsync(st.h_bad, st.d_bad, nb * sizeof(uint32_t), hipMemcpyDeviceToHost, st.stream));
			return GEC_OK;
		},
		[&](size_t ci, Staging &st) {
			const size_t b0 = ci * ch, nb = std::min(ch, nblocks - b0);
			for (size_t i = 0; i < nb; ++i)
				ok[b0 + i] = st.h_bad[i] ? 0 : 1;
		});
}

int gec_verify_hash_batch(const gec_codec *c, size_t nblocks, const uint8_t *const *shards, size_t S, uint8_t *ok,
			  uint8_t *shard_sums)
{
	if (!c)
		return fail(GEC_E_INVALID_ARG, "NULL codec");
	if (nblocks == 0)
		return GEC_OK;
	if (!shards || !ok || !shard_sums)
		return fail(GEC_E_INVALID_ARG, "NULL argument");
	if (S == 0)
		return fail(GEC_E_EMPTY_SHARD, "shard length is 0");
	if (S % 64)
		return fail(GEC_E_INCORRECT_SHARD_SIZE, "S must be a multiple of 64");
	const size_t k = c->k, m = c->m, n = k + m;
	for (size_t i = 0; i < nblocks * n; ++i)
		if (!shards[i])
			return fail(GEC_E_TOO_FEW_SHARDS, "verify needs all k+m shards");
	bool all_pinned = k <= (size_t)gec::PTR_KMAX && zero_copy_enabled();
	for (size_t i = 0; i < nblocks * n && all_pinned; ++i)
		all_pinned = aligned16(shards[i]) && pinned().contains(shards[i], S);
	if (!all_pinned) {
		// pageable shards: two staged trips (the scrub of shards a caller keeps in ordinary memory is not a hot path)
		int rc = gec_verify_batch(c, nblocks, shards, S, ok);
		if (rc)
			return rc;
		std::vector<size_t> lens(nblocks * n, S);
		return gec_shardsum_batch(c, nblocks * n, shards, lens.data(), shard_sums);
	}
	// one trip: the compare form of gf_apply_ptrs reads all k+m shards of a chunk over the link, leaves the verdicts in
	// pinned memory and everything it read in HBM, where the chunk's shard checksums are computed while the next
	// chunk is on the link (same stream pair as gec_encode_hash_batch)
	DeviceGuard dg(c->device);
	if (!dg.ok)
		return fail(GEC_E_DEVICE, "hipSetDevice failed");
	StagingLease lease(c);
	Staging &st = lease.st;
	const size_t stripe = n * S;
	const size_t zch = chunk_blocks(stripe, nblocks, 8 * kChunkBytes);
	const size_t nz = (nblocks + zch - 1) / zch;
	int rc = st.ensure(nblocks * n * 32 + 64, nblocks);
	if (!rc)
		rc = st.ensure_tab((nblocks * (k * 12 + m * 8) + 64 * nz) / sizeof(gec::CopyEntry) + 4 * nz + 4);
	if (!rc)
		rc = st.ensure_big(2 * (zch * stripe + zch * n * 32));
	if (!rc)
		rc = st.ensure_segments(c->num_cu);
	if (rc)
		return rc;
	hipStream_t up = st.stream_up ? st.stream_up : st.stream;
	hipStream_t chain = st.stream_chain ? st.stream_chain : st.stream2;
	std::memset(st.h_bad, 0, nblocks * sizeof(uint32_t));
	std::vector<const uint8_t *> in(zch * k);
	std::vector<uint32_t> valid(zch * k, (uint32_t)S);
	std::vector<uint8_t *> par(zch * m);
	for (size_t ci = 0; ci < nz && !rc; ++ci) {
		const size_t b0 = ci * zch, nb = std::min(zch, nblocks - b0);
		for (size_t i = 0; i < nb; ++i) {
			for (size_t t = 0; t < k; ++t)
				in[i * k + t] = pinned().dev(shards[(b0 + i) * n + t]);
			for (size_t r = 0; r < m; ++r)
				par[i * m + r] = const_cast<uint8_t *>(pinned().dev(shards[(b0 + i) * n + k + r]));
		}
		uint8_t *mir = st.d_big + (ci & 1) * (zch * stripe + zch * n * 32);
		if (ci >= 2 && hipStreamWaitEvent(up, st.ev_seg[2 + (ci & 1)], 0) != hipSuccess)
			rc = fail(GEC_E_DEVICE, "hipStreamWaitEvent");
		if (!rc)
			rc = launch_apply_ptrs(c, st, nb, in.data(), valid.data(), par.data(), (int)m, S, c->enc.row((int)k), up, mir, st.h_bad + b0);
		if (rc)
			continue;
		uint8_t *d_sums = mir + zch * stripe;
		hipError_t e = hipEventRecord(st.ev_seg[ci & 1], up);
		if (e == hipSuccess)
			e = hipStreamWaitEvent(chain, st.ev_seg[ci & 1], 0);
		if (e != hipSuccess) {
			rc = fail(GEC_E_DEVICE, "chunk event");
			continue;
		}
		rc = blake2_dev(c, nb * n, mir, nullptr, nullptr, S, S, d_sums, chain, 0, 0, 0, true);
		if (!rc && hipMemcpyAsync(st.h_buf + b0 * n * 32, d_sums, nb * n * 32, hipMemcpyDeviceToHost, chain) != hipSuccess)
			rc = fail(GEC_E_DEVICE, "hipMemcpyAsync (shard sums)");
		if (!rc && hipEventRecord(st.ev_seg[2 + (ci & 1)], chain) != hipSuccess)
			rc = fail(GEC_E_DEVICE, "hipEventRecord");
	}
	const hipError_t e1 = hipStreamSynchronize(up), e2 = hipStreamSynchronize(chain);
	if (rc)
		return rc;
	HIP_TRY(e1);
	HIP_TRY(e2);
	std::memcpy(shard_sums, st.h_buf, nblocks * n * 32);
	for (size_t b = 0; b < nblocks; ++b)
		ok[b] = st.h_bad[b] ? 0 : 1;
	return GEC_OK;
}

// in_sums / out_sums (both or neither): the shard checksums of the k shards READ per block and of the shards WRITTEN,
// at 32*(b*n + j), from the same trip (gec_reconstruct_hash_batch)
static int reconstruct_batch_impl(const gec_codec *c, size_t nblocks, const uint8_t *const *shards, uint8_t *const *out,
				  size_t S, int data_only, uint8_t *in_sums, uint8_t *out_sums)
{
	if (!c)
		return fail(GEC_E_INVALID_ARG, "NULL codec");
	if (nblocks == 0)
		return GEC_OK;
	if (!shards || !out)
		return fail(GEC_E_INVALID_ARG, "NULL argument");
	if (S == 0)
		return fail(GEC_E_EMPTY_SHARD, "shard length is 0");
	if (S % 64)
		return fail(GEC_E_INCORRECT_SHARD_SIZE, "S must be a multiple of 64");
	const size_t k = c->k, n = c->k + c->m;
	// bucket blocks by erasure pattern AND by which of the missing shards the caller wants back
	// (out entry non-NULL; with data_only parity is never wanted): one decode plan per bucket, one
	// launch per chunk, only the wanted rows computed.  key[j]: 1 present, 0 missing+wanted, 2 missing+unwanted
	std::map<std::string, std::vector<size_t>> buckets;
	for (size_t b = 0; b < nblocks; ++b) {
		std::string key(n, 0);
		size_t npresent = 0, nwanted = 0;
		for (size_t j = 0; j < n; ++j) {
			if (shards[b * n + j]) {
				key[j] = 1;
				++npresent;
			} else if ((data_only && j >= k) || !out[b * n + j]) {
				key[j] = 2;
			} else {
				++nwanted;
			}
		}
		if (npresent < k)
			return fail(GEC_E_TOO_FEW_PRESENT, "fewer than k shards present");
		if (nwanted)
			buckets[key].push_back(b);
	}
	CopyPool &pool = c->copy_pool();
	// one decode plan per bucket, cut down to the rows the caller wants
	struct Work {
		const std::vector<size_t> *ids;
		std::shared_ptr<const Plan> plan;
		bool all_pinned;
	};
	std::vector<Work> work;
	bool every_pinned = true;
	size_t tab_bytes = 0;
	for (auto &kv : buckets) {
		const std::vector<size_t> &ids = kv.second;
		std::string pres(kv.first);
		for (auto &ch : pres)
			ch = ch == 1 ? 1 : 0;
		std::shared_ptr<const Plan> full;
		int rc = get_plan(c, reinterpret_cast<const uint8_t *>(pres.data()), false, full);
		if (rc)
			return rc;
		auto sub = std::make_shared<Plan>();
		sub->valid = full->valid;
		for (size_t r = 0; r < full->missing.size(); ++r)
			if (kv.first[full->missing[r]] == 0)
				sub->missing.push_back(full->missing[r]);
		sub->rows = gec::Matrix((int)sub->missing.size(), (int)k);
		for (size_t r = 0, w = 0; r < full->missing.size(); ++r)
			if (kv.first[full->missing[r]] == 0)
				std::memcpy(&sub->rows.at((int)w++, 0), full->rows.row((int)r), k);
		const size_t nmiss = sub->missing.size();
		if (nmiss == 0)
			continue;
		bool all_pinned = true;
		for (size_t i = 0; i < ids.size() && all_pinned; ++i) {
			for (size_t t = 0; t < k && all_pinned; ++t)
				all_pinned = aligned16(shards[ids[i] * n + sub->valid[t]]) && pinned().contains(shards[ids[i] * n + sub->valid[t]], S);
			for (size_t r = 0; r < nmiss && all_pinned; ++r)
				all_pinned = aligned16(out[ids[i] * n + sub->missing[r]]) && pinned().contains(out[ids[i] * n + sub->missing[r]], S);
		}
		every_pinned = every_pinned && all_pinned;
		tab_bytes += ids.size() * (k * 12 + nmiss * 8) + 64;
		work.push_back({&ids, sub, all_pinned});
	}
	if (!work.empty() && every_pinned && k <= (size_t)gec::PTR_KMAX && zero_copy_enabled()) {
		// every shard and every output is device-addressable: one gf_apply_ptrs launch per erasure pattern reads
		// the k shards the decode uses and writes the rebuilt ones straight over the link
		DeviceGuard dg(c->device);
		if (!dg.ok)
			return fail(GEC_E_DEVICE, "hipSetDevice failed");
		StagingLease lease(c);
		Staging &st = lease.st;
		const bool sums = in_sums != nullptr;
		// with checksums: chunks of every pattern run through the two mirror halves in turn (chunk q+2 waits for
		// the checksums of chunk q), link kernels on the upload CUs, checksum kernels on the rest
		size_t sum_bytes = 0, max_ids = 0, nchunks_total = 0;
		const size_t zch_cap = chunk_blocks(n * S, nblocks, 8 * kChunkBytes);
		for (const Work &w : work) {
			sum_bytes += w.ids->size() * (k + w.plan->missing.size()) * 32;
			max_ids = std::max(max_ids, w.ids->size());
			nchunks_total += (w.ids->size() + zch_cap - 1) / zch_cap;
		}
		const size_t zch = std::min(zch_cap, std::max<size_t>(max_ids, 1));
		const size_t half = zch * n * S + zch * n * 32;
		int rc = st.ensure(sums ? sum_bytes + 64 : 64, 0);
		if (!rc)
			rc = st.ensure_tab(tab_bytes / sizeof(gec::CopyEntry) + 4 * (work.size() + nchunks_total) + 4);
		if (!rc && sums)
			rc = st.ensure_big(2 * half);
		if (!rc && sums)
			rc = st.ensure_segments(c->num_cu);
		if (rc)
			return rc;
		hipStream_t up = sums && st.stream_up ? st.stream_up : st.stream;
		hipStream_t chain = sums && st.stream_chain ? st.stream_chain : st.stream2;
		size_t q = 0, sum_off = 0;  // running chunk number, running offset into the pinned checksum area
		std::vector<size_t> sum_base(work.size());
		for (size_t wi = 0; wi < work.size() && !rc; ++wi) {
			const Work &w = work[wi];
			const std::vector<size_t> &ids = *w.ids;
			const size_t nmiss = w.plan->missing.size(), per = k + nmiss;
			sum_base[wi] = sum_off;
			std::vector<const uint8_t *> in(std::min(zch, ids.size()) * k);
			std::vector<uint32_t> valid(in.size(), (uint32_t)S);
			std::vector<uint8_t *> outp(std::min(zch, ids.size()) * nmiss);
			for (size_t i0 = 0; i0 < ids.size() && !rc; i0 += sums ? zch : ids.size(), ++q) {
				const size_t nb = sums ? std::min(zch, ids.size() - i0) : ids.size();
				if (!sums) {
					in.resize(nb * k);
					valid.assign(nb * k, (uint32_t)S);
					outp.resize(nb * nmiss);
				}
				for (size_t i = 0; i < nb; ++i) {
					for (size_t t = 0; t < k; ++t)
						in[i * k + t] = pinned().dev(shards[ids[i0 + i] * n + w.plan->valid[t]]);
					for (size_t r = 0; r < nmiss; ++r)
						outp[i * nmiss + r] = pinned().dev(out[ids[i0 + i] * n + w.plan->missing[r]]);
				}
				uint8_t *mir = sums ? st.d_big + (q & 1) * half : nullptr;
				if (sums && q >= 2 && hipStreamWaitEvent(up, st.ev_seg[2 + (q & 1)], 0) != hipSuccess)
					rc = fail(GEC_E_DEVICE, "hipStreamWaitEvent");
				if (!rc)
					rc = launch_apply_ptrs(c, st, nb, in.data(), valid.data(), outp.data(), (int)nmiss, S, w.plan->rows.v.data(), up, mir);
				if (rc || !sums)
					continue;
				uint8_t *d_sums = mir + zch * n * S;
				hipError_t e = hipEventRecord(st.ev_seg[q & 1], up);
				if (e == hipSuccess)
					e = hipStreamWaitEvent(chain, st.ev_seg[q & 1], 0);
				if (e != hipSuccess) {
					rc = fail(GEC_E_DEVICE, "chunk event");
					continue;
				}
				rc = blake2_dev(c, nb * per, mir, nullptr, nullptr, S, S, d_sums, chain, 0, 0, 0, true);
				if (!rc && hipMemcpyAsync(st.h_buf + sum_off, d_sums, nb * per * 32, hipMemcpyDeviceToHost, chain) != hipSuccess)
					rc = fail(GEC_E_DEVICE, "hipMemcpyAsync (shard sums)");
				if (!rc && hipEventRecord(st.ev_seg[2 + (q & 1)], chain) != hipSuccess)
					rc = fail(GEC_E_DEVICE, "hipEventRecord");
				sum_off += nb * per * 32;
			}
		}
		const hipError_t e1 = hipStreamSynchronize(up), e2 = sums ? hipStreamSynchronize(chain) : hipSuccess;  // also on error: queued launches read the tables
		if (rc)
			return rc;
		HIP_TRY(e1);
		HIP_TRY(e2);
		if (sums)
			for (size_t wi = 0; wi < work.size(); ++wi) {
				const Work &w = work[wi];
				const size_t nmiss = w.plan->missing.size(), per = k + nmiss;
				for (size_t i = 0; i < w.ids->size(); ++i) {
					const uint8_t *src = st.h_buf + sum_base[wi] + i * per * 32;
					const size_t b = (*w.ids)[i];
					for (size_t t = 0; t < k; ++t)
						std::memcpy(in_sums + 32 * (b * n + w.plan->valid[t]), src + 32 * t, 32);
					for (size_t r = 0; r < nmiss; ++r)
						std::memcpy(out_sums + 32 * (b * n + w.plan->missing[r]), src + 32 * (k + r), 32);
				}
			}
		return GEC_OK;
	}
	if (in_sums) {
		// buffers the device cannot address: the staged reconstruct, then the checksums of what was read and written in
		// a second trip
		int rc = reconstruct_batch_impl(c, nblocks, shards, out, S, data_only, nullptr, nullptr);
		if (rc)
			return rc;
		std::vector<const uint8_t *> msgs;
		std::vector<size_t> lens, where;
		std::vector<uint8_t *> dst;
		for (const Work &w : work)
			for (size_t b : *w.ids) {
				for (size_t t = 0; t < k; ++t) {
					msgs.push_back(shards[b * n + w.plan->valid[t]]);
					dst.push_back(in_sums + 32 * (b * n + w.plan->valid[t]));
				}
				for (int j : w.plan->missing) {
					msgs.push_back(out[b * n + j]);
					dst.push_back(out_sums + 32 * (b * n + j));
				}
			}
		lens.assign(msgs.size(), S);
		std::vector<uint8_t> tmp(msgs.size() * 32);
		rc = gec_shardsum_batch(c, msgs.size(), msgs.data(), lens.data(), tmp.data());
		for (size_t i = 0; !rc && i < msgs.size(); ++i)
			std::memcpy(dst[i], tmp.data() + 32 * i, 32);
		return rc;
	}
	for (const Work &wk : work) {
		const std::vector<size_t> &ids = *wk.ids;
		std::shared_ptr<const Plan> plan = wk.plan;
		const size_t nmiss = plan->missing.size();
		const size_t stripe = (k + nmiss) * S;
		const bool all_pinned = wk.all_pinned;
		int rc = GEC_OK;
		const size_t ch = chunk_blocks(stripe, ids.size(), all_pinned ? pinned_chunk_bytes() : kChunkBytes);
		std::vector<size_t> in_off(k), out_off(nmiss);
		for (size_t t = 0; t < k; ++t)
			in_off[t] = t * S;
		for (size_t r = 0; r < nmiss; ++r)
			out_off[r] = (k + r) * S;
		// chunks whose input shards (resp. output buffers) all lie in registered pinned memory go by
		// DMA straight from / to the caller's memory; adjacent shards (slices of one block buffer)
		// are merged into one copy
		const size_t nchunks = (ids.size() + ch - 1) / ch;
		std::vector<uint8_t> in_pinned(nchunks, 1), out_pinned(nchunks, 1);
		PipeChain chain;
		for (size_t i = 0; i < ids.size(); ++i) {
			const size_t ci = i / ch;
			for (size_t t = 0; t < k && in_pinned[ci]; ++t)
				if (!(aligned16(shards[ids[i] * n + plan->valid[t]]) && pinned().contains(shards[ids[i] * n + plan->valid[t]], S)))
					in_pinned[ci] = 0;
			for (size_t r = 0; r < nmiss && out_pinned[ci]; ++r)
				if (!(aligned16(out[ids[i] * n + plan->missing[r]]) && pinned().contains(out[ids[i] * n + plan->missing[r]], S)))
					out_pinned[ci] = 0;
		}
		rc = run_pipeline(
			c, nchunks, ch * stripe, 0,
			[&](size_t ci, Staging &st) {
				(void)st.ensure_tab(ch * (k + nmiss));
				if (in_pinned[ci])
					return;
				const size_t i0 = ci * ch, nb = std::min(ch, ids.size() - i0);
				pool.parallel_for(nb * k, [&](size_t q) {
					const size_t i = q / k, t = q % k;
					std::memcpy(st.h_buf + i * stripe + t * S, shards[ids[i0 + i] * n + plan->valid[t]], S);
				});
			},
			[&](size_t ci, Staging &st) -> int {
				const size_t i0 = ci * ch, nb = std::min(ch, ids.size() - i0);
				if (in_pinned[ci]) {
					std::vector<gec::CopyEntry> ents;
					ents.reserve(nb * 3);
					for (size_t i = 0; i < nb; ++i) {
						const uint8_t *const *sh = shards + ids[i0 + i] * n;
						for (size_t t = 0; t < k;) {  // adjacent shards (slices of one block buffer): one entry
							size_t run = 1;
							while (t + run < k && sh[plan->valid[t + run]] == sh[plan->valid[t]] + run * S)
								++run;
							ents.push_back({pinned().dev(sh[plan->valid[t]]), st.d_buf + i * stripe + t * S, run * S});
							t += run;
						}
					}
					int rct = chain.before(chain.last_in, st.stream);
					if (!rct)
						rct = launch_copy_table(st, ents, st.stream);
					if (!rct)
						rct = chain.after_in(st);
					if (rct)
						return rct;
				} else {
					HIP_TRY(hipMemcpy2DAsync(st.d_buf, stripe, st.h_buf, stripe, k * S, nb, hipMemcpyHostToDevice, st.stream));
				}
				int r2 = launch_apply(c, st.d_buf, stripe, st.d_buf, stripe, nullptr, 0, S, nb, in_off.data(),
						      out_off.data(), (int)nmiss, plan->rows.v.data(), gec::MODE_STORE, st.stream);
				if (r2)
					return r2;
				if (out_pinned[ci]) {
					std::vector<gec::CopyEntry> ents;
					ents.reserve(nb * nmiss);
					for (size_t i = 0; i < nb; ++i) {
						uint8_t *const *o = out + ids[i0 + i] * n;
						for (size_t r = 0; r < nmiss; ++r)
							ents.push_back({st.d_buf + i * stripe + (k + r) * S, pinned().dev(o[plan->missing[r]]), S});
					}
					int rct = chain.before(chain.last_out, st.stream);
					if (!rct)
						rct = launch_copy_table(st, ents, st.stream);
					if (!rct)
						rct = chain.after_out(st);
					if (rct)
						return rct;
				} else {
					HIP_TRY(hipMemcpy2DAsync(st.h_buf + k * S, stripe, st.d_buf + k * S, stripe, nmiss * S, nb, hipMemcpyDeviceToHost, st.stream));
				}
				return GEC_OK;
			},
			[&](size_t ci, Staging &st) {
				if (out_pinned[ci])
					return;
				const size_t i0 = ci * ch, nb = std::min(ch, ids.size() - i0);
				pool.parallel_for(nb * nmiss, [&](size_t q) {
					const size_t i = q / nmiss, r = q % nmiss;
					std::memcpy(out[ids[i0 + i] * n + plan->missing[r]], st.h_buf + i * stripe + (k + r) * S, S);
				});
			});
		if (rc)
			return rc;
	}
	return GEC_OK;
}

int gec_reconstruct_batch(const gec_codec *c, size_t nblocks, const uint8_t *const *shards, uint8_t *const *out, size_t S,
			  int data_only)
{
	return reconstruct_batch_impl(c, nblocks, shards, out, S, data_only, nullptr, nullptr);
}

int gec_reconstruct_hash_batch(const gec_codec *c, size_t nblocks, const uint8_t *const *shards, uint8_t *const *out, size_t S,
			       int data_only, uint8_t *in_sums, uint8_t *out_sums)
{
	if (!in_sums || !out_sums)
		return fail(GEC_E_INVALID_ARG, "NULL checksum output");
	return reconstruct_batch_impl(c, nblocks, shards, out, S, data_only, in_sums, out_sums);
}

}  // extern "C"
