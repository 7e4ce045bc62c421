// bm_internal.hpp -- what the translation units of libgarage_block share.
//
// Layout of the library (include/garage_block.h is the contract; reference anchors are cited there):
//   bm_core.cpp     errors, zstd, the environment table, manager life cycle and settings, test hooks, metrics
//   bm_node.cpp     the storage nodes behind ShardRpc: memory- and directory-backed (shard files, two-phase replace)
//   bm_gather.cpp   the shard gather every read-side path starts with (plain and hedged), PutShard to one node
//   bm_rw.cpp       rpc_put_block(s) / rpc_get_block(s): fan-out, the one-trip device calls, assembly, retries during a layout change
//   bm_stream.cpp   the streaming gets and the ranged get
//   bm_resync.cpp   refcounts (RcEntry), the resync queue and its workers, resync_block(s), list-errors / retry-now
//   bm_scrub.cpp    the scrub's steps, gbm_scrub, gbm_scrub_all, gbm_repair_all (RepairWorker)
//   bm_scrub_worker.cpp  the ScrubWorker's life cycle: commands, schedule, persisted checkpoint
//   bm_batcher.cpp  the coalescing queue in front of the FFI (PUT_BLOCKS_MAX_PARALLEL callers -> device batches)
// Pure host code: every shard byte and every large checksum batch is computed by libgarage_ec (include/garage_ec.h).
#pragma once

#include "../../include/garage_block.h"

#include <pthread.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <exception>
#include <functional>
#include <future>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <set>
#include <string>
#include <thread>
#include <tuple>
#include <unordered_map>
#include <unordered_set>
#include <vector>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "blake2b_mb.hpp"
#include "mlh64_host.hpp"

namespace gbmimpl {

// sets the calling thread's gbm_last_error() text and returns `code`
int fail(int code, const std::string &msg);
// function-try-block tail for the C entry points that allocate outside a try of their own: nothing may unwind across the C ABI
#define GBM_CATCH                                                                \
	catch (const std::exception &e)                                          \
	{                                                                        \
		try {                                                            \
			return ::gbmimpl::fail(GBM_E_IO, e.what());              \
		} catch (...) {                                                  \
			return GBM_E_IO;                                         \
		}                                                                \
	}                                                                        \
	catch (...)                                                              \
	{                                                                        \
		return GBM_E_IO;                                                 \
	}
const std::string &last_error();
int ec_fail(int rc, const char *what);

using b2host::blake2sum;
// One shard's checksum on this core, by shard-header version: 3 = MLH64 (mlh64_host.hpp: memory speed), 2 = BLAKE2b tree
// mode (its leaves are independent chains, eight at a time where the core can, blake2b_mb.hpp), 1 = plain blake2sum.
inline void shardsum_v(int ver, const uint8_t *data, size_t len, uint8_t out[32])
{
	if (ver == 3)
		mlh::shardsum3(data, len, out);
	else if (ver == 1)
		blake2sum(data, len, out);
	else
		b2host::shardsum_many(&data, &len, 1, out);
}

// ------------------------------------------------------------------ environment (bm_core.cpp holds the one table)
struct Env {
	bool trace;               // GBM_TRACE
	size_t put_slice;         // GBM_PUT_SLICE
	int put_threads;          // GBM_PUT_THREADS
	int batcher_workers;      // GBM_BATCHER_WORKERS
	size_t batcher_split_min; // GBM_BATCHER_SPLIT_MIN
	size_t batcher_get_split_min; // GBM_BATCHER_GET_SPLIT_MIN
	unsigned put_spot_check;  // GBM_PUT_SPOT_CHECK
};
const Env &env();
const char *env_table_text();

// --------------------------------------------------------------------- zstd
// DataBlock::from_buffer / zstd_encode (src/block/block.rs:85-106): one frame, level
// from the config, content checksum ON.  This image ships libzstd.so.1 but no headers,
// so the handful of entry points are resolved at run time.
struct Zstd {
	bool ok = false;
	// false on any error: the caller then stores the block Plain (block.rs:88-93)
	bool encode(const uint8_t *data, size_t len, int level, std::vector<uint8_t> &out) const;
	// verifies the frame checksum; false = corrupt.  `max_out` bounds the allocation.
	bool decode(const uint8_t *data, size_t len, size_t max_out, std::vector<uint8_t> &out) const;
	Zstd();
	// Incremental decoder of ONE frame (ZSTD_decompressStream): a compressed block is decoded shard by shard as its
	// shards are verified, and its plain bytes leave in chunks -- the reference streams through an async zstd decoder
	// the same way (src/block/manager.rs:344-363).  `streaming` is false when the library lacks the entry points.
	bool streaming = false;
	struct Stream {
		const Zstd *z;
		void *ds = nullptr;
		bool frame_done = false;
		explicit Stream(const Zstd &zz);
		~Stream();
		// feeds `in`; decoded bytes are appended to out (at most out_cap more bytes are accepted: kMaxDecompressed guard).
		// false = corrupt frame
		bool feed(const uint8_t *in, size_t len, const std::function<bool(const uint8_t *, size_t)> &emit);
	};

private:
	void *(*createCCtx)() = nullptr;
	size_t (*freeCCtx)(void *) = nullptr;
	size_t (*setParameter)(void *, int, int) = nullptr;
	size_t (*compress2)(void *, void *, size_t, const void *, size_t) = nullptr;
	size_t (*compressBound)(size_t) = nullptr;
	size_t (*decompress)(void *, size_t, const void *, size_t) = nullptr;
	unsigned long long (*getFrameContentSize)(const void *, size_t) = nullptr;
	unsigned (*isError)(size_t) = nullptr;
	void *(*createDStream)() = nullptr;
	size_t (*freeDStream)(void *) = nullptr;
	size_t (*initDStream)(void *) = nullptr;
	size_t (*decompressStream)(void *, void *, void *) = nullptr;
};
const Zstd &zstd();

using Hash = std::string;  // 32 raw bytes

inline std::string hex(const Hash &h)
{
	static const char *d = "0123456789abcdef";
	std::string s;
	for (unsigned char c : h) {
		s.push_back(d[c >> 4]);
		s.push_back(d[c & 15]);
	}
	return s;
}

// ------------------------------------------------------------- shard header
// 64-byte layout "<4sBBBBB3xQII32s" (tests/patterns.py parses it).
// The version names the checksum: 3 = MLH64 (GEC_SHARDSUM_MLH64, what a manager over a default codec writes), 2 = BLAKE2b
// tree mode (rounds 2-4), 1 = plain blake2sum (round 1).  A manager WRITES the version of its codec's checksum kind
// (gbm_manager::sumver) and reads all three: a shard of another version than its own is verified on the host with that
// version's checksum the first time it is read, then carried -- and rewritten on its node -- in the manager's own version,
// so that everything downstream of the gather sees one format and a store migrates as it is read (and as scrub walks it).
// Any other version is a format this build does not know: the shard is left alone (never renamed or deleted) and reported
// as unreadable.
struct ShardHeader {
	uint8_t version = 3;
	uint8_t k = 0, m = 0, idx = 0, compressed = 0;
	uint64_t orig_len = 0;
	uint32_t shard_len = 0;
	uint8_t checksum[32] = {0};

	enum Parse { OK = 0, GARBAGE = 1, UNKNOWN_VERSION = 2 };

	void pack(uint8_t out[GBM_SHARD_HEADER_SIZE]) const
	{
		std::memset(out, 0, GBM_SHARD_HEADER_SIZE);
		std::memcpy(out, "GECS", 4);
		out[4] = version;
		out[5] = k;
		out[6] = m;
		out[7] = idx;
		out[8] = compressed;
		std::memcpy(out + 12, &orig_len, 8);
		std::memcpy(out + 20, &shard_len, 4);
		std::memcpy(out + 28, checksum, 32);
	}
	Parse unpack(const uint8_t *in, size_t n)
	{
		if (n < GBM_SHARD_HEADER_SIZE || std::memcmp(in, "GECS", 4) != 0)
			return GARBAGE;
		version = in[4];
		if (version < 1 || version > 3)
			return UNKNOWN_VERSION;
		k = in[5];
		m = in[6];
		idx = in[7];
		compressed = in[8];
		std::memcpy(&orig_len, in + 12, 8);
		std::memcpy(&shard_len, in + 20, 4);
		std::memcpy(checksum, in + 28, 32);
		return OK;
	}
	bool same_geometry(const ShardHeader &o) const
	{
		return compressed == o.compressed && orig_len == o.orig_len && shard_len == o.shard_len;
	}
};

uint64_t real_now_ms();

// ------------------------------------------------------------------ thread pool
// fork-join: fn(i) for i in [0, n) on the workers and the calling thread.  Several callers may use it at
// once (they queue on call_mu_); work items must not call parallel_for themselves.
// Thread names (top -H, perf, tools/cpu_where.py): every thread the library starts says what it is.
inline void name_thread(const char *name)
{
#ifdef __linux__
	(void)pthread_setname_np(pthread_self(), name);  // <= 15 characters
#else
	(void)name;
#endif
}

// Threads a lane starts for itself run on its codec's memory node (gec_numa_bind_thread, include/garage_ec.h: a no-op for a CPU
// codec, a one-node box, GEC_NUMA=0): what they touch -- the lane's pinned shard buffers, the codec's staging slots -- lives there.
inline void lane_thread(const char *name, const gec_codec *near)
{
	name_thread(name);
	if (near)
		(void)gec_numa_bind_thread(near);
}

class Pool {
public:
	explicit Pool(unsigned n, const gec_codec *near = nullptr) : near_(near) { resize(n); }
	~Pool() { stop_all(); }
	void resize(unsigned n)
	{
		std::lock_guard<std::mutex> call(call_mu_);
		stop_all();
		stop_ = false;
		for (unsigned i = 0; i < n; ++i)
			workers_.emplace_back([this] {
				lane_thread("gbm-pool", near_);
				run();
			});
		nworkers_ = n;
	}
	unsigned workers() const { return nworkers_.load(std::memory_order_relaxed); }
	void parallel_for(size_t n, const std::function<void(size_t)> &fn)
	{
		if (n == 0)
			return;
		if (n == 1) {
			fn(0);
			return;
		}
		std::unique_lock<std::mutex> call_lock(call_mu_);
		if (workers_.empty()) {
			call_lock.unlock();
			for (size_t i = 0; i < n; ++i)
				fn(i);
			return;
		}
		{
			std::lock_guard<std::mutex> g(mu_);
			fn_ = &fn;
			n_ = n;
			next_ = 0;
			pending_ = n;
			err_ = nullptr;
			++epoch_;
		}
		cv_.notify_all();
		work();
		std::unique_lock<std::mutex> g(mu_);
		done_cv_.wait(g, [this] { return pending_ == 0; });
		fn_ = nullptr;
		// An item that threw (bad_alloc, as a rule): the items not yet started were skipped, every thread has left fn -- whose
		// captures live in the caller's frame -- and the first exception goes on from HERE, on the caller's thread, where the
		// C ABI's catch turns it into a code.  (Thrown straight out of work() it unwound that frame under the workers' feet;
		// thrown on a worker it was std::terminate.)
		if (err_) {
			std::exception_ptr e = err_;
			err_ = nullptr;
			g.unlock();
			std::rethrow_exception(e);
		}
	}

private:
	void stop_all()
	{
		{
			std::lock_guard<std::mutex> g(mu_);
			stop_ = true;
		}
		cv_.notify_all();
		for (auto &t : workers_)
			t.join();
		workers_.clear();
	}
	void work()
	{
		for (;;) {
			size_t i;
			const std::function<void(size_t)> *fn;
			bool skip;
			{
				std::lock_guard<std::mutex> g(mu_);
				if (!fn_ || next_ >= n_)
					return;
				i = next_++;
				fn = fn_;
				skip = (bool)err_;
			}
			if (!skip) {
				try {
					(*fn)(i);
				} catch (...) {
					std::lock_guard<std::mutex> g(mu_);
					if (!err_)
						err_ = std::current_exception();
				}
			}
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
	const gec_codec *near_ = nullptr;  // declared before the workers: they read it as they start
	std::atomic<unsigned> nworkers_{0};
	std::vector<std::thread> workers_;
	std::mutex mu_, call_mu_;
	std::condition_variable cv_, done_cv_;
	const std::function<void(size_t)> *fn_ = nullptr;
	size_t n_ = 0, next_ = 0, pending_ = 0;
	uint64_t epoch_ = 0;
	bool stop_ = false;
	std::exception_ptr err_;  // the first exception of the call in flight (under mu_)
};

// Fire-and-forget tasks for requests that may be abandoned (hedged reads): a task owns everything it touches
// through shared_ptrs, so nobody has to wait for a slow one.
class Async {
public:
	explicit Async(unsigned n, const gec_codec *near = nullptr)
	{
		for (unsigned i = 0; i < n; ++i)
			workers_.emplace_back([this, near] {
				lane_thread("gbm-async", near);
				run();
			});
	}
	~Async()
	{
		{
			std::lock_guard<std::mutex> g(mu_);
			stop_ = true;
		}
		cv_.notify_all();
		for (auto &t : workers_)
			t.join();
	}
	void submit(std::function<void()> fn)
	{
		{
			std::lock_guard<std::mutex> g(mu_);
			q_.push_back(std::move(fn));
		}
		cv_.notify_one();
	}

private:
	void run()
	{
		for (;;) {
			std::function<void()> fn;
			{
				std::unique_lock<std::mutex> g(mu_);
				cv_.wait(g, [&] { return stop_ || !q_.empty(); });
				if (q_.empty())
					return;  // stop_ and drained
				fn = std::move(q_.front());
				q_.pop_front();
			}
			try {
				fn();
			} catch (...) {
				// last resort (a task reports its own failures to whoever waits for it): an exception that leaves a
				// thread function is std::terminate for the whole process
			}
		}
	}
	std::vector<std::thread> workers_;
	std::mutex mu_;
	std::condition_variable cv_;
	std::deque<std::function<void()>> q_;
	bool stop_ = false;
};

// ------------------------------------------------------------------ buffers
// Bytes = shared, immutable-after-fill byte range; slices alias their parent (std::shared_ptr aliasing
// constructor), so a data shard is a view into its block's buffer and lives as long as any node keeps it.
struct Bytes {
	std::shared_ptr<uint8_t> p;
	size_t n = 0;
	const uint8_t *data() const { return p.get(); }
	uint8_t *mut() const { return p.get(); }
	bool empty() const { return !p; }
	Bytes slice(size_t off, size_t len) const
	{
		Bytes b;
		b.p = std::shared_ptr<uint8_t>(p, p.get() + off);
		b.n = len;
		return b;
	}
};

// Pool of pinned host buffers (gec_host_alloc): hipHostMalloc costs ~0.1 ms per MiB, so buffers are recycled
// by size.  On a host without a device gec_host_alloc hands out page-aligned ordinary memory (include/garage_ec.h).
class BufPool : public std::enable_shared_from_this<BufPool> {
public:
	static constexpr size_t kRetainMax = 2ull << 30;  // bytes kept for reuse; beyond that buffers are freed
	const gec_codec *near = nullptr;                  // the lane's codec (set once, before the first get)
	~BufPool()
	{
		for (auto &kv : free_)
			for (uint8_t *p : kv.second)
				gec_host_free(p);
	}
	Bytes get(size_t n)
	{
		const size_t cap = std::max<size_t>((n + 4095) / 4096 * 4096, 4096);
		uint8_t *raw = nullptr;
		{
			std::lock_guard<std::mutex> g(mu_);
			auto it = free_.find(cap);
			if (it != free_.end() && !it->second.empty()) {
				raw = it->second.back();
				it->second.pop_back();
				retained_ -= cap;
			}
		}
		if (!raw)  // (pinned, on the lane's codec's memory node whatever device the calling thread has current)
			raw = static_cast<uint8_t *>(near ? gec_host_alloc_near(near, cap) : gec_host_alloc(cap));
		if (!raw)
			throw std::bad_alloc();
		std::weak_ptr<BufPool> self = shared_from_this();
		Bytes b;
		b.n = n;
		b.p = std::shared_ptr<uint8_t>(raw, [self, cap](uint8_t *q) {
			if (auto sp = self.lock())
				sp->put_back(q, cap);
			else
				gec_host_free(q);
		});
		return b;
	}

private:
	void put_back(uint8_t *q, size_t cap)
	{
		{
			std::lock_guard<std::mutex> g(mu_);
			if (retained_ + cap <= kRetainMax) {
				free_[cap].push_back(q);
				retained_ += cap;
				return;
			}
		}
		gec_host_free(q);
	}
	std::mutex mu_;
	std::map<size_t, std::vector<uint8_t *>> free_;
	size_t retained_ = 0;
};

// ------------------------------------------------------------------ ShardRpc
// What the manager and a storage node say to each other: the per-shard analogue of BlockRpc
// (src/block/manager.rs:54-73).  PutShard carries the shard header (the DataBlockHeader plus the EC
// geometry and the shard's checksum) and the payload; GetShard is answered with a PutShard.
// A PutShard that would REPLACE a shard of a different geometry (the same block put again with another
// compression setting) is parked beside it and only takes its place on CommitShard, which the manager sends once
// the new stripe has reached its write quorum (AbortShard otherwise): a put that fails half way can no longer
// leave a block with too few shards of either geometry.
struct Shard {
	ShardHeader hd;
	Bytes data;  // shard_len bytes
};

enum class RpcKind { PutShard, GetShard, NeedShardQuery, DeleteShard, CommitShard, AbortShard };

struct ShardRpc {
	RpcKind kind;
	const Hash *hash;
	int idx;
	Shard shard;                        // PutShard
	const gbm_order_tag *tag = nullptr; // PutShard / GetShard
};

struct ShardResp {
	bool ok = false;      // PutShard stored / GetShard found / DeleteShard had something to delete
	bool needed = false;  // NeedShardReply
	bool have_hd = false; // NeedShardReply of a shard that is there: shard.hd holds its header (no payload)
	bool pending = false; // PutShard: parked beside a shard of another geometry, waiting for CommitShard
	Shard shard;          // answer to GetShard
};

struct Node {
	std::atomic<bool> down{false};  // flipped by gbm_node_set_down while other threads are talking to the node
	std::atomic<uint64_t> order_violations{0};
	std::atomic<uint64_t> latency_us{0};  // test hook: every request to this node takes this long
	std::atomic<uint64_t> ping_us{0};     // the node's avg_ping as the requester's peering knows it (gbm_node_set_ping; 0 = unknown)
	std::atomic<int> zone{0};             // the node's zone in the layout (gbm_node_set_zone; LayoutVersion::get_node_zone)
	std::atomic<uint64_t> requests{0};    // test hook: requests this node has been handed (down or not)
	std::shared_ptr<BufPool> bufs;
	virtual ~Node() = default;
	// stores the shard; *pending = it was parked because a shard of another geometry is in place
	virtual bool put(const Hash &h, int idx, const Shard &s, bool *pending = nullptr) = 0;
	virtual bool get(const Hash &h, int idx, Shard &s) = 0;  // false: absent / unreadable
	virtual bool has(const Hash &h, int idx) = 0;
	// the header of the shard in place, without its payload (false: no such shard, or a header that does not parse): what
	// the presence scan of a resync learns a shard's GEOMETRY from -- shards of a block are only usable together when they
	// were cut from the same payload the same way
	virtual bool header(const Hash &h, int idx, ShardHeader &hd) = 0;
	virtual bool del(const Hash &h, int idx) = 0;
	virtual bool commit(const Hash &h, int idx) = 0;  // a parked shard takes the place of the one in place
	virtual bool abort(const Hash &h, int idx) = 0;   // a parked shard is dropped
	virtual void mark_corrupted(const Hash &h, int idx) { del(h, idx); }
	virtual void set_fsync(bool) {}
	// every hash this node holds a shard of (BlockStoreIterator, src/block/repair.rs:196-233,634-752)
	virtual void list(std::set<Hash> &out) = 0;
	// the same, restricted to the hashes whose first byte is h0: one first-level directory of the store, the unit the
	// ScrubWorker's iterator walks and checkpoints by (BsiTodo::Directory, src/block/repair.rs:196-233,684-752)
	virtual void list_prefix(int h0, std::set<Hash> &out) = 0;
	// several consecutive first-level directories, lo <= h0 < hi (a sparse store is walked a few directories at a time)
	virtual void list_prefix_range(int lo, int hi, std::set<Hash> &out)
	{
		for (int p = lo; p < hi; ++p)
			list_prefix(p, out);
	}

	// the node's endpoint (StreamingEndpointHandler<BlockRpc>::handle, src/block/manager.rs:692-707);
	// false = could not be contacted
	bool handle(const ShardRpc &rq, ShardResp &rs);

private:
	void note_order(const gbm_order_tag *tag);
	std::mutex order_mu_;
	std::map<uint64_t, uint64_t> last_order_;
};

std::unique_ptr<Node> make_memory_node();
std::unique_ptr<Node> make_dir_node(const std::string &root);

// RcEntry (src/block/rc.rs:122-240)
struct RcEntry {
	enum Kind : uint8_t { Absent = 0, Present = 1, Deletable = 2 } kind = Absent;
	uint64_t v = 0;  // Present: count; Deletable: at_time (ms)
	bool is_zero() const { return kind != Present; }
	bool is_nonzero() const { return kind == Present; }
	bool is_deletable(uint64_t now) const { return kind == Absent || (kind == Deletable && now > v); }
	bool is_needed(uint64_t now) const { return !is_deletable(now); }
};

// ErrorCounter (src/block/resync.rs:604-648)
struct ErrorCounter {
	uint64_t errors = 0, last_try = 0;
	uint64_t delay_ms(uint64_t base) const
	{
		return base << std::min<uint64_t>(errors - 1, GBM_RESYNC_RETRY_MAX_BACKOFF_POWER);
	}
	uint64_t next_try(uint64_t base) const { return last_try + delay_ms(base); }
};

struct ScrubWorker;  // bm_scrub.cpp

// BlockManagerMetrics (src/block/metrics.rs:10-143): the value recorders as histograms over the boundaries the
// reference's Prometheus exporter is set up with (src/garage/server.rs:36-44), the counters the manager's metrics[] do
// not already hold.  Everything is a relaxed atomic: a recorder costs two increments on the request path.
struct Histogram {
	static constexpr int NB = GBM_HISTOGRAM_BUCKETS;
	static const double *bounds()
	{
		static const double b[NB] = {0.001, 0.0015, 0.002, 0.003, 0.005, 0.007, 0.01, 0.015, 0.02, 0.03, 0.05, 0.07, 0.1, 0.15, 0.2, 0.3, 0.5,
					     0.7,   1.,     1.5,   2.,    3.,    5.,    7.,   10.,   15.,  20.,  30.,  40.,  50., 60.,  70., 100.};
		return b;
	}
	std::atomic<uint64_t> bucket[NB + 1] = {};  // bucket[i]: observations in (bounds[i-1], bounds[i]]; [NB]: above the last bound
	std::atomic<uint64_t> sum_ns{0};  // (no separate count: a snapshot's count IS the sum of the buckets it read -- what the exposition
					  // format demands of the +Inf bucket -- however many observations land while it is being taken)
	void record(std::chrono::nanoseconds d)
	{
		const double s = (double)d.count() * 1e-9;
		const double *b = bounds();
		int i = 0;
		while (i < NB && s > b[i])
			++i;
		bucket[i].fetch_add(1, std::memory_order_relaxed);
		sum_ns.fetch_add((uint64_t)std::max<int64_t>(0, d.count()), std::memory_order_relaxed);
	}
};
// RecordDuration (src/util/metrics.rs:8-57) as a scope: the time from its construction to the end of the scope
struct DurationScope {
	Histogram *h;
	std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
	explicit DurationScope(Histogram &hist) : h(&hist) {}
	void cancel() { h = nullptr; }
	~DurationScope()
	{
		if (h)
			h->record(std::chrono::steady_clock::now() - t0);
	}
};
struct BlockMetrics {
	std::atomic<uint64_t> resync_counter{0}, resync_error_counter{0}, resync_send_counter{0}, resync_recv_counter{0}, delete_counter{0};
	std::atomic<uint64_t> unconfirmed_verdicts{0};  // "checksum does not match" from a trip that the host's own check did not confirm
	std::atomic<uint64_t> put_spot_checks{0}, put_spot_check_failures{0};  // gbm_set_put_spot_check
	Histogram resync_duration, read_duration, write_duration;
};

}  // namespace gbmimpl

struct gbm_manager {
	using Hash = gbmimpl::Hash;
	using Node = gbmimpl::Node;
	using BufPool = gbmimpl::BufPool;
	using Pool = gbmimpl::Pool;
	using Async = gbmimpl::Async;
	using RcEntry = gbmimpl::RcEntry;
	using ErrorCounter = gbmimpl::ErrorCounter;

	const gec_codec *codec = nullptr;  // the request path's codec (borrowed)
	int sumver = 3;  // the shard-header version this manager writes = its codec's checksum kind (gec_codec_shardsum): 3 or 2
	// Maintenance (scrub, resync rebuilds) runs on a BACKGROUND-class sibling of `codec` (gec_codec_background): its
	// device work yields to the request path's.  Owned; NULL when the sibling could not be created (then == codec).
	gec_codec *bg_codec_owned = nullptr;
	std::atomic<bool> maintenance_on_bg{true};  // gbm_set_maintenance_class (A/B: what the class buys)
	const gec_codec *bg_codec() const { return bg_codec_owned && maintenance_on_bg.load() ? bg_codec_owned : codec; }
	// Tranquilizer (src/util/tranquilizer.rs:38-69): after each maintenance batch that took t, sleep tranquility * t
	// (resync.rs:46,568 reads its value from the persisted worker config; scrub has its own, repair.rs:386-390)
	std::atomic<uint32_t> scrub_tranquility{0}, resync_tranquility{0};
	std::atomic<uint64_t> tranquilized_ms{0};
	int k = 0, m = 0, n = 0, write_quorum = 0;
	std::vector<std::shared_ptr<Node>> nodes;  // shared by the lanes of a multi-device manager

	// Several devices on one node (gbm_create_multi).  The manager the caller holds is then a FRONT: it owns one complete
	// manager per device (a "lane": that device's foreground codec, its BACKGROUND sibling, its own pinned-buffer pool, host
	// pool, refcount stripes, mutation locks and resync queue) and routes every call by gec_device_of_hash(hash, ndev) --
	// the same trick the reference plays with hash bytes for partitions, drives and mutation locks
	// (src/rpc/layout/version.rs:101-104, src/block/layout.rs:278-284, src/block/manager.rs:679-689).  A hash belongs to
	// exactly one lane, so everything keyed by hash lives in that lane alone and two devices never share a lock on the
	// request path; only the storage nodes (which lock internally, per hash stripe) are common.
	std::vector<std::unique_ptr<gbm_manager>> lanes;  // non-empty: this is a front (codec == NULL, no pool of its own)
	int lane_idx = 0, lane_cnt = 1;                   // a lane owns the hashes with gec_device_of_hash(h, lane_cnt) == lane_idx
	bool is_front() const { return !lanes.empty(); }
	bool owns(const Hash &h) const
	{
		return lane_cnt == 1 || gec_device_of_hash(reinterpret_cast<const uint8_t *>(h.data()), lane_cnt) == lane_idx;
	}
	// the manager that serves `hash`: a lane of a front, the manager itself otherwise
	gbm_manager *route(const uint8_t *hash)
	{
		return lanes.empty() ? this : lanes[(size_t)gec_device_of_hash(hash, (int)lanes.size())].get();
	}
	const gbm_manager *route(const uint8_t *hash) const { return const_cast<gbm_manager *>(this)->route(hash); }
	std::shared_ptr<BufPool> bufs = std::make_shared<BufPool>();
	std::unique_ptr<Pool> pool;

	// cluster layout versions (src/rpc/layout/): reads consult [current .. oldest]
	std::atomic<int> layout_cur{0}, layout_oldest{0};

	// refcounts, striped like mutation_lock (manager.rs:679-689)
	static constexpr int kRcStripes = 256;
	struct RcStripe {
		std::mutex mu;
		std::unordered_map<Hash, RcEntry> map;
	};
	RcStripe rc[kRcStripes];
	RcStripe &rc_of(const Hash &h) { return rc[(((unsigned char)h[0] << 8) | (unsigned char)h[1]) % kRcStripes]; }
	// mutation_lock (manager.rs:679-689): whoever changes which shards of a hash exist -- a put's "this block is
	// protected from now on" stamp, resync's delete branch -- does so under the hash's lock, and resync re-reads
	// the refcount under it right before it deletes (delete_if_unneeded, manager.rs:619-623,821-830)
	std::mutex mutation_lock[kRcStripes];
	std::mutex &lock_mutate(const Hash &h) { return mutation_lock[(((unsigned char)h[0] << 8) | (unsigned char)h[1]) % kRcStripes]; }

	// resync.queue / resync.errors (resync.rs:170-253)
	mutable std::mutex rs_mu;
	std::condition_variable rs_cv;
	std::set<std::pair<uint64_t, Hash>> rs_queue;
	std::unordered_map<Hash, ErrorCounter> rs_errors;
	// ResyncWorker x n_workers (resync.rs:513-602; `resync-worker-count`, 1..MAX_RESYNC_WORKERS, :136-152): each runs passes
	// over what is due; a hash one pass has taken is in the busy set until that pass is over (BusySet, :74-85,339-352)
	std::vector<std::thread> rs_workers;
	bool rs_worker_stop = false;
	int rs_n_workers = 1;
	std::unordered_set<Hash> rs_busy;
	std::string rs_cfg_path;  // ResyncPersistedConfig's file ("" = not persisted); gbm_resync_config_persist
	std::atomic<bool> resync_tranquility_set{false};

	std::atomic<uint64_t> gc_delay_ms{GBM_BLOCK_GC_DELAY_MS}, retry_delay_ms{GBM_RESYNC_RETRY_DELAY_MS},
		incref_delay_ms{2 * 300000ull};  // 2 * rpc_timeout, DEFAULT_TIMEOUT = 300 s (rpc_helper.rs:33)
	std::atomic<uint64_t> clock_skew_ms{0};
	uint64_t now() const { return gbmimpl::real_now_ms() + clock_skew_ms.load(); }

	// ScrubWorkerPersisted (src/block/repair.rs:169-194)
	std::atomic<uint64_t> scrub_corruptions{0}, scrub_last_complete_ms{0};
	// the continuously running ScrubWorker (repair.rs:156-500): gbm_scrub_worker_start creates it (one per lane)
	std::mutex scrub_worker_mu;
	std::shared_ptr<gbmimpl::ScrubWorker> scrub_worker;
	std::atomic<bool> scrub_tranquility_set{false};  // gbm_set_tranquility has been called: INITIAL_SCRUB_TRANQUILITY does not apply
	// Put batches whose device-computed shard checksums are spot-checked on the host (every Nth; 0 = never), and -- a test hook --
	// how many of the next put trips come back with one checksum falsified (a device fault, simulated)
	std::atomic<uint32_t> put_spot_every{16};
	std::atomic<uint64_t> put_trips{0};
	std::atomic<int> test_bad_put_sums{0};
	std::atomic<uint64_t> metrics[6] = {};
	gbmimpl::BlockMetrics bmx;  // the rest of BlockManagerMetrics (gbm_block_metrics_get, gbm_metrics_prometheus)
	std::atomic<uint64_t> gpu_hashed{0};
	std::atomic<bool> compress{false};    // Config.compression_level (src/util/config.rs:52-58); Garage's default is Some(1)
	std::atomic<int> compression_level{1};
	// the requester's end-to-end block hash (gbm_set_verify_block_hash).  Default by shard-header version (set at creation):
	// ALWAYS over MLH64 shard checksums (version 3: fast but not cryptographic, so the block's own name is checked on every Plain
	// read as the reference's read path does, block.rs:69-76 / manager.rs:592), REBUILT over the BLAKE2b tree (version 2)
	std::atomic<int> verify_mode{GBM_VERIFY_REBUILT};
	// a shard of another header version met by a READ is verified and carried in this manager's version but rewritten on its node
	// only when this is set (gbm_set_migrate_on_read); scrub and resync always migrate what they touch
	std::atomic<bool> migrate_on_read{false};
	std::atomic<uint64_t> shards_migrated{0};
	std::atomic<uint64_t> shared_host_rate{2000000000ull};  // bytes/s one pool thread checks + assembles + hashes (measured by the shared form)
	std::atomic<size_t> cpu_block_hash_max{96};  // gets of up to this many blocks hash them on the host (gbm_set_threads rescales)

	// who is asking (request_order's our_node_id / our_zone, rpc_helper.rs:626-628): -1 = not one of the storage nodes
	std::atomic<int> self_node{-1}, self_zone{0};
	std::atomic<bool> locality_set{false};  // a zone, a ping or the requester's identity has been given: the streaming forms consult the order
	// hedged reads: 0 = the k requests of a read are issued and awaited in order
	std::atomic<uint64_t> hedge_us{0}, hedged_reads{0};
	std::mutex async_mu;
	std::shared_ptr<Async> async;  // created when hedging is first switched on
	std::shared_ptr<Async> async_pool()
	{
		std::lock_guard<std::mutex> g(async_mu);
		if (!async)
			async = std::make_shared<Async>(32, codec);
		return async;
	}

	// storage nodes of a hash in layout version v: a deterministic stand-in for
	// ClusterLayout::storage_nodes_of (partition = top bits of the hash, src/rpc/layout/version.rs:101-118)
	void nodes_of(const Hash &h, int version, std::vector<int> &who) const
	{
		const size_t N = nodes.size();
		const size_t start = ((unsigned char)h[0] * 31u + (unsigned char)h[1] + (size_t)version * (N / 2 + 1)) % N;
		who.resize(n);
		for (int j = 0; j < n; ++j)
			who[j] = (int)((start + j) % N);
	}
	void nodes_of(const Hash &h, std::vector<int> &who) const { nodes_of(h, layout_cur.load(), who); }

	RcEntry get_rc(const Hash &h)
	{
		RcStripe &s = rc_of(h);
		std::lock_guard<std::mutex> g(s.mu);
		auto it = s.map.find(h);
		return it == s.map.end() ? RcEntry() : it->second;
	}
	void put_to_resync_at(const Hash &h, uint64_t when)
	{
		{
			std::lock_guard<std::mutex> g(rs_mu);
			rs_queue.insert({when, h});
		}
		rs_cv.notify_all();
	}
	void put_to_resync(const Hash &h, uint64_t delay) { put_to_resync_at(h, now() + delay); }
};

namespace gbmimpl {

// GBM_TRACE=1: stage timings of the batched put / get on stderr (tools/host_path_bench.py reads them off)
struct Trace {
	const char *what;
	std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
	std::string line;
	explicit Trace(const char *w) : what(w) {}
	void lap(const char *stage);
	~Trace();
};

// largest block a Garage node will ever hold decompresses to: block_size is configurable, 1 MiB by default and
// "a few MiB" in practice; 1 GiB is far above any of it and still a harmless allocation bound
constexpr size_t kMaxDecompressed = 1ull << 30;

// shard checksums of many buffers: on the GPU (gec_shardsum_batch) once the batch is big enough to beat the
// CPU pool through PCIe, else on the pool's threads.  SURVEY.md section 8 row f4.
int hash_many(gbm_manager *mg, const std::vector<const uint8_t *> &ptrs, const std::vector<size_t> &lens, std::vector<uint8_t> &sums);

// fn(lane, lane index) for every lane of a front, side by side on threads of their own; the last non-zero result, its
// error text re-published on the calling thread (bm_core.cpp)
int for_lanes(gbm_manager *front, const std::function<int(gbm_manager *, size_t)> &fn);
// block indices [0, nb) of a batch call on a front, by owning lane, in the caller's order
inline std::vector<std::vector<size_t>> split_by_lane(const gbm_manager *front, size_t nb, const uint8_t *hashes)
{
	std::vector<std::vector<size_t>> ids(front->lanes.size());
	for (size_t b = 0; b < nb; ++b)
		ids[(size_t)gec_device_of_hash(hashes + 32 * b, (int)front->lanes.size())].push_back(b);
	return ids;
}

// Shards of one block are only usable together when they were cut from the same
// payload with the same geometry.  A block can legitimately have shards of two
// geometries on disk at once -- e.g. it was first stored Plain and a later put with
// compression enabled reached only some nodes before failing its quorum -- so shards
// are grouped by geometry and the largest consistent group is used (find_block makes
// the same kind of choice between <hash> and <hash>.zst, manager.rs:627-662).
struct Geometry {
	uint8_t compressed = 0;
	uint64_t orig_len = 0;
	uint32_t shard_len = 0;
	bool operator<(const Geometry &o) const
	{
		return std::tie(compressed, orig_len, shard_len) < std::tie(o.compressed, o.orig_len, o.shard_len);
	}
};

struct Gathered {
	std::vector<Bytes> shard;  // n entries; empty = not in hand
	std::vector<std::array<uint8_t, 32>> sum;  // the checksum each shard's header promises
	std::vector<int> node;                     // where each shard came from
	ShardHeader meta;
	bool have_meta = false;
	int count = 0;
	std::vector<uint8_t> tried;  // candidates (version-major, shard index minor) that have been asked
	bool mixed = false;
	bool settled = false;  // a geometry has been chosen; later candidates must match it
	bool corrupt_seen = false;  // a shard of this block was there but failed its header / checksum check during this read
	bool down_seen = false;     // a holder of this block could not be contacted during this read (its shard may exist)
	std::vector<uint32_t> order;  // the candidates (index = version-major, shard minor) in the order they are asked (read_candidate_order)
	struct Group {
		ShardHeader meta;
		std::vector<Bytes> shard;
		std::vector<std::array<uint8_t, 32>> sum;
		std::vector<int> node;
		int count = 0;
	};
	std::map<Geometry, Group> groups;
	int best() const
	{
		int c = 0;
		for (auto &kv : groups)
			c = std::max(c, kv.second.count);
		return c;
	}
	bool have_idx(int j) const
	{
		for (auto &kv : groups)
			if (!kv.second.shard[j].empty())
				return true;
		return false;
	}
};

// Fetch shards until every block has `want` valid ones of one geometry in hand (or ran out of nodes) -- bm_gather.cpp
int gather_many(gbm_manager *mg, const std::vector<Hash> &hs, const gbm_order_tag *tags, int want, std::vector<Gathered> &gs,
		bool verify = true, const std::vector<uint8_t> *only = nullptr, bool migrate = false);
// The order in which a read asks the holders of a block's shards (block_read_nodes_of + request_order, rpc_helper.rs:570-660,
// applied to shards): `order` = candidate indices c = (version - vold) * n + shard -- bm_gather.cpp
void read_candidate_order(const gbm_manager *mg, const Hash &h, int vold, int vcur, std::vector<uint32_t> &order);
// PutShard to one node; false = the node could not be contacted or refused
bool send_shard(gbm_manager *mg, int node, const Hash &h, int idx, const Bytes &payload, size_t S, uint64_t orig_len, bool compressed,
		const uint8_t *checksum, const gbm_order_tag *tag, bool *pending = nullptr);
// order gate of the batcher: batches that carry order tags hand their shards to the nodes in the order they were formed
struct FanoutGate {
	std::function<void()> before, after;  // around the whole fan-out of a batch that carries tags (one device)
	// around the batch's device trip: the batcher lets ONE batch of a device onto the link at a time -- two trips that
	// share it only lengthen each other, while a batch that waits its turn leaves its worker's host stages (copy into
	// the shard buffers, fan-out) overlapping the other batch's trip
	std::function<void()> device_enter, device_exit;
	// several devices: blocks of one OrderTag stream are encoded on different devices, in different batches.  Tagged
	// blocks then go out one block at a time, in `block_order` (the order the blocks were submitted in, over all
	// devices), each one between before_block(b) -- returns once the stream's previous block has reached its nodes,
	// whichever device had it -- and after_block(b).
	const std::vector<size_t> *block_order = nullptr;
	std::function<void(size_t)> before_block, after_block;
	// gets through the coalescing queue under GBM_VERIFY_ALWAYS: the end-to-end hash of a healthy Plain block is left to the
	// CALLER that waits for it ((*defer_block_hash)[b] = 1: "delivered, not yet checked against its name") -- every reader hashes
	// its own block on its own thread, as every request of the reference verifies its own read, instead of the batch's one
	// pool doing all of them; blocks that went through a decode are still hashed by the batch
	std::vector<uint8_t> *defer_block_hash = nullptr;
};
// RAII form of device_enter / device_exit
struct DeviceTurn {
	const FanoutGate *g;
	explicit DeviceTurn(const FanoutGate *gate) : g(gate)
	{
		if (g && g->device_enter)
			g->device_enter();
	}
	~DeviceTurn()
	{
		if (g && g->device_exit)
			g->device_exit();
	}
};
// raw == true: rpc_get_raw_block (stored bytes + header); false: rpc_get_block (plain bytes).  bm_rw.cpp
int get_blocks_impl(gbm_manager *mg, size_t nb, const uint8_t *hashes, const gbm_order_tag *tags, uint8_t *const *out,
		    const size_t *cap, size_t *len_out, int *rcs, bool raw, gbm_data_block_header *headers,
		    const FanoutGate *gate = nullptr);
int put_blocks_impl(gbm_manager *mg, size_t nb, const uint8_t *hashes, const uint8_t *const *data, const size_t *len,
		    const uint8_t *prevent_compression, const gbm_order_tag *tags, int *rcs, const FanoutGate *gate = nullptr);
// hash_early(blocks, n, sums, ok): the blake2sum of n blocks the caller has ALREADY assembled (overlap_block), eight chains at a
// time; ok[i] = 0 where block i was not assembled (compressed, a short buffer): then nothing is written for it.  Given (with
// overlap_block), a big want_block_sums == 1 batch over header version 3 is SHARED between the device trip and the pool: the pool
// checks, assembles and hashes as many healthy blocks as it gets through while the device does the same for the rest.
using EarlyHashFn = std::function<void(const size_t *blocks, size_t n, uint8_t *sums, uint8_t *ok)>;
// want_block_sums: 0 = no, 1 = the blake2sum of every block, 2 = of the blocks of every trip that rebuilds something
// (block_sums[32*b..] is only meaningful where have_sum[b] is set)
// overlap: host work done while the first device trip is in flight (the caller's early assembly); overlap_block(b): the same work
// for ONE block -- given, a big batch whose shards are checked on the host (header version 3) does check and assembly of a block in
// one pool task (the shard is copied while it is still in that core's cache) instead of two passes that queue on the pool
int fetch_blocks(gbm_manager *mg, const std::vector<Hash> &hs, const gbm_order_tag *tags, std::vector<Gathered> &g, int *rcs,
		 int want_block_sums, std::vector<uint8_t> &block_sums, const std::function<void()> &overlap = nullptr,
		 std::vector<uint8_t> *changed = nullptr, const FanoutGate *gate = nullptr, std::vector<uint8_t> *have_sum = nullptr,
		 const std::function<void(size_t)> *overlap_block = nullptr, const EarlyHashFn *hash_early = nullptr);
void assemble(const Gathered &g, int k, uint8_t *dst);
int one_block_rc(int rc1);
// A shard is set aside (renamed *.corrupted, rebuilt by resync) only on the HOST's word.  Whoever found a checksum that does
// not match -- a device trip, the pool -- has the shard's bytes in hand, and corruption is rare: the verdict is confirmed
// with the host's own restatement of the checksum before anything is renamed.  A verdict that is not confirmed is counted
// (block_ec_unconfirmed_verdicts), logged, and the shard stays where it is: a fault in the checker must not become the loss
// of the k good shards it was shown.  true = the shard really does not match `header_sum`.  (bm_core.cpp)
bool confirmed_corrupt(gbm_manager *mg, const uint8_t *data, size_t S, const uint8_t header_sum[32], const char *who);
// every hash any reachable node holds a shard of (bm_scrub.cpp)
void list_all_nodes(gbm_manager *mg, std::set<Hash> &all);
// One step of the scrub (bm_scrub.cpp; gbm_scrub_all and the ScrubWorker take the same steps): a batch of hashes with their
// shards as the nodes hold them -- accepted on their headers: the checksums come back from the same device trip that checks the
// stripe against the code --, then that trip and the bookkeeping of what it found.  st: [0] blocks scrubbed, [1] corruptions
// detected, [2] device verify calls, [3] shards located and set aside; after_trip(t) is called behind every device trip with
// the time it took (the tranquilizer's place).
struct ScrubBatch {
	std::vector<Hash> batch;
	std::vector<Gathered> g;
	int rc = GBM_OK;
	std::string err;
};
ScrubBatch read_scrub_batch(gbm_manager *mg, std::vector<Hash> hashes);
int verify_scrub_batch(gbm_manager *mg, ScrubBatch &cur, uint64_t st[4], Trace &tr, const std::function<void(std::chrono::nanoseconds)> &after_trip);
// the coalescing queue's figures for the metrics: out[0] = free RAM permits (KiB), [1] put batches, [2] put blocks,
// [3] get batches, [4] get blocks; summed over the lanes of a front (bm_batcher.cpp)
void batcher_snapshot(gbm_batcher *b, uint64_t out[5]);
// gbm_set_tranquility changed the scrub's value: a running ScrubWorker persists it (repair.rs:26-27)
void scrub_worker_tranquility_changed(gbm_manager *mg);
// the resync workers' settings changed (gbm_set_tranquility, gbm_set_resync_workers): ResyncPersistedConfig is saved
void resync_config_changed(gbm_manager *mg);
// the clock moved (gbm_clock_advance): a pause may be over, the next run may be due
void scrub_worker_wake(gbm_manager *mg);

}  // namespace gbmimpl
