// block_manager.cpp -- libgarage_block.so (include/garage_block.h): C++ host-side
// mirror of garage_block::BlockManager with erasure-coded shard fan-out.
//
// Pure host code: it moves buffers between "nodes" and calls libgarage_ec's C ABI
// (gec_encode_hash_batch / gec_reconstruct_batch / gec_verify_batch /
// gec_blake2sum_batch) for every shard byte and every large hash batch it needs
// computed.  No GF arithmetic happens here.
//
// Shape (reference anchors in include/garage_block.h):
//  * buffers are reference-counted and never copied once they exist: a put copies the block ONCE into a
//    zero-padded k*S buffer (the k data shards are slices of it, the way the reference clones one `Bytes`
//    per peer, src/rpc/rpc_helper.rs:493), parity lands in one m*S buffer per block; both come from a pool
//    of pinned host memory (gec_host_alloc) so that libgarage_ec moves them by DMA without staging;
//  * the manager talks to its nodes in ShardRpc messages (PutShard / GetShard / NeedShardQuery /
//    DeleteShard), the per-shard analogue of BlockRpc (src/block/manager.rs:54-73);
//  * refcounts are RcEntry {Present, Deletable{at}, Absent} (src/block/rc.rs), the resync queue is ordered
//    by (due time, hash) with an ErrorCounter table for exponential back-off (src/block/resync.rs);
//  * bulk work (copies, compression, fan-out, gathers) runs on an internal thread pool; rc and queue
//    are striped / locked, nodes lock internally.
#include "../../include/garage_block.h"

#include <dirent.h>
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <sys/uio.h>
#include <unistd.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <future>
#include <thread>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <set>
#include <string>
#include <tuple>
#include <unordered_map>
#include <vector>

namespace {

thread_local std::string g_err;

int fail(int code, const std::string &msg)
{
	g_err = msg;
	return code;
}

// ------------------------------------------------------------------ blake2b
// RFC 7693, unkeyed, 64-byte digest.  Garage's blake2sum keeps the first 32 bytes
// of blake2b-512 (src/util/data.rs:130-138).
const uint64_t B2_IV[8] = {0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL, 0xa54ff53a5f1d36f1ULL,
			   0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
const uint8_t B2_SIGMA[12][16] = {
	{0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
	{11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
	{9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
	{12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
	{6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0},
	{0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3}};

inline uint64_t rotr64(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }

void b2_compress(uint64_t h[8], const uint8_t block[128], uint64_t t, bool last, bool last_node = false)
{
	uint64_t m[16], v[16];
	std::memcpy(m, block, 128);  // little-endian host
	for (int i = 0; i < 8; ++i) {
		v[i] = h[i];
		v[i + 8] = B2_IV[i];
	}
	v[12] ^= t;  // t fits 64 bits here
	if (last)
		v[14] = ~v[14];
	if (last && last_node)
		v[15] = ~v[15];  // f1, tree mode
#define B2_G(a, b, c, d, x, y)                  \
	v[a] = v[a] + v[b] + (x);               \
	v[d] = rotr64(v[d] ^ v[a], 32);         \
	v[c] = v[c] + v[d];                     \
	v[b] = rotr64(v[b] ^ v[c], 24);         \
	v[a] = v[a] + v[b] + (y);               \
	v[d] = rotr64(v[d] ^ v[a], 16);         \
	v[c] = v[c] + v[d];                     \
	v[b] = rotr64(v[b] ^ v[c], 63);
	for (int r = 0; r < 12; ++r) {
		const uint8_t *s = B2_SIGMA[r];
		B2_G(0, 4, 8, 12, m[s[0]], m[s[1]]);
		B2_G(1, 5, 9, 13, m[s[2]], m[s[3]]);
		B2_G(2, 6, 10, 14, m[s[4]], m[s[5]]);
		B2_G(3, 7, 11, 15, m[s[6]], m[s[7]]);
		B2_G(0, 5, 10, 15, m[s[8]], m[s[9]]);
		B2_G(1, 6, 11, 12, m[s[10]], m[s[11]]);
		B2_G(2, 7, 8, 13, m[s[12]], m[s[13]]);
		B2_G(3, 4, 9, 14, m[s[14]], m[s[15]]);
	}
#undef B2_G
	for (int i = 0; i < 8; ++i)
		h[i] ^= v[i] ^ v[i + 8];
}

// BLAKE2b-512 with an explicit parameter block (words 0..2) and the tree-mode "last node" flag; full digest out
void blake2b_params(const uint8_t *data, size_t len, uint64_t p0, uint64_t p1, uint64_t p2, bool last_node, uint8_t out[64])
{
	uint64_t h[8];
	for (int i = 0; i < 8; ++i)
		h[i] = B2_IV[i];
	h[0] ^= p0;
	h[1] ^= p1;
	h[2] ^= p2;
	size_t off = 0;
	while (len - off > 128) {
		b2_compress(h, data + off, off + 128, false);
		off += 128;
	}
	uint8_t last[128] = {0};
	if (len > off)  // data may be NULL for the empty message
		std::memcpy(last, data + off, len - off);
	b2_compress(h, last, len, true, last_node);
	std::memcpy(out, h, 64);
}

void blake2sum(const uint8_t *data, size_t len, uint8_t out[32])
{
	uint8_t full[64];
	blake2b_params(data, len, 0x01010000ULL ^ 64 /* digest length 64, no key, fanout 1, depth 1 */, 0, 0, false, full);
	std::memcpy(out, full, 32);
}

// The shard checksum: BLAKE2b tree mode, GEC_SHARDSUM_LEAF-byte leaves, unlimited fanout, depth 2, 64-byte inner
// digests, root truncated to 32 bytes (include/garage_ec.h has the definition and the reason).  CPU restatement for the
// few shards the manager checksums itself (a repair, a small read); batches go to gec_shardsum_batch.
void shardsum(const uint8_t *data, size_t len, uint8_t out[32])
{
	const uint64_t P0 = 64ull | (2ull << 24) | ((uint64_t)GEC_SHARDSUM_LEAF << 32);
	const size_t nleaf = len ? (len + GEC_SHARDSUM_LEAF - 1) / GEC_SHARDSUM_LEAF : 1;
	std::vector<uint8_t> digs(nleaf * 64);
	for (size_t i = 0; i < nleaf; ++i) {
		const size_t lo = i * GEC_SHARDSUM_LEAF, n = len > lo ? std::min<size_t>(GEC_SHARDSUM_LEAF, len - lo) : 0;
		blake2b_params(n ? data + lo : nullptr, n, P0, i, 64ull << 8, i + 1 == nleaf, digs.data() + 64 * i);
	}
	uint8_t full[64];
	blake2b_params(digs.data(), digs.size(), P0, 0, 1ull | (64ull << 8), true, full);
	std::memcpy(out, full, 32);
}

// --------------------------------------------------------------------- zstd
// DataBlock::from_buffer / zstd_encode (src/block/block.rs:85-106): one frame, level
// from the config, content checksum ON.  This image ships libzstd.so.1 but no headers,
// so the handful of entry points are resolved at run time.
struct Zstd {
	void *(*createCCtx)() = nullptr;
	size_t (*freeCCtx)(void *) = nullptr;
	size_t (*setParameter)(void *, int, int) = nullptr;
	size_t (*compress2)(void *, void *, size_t, const void *, size_t) = nullptr;
	size_t (*compressBound)(size_t) = nullptr;
	size_t (*decompress)(void *, size_t, const void *, size_t) = nullptr;
	unsigned long long (*getFrameContentSize)(const void *, size_t) = nullptr;
	unsigned (*isError)(size_t) = nullptr;
	bool ok = false;

	Zstd()
	{
		void *h = dlopen("libzstd.so.1", RTLD_NOW | RTLD_LOCAL);
		if (!h)
			return;
#define GBM_SYM(field, name) field = reinterpret_cast<decltype(field)>(dlsym(h, name))
		GBM_SYM(createCCtx, "ZSTD_createCCtx");
		GBM_SYM(freeCCtx, "ZSTD_freeCCtx");
		GBM_SYM(setParameter, "ZSTD_CCtx_setParameter");
		GBM_SYM(compress2, "ZSTD_compress2");
		GBM_SYM(compressBound, "ZSTD_compressBound");
		GBM_SYM(decompress, "ZSTD_decompress");
		GBM_SYM(getFrameContentSize, "ZSTD_getFrameContentSize");
		GBM_SYM(isError, "ZSTD_isError");
#undef GBM_SYM
		ok = createCCtx && freeCCtx && setParameter && compress2 && compressBound && decompress &&
		     getFrameContentSize && isError;
	}
	// false on any error: the caller then stores the block Plain (block.rs:88-93)
	bool encode(const uint8_t *data, size_t len, int level, std::vector<uint8_t> &out) const
	{
		if (!ok)
			return false;
		void *c = createCCtx();
		if (!c)
			return false;
		bool good = !isError(setParameter(c, 100 /* ZSTD_c_compressionLevel */, level)) &&
			    !isError(setParameter(c, 201 /* ZSTD_c_checksumFlag */, 1));
		if (good) {
			out.resize(compressBound(len));
			size_t n = compress2(c, out.data(), out.size(), data, len);
			good = !isError(n);
			if (good)
				out.resize(n);
		}
		freeCCtx(c);
		return good;
	}
	// verifies the frame checksum; false = corrupt.  `max_out` bounds the allocation: a damaged or
	// forged frame header must not be able to ask for terabytes (Garage blocks are <= block_size, and a
	// zstd frame cannot expand by more than ~2^17 per byte; the caller passes a generous multiple of
	// block_size).  Frames without a content-size field (streaming encoders write those) are decoded
	// into a buffer that grows up to the same bound.
	bool decode(const uint8_t *data, size_t len, size_t max_out, std::vector<uint8_t> &out) const
	{
		if (!ok)
			return false;
		const unsigned long long sz = getFrameContentSize(data, len);
		const unsigned long long UNKNOWN = 0ULL - 1, ERROR_ = 0ULL - 2;
		if (sz == ERROR_)
			return false;
		try {
			if (sz != UNKNOWN) {
				if (sz > max_out)
					return false;
				out.resize((size_t)sz);
				uint8_t dummy;
				size_t n = decompress(sz ? out.data() : &dummy, (size_t)sz, data, len);
				return !isError(n) && n == sz;
			}
			size_t cap = std::min<size_t>(std::max<size_t>(4 * len, 1 << 16), max_out);
			for (;;) {
				out.resize(cap);
				size_t n = decompress(out.data(), cap, data, len);
				if (!isError(n)) {
					out.resize(n);
					return true;
				}
				if (cap >= max_out)
					return false;  // corrupt, or larger than any block can be
				cap = std::min(cap * 4, max_out);
			}
		} catch (const std::bad_alloc &) {
			return false;
		}
	}
};

const Zstd &zstd()
{
	static const Zstd z;
	return z;
}

using Hash = std::string;  // 32 raw bytes

std::string hex(const Hash &h)
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
// Same 64-byte layout as garage_amd/block_manager.py::ShardHeader ("<4sBBBBB3xQII32s").
struct ShardHeader {
	uint8_t k = 0, m = 0, idx = 0, compressed = 0;
	uint64_t orig_len = 0;
	uint32_t shard_len = 0;
	uint8_t checksum[32] = {0};

	void pack(uint8_t out[GBM_SHARD_HEADER_SIZE]) const
	{
		std::memset(out, 0, GBM_SHARD_HEADER_SIZE);
		std::memcpy(out, "GECS", 4);
		out[4] = 2;  // version 2: the checksum is the tree-mode shardsum (version 1 was plain blake2sum)
		out[5] = k;
		out[6] = m;
		out[7] = idx;
		out[8] = compressed;
		std::memcpy(out + 12, &orig_len, 8);
		std::memcpy(out + 20, &shard_len, 4);
		std::memcpy(out + 28, checksum, 32);
	}
	bool unpack(const uint8_t *in, size_t n)
	{
		if (n < GBM_SHARD_HEADER_SIZE || std::memcmp(in, "GECS", 4) != 0 || in[4] != 2)
			return false;
		k = in[5];
		m = in[6];
		idx = in[7];
		compressed = in[8];
		std::memcpy(&orig_len, in + 12, 8);
		std::memcpy(&shard_len, in + 20, 4);
		std::memcpy(checksum, in + 28, 32);
		return true;
	}
};

uint64_t real_now_ms()
{
	return (uint64_t)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::system_clock::now().time_since_epoch()).count();
}

// ------------------------------------------------------------------ thread pool
// fork-join: fn(i) for i in [0, n) on the workers and the calling thread.  Several callers may use it at
// once (they queue on call_mu_); work items must not call parallel_for themselves.
class Pool {
public:
	explicit Pool(unsigned n) { resize(n); }
	~Pool() { stop_all(); }
	void resize(unsigned n)
	{
		std::lock_guard<std::mutex> call(call_mu_);
		stop_all();
		stop_ = false;
		for (unsigned i = 0; i < n; ++i)
			workers_.emplace_back([this] { run(); });
	}
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
			++epoch_;
		}
		cv_.notify_all();
		work();
		std::unique_lock<std::mutex> g(mu_);
		done_cv_.wait(g, [this] { return pending_ == 0; });
		fn_ = nullptr;
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

// Fire-and-forget tasks for requests that may be abandoned (hedged reads): a task owns everything it touches
// through shared_ptrs, so nobody has to wait for a slow one.
class Async {
public:
	explicit Async(unsigned n)
	{
		for (unsigned i = 0; i < n; ++i)
			workers_.emplace_back([this] { run(); });
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
			fn();
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
// by size.  Without a device (CPU tests against the oracle stub) gec_host_alloc is plain malloc.
class BufPool : public std::enable_shared_from_this<BufPool> {
public:
	static constexpr size_t kRetainMax = 2ull << 30;  // bytes kept for reuse; beyond that buffers are freed
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
		if (!raw)
			raw = static_cast<uint8_t *>(gec_host_alloc(cap));
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
struct Shard {
	ShardHeader hd;
	Bytes data;  // shard_len bytes
};

enum class RpcKind { PutShard, GetShard, NeedShardQuery, DeleteShard };

struct ShardRpc {
	RpcKind kind;
	const Hash *hash;
	int idx;
	Shard shard;                        // PutShard
	const gbm_order_tag *tag = nullptr; // PutShard / GetShard
};

struct ShardResp {
	bool ok = false;     // PutShard stored / GetShard found / DeleteShard had something to delete
	bool needed = false; // NeedShardReply
	Shard shard;         // answer to GetShard
};

struct Node {
	std::atomic<bool> down{false};  // flipped by gbm_node_set_down while other threads are talking to the node
	std::atomic<uint64_t> order_violations{0};
	std::atomic<uint64_t> latency_us{0};  // test hook: every request to this node takes this long
	std::shared_ptr<BufPool> bufs;
	virtual ~Node() = default;
	virtual bool put(const Hash &h, int idx, const Shard &s) = 0;
	virtual bool get(const Hash &h, int idx, Shard &s) = 0;  // false: absent / unreadable
	virtual bool has(const Hash &h, int idx) = 0;
	virtual bool del(const Hash &h, int idx) = 0;
	virtual void mark_corrupted(const Hash &h, int idx) { del(h, idx); }
	// every hash this node holds a shard of (BlockStoreIterator, src/block/repair.rs:196-233,634-752)
	virtual void list(std::set<Hash> &out) = 0;

	// the node's endpoint (StreamingEndpointHandler<BlockRpc>::handle, src/block/manager.rs:692-707);
	// false = could not be contacted
	bool handle(const ShardRpc &rq, ShardResp &rs)
	{
		if (down.load(std::memory_order_acquire))
			return false;
		if (const uint64_t us = latency_us.load(std::memory_order_relaxed))
			std::this_thread::sleep_for(std::chrono::microseconds(us));
		switch (rq.kind) {
		case RpcKind::PutShard:
			note_order(rq.tag);
			rs.ok = put(*rq.hash, rq.idx, rq.shard);
			return true;
		case RpcKind::GetShard:
			rs.ok = get(*rq.hash, rq.idx, rs.shard);
			return true;
		case RpcKind::NeedShardQuery:
			rs.ok = true;
			rs.needed = !has(*rq.hash, rq.idx);
			return true;
		case RpcKind::DeleteShard:
			rs.ok = del(*rq.hash, rq.idx);
			return true;
		}
		return false;
	}

private:
	void note_order(const gbm_order_tag *tag)
	{
		if (!tag)
			return;
		std::lock_guard<std::mutex> g(order_mu_);
		auto it = last_order_.find(tag->stream_id);
		if (it != last_order_.end() && tag->order < it->second)
			order_violations.fetch_add(1);
		if (it == last_order_.end() || tag->order > it->second)
			last_order_[tag->stream_id] = tag->order;
		if (last_order_.size() > 4096)  // streams are short-lived (one per PutObject / GetObject)
			last_order_.erase(last_order_.begin());
	}
	std::mutex order_mu_;
	std::map<uint64_t, uint64_t> last_order_;
};

std::string shard_key(const Hash &h, int idx)
{
	std::string k(h);
	k.push_back((char)idx);
	return k;
}

struct MemoryNode : Node {
	static constexpr int kStripes = 64;  // puts and gets of different hashes do not contend
	struct Stripe {
		std::mutex mu;
		std::unordered_map<std::string, Shard> files;
	};
	Stripe stripes[kStripes];
	Stripe &stripe_of(const Hash &h) { return stripes[((unsigned char)h[2] ^ (unsigned char)h[3]) % kStripes]; }
	bool put(const Hash &h, int idx, const Shard &s) override
	{
		Stripe &st = stripe_of(h);
		std::lock_guard<std::mutex> g(st.mu);
		st.files[shard_key(h, idx)] = s;
		return true;
	}
	bool get(const Hash &h, int idx, Shard &s) override
	{
		Stripe &st = stripe_of(h);
		std::lock_guard<std::mutex> g(st.mu);
		auto it = st.files.find(shard_key(h, idx));
		if (it == st.files.end())
			return false;
		s = it->second;
		return true;
	}
	bool has(const Hash &h, int idx) override
	{
		Stripe &st = stripe_of(h);
		std::lock_guard<std::mutex> g(st.mu);
		return st.files.count(shard_key(h, idx)) != 0;
	}
	bool del(const Hash &h, int idx) override
	{
		Stripe &st = stripe_of(h);
		std::lock_guard<std::mutex> g(st.mu);
		return st.files.erase(shard_key(h, idx)) != 0;
	}
	void list(std::set<Hash> &out) override
	{
		for (Stripe &st : stripes) {
			std::lock_guard<std::mutex> g(st.mu);
			for (auto &kv : st.files)
				out.insert(kv.first.substr(0, 32));
		}
	}
};

// <root>/<h0>/<h1>/<hex>.s<idx>, tmp file + rename (write_block_inner, manager.rs:720-805);
// a corrupt shard is renamed *.corrupted (manager.rs:807-819).  File = 64-byte header + payload.
struct DirNode : Node {
	std::string root;
	std::atomic<bool> fsync_data{false};  // Config.data_fsync (src/util/config.rs:22-24), off by default
	explicit DirNode(std::string r) : root(std::move(r)) {}
	std::string dir(const Hash &h) const
	{
		std::string hx = hex(h);
		return root + "/" + hx.substr(0, 2) + "/" + hx.substr(2, 2);
	}
	std::string path(const Hash &h, int idx) const { return dir(h) + "/" + hex(h) + ".s" + std::to_string(idx); }
	static void mkdirs(const std::string &p)
	{
		for (size_t i = 1; i <= p.size(); ++i)
			if (i == p.size() || p[i] == '/')
				::mkdir(p.substr(0, i).c_str(), 0755);
	}
	// raw descriptors, one writev / two preads per shard: a shard file is written and read whole, stdio's buffer
	// would only add a copy (7168 files per 512-block batch: the syscall count is what the node's rate is made of)
	bool put(const Hash &h, int idx, const Shard &s) override
	{
		static std::atomic<uint64_t> seq{0};  // unique per writer: two threads may store the same shard
		const std::string d = dir(h);
		std::string p = path(h, idx), tmp = p + ".tmp" + std::to_string(::getpid()) + "_" + std::to_string(seq++);
		int fd = ::open(tmp.c_str(), O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0644);
		if (fd < 0 && errno == ENOENT) {  // first shard of this prefix
			mkdirs(d);
			fd = ::open(tmp.c_str(), O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0644);
		}
		if (fd < 0)
			return false;
		uint8_t hdr[GBM_SHARD_HEADER_SIZE];
		s.hd.pack(hdr);
		struct iovec iov[2] = {{hdr, sizeof(hdr)}, {const_cast<uint8_t *>(s.data.data()), s.data.n}};
		size_t left = sizeof(hdr) + s.data.n;
		bool ok = true;
		int cur = 0;
		while (left && ok) {  // (short writes: continue where the kernel stopped)
			const ssize_t w = ::writev(fd, iov + cur, 2 - cur);
			if (w < 0) {
				ok = errno == EINTR;
				continue;
			}
			left -= (size_t)w;
			size_t adv = (size_t)w;
			while (cur < 2 && adv >= iov[cur].iov_len)
				adv -= iov[cur++].iov_len;
			if (cur < 2) {
				iov[cur].iov_base = (uint8_t *)iov[cur].iov_base + adv;
				iov[cur].iov_len -= adv;
			}
		}
		const bool sync = fsync_data.load();
		if (ok && sync)  // file first, then (after the rename) its directory: manager.rs:775-800
			ok = ::fsync(fd) == 0;
		ok = (::close(fd) == 0) && ok;
		if (ok)
			ok = std::rename(tmp.c_str(), p.c_str()) == 0;
		if (!ok)
			std::remove(tmp.c_str());
		if (ok && sync) {
			int dfd = ::open(d.c_str(), O_RDONLY | O_DIRECTORY);
			if (dfd >= 0) {
				ok = ::fsync(dfd) == 0;
				::close(dfd);
			} else {
				ok = false;
			}
		}
		return ok;
	}
	static bool read_all(int fd, uint8_t *dst, size_t len, off_t off)
	{
		while (len) {
			const ssize_t r = ::pread(fd, dst, len, off);
			if (r < 0 && errno == EINTR)
				continue;
			if (r <= 0)
				return false;
			dst += r;
			len -= (size_t)r;
			off += r;
		}
		return true;
	}
	bool get(const Hash &h, int idx, Shard &s) override
	{
		const int fd = ::open(path(h, idx).c_str(), O_RDONLY | O_CLOEXEC);
		if (fd < 0)
			return false;
		struct stat stt;
		uint8_t hdr[GBM_SHARD_HEADER_SIZE];
		bool ok = ::fstat(fd, &stt) == 0 && stt.st_size >= (off_t)sizeof(hdr) && read_all(fd, hdr, sizeof(hdr), 0);
		// a file whose header does not parse is handed up as an invalid shard (shard_len != size) so that
		// the reader treats it like a checksum failure: *.corrupted + resync
		if (ok && !s.hd.unpack(hdr, sizeof(hdr))) {
			s.hd = ShardHeader();
			s.hd.idx = 0xff;
		}
		if (ok) {
			const size_t len = (size_t)stt.st_size - sizeof(hdr);
			s.data = bufs->get(len);
			ok = read_all(fd, s.data.mut(), len, sizeof(hdr));
		}
		::close(fd);
		return ok;
	}
	bool has(const Hash &h, int idx) override
	{
		struct stat st;
		return ::stat(path(h, idx).c_str(), &st) == 0;
	}
	bool del(const Hash &h, int idx) override { return std::remove(path(h, idx).c_str()) == 0; }
	void mark_corrupted(const Hash &h, int idx) override
	{
		std::string p = path(h, idx);
		std::rename(p.c_str(), (p + ".corrupted").c_str());
	}
	void list(std::set<Hash> &out) override
	{
		// <root>/<h0>/<h1>/<64 hex digits>.s<idx>
		auto each = [](const std::string &d, const std::function<void(const std::string &)> &fn) {
			if (DIR *dp = ::opendir(d.c_str())) {
				while (struct dirent *e = ::readdir(dp))
					if (e->d_name[0] != '.')
						fn(e->d_name);
				::closedir(dp);
			}
		};
		each(root, [&](const std::string &a) {
			each(root + "/" + a, [&](const std::string &b) {
				each(root + "/" + a + "/" + b, [&](const std::string &f) {
					const size_t dot = f.find(".s");
					if (dot != 64 || f.find_first_not_of("0123456789", dot + 2) != std::string::npos || f.size() == dot + 2)
						return;
					Hash h(32, 0);
					for (int i = 0; i < 32; ++i) {
						auto nib = [](char c) { return c >= 'a' ? c - 'a' + 10 : c - '0'; };
						h[i] = (char)((nib(f[2 * i]) << 4) | nib(f[2 * i + 1]));
					}
					out.insert(h);
				});
			});
		});
	}
};

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

}  // namespace

struct gbm_manager {
	const gec_codec *codec = nullptr;
	int k = 0, m = 0, n = 0, write_quorum = 0;
	std::vector<std::unique_ptr<Node>> nodes;
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

	// resync.queue / resync.errors (resync.rs:170-253)
	mutable std::mutex rs_mu;
	std::condition_variable rs_cv;
	std::set<std::pair<uint64_t, Hash>> rs_queue;
	std::unordered_map<Hash, ErrorCounter> rs_errors;
	std::thread rs_worker;
	bool rs_worker_stop = false;

	std::atomic<uint64_t> gc_delay_ms{GBM_BLOCK_GC_DELAY_MS}, retry_delay_ms{GBM_RESYNC_RETRY_DELAY_MS},
		incref_delay_ms{2 * 300000ull};  // 2 * rpc_timeout, DEFAULT_TIMEOUT = 300 s (rpc_helper.rs:33)
	std::atomic<uint64_t> clock_skew_ms{0};
	uint64_t now() const { return real_now_ms() + clock_skew_ms.load(); }

	// ScrubWorkerPersisted (src/block/repair.rs:169-194)
	std::atomic<uint64_t> scrub_corruptions{0}, scrub_last_complete_ms{0};
	std::atomic<uint64_t> metrics[6] = {};
	std::atomic<uint64_t> gpu_hashed{0};
	std::atomic<bool> compress{false};    // Config.compression_level (src/util/config.rs:52-58); Garage's default is Some(1)
	std::atomic<int> compression_level{1};
	std::atomic<bool> verify_block_hash{true};
	std::atomic<size_t> cpu_block_hash_max{96};  // gets of up to this many blocks hash them on the host (gbm_set_threads rescales)

	// hedged reads: 0 = the k requests of a read are issued and awaited in order
	std::atomic<uint64_t> hedge_us{0}, hedged_reads{0};
	std::mutex async_mu;
	std::shared_ptr<Async> async;  // created when hedging is first switched on
	std::shared_ptr<Async> async_pool()
	{
		std::lock_guard<std::mutex> g(async_mu);
		if (!async)
			async = std::make_shared<Async>(32);
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

namespace {

// GBM_TRACE=1: stage timings of the batched put / get on stderr (tools/host_path_bench.py reads them off)
bool trace_on()
{
	static const bool on = [] {
		const char *e = std::getenv("GBM_TRACE");
		return e && e[0] == '1';
	}();
	return on;
}
struct Trace {
	const char *what;
	std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
	std::string line;
	explicit Trace(const char *w) : what(w) {}
	void lap(const char *stage)
	{
		if (!trace_on())
			return;
		const auto t = std::chrono::steady_clock::now();
		char buf[64];
		std::snprintf(buf, sizeof buf, " %s %.2f ms", stage, std::chrono::duration<double, std::milli>(t - t0).count());
		line += buf;
		t0 = t;
	}
	~Trace()
	{
		if (trace_on() && !line.empty())
			std::fprintf(stderr, "[gbm] %s:%s\n", what, line.c_str());
	}
};

int ec_fail(int rc, const char *what)
{
	return fail(GBM_E_EC, std::string(what) + ": " + gec_strerror(rc) + " (" + gec_last_error() + ")");
}

// largest block a Garage node will ever hold decompresses to: block_size is configurable, 1 MiB by default and
// "a few MiB" in practice; 1 GiB is far above any of it and still a harmless allocation bound
constexpr size_t kMaxDecompressed = 1ull << 30;

// shard checksums of many buffers: on the GPU (gec_shardsum_batch) once the batch is big enough to beat the
// CPU pool through PCIe, else on the pool's threads.  SURVEY.md section 8 row f4.
constexpr size_t kGpuHashMinMessages = 64;
constexpr size_t kGpuHashMinBytes = 8u << 20;

int hash_many(gbm_manager *mg, const std::vector<const uint8_t *> &ptrs, const std::vector<size_t> &lens,
	      std::vector<uint8_t> &sums)
{
	sums.resize(ptrs.size() * 32);
	size_t total = 0;
	for (size_t l : lens)
		total += l;
	if (ptrs.size() >= kGpuHashMinMessages && total >= kGpuHashMinBytes) {
		int rc = gec_shardsum_batch(mg->codec, ptrs.size(), ptrs.data(), lens.data(), sums.data());
		if (rc)
			return ec_fail(rc, "gec_shardsum_batch");
		mg->gpu_hashed += ptrs.size();
		return GBM_OK;
	}
	mg->pool->parallel_for(ptrs.size(), [&](size_t i) { shardsum(ptrs[i], lens[i], sums.data() + 32 * i); });
	return GBM_OK;
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
	size_t next = 0;  // next candidate (version-major, shard index minor) to try
	bool mixed = false;
	bool settled = false;  // a geometry has been chosen; later candidates must match it
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

// Fetch shards until every block has `want` valid ones of one geometry in hand (or ran out of nodes): shard
// index order within the current layout version, then older versions (block_read_nodes_of interleaves
// versions the same way, rpc_helper.rs:570-619).  The checksums of each round's candidates are verified in
// ONE batch; a shard whose checksum or header does not match is treated as missing, renamed *.corrupted and
// queued for resync (read_block_from's behaviour, manager.rs:577-609), and the next node is tried in the
// following round.
int gather_many(gbm_manager *mg, const std::vector<Hash> &hs, const gbm_order_tag *tags, int want, std::vector<Gathered> &gs,
		bool verify = true, const std::vector<uint8_t> *only = nullptr)
{
	// verify == false: shards are accepted on their header alone; the caller checks the checksums in the same
	// device trip that decodes (gec_decode_verify_batch) and comes back for more (`only` = blocks to continue)
	const int n = mg->n;
	const int vcur = mg->layout_cur.load(), vold = mg->layout_oldest.load();
	const size_t ncand = (size_t)(vcur - vold + 1) * n;
	if (!only)
		gs.assign(hs.size(), Gathered());
	struct Cand {
		size_t b;
		int j, node;
		Shard s;
	};
	auto have = [&](const Gathered &g, int j) { return g.settled ? !g.shard[j].empty() : g.have_idx(j); };
	auto in_hand = [&](const Gathered &g) { return g.settled ? g.count : g.best(); };
	// header checks every fetched shard passes before it becomes a candidate; called by one thread per block
	auto accept = [&](std::vector<Cand> &mine, size_t b, int j, int node, Shard &&sh) -> bool {
		Gathered &g = gs[b];
		const ShardHeader &hd = sh.hd;
		const bool ok = hd.idx == j && hd.k == mg->k && hd.m == mg->m && sh.data.n == hd.shard_len && hd.shard_len > 0 &&
				hd.shard_len % 64 == 0;
		if (!ok) {
			mg->metrics[2]++;
			mg->nodes[node]->mark_corrupted(hs[b], j);
			mg->put_to_resync(hs[b], 0);
			return false;
		}
		if (g.settled &&
		    (hd.compressed != g.meta.compressed || hd.orig_len != g.meta.orig_len || hd.shard_len != g.meta.shard_len)) {
			g.mixed = true;  // a stale shard of another geometry: resync will overwrite it
			return false;
		}
		mine.push_back(Cand{b, j, node, std::move(sh)});
		return true;
	};
	// next (version, shard index) candidate of block b that is not in hand and not already asked for this round
	// (`taken(j)`: shard j is already covered this round; a j whose request failed is asked again from the holder
	// in the next older layout version)
	auto next_candidate = [&](size_t b, const std::function<bool(int)> &taken, std::vector<int> &who, int &who_v,
				  int &j_out) -> bool {
		Gathered &g = gs[b];
		while (g.next < ncand) {
			const size_t c = g.next++;
			const int v = vcur - (int)(c / n), j = (int)(c % n);
			if (have(g, j) || taken(j))
				continue;
			if (v != who_v) {
				mg->nodes_of(hs[b], v, who);
				who_v = v;
			}
			j_out = j;
			return true;
		}
		return false;
	};
	const uint64_t hedge_us = mg->hedge_us.load();
	for (;;) {
		std::vector<std::vector<Cand>> per(hs.size());
		if (hedge_us == 0) {
			mg->pool->parallel_for(hs.size(), [&](size_t b) {
				if (only && !(*only)[b])
					return;
				Gathered &g = gs[b];
				int pending = 0, who_v = -1, j = 0;
				std::vector<int> who;
				auto taken = [&](int jj) {
					for (const Cand &pc : per[b])
						if (pc.j == jj)
							return true;
					return false;
				};
				while (in_hand(g) + pending < want && next_candidate(b, taken, who, who_v, j)) {
					ShardRpc rq{RpcKind::GetShard, &hs[b], j, Shard(), tags ? &tags[b] : nullptr};
					ShardResp rs;
					if (!mg->nodes[who[j]]->handle(rq, rs) || !rs.ok)
						continue;
					if (accept(per[b], b, j, who[j], std::move(rs.shard)))
						++pending;
				}
			});
		} else {
			// Hedged round: every request of the round is in flight at once; when some have not answered
			// after hedge_us, the next candidates (the parity holders, then older layout versions) are
			// asked as well, and a block moves on as soon as it has its shards from whoever answered
			// first.  Requests that lose the race are abandoned, not cancelled: they own their state.
			struct Flight {
				size_t b;
				int j, node;
				Hash h;
				gbm_order_tag tag;
				bool has_tag, answered = false, done = false;
				ShardResp rs;
			};
			struct Round {
				std::mutex mu;
				std::condition_variable cv;
				std::vector<int> need, ok, outstanding;
				size_t unsatisfied = 0;
				std::atomic<bool> over{false};  // the round has what it needs: requests not yet started are dropped
				bool satisfied(size_t b) const { return ok[b] >= need[b] || outstanding[b] == 0; }
			};
			auto rd = std::make_shared<Round>();
			rd->need.assign(hs.size(), 0);
			rd->ok.assign(hs.size(), 0);
			rd->outstanding.assign(hs.size(), 0);
			std::vector<std::shared_ptr<Flight>> flights;
			std::vector<std::vector<size_t>> flights_of(hs.size());
			std::vector<std::vector<int>> who(hs.size());
			std::vector<int> who_v(hs.size(), -1);
			std::shared_ptr<Async> async = mg->async_pool();
			// caller holds rd->mu
			auto launch = [&](size_t b, int count) -> int {
				int launched = 0, j = 0;
				auto taken = [&](int jj) {  // in flight, or answered with a shard
					for (size_t fi : flights_of[b]) {
						const Flight &f = *flights[fi];
						if (f.j == jj && (!f.done || (f.answered && f.rs.ok)))
							return true;
					}
					return false;
				};
				while (launched < count && next_candidate(b, taken, who[b], who_v[b], j)) {
					flights_of[b].push_back(flights.size());
					auto f = std::make_shared<Flight>();
					f->b = b;
					f->j = j;
					f->node = who[b][j];
					f->h = hs[b];
					f->has_tag = tags != nullptr;
					if (tags)
						f->tag = tags[b];
					flights.push_back(f);
					const bool was = rd->satisfied(b);
					rd->outstanding[b]++;
					if (was && !rd->satisfied(b))
						rd->unsatisfied++;
					Node *nd = mg->nodes[f->node].get();
					async->submit([rd, f, nd] {
						ShardRpc rq{RpcKind::GetShard, &f->h, f->j, Shard(), f->has_tag ? &f->tag : nullptr};
						ShardResp rs;
						const bool answered = !rd->over.load() && nd->handle(rq, rs);
						{
							std::lock_guard<std::mutex> g(rd->mu);
							f->rs = std::move(rs);
							f->answered = answered;
							f->done = true;
							const bool was_sat = rd->satisfied(f->b);
							rd->outstanding[f->b]--;
							if (answered && f->rs.ok)
								rd->ok[f->b]++;
							if (!was_sat && rd->satisfied(f->b))
								rd->unsatisfied--;
						}
						rd->cv.notify_all();
					});
					++launched;
				}
				return launched;
			};
			std::unique_lock<std::mutex> lk(rd->mu);
			for (size_t b = 0; b < hs.size(); ++b) {
				if (only && !(*only)[b])
					continue;
				rd->need[b] = std::max(0, want - in_hand(gs[b]));
				launch(b, rd->need[b]);
			}
			// (system_clock: pthread_cond_timedwait, which ThreadSanitizer understands; gcc 11's does not know
			// the pthread_cond_clockwait a steady_clock deadline turns into)
			const auto deadline = std::chrono::system_clock::now() + std::chrono::microseconds(hedge_us);
			if (!rd->cv.wait_until(lk, deadline, [&] { return rd->unsatisfied == 0; })) {
				uint64_t hedges = 0;
				for (size_t b = 0; b < hs.size(); ++b)
					if (!rd->satisfied(b))
						hedges += launch(b, rd->need[b] - rd->ok[b]);
				mg->hedged_reads += hedges;
				rd->cv.wait(lk, [&] { return rd->unsatisfied == 0; });
			}
			rd->over = true;
			for (auto &f : flights)
				if (f->done && f->answered && f->rs.ok)
					accept(per[f->b], f->b, f->j, f->node, std::move(f->rs.shard));
		}
		std::vector<Cand *> cands;
		for (auto &v : per)
			for (Cand &c : v)
				cands.push_back(&c);
		if (cands.empty())
			break;
		std::vector<uint8_t> sums;
		if (verify) {
			std::vector<const uint8_t *> ptrs(cands.size());
			std::vector<size_t> lens(cands.size());
			for (size_t i = 0; i < cands.size(); ++i) {
				ptrs[i] = cands[i]->s.data.data();
				lens[i] = cands[i]->s.hd.shard_len;
			}
			int rc = hash_many(mg, ptrs, lens, sums);
			if (rc)
				return rc;
		}
		for (size_t i = 0; i < cands.size(); ++i) {
			Cand &c = *cands[i];
			Gathered &g = gs[c.b];
			if (verify && std::memcmp(sums.data() + 32 * i, c.s.hd.checksum, 32) != 0) {
				mg->metrics[2]++;
				mg->nodes[c.node]->mark_corrupted(hs[c.b], c.j);
				mg->put_to_resync(hs[c.b], 0);
				continue;
			}
			mg->metrics[1] += c.s.hd.shard_len;
			std::array<uint8_t, 32> want_sum;
			std::memcpy(want_sum.data(), c.s.hd.checksum, 32);
			if (g.settled) {
				g.shard[c.j] = std::move(c.s.data);
				g.sum[c.j] = want_sum;
				g.node[c.j] = c.node;
				g.count++;
				continue;
			}
			Geometry geo;
			geo.compressed = c.s.hd.compressed;
			geo.orig_len = c.s.hd.orig_len;
			geo.shard_len = c.s.hd.shard_len;
			Gathered::Group &grp = g.groups[geo];
			if (grp.shard.empty()) {
				grp.shard.assign(n, Bytes());
				grp.sum.assign(n, {});
				grp.node.assign(n, -1);
				grp.meta = c.s.hd;
			}
			grp.shard[c.j] = std::move(c.s.data);
			grp.sum[c.j] = want_sum;
			grp.node[c.j] = c.node;
			grp.count++;
		}
	}
	// settle on the largest consistent group; the stragglers of other geometries are
	// stale leftovers that resync will overwrite
	for (size_t b = 0; b < hs.size(); ++b) {
		Gathered &g = gs[b];
		if (g.settled || (only && !(*only)[b]))
			continue;
		Gathered::Group *bestg = nullptr;
		for (auto &kv : g.groups)
			if (!bestg || kv.second.count > bestg->count)
				bestg = &kv.second;
		if (bestg) {
			g.shard = std::move(bestg->shard);
			g.sum = std::move(bestg->sum);
			g.node = std::move(bestg->node);
			g.meta = bestg->meta;
			g.have_meta = true;
			g.count = bestg->count;
			g.mixed = g.groups.size() > 1;
		} else {
			g.shard.assign(n, Bytes());
			g.sum.assign(n, {});
			g.node.assign(n, -1);
		}
		g.settled = true;
		g.groups.clear();
	}
	for (size_t b = 0; b < hs.size(); ++b)
		if (gs[b].mixed && (!only || (*only)[b]))
			mg->put_to_resync(hs[b], 0);
	return GBM_OK;
}

// PutShard to one node; false = the node could not be contacted or refused
bool send_shard(gbm_manager *mg, int node, const Hash &h, int idx, const Bytes &payload, size_t S, uint64_t orig_len,
		bool compressed, const uint8_t *checksum, const gbm_order_tag *tag)
{
	ShardRpc rq{RpcKind::PutShard, &h, idx, Shard(), tag};
	ShardHeader &hd = rq.shard.hd;
	hd.k = (uint8_t)mg->k;
	hd.m = (uint8_t)mg->m;
	hd.idx = (uint8_t)idx;
	hd.compressed = compressed ? 1 : 0;
	hd.orig_len = orig_len;
	hd.shard_len = (uint32_t)S;
	if (checksum)
		std::memcpy(hd.checksum, checksum, 32);
	else
		shardsum(payload.data(), S, hd.checksum);
	rq.shard.data = payload;
	ShardResp rs;
	return mg->nodes[node]->handle(rq, rs) && rs.ok;
}

// rcs (optional): per-block result, GBM_OK or GBM_E_QUORUM; the return value is the last failure.  The device
// work of the whole batch happens before anything is sent to a node, so a device error (GBM_E_EC) fails
// every block of the batch and leaves no partial state behind.
int put_blocks_impl(gbm_manager *mg, size_t nb, const uint8_t *hashes, const uint8_t *const *data, const size_t *len,
		    const uint8_t *prevent_compression, const gbm_order_tag *tags, int *rcs)
{
	if (!mg || (nb && (!hashes || !data || !len)))
		return fail(GBM_E_INVALID_ARG, "NULL argument");
	if (nb == 0)
		return GBM_OK;
	for (size_t b = 0; b < nb; ++b)
		if (!data[b] && len[b])
			return fail(GBM_E_INVALID_ARG, "NULL block pointer");
	if (rcs)
		std::fill(rcs, rcs + nb, GBM_OK);
	const int k = mg->k, m = mg->m, n = mg->n;
	const bool compress = mg->compress.load();
	const int level = mg->compression_level.load();
	// -- DataBlock::from_buffer (zstd when a level is configured and the caller did not forbid it, Plain on any
	//    encoder error), then ONE copy of the payload into a zero-padded k*S buffer whose slices are the k data
	//    shards.  Shard geometry is a pure function of the block: S = gec_shard_len(k, payload length) --
	//    never the batch maximum: a later put of the same block must produce compatible shards.
	struct Prep {
		Bytes block, parity;
		size_t plen = 0, S = 0;
		bool z = false;
	};
	std::vector<Prep> prep(nb);
	std::atomic<bool> oom{false};
	Trace tr("put");
	mg->pool->parallel_for(nb, [&](size_t b) {
		try {
			Prep &p = prep[b];
			std::vector<uint8_t> zbuf;
			const uint8_t *src = data[b];
			p.plen = len[b];
			if (compress && !(prevent_compression && prevent_compression[b]) &&
			    zstd().encode(data[b], len[b], level, zbuf)) {
				src = zbuf.data();
				p.plen = zbuf.size();
				p.z = true;
			}
			p.S = gec_shard_len(k, p.plen);
			p.block = mg->bufs->get((size_t)k * p.S);
			if (p.plen)
				std::memcpy(p.block.mut(), src, p.plen);
			std::memset(p.block.mut() + p.plen, 0, (size_t)k * p.S - p.plen);
			p.parity = mg->bufs->get((size_t)m * p.S);
		} catch (const std::bad_alloc &) {
			oom = true;
		}
	});
	if (oom)
		return fail(GBM_E_IO, "out of (pinned) host memory for the shard buffers");
	tr.lap("prep");
	// Blocks of equal S -- in practice all full block_size blocks -- share ONE device call that returns
	// parity and the checksums of all k+m shards.
	std::map<size_t, std::vector<size_t>> by_s;
	for (size_t b = 0; b < nb; ++b)
		by_s[prep[b].S].push_back(b);
	std::vector<uint8_t> sums(nb * (size_t)n * 32);
	for (auto &kv : by_s) {
		const size_t S = kv.first;
		const std::vector<size_t> &ids = kv.second;
		const size_t gn = ids.size();
		std::vector<uint8_t *> pp(gn);
		std::vector<const uint8_t *> gd(gn);
		std::vector<size_t> gl(gn, (size_t)k * S);  // the buffers are already padded: whole data area
		std::vector<uint8_t> gsums(gn * (size_t)n * 32);
		for (size_t i = 0; i < gn; ++i) {
			pp[i] = prep[ids[i]].parity.mut();
			gd[i] = prep[ids[i]].block.data();
		}
		int rc = gec_encode_hash_batch(mg->codec, gn, gd.data(), gl.data(), S, pp.data(), gsums.data());
		if (rc) {  // nothing has been sent to any node yet: the whole batch fails
			if (rcs)
				std::fill(rcs, rcs + nb, GBM_E_EC);
			return ec_fail(rc, "gec_encode_hash_batch");
		}
		mg->gpu_hashed += gn * (size_t)n;
		for (size_t i = 0; i < gn; ++i)
			std::memcpy(sums.data() + ids[i] * (size_t)n * 32, gsums.data() + i * (size_t)n * 32, (size_t)n * 32);
	}
	tr.lap("encode+hash");
	// fan-out: shard j of every block to nodes_of(hash)[j].  With order tags the blocks go out one after the
	// other in (stream, order) order -- requests of one stream reach a node in `order` order, whatever their
	// shard geometry; without tags the blocks are independent and go out from the pool's threads.
	std::vector<int> oks(nb, 0);
	auto fan_out = [&](size_t b) {
		Hash h((const char *)hashes + 32 * b, 32);
		std::vector<int> who;
		mg->nodes_of(h, who);
		const size_t S = prep[b].S;
		int ok = 0;
		for (int j = 0; j < n; ++j) {
			const Bytes payload = j < k ? prep[b].block.slice((size_t)j * S, S) : prep[b].parity.slice((size_t)(j - k) * S, S);
			if (send_shard(mg, who[j], h, j, payload, S, prep[b].plen, prep[b].z, sums.data() + (b * n + j) * 32,
				       tags ? &tags[b] : nullptr)) {
				++ok;
				mg->metrics[0] += S;
			}
		}
		oks[b] = ok;
	};
	if (tags) {
		// order is a per-node property (requests of one stream reach a NODE in `order` order): the nodes are served
		// side by side, each one walking the blocks in (stream, order) order and taking the shards that are its own
		std::vector<size_t> order(nb);
		for (size_t i = 0; i < nb; ++i)
			order[i] = i;
		std::stable_sort(order.begin(), order.end(), [&](size_t x, size_t y) {
			return std::tie(tags[x].stream_id, tags[x].order) < std::tie(tags[y].stream_id, tags[y].order);
		});
		std::vector<std::vector<int>> who(nb);
		for (size_t b = 0; b < nb; ++b)
			mg->nodes_of(Hash((const char *)hashes + 32 * b, 32), who[b]);
		std::vector<std::atomic<int>> okc(nb);
		for (auto &x : okc)
			x = 0;
		mg->pool->parallel_for(mg->nodes.size(), [&](size_t node) {
			for (size_t b : order) {
				const size_t S = prep[b].S;
				for (int j = 0; j < n; ++j) {
					if (who[b][j] != (int)node)
						continue;
					const Hash h((const char *)hashes + 32 * b, 32);
					const Bytes payload = j < k ? prep[b].block.slice((size_t)j * S, S) : prep[b].parity.slice((size_t)(j - k) * S, S);
					if (send_shard(mg, (int)node, h, j, payload, S, prep[b].plen, prep[b].z, sums.data() + (b * n + j) * 32, &tags[b])) {
						++okc[b];
						mg->metrics[0] += S;
					}
				}
			}
		});
		for (size_t b = 0; b < nb; ++b)
			oks[b] = okc[b].load();
	} else {
		mg->pool->parallel_for(nb, fan_out);
	}
	tr.lap("fan-out");
	int result = GBM_OK;
	for (size_t b = 0; b < nb; ++b) {
		Hash h((const char *)hashes + 32 * b, 32);
		mg->metrics[4]++;
		if (oks[b] < mg->write_quorum) {
			result = fail(GBM_E_QUORUM, "Could not reach quorum of " + std::to_string(mg->write_quorum) + ". " +
							    std::to_string(oks[b]) + " of " + std::to_string(n) + " request succeeded");
			if (rcs)
				rcs[b] = GBM_E_QUORUM;
			continue;
		}
		// A block that nobody references yet (PutObject runs the put and the block_ref incref
		// concurrently, src/api/s3/put.rs:545-581) is protected for BLOCK_GC_DELAY exactly like one
		// whose count just dropped to zero: resync must never delete what a put just acknowledged.
		{
			gbm_manager::RcStripe &st = mg->rc_of(h);
			std::lock_guard<std::mutex> g(st.mu);
			RcEntry &e = st.map[h];
			if (e.kind != RcEntry::Present) {
				e.kind = RcEntry::Deletable;
				e.v = std::max(e.v, mg->now() + mg->gc_delay_ms.load());
			}
		}
		if (oks[b] < n)
			mg->put_to_resync(h, 0);  // stragglers: resync rebuilds what is absent (it only REBUILDS while the block is needed)
	}
	return result;
}

// gather + verify + decode, in rounds of ONE device trip each (gec_decode_verify_batch: shard checksums, rebuild
// of missing data shards and the block's own blake2sum from a single upload).  A shard whose checksum does not
// match its header is treated the way read_block_from treats a corrupt file (manager.rs:577-609): renamed
// *.corrupted, queued for resync, and the read carries on with the next node.
// On return, for every block with rcs[b] == GBM_OK, g[b].shard[0..k-1] hold the stored DataBlock (plain bytes or
// one zstd frame, orig_len bytes) and block_sums[32*b..] its blake2sum (when want_block_sums).
// `overlap` (optional) runs on a helper thread while the first device trip is in flight -- the caller assembles the
// blocks that need no decode into its output buffers meanwhile; `changed[b]` is set for every block whose shard set
// changed after that point (bit 0: a shard failed its checksum and was replaced, bit 1: a data shard was rebuilt).
int fetch_blocks(gbm_manager *mg, const std::vector<Hash> &hs, const gbm_order_tag *tags, std::vector<Gathered> &g, int *rcs,
		 bool want_block_sums, std::vector<uint8_t> &block_sums, const std::function<void()> &overlap = nullptr,
		 std::vector<uint8_t> *changed = nullptr)
{
	const int k = mg->k, n = mg->n;
	const size_t nb = hs.size();
	block_sums.assign(want_block_sums ? nb * 32 : 0, 0);
	if (changed)
		changed->assign(nb, 0);
	Trace tr("get");
	int grc = gather_many(mg, hs, tags, k, g, /*verify=*/false);
	if (grc)
		return grc;
	tr.lap("gather");
	std::thread helper;
	struct Joiner {
		std::thread &t;
		~Joiner()
		{
			if (t.joinable())
				t.join();
		}
	} joiner{helper};
	if (overlap)
		helper = std::thread(overlap);
	std::vector<uint8_t> todo(nb, 1);
	for (int round = 0; round <= n; ++round) {
		if (round == 1 && helper.joinable())
			helper.join();
		std::map<size_t, std::vector<size_t>> by_s;
		for (size_t b = 0; b < nb; ++b) {
			if (!todo[b])
				continue;
			if (!g[b].have_meta || g[b].count < k) {
				rcs[b] = GBM_E_MISSING_BLOCK;
				todo[b] = 0;
				continue;
			}
			if (g[b].meta.orig_len > (uint64_t)k * g[b].meta.shard_len) {
				rcs[b] = GBM_E_CORRUPT_DATA;
				todo[b] = 0;
				continue;
			}
			by_s[g[b].meta.shard_len].push_back(b);
		}
		if (by_s.empty())
			break;
		std::vector<uint8_t> again(nb, 0);
		bool any_again = false;
		for (auto &kv : by_s) {
			const size_t S = kv.first;
			const std::vector<size_t> &ids = kv.second;
			std::vector<const uint8_t *> sp(ids.size() * n, nullptr);
			std::vector<uint8_t *> op(ids.size() * n, nullptr);
			std::vector<size_t> lens(ids.size());
			std::vector<uint8_t> ssums(ids.size() * (size_t)n * 32), bsums(want_block_sums ? ids.size() * 32 : 0);
			std::vector<std::vector<Bytes>> fresh(ids.size(), std::vector<Bytes>(k));
			size_t nrebuild = 0;
			try {
				for (size_t i = 0; i < ids.size(); ++i) {
					Gathered &gb = g[ids[i]];
					lens[i] = gb.meta.orig_len;
					for (int j = 0; j < n; ++j) {
						if (!gb.shard[j].empty()) {
							sp[i * n + j] = gb.shard[j].data();
						} else if (j < k) {
							fresh[i][j] = mg->bufs->get(S);
							op[i * n + j] = fresh[i][j].mut();
							++nrebuild;
						}
					}
				}
			} catch (const std::bad_alloc &) {
				return fail(GBM_E_IO, "out of (pinned) host memory");
			}
			int rc = gec_decode_verify_batch(mg->codec, ids.size(), sp.data(), S, lens.data(), op.data(), ssums.data(),
							 want_block_sums ? bsums.data() : nullptr);
			tr.lap("decode+verify");
			if (helper.joinable())
				helper.join();  // the overlapped host work reads g: it must be done before the results below change it
			tr.lap("join overlapped assembly");
			if (rc)
				return ec_fail(rc, "gec_decode_verify_batch");
			mg->gpu_hashed += ids.size() * (size_t)k + (want_block_sums ? ids.size() : 0);
			for (size_t i = 0; i < ids.size(); ++i) {
				const size_t b = ids[i];
				Gathered &gb = g[b];
				// the shards that were read: the first k present, in index order
				int seen = 0;
				bool bad = false;
				for (int j = 0; j < n && seen < k; ++j) {
					if (gb.shard[j].empty())
						continue;
					++seen;
					if (std::memcmp(ssums.data() + (i * n + j) * 32, gb.sum[j].data(), 32) != 0) {
						mg->metrics[2]++;
						if (gb.node[j] >= 0)
							mg->nodes[gb.node[j]]->mark_corrupted(hs[b], j);
						mg->put_to_resync(hs[b], 0);
						gb.shard[j] = Bytes();
						gb.count--;
						bad = true;
					}
				}
				if (bad) {
					again[b] = 1;
					any_again = true;
					if (changed)
						(*changed)[b] |= 1;  // a shard in hand was replaced
					continue;
				}
				bool rebuilt_any = false;
				for (int j = 0; j < k; ++j)
					if (!fresh[i][j].empty()) {
						gb.shard[j] = fresh[i][j];
						rebuilt_any = true;
					}
				if (rebuilt_any) {
					mg->metrics[3]++;
					if (changed)
						(*changed)[b] |= 2;  // missing data shards were filled in
				}
				if (want_block_sums)
					std::memcpy(block_sums.data() + 32 * b, bsums.data() + 32 * i, 32);
				rcs[b] = GBM_OK;
				todo[b] = 0;
			}
			(void)nrebuild;
		}
		if (!any_again)
			break;
		if (helper.joinable())
			helper.join();
		grc = gather_many(mg, hs, tags, k, g, /*verify=*/false, &again);  // the next nodes, for the blocks that lost a shard
		if (grc)
			return grc;
	}
	for (size_t b = 0; b < nb; ++b)
		if (todo[b])
			rcs[b] = GBM_E_MISSING_BLOCK;
	return GBM_OK;
}

void assemble(const Gathered &g, int k, uint8_t *dst)
{
	const size_t L = g.meta.orig_len, S = g.meta.shard_len;
	for (int j = 0; j < k; ++j) {
		const size_t lo = (size_t)j * S;
		if (lo >= L)
			break;
		std::memcpy(dst + lo, g.shard[j].data(), std::min(S, L - lo));
	}
}

// raw == true: rpc_get_raw_block (stored bytes + header); false: rpc_get_block (plain bytes).
int get_blocks_impl(gbm_manager *mg, size_t nb, const uint8_t *hashes, const gbm_order_tag *tags, uint8_t *const *out,
		    const size_t *cap, size_t *len_out, int *rcs, bool raw, gbm_data_block_header *headers)
{
	if (!mg || (nb && (!hashes || !out || !cap || !len_out || !rcs)))
		return fail(GBM_E_INVALID_ARG, "NULL argument");
	const int k = mg->k;
	std::vector<Hash> hs(nb);
	for (size_t b = 0; b < nb; ++b)
		hs[b].assign((const char *)hashes + 32 * b, 32);
	std::vector<Gathered> g;
	std::vector<uint8_t> block_sums, changed, early(nb, 0);
	const bool verify = mg->verify_block_hash.load();
	// Where the block's own checksum is computed.  It is one serial BLAKE2b chain per block: ~11 ms per MiB on the
	// device however many blocks run beside it, ~1 ms per MiB on a host core.  Small requests -- a GetObject reads
	// its blocks a few at a time -- are hashed by the pool from the assembled bytes; big batches on the device,
	// behind the upload (gec_decode_verify_batch).
	const bool cpu_hash = verify && nb <= mg->cpu_block_hash_max.load();
	// While the device checks the shards, the host already copies the blocks that need no decode (all k data shards in
	// hand, stored Plain) into the caller's buffers: a block that then fails a checksum is reported as such (its buffer
	// contents are unspecified on error) or is assembled again from the replaced shards.
	// (a block with data shards to rebuild gets the shards it has; the rebuilt ones follow after the trip)
	std::vector<std::vector<uint8_t>> missing_early(nb);  // data shard indices that were not in hand at that point
	auto assemble_early = [&] {
		mg->pool->parallel_for(nb, [&](size_t b) {
			const Gathered &gb = g[b];
			if (!gb.have_meta || gb.count < k || gb.meta.compressed || gb.meta.orig_len > (uint64_t)k * gb.meta.shard_len ||
			    cap[b] < gb.meta.orig_len)
				return;
			const size_t L = gb.meta.orig_len, S = gb.meta.shard_len;
			for (int j = 0; j < k && (size_t)j * S < L; ++j) {
				if (gb.shard[j].empty())
					missing_early[b].push_back((uint8_t)j);
				else
					std::memcpy(out[b] + (size_t)j * S, gb.shard[j].data(), std::min(S, L - (size_t)j * S));
			}
			early[b] = 1;
		});
	};
	int frc = fetch_blocks(mg, hs, tags, g, rcs, verify && !cpu_hash, block_sums, assemble_early, &changed);
	if (frc)
		return frc;
	// assemble (parallel), then check every Plain block's content against its name (DataBlock::verify,
	// block.rs:69-77) -- all block hashes in one batch.  Plain blocks are assembled straight into the
	// caller's buffer and hashed from there (on CORRUPT_DATA its contents are unspecified); compressed
	// blocks go through an intermediate for the zstd frame, whose checksum is their verify.
	mg->pool->parallel_for(nb, [&](size_t b) {
		len_out[b] = 0;
		if (rcs[b] != GBM_OK)
			return;
		const size_t L = g[b].meta.orig_len;
		const bool z = g[b].meta.compressed != 0;
		if (headers)
			headers[b].kind = z ? GBM_HEADER_COMPRESSED : GBM_HEADER_PLAIN;
		len_out[b] = L;
		// DataBlock::verify (block.rs:69-83): Plain = content against its name -- the block's blake2sum came back
		// from the same device trip that decoded it (or is computed below, cpu_hash); Compressed = the zstd frame
		// (with its checksum) decodes
		if (!z && verify && !cpu_hash && std::memcmp(block_sums.data() + 32 * b, hashes + 32 * b, 32) != 0) {
			rcs[b] = GBM_E_CORRUPT_DATA;
			return;
		}
		if (z && !raw) {
			std::vector<uint8_t> frame(L), plain;
			assemble(g[b], k, frame.data());
			if (!zstd().decode(frame.data(), L, kMaxDecompressed, plain)) {
				rcs[b] = GBM_E_CORRUPT_DATA;
				return;
			}
			len_out[b] = plain.size();
			if (cap[b] < plain.size()) {
				rcs[b] = GBM_E_BUFFER_TOO_SMALL;
				return;
			}
			if (!plain.empty())
				std::memcpy(out[b], plain.data(), plain.size());
			mg->metrics[5]++;
			return;
		}
		if (cap[b] < L) {
			rcs[b] = GBM_E_BUFFER_TOO_SMALL;
			return;
		}
		if (early[b] && !(changed[b] & 1)) {
			const size_t S = g[b].meta.shard_len;
			for (uint8_t j : missing_early[b])  // rebuilt since
				std::memcpy(out[b] + (size_t)j * S, g[b].shard[j].data(), std::min(S, L - (size_t)j * S));
		} else {
			assemble(g[b], k, out[b]);
		}
		if (!z && cpu_hash) {
			uint8_t sum[32];
			blake2sum(out[b], L, sum);
			if (std::memcmp(sum, hashes + 32 * b, 32) != 0) {
				rcs[b] = GBM_E_CORRUPT_DATA;
				return;
			}
		}
		mg->metrics[5]++;
	});
	return GBM_OK;
}

int one_block_rc(int rc1)
{
	switch (rc1) {
	case GBM_E_MISSING_BLOCK: return fail(rc1, "Missing block: no node returned a valid block");
	case GBM_E_CORRUPT_DATA: return fail(rc1, "Corrupt data: does not match hash");
	case GBM_E_BUFFER_TOO_SMALL: return fail(rc1, "output buffer too small");
	default: return rc1;
	}
}

// ------------------------------------------------------------------ resync
struct ResyncStats {
	uint64_t taken = 0, ok = 0, errors = 0, skipped = 0, rebuilt = 0, deleted = 0, offloaded = 0, device_calls = 0;
};

// What one block needs, decided from the refcount and a presence scan (NeedShardQuery to every node that
// could hold a shard: no payload moves).
struct ResyncTask {
	Hash h;
	std::vector<int> who;              // current layout
	std::vector<uint8_t> present_cur;  // shard j present on its current node
	std::vector<uint8_t> reachable;    // current node of shard j is up
	struct Stray {
		int version, idx, node;
	};
	std::vector<Stray> strays;         // shards sitting on nodes of older layout versions
	bool exists = false;
	RcEntry rc;
	std::string error;
	int changed = 0;
	// rebuild
	std::vector<int> want;             // absent on a reachable current node, not recoverable by offload
	Gathered g;
};

void scan_block(gbm_manager *mg, ResyncTask &t)
{
	const int n = mg->n;
	const int vcur = mg->layout_cur.load(), vold = mg->layout_oldest.load();
	mg->nodes_of(t.h, vcur, t.who);
	t.present_cur.assign(n, 0);
	t.reachable.assign(n, 0);
	for (int j = 0; j < n; ++j) {
		ShardRpc rq{RpcKind::NeedShardQuery, &t.h, j, Shard(), nullptr};
		ShardResp rs;
		if (mg->nodes[t.who[j]]->handle(rq, rs)) {
			t.reachable[j] = 1;
			t.present_cur[j] = rs.needed ? 0 : 1;
			t.exists = t.exists || !rs.needed;
		}
	}
	std::vector<int> who;
	for (int v = vcur - 1; v >= vold; --v) {
		mg->nodes_of(t.h, v, who);
		for (int j = 0; j < n; ++j) {
			if (who[j] == t.who[j])
				continue;
			ShardRpc rq{RpcKind::NeedShardQuery, &t.h, j, Shard(), nullptr};
			ShardResp rs;
			if (mg->nodes[who[j]]->handle(rq, rs) && !rs.needed) {
				t.strays.push_back({v, j, who[j]});
				t.exists = true;
			}
		}
	}
	t.rc = mg->get_rc(t.h);
}

// resync_block for a set of blocks (src/block/resync.rs:354-503), with the device work of all of them batched.
void resync_blocks(gbm_manager *mg, std::vector<ResyncTask> &tasks, ResyncStats &st)
{
	const int k = mg->k, n = mg->n;
	const uint64_t now = mg->now();
	Trace tr("resync");
	mg->pool->parallel_for(tasks.size(), [&](size_t i) { scan_block(mg, tasks[i]); });
	tr.lap("presence scan");
	std::vector<size_t> rebuild;
	std::atomic<uint64_t> deleted{0}, offloaded{0};
	mg->pool->parallel_for(tasks.size(), [&](size_t i) {
		ResyncTask &t = tasks[i];
		if (t.exists && t.rc.is_deletable(now)) {
			// "offloading and deleting" -- with one refcount for the whole (in-process) cluster a deletable
			// block is needed by nobody, so NeedShardQuery has no taker and the offload set is empty
			for (int j = 0; j < n; ++j)
				if (t.present_cur[j]) {
					ShardRpc rq{RpcKind::DeleteShard, &t.h, j, Shard(), nullptr};
					ShardResp rs;
					if (mg->nodes[t.who[j]]->handle(rq, rs) && rs.ok)
						++t.changed;
				}
			for (auto &s : t.strays) {
				ShardRpc rq{RpcKind::DeleteShard, &t.h, s.idx, Shard(), nullptr};
				ShardResp rs;
				if (mg->nodes[s.node]->handle(rq, rs) && rs.ok)
					++t.changed;
			}
			deleted += t.changed;
			// clear_deleted_block_rc
			gbm_manager::RcStripe &rs = mg->rc_of(t.h);
			std::lock_guard<std::mutex> g(rs.mu);
			auto it = rs.map.find(t.h);
			if (it != rs.map.end() && it->second.kind == RcEntry::Deletable && now > it->second.v)
				rs.map.erase(it);
			return;
		}
		if (!t.rc.is_needed(now))
			return;  // nothing stored, nothing needed
		// needed.  First the offload branch: a shard that a layout change left on its old node is sent to the
		// owner that lacks it (PutShard), then deleted where it no longer belongs.
		for (auto &s : t.strays) {
			if (!t.present_cur[s.idx] && t.reachable[s.idx]) {
				ShardRpc rq{RpcKind::GetShard, &t.h, s.idx, Shard(), nullptr};
				ShardResp rs;
				if (!mg->nodes[s.node]->handle(rq, rs) || !rs.ok)
					continue;
				uint8_t sum[32];
				shardsum(rs.shard.data.data(), rs.shard.data.n, sum);
				if (rs.shard.data.n != rs.shard.hd.shard_len || std::memcmp(sum, rs.shard.hd.checksum, 32) != 0) {
					mg->metrics[2]++;
					mg->nodes[s.node]->mark_corrupted(t.h, s.idx);
					continue;
				}
				ShardRpc pq{RpcKind::PutShard, &t.h, s.idx, rs.shard, nullptr};
				ShardResp ps;
				if (!mg->nodes[t.who[s.idx]]->handle(pq, ps) || !ps.ok) {
					t.error = "offload: PutShard to the new owner failed";
					continue;
				}
				t.present_cur[s.idx] = 1;
				++t.changed;
				++offloaded;
			}
			if (t.present_cur[s.idx]) {  // the owner has it: the stray copy is unneeded
				ShardRpc rq{RpcKind::DeleteShard, &t.h, s.idx, Shard(), nullptr};
				ShardResp rs;
				(void)mg->nodes[s.node]->handle(rq, rs);
			}
		}
		for (int j = 0; j < n; ++j)
			if (!t.present_cur[j]) {
				if (t.reachable[j])
					t.want.push_back(j);
				else
					t.error = "storage node of shard " + std::to_string(j) + " could not be contacted";
			}
	});
	st.deleted += deleted.load();
	st.offloaded += offloaded.load();
	tr.lap("delete / offload");
	for (size_t i = 0; i < tasks.size(); ++i)
		if (!tasks[i].want.empty())
			rebuild.push_back(i);
	// "fetching absent but needed block" (resync.rs:485-499): gather exactly k shards per block, rebuild what is wanted,
	// PutShard.  First pass: shards are accepted on their headers and ONE device trip per group both rebuilds and
	// returns the checksums of what it read (compared with the headers) and of what it wrote (stamped into the new
	// headers) -- gec_reconstruct_hash_batch.  A block that turns out to have read a corrupt shard (set aside,
	// queued) goes through a second pass whose gather verifies checksums first and moves on to the next holder.
	auto rebuild_pass = [&](const std::vector<size_t> &todo, bool verify_in_gather) -> std::vector<size_t> {
		std::vector<size_t> again;
		std::vector<Hash> hs;
		for (size_t i : todo)
			hs.push_back(tasks[i].h);
		std::vector<Gathered> gs;
		int grc = gather_many(mg, hs, nullptr, k, gs, verify_in_gather);
		tr.lap(verify_in_gather ? "gather k + checksums" : "gather k");
		for (size_t q = 0; q < todo.size(); ++q) {
			ResyncTask &t = tasks[todo[q]];
			if (grc) {
				t.error = std::string("gather: ") + g_err;
				t.want.clear();
				continue;
			}
			t.g = std::move(gs[q]);
			if (!t.g.have_meta || t.g.count < k) {
				t.error = "Missing block: fewer than k shards reachable";
				t.want.clear();
				continue;
			}
			// the read may have found corrupt shards (renamed away): those are absent now as well
			for (int j = 0; j < n; ++j)
				if (t.reachable[j] && t.g.shard[j].empty() && std::find(t.want.begin(), t.want.end(), j) == t.want.end()) {
					ShardRpc rq{RpcKind::NeedShardQuery, &t.h, j, Shard(), nullptr};
					ShardResp rs;
					if (mg->nodes[t.who[j]]->handle(rq, rs) && rs.needed)
						t.want.push_back(j);
				}
			// shards of a minority geometry are stale: overwrite them
			if (t.g.mixed)
				for (int j = 0; j < n; ++j)
					if (t.reachable[j] && t.g.shard[j].empty() && std::find(t.want.begin(), t.want.end(), j) == t.want.end())
						t.want.push_back(j);
		}
		// ONE device call per shard length: inside it gec_reconstruct_hash_batch buckets the blocks by (which shards
		// are in hand, which are wanted) -- one decode plan and one kernel launch per such erasure pattern, the patterns'
		// chunks pipelined through the link without a host round trip in between (one call per pattern: 14 calls,
		// 20 ms for a lost node's 449 shards; one call: see tools/host_path_bench.py maintenance)
		std::map<size_t, std::vector<size_t>> groups;
		for (size_t i : todo)
			if (!tasks[i].want.empty())
				groups[tasks[i].g.meta.shard_len].push_back(i);
		for (auto &kv : groups) {
			const size_t S = kv.first;
			const std::vector<size_t> &ids = kv.second;
			std::vector<const uint8_t *> sp(ids.size() * n, nullptr);
			std::vector<uint8_t *> op(ids.size() * n, nullptr);
			std::vector<std::vector<Bytes>> outb(ids.size(), std::vector<Bytes>(n));
			std::vector<uint8_t> in_sums(ids.size() * (size_t)n * 32), out_sums(ids.size() * (size_t)n * 32);
			bool oom = false;
			// one pinned slab for the group's rebuilt shards (a first-time allocation per shard costs more than
			// the decode), sliced per shard: the nodes keep the slices, the slab lives as long as any of them
			size_t nwant = 0;
			for (size_t q = 0; q < ids.size(); ++q)
				nwant += tasks[ids[q]].want.size();
			Bytes slab;
			try {
				slab = mg->bufs->get(nwant * S);
			} catch (const std::bad_alloc &) {
				oom = true;
			}
			size_t slot = 0;
			for (size_t q = 0; q < ids.size() && !oom; ++q) {
				ResyncTask &t = tasks[ids[q]];
				for (int j = 0; j < n; ++j)
					if (!t.g.shard[j].empty())
						sp[q * n + j] = t.g.shard[j].data();
				for (int j : t.want) {
					outb[q][j] = slab.slice(slot * S, S);
					op[q * n + j] = outb[q][j].mut();
					++slot;
				}
			}
			tr.lap("group setup");
			int rc = oom ? GEC_E_NOMEM
				     : gec_reconstruct_hash_batch(mg->codec, ids.size(), sp.data(), op.data(), S, 0, in_sums.data(), out_sums.data());
			tr.lap("reconstruct + checksums");
			++st.device_calls;
			if (rc) {
				ec_fail(rc, "gec_reconstruct_hash_batch");
				for (size_t i : ids)
					tasks[i].error = g_err;
				continue;
			}
			mg->gpu_hashed += ids.size() * (size_t)k + nwant;
			std::vector<uint8_t> good(ids.size(), 1);
			if (!verify_in_gather) {
				// what was read: the first k shards in hand, in index order
				for (size_t q = 0; q < ids.size(); ++q) {
					ResyncTask &t = tasks[ids[q]];
					int seen = 0;
					for (int j = 0; j < n && seen < k; ++j) {
						if (t.g.shard[j].empty())
							continue;
						++seen;
						if (std::memcmp(in_sums.data() + (q * n + j) * 32, t.g.sum[j].data(), 32) != 0) {
							mg->metrics[2]++;
							if (t.g.node[j] >= 0)
								mg->nodes[t.g.node[j]]->mark_corrupted(t.h, j);
							good[q] = 0;
						}
					}
					if (!good[q]) {
						t.want.clear();  // decided again by the second pass
						again.push_back(ids[q]);
					}
				}
			}
			mg->metrics[3] += ids.size();
			std::atomic<uint64_t> rebuilt{0};
			mg->pool->parallel_for(ids.size(), [&](size_t q) {
				if (!good[q])
					return;
				ResyncTask &t = tasks[ids[q]];
				for (int j : t.want) {
					if (send_shard(mg, t.who[j], t.h, j, outb[q][j], S, t.g.meta.orig_len, t.g.meta.compressed != 0,
						       out_sums.data() + (q * n + j) * 32, nullptr)) {
						++t.changed;
						++rebuilt;
					} else {
						t.error = "PutShard of a rebuilt shard failed";
					}
				}
			});
			st.rebuilt += rebuilt.load();
			tr.lap("PutShard");
		}
		return again;
	};
	if (!rebuild.empty()) {
		std::vector<size_t> again = rebuild_pass(rebuild, false);
		if (!again.empty()) {
			// the presence of the shards that were set aside has changed: scan those blocks again
			for (size_t i : again) {
				ResyncTask &t = tasks[i];
				for (int j = 0; j < n; ++j)
					if (t.reachable[j]) {
						ShardRpc rq{RpcKind::NeedShardQuery, &t.h, j, Shard(), nullptr};
						ShardResp rs;
						if (mg->nodes[t.who[j]]->handle(rq, rs) && rs.needed)
							t.want.push_back(j);
					}
			}
			(void)rebuild_pass(again, true);
		}
	}
}

}  // namespace

extern "C" {

const char *gbm_last_error(void) { return g_err.c_str(); }

void gbm_blake2sum(const uint8_t *data, size_t len, uint8_t out[32]) { blake2sum(data, len, out); }

void gbm_shardsum(const uint8_t *data, size_t len, uint8_t out[32]) { shardsum(data, len, out); }

int gbm_create(const gec_codec *codec, int nnodes, const char *const *node_dirs, int write_quorum, gbm_manager **out)
{
	if (!codec || !out)
		return fail(GBM_E_INVALID_ARG, "NULL argument");
	*out = nullptr;
	const int k = gec_codec_k(codec), m = gec_codec_m(codec);
	if (nnodes < k + m)
		return fail(GBM_E_INVALID_ARG, "RS(k,m) needs at least k+m storage nodes (replication_factor == k+m)");
	if (k + m > 255)
		return fail(GBM_E_INVALID_ARG, "shard index must fit a byte");
	auto mg = std::make_unique<gbm_manager>();
	mg->codec = codec;
	mg->k = k;
	mg->m = m;
	mg->n = k + m;
	mg->write_quorum = write_quorum > 0 ? write_quorum : k + (m + 1) / 2;
	if (mg->write_quorum < k || mg->write_quorum > mg->n)
		return fail(GBM_E_INVALID_ARG, "write quorum must be in [k, k+m]");
	for (int i = 0; i < nnodes; ++i) {
		if (node_dirs)
			mg->nodes.emplace_back(new DirNode(node_dirs[i]));
		else
			mg->nodes.emplace_back(new MemoryNode());
		mg->nodes.back()->bufs = mg->bufs;
	}
	const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
	mg->pool.reset(new Pool(std::min(15u, hw - 1)));
	*out = mg.release();
	return GBM_OK;
}

void gbm_destroy(gbm_manager *m)
{
	if (!m)
		return;
	gbm_resync_worker_stop(m);
	m->async.reset();  // drains: abandoned hedged requests still point at the nodes
	delete m;
}

int gbm_set_threads(gbm_manager *m, int nthreads)
{
	if (!m || nthreads < 1 || nthreads > 256)
		return fail(GBM_E_INVALID_ARG, "need 1 <= nthreads <= 256");
	m->pool->resize((unsigned)nthreads - 1);  // the calling thread works too
	m->cpu_block_hash_max = 6 * (size_t)nthreads;
	return GBM_OK;
}

int gbm_set_host_block_hash_max(gbm_manager *m, size_t nblocks)
{
	if (!m)
		return fail(GBM_E_INVALID_ARG, "NULL manager");
	m->cpu_block_hash_max = nblocks;
	return GBM_OK;
}

int gbm_set_data_fsync(gbm_manager *m, int enabled)
{
	if (!m)
		return fail(GBM_E_INVALID_ARG, "NULL manager");
	for (auto &nd : m->nodes)
		if (DirNode *d = dynamic_cast<DirNode *>(nd.get()))
			d->fsync_data = enabled != 0;
	return GBM_OK;
}

int gbm_set_verify_block_hash(gbm_manager *m, int enabled)
{
	if (!m)
		return fail(GBM_E_INVALID_ARG, "NULL manager");
	m->verify_block_hash = enabled != 0;
	return GBM_OK;
}

int gbm_set_read_hedge(gbm_manager *m, uint64_t hedge_us)
{
	if (!m)
		return fail(GBM_E_INVALID_ARG, "NULL manager");
	m->hedge_us = hedge_us;
	return GBM_OK;
}

uint64_t gbm_hedged_reads(const gbm_manager *m) { return m ? m->hedged_reads.load() : 0; }

int gbm_node_set_latency(gbm_manager *m, int node, uint64_t latency_us)
{
	if (!m || node < 0 || node >= (int)m->nodes.size())
		return fail(GBM_E_INVALID_ARG, "bad node");
	m->nodes[node]->latency_us = latency_us;
	return GBM_OK;
}

int gbm_set_timing(gbm_manager *m, int64_t gc_delay_ms, int64_t resync_retry_delay_ms, int64_t incref_check_delay_ms)
{
	if (!m)
		return fail(GBM_E_INVALID_ARG, "NULL manager");
	if (gc_delay_ms >= 0)
		m->gc_delay_ms = (uint64_t)gc_delay_ms;
	if (resync_retry_delay_ms >= 0)
		m->retry_delay_ms = (uint64_t)resync_retry_delay_ms;
	if (incref_check_delay_ms >= 0)
		m->incref_delay_ms = (uint64_t)incref_check_delay_ms;
	return GBM_OK;
}

int gbm_clock_advance(gbm_manager *m, uint64_t ms)
{
	if (!m)
		return fail(GBM_E_INVALID_ARG, "NULL manager");
	m->clock_skew_ms += ms;
	m->rs_cv.notify_all();
	return GBM_OK;
}

int gbm_storage_nodes_of(const gbm_manager *m, const uint8_t hash[32], int *nodes_out)
{
	if (!m || !hash || !nodes_out)
		return fail(GBM_E_INVALID_ARG, "NULL argument");
	std::vector<int> who;
	m->nodes_of(Hash((const char *)hash, 32), who);
	std::copy(who.begin(), who.end(), nodes_out);
	return GBM_OK;
}

int gbm_layout_update(gbm_manager *m)
{
	if (!m)
		return fail(GBM_E_INVALID_ARG, "NULL manager");
	return ++m->layout_cur;
}

int gbm_layout_trim(gbm_manager *m)
{
	if (!m)
		return fail(GBM_E_INVALID_ARG, "NULL manager");
	m->layout_oldest = m->layout_cur.load();
	return GBM_OK;
}

int gbm_rpc_put_blocks(gbm_manager *mg, size_t nb, const uint8_t *hashes, const uint8_t *const *data, const size_t *len,
		       const uint8_t *prevent_compression, const gbm_order_tag *order_tags)
{
	// Large untagged batches go through in slices of 64 blocks on four threads, so that one slice's host work (the copy
	// into the shard buffers, the fan-out) runs while other slices are on the link: 512 x 1 MiB blocks 26 -> 38 GiB/s
	// (tools/bm_sweep.sh: 128 x 2 threads 33, 128 x 3 37, 64 x 4 38.6, 64 x 8 39.5).  Tagged batches keep their order.
	static const size_t kSlice = [] {
		const char *e = std::getenv("GBM_PUT_SLICE");
		const long v = e ? std::atol(e) : 0;
		return (size_t)(v > 0 ? v : 64);
	}();
	static const int kThreads = [] {
		const char *e = std::getenv("GBM_PUT_THREADS");
		const int v = e ? std::atoi(e) : 0;
		return v > 0 && v <= 8 ? v : 4;
	}();
	try {
		if (!mg || order_tags || nb < 2 * kSlice)
			return put_blocks_impl(mg, nb, hashes, data, len, prevent_compression, order_tags, nullptr);
		std::atomic<size_t> next{0};
		std::mutex mu;
		int result = GBM_OK;
		std::string err;
		auto run = [&] {
			for (;;) {
				const size_t b0 = next.fetch_add(kSlice);
				if (b0 >= nb)
					return;
				const size_t cnt = std::min(kSlice, nb - b0);
				int rc;
				try {
					rc = put_blocks_impl(mg, cnt, hashes + 32 * b0, data + b0, len + b0,
							     prevent_compression ? prevent_compression + b0 : nullptr, nullptr, nullptr);
				} catch (const std::exception &e) {
					rc = fail(GBM_E_IO, std::string("rpc_put_blocks: ") + e.what());
				}
				if (rc) {
					std::lock_guard<std::mutex> g(mu);
					result = rc;
					err = g_err;  // the error text is thread-local: carry it to the caller's thread
				}
			}
		};
		std::vector<std::thread> others;
		for (int t = 1; t < kThreads; ++t)
			others.emplace_back(run);
		run();
		for (auto &t : others)
			t.join();
		if (result)
			return fail(result, err);
		return GBM_OK;
	} catch (const std::exception &e) {
		return fail(GBM_E_IO, std::string("rpc_put_blocks: ") + e.what());
	}
}

int gbm_rpc_put_block(gbm_manager *m, const uint8_t hash[32], const uint8_t *data, size_t len, int prevent_compression,
		      const gbm_order_tag *order_tag)
{
	const uint8_t *d[1] = {data};
	const uint8_t pc = prevent_compression ? 1 : 0;
	return gbm_rpc_put_blocks(m, 1, hash, d, &len, &pc, order_tag);
}

int gbm_rpc_get_blocks(gbm_manager *mg, size_t nb, const uint8_t *hashes, const gbm_order_tag *order_tags, uint8_t *const *out,
		       const size_t *cap, size_t *len_out, int *rcs)
{
	try {
		return get_blocks_impl(mg, nb, hashes, order_tags, out, cap, len_out, rcs, false, nullptr);
	} catch (const std::exception &e) {
		return fail(GBM_E_IO, std::string("rpc_get_blocks: ") + e.what());
	}
}

int gbm_rpc_get_block(gbm_manager *m, const uint8_t hash[32], const gbm_order_tag *order_tag, uint8_t *out, size_t cap,
		      size_t *len_out)
{
	if (!len_out)
		return fail(GBM_E_INVALID_ARG, "NULL len_out");
	uint8_t *o[1] = {out};
	int rc1 = GBM_OK;
	int rc = gbm_rpc_get_blocks(m, 1, hash, order_tag, o, &cap, len_out, &rc1);
	return rc ? rc : one_block_rc(rc1);
}

int gbm_rpc_get_raw_block(gbm_manager *m, const uint8_t hash[32], const gbm_order_tag *order_tag,
			  gbm_data_block_header *header_out, uint8_t *out, size_t cap, size_t *len_out)
{
	if (!len_out || !header_out)
		return fail(GBM_E_INVALID_ARG, "NULL argument");
	uint8_t *o[1] = {out};
	int rc1 = GBM_OK;
	int rc;
	try {
		rc = get_blocks_impl(m, 1, hash, order_tag, o, &cap, len_out, &rc1, true, header_out);
	} catch (const std::exception &e) {
		return fail(GBM_E_IO, std::string("rpc_get_raw_block: ") + e.what());
	}
	return rc ? rc : one_block_rc(rc1);
}

static int stream_out(const uint8_t *p, size_t len, size_t chunk_bytes, gbm_chunk_fn sink, void *ctx)
{
	const size_t ch = chunk_bytes ? chunk_bytes : 65536;
	for (size_t off = 0; off < len; off += ch)
		if (sink(ctx, p + off, std::min(ch, len - off)) != 0)
			return fail(GBM_E_ABORTED, "the stream's consumer stopped");
	return GBM_OK;
}

// one block into a buffer this function owns (the streaming forms do not know the size up front)
static int get_block_owned(gbm_manager *mg, const uint8_t hash[32], const gbm_order_tag *tag, bool raw,
			   gbm_data_block_header *hdr, std::vector<uint8_t> &out)
{
	const int k = mg->k;
	std::vector<Hash> hs(1, Hash((const char *)hash, 32));
	std::vector<Gathered> g;
	std::vector<uint8_t> block_sums;
	int rc1 = GBM_OK;
	const bool verify = mg->verify_block_hash.load();
	int frc = fetch_blocks(mg, hs, tag, g, &rc1, /*want_block_sums=*/false, block_sums);  // one block: hashed on the host, below
	if (frc)
		return frc;
	if (rc1 != GBM_OK)
		return one_block_rc(rc1);
	const size_t L = g[0].meta.orig_len;
	const bool z = g[0].meta.compressed != 0;
	if (hdr)
		hdr->kind = z ? GBM_HEADER_COMPRESSED : GBM_HEADER_PLAIN;
	std::vector<uint8_t> stored(L);
	assemble(g[0], k, stored.data());
	if (!z && verify) {
		uint8_t sum[32];
		blake2sum(stored.data(), L, sum);
		if (std::memcmp(sum, hash, 32) != 0)
			return one_block_rc(GBM_E_CORRUPT_DATA);
	}
	if (z && !raw) {
		if (!zstd().decode(stored.data(), L, kMaxDecompressed, out))
			return one_block_rc(GBM_E_CORRUPT_DATA);
	} else {
		out.swap(stored);
	}
	mg->metrics[5]++;
	return GBM_OK;
}

static int get_streaming(gbm_manager *m, const uint8_t hash[32], const gbm_order_tag *order_tag, gbm_data_block_header *hdr,
			 size_t chunk_bytes, gbm_chunk_fn sink, void *ctx, bool raw)
{
	if (!m || !hash || !sink)
		return fail(GBM_E_INVALID_ARG, "NULL argument");
	// the shards have to be complete before any byte can be trusted (checksums, decode), so the block is gathered
	// first and then handed out in order
	std::vector<uint8_t> buf;
	gbm_data_block_header h0{};
	int rc = get_block_owned(m, hash, order_tag, raw, &h0, buf);
	if (rc)
		return rc;
	if (hdr)
		*hdr = h0;
	return stream_out(buf.data(), buf.size(), chunk_bytes, sink, ctx);
}

int gbm_rpc_get_block_streaming(gbm_manager *m, const uint8_t hash[32], const gbm_order_tag *order_tag, size_t chunk_bytes,
				gbm_chunk_fn sink, void *ctx)
{
	try {
		return get_streaming(m, hash, order_tag, nullptr, chunk_bytes, sink, ctx, false);
	} catch (const std::exception &e) {
		return fail(GBM_E_IO, std::string("rpc_get_block_streaming: ") + e.what());
	}
}

int gbm_rpc_get_raw_block_streaming(gbm_manager *m, const uint8_t hash[32], const gbm_order_tag *order_tag,
				    gbm_data_block_header *header_out, size_t chunk_bytes, gbm_chunk_fn sink, void *ctx)
{
	if (!header_out)
		return fail(GBM_E_INVALID_ARG, "NULL header_out");
	try {
		return get_streaming(m, hash, order_tag, header_out, chunk_bytes, sink, ctx, true);
	} catch (const std::exception &e) {
		return fail(GBM_E_IO, std::string("rpc_get_raw_block_streaming: ") + e.what());
	}
}

int gbm_block_incref(gbm_manager *m, const uint8_t hash[32])
{
	if (!m || !hash)
		return fail(GBM_E_INVALID_ARG, "NULL argument");
	Hash h((const char *)hash, 32);
	bool was_zero;
	{
		gbm_manager::RcStripe &s = m->rc_of(h);
		std::lock_guard<std::mutex> lk(s.mu);
		RcEntry &e = s.map[h];
		was_zero = e.is_zero();
		e.v = e.kind == RcEntry::Present ? e.v + 1 : 1;
		e.kind = RcEntry::Present;
	}
	// "there is normally a node that is responsible for sending us the data of the block.  However that
	// operation may fail, so in all cases we add the block here to the todo list" (manager.rs:452-475)
	if (was_zero)
		m->put_to_resync(h, m->incref_delay_ms.load());
	return GBM_OK;
}

int gbm_block_decref(gbm_manager *m, const uint8_t hash[32])
{
	if (!m || !hash)
		return fail(GBM_E_INVALID_ARG, "NULL argument");
	Hash h((const char *)hash, 32);
	bool deletable = false;
	{
		gbm_manager::RcStripe &s = m->rc_of(h);
		std::lock_guard<std::mutex> lk(s.mu);
		auto it = s.map.find(h);
		if (it != s.map.end() && it->second.kind == RcEntry::Present) {
			if (it->second.v > 1) {
				--it->second.v;
			} else {
				it->second.kind = RcEntry::Deletable;
				it->second.v = m->now() + m->gc_delay_ms.load();
				deletable = true;
			}
		}  // Deletable / Absent stay what they are (RcEntry::decrement)
	}
	if (deletable)  // handled in the resync loop after the GC delay has passed (manager.rs:478-500)
		m->put_to_resync(h, m->gc_delay_ms.load() + 10000);
	return GBM_OK;
}

int gbm_block_rc(gbm_manager *m, const uint8_t hash[32], uint64_t out[3])
{
	if (!m || !hash || !out)
		return fail(GBM_E_INVALID_ARG, "NULL argument");
	RcEntry e = m->get_rc(Hash((const char *)hash, 32));
	out[0] = e.kind == RcEntry::Present ? e.v : 0;
	out[1] = e.kind;
	out[2] = e.kind == RcEntry::Deletable ? e.v : 0;
	return GBM_OK;
}

int gbm_put_to_resync(gbm_manager *m, const uint8_t hash[32], uint64_t delay_ms)
{
	if (!m || !hash)
		return fail(GBM_E_INVALID_ARG, "NULL argument");
	m->put_to_resync(Hash((const char *)hash, 32), delay_ms);
	return GBM_OK;
}

// One pass of resync_iter over everything that is due (resync.rs:255-337).
int gbm_resync_run(gbm_manager *mg, size_t max_blocks, uint64_t stats[8])
{
	if (!mg)
		return fail(GBM_E_INVALID_ARG, "NULL manager");
	ResyncStats st;
	std::vector<ResyncTask> tasks;
	std::vector<std::pair<uint64_t, Hash>> taken;
	const uint64_t now = mg->now(), base = mg->retry_delay_ms.load();
	{
		std::lock_guard<std::mutex> g(mg->rs_mu);
		std::set<Hash> seen;
		for (auto it = mg->rs_queue.begin(); it != mg->rs_queue.end() && it->first <= now;) {
			if (max_blocks && taken.size() >= max_blocks)
				break;
			const Hash &h = it->second;
			auto ec = mg->rs_errors.find(h);
			if (ec != mg->rs_errors.end() && now < ec->second.next_try(base)) {
				// still inside the back-off: keep the entry, at the time it may be retried
				mg->rs_queue.insert({ec->second.next_try(base), h});
				it = mg->rs_queue.erase(it);
				++st.skipped;
				continue;
			}
			if (seen.insert(h).second)
				taken.push_back(*it);
			it = mg->rs_queue.erase(it);
		}
	}
	st.taken = taken.size();
	tasks.resize(taken.size());
	for (size_t i = 0; i < taken.size(); ++i)
		tasks[i].h = taken[i].second;
	int result = GBM_OK;
	try {
		resync_blocks(mg, tasks, st);
	} catch (const std::exception &e) {
		for (auto &t : tasks)
			if (t.error.empty())
				t.error = e.what();
	}
	{
		std::lock_guard<std::mutex> g(mg->rs_mu);
		for (auto &t : tasks) {
			if (t.error.empty()) {
				mg->rs_errors.erase(t.h);
				++st.ok;
				continue;
			}
			++st.errors;
			result = fail(t.error.rfind("Missing block", 0) == 0 ? GBM_E_MISSING_BLOCK : GBM_E_IO, t.error);
			ErrorCounter &ec = mg->rs_errors[t.h];
			ec.errors += 1;
			ec.last_try = now + 1;
			mg->rs_queue.insert({ec.next_try(base), t.h});
		}
	}
	if (stats) {
		const uint64_t v[8] = {st.taken, st.ok, st.errors, st.skipped, st.rebuilt, st.deleted, st.offloaded, st.device_calls};
		std::copy(v, v + 8, stats);
	}
	return result;
}

int gbm_resync_block(gbm_manager *mg, const uint8_t hash[32], int *changed)
{
	if (!mg || !hash)
		return fail(GBM_E_INVALID_ARG, "NULL argument");
	std::vector<ResyncTask> tasks(1);
	tasks[0].h.assign((const char *)hash, 32);
	ResyncStats st;
	try {
		resync_blocks(mg, tasks, st);
	} catch (const std::exception &e) {
		return fail(GBM_E_IO, std::string("resync_block: ") + e.what());
	}
	if (changed)
		*changed = tasks[0].changed;
	if (!tasks[0].error.empty())
		return fail(tasks[0].error.rfind("Missing block", 0) == 0 ? GBM_E_MISSING_BLOCK : GBM_E_IO, tasks[0].error);
	return GBM_OK;
}

int gbm_resync_all(gbm_manager *mg, int *changed)
{
	if (!mg)
		return fail(GBM_E_INVALID_ARG, "NULL manager");
	int total = 0, result = GBM_OK;
	for (int round = 0; round < 64; ++round) {
		uint64_t st[8];
		int rc = gbm_resync_run(mg, 0, st);
		if (rc)
			result = rc;
		total += (int)(st[4] + st[5] + st[6]);
		if (st[0] == 0)
			break;
	}
	if (changed)
		*changed = total;
	return result;
}

size_t gbm_resync_queue_len(const gbm_manager *m)
{
	if (!m)
		return 0;
	std::lock_guard<std::mutex> lk(m->rs_mu);
	return m->rs_queue.size();
}

size_t gbm_resync_errors_len(const gbm_manager *m)
{
	if (!m)
		return 0;
	std::lock_guard<std::mutex> lk(m->rs_mu);
	return m->rs_errors.size();
}

int gbm_resync_worker_start(gbm_manager *m)
{
	if (!m)
		return fail(GBM_E_INVALID_ARG, "NULL manager");
	std::lock_guard<std::mutex> g(m->rs_mu);
	if (m->rs_worker.joinable())
		return GBM_OK;
	m->rs_worker_stop = false;
	m->rs_worker = std::thread([m] {
		std::unique_lock<std::mutex> lk(m->rs_mu);
		while (!m->rs_worker_stop) {
			const uint64_t now = m->now();
			if (!m->rs_queue.empty() && m->rs_queue.begin()->first <= now) {
				lk.unlock();
				(void)gbm_resync_run(m, 1024, nullptr);
				lk.lock();
				continue;
			}
			// idle until the first entry is due, something is queued, or 10 s pass (resync.rs:325-336)
			uint64_t wait_ms = 10000;
			if (!m->rs_queue.empty())
				wait_ms = std::min<uint64_t>(wait_ms, m->rs_queue.begin()->first - now);
			m->rs_cv.wait_until(lk, std::chrono::system_clock::now() + std::chrono::milliseconds(std::max<uint64_t>(wait_ms, 1)));
		}
	});
	return GBM_OK;
}

int gbm_resync_worker_stop(gbm_manager *m)
{
	if (!m)
		return fail(GBM_E_INVALID_ARG, "NULL manager");
	std::thread t;
	{
		std::lock_guard<std::mutex> g(m->rs_mu);
		if (!m->rs_worker.joinable())
			return GBM_OK;
		m->rs_worker_stop = true;
		t = std::move(m->rs_worker);
	}
	m->rs_cv.notify_all();
	t.join();
	return GBM_OK;
}

int gbm_scrub(gbm_manager *mg, size_t nb, const uint8_t *hashes, uint8_t *bad_out)
{
	if (!mg || (nb && (!hashes || !bad_out)))
		return fail(GBM_E_INVALID_ARG, "NULL argument");
	try {
		std::vector<Hash> hs(nb);
		for (size_t b = 0; b < nb; ++b)
			hs[b].assign((const char *)hashes + 32 * b, 32);
		std::vector<Gathered> g;
		int grc = gather_many(mg, hs, nullptr, mg->n, g);
		if (grc)
			return grc;
		std::map<size_t, std::vector<size_t>> by_len;
		for (size_t b = 0; b < nb; ++b) {
			bad_out[b] = g[b].count == mg->n ? 0 : 1;
			if (!bad_out[b])
				by_len[g[b].meta.shard_len].push_back(b);
		}
		for (auto &kv : by_len) {
			const std::vector<size_t> &ids = kv.second;
			std::vector<const uint8_t *> sp(ids.size() * mg->n);
			for (size_t i = 0; i < ids.size(); ++i)
				for (int j = 0; j < mg->n; ++j)
					sp[i * mg->n + j] = g[ids[i]].shard[j].data();
			std::vector<uint8_t> ok(ids.size());
			int rc = gec_verify_batch(mg->codec, ids.size(), sp.data(), kv.first, ok.data());
			if (rc)
				return ec_fail(rc, "gec_verify_batch");
			for (size_t i = 0; i < ids.size(); ++i)
				bad_out[ids[i]] = ok[i] ? 0 : 1;
		}
	} catch (const std::exception &e) {
		return fail(GBM_E_IO, std::string("scrub: ") + e.what());
	}
	return GBM_OK;
}

// RepairWorker (src/block/repair.rs:30-150): phase 1 queues every hash of the refcount table, phase 2 every hash that
// is actually stored somewhere ("blocks we are storing but don't actually need").
// every hash any reachable node holds a shard of: the nodes are walked side by side (a directory node's walk is one
// opendir per prefix directory -- 16 nodes x hundreds of directories, tens of milliseconds when done one after the other)
static void list_all_nodes(gbm_manager *mg, std::set<Hash> &all)
{
	std::vector<std::set<Hash>> per(mg->nodes.size());
	mg->pool->parallel_for(mg->nodes.size(), [&](size_t i) {
		if (!mg->nodes[i]->down.load())
			mg->nodes[i]->list(per[i]);
	});
	for (auto &s : per)
		all.insert(s.begin(), s.end());
}

int gbm_repair_all(gbm_manager *mg, size_t *queued)
{
	if (!mg)
		return fail(GBM_E_INVALID_ARG, "NULL manager");
	std::set<Hash> all;
	for (auto &st : mg->rc) {
		std::lock_guard<std::mutex> g(st.mu);
		for (auto &kv : st.map)
			all.insert(kv.first);
	}
	list_all_nodes(mg, all);
	for (const Hash &h : all)
		mg->put_to_resync(h, 0);
	if (queued)
		*queued = all.size();
	return GBM_OK;
}

// Which single shard of an RS-inconsistent stripe is the wrong one?  For every candidate j the stripe is re-derived
// from the first k of the OTHER shards; the candidate is the culprit iff all the others then agree with what is
// stored (needs m >= 2).  One gec_reconstruct_batch call: the n candidates are n "blocks" with n erasure patterns.
static int locate_bad_shard(gbm_manager *mg, const Gathered &g)
{
	const int n = mg->n, k = mg->k;
	if (mg->m < 2)
		return -1;
	const size_t S = g.meta.shard_len;
	std::vector<const uint8_t *> sp((size_t)n * n, nullptr);
	std::vector<uint8_t *> op((size_t)n * n, nullptr);
	std::vector<std::vector<Bytes>> outb(n, std::vector<Bytes>(n));
	for (int c = 0; c < n; ++c) {
		// candidate c erased; of the rest the first k are read, the others are rebuilt and compared
		int used = 0;
		for (int j = 0; j < n; ++j) {
			if (j == c)
				continue;
			if (used < k) {
				sp[(size_t)c * n + j] = g.shard[j].data();
				++used;
			} else {
				outb[c][j] = mg->bufs->get(S);
				op[(size_t)c * n + j] = outb[c][j].mut();
			}
		}
	}
	if (gec_reconstruct_batch(mg->codec, n, sp.data(), op.data(), S, 0) != GEC_OK)
		return -1;
	int culprit = -1;
	for (int c = 0; c < n; ++c) {
		bool agree = true;
		for (int j = 0; j < n && agree; ++j)
			if (!outb[c][j].empty())
				agree = std::memcmp(outb[c][j].data(), g.shard[j].data(), S) == 0;
		if (agree) {
			if (culprit >= 0)
				return -1;  // ambiguous: more than one shard is wrong
			culprit = c;
		}
	}
	return culprit;
}

// ScrubWorker (src/block/repair.rs:234-500): walk everything that is stored, batch by batch, verify on the device;
// a corrupt block is counted and queued for resync.  Where the reference can only say "this file no longer matches
// its name", the code can say WHICH shard of an inconsistent stripe is wrong (if only one is): that shard is set
// aside as *.corrupted, and resync rebuilds it.
// stats (may be NULL): [0] blocks scrubbed, [1] corruptions detected, [2] device verify calls, [3] shards located and set aside
int gbm_scrub_all(gbm_manager *mg, size_t batch_blocks, uint64_t stats[4])
{
	if (!mg)
		return fail(GBM_E_INVALID_ARG, "NULL manager");
	if (batch_blocks == 0)
		batch_blocks = 1024;
	uint64_t st[4] = {0, 0, 0, 0};
	try {
		std::set<Hash> all;
		Trace tr("scrub");
		list_all_nodes(mg, all);
		tr.lap("list");
		std::vector<Hash> hs(all.begin(), all.end());
		// the next batch's shards are read from the nodes while the current batch is on the device
		struct Batch {
			std::vector<Hash> batch;
			std::vector<Gathered> g;
			int rc = GBM_OK;
			std::string err;
		};
		auto read_batch = [&](size_t b0) {
			Batch bt;
			const size_t nb = std::min(batch_blocks, hs.size() - b0);
			bt.batch.assign(hs.begin() + b0, hs.begin() + b0 + nb);
			// shards are accepted on their headers; their checksums come back from the same device trip that checks
			// the stripe against the code (every byte crosses the link once)
			try {
				bt.rc = gather_many(mg, bt.batch, nullptr, mg->n, bt.g, /*verify=*/false);
				if (bt.rc)
					bt.err = g_err;  // thread-local: carried to the caller's thread
			} catch (const std::exception &e) {
				bt.rc = GBM_E_IO;
				bt.err = e.what();
			}
			return bt;
		};
		std::future<Batch> next;
		if (!hs.empty())
			next = std::async(std::launch::async, read_batch, (size_t)0);
		for (size_t b0 = 0; b0 < hs.size(); b0 += batch_blocks) {
			Batch cur = next.get();
			tr.lap("wait for the batch's shards");
			if (b0 + batch_blocks < hs.size())
				next = std::async(std::launch::async, read_batch, b0 + batch_blocks);
			if (cur.rc) {
				if (next.valid())
					next.wait();
				return fail(cur.rc, cur.err);
			}
			const size_t nb = cur.batch.size();
			std::vector<Hash> &batch = cur.batch;
			std::vector<Gathered> &g = cur.g;
			std::map<size_t, std::vector<size_t>> by_len;
			auto unreadable = [&](size_t b) {
				if (mg->get_rc(batch[b]).is_nonzero()) {
					++st[1];  // a needed block that is not fully readable
					mg->put_to_resync(batch[b], 0);
				}
			};
			for (size_t b = 0; b < nb; ++b) {
				++st[0];
				if (g[b].count == mg->n)
					by_len[g[b].meta.shard_len].push_back(b);
				else
					unreadable(b);
			}
			for (auto &kv : by_len) {
				const std::vector<size_t> &ids = kv.second;
				std::vector<const uint8_t *> sp(ids.size() * mg->n);
				for (size_t i = 0; i < ids.size(); ++i)
					for (int j = 0; j < mg->n; ++j)
						sp[i * mg->n + j] = g[ids[i]].shard[j].data();
				std::vector<uint8_t> ok(ids.size()), sums(ids.size() * (size_t)mg->n * 32);
				int rc = gec_verify_hash_batch(mg->codec, ids.size(), sp.data(), kv.first, ok.data(), sums.data());
				++st[2];
				tr.lap("verify + checksums");
				if (rc)
					return ec_fail(rc, "gec_verify_hash_batch");
				mg->gpu_hashed += ids.size() * (size_t)mg->n;
				for (size_t i = 0; i < ids.size(); ++i) {
					// a shard that does not match the checksum in its own header: read_block_from's corrupt-file
					// case (manager.rs:577-609) -- set aside, queued; the stripe's verdict follows from it
					bool sum_bad = false;
					for (int j = 0; j < mg->n; ++j) {
						const Gathered &gb = g[ids[i]];
						if (std::memcmp(sums.data() + (i * mg->n + j) * 32, gb.sum[j].data(), 32) != 0) {
							mg->metrics[2]++;
							if (gb.node[j] >= 0)
								mg->nodes[gb.node[j]]->mark_corrupted(batch[ids[i]], j);
							mg->put_to_resync(batch[ids[i]], 0);
							sum_bad = true;
						}
					}
					if (sum_bad) {
						unreadable(ids[i]);
						continue;
					}
					if (ok[i])
						continue;
					++st[1];
					mg->metrics[2]++;
					const Gathered &gb = g[ids[i]];
					const int bad = locate_bad_shard(mg, gb);
					if (bad >= 0 && gb.node[bad] >= 0) {
						mg->nodes[gb.node[bad]]->mark_corrupted(batch[ids[i]], bad);
						++st[3];
					}
					mg->put_to_resync(batch[ids[i]], 0);
				}
			}
		}
	} catch (const std::exception &e) {
		return fail(GBM_E_IO, std::string("scrub_all: ") + e.what());
	}
	mg->scrub_corruptions += st[1];
	mg->scrub_last_complete_ms = mg->now();
	if (stats)
		std::copy(st, st + 4, stats);
	return GBM_OK;
}

int gbm_scrub_state(const gbm_manager *m, uint64_t out[2])
{
	if (!m || !out)
		return fail(GBM_E_INVALID_ARG, "NULL argument");
	out[0] = m->scrub_corruptions.load();
	out[1] = m->scrub_last_complete_ms.load();
	return GBM_OK;
}

int gbm_node_set_down(gbm_manager *m, int node, int down)
{
	if (!m || node < 0 || node >= (int)m->nodes.size())
		return fail(GBM_E_INVALID_ARG, "bad node index");
	m->nodes[node]->down = down != 0;
	return GBM_OK;
}

int gbm_node_has_shard(gbm_manager *m, int node, const uint8_t hash[32], int idx)
{
	if (!m || node < 0 || node >= (int)m->nodes.size())
		return 0;
	return m->nodes[node]->has(Hash((const char *)hash, 32), idx) ? 1 : 0;
}

int gbm_node_delete_shard(gbm_manager *m, int node, const uint8_t hash[32], int idx)
{
	if (!m || node < 0 || node >= (int)m->nodes.size())
		return fail(GBM_E_INVALID_ARG, "bad node index");
	m->nodes[node]->del(Hash((const char *)hash, 32), idx);
	return GBM_OK;
}

int gbm_node_corrupt_shard(gbm_manager *m, int node, const uint8_t hash[32], int idx, size_t offset, uint8_t mask,
			   int fix_checksum)
{
	if (!m || node < 0 || node >= (int)m->nodes.size())
		return fail(GBM_E_INVALID_ARG, "bad node index");
	Hash h((const char *)hash, 32);
	Shard s;
	if (!m->nodes[node]->get(h, idx, s) || s.data.n <= offset)
		return fail(GBM_E_IO, "no such shard / offset");
	try {
		// shard buffers are shared (a data shard is a slice of its block's buffer): corrupt a private copy
		Bytes copy = m->bufs->get(s.data.n);
		std::memcpy(copy.mut(), s.data.data(), s.data.n);
		copy.mut()[offset] ^= mask;
		s.data = copy;
	} catch (const std::bad_alloc &) {
		return fail(GBM_E_IO, "out of memory");
	}
	if (fix_checksum)
		shardsum(s.data.data(), s.data.n, s.hd.checksum);
	return m->nodes[node]->put(h, idx, s) ? GBM_OK : fail(GBM_E_IO, "rewrite failed");
}

int gbm_node_shard_header(gbm_manager *m, int node, const uint8_t hash[32], int idx, uint8_t out[GBM_SHARD_HEADER_SIZE])
{
	if (!m || node < 0 || node >= (int)m->nodes.size() || !hash || !out)
		return fail(GBM_E_INVALID_ARG, "bad argument");
	Shard s;
	if (!m->nodes[node]->get(Hash((const char *)hash, 32), idx, s))
		return fail(GBM_E_IO, "no such shard");
	s.hd.pack(out);
	return GBM_OK;
}

uint64_t gbm_node_order_violations(gbm_manager *m, int node)
{
	if (!m || node < 0 || node >= (int)m->nodes.size())
		return 0;
	return m->nodes[node]->order_violations.load();
}

int gbm_set_compression_level(gbm_manager *m, int enabled, int level)
{
	if (!m)
		return fail(GBM_E_INVALID_ARG, "NULL manager");
	if (enabled && !zstd().ok)
		return fail(GBM_E_IO, "libzstd.so.1 not available");
	m->compression_level = level;
	m->compress = enabled != 0;
	return GBM_OK;
}

// ------------------------------------------------------------------ batcher
// The coalescing queue in front of the FFI.  Garage keeps <= 3 block puts in flight
// per PutObject (PUT_BLOCKS_MAX_PARALLEL, src/api/s3/put.rs:42,486-511) and serves
// many requests at once; each caller blocks in gbm_batcher_put_block (the way
// `rpc_put_block(...).await` suspends) while ONE worker thread turns whatever has
// queued up within max_wait_us (or max_blocks) into a single device batch.
struct gbm_batcher {
	struct Item {
		const uint8_t *hash, *data;
		size_t len;
		uint8_t prevent_compression = 0;
		bool has_tag = false;
		gbm_order_tag tag{0, 0};
		int rc = GBM_OK;
		bool done = false;
	};
	gbm_manager *mg = nullptr;
	size_t max_blocks = 64;
	unsigned max_wait_us = 200;
	// buffer_kb_semaphore (src/block/manager.rs:96,156,380-384): KiB permits for the bytes of blocks on
	// their way to the storage nodes, Config.block_ram_buffer_max (default 256 MiB, src/util/config.rs:276-278)
	size_t ram_permits_kb = 256 * 1024, ram_in_use_kb = 0;
	std::mutex mu;
	std::condition_variable cv_work, cv_done, cv_ram;
	std::deque<Item *> queue;
	bool stop = false, forming = false;
	uint64_t batches = 0, blocks = 0, max_batch = 0;
	// two workers: while one batch is on the device the next one forms and starts (the device trip has a latency
	// floor -- the checksum chain -- that a single worker would pay serially)
	std::vector<std::thread> workers;

	void run()
	{
		std::unique_lock<std::mutex> lk(mu);
		for (;;) {
			// one worker forms a batch at a time; the other one is either on the device or waits its turn
			cv_work.wait(lk, [&] { return stop || (!queue.empty() && !forming); });
			if (queue.empty()) {
				if (stop)
					return;
				continue;
			}
			forming = true;
			// linger a little so concurrent callers land in the same batch
			// system_clock: libstdc++ maps it to pthread_cond_timedwait, which ThreadSanitizer
			// understands (steady_clock -> pthread_cond_clockwait is not intercepted by gcc 11's
			// TSan and floods the report with false "double lock" findings)
			// The linger ends early once arrivals stop: callers come in bursts (the <= 3 parallel puts of a PutObject,
			// or everybody at once when a batch completes), and waiting out the full linger after the burst is pure
			// latency -- 3 callers: 0.80 -> 0.55 ms per put.
			const auto deadline = std::chrono::system_clock::now() + std::chrono::microseconds(max_wait_us);
			const auto gap = std::chrono::microseconds(std::max(20u, max_wait_us / 6));
			size_t seen = queue.size();
			while (!stop && queue.size() < max_blocks) {
				const auto now = std::chrono::system_clock::now();
				if (now >= deadline)
					break;
				if (cv_work.wait_until(lk, std::min(deadline, now + gap)) == std::cv_status::timeout && queue.size() == seen)
					break;  // nobody arrived during the gap
				seen = queue.size();
			}
			std::vector<Item *> batch;
			while (!queue.empty() && batch.size() < max_blocks) {
				batch.push_back(queue.front());
				queue.pop_front();
			}
			forming = false;
			cv_work.notify_all();
			lk.unlock();
			const size_t nb = batch.size();
			std::vector<uint8_t> hashes(nb * 32), pc(nb);
			std::vector<const uint8_t *> data(nb);
			std::vector<size_t> lens(nb);
			std::vector<gbm_order_tag> tags(nb);
			std::vector<int> rcs(nb, GBM_OK);
			bool any_tag = false;
			static const uint8_t kEmpty = 0;  // a zero-length block may come with a NULL pointer
			for (size_t i = 0; i < nb; ++i) {
				std::memcpy(hashes.data() + 32 * i, batch[i]->hash, 32);
				data[i] = batch[i]->data ? batch[i]->data : &kEmpty;
				lens[i] = batch[i]->len;
				pc[i] = batch[i]->prevent_compression;
				// untagged blocks sort after tagged ones of the same batch; their relative order is free
				tags[i] = batch[i]->has_tag ? batch[i]->tag : gbm_order_tag{~0ull, i};
				any_tag = any_tag || batch[i]->has_tag;
			}
			int rc;
			try {
				rc = put_blocks_impl(mg, nb, hashes.data(), data.data(), lens.data(), pc.data(), any_tag ? tags.data() : nullptr,
						     rcs.data());
			} catch (const std::exception &) {
				rc = GBM_E_IO;
				std::fill(rcs.begin(), rcs.end(), GBM_E_IO);
			}
			(void)rc;  // per-block results are in rcs (a whole-batch failure marks every block that was not stored)
			lk.lock();
			for (size_t i = 0; i < nb; ++i) {
				batch[i]->rc = rcs[i];
				batch[i]->done = true;
				ram_in_use_kb -= batch[i]->len / 1024;  // the permit is dropped once all sends finished
			}
			cv_ram.notify_all();
			++batches;
			blocks += nb;
			max_batch = std::max<uint64_t>(max_batch, nb);
			cv_done.notify_all();
		}
	}
};

int gbm_batcher_create(gbm_manager *m, size_t max_blocks, unsigned max_wait_us, gbm_batcher **out)
{
	if (!m || !out || max_blocks == 0)
		return fail(GBM_E_INVALID_ARG, "bad batcher arguments");
	auto *b = new gbm_batcher();
	b->mg = m;
	b->max_blocks = max_blocks;
	b->max_wait_us = max_wait_us;
	static const int nworkers = [] {
		const char *e = std::getenv("GBM_BATCHER_WORKERS");
		const int v = e ? std::atoi(e) : 0;
		return v >= 1 && v <= 16 ? v : 2;
	}();
	for (int i = 0; i < nworkers; ++i)
		b->workers.emplace_back([b] { b->run(); });
	*out = b;
	return GBM_OK;
}

void gbm_batcher_destroy(gbm_batcher *b)
{
	if (!b)
		return;
	{
		std::lock_guard<std::mutex> g(b->mu);
		b->stop = true;
	}
	b->cv_work.notify_all();
	b->cv_ram.notify_all();
	for (auto &t : b->workers)
		t.join();
	delete b;
}

int gbm_batcher_put_block(gbm_batcher *b, const uint8_t hash[32], const uint8_t *data, size_t len, int prevent_compression,
			  const gbm_order_tag *order_tag)
{
	if (!b || !hash || (!data && len))
		return fail(GBM_E_INVALID_ARG, "NULL argument");
	gbm_batcher::Item it;
	it.hash = hash;
	it.data = data;
	it.len = len;
	it.prevent_compression = prevent_compression ? 1 : 0;
	if (order_tag) {
		it.has_tag = true;
		it.tag = *order_tag;
	}
	std::unique_lock<std::mutex> lk(b->mu);
	// acquire len/1024 permits; a block larger than the whole budget could never be sent (Garage's
	// acquire_many would wait forever): refuse it instead
	const size_t need_kb = len / 1024;
	if (need_kb > b->ram_permits_kb)
		return fail(GBM_E_INVALID_ARG, "could not reserve space for buffer of data to send to remote nodes");
	b->cv_ram.wait(lk, [&] { return b->stop || b->ram_in_use_kb + need_kb <= b->ram_permits_kb; });
	if (b->stop)
		return fail(GBM_E_INVALID_ARG, "batcher is shutting down");
	b->ram_in_use_kb += need_kb;
	b->queue.push_back(&it);
	b->cv_work.notify_all();
	b->cv_done.wait(lk, [&] { return it.done; });
	if (it.rc == GBM_E_QUORUM)
		return fail(it.rc, "Could not reach quorum");
	if (it.rc != GBM_OK)
		return fail(it.rc, "device batch failed");
	return GBM_OK;
}

int gbm_batcher_set_ram_buffer_max(gbm_batcher *b, size_t bytes)
{
	if (!b || bytes < 1024)
		return fail(GBM_E_INVALID_ARG, "bad ram buffer size");
	std::lock_guard<std::mutex> g(b->mu);
	b->ram_permits_kb = bytes / 1024;
	b->cv_ram.notify_all();
	return GBM_OK;
}

int gbm_batcher_stats(gbm_batcher *b, uint64_t out[3])
{
	if (!b || !out)
		return fail(GBM_E_INVALID_ARG, "NULL argument");
	std::lock_guard<std::mutex> g(b->mu);
	out[0] = b->batches;
	out[1] = b->blocks;
	out[2] = b->max_batch;
	return GBM_OK;
}

uint64_t gbm_gpu_hashed(const gbm_manager *m) { return m ? m->gpu_hashed.load() : 0; }

int gbm_metrics(const gbm_manager *m, uint64_t out[6])
{
	if (!m || !out)
		return fail(GBM_E_INVALID_ARG, "NULL argument");
	for (int i = 0; i < 6; ++i)
		out[i] = m->metrics[i].load();
	return GBM_OK;
}

}  // extern "C"
