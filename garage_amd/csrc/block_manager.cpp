// block_manager.cpp -- libgarage_block.so (include/garage_block.h): C++ host-side
// mirror of garage_block::BlockManager with erasure-coded shard fan-out.
//
// Pure host code: it moves buffers between "nodes" and calls libgarage_ec's C ABI
// (gec_encode_hash_batch / gec_reconstruct_batch / gec_verify_batch /
// gec_blake2sum_batch) for every shard byte and every large hash batch it needs
// computed.  No GF arithmetic happens here.
#include "../../include/garage_block.h"

#include <dlfcn.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <thread>
#include <cerrno>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
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

void b2_compress(uint64_t h[8], const uint8_t block[128], uint64_t t, bool last)
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

void blake2sum(const uint8_t *data, size_t len, uint8_t out[32])
{
	uint64_t h[8];
	for (int i = 0; i < 8; ++i)
		h[i] = B2_IV[i];
	h[0] ^= 0x01010000ULL ^ 64;  // digest length 64, no key, fanout 1, depth 1
	size_t off = 0;
	while (len - off > 128) {
		b2_compress(h, data + off, off + 128, false);
		off += 128;
	}
	uint8_t last[128] = {0};
	if (len > off)  // data may be NULL for the empty message
		std::memcpy(last, data + off, len - off);
	b2_compress(h, last, len, true);
	std::memcpy(out, h, 32);
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
	// verifies the frame checksum; false = corrupt
	bool decode(const uint8_t *data, size_t len, std::vector<uint8_t> &out) const
	{
		if (!ok)
			return false;
		unsigned long long sz = getFrameContentSize(data, len);
		if (sz >= (1ull << 40))  // CONTENTSIZE_ERROR / UNKNOWN are huge sentinels
			return false;
		out.resize((size_t)sz);
		uint8_t dummy;
		size_t n = decompress(sz ? out.data() : &dummy, (size_t)sz, data, len);
		return !isError(n) && n == sz;
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
		out[4] = 1;
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
		if (n < GBM_SHARD_HEADER_SIZE || std::memcmp(in, "GECS", 4) != 0 || in[4] != 1)
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

// -------------------------------------------------------------------- nodes
struct Node {
	std::atomic<bool> down{false};  // flipped by gbm_node_set_down while the batcher thread may be fanning out
	virtual ~Node() = default;
	virtual bool put(const Hash &h, int idx, std::vector<uint8_t> &&raw) = 0;
	virtual bool get(const Hash &h, int idx, std::vector<uint8_t> &raw) = 0;  // false: absent
	virtual void del(const Hash &h, int idx) = 0;
	virtual void mark_corrupted(const Hash &h, int idx) { del(h, idx); }
};

struct MemoryNode : Node {
	std::mutex mu;  // puts (batcher thread) and gets (caller threads) may overlap
	std::map<std::pair<Hash, int>, std::vector<uint8_t>> files;
	bool put(const Hash &h, int idx, std::vector<uint8_t> &&raw) override
	{
		std::lock_guard<std::mutex> g(mu);
		files[{h, idx}] = std::move(raw);
		return true;
	}
	bool get(const Hash &h, int idx, std::vector<uint8_t> &raw) override
	{
		std::lock_guard<std::mutex> g(mu);
		auto it = files.find({h, idx});
		if (it == files.end())
			return false;
		raw = it->second;
		return true;
	}
	void del(const Hash &h, int idx) override
	{
		std::lock_guard<std::mutex> g(mu);
		files.erase({h, idx});
	}
};

// <root>/<h0>/<h1>/<hex>.s<idx>, tmp file + rename (write_block_inner, manager.rs:720-805);
// a corrupt shard is renamed *.corrupted (manager.rs:807-819).
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
	bool put(const Hash &h, int idx, std::vector<uint8_t> &&raw) override
	{
		mkdirs(dir(h));
		static std::atomic<uint64_t> seq{0};  // unique per writer: two threads may store the same shard
		std::string p = path(h, idx), tmp = p + ".tmp" + std::to_string(::getpid()) + "_" + std::to_string(seq++);
		FILE *f = std::fopen(tmp.c_str(), "wb");
		if (!f)
			return false;
		bool ok = std::fwrite(raw.data(), 1, raw.size(), f) == raw.size();
		const bool sync = fsync_data.load();
		if (ok && sync)  // file first, then (after the rename) its directory: manager.rs:775-800
			ok = std::fflush(f) == 0 && ::fsync(::fileno(f)) == 0;
		ok = (std::fclose(f) == 0) && ok;
		if (ok)
			ok = std::rename(tmp.c_str(), p.c_str()) == 0;
		if (!ok)
			std::remove(tmp.c_str());
		if (ok && sync) {
			int dfd = ::open(dir(h).c_str(), O_RDONLY | O_DIRECTORY);
			if (dfd >= 0) {
				ok = ::fsync(dfd) == 0;
				::close(dfd);
			} else {
				ok = false;
			}
		}
		return ok;
	}
	bool get(const Hash &h, int idx, std::vector<uint8_t> &raw) override
	{
		FILE *f = std::fopen(path(h, idx).c_str(), "rb");
		if (!f)
			return false;
		std::fseek(f, 0, SEEK_END);
		long n = std::ftell(f);
		std::fseek(f, 0, SEEK_SET);
		raw.resize(n > 0 ? (size_t)n : 0);
		bool ok = n >= 0 && std::fread(raw.data(), 1, raw.size(), f) == raw.size();
		std::fclose(f);
		return ok;
	}
	void del(const Hash &h, int idx) override { std::remove(path(h, idx).c_str()); }
	void mark_corrupted(const Hash &h, int idx) override
	{
		std::string p = path(h, idx);
		std::rename(p.c_str(), (p + ".corrupted").c_str());
	}
};

}  // namespace

struct gbm_manager {
	const gec_codec *codec = nullptr;
	int k = 0, m = 0, n = 0, write_quorum = 0;
	std::vector<std::unique_ptr<Node>> nodes;
	std::mutex mu;  // rc / resync queue / metrics (lock_mutate's role, manager.rs:679-689)
	std::unordered_map<Hash, uint64_t> rc;
	std::vector<Hash> resync_queue;
	uint64_t metrics[6] = {0, 0, 0, 0, 0, 0};
	uint64_t gpu_hashed = 0;  // messages hashed on the device
	bool compress = false;    // Config.compression_level (src/util/config.rs:52-58); Garage's default is Some(1)
	int compression_level = 1;

	void nodes_of(const Hash &h, std::vector<int> &who) const
	{
		// partition = top byte(s) of the hash (src/rpc/layout/version.rs:101-104)
		size_t start = ((unsigned char)h[0] * 31u + (unsigned char)h[1]) % nodes.size();
		who.resize(n);
		for (int j = 0; j < n; ++j)
			who[j] = (int)((start + j) % nodes.size());
	}
	void enqueue(const Hash &h)
	{
		std::lock_guard<std::mutex> g(mu);
		resync_queue.push_back(h);
	}
};

namespace {

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
	std::vector<std::vector<uint8_t>> shard;  // n entries; empty = not in hand
	ShardHeader meta;
	bool have_meta = false;
	int count = 0;
	int next = 0;  // next shard index to try
	struct Group {
		ShardHeader meta;
		std::vector<std::vector<uint8_t>> shard;
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
};

// blake2sum of many buffers: on the GPU (gec_blake2sum_batch) once the batch is big
// enough to beat one CPU thread (~1 GiB/s) through PCIe + the kernel's ~3.5 ms chain
// latency, else inline.  SURVEY.md section 8 row f4.
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
		int rc = gec_blake2sum_batch(mg->codec, ptrs.size(), ptrs.data(), lens.data(), sums.data());
		if (rc)
			return fail(GBM_E_EC, std::string("gec_blake2sum_batch: ") + gec_strerror(rc) + " (" + gec_last_error() + ")");
		std::lock_guard<std::mutex> lk(mg->mu);
		mg->gpu_hashed += ptrs.size();
		return GBM_OK;
	}
	for (size_t i = 0; i < ptrs.size(); ++i)
		blake2sum(ptrs[i], lens[i], sums.data() + 32 * i);
	return GBM_OK;
}

// Fetch shards in node order until every block has `want` valid ones in hand (or ran
// out of nodes).  Checksums of each round's candidates are verified in ONE batch; a
// shard whose checksum or geometry does not match is treated as missing, renamed
// *.corrupted and queued for resync (read_block_from's behaviour, manager.rs:577-609),
// and the next node is tried in the following round.
int gather_many(gbm_manager *mg, const std::vector<Hash> &hs, int want, std::vector<Gathered> &gs)
{
	const int n = mg->n;
	gs.assign(hs.size(), Gathered());
	std::vector<std::vector<int>> who(hs.size());
	for (size_t b = 0; b < hs.size(); ++b) {
		mg->nodes_of(hs[b], who[b]);
		gs[b].shard.assign(n, {});
	}
	struct Cand {
		size_t b;
		int j;
		ShardHeader hd;
		std::vector<uint8_t> raw;
	};
	for (;;) {
		std::vector<Cand> cands;
		for (size_t b = 0; b < hs.size(); ++b) {
			Gathered &g = gs[b];
			int pending = 0;
			while (g.next < n && g.best() + pending < want) {
				const int j = g.next++;
				Node &nd = *mg->nodes[who[b][j]];
				if (nd.down)
					continue;
				Cand c{b, j, ShardHeader(), {}};
				if (!nd.get(hs[b], j, c.raw))
					continue;
				const bool ok = c.hd.unpack(c.raw.data(), c.raw.size()) && c.hd.idx == j && c.hd.k == mg->k &&
						c.hd.m == mg->m && c.raw.size() == (size_t)GBM_SHARD_HEADER_SIZE + c.hd.shard_len;
				if (!ok) {
					{
						std::lock_guard<std::mutex> lk(mg->mu);
						mg->metrics[2]++;
						mg->resync_queue.push_back(hs[b]);
					}
					nd.mark_corrupted(hs[b], j);
					continue;
				}
				cands.push_back(std::move(c));
				++pending;
			}
		}
		if (cands.empty())
			break;
		std::vector<const uint8_t *> ptrs(cands.size());
		std::vector<size_t> lens(cands.size());
		for (size_t i = 0; i < cands.size(); ++i) {
			ptrs[i] = cands[i].raw.data() + GBM_SHARD_HEADER_SIZE;
			lens[i] = cands[i].hd.shard_len;
		}
		std::vector<uint8_t> sums;
		int rc = hash_many(mg, ptrs, lens, sums);
		if (rc)
			return rc;
		for (size_t i = 0; i < cands.size(); ++i) {
			Cand &c = cands[i];
			Gathered &g = gs[c.b];
			if (std::memcmp(sums.data() + 32 * i, c.hd.checksum, 32) != 0) {
				{
					std::lock_guard<std::mutex> lk(mg->mu);
					mg->metrics[2]++;
					mg->resync_queue.push_back(hs[c.b]);
				}
				mg->nodes[who[c.b][c.j]]->mark_corrupted(hs[c.b], c.j);
				continue;
			}
			Geometry geo;
			geo.compressed = c.hd.compressed;
			geo.orig_len = c.hd.orig_len;
			geo.shard_len = c.hd.shard_len;
			Gathered::Group &grp = g.groups[geo];
			if (grp.shard.empty()) {
				grp.shard.assign(n, {});
				grp.meta = c.hd;
			}
			c.raw.erase(c.raw.begin(), c.raw.begin() + GBM_SHARD_HEADER_SIZE);
			grp.shard[c.j] = std::move(c.raw);
			grp.count++;
			std::lock_guard<std::mutex> lk(mg->mu);
			mg->metrics[1] += c.hd.shard_len;
		}
	}
	// settle on the largest consistent group; the stragglers of other geometries are
	// stale leftovers that resync will overwrite
	for (size_t b = 0; b < hs.size(); ++b) {
		Gathered &g = gs[b];
		Gathered::Group *bestg = nullptr;
		for (auto &kv : g.groups)
			if (!bestg || kv.second.count > bestg->count)
				bestg = &kv.second;
		if (bestg) {
			g.shard = std::move(bestg->shard);
			g.meta = bestg->meta;
			g.have_meta = true;
			g.count = bestg->count;
			if (g.groups.size() > 1) {
				std::lock_guard<std::mutex> lk(mg->mu);
				mg->resync_queue.push_back(hs[b]);
			}
		}
		g.groups.clear();
	}
	return GBM_OK;
}

int gather(gbm_manager *mg, const Hash &h, int want, Gathered &g)
{
	std::vector<Gathered> gs;
	int rc = gather_many(mg, {h}, want, gs);
	g = std::move(gs[0]);
	return rc;
}

int store_shard(gbm_manager *mg, int node, const Hash &h, int idx, const uint8_t *payload, size_t S,
		uint64_t orig_len, bool compressed, const uint8_t *checksum = nullptr)
{
	Node &nd = *mg->nodes[node];
	if (nd.down)
		return -1;
	ShardHeader hd;
	hd.k = (uint8_t)mg->k;
	hd.m = (uint8_t)mg->m;
	hd.idx = (uint8_t)idx;
	hd.compressed = compressed ? 1 : 0;
	hd.orig_len = orig_len;
	hd.shard_len = (uint32_t)S;
	if (checksum)
		std::memcpy(hd.checksum, checksum, 32);
	else
		blake2sum(payload, S, hd.checksum);
	std::vector<uint8_t> raw(GBM_SHARD_HEADER_SIZE + S);
	hd.pack(raw.data());
	std::memcpy(raw.data() + GBM_SHARD_HEADER_SIZE, payload, S);
	return nd.put(h, idx, std::move(raw)) ? 0 : -1;
}

int ec_fail(int rc, const char *what)
{
	return fail(GBM_E_EC, std::string(what) + ": " + gec_strerror(rc) + " (" + gec_last_error() + ")");
}

}  // namespace

extern "C" {

const char *gbm_last_error(void) { return g_err.c_str(); }

void gbm_blake2sum(const uint8_t *data, size_t len, uint8_t out[32]) { blake2sum(data, len, out); }

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
	}
	*out = mg.release();
	return GBM_OK;
}

void gbm_destroy(gbm_manager *m) { delete m; }

int gbm_set_data_fsync(gbm_manager *m, int enabled)
{
	if (!m)
		return fail(GBM_E_INVALID_ARG, "NULL manager");
	for (auto &nd : m->nodes)
		if (DirNode *d = dynamic_cast<DirNode *>(nd.get()))
			d->fsync_data = enabled != 0;
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

// rcs (optional): per-block result, GBM_OK or GBM_E_QUORUM; the return value is the last
// failure (or a whole-batch error such as GBM_E_EC).
static int put_blocks_impl(gbm_manager *mg, size_t nb, const uint8_t *hashes, const uint8_t *const *data,
			   const size_t *len, int *rcs)
{
	if (!mg || (nb && (!hashes || !data || !len)))
		return fail(GBM_E_INVALID_ARG, "NULL argument");
	if (nb == 0)
		return GBM_OK;
	if (rcs)
		std::fill(rcs, rcs + nb, GBM_OK);
	const int k = mg->k, m = mg->m, n = mg->n;
	// DataBlock::from_buffer: zstd when a level is configured, Plain on any encoder error
	std::vector<std::vector<uint8_t>> zbuf(nb);
	std::vector<const uint8_t *> dptr(data, data + nb);
	std::vector<size_t> dlen(len, len + nb);
	std::vector<uint8_t> is_z(nb, 0);
	if (mg->compress)
		for (size_t b = 0; b < nb; ++b)
			if (zstd().encode(data[b], len[b], mg->compression_level, zbuf[b])) {
				dptr[b] = zbuf[b].data();
				dlen[b] = zbuf[b].size();
				is_z[b] = 1;
			}
	data = dptr.data();
	len = dlen.data();
	// Shard geometry is a pure function of the block: S = gec_shard_len(k, payload length).
	// (Never the batch maximum: a later put of the same block must produce compatible
	// shards.)  Blocks of equal S -- in practice all full block_size blocks -- share ONE
	// device call that returns parity and the checksums of all k+m shards.
	std::map<size_t, std::vector<size_t>> by_s;
	for (size_t b = 0; b < nb; ++b)
		by_s[gec_shard_len(k, len[b])].push_back(b);
	int result = GBM_OK;
	for (auto &kv : by_s) {
		const size_t S = kv.first;
		const std::vector<size_t> &ids = kv.second;
		const size_t gn = ids.size();
		std::vector<uint8_t> parity(gn * (size_t)m * S), sums(gn * (size_t)n * 32), last(S);
		std::vector<uint8_t *> pp(gn);
		std::vector<const uint8_t *> gd(gn);
		std::vector<size_t> gl(gn);
		for (size_t i = 0; i < gn; ++i) {
			pp[i] = parity.data() + i * (size_t)m * S;
			gd[i] = data[ids[i]];
			gl[i] = len[ids[i]];
		}
		int rc = gec_encode_hash_batch(mg->codec, gn, gd.data(), gl.data(), S, pp.data(), sums.data());
		if (rc)
			return ec_fail(rc, "gec_encode_hash_batch");
		{
			std::lock_guard<std::mutex> lk(mg->mu);
			mg->gpu_hashed += gn * (size_t)n;
		}
		for (size_t i = 0; i < gn; ++i) {
			const size_t b = ids[i];
			Hash h((const char *)hashes + 32 * b, 32);
			std::vector<int> who;
			mg->nodes_of(h, who);
			int ok = 0;
			for (int j = 0; j < n; ++j) {
				const uint8_t *payload;
				if (j < k) {
					// data shard j = payload bytes [j*S, (j+1)*S) zero-extended
					size_t lo = (size_t)j * S, hi = std::min(len[b], lo + S);
					if (hi >= lo + S) {
						payload = data[b] + lo;
					} else {
						std::fill(last.begin(), last.end(), 0);
						if (hi > lo)
							std::memcpy(last.data(), data[b] + lo, hi - lo);
						payload = last.data();
					}
				} else {
					payload = pp[i] + (size_t)(j - k) * S;
				}
				if (store_shard(mg, who[j], h, j, payload, S, len[b], is_z[b] != 0, sums.data() + (i * n + j) * 32) == 0) {
					++ok;
					std::lock_guard<std::mutex> lk(mg->mu);
					mg->metrics[0] += S;
				}
			}
			{
				std::lock_guard<std::mutex> lk(mg->mu);
				mg->metrics[4]++;
			}
			if (ok < mg->write_quorum) {
				result = fail(GBM_E_QUORUM, "Could not reach quorum of " + std::to_string(mg->write_quorum) + ". " +
								    std::to_string(ok) + " of " + std::to_string(n) +
								    " request succeeded");
				if (rcs)
					rcs[b] = GBM_E_QUORUM;
			} else if (ok < n) {
				mg->enqueue(h);  // stragglers are finished by resync
			}
		}
	}
	return result;
}

int gbm_rpc_put_blocks(gbm_manager *mg, size_t nb, const uint8_t *hashes, const uint8_t *const *data,
		       const size_t *len)
{
	return put_blocks_impl(mg, nb, hashes, data, len, nullptr);
}

int gbm_rpc_put_block(gbm_manager *m, const uint8_t hash[32], const uint8_t *data, size_t len)
{
	const uint8_t *d[1] = {data};
	return gbm_rpc_put_blocks(m, 1, hash, d, &len);
}

int gbm_rpc_get_blocks(gbm_manager *mg, size_t nb, const uint8_t *hashes, uint8_t *const *out, const size_t *cap,
		       size_t *len_out, int *rcs)
{
	if (!mg || (nb && (!hashes || !out || !cap || !len_out || !rcs)))
		return fail(GBM_E_INVALID_ARG, "NULL argument");
	const int k = mg->k, n = mg->n;
	std::vector<Hash> hs(nb);
	for (size_t b = 0; b < nb; ++b)
		hs[b].assign((const char *)hashes + 32 * b, 32);
	std::vector<Gathered> g;
	int grc = gather_many(mg, hs, k, g);  // shard checksums verified in batches (GPU when large)
	if (grc)
		return grc;
	// blocks that need a decode, grouped by shard length (one device call per group)
	std::map<size_t, std::vector<size_t>> need;
	for (size_t b = 0; b < nb; ++b) {
		len_out[b] = 0;
		if (!g[b].have_meta || g[b].count < k) {
			rcs[b] = GBM_E_MISSING_BLOCK;
			continue;
		}
		rcs[b] = GBM_OK;
		len_out[b] = g[b].meta.orig_len;
		for (int j = 0; j < k; ++j)
			if (g[b].shard[j].empty()) {
				need[g[b].meta.shard_len].push_back(b);
				break;
			}
	}
	for (auto &kv : need) {
		const size_t S = kv.first;
		const std::vector<size_t> &ids = kv.second;
		std::vector<const uint8_t *> sp(ids.size() * n, nullptr);
		std::vector<uint8_t *> op(ids.size() * n, nullptr);
		for (size_t i = 0; i < ids.size(); ++i) {
			Gathered &gb = g[ids[i]];
			for (int j = 0; j < n; ++j) {
				if (!gb.shard[j].empty()) {
					sp[i * n + j] = gb.shard[j].data();
				} else if (j < k) {
					gb.shard[j].resize(S);
					op[i * n + j] = gb.shard[j].data();
				}
			}
		}
		int rc = gec_reconstruct_batch(mg->codec, ids.size(), sp.data(), op.data(), S, /*data_only=*/1);
		if (rc)
			return ec_fail(rc, "gec_reconstruct_batch");
		std::lock_guard<std::mutex> lk(mg->mu);
		mg->metrics[3] += ids.size();
	}
	// assemble, then check every block's content against its name (DataBlock::verify,
	// block.rs:69-77) -- all block hashes in one batch.  Plain blocks are assembled straight
	// into the caller's buffer and hashed from there (on CORRUPT_DATA its contents are
	// unspecified); only compressed blocks need an intermediate for the zstd frame.
	std::vector<const uint8_t *> ptrs;
	std::vector<size_t> lens, idx;
	auto assemble = [&](size_t b, uint8_t *dst) {
		const size_t L = g[b].meta.orig_len, S = g[b].meta.shard_len;
		for (int j = 0; j < k; ++j) {
			size_t lo = (size_t)j * S;
			if (lo >= L)
				break;
			std::memcpy(dst + lo, g[b].shard[j].data(), std::min(S, L - lo));
		}
	};
	for (size_t b = 0; b < nb; ++b) {
		if (rcs[b] != GBM_OK)
			continue;
		const size_t L = g[b].meta.orig_len, S = g[b].meta.shard_len;
		if (L > (size_t)k * S) {
			rcs[b] = GBM_E_CORRUPT_DATA;
			continue;
		}
		if (g[b].meta.compressed) {
			// DataBlock::verify for Compressed = "the zstd stream decodes" (frame checksum)
			std::vector<uint8_t> frame(L), plain;
			assemble(b, frame.data());
			if (!zstd().decode(frame.data(), L, plain)) {
				rcs[b] = GBM_E_CORRUPT_DATA;
				continue;
			}
			len_out[b] = plain.size();
			if (cap[b] < plain.size()) {
				rcs[b] = GBM_E_BUFFER_TOO_SMALL;
				continue;
			}
			std::memcpy(out[b], plain.data(), plain.size());
			std::lock_guard<std::mutex> lk(mg->mu);
			mg->metrics[5]++;
			continue;
		}
		if (cap[b] < L) {
			rcs[b] = GBM_E_BUFFER_TOO_SMALL;
			continue;
		}
		assemble(b, out[b]);
		ptrs.push_back(out[b]);
		lens.push_back(L);
		idx.push_back(b);
	}
	std::vector<uint8_t> sums;
	int hrc = hash_many(mg, ptrs, lens, sums);
	if (hrc)
		return hrc;
	for (size_t i = 0; i < idx.size(); ++i) {
		const size_t b = idx[i];
		if (std::memcmp(sums.data() + 32 * i, hashes + 32 * b, 32) != 0) {
			rcs[b] = GBM_E_CORRUPT_DATA;
			continue;
		}
		std::lock_guard<std::mutex> lk(mg->mu);
		mg->metrics[5]++;
	}
	return GBM_OK;
}

int gbm_rpc_get_block(gbm_manager *m, const uint8_t hash[32], uint8_t *out, size_t cap, size_t *len_out)
{
	if (!len_out)
		return fail(GBM_E_INVALID_ARG, "NULL len_out");
	uint8_t *o[1] = {out};
	int rc1 = GBM_OK;
	int rc = gbm_rpc_get_blocks(m, 1, hash, o, &cap, len_out, &rc1);
	if (rc)
		return rc;
	switch (rc1) {
	case GBM_E_MISSING_BLOCK: return fail(rc1, "Missing block: no node returned a valid block");
	case GBM_E_CORRUPT_DATA: return fail(rc1, "Corrupt data: does not match hash");
	case GBM_E_BUFFER_TOO_SMALL: return fail(rc1, "output buffer too small");
	default: return rc1;
	}
}

int gbm_block_incref(gbm_manager *m, const uint8_t hash[32])
{
	if (!m || !hash)
		return fail(GBM_E_INVALID_ARG, "NULL argument");
	Hash h((const char *)hash, 32);
	std::lock_guard<std::mutex> lk(m->mu);
	if (++m->rc[h] == 1)
		m->resync_queue.push_back(h);  // presence check later (manager.rs:452-475)
	return GBM_OK;
}

int gbm_block_decref(gbm_manager *m, const uint8_t hash[32])
{
	if (!m || !hash)
		return fail(GBM_E_INVALID_ARG, "NULL argument");
	Hash h((const char *)hash, 32);
	std::lock_guard<std::mutex> lk(m->mu);
	uint64_t &c = m->rc[h];
	if (c > 0)
		--c;
	if (c == 0)
		m->resync_queue.push_back(h);
	return GBM_OK;
}

int gbm_resync_block(gbm_manager *mg, const uint8_t hash[32], int *changed)
{
	if (!mg || !hash)
		return fail(GBM_E_INVALID_ARG, "NULL argument");
	Hash h((const char *)hash, 32);
	std::vector<int> who;
	mg->nodes_of(h, who);
	int nchanged = 0;
	uint64_t refs;
	{
		std::lock_guard<std::mutex> lk(mg->mu);
		auto it = mg->rc.find(h);
		refs = it == mg->rc.end() ? 0 : it->second;
	}
	if (refs == 0) {  // unneeded: delete everywhere (resync.rs:369-458, without the offload step)
		for (int j = 0; j < mg->n; ++j) {
			Node &nd = *mg->nodes[who[j]];
			std::vector<uint8_t> raw;
			if (!nd.down && nd.get(h, j, raw)) {
				nd.del(h, j);
				++nchanged;
			}
		}
		if (changed)
			*changed = nchanged;
		return GBM_OK;
	}
	Gathered g;
	gather(mg, h, mg->n, g);
	if (!g.have_meta || g.count < mg->k)
		return fail(GBM_E_MISSING_BLOCK, "Missing block: fewer than k shards reachable");
	if (g.count < mg->n) {  // needed but absent somewhere (resync.rs:460-500): rebuild and rewrite
		const size_t S = g.meta.shard_len;
		std::vector<const uint8_t *> sp(mg->n, nullptr);
		std::vector<uint8_t *> op(mg->n, nullptr);
		for (int j = 0; j < mg->n; ++j) {
			if (!g.shard[j].empty()) {
				sp[j] = g.shard[j].data();
			} else {
				g.shard[j].resize(S);
				op[j] = g.shard[j].data();
			}
		}
		int rc = gec_reconstruct_batch(mg->codec, 1, sp.data(), op.data(), S, 0);
		if (rc)
			return ec_fail(rc, "gec_reconstruct_batch");
		{
			std::lock_guard<std::mutex> lk(mg->mu);
			mg->metrics[3]++;
		}
		for (int j = 0; j < mg->n; ++j) {
			if (sp[j])
				continue;
			if (store_shard(mg, who[j], h, j, g.shard[j].data(), S, g.meta.orig_len, g.meta.compressed != 0) == 0)
				++nchanged;
			else
				mg->enqueue(h);
		}
	}
	if (changed)
		*changed = nchanged;
	return GBM_OK;
}

int gbm_resync_all(gbm_manager *mg, int *changed)
{
	if (!mg)
		return fail(GBM_E_INVALID_ARG, "NULL manager");
	std::vector<Hash> todo;
	{
		std::lock_guard<std::mutex> lk(mg->mu);
		todo.swap(mg->resync_queue);
	}
	std::sort(todo.begin(), todo.end());
	todo.erase(std::unique(todo.begin(), todo.end()), todo.end());
	int total = 0, result = GBM_OK;
	for (const Hash &h : todo) {
		int c = 0;
		int rc = gbm_resync_block(mg, (const uint8_t *)h.data(), &c);
		if (rc)
			result = rc;
		total += c;
	}
	if (changed)
		*changed = total;
	return result;
}

size_t gbm_resync_queue_len(const gbm_manager *m)
{
	if (!m)
		return 0;
	std::lock_guard<std::mutex> lk(const_cast<gbm_manager *>(m)->mu);
	return m->resync_queue.size();
}

int gbm_scrub(gbm_manager *mg, size_t nb, const uint8_t *hashes, uint8_t *bad_out)
{
	if (!mg || (nb && (!hashes || !bad_out)))
		return fail(GBM_E_INVALID_ARG, "NULL argument");
	std::vector<Hash> hs(nb);
	for (size_t b = 0; b < nb; ++b)
		hs[b].assign((const char *)hashes + 32 * b, 32);
	std::vector<Gathered> g;
	int grc = gather_many(mg, hs, mg->n, g);
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
	std::vector<uint8_t> raw;
	return m->nodes[node]->get(Hash((const char *)hash, 32), idx, raw) ? 1 : 0;
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
	std::vector<uint8_t> raw;
	if (!m->nodes[node]->get(h, idx, raw) || raw.size() <= GBM_SHARD_HEADER_SIZE + offset)
		return fail(GBM_E_IO, "no such shard / offset");
	raw[GBM_SHARD_HEADER_SIZE + offset] ^= mask;
	if (fix_checksum) {
		ShardHeader hd;
		hd.unpack(raw.data(), raw.size());
		blake2sum(raw.data() + GBM_SHARD_HEADER_SIZE, raw.size() - GBM_SHARD_HEADER_SIZE, hd.checksum);
		hd.pack(raw.data());
	}
	return m->nodes[node]->put(h, idx, std::move(raw)) ? GBM_OK : fail(GBM_E_IO, "rewrite failed");
}

int gbm_set_compression_level(gbm_manager *m, int enabled, int level)
{
	if (!m)
		return fail(GBM_E_INVALID_ARG, "NULL manager");
	if (enabled && !zstd().ok)
		return fail(GBM_E_IO, "libzstd.so.1 not available");
	m->compress = enabled != 0;
	m->compression_level = level;
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
	bool stop = false;
	uint64_t batches = 0, blocks = 0, max_batch = 0;
	std::thread worker;

	void run()
	{
		std::unique_lock<std::mutex> lk(mu);
		for (;;) {
			cv_work.wait(lk, [&] { return stop || !queue.empty(); });
			if (queue.empty()) {
				if (stop)
					return;
				continue;
			}
			// linger a little so concurrent callers land in the same batch
			// system_clock: libstdc++ maps it to pthread_cond_timedwait, which ThreadSanitizer
			// understands (steady_clock -> pthread_cond_clockwait is not intercepted by gcc 11's
			// TSan and floods the report with false "double lock" findings)
			const auto deadline = std::chrono::system_clock::now() + std::chrono::microseconds(max_wait_us);
			while (!stop && queue.size() < max_blocks &&
			       cv_work.wait_until(lk, deadline) != std::cv_status::timeout) {
			}
			std::vector<Item *> batch;
			while (!queue.empty() && batch.size() < max_blocks) {
				batch.push_back(queue.front());
				queue.pop_front();
			}
			lk.unlock();
			const size_t nb = batch.size();
			std::vector<uint8_t> hashes(nb * 32);
			std::vector<const uint8_t *> data(nb);
			std::vector<size_t> lens(nb);
			std::vector<int> rcs(nb, GBM_OK);
			for (size_t i = 0; i < nb; ++i) {
				std::memcpy(hashes.data() + 32 * i, batch[i]->hash, 32);
				data[i] = batch[i]->data;
				lens[i] = batch[i]->len;
			}
			int rc = put_blocks_impl(mg, nb, hashes.data(), data.data(), lens.data(), rcs.data());
			lk.lock();
			for (size_t i = 0; i < nb; ++i) {
				// a whole-batch failure (device error) hits every block of the batch
				batch[i]->rc = (rc != GBM_OK && rc != GBM_E_QUORUM) ? rc : rcs[i];
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
	b->worker = std::thread([b] { b->run(); });
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
	b->worker.join();
	delete b;
}

int gbm_batcher_put_block(gbm_batcher *b, const uint8_t hash[32], const uint8_t *data, size_t len)
{
	if (!b || !hash || (!data && len))
		return fail(GBM_E_INVALID_ARG, "NULL argument");
	gbm_batcher::Item it;
	it.hash = hash;
	it.data = data;
	it.len = len;
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
	b->cv_work.notify_one();
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

uint64_t gbm_gpu_hashed(const gbm_manager *m)
{
	if (!m)
		return 0;
	std::lock_guard<std::mutex> lk(const_cast<gbm_manager *>(m)->mu);
	return m->gpu_hashed;
}

int gbm_metrics(const gbm_manager *m, uint64_t out[6])
{
	if (!m || !out)
		return fail(GBM_E_INVALID_ARG, "NULL argument");
	std::lock_guard<std::mutex> lk(const_cast<gbm_manager *>(m)->mu);
	std::copy(m->metrics, m->metrics + 6, out);
	return GBM_OK;
}

}  // extern "C"
