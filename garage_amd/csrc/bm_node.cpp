// bm_node.cpp -- the storage nodes behind ShardRpc: in-memory (64 lock stripes) and directory-backed
// (<root>/<h0>/<h1>/<hex>.s<idx>, tmp + rename like write_block_inner, src/block/manager.rs:720-805).
#include "bm_internal.hpp"

#include <dirent.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <sys/uio.h>
#include <unistd.h>

#include <cerrno>

namespace gbmimpl {

bool Node::handle(const ShardRpc &rq, ShardResp &rs)
{
	requests.fetch_add(1, std::memory_order_relaxed);
	if (down.load(std::memory_order_acquire))
		return false;
	if (const uint64_t us = latency_us.load(std::memory_order_relaxed))
		std::this_thread::sleep_for(std::chrono::microseconds(us));
	switch (rq.kind) {
	case RpcKind::PutShard:
		note_order(rq.tag);
		rs.ok = put(*rq.hash, rq.idx, rq.shard, &rs.pending);
		return true;
	case RpcKind::GetShard:
		rs.ok = get(*rq.hash, rq.idx, rs.shard);
		return true;
	case RpcKind::NeedShardQuery:
		rs.ok = true;
		rs.have_hd = header(*rq.hash, rq.idx, rs.shard.hd);
		rs.needed = !rs.have_hd && !has(*rq.hash, rq.idx);  // (a shard whose header does not parse is there all the same)
		return true;
	case RpcKind::DeleteShard:
		rs.ok = del(*rq.hash, rq.idx);
		return true;
	case RpcKind::CommitShard:
		rs.ok = commit(*rq.hash, rq.idx);
		return true;
	case RpcKind::AbortShard:
		rs.ok = abort(*rq.hash, rq.idx);
		return true;
	}
	return false;
}

void Node::note_order(const gbm_order_tag *tag)
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

namespace {

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
		std::unordered_map<std::string, Shard> files, parked;  // parked: waiting for CommitShard
	};
	Stripe stripes[kStripes];
	Stripe &stripe_of(const Hash &h) { return stripes[((unsigned char)h[2] ^ (unsigned char)h[3]) % kStripes]; }
	bool put(const Hash &h, int idx, const Shard &s, bool *pending) override
	{
		Stripe &st = stripe_of(h);
		std::lock_guard<std::mutex> g(st.mu);
		const std::string key = shard_key(h, idx);
		auto it = st.files.find(key);
		const bool park = it != st.files.end() && !it->second.hd.same_geometry(s.hd);
		if (pending)
			*pending = park;
		(park ? st.parked : st.files)[key] = s;
		return true;
	}
	bool commit(const Hash &h, int idx) override
	{
		Stripe &st = stripe_of(h);
		std::lock_guard<std::mutex> g(st.mu);
		const std::string key = shard_key(h, idx);
		auto it = st.parked.find(key);
		if (it == st.parked.end())
			return false;
		st.files[key] = std::move(it->second);
		st.parked.erase(it);
		return true;
	}
	bool abort(const Hash &h, int idx) override
	{
		Stripe &st = stripe_of(h);
		std::lock_guard<std::mutex> g(st.mu);
		return st.parked.erase(shard_key(h, idx)) != 0;
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
	bool header(const Hash &h, int idx, ShardHeader &hd) override
	{
		Stripe &st = stripe_of(h);
		std::lock_guard<std::mutex> g(st.mu);
		auto it = st.files.find(shard_key(h, idx));
		if (it == st.files.end())
			return false;
		hd = it->second.hd;
		return true;
	}
	bool del(const Hash &h, int idx) override
	{
		Stripe &st = stripe_of(h);
		std::lock_guard<std::mutex> g(st.mu);
		st.parked.erase(shard_key(h, idx));
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
	void list_prefix(int h0, std::set<Hash> &out) override { list_prefix_range(h0, h0 + 1, out); }
	void list_prefix_range(int lo, int hi, std::set<Hash> &out) override  // one scan whatever the width
	{
		for (Stripe &st : stripes) {
			std::lock_guard<std::mutex> g(st.mu);
			for (auto &kv : st.files) {
				const int h0 = (unsigned char)kv.first[0];
				if (h0 >= lo && h0 < hi)
					out.insert(kv.first.substr(0, 32));
			}
		}
	}
};

// <root>/<h0>/<h1>/<hex>.s<idx>, tmp file + rename (write_block_inner, manager.rs:720-805);
// a corrupt shard is renamed *.corrupted (manager.rs:807-819).  File = 64-byte header + payload.
struct DirNode : Node {
	std::string root;
	std::atomic<bool> fsync_data{false};  // Config.data_fsync (src/util/config.rs:22-24), off by default
	explicit DirNode(std::string r) : root(std::move(r)) {}
	void set_fsync(bool on) override { fsync_data = on; }
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
	// the geometry fields of the shard file in place, if there is one (header only: 64 bytes)
	bool header_in_place(const std::string &p, ShardHeader &hd) const
	{
		const int fd = ::open(p.c_str(), O_RDONLY | O_CLOEXEC);
		if (fd < 0)
			return false;
		uint8_t hdr[GBM_SHARD_HEADER_SIZE];
		const bool ok = read_all(fd, hdr, sizeof(hdr), 0) && hd.unpack(hdr, sizeof(hdr)) == ShardHeader::OK;
		::close(fd);
		return ok;
	}
	bool put(const Hash &h, int idx, const Shard &s, bool *pending) override
	{
		static std::atomic<uint64_t> seq{0};  // unique per writer: two threads may store the same shard
		const std::string d = dir(h);
		std::string p = path(h, idx);
		// a shard of ANOTHER geometry in place (the block was stored before with a different compression setting):
		// the new one is parked as <name>.parked until the manager commits it.  The probe costs one open() that
		// fails with ENOENT for every first-time put.
		ShardHeader old;
		const bool park = header_in_place(p, old) && !old.same_geometry(s.hd);
		if (pending)
			*pending = park;
		if (park)
			p += ".parked";
		std::string tmp = p + ".tmp" + std::to_string(::getpid()) + "_" + std::to_string(seq++);
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
		// a file whose header is garbage is handed up as an invalid shard (idx 0xff) so that the reader treats
		// it like a checksum failure: *.corrupted + resync.  A header of a VERSION this build does not know is
		// handed up with that version: the reader leaves such a file alone (it may belong to a newer build).
		if (ok) {
			const ShardHeader::Parse pr = s.hd.unpack(hdr, sizeof(hdr));
			if (pr == ShardHeader::GARBAGE) {
				s.hd = ShardHeader();
				s.hd.idx = 0xff;
			}
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
	bool header(const Hash &h, int idx, ShardHeader &hd) override { return header_in_place(path(h, idx), hd); }
	bool del(const Hash &h, int idx) override
	{
		std::remove((path(h, idx) + ".parked").c_str());
		return std::remove(path(h, idx).c_str()) == 0;
	}
	bool commit(const Hash &h, int idx) override
	{
		const std::string p = path(h, idx);
		return std::rename((p + ".parked").c_str(), p.c_str()) == 0;
	}
	bool abort(const Hash &h, int idx) override { return std::remove((path(h, idx) + ".parked").c_str()) == 0; }
	void mark_corrupted(const Hash &h, int idx) override
	{
		std::string p = path(h, idx);
		std::rename(p.c_str(), (p + ".corrupted").c_str());
	}
	// <root>/<h0>/<h1>/<64 hex digits>.s<idx>
	static void each_entry(const std::string &d, const std::function<void(const std::string &)> &fn)
	{
		if (DIR *dp = ::opendir(d.c_str())) {
			while (struct dirent *e = ::readdir(dp))
				if (e->d_name[0] != '.')
					fn(e->d_name);
			::closedir(dp);
		}
	}
	// every hash with a shard file under <root>/<a>
	void list_first_level(const std::string &a, std::set<Hash> &out) const
	{
		each_entry(root + "/" + a, [&](const std::string &b) {
			each_entry(root + "/" + a + "/" + b, [&](const std::string &f) {
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
	}
	void list(std::set<Hash> &out) override
	{
		each_entry(root, [&](const std::string &a) { list_first_level(a, out); });
	}
	void list_prefix(int h0, std::set<Hash> &out) override
	{
		static const char *hx = "0123456789abcdef";
		list_first_level(std::string{hx[(h0 >> 4) & 15], hx[h0 & 15]}, out);
	}
};

}  // namespace

std::unique_ptr<Node> make_memory_node() { return std::unique_ptr<Node>(new MemoryNode()); }
std::unique_ptr<Node> make_dir_node(const std::string &root) { return std::unique_ptr<Node>(new DirNode(root)); }

}  // namespace gbmimpl
