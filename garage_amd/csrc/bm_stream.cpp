// bm_stream.cpp -- the streaming gets (rpc_get_block_streaming, rpc_get_raw_block_streaming, src/block/manager.rs:243-256,344-363)
// and the ranged get (body_from_blocks_range, src/api/s3/get.rs:650-743).
#include "bm_internal.hpp"

using namespace gbmimpl;

// ------------------------------------------------------------------ the streaming gets
// rpc_get_block_streaming hands the network stream through (manager.rs:344-363).  Here the "stream" is the block's data
// shards in index order: shard i IS the bytes [i*S, (i+1)*S) of the stored DataBlock, so it can leave as soon as its own
// checksum has matched.  The k shards in hand are checked side by side on the async pool (a shard of a 1 MiB block:
// ~35 us on a core); the calling thread walks the shards in order and hands each one to the sink straight out of its
// buffer the moment its verdict is in; a missing data shard is rebuilt (one small decode on the request path's codec)
// when the walk reaches it; the end-to-end hash, when its mode asks for it, runs on a thread of its own BEHIND the walk
// and only decides the final result.
namespace {

struct StreamChecks {  // shared with the async tasks: they own what they touch
	std::mutex mu;
	std::condition_variable cv;
	std::vector<int> verdict;  // per shard index: 0 = pending, 1 = matches its header's checksum, -1 = does not
	std::vector<Bytes> shard;
	std::vector<std::array<uint8_t, 32>> sum;
	// A shard's check is its checksum tree: the leaves in `groups` pieces (claimed in order, first shard first), then
	// the root by whoever finishes the shard's last piece.
	std::vector<int> used;                      // the read set, in index order
	size_t S = 0, nleaf = 0, groups = 1;
	int ver = 3;                                // the checksums' kind (the manager's shard-header version: the gather normalises)
	std::vector<std::vector<uint8_t>> dig;      // per entry of `used`: nleaf leaf digests (v2: 64 bytes each) / leaf sums (v3: 8 bytes)
	std::unique_ptr<std::atomic<int>[]> left;   // per entry of `used`: pieces not yet hashed
	std::atomic<size_t> next{0};                // next piece to claim: entry = next / groups, piece = next % groups

	bool check_next()  // false: nothing left to claim
	{
		const size_t t = next.fetch_add(1);
		if (t >= used.size() * groups)
			return false;
		const size_t e = t / groups, gi = t % groups;
		const int j = used[e];
		const size_t lo = nleaf * gi / groups, hi = nleaf * (gi + 1) / groups;
		if (hi > lo) {
			if (ver == 3)
				mlh::leaf_range(shard[j].data(), S, lo, hi, reinterpret_cast<uint64_t *>(dig[e].data()));
			else
				b2host::shardsum_leaf_range(shard[j].data(), S, lo, hi, dig[e].data());
		}
		if (left[e].fetch_sub(1) == 1) {  // the shard's last piece: its root, its verdict
			uint8_t got[32];
			if (ver == 3)
				mlh::root(S, reinterpret_cast<const uint64_t *>(dig[e].data()), nleaf, got);
			else
				b2host::shardsum_root(dig[e].data(), nleaf, got);
			const int v = std::memcmp(got, sum[j].data(), 32) == 0 ? 1 : -1;
			{
				std::lock_guard<std::mutex> lk(mu);
				verdict[j] = v;
			}
			cv.notify_all();
		}
		return true;
	}
};

// the block hash behind the stream: segments are pushed in order by the walk, hashed by a thread of its own
struct TailHash {
	std::mutex mu;
	std::condition_variable cv;
	std::deque<std::pair<Bytes, std::pair<const uint8_t *, size_t>>> q;  // (owner, range)
	bool closed = false;
	b2host::State st;
	std::thread th;
	void start()
	{
		th = std::thread([this] {
			name_thread("gbm-tail-hash");
			for (;;) {
				std::pair<Bytes, std::pair<const uint8_t *, size_t>> seg;
				{
					std::unique_lock<std::mutex> g(mu);
					cv.wait(g, [&] { return closed || !q.empty(); });
					if (q.empty())
						return;
					seg = std::move(q.front());
					q.pop_front();
				}
				st.update(seg.second.first, seg.second.second);
			}
		});
	}
	void push(const Bytes &owner, const uint8_t *p, size_t n)
	{
		{
			std::lock_guard<std::mutex> g(mu);
			q.emplace_back(owner, std::make_pair(p, n));
		}
		cv.notify_one();
	}
	// waits for the hasher; the digest of everything pushed
	void finish(uint8_t out[32])
	{
		{
			std::lock_guard<std::mutex> g(mu);
			closed = true;
		}
		cv.notify_one();
		if (th.joinable())
			th.join();
		uint8_t full[64];
		st.final(full);
		std::memcpy(out, full, 32);
	}
	~TailHash()
	{
		{
			std::lock_guard<std::mutex> g(mu);
			closed = true;
			q.clear();
		}
		cv.notify_one();
		if (th.joinable())
			th.join();
	}
};

// Where a stream's bytes go: the sink (through the incremental zstd decoder for a Compressed block read as plain bytes)
// and, when the mode asks for it, the hash behind the stream.
struct StreamOut {
	gbm_chunk_fn sink;
	void *ctx;
	size_t ch;
	bool z = false, raw = false;
	bool aborted = false, frame_bad = false, hashing = false;
	TailHash tail;
	std::unique_ptr<Zstd::Stream> zs;
	std::vector<uint8_t> zbuf, whole;  // decoder output not yet handed out / the frame, when the library cannot stream
	size_t plain_len = 0;
	std::vector<std::pair<Bytes, size_t>> sent;  // what has been delivered (owner, bytes): a hash that starts late catches up

	StreamOut(gbm_chunk_fn s, void *c, size_t chunk) : sink(s), ctx(c), ch(chunk ? chunk : 65536) {}
	void open(bool compressed, bool raw_)
	{
		z = compressed;
		raw = raw_;
		if (z && !raw && zstd().streaming) {
			zs.reset(new Zstd::Stream(zstd()));
			if (!zs->ds)
				zs.reset();  // no decoder (an allocation failed): the whole-frame form below serves -- NOT "corrupt data"
			else
				zbuf.reserve(ch);
		}
	}
	void start_hash()  // (from the first byte: whatever went out before is hashed first)
	{
		if (hashing || z)
			return;
		hashing = true;
		tail.start();
		for (auto &pr : sent)
			tail.push(pr.first, pr.first.data(), pr.second);
	}
	bool to_sink(const uint8_t *p, size_t len)  // chunks of at most `ch` bytes
	{
		for (size_t off = 0; off < len && !aborted; off += ch)
			if (sink(ctx, p + off, std::min(ch, len - off)) != 0)
				aborted = true;
		return !aborted;
	}
	// `len` stored bytes of the block, in order, at the start of `owner`.  false: stop (corrupt frame / abort)
	bool deliver(const Bytes &owner, size_t len)
	{
		const uint8_t *p = owner.data();
		sent.emplace_back(owner, len);
		if (hashing)
			tail.push(owner, p, len);
		if (!z || raw)
			return to_sink(p, len);
		if (!zs) {  // no incremental decoder in this libzstd: the frame is collected and decoded at the end
			whole.insert(whole.end(), p, p + len);
			return true;
		}
		const bool ok = zs->feed(p, len, [&](const uint8_t *o, size_t on) {
			plain_len += on;
			if (plain_len > kMaxDecompressed)
				return false;
			while (on) {  // hand out full chunks, keep the rest
				const size_t take = std::min(on, ch - zbuf.size());
				zbuf.insert(zbuf.end(), o, o + take);
				o += take;
				on -= take;
				if (zbuf.size() == ch) {
					if (!to_sink(zbuf.data(), zbuf.size()))
						return false;
					zbuf.clear();
				}
			}
			return true;
		});
		if (!ok && !aborted)
			frame_bad = true;
		return ok;
	}
	// the tail: what is left in the decoder, then the checks that can only be made once everything has gone by
	int finish(const uint8_t hash[32])
	{
		if (aborted)
			return fail(GBM_E_ABORTED, "the stream's consumer stopped");
		if (z && !raw) {
			if (!zs) {
				std::vector<uint8_t> plain;
				if (frame_bad || !zstd().decode(whole.data(), whole.size(), kMaxDecompressed, plain))
					return one_block_rc(GBM_E_CORRUPT_DATA);
				if (!to_sink(plain.data(), plain.size()))
					return fail(GBM_E_ABORTED, "the stream's consumer stopped");
			} else {
				if (frame_bad || !zs->frame_done)  // a frame that does not end, or whose checksum does not match (block.rs:78-83)
					return one_block_rc(GBM_E_CORRUPT_DATA);
				if (!zbuf.empty() && !to_sink(zbuf.data(), zbuf.size()))
					return fail(GBM_E_ABORTED, "the stream's consumer stopped");
			}
		}
		if (hashing) {
			uint8_t sum[32];
			tail.finish(sum);
			if (std::memcmp(sum, hash, 32) != 0)
				return one_block_rc(GBM_E_CORRUPT_DATA);
		}
		return GBM_OK;
	}
};

struct StreamGeom {
	size_t L = 0, S = 0;
	bool z = false;
};

// The general form: gather k shards (any holders, older layout versions, parity), check them side by side, rebuild what is
// missing, deliver from byte `skip` on (everything before it has gone out already: the fast path below hands over here
// when a shard is not where it should be).  `geom` != NULL: the geometry the stream has been opened with.
int stream_general(gbm_manager *m, const std::vector<Hash> &hs, const uint8_t hash[32], const gbm_order_tag *order_tag,
		   gbm_data_block_header *hdr, bool raw, StreamOut &out, size_t skip, const StreamGeom *geom)
{
	const int k = m->k, n = m->n;
	std::vector<Gathered> g;
	Trace tr("streaming get (general)");
	int grc = gather_many(m, hs, order_tag, k, g, /*verify=*/false);
	if (grc)
		return grc;
	tr.lap("gather");
	if (!g[0].have_meta || g[0].count < k)
		return one_block_rc(g[0].corrupt_seen || skip ? GBM_E_CORRUPT_DATA : GBM_E_MISSING_BLOCK);
	if (g[0].meta.orig_len > (uint64_t)k * g[0].meta.shard_len)
		return one_block_rc(GBM_E_CORRUPT_DATA);
	const size_t L = g[0].meta.orig_len, S = g[0].meta.shard_len;
	const bool z = g[0].meta.compressed != 0;
	if (geom && (geom->L != L || geom->S != S || geom->z != z))
		return one_block_rc(GBM_E_CORRUPT_DATA);  // another geometry took over mid-stream
	if (!geom) {
		if (hdr)
			hdr->kind = z ? GBM_HEADER_COMPRESSED : GBM_HEADER_PLAIN;  // reported before the first byte
		out.open(z, raw);
	}
	const int mode = m->verify_mode.load();

	// ---- the k shards the block is read from (the first k in hand, in index order): checked side by side
	auto ck = std::make_shared<StreamChecks>();
	ck->verdict.assign(n, 0);
	ck->shard = g[0].shard;
	ck->sum = g[0].sum;
	std::vector<int> used;
	bool need_decode = false;
	for (int j = 0; j < n && (int)used.size() < k; ++j)
		if (!g[0].shard[j].empty())
			used.push_back(j);
	for (int j = 0; j < k; ++j)
		need_decode = need_decode || g[0].shard[j].empty();
	// The checks are claimed piece by piece, the first shard's pieces first -- by a few helpers on the async pool and by
	// the walk itself while it waits: shard 0's verdict takes a fraction of one shard's hashing time (its leaves are
	// independent chains), the others' follow in index order as the stream advances.
	ck->used = used;
	ck->S = S;
	ck->ver = m->sumver;
	ck->nleaf = ck->ver == 3 ? mlh::nleaf(S) : b2host::shardsum_nleaf(S);
	// pieces of ~100 KiB: smaller ones are over before a helper has even woken up (1 MiB blocks: a shard is one piece and the
	// walk checks shard 0 itself, 35 us; 4 MiB blocks: four pieces per shard)
	ck->groups = std::min<size_t>(8, std::max<size_t>(1, S / (96u << 10)));
	ck->dig.assign(used.size(), std::vector<uint8_t>(ck->nleaf * 64));
	ck->left.reset(new std::atomic<int>[used.size()]);
	for (size_t e = 0; e < used.size(); ++e)
		ck->left[e] = (int)ck->groups;
	{
		const unsigned hw = std::max(2u, std::thread::hardware_concurrency());
		const size_t pieces = used.size() * ck->groups;
		const size_t helpers = std::min<size_t>(pieces > 1 ? pieces - 1 : 0, std::max(1u, hw - 2));
		std::shared_ptr<gbm_manager::Async> async = m->async_pool();
		for (size_t i = 0; i < helpers; ++i)
			async->submit([ck] {
				while (ck->check_next()) {
				}
			});
	}
	auto wait_verdict = [&](int j) {
		for (;;) {
			{
				std::lock_guard<std::mutex> lk(ck->mu);
				if (ck->verdict[j] != 0)
					return ck->verdict[j];
			}
			if (!ck->check_next())  // everything is claimed: the verdict is on its way
				break;
		}
		std::unique_lock<std::mutex> lk(ck->mu);
		ck->cv.wait(lk, [&] { return ck->verdict[j] != 0; });
		return ck->verdict[j];
	};
	if (mode == GBM_VERIFY_ALWAYS || (mode == GBM_VERIFY_REBUILT && need_decode))
		out.start_hash();

	// ---- the walk
	std::vector<Bytes> rebuilt(k);
	bool decoded = false;
	size_t pos = 0;  // stored bytes walked over so far (delivered, or below `skip`)
	int bad_shard = -1;
	for (int j = 0; j < k && pos < L && !out.aborted && !out.frame_bad; ++j) {
		const size_t len = std::min(S, L - pos);
		if (pos + len <= skip) {  // went out before the hand-over
			pos += len;
			continue;
		}
		if (!g[0].shard[j].empty()) {
			if (wait_verdict(j) < 0) {
				bad_shard = j;
				break;
			}
			if (j == 0)
				tr.lap("first shard checked");
			if (!out.deliver(g[0].shard[j], len))
				break;
			pos += len;
			continue;
		}
		if (!decoded) {
			// a missing data shard: every shard the decode reads must have matched first
			for (int u : used)
				if (wait_verdict(u) < 0) {
					bad_shard = u;
					break;
				}
			if (bad_shard >= 0)
				break;
			std::vector<const uint8_t *> sp(n, nullptr);
			std::vector<uint8_t *> op(n, nullptr);
			try {
				for (int t = 0; t < k; ++t)
					if (g[0].shard[t].empty()) {
						rebuilt[t] = m->bufs->get(S);
						op[t] = rebuilt[t].mut();
					}
			} catch (const std::bad_alloc &) {
				return fail(GBM_E_IO, "out of (pinned) host memory");
			}
			for (int u : used)
				sp[u] = g[0].shard[u].data();
			int rc = gec_reconstruct_batch(m->codec, 1, sp.data(), op.data(), S, /*data_only=*/1);
			if (rc)
				return ec_fail(rc, "gec_reconstruct_batch");
			m->metrics[3]++;
			decoded = true;
		}
		if (!out.deliver(rebuilt[j], len))
			break;
		pos += len;
	}
	if (bad_shard >= 0) {
		// read_block_from's corrupt-file case (manager.rs:577-609), met mid-stream: the shard is set aside and queued, and
		// the rest of the block comes from the batch path's gather / check / decode rounds (what was already delivered had
		// matched its checksums and stays delivered)
		m->metrics[2]++;
		if (g[0].node[bad_shard] >= 0)
			m->nodes[g[0].node[bad_shard]]->mark_corrupted(hs[0], bad_shard);
		m->put_to_resync(hs[0], 0);
		std::vector<Gathered> g2;
		std::vector<uint8_t> bsums;
		int rc1 = GBM_OK;
		int frc = fetch_blocks(m, hs, order_tag, g2, &rc1, 0, bsums);
		if (frc)
			return frc;
		if (rc1 != GBM_OK)
			return one_block_rc(rc1 == GBM_E_MISSING_BLOCK ? GBM_E_CORRUPT_DATA : rc1);  // shards were there: they were corrupt
		if (g2[0].meta.orig_len != L || g2[0].meta.shard_len != S || (g2[0].meta.compressed != 0) != z)
			return one_block_rc(GBM_E_CORRUPT_DATA);  // another geometry took over mid-stream
		if (mode == GBM_VERIFY_REBUILT)
			out.start_hash();  // the replacement comes out of a decode after all: the block is hashed, from its first byte
		for (int j = (int)(pos / S); j < k && pos < L && !out.aborted && !out.frame_bad; ++j) {
			const size_t len = std::min(S, L - pos);
			if (!out.deliver(g2[0].shard[j], len))
				break;
			pos += len;
		}
	}
	tr.lap("last shard delivered");
	int rc = out.finish(hash);
	if (rc == GBM_OK)
		m->metrics[5]++;
	return rc;
}

// true when the k holders a read asks first (read_candidate_order) are the k data shards' holders in the oldest active layout
// version -- always the case for a manager that knows nothing about zones and pings
static bool nearest_k_are_data(const gbm_manager *m, const Hash &h)
{
	if (!m->locality_set.load(std::memory_order_relaxed))
		return true;
	std::vector<uint32_t> order;
	read_candidate_order(m, h, m->layout_oldest.load(), m->layout_cur.load(), order);
	for (int i = 0; i < m->k && i < (int)order.size(); ++i)
		if (order[i] >= (uint32_t)m->k)  // (candidate index = version-major, shard minor: < k means a data shard of the oldest version)
			return false;
	return true;
}

// The fast path's requests: data shards lo..hi asked for AT ONCE, each from the node that should hold it in the current
// layout version; a shard is checked (header, checksum) by the task that fetched it.  Shard `lo` is fetched by the calling
// thread itself: the first byte waits for no other thread to wake up.
struct Fast {
	std::mutex mu;
	std::condition_variable cv;
	std::vector<int> st;  // 0 pending, 1 arrived and matches its own checksum, -1 not usable
	std::vector<Shard> shard;
	Hash h;
	gbm_order_tag tag{0, 0};
	bool has_tag = false;
	Fast(int k, const Hash &hash, const gbm_order_tag *order_tag) : st(k, 0), shard(k), h(hash)
	{
		if (order_tag) {
			tag = *order_tag;
			has_tag = true;
		}
	}
	bool arrived(int j)
	{
		std::unique_lock<std::mutex> lk(mu);
		cv.wait(lk, [&] { return st[j] != 0; });
		return st[j] == 1;
	}
};

void fast_fetch(gbm_manager *m, const std::shared_ptr<Fast> &fs, const std::vector<int> &who, int lo, int hi)
{
	std::shared_ptr<gbm_manager::Async> async = m->async_pool();
	const int mk = m->k, mm = m->m;
	auto fetch = [fs, mk, mm](Node *nd, int j) {
		ShardRpc rq{RpcKind::GetShard, &fs->h, j, Shard(), fs->has_tag ? &fs->tag : nullptr};
		ShardResp rs;
		int v = -1;
		try {
			if (nd->handle(rq, rs) && rs.ok) {
				const ShardHeader &hd = rs.shard.hd;
				if (hd.version >= 2 && hd.version <= 3 && hd.idx == j && hd.k == mk && hd.m == mm && hd.shard_len > 0 && hd.shard_len % 64 == 0 &&
				    rs.shard.data.n == hd.shard_len) {
					uint8_t sum[32];
					shardsum_v(hd.version, rs.shard.data.data(), hd.shard_len, sum);
					if (std::memcmp(sum, hd.checksum, 32) == 0)
						v = 1;
				}
			}
		} catch (...) {  // (no memory for the shard's copy, as a rule) "not there": the general form takes over; on a helper
			v = -1;  // thread nothing may escape, and whoever waits in arrived(j) must be woken
		}
		{
			std::lock_guard<std::mutex> lk(fs->mu);
			if (v == 1)
				fs->shard[j] = std::move(rs.shard);
			fs->st[j] = v;
		}
		fs->cv.notify_all();
	};
	for (int j = lo + 1; j <= hi; ++j) {
		Node *nd = m->nodes[who[j]].get();
		async->submit([fetch, nd, j] { fetch(nd, j); });
	}
	fetch(m->nodes[who[lo]].get(), lo);
}

// The streaming get.  The fast path is the healthy block: its k data shards are asked for AT ONCE, each from the node that
// should hold it in the current layout version; a shard is checked (header, checksum) by the task that fetched it, and
// the walk hands shard i to the sink as soon as shards 0..i have arrived and matched -- the first byte waits for ONE
// node's answer and one shard's checksum, not for the slowest of k nodes.  The moment a shard is not there, not
// consistent with shard 0's geometry, or does not match, the general form takes over from the byte the walk has reached
// (other holders, older layout versions, parity + decode, the corrupt-shard bookkeeping).
int get_streaming(gbm_manager *m, const uint8_t hash[32], const gbm_order_tag *order_tag, gbm_data_block_header *hdr,
		  size_t chunk_bytes, gbm_chunk_fn sink, void *ctx, bool raw)
{
	if (!m || !hash || !sink)
		return fail(GBM_E_INVALID_ARG, "NULL argument");
	m = m->route(hash);
	DurationScope read_time(m->bmx.read_duration);
	const int k = m->k;
	std::vector<Hash> hs(1, Hash((const char *)hash, 32));
	StreamOut out(sink, ctx, chunk_bytes);
	auto fs = std::make_shared<Fast>(k, hs[0], order_tag);
	std::vector<int> who;
	// (the oldest active layout version's holders -- where the general form's walk starts as well, bm_gather.cpp; in the
	// steady state that is the current version)
	m->nodes_of(hs[0], m->layout_oldest.load(), who);
	// (zones / pings known and a data shard's holder is not among the k nearest: the general form asks the k nearest instead and
	// decodes -- a far data shard would cost the stream a WAN round trip, request_order's whole point, rpc_helper.rs:621-660)
	if (nearest_k_are_data(m, hs[0]))
		fast_fetch(m, fs, who, 0, k - 1);
	else
		std::fill(fs->st.begin(), fs->st.end(), -1);
	Trace tr("streaming get");
	StreamGeom geom;
	size_t pos = 0;
	bool opened = false, handover = false;
	for (int j = 0; j < k; ++j) {
		if (!fs->arrived(j)) {
			handover = true;
			break;
		}
		const ShardHeader &hd = fs->shard[j].hd;
		if (j == 0) {
			if (hd.orig_len > (uint64_t)k * hd.shard_len) {
				handover = true;
				break;
			}
			geom.L = hd.orig_len;
			geom.S = hd.shard_len;
			geom.z = hd.compressed != 0;
			if (hdr)
				hdr->kind = geom.z ? GBM_HEADER_COMPRESSED : GBM_HEADER_PLAIN;  // reported before the first byte
			out.open(geom.z, raw);
			opened = true;
			if (m->verify_mode.load() == GBM_VERIFY_ALWAYS)
				out.start_hash();
			tr.lap("first shard arrived and checked");
		} else if (hd.orig_len != geom.L || hd.shard_len != geom.S || (hd.compressed != 0) != geom.z) {
			handover = true;  // a stale shard of another geometry: the general form sorts the groups out
			break;
		}
		if (pos >= geom.L)
			break;
		const size_t len = std::min(geom.S, geom.L - pos);
		m->metrics[1] += hd.shard_len;
		if (!out.deliver(fs->shard[j].data, len))
			break;
		pos += len;
	}
	if (handover) {
		const size_t sent_before = out.sent.size();
		int rc = stream_general(m, hs, hash, order_tag, hdr, raw, out, pos, opened ? &geom : nullptr);
		// (a block whose shards were being moved to their new owners under the walk -- get_blocks_impl has the story -- is asked
		// for once more, as long as the general form has not delivered anything itself: it takes up at byte `pos` again)
		for (int attempt = 1; attempt <= 2 && (rc == GBM_E_MISSING_BLOCK || rc == GBM_E_CORRUPT_DATA) && out.sent.size() == sent_before &&
				      !out.aborted && !out.frame_bad && m->layout_cur.load() != m->layout_oldest.load();
		     ++attempt) {
			std::this_thread::sleep_for(std::chrono::milliseconds(attempt == 1 ? 1 : 5));
			rc = stream_general(m, hs, hash, order_tag, hdr, raw, out, pos, opened ? &geom : nullptr);
		}
		return rc;
	}
	tr.lap("last shard delivered");
	int rc = out.finish(hash);
	if (rc == GBM_OK)
		m->metrics[5]++;
	return rc;
}

// The reference's scan over a whole block's stream (body_from_blocks_range, src/api/s3/get.rs:687-723): chunks before
// `begin` are dropped, the ones that overlap the range are cut to it, and once `end` is behind it the stream is let go.
struct RangeSlice {
	gbm_chunk_fn sink;
	void *ctx;
	size_t begin, end, off = 0;
	bool done = false, consumer_stopped = false;
	static int fn(void *c, const uint8_t *p, size_t n)
	{
		RangeSlice *r = static_cast<RangeSlice *>(c);
		const size_t lo = r->off, hi = lo + n;
		r->off = hi;
		if (hi <= r->begin)
			return 0;
		const size_t a = std::max(lo, r->begin), b = std::min(hi, r->end);
		if (b > a && r->sink(r->ctx, p + (a - lo), b - a) != 0) {
			r->consumer_stopped = true;
			return 1;
		}
		if (hi >= r->end) {
			r->done = true;  // "the rest of the stream will be ignored" (get.rs:691-695)
			return 1;
		}
		return 0;
	}
};

// A byte range [begin, end) of one block.  Data shard i of a Plain block IS its bytes [i*S, (i+1)*S), so the range needs
// only the data shards it touches: `block_size` (the VersionBlock's size the caller's version table holds) says what S
// must be, those shards are asked for at once and each is checked against its own checksum before a byte of it goes
// out.  Whatever does not fit that picture -- the block is stored Compressed, its stored geometry is not what
// block_size implies, a shard is missing or does not match -- is the whole-block stream's business: it takes over
// behind a slicing sink, from the byte the range has reached, with everything a streaming get does (other holders,
// parity + decode, the corrupt-shard bookkeeping).
int get_range(gbm_manager *m, const uint8_t hash[32], const gbm_order_tag *order_tag, size_t block_size, size_t begin, size_t end,
	      size_t chunk_bytes, gbm_chunk_fn sink, void *ctx)
{
	if (!m || !hash || !sink)
		return fail(GBM_E_INVALID_ARG, "NULL argument");
	if (begin > end)
		return fail(GBM_E_INVALID_ARG, "range begins after its end");
	m = m->route(hash);
	DurationScope read_time(m->bmx.read_duration);
	const int k = m->k;
	const size_t ch = chunk_bytes ? chunk_bytes : 65536;
	size_t pos = begin;  // the next byte of the block the consumer is owed
	auto whole_block = [&]() {
		read_time.cancel();  // (the whole-block stream records its own)
		RangeSlice rs{sink, ctx, pos, end};
		int rc = get_streaming(m, hash, order_tag, nullptr, chunk_bytes, RangeSlice::fn, &rs, false);
		if (rc == GBM_E_ABORTED && rs.done && !rs.consumer_stopped)
			return (int)GBM_OK;  // let go on purpose
		return rc;
	};
	if (begin == end || block_size == 0 || begin >= block_size)
		return begin == end ? (int)GBM_OK : whole_block();  // (a range beyond block_size: the stored block decides)
	const size_t S = gec_shard_len(k, block_size);
	if (S == 0)
		return whole_block();
	const size_t last = std::min(end, block_size) - 1;
	const int j0 = (int)(begin / S), j1 = (int)(last / S);
	Hash h((const char *)hash, 32);
	auto fs = std::make_shared<Fast>(k, h, order_tag);
	std::vector<int> who;
	m->nodes_of(h, m->layout_oldest.load(), who);
	if (!nearest_k_are_data(m, h))
		return whole_block();
	Trace tr("range get");
	fast_fetch(m, fs, who, j0, j1);
	for (int j = j0; j <= j1; ++j) {
		if (!fs->arrived(j))
			return whole_block();
		const ShardHeader &hd = fs->shard[j].hd;
		if (hd.compressed != 0 || hd.orig_len != block_size || hd.shard_len != S)
			return whole_block();
		if (j == j0)
			tr.lap("first shard arrived and checked");
		m->metrics[1] += hd.shard_len;
		const size_t lo = (size_t)j * S;                       // block offset of this shard's first byte
		const size_t a = pos - lo, b = std::min(S, std::min(end, block_size) - lo);
		const uint8_t *p = fs->shard[j].data.data();
		for (size_t off = a; off < b; off += ch) {
			const size_t n = std::min(ch, b - off);
			if (sink(ctx, p + off, n) != 0)
				return fail(GBM_E_ABORTED, "the stream's consumer stopped");
			pos += n;
		}
	}
	tr.lap("last shard delivered");
	m->metrics[5]++;
	return GBM_OK;
}

}  // namespace

extern "C" {

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

int gbm_rpc_get_block_range_streaming(gbm_manager *m, const uint8_t hash[32], const gbm_order_tag *order_tag, size_t block_size,
				      size_t begin, size_t end, size_t chunk_bytes, gbm_chunk_fn sink, void *ctx)
{
	try {
		return get_range(m, hash, order_tag, block_size, begin, end, chunk_bytes, sink, ctx);
	} catch (const std::exception &e) {
		return fail(GBM_E_IO, std::string("rpc_get_block_range_streaming: ") + e.what());
	}
}

}  // extern "C"
