// bm_rw.cpp -- the request path: rpc_put_block(s) (encode + checksums in ONE device trip, per-node fan-out, write
// quorum) and rpc_get_block(s) (gather k shards, ONE decode + verify trip, assembly), plus their C entry points.
#include "bm_internal.hpp"

namespace gbmimpl {

// Fetch shards until every block has `want` valid ones of one geometry in hand (or ran out of nodes): shard
// index order within the current layout version, then older versions (block_read_nodes_of interleaves
// versions the same way, rpc_helper.rs:570-619).  The checksums of each round's candidates are verified in
// ONE batch; a shard whose checksum or header does not match is treated as missing, renamed *.corrupted and
// queued for resync (read_block_from's behaviour, manager.rs:577-609), and the next node is tried in the
// following round.
int gather_many(gbm_manager *mg, const std::vector<Hash> &hs, const gbm_order_tag *tags, int want, std::vector<Gathered> &gs,
		bool verify, const std::vector<uint8_t> *only)
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
		ShardHeader &hd = sh.hd;
		if (hd.version != 1 && hd.version != 2)
			return false;  // a shard format this build does not know: unreadable for us, but left alone (never renamed)
		bool ok = hd.idx == j && hd.k == mg->k && hd.m == mg->m && sh.data.n == hd.shard_len && hd.shard_len > 0 &&
			  hd.shard_len % 64 == 0;
		if (ok && hd.version == 1) {
			// round 1's format: the checksum is plain blake2sum.  Verified here, on the host (there are at most a
			// cluster's worth of such shards and each is read this way once), then carried -- and rewritten on its
			// node -- as version 2, so that everything downstream sees one format.
			uint8_t sum[32];
			blake2sum(sh.data.data(), hd.shard_len, sum);
			ok = std::memcmp(sum, hd.checksum, 32) == 0;
			if (ok) {
				hd.version = 2;
				shardsum(sh.data.data(), hd.shard_len, hd.checksum);
				ShardRpc up{RpcKind::PutShard, &hs[b], j, sh, nullptr};
				ShardResp ur;
				(void)mg->nodes[node]->handle(up, ur);
			}
		}
		if (!ok) {
			mg->metrics[2]++;
			mg->nodes[node]->mark_corrupted(hs[b], j);
			mg->put_to_resync(hs[b], 0);
			g.corrupt_seen = true;
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
				  int &j_out, size_t *c_out = nullptr) -> bool {
		Gathered &g = gs[b];
		if (g.tried.size() != ncand)
			g.tried.assign(ncand, 0);
		// a candidate is consumed when it is ASKED, not when it is passed over: shard j being covered by a request that is
		// still in flight says nothing about j's other holders, which are needed the moment that request fails (a hedge
		// timer that fired while all n first requests were in flight used to use up every older-version candidate)
		for (size_t c = 0; c < ncand; ++c) {
			if (g.tried[c])
				continue;
			const int v = vcur - (int)(c / n), j = (int)(c % n);
			if (have(g, j) || taken(j))
				continue;
			g.tried[c] = 1;
			if (c_out)
				*c_out = c;
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
				size_t b, cand = 0;  // cand: the candidate's index (it is given back when the request is abandoned)
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
				size_t cand = 0;
				while (launched < count && next_candidate(b, taken, who[b], who_v[b], j, &cand)) {
					flights_of[b].push_back(flights.size());
					auto f = std::make_shared<Flight>();
					f->b = b;
					f->cand = cand;
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
			for (;;) {
				const auto deadline = std::chrono::system_clock::now() + std::chrono::microseconds(hedge_us);
				if (!rd->cv.wait_until(lk, deadline, [&] { return rd->unsatisfied == 0; })) {
					uint64_t hedges = 0;
					for (size_t b = 0; b < hs.size(); ++b)
						if (!rd->satisfied(b))
							hedges += launch(b, rd->need[b] - rd->ok[b]);
					mg->hedged_reads += hedges;
					rd->cv.wait(lk, [&] { return rd->unsatisfied == 0; });
				}
				// "start another on each failure" (try_call_many_inner, rpc_helper.rs:323-411): a block whose requests have all
				// come back and that is still short asks its next holders -- the parity shards' nodes, then the older layout
				// versions' -- instead of ending the round empty-handed (a block whose shards are all still on the previous
				// layout's nodes was "missing" to a hedged read)
				int more = 0;
				for (size_t b = 0; b < hs.size(); ++b)
					if ((!only || (*only)[b]) && rd->ok[b] < rd->need[b] && rd->outstanding[b] == 0)
						more += launch(b, rd->need[b] - rd->ok[b]);
				if (!more)
					break;
			}
			rd->over = true;
			for (auto &f : flights) {
				if (f->done && f->answered && f->rs.ok)
					accept(per[f->b], f->b, f->j, f->node, std::move(f->rs.shard));
				else if (!f->done)
					// abandoned, not answered: its holder has not been heard -- if what the round did bring in does not hold
					// up (a shard that fails its checksum), the next round may ask it again.  (A round that was satisfied by a
					// parity shard which then proved corrupt used to find the slow data shard's holder "already asked" and
					// gave the block up as corrupt, with one good shard more than it needed still out there.)
					gs[f->b].tried[f->cand] = 0;
			}
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
			if (verify && std::memcmp(sums.data() + 32 * i, c.s.hd.checksum, 32) != 0 &&
			    confirmed_corrupt(mg, c.s.data.data(), c.s.hd.shard_len, c.s.hd.checksum, "the gather's checksum pass")) {
				mg->metrics[2]++;
				mg->nodes[c.node]->mark_corrupted(hs[c.b], c.j);
				mg->put_to_resync(hs[c.b], 0);
				g.corrupt_seen = true;
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
		bool compressed, const uint8_t *checksum, const gbm_order_tag *tag, bool *pending)
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
	const bool ok = mg->nodes[node]->handle(rq, rs) && rs.ok;
	if (pending)
		*pending = ok && rs.pending;
	return ok;
}

// rcs (optional): per-block result, GBM_OK or GBM_E_QUORUM; the return value is the last failure.  The device
// work of the whole batch happens before anything is sent to a node, so a device error (GBM_E_EC) fails
// every block of the batch and leaves no partial state behind.  (gbm_rpc_put_blocks cuts a big untagged request into
// slices that are independent puts: a device error in one slice does not undo the others.)
// Whatever the outcome, every entry of rcs is set: a whole-batch failure marks every block.
int put_blocks_impl(gbm_manager *mg, size_t nb, const uint8_t *hashes, const uint8_t *const *data, const size_t *len,
		    const uint8_t *prevent_compression, const gbm_order_tag *tags, int *rcs, const FanoutGate *gate)
{
	auto fail_all = [&](int code, const std::string &msg) {
		if (rcs)
			std::fill(rcs, rcs + nb, code);
		return fail(code, msg);
	};
	if (!mg || (nb && (!hashes || !data || !len)))
		return fail_all(GBM_E_INVALID_ARG, "NULL argument");
	if (nb == 0)
		return GBM_OK;
	for (size_t b = 0; b < nb; ++b)
		if (!data[b] && len[b])
			return fail_all(GBM_E_INVALID_ARG, "NULL block pointer");
	DurationScope write_time(mg->bmx.write_duration);  // block.write_duration (metrics.rs:127-131): one observation per call
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
	// (zstd, when asked for, holds the compressed payload until it is copied)
	std::vector<std::vector<uint8_t>> zbufs(nb);
	mg->pool->parallel_for(nb, [&](size_t b) {
		try {
			Prep &p = prep[b];
			p.plen = len[b];
			if (compress && !(prevent_compression && prevent_compression[b]) && zstd().encode(data[b], len[b], level, zbufs[b])) {
				p.plen = zbufs[b].size();
				p.z = true;
			}
			p.S = gec_shard_len(k, p.plen);
			p.block = mg->bufs->get((size_t)k * p.S);
			p.parity = mg->bufs->get((size_t)m * p.S);
		} catch (const std::bad_alloc &) {
			oom = true;
		}
	});
	if (!oom) {
		// the ONE copy of the payload, shard by shard: a PutObject's few blocks are cut into k pieces each so that the copy
		// of a single 1 MiB block is not one core's 80 us (it is most of what the host adds to a small put's latency)
		const size_t pieces = nb >= 32 ? 1 : (size_t)k;
		mg->pool->parallel_for(nb * pieces, [&](size_t i) {
			const size_t b = i / pieces, pc = i % pieces;
			Prep &p = prep[b];
			const uint8_t *src = p.z ? zbufs[b].data() : data[b];
			const size_t total = (size_t)k * p.S;
			const size_t lo = total * pc / pieces / 64 * 64, hi = pc + 1 == pieces ? total : total * (pc + 1) / pieces / 64 * 64;
			const size_t cp_hi = std::min(hi, p.plen);
			if (cp_hi > lo)
				std::memcpy(p.block.mut() + lo, src + lo, cp_hi - lo);
			const size_t z_lo = std::max(lo, p.plen);
			if (hi > z_lo)
				std::memset(p.block.mut() + z_lo, 0, hi - z_lo);
		});
	}
	if (oom)
		return fail_all(GBM_E_IO, "out of (pinned) host memory for the shard buffers");
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
		int rc;
		{
			DeviceTurn turn(gate);
			rc = gec_encode_hash_batch(mg->codec, gn, gd.data(), gl.data(), S, pp.data(), gsums.data());
		}
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
	// From here on shards of these blocks start to exist.  A block that nobody references yet (PutObject runs the
	// put and the block_ref incref concurrently, src/api/s3/put.rs:545-581) is protected for BLOCK_GC_DELAY exactly
	// like one whose count just dropped to zero -- and it is protected BEFORE its first shard is written, under the
	// hash's mutation lock: resync's delete branch re-reads the refcount under the same lock right before it
	// deletes, so it either finishes before this stamp (and the shards written below are new) or sees it.
	for (size_t b = 0; b < nb; ++b) {
		Hash h((const char *)hashes + 32 * b, 32);
		std::lock_guard<std::mutex> ml(mg->lock_mutate(h));
		gbm_manager::RcStripe &st = mg->rc_of(h);
		std::lock_guard<std::mutex> g(st.mu);
		RcEntry &e = st.map[h];
		if (e.kind != RcEntry::Present) {
			e.kind = RcEntry::Deletable;
			e.v = std::max(e.v, mg->now() + mg->gc_delay_ms.load());
		}
	}
	// fan-out: shard j of every block to nodes_of(hash)[j].  With order tags the blocks go out one after the
	// other in (stream, order) order -- requests of one stream reach a node in `order` order, whatever their
	// shard geometry; without tags the blocks are independent and go out from the pool's threads.
	std::vector<int> oks(nb, 0);
	// shards that a node parked beside a shard of another geometry (ShardResp::pending): committed once the block has
	// its quorum, dropped otherwise
	std::mutex parked_mu;
	std::vector<std::tuple<size_t, int, int>> parked;  // (block, shard idx, node)
	auto note_parked = [&](size_t b, int j, int node) {
		std::lock_guard<std::mutex> g(parked_mu);
		parked.emplace_back(b, j, node);
	};
	// (a tag array may hold untagged blocks -- the coalescing queue mixes requests: GBM_NO_STREAM marks them; they are
	// ordered behind the tagged ones and reach the nodes without a tag, like a put with order_tag = None)
	auto tag_of = [&](size_t b) -> const gbm_order_tag * { return tags && tags[b].stream_id != GBM_NO_STREAM ? &tags[b] : nullptr; };
	if (gate && gate->before)
		gate->before();
	auto fan_out = [&](size_t b) {
		Hash h((const char *)hashes + 32 * b, 32);
		std::vector<int> who;
		mg->nodes_of(h, who);
		const size_t S = prep[b].S;
		int ok = 0;
		for (int j = 0; j < n; ++j) {
			const Bytes payload = j < k ? prep[b].block.slice((size_t)j * S, S) : prep[b].parity.slice((size_t)(j - k) * S, S);
			bool pend = false;
			if (send_shard(mg, who[j], h, j, payload, S, prep[b].plen, prep[b].z, sums.data() + (b * n + j) * 32,
				       tag_of(b), &pend)) {
				++ok;
				mg->metrics[0] += S;
				if (pend)
					note_parked(b, j, who[j]);
			}
		}
		oks[b] = ok;
	};
	if (tags && gate && gate->block_order) {
		// several devices share the streams: one block at a time, in submission order, each behind its stream's previous
		// block (which another device's batch may hold)
		for (size_t b : *gate->block_order) {
			Hash h((const char *)hashes + 32 * b, 32);
			std::vector<int> who;
			mg->nodes_of(h, who);
			const size_t S = prep[b].S;
			std::atomic<int> okc{0};
			if (gate->before_block)
				gate->before_block(b);
			mg->pool->parallel_for((size_t)n, [&](size_t jj) {
				const int j = (int)jj;
				const Bytes payload = j < k ? prep[b].block.slice((size_t)j * S, S) : prep[b].parity.slice((size_t)(j - k) * S, S);
				bool pend = false;
				if (send_shard(mg, who[j], h, j, payload, S, prep[b].plen, prep[b].z, sums.data() + (b * n + j) * 32, tag_of(b), &pend)) {
					++okc;
					mg->metrics[0] += S;
					if (pend)
						note_parked(b, j, who[j]);
				}
			});
			if (gate->after_block)
				gate->after_block(b);
			oks[b] = okc.load();
		}
	} else if (tags) {
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
					bool pend = false;
					if (send_shard(mg, (int)node, h, j, payload, S, prep[b].plen, prep[b].z, sums.data() + (b * n + j) * 32, tag_of(b), &pend)) {
						++okc[b];
						mg->metrics[0] += S;
						if (pend)
							note_parked(b, j, (int)node);
					}
				}
			}
		});
		for (size_t b = 0; b < nb; ++b)
			oks[b] = okc[b].load();
	} else {
		mg->pool->parallel_for(nb, fan_out);
	}
	if (gate && gate->after)
		gate->after();
	tr.lap("fan-out");
	for (auto &pk : parked) {  // rare: the block existed with another geometry
		const size_t b = std::get<0>(pk);
		const Hash h((const char *)hashes + 32 * b, 32);
		ShardRpc rq{oks[b] >= mg->write_quorum ? RpcKind::CommitShard : RpcKind::AbortShard, &h, std::get<1>(pk), Shard(), nullptr};
		ShardResp rs;
		(void)mg->nodes[std::get<2>(pk)]->handle(rq, rs);
	}
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
		 int want_block_sums, std::vector<uint8_t> &block_sums, const std::function<void()> &overlap,
		 std::vector<uint8_t> *changed, const FanoutGate *gate, std::vector<uint8_t> *have_sum)
{
	const int k = mg->k, n = mg->n;
	const size_t nb = hs.size();
	block_sums.assign(want_block_sums ? nb * 32 : 0, 0);
	if (have_sum)
		have_sum->assign(nb, 0);
	// a block that cannot be read: MissingBlock when too few shards exist, CorruptData when shards were there but
	// failed their checks (Error::CorruptData is what the serving node's read_block_from answers, manager.rs:577-609)
	auto unreadable = [&](size_t b) { return g[b].corrupt_seen ? GBM_E_CORRUPT_DATA : GBM_E_MISSING_BLOCK; };
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
		helper = std::thread([&overlap] {
			name_thread("gbm-get-helper");
			overlap();
		});
	std::vector<uint8_t> todo(nb, 1);
	for (int round = 0; round <= n; ++round) {
		if (round == 1 && helper.joinable())
			helper.join();
		std::map<size_t, std::vector<size_t>> by_s;
		for (size_t b = 0; b < nb; ++b) {
			if (!todo[b])
				continue;
			if (!g[b].have_meta || g[b].count < k) {
				rcs[b] = unreadable(b);
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
			std::vector<uint8_t> ssums(ids.size() * (size_t)n * 32), bsums;
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
			const bool trip_sums = want_block_sums == 1 || (want_block_sums == 2 && nrebuild > 0);
			if (trip_sums)
				bsums.assign(ids.size() * 32, 0);
			int rc;
			{
				DeviceTurn turn(gate);
				rc = gec_decode_verify_batch(mg->codec, ids.size(), sp.data(), S, lens.data(), op.data(), ssums.data(),
							     trip_sums ? bsums.data() : nullptr);
			}
			tr.lap("decode+verify");
			if (helper.joinable())
				helper.join();  // the overlapped host work reads g: it must be done before the results below change it
			tr.lap("join overlapped assembly");
			if (rc)
				return ec_fail(rc, "gec_decode_verify_batch");
			mg->gpu_hashed += ids.size() * (size_t)k + (trip_sums ? ids.size() : 0);
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
						if (!confirmed_corrupt(mg, gb.shard[j].data(), S, gb.sum[j].data(), "gec_decode_verify_batch"))
							return fail(GBM_E_EC, "a read trip's shard checksums are not what the host computes: nothing was set aside, the read is refused");
						mg->metrics[2]++;
						if (gb.node[j] >= 0)
							mg->nodes[gb.node[j]]->mark_corrupted(hs[b], j);
						mg->put_to_resync(hs[b], 0);
						gb.shard[j] = Bytes();
						gb.count--;
						gb.corrupt_seen = true;
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
				if (trip_sums) {
					std::memcpy(block_sums.data() + 32 * b, bsums.data() + 32 * i, 32);
					if (have_sum)
						(*have_sum)[b] = 1;
				}
				rcs[b] = GBM_OK;
				todo[b] = 0;
			}
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
			rcs[b] = unreadable(b);
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

static int get_blocks_once(gbm_manager *mg, size_t nb, const uint8_t *hashes, const gbm_order_tag *tags, uint8_t *const *out,
			   const size_t *cap, size_t *len_out, int *rcs, bool raw, gbm_data_block_header *headers, const FanoutGate *gate);
static int get_blocks_again(gbm_manager *mg, size_t nb, const uint8_t *hashes, const gbm_order_tag *tags, uint8_t *const *out,
			    const size_t *cap, size_t *len_out, int *rcs, bool raw, gbm_data_block_header *headers, const FanoutGate *gate);

// raw == true: rpc_get_raw_block (stored bytes + header); false: rpc_get_block (plain bytes).
// While a layout change is being followed (more than one version is active) resync MOVES shards: PutShard to the new owner,
// then DeleteShard at the old one.  A read that asked the new owners before the puts and the old ones after the deletes
// finds the block "missing" although it was whole the whole time -- the reference's readers have the same window
// (block_read_nodes_of walks the versions in order, rpc_helper.rs:570-619) and leave it to the client's retry; with k
// holders to hear from instead of one it is wider here, so a block that comes back Missing during a transition is asked
// for again (twice at most): moves only go forward, a later walk meets the shards where an earlier one's came from.
int get_blocks_impl(gbm_manager *mg, size_t nb, const uint8_t *hashes, const gbm_order_tag *tags, uint8_t *const *out,
		    const size_t *cap, size_t *len_out, int *rcs, bool raw, gbm_data_block_header *headers, const FanoutGate *gate)
{
	if (!mg || (nb && (!hashes || !out || !cap || !len_out || !rcs)))
		return fail(GBM_E_INVALID_ARG, "NULL argument");
	DurationScope read_time(mg->bmx.read_duration);  // block.read_duration (metrics.rs:117-121): one observation per call
	int rc = get_blocks_once(mg, nb, hashes, tags, out, cap, len_out, rcs, raw, headers, gate);
	// (twice more at most, a millisecond and five apart: a slow reader beside a fast mover can lose several shards of one block to
	// the window in one walk; the mover is done with a block in well under that)
	for (int attempt = 1; attempt <= 2 && rc == GBM_OK && mg->layout_cur.load() != mg->layout_oldest.load(); ++attempt) {
		bool any = false;
		for (size_t b = 0; b < nb && !any; ++b)
			any = rcs[b] == GBM_E_MISSING_BLOCK || rcs[b] == GBM_E_CORRUPT_DATA;
		if (!any)
			break;
		if (attempt == 2)
			std::this_thread::sleep_for(std::chrono::milliseconds(5));
		else
			std::this_thread::sleep_for(std::chrono::milliseconds(1));
		rc = get_blocks_again(mg, nb, hashes, tags, out, cap, len_out, rcs, raw, headers, gate);
	}
	return rc;
}

// the blocks of a call that came back Missing (or Corrupt with too few shards found), asked for once more
static int get_blocks_again(gbm_manager *mg, size_t nb, const uint8_t *hashes, const gbm_order_tag *tags, uint8_t *const *out,
			    const size_t *cap, size_t *len_out, int *rcs, bool raw, gbm_data_block_header *headers, const FanoutGate *gate)
{
	int rc = GBM_OK;
	std::vector<size_t> again;
	for (size_t b = 0; b < nb; ++b)
		if (rcs[b] == GBM_E_MISSING_BLOCK || rcs[b] == GBM_E_CORRUPT_DATA)  // (Corrupt: a bad shard was met AND too few others were found)
			again.push_back(b);
	if (again.empty())
		return rc;
	const size_t na = again.size();
	std::vector<uint8_t> hh(na * 32);
	std::vector<gbm_order_tag> tt(tags ? na : 0);
	std::vector<uint8_t *> oo(na);
	std::vector<size_t> cc(na), ll(na, 0);
	std::vector<int> rr(na, GBM_E_MISSING_BLOCK);
	std::vector<gbm_data_block_header> hd(headers ? na : 0);
	for (size_t i = 0; i < na; ++i) {
		std::memcpy(hh.data() + 32 * i, hashes + 32 * again[i], 32);
		if (tags)
			tt[i] = tags[again[i]];
		oo[i] = out[again[i]];
		cc[i] = cap[again[i]];
	}
	rc = get_blocks_once(mg, na, hh.data(), tags ? tt.data() : nullptr, oo.data(), cc.data(), ll.data(), rr.data(), raw,
			     headers ? hd.data() : nullptr, gate);
	if (rc != GBM_OK)
		return rc;
	for (size_t i = 0; i < na; ++i) {
		rcs[again[i]] = rr[i];
		len_out[again[i]] = ll[i];
		if (headers)
			headers[again[i]] = hd[i];
	}
	return GBM_OK;
}

static int get_blocks_once(gbm_manager *mg, size_t nb, const uint8_t *hashes, const gbm_order_tag *tags, uint8_t *const *out,
			   const size_t *cap, size_t *len_out, int *rcs, bool raw, gbm_data_block_header *headers, const FanoutGate *gate)
{
	const int k = mg->k;
	std::vector<Hash> hs(nb);
	for (size_t b = 0; b < nb; ++b)
		hs[b].assign((const char *)hashes + 32 * b, 32);
	std::vector<Gathered> g;
	std::vector<uint8_t> block_sums, changed, have_sum, early(nb, 0);
	// The requester's end-to-end check (gbm_set_verify_block_hash): every Plain block, only the blocks that went through a
	// decode, or -- the default, the reference's read path -- none: the shard checksums of the same trip are the serving
	// node's verify (read_block_from, manager.rs:577-609), and they are always checked.
	const int mode = mg->verify_mode.load();
	const bool verify = mode != GBM_VERIFY_OFF, only_rebuilt = mode == GBM_VERIFY_REBUILT;
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
	Trace tr("get (whole call)");
	int frc = fetch_blocks(mg, hs, tags, g, rcs, verify && !cpu_hash ? (only_rebuilt ? 2 : 1) : 0, block_sums, assemble_early, &changed,
			       gate, &have_sum);
	if (frc)
		return frc;
	tr.lap("fetch");
	// assemble (parallel), then check every Plain block's content against its name (DataBlock::verify,
	// block.rs:69-77) -- all block hashes in one batch.  Plain blocks are assembled straight into the
	// caller's buffer and hashed from there (on CORRUPT_DATA its contents are unspecified); compressed
	// blocks go through an intermediate for the zstd frame, whose checksum is their verify.
	std::vector<uint8_t> hash_here(nb, 0);
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
		const bool check = verify && !z && (!only_rebuilt || (changed[b] & 2));
		if (check && !cpu_hash && have_sum[b] && std::memcmp(block_sums.data() + 32 * b, hashes + 32 * b, 32) != 0) {
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
		if (check && (cpu_hash || !have_sum[b])) {
			hash_here[b] = 1;  // checked below, eight blocks per core at a time
			return;
		}
		mg->metrics[5]++;
	});
	if (verify) {
		std::vector<size_t> idx;
		for (size_t b = 0; b < nb; ++b)
			if (hash_here[b])
				idx.push_back(b);
		// eight blocks per task where a core hashes eight chains at once (AVX-512); one block per task otherwise, so the
		// scalar fallback keeps one message per pool thread (8 x 1 MiB: 6.7 ms either way instead of 16 ms on one core)
		const size_t per = b2host::mb_available() ? 8 : 1;
		mg->pool->parallel_for((idx.size() + per - 1) / per, [&](size_t grp) {
			const size_t i0 = grp * per, cnt = std::min<size_t>(per, idx.size() - i0);
			const uint8_t *ptr[8];
			size_t len[8];
			uint8_t sums[8 * 32];
			for (size_t i = 0; i < cnt; ++i) {
				ptr[i] = out[idx[i0 + i]];
				len[i] = len_out[idx[i0 + i]];
			}
			b2host::blake2sum_many(ptr, len, cnt, sums);
			for (size_t i = 0; i < cnt; ++i) {
				const size_t b = idx[i0 + i];
				if (std::memcmp(sums + 32 * i, hashes + 32 * b, 32) != 0)
					rcs[b] = GBM_E_CORRUPT_DATA;
				else
					mg->metrics[5]++;
			}
		});
	}
	tr.lap("finish");
	g.clear();
	tr.lap("release");
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

}  // namespace gbmimpl

using namespace gbmimpl;

extern "C" {

int gbm_rpc_put_blocks(gbm_manager *mg, size_t nb, const uint8_t *hashes, const uint8_t *const *data, const size_t *len,
		       const uint8_t *prevent_compression, const gbm_order_tag *order_tags)
{
	// Large untagged batches go through in slices of 64 blocks on four threads, so that one slice's host work (the copy
	// into the shard buffers, the fan-out) runs while other slices are on the link: 512 x 1 MiB blocks 26 -> 38 GiB/s
	// (tools/bm_sweep.sh: 128 x 2 threads 33, 128 x 3 37, 64 x 4 38.6, 64 x 8 39.5).  Tagged batches keep their order.
	const size_t kSlice = env().put_slice;
	const int kThreads = env().put_threads;
	if (mg && mg->is_front() && nb && hashes && data && len) {
		// several devices: every block goes to the lane gec_device_of_hash names
		auto sub_put = [&](gbm_manager *lane, const std::vector<size_t> &ids) {
			const size_t cnt = ids.size();
			if (!cnt)
				return (int)GBM_OK;
			std::vector<uint8_t> hh(cnt * 32), pc(cnt, 0);
			std::vector<const uint8_t *> dd(cnt);
			std::vector<size_t> ll(cnt);
			std::vector<gbm_order_tag> tt(order_tags ? cnt : 0);
			for (size_t i = 0; i < cnt; ++i) {
				const size_t b = ids[i];
				std::memcpy(hh.data() + 32 * i, hashes + 32 * b, 32);
				dd[i] = data[b];
				ll[i] = len[b];
				if (prevent_compression)
					pc[i] = prevent_compression[b];
				if (order_tags)
					tt[i] = order_tags[b];
			}
			return gbm_rpc_put_blocks(lane, cnt, hh.data(), dd.data(), ll.data(), pc.data(), order_tags ? tt.data() : nullptr);
		};
		try {
			if (!order_tags) {
				const auto ids = split_by_lane(mg, nb, hashes);
				return for_lanes(mg, [&](gbm_manager *lane, size_t l) { return sub_put(lane, ids[l]); });
			}
			// Tagged blocks must reach every node in (stream, order) order whichever device encodes them: the batch is
			// walked in that order and cut into runs of consecutive blocks of one device, one put per run.  (The
			// coalescing queue, gbm_batcher_*, keeps both the order and the devices busy; this form is the simple one.)
			std::vector<size_t> order(nb);
			for (size_t i = 0; i < nb; ++i)
				order[i] = i;
			std::stable_sort(order.begin(), order.end(), [&](size_t x, size_t y) {
				return std::tie(order_tags[x].stream_id, order_tags[x].order) < std::tie(order_tags[y].stream_id, order_tags[y].order);
			});
			int result = GBM_OK;
			std::string err;
			for (size_t i = 0; i < nb;) {
				gbm_manager *lane = mg->route(hashes + 32 * order[i]);
				std::vector<size_t> run;
				for (; i < nb && mg->route(hashes + 32 * order[i]) == lane; ++i)
					run.push_back(order[i]);
				int rc = sub_put(lane, run);
				if (rc) {
					result = rc;
					err = last_error();
				}
			}
			return result ? fail(result, err) : GBM_OK;
		} catch (const std::exception &e) {
			return fail(GBM_E_IO, std::string("rpc_put_blocks: ") + e.what());
		}
	}
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
					err = last_error();  // the error text is thread-local: carry it to the caller's thread
				}
			}
		};
		std::vector<std::thread> others;
		for (int t = 1; t < kThreads; ++t)
			others.emplace_back([&run] {
				name_thread("gbm-put-slice");
				run();
			});
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
	if (m && hash)
		m = m->route(hash);  // one block: straight to its device's lane
	return gbm_rpc_put_blocks(m, 1, hash, d, &len, &pc, order_tag);
}

int gbm_rpc_get_blocks(gbm_manager *mg, size_t nb, const uint8_t *hashes, const gbm_order_tag *order_tags, uint8_t *const *out,
		       const size_t *cap, size_t *len_out, int *rcs)
{
	try {
		if (mg && mg->is_front() && nb && hashes && out && cap && len_out && rcs) {
			// several devices: every block is read, checked and decoded on the device that owns its hash
			const auto ids = split_by_lane(mg, nb, hashes);
			return for_lanes(mg, [&](gbm_manager *lane, size_t l) {
				const size_t cnt = ids[l].size();
				if (!cnt)
					return (int)GBM_OK;
				std::vector<uint8_t> hh(cnt * 32);
				std::vector<uint8_t *> oo(cnt);
				std::vector<size_t> cc(cnt), ll(cnt, 0);
				std::vector<int> rr(cnt, GBM_E_MISSING_BLOCK);
				std::vector<gbm_order_tag> tt(order_tags ? cnt : 0);
				for (size_t i = 0; i < cnt; ++i) {
					const size_t b = ids[l][i];
					std::memcpy(hh.data() + 32 * i, hashes + 32 * b, 32);
					oo[i] = out[b];
					cc[i] = cap[b];
					if (order_tags)
						tt[i] = order_tags[b];
				}
				int rc = get_blocks_impl(lane, cnt, hh.data(), order_tags ? tt.data() : nullptr, oo.data(), cc.data(), ll.data(),
							 rr.data(), false, nullptr);
				for (size_t i = 0; i < cnt; ++i) {
					len_out[ids[l][i]] = ll[i];
					rcs[ids[l][i]] = rc ? rc : rr[i];
				}
				return rc;
			});
		}
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
	if (m && hash)
		m = m->route(hash);
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
	if (m && hash)
		m = m->route(hash);
	try {
		rc = get_blocks_impl(m, 1, hash, order_tag, o, &cap, len_out, &rc1, true, header_out);
	} catch (const std::exception &e) {
		return fail(GBM_E_IO, std::string("rpc_get_raw_block: ") + e.what());
	}
	return rc ? rc : one_block_rc(rc1);
}

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
	std::vector<std::vector<uint8_t>> dig;      // per entry of `used`: nleaf leaf digests
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
		if (hi > lo)
			b2host::shardsum_leaf_range(shard[j].data(), S, lo, hi, dig[e].data());
		if (left[e].fetch_sub(1) == 1) {  // the shard's last piece: its root, its verdict
			uint8_t got[32];
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
	ck->nleaf = b2host::shardsum_nleaf(S);
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
		if (nd->handle(rq, rs) && rs.ok) {
			const ShardHeader &hd = rs.shard.hd;
			if (hd.version == 2 && hd.idx == j && hd.k == mk && hd.m == mm && hd.shard_len > 0 && hd.shard_len % 64 == 0 &&
			    rs.shard.data.n == hd.shard_len) {
				uint8_t sum[32];
				shardsum(rs.shard.data.data(), hd.shard_len, sum);
				if (std::memcmp(sum, hd.checksum, 32) == 0)
					v = 1;
			}
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
	m->nodes_of(hs[0], who);
	fast_fetch(m, fs, who, 0, k - 1);
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
	m->nodes_of(h, who);
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
