// bm_gather.cpp -- the shard gather every read-side path starts with (rpc_get_raw_block_internal's probing of the holders,
// src/block/manager.rs:276-339; block_read_nodes_of, src/rpc/rpc_helper.rs:570-619; the hedged form after try_call_many_inner,
// :323-411), and PutShard to one node.
#include "bm_internal.hpp"

namespace gbmimpl {

// block_read_nodes_of + request_order (src/rpc/rpc_helper.rs:570-660) applied to SHARDS.  The reference sorts the nodes that
// may hold a block by (is another node, is another zone, avg ping) and asks "the preferred node in all layout versions (older
// to newer), then the second preferred one in all versions", itself first.  Here every holder has a different shard and a read
// needs k of them, so the same order decides WHICH k are asked: the requester's own shard, then the same zone's, then the lowest
// pings -- a near parity shard and a 0.1 ms decode instead of a far data shard and a WAN round trip; hedged reads go on to
// the next nearest.  Ties keep shard-index order (data before parity: no decode when nothing distinguishes the holders), so a
// manager that was told nothing about zones and pings asks as it always did.  A node's ping is what gbm_node_set_ping says
// (unknown = 10 s, the reference's default, rpc_helper.rs:641).
void read_candidate_order(const gbm_manager *mg, const Hash &h, int vold, int vcur, std::vector<uint32_t> &order)
{
	const int n = mg->n, nver = vcur - vold + 1;
	const int self = mg->self_node.load(), our_zone = mg->self_zone.load();
	struct Key {
		bool other_node, other_zone;
		uint64_t ping;
		int node, j;
	};
	std::vector<std::vector<Key>> ver(nver);
	std::vector<int> who;
	for (int v = 0; v < nver; ++v) {
		mg->nodes_of(h, vold + v, who);
		ver[v].resize(n);
		for (int j = 0; j < n; ++j) {
			const Node &nd = *mg->nodes[who[j]];
			const uint64_t ping = nd.ping_us.load(std::memory_order_relaxed);
			ver[v][j] = Key{who[j] != self, nd.zone.load(std::memory_order_relaxed) != our_zone, ping ? ping : 10000000ull, who[j], j};
		}
		std::stable_sort(ver[v].begin(), ver[v].end(), [](const Key &a, const Key &b) {
			return std::tie(a.other_node, a.other_zone, a.ping) < std::tie(b.other_node, b.other_zone, b.ping);
		});
	}
	order.clear();
	order.reserve((size_t)nver * n);
	std::vector<std::pair<int, int>> seen;  // (node, shard): the same request under two versions is made once
	for (int i = 0; i < n; ++i)
		for (int v = 0; v < nver; ++v) {
			const Key &c = ver[v][i];
			const std::pair<int, int> what(c.node, c.j);
			if (std::find(seen.begin(), seen.end(), what) != seen.end())
				continue;
			seen.push_back(what);
			const uint32_t idx = (uint32_t)(v * n + c.j);
			if (nver > 1 && c.node == self)
				order.insert(order.begin(), idx);  // "it's always fast (almost free) to ask locally" (rpc_helper.rs:594-597)
			else
				order.push_back(idx);
		}
}

// Fetch shards until every block has `want` valid ones of one geometry in hand (or ran out of nodes): shard
// index order within the current layout version, then older versions (block_read_nodes_of interleaves
// versions the same way, rpc_helper.rs:570-619).  The checksums of each round's candidates are verified in
// ONE batch; a shard whose checksum or header does not match is treated as missing, renamed *.corrupted and
// queued for resync (read_block_from's behaviour, manager.rs:577-609), and the next node is tried in the
// following round.
int gather_many(gbm_manager *mg, const std::vector<Hash> &hs, const gbm_order_tag *tags, int want, std::vector<Gathered> &gs,
		bool verify, const std::vector<uint8_t> *only, bool migrate)
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
		if (hd.version < 1 || hd.version > 3)
			return false;  // a shard format this build does not know: unreadable for us, but left alone (never renamed)
		bool ok = hd.idx == j && hd.k == mg->k && hd.m == mg->m && sh.data.n == hd.shard_len && hd.shard_len > 0 &&
			  hd.shard_len % 64 == 0;
		if (ok && hd.version != mg->sumver) {
			// A shard of another format than this manager writes -- round 1's plain blake2sum, rounds 2-4's BLAKE2b tree in a
			// store that has moved on to MLH64 (or the reverse, a v2 manager over a newer store).  Verified here, on the host,
			// with ITS version's checksum (each such shard is read this way once), then carried -- and rewritten on its node --
			// in the manager's version, so that everything downstream sees one format and the store migrates as it is read.
			uint8_t sum[32];
			shardsum_v(hd.version, sh.data.data(), hd.shard_len, sum);
			ok = std::memcmp(sum, hd.checksum, 32) == 0;
			if (ok) {
				const int was = hd.version;
				hd.version = (uint8_t)mg->sumver;
				shardsum_v(mg->sumver, sh.data.data(), hd.shard_len, hd.checksum);
				// The REWRITE on its node is maintenance: scrub and resync do it (`migrate`), and only UPWARDS (an older header
				// version into this manager's newer one) -- a manager of the older kind reads newer shards and leaves them as
				// they are, so two managers of different kinds over one store converge instead of rewriting each other's shards
				// at every scrub.  A read rewrites (either way) only when the operator asked for it (gbm_set_migrate_on_read):
				// a get has no business writing to a store.
				if ((migrate && was < mg->sumver) || mg->migrate_on_read.load()) {
					ShardRpc up{RpcKind::PutShard, &hs[b], j, sh, nullptr};
					ShardResp ur;
					if (mg->nodes[node]->handle(up, ur) && ur.ok)
						mg->shards_migrated++;
				}
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
		if (g.tried.size() != ncand) {
			g.tried.assign(ncand, 0);
			read_candidate_order(mg, hs[b], vold, vcur, g.order);
		}
		// a candidate is consumed when it is ASKED, not when it is passed over: shard j being covered by a request that is
		// still in flight says nothing about j's other holders, which are needed the moment that request fails (a hedge
		// timer that fired while all n first requests were in flight used to use up every older-version candidate)
		for (const uint32_t c : g.order) {
			if (g.tried[c])
				continue;
			// The order is read_candidate_order's: nearest holders first, and for one shard its holder in the OLDEST layout version
			// before the newer ones' -- block_read_nodes_of's order (rpc_helper.rs:559-563, 583-603: "ask the preferred node in all
			// layout versions (older to newer)", because most blocks were saved before the change).  Here that is also what makes a
			// read safe beside the mover: resync moves a shard with PutShard to its new owner and only then DeleteShard at the old
			// one, so whoever asks the OLD holder first cannot miss a shard in motion -- it is either still there, or already at the
			// new owner by the time that one is asked.  Newest-first (rounds 2 - 3) had a window: new owner asked before the put,
			// old one after the delete; with k holders to hear from instead of one that window made whole blocks read as Missing
			// under the soak.  The price, for the length of the transition: one missed request for every shard that has already moved.
			const int v = vold + (int)(c / n), j = (int)(c % n);
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
					if (!mg->nodes[who[j]]->handle(rq, rs)) {
						g.down_seen = true;  // a holder that could not be asked: the shard may well be there
						continue;
					}
					if (!rs.ok)
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
				std::atomic<bool> unreachable{false};  // its holder was down (not: the round ended before it was asked)
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
						bool answered = false;
						try {
							if (!rd->over.load()) {
								answered = nd->handle(rq, rs);
								f->unreachable = !answered;
							}
						} catch (...) {  // (out of memory for the shard's copy, as a rule) a holder that did not answer
							rs = ShardResp();
						}
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
				if (f->unreachable.load())
					gs[f->b].down_seen = true;
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
	hd.version = (uint8_t)mg->sumver;
	if (checksum)
		std::memcpy(hd.checksum, checksum, 32);
	else
		shardsum_v(mg->sumver, payload.data(), S, hd.checksum);
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

}  // namespace gbmimpl
