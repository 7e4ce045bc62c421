// bm_resync.cpp -- refcounts (RcEntry, src/block/rc.rs), the resync queue with its ErrorCounter back-off and
// background worker (src/block/resync.rs:170-337,513-648), and resync_block for a whole pass of blocks at once
// (:354-503) with the device work of all of them batched on the manager's BACKGROUND-class codec.
#include "bm_internal.hpp"

namespace gbmimpl {
namespace {

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
	// the geometry of the shard each current node holds (NeedShardReply carries the header): shards of a block are only
	// usable together when they were cut from the same payload the same way.  A node that was down while the block was
	// put again with another compression setting still holds a shard of the OLD geometry: present, readable, useless --
	// it is replaced like an absent one (PutShard parks the new shard beside it, CommitShard swaps them).
	std::vector<ShardHeader> geo;
	std::vector<uint8_t> have_geo;
	bool mixed = false;                // the shards in place are not all of one geometry
	std::vector<int> replace;          // of `want`: a shard of another geometry is in place
	Gathered g;
};

void scan_block(gbm_manager *mg, ResyncTask &t)
{
	const int n = mg->n;
	const int vcur = mg->layout_cur.load(), vold = mg->layout_oldest.load();
	mg->nodes_of(t.h, vcur, t.who);
	t.present_cur.assign(n, 0);
	t.reachable.assign(n, 0);
	t.geo.assign(n, ShardHeader());
	t.have_geo.assign(n, 0);
	int first = -1;
	for (int j = 0; j < n; ++j) {
		ShardRpc rq{RpcKind::NeedShardQuery, &t.h, j, Shard(), nullptr};
		ShardResp rs;
		if (mg->nodes[t.who[j]]->handle(rq, rs)) {
			t.reachable[j] = 1;
			t.present_cur[j] = rs.needed ? 0 : 1;
			t.exists = t.exists || !rs.needed;
			if (rs.have_hd && rs.shard.hd.version >= 2) {
				t.geo[j] = rs.shard.hd;
				t.have_geo[j] = 1;
				if (first < 0)
					first = j;
				else if (!t.geo[j].same_geometry(t.geo[first]))
					t.mixed = true;
			}
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
			// block is needed by nobody, so NeedShardQuery has no taker and the offload set is empty.
			// The refcount read by the presence scan may be stale by now (the scan of a 1024-block pass takes a
			// while): it is read AGAIN under the hash's mutation lock, and the shards are deleted under that lock
			// (delete_if_unneeded re-checks under lock_mutate, manager.rs:619-623,821-830).  A put of the same hash
			// stamps its protection under the same lock before it writes its first shard (put_blocks_impl).
			std::lock_guard<std::mutex> ml(mg->lock_mutate(t.h));
			t.rc = mg->get_rc(t.h);
			if (!t.rc.is_deletable(mg->now()))
				return;  // referenced, or put, in the meantime: nothing to delete; stragglers are queued by the put
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
				shardsum_v(rs.shard.hd.version, rs.shard.data.data(), rs.shard.data.n, sum);  // (a stray keeps its header, whatever its version)
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
		if (!tasks[i].want.empty() || (tasks[i].mixed && tasks[i].rc.is_needed(now)))
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
		int grc = gather_many(mg, hs, nullptr, k, gs, verify_in_gather, nullptr, /*migrate=*/true);
		tr.lap(verify_in_gather ? "gather k + checksums" : "gather k");
		for (size_t q = 0; q < todo.size(); ++q) {
			ResyncTask &t = tasks[todo[q]];
			if (grc) {
				t.error = std::string("gather: ") + last_error();
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
			// shards of another geometry than the one the block is read in (the gather's choice: the largest consistent
			// group) are stale: they are replaced.  Where the scan could not read a header, the gather's own verdict decides.
			if (t.mixed || t.g.mixed)
				for (int j = 0; j < n; ++j) {
					if (!t.reachable[j] || !t.g.shard[j].empty() || std::find(t.want.begin(), t.want.end(), j) != t.want.end())
						continue;
					if (t.have_geo[j] ? !t.geo[j].same_geometry(t.g.meta) : t.g.mixed) {
						t.want.push_back(j);
						t.replace.push_back(j);
					}
				}
		}
		// A wanted shard that the gather has brought in all the same -- from another holder than the one the scan asked: a
		// previous layout version's node that was unreachable a moment ago, a stray the offload branch could not fetch -- is not
		// rebuilt, it is handed to its owner as it is (its checksum is verified first: nobody has looked at it yet).  It must
		// not reach the trip as "present AND wanted": the codec skips a block nothing is wanted of, its in_sums stay whatever
		// they were, and comparing those with the headers set aside all k good shards of the block.
		for (size_t i : todo) {
			ResyncTask &t = tasks[i];
			if (!t.g.have_meta || t.want.empty())  // (an error noted by the scan -- a node that is down -- does not stop this: the
				continue;                      // task is tried again later, what can be done now is done now)
			const std::string noted = t.error;
			t.error.clear();
			for (auto it = t.want.begin(); it != t.want.end();) {
				const int j = *it;
				if (t.g.shard[j].empty()) {
					++it;
					continue;
				}
				uint8_t sum[32];
				shardsum_v(mg->sumver, t.g.shard[j].data(), t.g.meta.shard_len, sum);  // (the gather carries every shard in the manager's version)
				if (std::memcmp(sum, t.g.sum[j].data(), 32) != 0) {
					mg->metrics[2]++;
					if (t.g.node[j] >= 0)
						mg->nodes[t.g.node[j]]->mark_corrupted(t.h, j);
					t.error = "a shard fetched for hand-over does not match its checksum";  // the block is tried again: fewer holders now
					break;
				}
				bool pend = false;
				if (send_shard(mg, t.who[j], t.h, j, t.g.shard[j], t.g.meta.shard_len, t.g.meta.orig_len, t.g.meta.compressed != 0,
					       t.g.sum[j].data(), nullptr, &pend)) {
					if (pend) {
						ShardRpc cq{RpcKind::CommitShard, &t.h, j, Shard(), nullptr};
						ShardResp cs;
						(void)mg->nodes[t.who[j]]->handle(cq, cs);
					}
					++t.changed;
					++st.offloaded;
				} else {
					t.error = "PutShard of a shard in hand to its owner failed";
				}
				it = t.want.erase(it);
			}
			if (!t.error.empty())
				t.want.clear();
			else
				t.error = noted;
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
			const auto t_dev = std::chrono::steady_clock::now();
			int rc = oom ? GEC_E_NOMEM
				     : gec_reconstruct_hash_batch(mg->bg_codec(), ids.size(), sp.data(), op.data(), S, 0, in_sums.data(), out_sums.data());
			tr.lap("reconstruct + checksums");
			if (const uint32_t tranq = mg->resync_tranquility.load()) {  // Tranquilizer::tranquilize (tranquilizer.rs:38-69)
				const auto spent = std::chrono::steady_clock::now() - t_dev;
				std::this_thread::sleep_for(spent * tranq);
				mg->tranquilized_ms += (uint64_t)std::chrono::duration_cast<std::chrono::milliseconds>(spent * tranq).count();
			}
			++st.device_calls;
			if (rc) {
				ec_fail(rc, "gec_reconstruct_hash_batch");
				for (size_t i : ids)
					tasks[i].error = last_error();
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
							std::string who = "gec_reconstruct_hash_batch (shard " + std::to_string(j) + ", in hand:";
							for (int x = 0; x < n; ++x)
								if (!t.g.shard[x].empty())
									who += " " + std::to_string(x);
							who += "; wanted:";
							for (int x : t.want)
								who += " " + std::to_string(x);
							who += ")";
							if (!confirmed_corrupt(mg, t.g.shard[j].data(), S, t.g.sum[j].data(), who.c_str())) {
								t.error = "the rebuild trip's shard checksums are not what the host computes: nothing was set aside";
								good[q] = 0;
								t.want.clear();
								break;
							}
							mg->metrics[2]++;
							if (t.g.node[j] >= 0)
								mg->nodes[t.g.node[j]]->mark_corrupted(t.h, j);
							good[q] = 0;
						}
					}
					if (!good[q] && t.error.empty()) {
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
					bool pend = false;
					if (send_shard(mg, t.who[j], t.h, j, outb[q][j], S, t.g.meta.orig_len, t.g.meta.compressed != 0,
						       out_sums.data() + (q * n + j) * 32, nullptr, &pend)) {
						if (pend) {
							// a shard of another geometry was in place: the node parked the new one beside it; the block is
							// being read in the new shard's geometry (>= k shards of it exist), so it takes the old one's place
							ShardRpc cq{RpcKind::CommitShard, &t.h, j, Shard(), nullptr};
							ShardResp cs;
							if (!mg->nodes[t.who[j]]->handle(cq, cs) || !cs.ok) {
								t.error = "CommitShard of a rebuilt shard failed";
								continue;
							}
						}
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
}  // namespace gbmimpl

using namespace gbmimpl;

extern "C" {

int gbm_block_incref(gbm_manager *m, const uint8_t hash[32])
{
	if (!m || !hash)
		return fail(GBM_E_INVALID_ARG, "NULL argument");
	m = m->route(hash);
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
	m = m->route(hash);
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
	RcEntry e = m->route(hash)->get_rc(Hash((const char *)hash, 32));
	out[0] = e.kind == RcEntry::Present ? e.v : 0;
	out[1] = e.kind;
	out[2] = e.kind == RcEntry::Deletable ? e.v : 0;
	return GBM_OK;
}

int gbm_put_to_resync(gbm_manager *m, const uint8_t hash[32], uint64_t delay_ms)
{
	if (!m || !hash)
		return fail(GBM_E_INVALID_ARG, "NULL argument");
	m->route(hash)->put_to_resync(Hash((const char *)hash, 32), delay_ms);
	return GBM_OK;
}

// resync_counter / resync_error_counter (resync.rs:298-302), blocks sent and received (:441-497), blocks deleted
// (manager.rs:827) -- per shard here: this manager stands in front of all its nodes
static void note_resync(gbm_manager *mg, const ResyncStats &st)
{
	mg->bmx.resync_counter += st.taken;
	mg->bmx.resync_error_counter += st.errors;
	mg->bmx.resync_send_counter += st.offloaded;
	mg->bmx.resync_recv_counter += st.rebuilt;
	mg->bmx.delete_counter += st.deleted;
}

// One pass of resync_iter over everything that is due (resync.rs:255-337).
int gbm_resync_run(gbm_manager *mg, size_t max_blocks, uint64_t stats[8])
{
	if (!mg)
		return fail(GBM_E_INVALID_ARG, "NULL manager");
	if (mg->is_front()) {
		// every device's queue side by side, each on its own BACKGROUND codec; max_blocks is shared out evenly
		const size_t nl = mg->lanes.size(), share = max_blocks ? (max_blocks + nl - 1) / nl : 0;
		std::vector<std::array<uint64_t, 8>> per(nl);
		int rc = for_lanes(mg, [&](gbm_manager *lane, size_t i) { return gbm_resync_run(lane, share, per[i].data()); });
		if (stats)
			for (int j = 0; j < 8; ++j) {
				stats[j] = 0;
				for (auto &p : per)
					stats[j] += p[j];
			}
		return rc;
	}
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
			if (mg->rs_busy.count(h)) {  // another worker's pass has this block in hand (get_block_to_resync, resync.rs:339-352)
				++it;
				continue;
			}
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
		for (auto &t : taken)
			mg->rs_busy.insert(t.second);
	}
	st.taken = taken.size();
	tasks.resize(taken.size());
	for (size_t i = 0; i < taken.size(); ++i)
		tasks[i].h = taken[i].second;
	int result = GBM_OK;
	{
		DurationScope pass_time(mg->bmx.resync_duration);  // block.resync_duration (resync.rs:290-296): one pass over what is due
		if (tasks.empty())
			pass_time.cancel();
		try {
			resync_blocks(mg, tasks, st);
		} catch (const std::exception &e) {
			for (auto &t : tasks)
				if (t.error.empty())
					t.error = e.what();
		}
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
		for (auto &t : taken)  // BusyBlock's drop (resync.rs:506-511)
			mg->rs_busy.erase(t.second);
	}
	if (!taken.empty())
		mg->rs_cv.notify_all();  // entries another worker had to pass over may be taken now
	note_resync(mg, st);
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
	mg = mg->route(hash);
	std::vector<ResyncTask> tasks(1);
	tasks[0].h.assign((const char *)hash, 32);
	ResyncStats st;
	st.taken = 1;
	{
		DurationScope one_time(mg->bmx.resync_duration);
		try {
			resync_blocks(mg, tasks, st);
		} catch (const std::exception &e) {
			st.errors = 1;
			note_resync(mg, st);
			return fail(GBM_E_IO, std::string("resync_block: ") + e.what());
		}
	}
	st.errors = tasks[0].error.empty() ? 0 : 1;
	note_resync(mg, st);
	if (changed)
		*changed = tasks[0].changed;
	if (!tasks[0].error.empty())
		return fail(tasks[0].error.rfind("Missing block", 0) == 0 ? GBM_E_MISSING_BLOCK : GBM_E_IO, tasks[0].error);
	return GBM_OK;
}

// true once no pass of any worker has a block of this manager (or of its lanes) in hand; waits for that at most wait_ms
static bool resync_passes_over(gbm_manager *mg, uint64_t wait_ms)
{
	auto one = [&](gbm_manager *x) {
		std::unique_lock<std::mutex> lk(x->rs_mu);
		return x->rs_cv.wait_until(lk, std::chrono::system_clock::now() + std::chrono::milliseconds(wait_ms), [&] { return x->rs_busy.empty(); });
	};
	bool over = one(mg);
	for (auto &l : mg->lanes)
		over = one(l.get()) && over;
	return over;
}

// gbm_resync_run until nothing is due any more -- and no OTHER worker's pass is still under way: a background worker takes
// its entries out of the queue for the length of its pass, so an empty queue alone does not mean that everything due has
// been done (a caller that trims the layout on the strength of this call would strand the shards such a pass was about to
// move).
int gbm_resync_all(gbm_manager *mg, int *changed)
{
	if (!mg)
		return fail(GBM_E_INVALID_ARG, "NULL manager");
	int total = 0, result = GBM_OK;
	for (int round = 0; round < 256; ++round) {
		uint64_t st[8];
		int rc = gbm_resync_run(mg, 0, st);
		if (rc)
			result = rc;
		total += (int)(st[4] + st[5] + st[6]);
		if (st[0] != 0)
			continue;
		if (resync_passes_over(mg, 0))
			break;  // nothing was due and nothing is in flight
		(void)resync_passes_over(mg, 1000);  // another worker's pass: what it leaves behind (errors re-queued, follow-ups) is looked at next
	}
	if (changed)
		*changed = total;
	return result;
}

size_t gbm_resync_queue_len(const gbm_manager *m)
{
	if (!m)
		return 0;
	size_t total = 0;
	for (auto &l : m->lanes)
		total += gbm_resync_queue_len(l.get());
	std::lock_guard<std::mutex> lk(m->rs_mu);
	return total + m->rs_queue.size();
}

size_t gbm_resync_errors_len(const gbm_manager *m)
{
	if (!m)
		return 0;
	size_t total = 0;
	for (auto &l : m->lanes)
		total += gbm_resync_errors_len(l.get());
	std::lock_guard<std::mutex> lk(m->rs_mu);
	return total + m->rs_errors.size();
}

int gbm_list_resync_errors(gbm_manager *m, gbm_resync_error_info *out, size_t cap, size_t *n_out)
{
	if (!m || !n_out || (cap && !out))
		return fail(GBM_E_INVALID_ARG, "NULL argument");
	try {
		std::vector<gbm_resync_error_info> all;
		auto one = [&](gbm_manager *x) {
			const uint64_t base = x->retry_delay_ms.load();
			const size_t first = all.size();
			{
				std::lock_guard<std::mutex> g(x->rs_mu);
				for (auto &kv : x->rs_errors) {
					gbm_resync_error_info e{};
					std::memcpy(e.hash, kv.first.data(), 32);
					e.error_count = kv.second.errors;
					e.last_try_ms = kv.second.last_try;
					e.next_try_ms = kv.second.next_try(base);
					all.push_back(e);
				}
			}
			for (size_t i = first; i < all.size(); ++i) {  // (outside the queue's lock, as the reference's second loop)
				const RcEntry rc = x->get_rc(Hash((const char *)all[i].hash, 32));
				all[i].refcount = rc.kind == RcEntry::Present ? rc.v : 0;  // RcEntry::as_u64 (rc.rs:234-240)
			}
		};
		one(m);
		for (auto &l : m->lanes)
			one(l.get());
		std::sort(all.begin(), all.end(),
			  [](const gbm_resync_error_info &a, const gbm_resync_error_info &b) { return std::memcmp(a.hash, b.hash, 32) < 0; });
		*n_out = all.size();
		std::copy(all.begin(), all.begin() + (ptrdiff_t)std::min(cap, all.size()), out);
	} catch (const std::exception &e) {
		return fail(GBM_E_IO, std::string("gbm_list_resync_errors: ") + e.what());
	}
	return GBM_OK;
}

int gbm_resync_clear_backoff(gbm_manager *m, const uint8_t hash[32])
{
	if (!m || !hash)
		return fail(GBM_E_INVALID_ARG, "NULL argument");
	m = m->route(hash);
	const Hash h((const char *)hash, 32);
	const uint64_t now = m->now(), base = m->retry_delay_ms.load();
	{
		std::lock_guard<std::mutex> g(m->rs_mu);
		auto it = m->rs_errors.find(h);
		if (it == m->rs_errors.end() || it->second.errors == 0)
			return fail(GBM_E_INVALID_ARG, "Block " + hex(h) + " was not in an errored state");
		const uint64_t d = it->second.delay_ms(base);
		it->second.last_try = now > d ? now - d : 0;
		m->rs_queue.insert({now, h});
	}
	m->rs_cv.notify_all();
	return GBM_OK;
}

static void resync_worker_loop(gbm_manager *m)
{
	std::unique_lock<std::mutex> lk(m->rs_mu);
	while (!m->rs_worker_stop) {
		const uint64_t now = m->now();
		bool due = false;  // something is due that no other worker has in hand
		for (auto it = m->rs_queue.begin(); it != m->rs_queue.end() && it->first <= now && !due; ++it)
			due = !m->rs_busy.count(it->second);
		if (due) {
			lk.unlock();
			(void)gbm_resync_run(m, 1024, nullptr);
			lk.lock();
			continue;
		}
		// idle until the first entry is due, something is queued, another worker's pass ends, or 10 s pass (resync.rs:325-336)
		uint64_t wait_ms = 10000;
		if (!m->rs_queue.empty() && m->rs_queue.begin()->first > now)
			wait_ms = std::min<uint64_t>(wait_ms, m->rs_queue.begin()->first - now);
		m->rs_cv.wait_until(lk, std::chrono::system_clock::now() + std::chrono::milliseconds(std::max<uint64_t>(wait_ms, 1)));
	}
}

int gbm_resync_worker_start(gbm_manager *m)
{
	if (!m)
		return fail(GBM_E_INVALID_ARG, "NULL manager");
	if (m->is_front()) {  // ResyncWorkers per device queue
		for (auto &l : m->lanes)
			(void)gbm_resync_worker_start(l.get());
		return GBM_OK;
	}
	try {
		std::lock_guard<std::mutex> g(m->rs_mu);
		if (!m->rs_workers.empty())
			return GBM_OK;
		m->rs_worker_stop = false;
		for (int i = 0; i < m->rs_n_workers; ++i)
			m->rs_workers.emplace_back([m] {
				lane_thread("gbm-resync", m->codec);
				resync_worker_loop(m);
			});
	} catch (const std::exception &e) {
		return fail(GBM_E_IO, std::string("gbm_resync_worker_start: ") + e.what());
	}
	return GBM_OK;
}

int gbm_resync_worker_stop(gbm_manager *m)
try {
	if (!m)
		return fail(GBM_E_INVALID_ARG, "NULL manager");
	for (auto &l : m->lanes)
		(void)gbm_resync_worker_stop(l.get());
	std::vector<std::thread> ts;
	{
		std::lock_guard<std::mutex> g(m->rs_mu);
		if (m->rs_workers.empty())
			return GBM_OK;
		m->rs_worker_stop = true;
		ts.swap(m->rs_workers);
	}
	m->rs_cv.notify_all();
	for (auto &t : ts)
		t.join();
	return GBM_OK;
}
GBM_CATCH

// ResyncPersistedConfig (resync.rs:58-71): this library's own 16-byte little-endian record, <path>.tmp + rename
static void resync_config_save(gbm_manager *m)
{
	std::string path;
	uint32_t n;
	{
		std::lock_guard<std::mutex> g(m->rs_mu);
		path = m->rs_cfg_path;
		n = (uint32_t)m->rs_n_workers;
	}
	if (path.empty())
		return;
	const uint32_t tq = m->resync_tranquility.load();
	uint8_t b[16] = {'G', 'B', 'M', 'r', 'c', 'f', 'g', '1'};
	for (int i = 0; i < 4; ++i) {
		b[8 + i] = (uint8_t)(n >> (8 * i));
		b[12 + i] = (uint8_t)(tq >> (8 * i));
	}
	const std::string tmp = path + ".tmp";
	bool ok = false;
	if (FILE *f = std::fopen(tmp.c_str(), "wb")) {
		ok = std::fwrite(b, 1, sizeof(b), f) == sizeof(b);
		ok = (std::fclose(f) == 0) && ok;
	}
	if (!ok || std::rename(tmp.c_str(), path.c_str()) != 0) {
		std::remove(tmp.c_str());
		std::fprintf(stderr, "garage_block: could not save the resync configuration to %s\n", path.c_str());
	}
}

int gbm_set_resync_workers(gbm_manager *m, int n_workers)
{
	if (!m)
		return fail(GBM_E_INVALID_ARG, "NULL manager");
	if (n_workers < 1 || n_workers > GBM_MAX_RESYNC_WORKERS)
		return fail(GBM_E_INVALID_ARG, "Invalid number of resync workers, must be between 1 and " + std::to_string(GBM_MAX_RESYNC_WORKERS));
	auto one = [&](gbm_manager *x) {
		bool running;
		{
			std::lock_guard<std::mutex> g(x->rs_mu);
			running = !x->rs_workers.empty();
			if (x->rs_n_workers == n_workers)
				return;
			x->rs_n_workers = n_workers;
		}
		if (running && !x->is_front()) {
			(void)gbm_resync_worker_stop(x);
			(void)gbm_resync_worker_start(x);
		}
	};
	one(m);
	for (auto &l : m->lanes)
		one(l.get());
	resync_config_save(m);
	return GBM_OK;
}

int gbm_get_resync_workers(const gbm_manager *m)
{
	if (!m)
		return 0;
	std::lock_guard<std::mutex> g(m->rs_mu);
	return m->rs_n_workers;
}

int gbm_resync_config_persist(gbm_manager *m, const char *path)
{
	if (!m || !path || !*path)
		return fail(GBM_E_INVALID_ARG, "NULL argument");
	uint8_t b[17];
	size_t n = 0;
	if (FILE *f = std::fopen(path, "rb")) {
		n = std::fread(b, 1, sizeof(b), f);
		std::fclose(f);
	}
	{
		std::lock_guard<std::mutex> g(m->rs_mu);
		m->rs_cfg_path = path;
	}
	auto u32 = [&](int off) { return (uint32_t)b[off] | (uint32_t)b[off + 1] << 8 | (uint32_t)b[off + 2] << 16 | (uint32_t)b[off + 3] << 24; };
	if (n == 16 && std::memcmp(b, "GBMrcfg1", 8) == 0 && u32(8) >= 1 && u32(8) <= GBM_MAX_RESYNC_WORKERS) {
		// the persisted values win (PersisterShared::new, persister.rs:97-101)
		const int rc = gbm_set_tranquility(m, -1, (int)std::min<uint32_t>(u32(12), 0x7fffffff));
		if (rc)
			return rc;
		return gbm_set_resync_workers(m, (int)u32(8));
	}
	// no record (or one that does not decode): ResyncPersistedConfig::default, unless the caller has chosen already
	if (!m->resync_tranquility_set.load())
		(void)gbm_set_tranquility(m, -1, GBM_INITIAL_RESYNC_TRANQUILITY);
	resync_config_save(m);
	return GBM_OK;
}

}  // extern "C"

void gbmimpl::resync_config_changed(gbm_manager *mg)
{
	resync_config_save(mg);
}
