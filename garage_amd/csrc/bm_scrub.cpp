// bm_scrub.cpp -- ScrubWorker / RepairWorker (src/block/repair.rs): gbm_scrub (a given set of blocks), gbm_scrub_all
// (everything stored, batch by batch, one gec_verify_hash_batch trip each on the BACKGROUND-class codec, with
// leave-one-out location of a silently wrong shard), gbm_repair_all (queue everything for resync).
#include "bm_internal.hpp"

using namespace gbmimpl;

// RepairWorker (src/block/repair.rs:30-150): phase 1 queues every hash of the refcount table, phase 2 every hash that
// is actually stored somewhere ("blocks we are storing but don't actually need").
// every hash any reachable node holds a shard of: the nodes are walked side by side (a directory node's walk is one
// opendir per prefix directory -- 16 nodes x hundreds of directories, tens of milliseconds when done one after the other)
void gbmimpl::list_all_nodes(gbm_manager *mg, std::set<Hash> &all)
{
	std::vector<std::set<Hash>> per(mg->nodes.size());
	mg->pool->parallel_for(mg->nodes.size(), [&](size_t i) {
		if (!mg->nodes[i]->down.load())
			mg->nodes[i]->list(per[i]);
	});
	for (auto &s : per)
		for (const Hash &h : s)
			if (mg->owns(h))  // a lane of a multi-device manager walks the hashes of its own device only
				all.insert(h);
}

extern "C" {

int gbm_scrub(gbm_manager *mg, size_t nb, const uint8_t *hashes, uint8_t *bad_out)
{
	if (!mg || (nb && (!hashes || !bad_out)))
		return fail(GBM_E_INVALID_ARG, "NULL argument");
	if (mg->is_front()) {
		const auto ids = split_by_lane(mg, nb, hashes);
		return for_lanes(mg, [&](gbm_manager *lane, size_t l) {
			const size_t cnt = ids[l].size();
			if (!cnt)
				return (int)GBM_OK;
			std::vector<uint8_t> hh(cnt * 32), bad(cnt);
			for (size_t i = 0; i < cnt; ++i)
				std::memcpy(hh.data() + 32 * i, hashes + 32 * ids[l][i], 32);
			int rc = gbm_scrub(lane, cnt, hh.data(), bad.data());
			for (size_t i = 0; i < cnt; ++i)
				bad_out[ids[l][i]] = bad[i];
			return rc;
		});
	}
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
			int rc = gec_verify_batch(mg->bg_codec(), ids.size(), sp.data(), kv.first, ok.data());
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

int gbm_repair_all(gbm_manager *mg, size_t *queued)
{
	if (!mg)
		return fail(GBM_E_INVALID_ARG, "NULL manager");
	if (mg->is_front()) {
		std::vector<size_t> q(mg->lanes.size(), 0);
		int rc = for_lanes(mg, [&](gbm_manager *lane, size_t l) { return gbm_repair_all(lane, &q[l]); });
		if (queued) {
			*queued = 0;
			for (size_t v : q)
				*queued += v;
		}
		return rc;
	}
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
	if (gec_reconstruct_batch(mg->bg_codec(), n, sp.data(), op.data(), S, 0) != GEC_OK)
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
	if (mg->is_front()) {
		// one ScrubWorker per device, side by side: each walks the hashes its device owns on its own BACKGROUND codec
		std::vector<std::array<uint64_t, 4>> per(mg->lanes.size());
		int rc = for_lanes(mg, [&](gbm_manager *lane, size_t l) { return gbm_scrub_all(lane, batch_blocks, per[l].data()); });
		if (stats)
			for (int j = 0; j < 4; ++j) {
				stats[j] = 0;
				for (auto &p : per)
					stats[j] += p[j];
			}
		if (rc == GBM_OK)
			mg->scrub_last_complete_ms = mg->lanes[0]->now();
		return rc;
	}
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
					bt.err = last_error();  // thread-local: carried to the caller's thread
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
				const auto t_dev = std::chrono::steady_clock::now();
				int rc = gec_verify_hash_batch(mg->bg_codec(), ids.size(), sp.data(), kv.first, ok.data(), sums.data());
				++st[2];
				tr.lap("verify + checksums");
				if (const uint32_t tranq = mg->scrub_tranquility.load()) {  // Tranquilizer::tranquilize (tranquilizer.rs:38-69)
					const auto spent = std::chrono::steady_clock::now() - t_dev;
					std::this_thread::sleep_for(spent * tranq);
					mg->tranquilized_ms += (uint64_t)std::chrono::duration_cast<std::chrono::milliseconds>(spent * tranq).count();
				}
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
	for (auto &l : m->lanes)
		out[0] += l->scrub_corruptions.load();
	out[1] = m->scrub_last_complete_ms.load();
	return GBM_OK;
}

}  // extern "C"
