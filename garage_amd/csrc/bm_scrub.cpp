// bm_scrub.cpp -- ScrubWorker / RepairWorker (src/block/repair.rs): gbm_scrub (a given set of blocks), gbm_scrub_all
// (everything stored, batch by batch, one gec_verify_hash_batch trip each on the BACKGROUND-class codec, with
// leave-one-out location of a silently wrong shard), gbm_repair_all (queue everything for resync).
#include "bm_internal.hpp"

#include <cstdio>
#include <random>

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

namespace {

// Which single shard of an RS-inconsistent stripe is the wrong one?  For every candidate j the stripe is re-derived
// from the first k of the OTHER shards; the candidate is the culprit iff all the others then agree with what is
// stored (needs m >= 2; with m == 1 the block's own name decides: locate_by_content).  One gec_reconstruct_batch call: the n candidates are n "blocks" with n erasure patterns.
// With ONE parity shard there is nothing to compare a re-derived stripe with -- but the block's name is the hash of its
// plain bytes (and a compressed block is a zstd frame with its content checksum): the candidate without which the data
// shards, rebuilt from the rest, give back the block that the name promises is the culprit.  k + 1 small decodes and as
// many hashes of one block on the host, for a stripe the scrub has already found inconsistent.
int locate_by_content(gbm_manager *mg, const Hash &h, const Gathered &g)
{
	const int n = mg->n, k = mg->k;
	const size_t S = g.meta.shard_len, L = g.meta.orig_len;
	if (L > (size_t)k * S)
		return -1;
	auto names_the_block = [&](const std::vector<const uint8_t *> &data_shard) {
		std::vector<uint8_t> stored(L);
		for (int j = 0; j < k; ++j) {
			const size_t lo = (size_t)j * S;
			if (lo < L)
				std::memcpy(stored.data() + lo, data_shard[j], std::min(S, L - lo));
		}
		if (g.meta.compressed) {  // DataBlock::verify of a Compressed block: the frame decodes cleanly (block.rs:78-83)
			std::vector<uint8_t> plain;
			return zstd().decode(stored.data(), stored.size(), kMaxDecompressed, plain);
		}
		uint8_t sum[32];
		blake2sum(stored.data(), stored.size(), sum);
		return std::memcmp(sum, h.data(), 32) == 0;
	};
	std::vector<const uint8_t *> data_shard(k);
	for (int j = 0; j < k; ++j)
		data_shard[j] = g.shard[j].data();
	if (names_the_block(data_shard))
		return k;  // the data is sound: the parity shard is the odd one
	int culprit = -1;
	for (int c = 0; c < k; ++c) {  // data shard c erased and rebuilt from the others
		std::vector<const uint8_t *> sp(n, nullptr);
		std::vector<uint8_t *> op(n, nullptr);
		for (int j = 0; j < n; ++j)
			if (j != c)
				sp[j] = g.shard[j].data();
		Bytes rebuilt = mg->bufs->get(S);
		op[c] = rebuilt.mut();
		if (gec_reconstruct_batch(mg->bg_codec(), 1, sp.data(), op.data(), S, 1) != GEC_OK)
			return -1;
		data_shard[c] = rebuilt.data();
		const bool sound = names_the_block(data_shard);
		data_shard[c] = g.shard[c].data();
		if (sound) {
			if (culprit >= 0)
				return -1;  // ambiguous
			culprit = c;
		}
	}
	return culprit;
}

int locate_bad_shard(gbm_manager *mg, const Hash &h, const Gathered &g)
{
	const int n = mg->n, k = mg->k;
	if (mg->m < 2)
		return locate_by_content(mg, h, g);
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

}  // namespace

// shards are accepted on their headers; their checksums come back from the same device trip that checks the stripe
// against the code (every byte crosses the link once)
ScrubBatch gbmimpl::read_scrub_batch(gbm_manager *mg, std::vector<Hash> hashes)
{
	ScrubBatch bt;
	bt.batch = std::move(hashes);
	try {
		bt.rc = gather_many(mg, bt.batch, nullptr, mg->n, bt.g, /*verify=*/false, nullptr, /*migrate=*/true);
		if (bt.rc)
			bt.err = last_error();  // thread-local: carried to the caller's thread
	} catch (const std::exception &e) {
		bt.rc = GBM_E_IO;
		bt.err = e.what();
	}
	return bt;
}

// The batch on the device: ONE gec_verify_hash_batch trip per shard length, then the bookkeeping of what it found.
// st: [0] blocks scrubbed, [1] corruptions detected, [2] device verify calls, [3] shards located and set aside;
// after_trip(t) is called behind every device trip with the time it took (the tranquilizer's place).
int gbmimpl::verify_scrub_batch(gbm_manager *mg, ScrubBatch &cur, uint64_t st[4], Trace &tr,
				const std::function<void(std::chrono::nanoseconds)> &after_trip)
{
	const size_t nb = cur.batch.size();
	std::vector<Hash> &batch = cur.batch;
	std::vector<Gathered> &g = cur.g;
	std::map<size_t, std::vector<size_t>> by_len;
	// a needed block that is not fully readable is queued for resync.  It counts as a corruption only when something that
	// was read did not match (a checksum, the code): the reference's scrub counts Error::CorruptData and nothing else
	// (repair.rs:455-458) -- a shard that is simply not there (a node that is down, a put that is still catching up) is
	// the resync's business, not a corruption.
	auto unreadable = [&](size_t b, bool corrupt) {
		if (mg->get_rc(batch[b]).is_nonzero()) {
			if (corrupt)
				++st[1];
			mg->put_to_resync(batch[b], 0);
		}
	};
	for (size_t b = 0; b < nb; ++b) {
		++st[0];
		if (g[b].count == mg->n)
			by_len[g[b].meta.shard_len].push_back(b);
		else
			unreadable(b, g[b].corrupt_seen);
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
		if (after_trip)
			after_trip(std::chrono::steady_clock::now() - t_dev);
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
					if (!confirmed_corrupt(mg, gb.shard[j].data(), kv.first, gb.sum[j].data(), "gec_verify_hash_batch"))
						return fail(GBM_E_EC, "the scrub trip's shard checksums are not what the host computes: nothing was set aside");
					mg->metrics[2]++;
					if (gb.node[j] >= 0)
						mg->nodes[gb.node[j]]->mark_corrupted(batch[ids[i]], j);
					mg->put_to_resync(batch[ids[i]], 0);
					sum_bad = true;
				}
			}
			if (sum_bad) {
				unreadable(ids[i], true);
				continue;
			}
			if (ok[i])
				continue;
			++st[1];
			mg->metrics[2]++;
			const Gathered &gb = g[ids[i]];
			const int bad = locate_bad_shard(mg, batch[ids[i]], gb);
			if (bad >= 0 && gb.node[bad] >= 0) {
				mg->nodes[gb.node[bad]]->mark_corrupted(batch[ids[i]], bad);
				++st[3];
			}
			mg->put_to_resync(batch[ids[i]], 0);
		}
	}
	return GBM_OK;
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
		int grc = gather_many(mg, hs, nullptr, mg->n, g, true, nullptr, /*migrate=*/true);
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
	// phase 1: every hash of the refcount table; phase 2: every hash a node stores ("blocks we are storing but don't
	// actually need"), one first-level directory at a time -- the walk never holds more than a 256th of the store
	size_t total = 0;
	try {
		std::set<Hash> known;
		for (auto &st : mg->rc) {
			std::lock_guard<std::mutex> g(st.mu);
			for (auto &kv : st.map)
				known.insert(kv.first);
		}
		for (const Hash &h : known)
			mg->put_to_resync(h, 0);
		total = known.size();
		// Phase 2 hands "blocks we are storing but don't actually need" to a resync that DELETES what nothing references
		// (RcEntry::Absent is deletable, rc.rs:222-228).  The reference's refcount table is durable; this mirror's lives in
		// memory, so right after a restart it is empty and every stored block would look unneeded.  An empty table beside a
		// store that is not empty is that situation, not a cluster full of garbage: refused.
		if (known.empty()) {
			std::set<Hash> any;
			for (int p = 0; p < 256 && any.empty(); ++p)
				for (auto &nd : mg->nodes)
					if (!nd->down.load()) {
						nd->list_prefix(p, any);
						if (!any.empty())
							break;
					}
			bool mine = false;
			for (const Hash &h : any)
				mine = mine || mg->owns(h);
			if (mine)
				return fail(GBM_E_INVALID_ARG, "the refcount table is empty while blocks are stored: count the references again "
								   "(gbm_block_incref) before a repair -- resync deletes what nothing references");
		}
		for (int p = 0; p < 256; ++p) {
			std::vector<std::set<Hash>> per(mg->nodes.size());
			mg->pool->parallel_for(mg->nodes.size(), [&](size_t i) {
				if (!mg->nodes[i]->down.load())
					mg->nodes[i]->list_prefix(p, per[i]);
			});
			std::set<Hash> stored;
			for (auto &st : per)
				for (const Hash &h : st)
					if (mg->owns(h) && !known.count(h))
						stored.insert(h);
			for (const Hash &h : stored)
				mg->put_to_resync(h, 0);
			total += stored.size();
		}
	} catch (const std::exception &e) {
		return fail(GBM_E_IO, std::string("repair_all: ") + e.what());
	}
	if (queued)
		*queued = total;
	return GBM_OK;
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
		auto read_at = [&](size_t b0) {
			const size_t nb = std::min(batch_blocks, hs.size() - b0);
			return read_scrub_batch(mg, std::vector<Hash>(hs.begin() + b0, hs.begin() + b0 + nb));
		};
		// Tranquilizer::tranquilize (tranquilizer.rs:38-69): after a device trip that took t, sleep tranquility * t
		const std::function<void(std::chrono::nanoseconds)> tranquilize = [&](std::chrono::nanoseconds spent) {
			if (const uint32_t tranq = mg->scrub_tranquility.load()) {
				std::this_thread::sleep_for(spent * tranq);
				mg->tranquilized_ms += (uint64_t)std::chrono::duration_cast<std::chrono::milliseconds>(spent * tranq).count();
			}
		};
		std::future<ScrubBatch> next;
		if (!hs.empty())
			next = std::async(std::launch::async, read_at, (size_t)0);
		for (size_t b0 = 0; b0 < hs.size(); b0 += batch_blocks) {
			ScrubBatch cur = next.get();
			tr.lap("wait for the batch's shards");
			if (b0 + batch_blocks < hs.size())
				next = std::async(std::launch::async, read_at, b0 + batch_blocks);
			int rc = cur.rc ? fail(cur.rc, cur.err) : verify_scrub_batch(mg, cur, st, tr, tranquilize);
			if (rc) {
				if (next.valid())
					next.wait();
				return rc;
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
