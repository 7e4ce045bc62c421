// bm_rw.cpp -- the request path: rpc_put_block(s) (encode + checksums in ONE device trip, per-node fan-out, write
// quorum) and rpc_get_block(s) (gather k shards, ONE decode + verify trip, assembly), plus their C entry points.
#include "bm_internal.hpp"

namespace gbmimpl {

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
		if (mg->test_bad_put_sums.load() > 0 && mg->test_bad_put_sums.fetch_sub(1) > 0)
			gsums[(gn - 1) * (size_t)n * 32 + 5] ^= 0x40;  // (test hook: the device got shard 0 of the trip's last block wrong)
		// The spot check (gbm_set_put_spot_check): one shard of one block of this trip -- data or parity -- is hashed again on
		// the host before anything is sent; a trip whose checksums the host cannot reproduce stores nothing.  ("every trip"
		// looks at shard 0 of every block as well: the setting of the tests and of the paranoid.)
		if (const uint32_t every = mg->put_spot_every.load()) {
			const uint64_t trip = mg->put_trips.fetch_add(1);
			if (trip % every == 0) {
				auto matches = [&](size_t i, size_t j) {
					const uint8_t *p = j < (size_t)k ? prep[ids[i]].block.data() + j * S : prep[ids[i]].parity.data() + (j - (size_t)k) * S;
					uint8_t host_sum[32];
					shardsum_v(mg->sumver, p, S, host_sum);
					return std::memcmp(host_sum, gsums.data() + (i * (size_t)n + j) * 32, 32) == 0;
				};
				const uint64_t draw = (trip / every) * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull;
				mg->bmx.put_spot_checks++;
				bool ok = matches((size_t)(draw >> 33) % gn, (size_t)(draw >> 11) % (size_t)n);
				for (size_t i = 0; every == 1 && ok && i < gn; ++i)
					ok = matches(i, 0);
				if (!ok) {
					mg->bmx.put_spot_check_failures++;
					if (rcs)
						std::fill(rcs, rcs + nb, GBM_E_EC);
					return fail(GBM_E_EC, "a put trip's shard checksums are not what the host computes: nothing was sent to any node");
				}
			}
		}
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
		 std::vector<uint8_t> *changed, const FanoutGate *gate, std::vector<uint8_t> *have_sum, const std::function<void(size_t)> *overlap_block,
		 const EarlyHashFn *hash_early)
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
	// want_block_sums == 2 asks the device trip for the end-to-end hash of the blocks it REBUILDS.  When the gather came back
	// healthy -- or with few enough degraded blocks that the pool hashes them at once (cpu_block_hash_max, the same bound
	// get_blocks_once applies to a whole request) -- no block needs a hash from the trip: the batch takes the host check
	// (fused with the copy-out for a big batch), blocks that miss a data shard go to the device for the decode alone, and
	// the caller hashes what was rebuilt afterwards (its have_sum[b] stays 0).  A fully healthy default-mode get is then the
	// same work as GBM_VERIFY_OFF.
	if (want_block_sums == 2 && mg->sumver == 3) {
		size_t need = 0;
		for (size_t b = 0; b < nb; ++b) {
			if (!g[b].have_meta || g[b].count < k)
				continue;
			for (int j = 0; j < k; ++j)
				if (g[b].shard[j].empty()) {
					++need;
					break;
				}
		}
		if (need <= mg->cpu_block_hash_max.load())
			want_block_sums = 0;
	}
	std::exception_ptr helper_err;  // what the overlapped work threw: carried to this thread (on the helper it would be std::terminate)
	std::thread helper;
	struct Joiner {
		std::thread &t;
		~Joiner()
		{
			if (t.joinable())
				t.join();
		}
	} joiner{helper};
	// header version 3, a batch of some size, no block checksums wanted from a device trip: the early assembly rides in the
	// shard-check tasks below (one pool task per block: check its k shards, then copy them out while they are in that core's cache)
	const bool fused_host = overlap_block && mg->sumver == 3 && want_block_sums == 0 && nb >= 16;
	// ... and with the hash of EVERY block wanted (GBM_VERIFY_ALWAYS, the default over header version 3) a big batch is shared: the
	// device trip is ~11 ms of BLAKE2b chain per MiB of block however few blocks it carries, and the pool idles beside it -- so the
	// pool takes as many healthy blocks as it checks, assembles and hashes in that time (`shared` below)
	const bool shared_ok = overlap_block && hash_early && mg->sumver == 3 && want_block_sums == 1 && nb >= 64;
	if (overlap && !fused_host && !shared_ok)
		helper = std::thread([&overlap, &helper_err] {
			name_thread("gbm-get-helper");
			try {
				overlap();
			} catch (...) {
				helper_err = std::current_exception();
			}
		});
	auto join_helper = [&] {
		helper.join();
		if (helper_err)
			std::rethrow_exception(helper_err);
	};
	std::vector<uint8_t> todo(nb, 1);
	for (int round = 0; round <= n; ++round) {
		if (round == 1 && helper.joinable())
			join_helper();
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
			int rc = GEC_OK;
			bool shared_now = false;             // this group went through the shared form below
			std::vector<uint8_t> host_sum_ok;    // ... and bsums[32*i] is meaningful for these
			// Shard-header version 3 (MLH64): a host core checks a shard at memory speed, so the shards of a read are verified
			// HERE, on the pool -- a healthy get does not cross the link at all -- and only blocks that miss a data shard go to
			// the device, for the decode alone (gec_reconstruct_batch).  (With block checksums wanted from the trip the one-trip
			// form below still serves: the shards have to be on the device for those anyway.)
			const bool host_check = mg->sumver == 3 && !trip_sums;
			if (host_check) {
				struct Item {
					uint32_t i, j;
				};
				std::vector<Item> items;
				items.reserve(ids.size() * (size_t)k);
				for (size_t i = 0; i < ids.size(); ++i) {
					int seen = 0;
					for (int j = 0; j < n && seen < k; ++j)
						if (sp[i * n + j]) {
							items.push_back(Item{(uint32_t)i, (uint32_t)j});
							++seen;
						}
				}
				// The decode does not wait for the verdicts: it reads the same (immutable) shard buffers the pool is checking, so it
				// runs BESIDE the check -- on the device, from a thread of its own -- and what it rebuilt from a shard that then
				// fails its checksum is simply not used (the block goes round again below).  Exactly the first k present shards
				// of a block are handed over: the decode reads what is being verified.
				std::vector<const uint8_t *> dsp;
				std::vector<uint8_t *> dop;
				size_t ndec = 0;
				for (size_t i = 0; i < ids.size() && nrebuild; ++i) {
					bool wants = false;
					for (int j = 0; j < k; ++j)
						wants = wants || op[i * n + j];
					if (!wants)
						continue;
					int seen = 0;
					for (int j = 0; j < n; ++j) {
						const bool use = sp[i * n + j] && seen < k;
						seen += use ? 1 : 0;
						dsp.push_back(use ? sp[i * n + j] : nullptr);
						dop.push_back(op[i * n + j]);
					}
					++ndec;
				}
				int drc = GEC_OK;
				std::string derr;
				std::thread decoder;
				if (ndec)
					decoder = std::thread([&] {
						// (a thread of its own inside a C entry point: nothing may leave it -- an exception here would be std::terminate)
						try {
							name_thread("gbm-get-decode");
							DeviceTurn turn(gate);
							drc = gec_reconstruct_batch(mg->codec, ndec, dsp.data(), dop.data(), S, /*data_only=*/1);
							if (drc)
								derr = gec_last_error();  // (thread-local over there: carried to the caller's thread)
						} catch (const std::exception &e) {
							drc = GEC_E_NOMEM;
							try {
								derr = e.what();
							} catch (...) {
							}
						} catch (...) {
							drc = GEC_E_NOMEM;
						}
					});
				struct JoinDecoder {
					std::thread &t;
					~JoinDecoder()
					{
						if (t.joinable())
							t.join();
					}
				} join_decoder{decoder};
				const size_t per = 4;  // shards per pool task
				auto check = [&](size_t t) {
					for (size_t q = t * per; q < std::min(items.size(), (t + 1) * per); ++q)
						mlh::shardsum3(sp[items[q].i * (size_t)n + items[q].j], S, ssums.data() + (items[q].i * (size_t)n + items[q].j) * 32);
				};
				const size_t ntask = (items.size() + per - 1) / per;
				if (fused_host && round == 0) {
					// one task per block: its shards are consecutive in `items` (k of them, the first k present)
					mg->pool->parallel_for(ids.size(), [&](size_t i) {
						int seen = 0;
						for (int j = 0; j < n && seen < k; ++j)
							if (sp[i * n + j]) {
								mlh::shardsum3(sp[i * n + j], S, ssums.data() + (i * (size_t)n + j) * 32);
								++seen;
							}
						(*overlap_block)(ids[i]);
					});
				} else if (ntask <= 2) {
					for (size_t t = 0; t < ntask; ++t)
						check(t);
				} else {
					mg->pool->parallel_for(ntask, check);
				}
				tr.lap("shard checksums on the host");
				if (decoder.joinable())
					decoder.join();
				if (ndec)
					tr.lap("decode (beside the checks)");
				if (drc)
					return fail(GBM_E_EC, std::string("gec_reconstruct_batch: ") + gec_strerror(drc) + " (" + derr + ")");
			} else if (shared_ok && round == 0 && trip_sums) {
				// -- the shared form.  How many blocks the pool takes: what it gets through (check 1.0 + copy 1.0 + hash 1.0 bytes per
				// byte: ~1 GB/s per thread with eight chains per core) while the device needs max(its share over the link at ~50 GB/s,
				// one block's chain: ~11 ms per MiB) + ~2 ms for the rest.  Blocks that miss a data shard stay with the device.
				// (the pool's rate is MEASURED: every shared call leaves what its share achieved per thread -- 2.5 GB/s on a Zen 5
				// core with eight chains at a time, a third of that on a host without AVX-512 -- and the next call splits by it)
				const double Lb = (double)k * (double)S, nthr = (double)mg->pool->workers() + 1.0;
				const double host_rate = std::min(8e9, std::max(0.2e9, (double)mg->shared_host_rate.load(std::memory_order_relaxed)));
				const double t_host = Lb / host_rate / nthr, t_link = Lb / 50.0e9, t_chain = 11e-3 * Lb / 1048576.0, t_fixed = 2e-3;
				std::vector<uint8_t> healthy(ids.size(), 0);
				size_t nhealthy = 0;
				for (size_t i = 0; i < ids.size(); ++i) {
					bool whole = true;
					for (int j = 0; j < k; ++j)
						whole = whole && sp[i * n + j];
					healthy[i] = whole;
					nhealthy += whole;
				}
				double want_host = (t_chain + t_fixed) / t_host;                               // chain-bound device share
				if (((double)ids.size() - want_host) * t_link > t_chain)                        // link-bound device share
					want_host = (t_fixed + (double)ids.size() * t_link) / (t_host + t_link);
				const size_t nhost = std::min(nhealthy, (size_t)std::max(0.0, want_host));
				std::vector<size_t> host_i, dev_i;
				for (size_t i = 0; i < ids.size(); ++i)
					(healthy[i] && host_i.size() < nhost ? host_i : dev_i).push_back(i);
				// the device's share, on a thread of its own (contiguous argument arrays for the sub-batch, results scattered back)
				const size_t nd = dev_i.size();
				std::vector<const uint8_t *> dsp(nd * n);
				std::vector<uint8_t *> dop(nd * n);
				std::vector<size_t> dlens(nd);
				std::vector<uint8_t> dss(nd * (size_t)n * 32), dbs(nd * 32);
				for (size_t q = 0; q < nd; ++q) {
					std::copy(sp.begin() + dev_i[q] * n, sp.begin() + (dev_i[q] + 1) * n, dsp.begin() + q * n);
					std::copy(op.begin() + dev_i[q] * n, op.begin() + (dev_i[q] + 1) * n, dop.begin() + q * n);
					dlens[q] = lens[dev_i[q]];
				}
				int drc = GEC_OK;
				std::string derr;
				std::thread trip;
				struct JoinTrip {
					std::thread &t;
					~JoinTrip()
					{
						if (t.joinable())
							t.join();
					}
				} join_trip{trip};
				if (nd)
					trip = std::thread([&] {
						try {  // (a thread of its own inside a C entry point: nothing may leave it)
							name_thread("gbm-get-trip");
							DeviceTurn turn(gate);
							drc = gec_decode_verify_batch(mg->codec, nd, dsp.data(), S, dlens.data(), dop.data(), dss.data(), dbs.data());
							if (drc)
								derr = gec_last_error();
						} catch (...) {
							drc = GEC_E_NOMEM;
						}
					});
				// the pool's share: eight blocks per task (the eight chains of one core), check -> assemble -> hash while the block is
				// in that core's cache; then the early assembly of the device's blocks (what the helper thread does otherwise)
				host_sum_ok.assign(ids.size(), 0);
				const size_t ngrp = (host_i.size() + 7) / 8;
				const auto pool_t0 = std::chrono::steady_clock::now();
				mg->pool->parallel_for(ngrp + nd, [&](size_t t) {
					if (t >= ngrp) {
						(*overlap_block)(ids[dev_i[t - ngrp]]);
						return;
					}
					size_t bs[8];
					uint8_t ok8[8] = {0}, sums8[8 * 32];
					const size_t i0 = t * 8, cnt = std::min<size_t>(8, host_i.size() - i0);
					for (size_t q = 0; q < cnt; ++q) {
						const size_t i = host_i[i0 + q];
						for (int j = 0; j < k; ++j)  // (healthy: the first k present shards are the data shards)
							mlh::shardsum3(sp[i * n + j], S, ssums.data() + (i * (size_t)n + j) * 32);
						bs[q] = ids[i];
						(*overlap_block)(bs[q]);
					}
					(*hash_early)(bs, cnt, sums8, ok8);
					for (size_t q = 0; q < cnt; ++q)
						if (ok8[q]) {
							std::memcpy(bsums.data() + 32 * host_i[i0 + q], sums8 + 32 * q, 32);
							host_sum_ok[host_i[i0 + q]] = 1;
						}
				});
				if (host_i.size() >= 32) {  // (enough blocks for the figure to mean something: half old, half new)
					const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - pool_t0).count();
					if (dt > 0) {
						const double seen = (double)host_i.size() * Lb / dt / nthr;
						mg->shared_host_rate = (uint64_t)(0.5 * host_rate + 0.5 * std::min(8e9, std::max(0.2e9, seen)));
					}
				}
				tr.lap("the pool's share: check + assemble + hash");
				if (trip.joinable())
					trip.join();
				tr.lap("the device's share: decode+verify");
				if (drc)
					return fail(GBM_E_EC, std::string("gec_decode_verify_batch: ") + gec_strerror(drc) + " (" + derr + ")");
				for (size_t q = 0; q < nd; ++q) {
					std::memcpy(ssums.data() + dev_i[q] * (size_t)n * 32, dss.data() + q * (size_t)n * 32, (size_t)n * 32);
					std::memcpy(bsums.data() + 32 * dev_i[q], dbs.data() + 32 * q, 32);
					host_sum_ok[dev_i[q]] = 1;
				}
				shared_now = true;
				mg->gpu_hashed += nd * (size_t)k + nd;
			} else {
				DeviceTurn turn(gate);
				rc = gec_decode_verify_batch(mg->codec, ids.size(), sp.data(), S, lens.data(), op.data(), ssums.data(),
							     trip_sums ? bsums.data() : nullptr);
				tr.lap("decode+verify");
			}
			if (helper.joinable())
				join_helper();  // the overlapped host work reads g: it must be done before the results below change it
			tr.lap("join overlapped assembly");
			if (rc)
				return ec_fail(rc, host_check ? "gec_reconstruct_batch" : "gec_decode_verify_batch");
			if (!host_check && !shared_now)
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
				if (trip_sums && (!shared_now || host_sum_ok[i])) {  // (shared form: a block the pool could not hash early is hashed by the caller)
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
			join_helper();
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
			   const size_t *cap, size_t *len_out, int *rcs, bool raw, gbm_data_block_header *headers, const FanoutGate *gate,
			   std::vector<uint8_t> &worth_retry);
static int get_blocks_again(gbm_manager *mg, size_t nb, const uint8_t *hashes, const gbm_order_tag *tags, uint8_t *const *out,
			    const size_t *cap, size_t *len_out, int *rcs, bool raw, gbm_data_block_header *headers, const FanoutGate *gate,
			    std::vector<uint8_t> &worth_retry);

// raw == true: rpc_get_raw_block (stored bytes + header); false: rpc_get_block (plain bytes).
// While a layout change is being followed (more than one version is active) resync MOVES shards: PutShard to the new owner,
// then DeleteShard at the old one.  The gather asks a shard's holders oldest version first (bm_gather.cpp, block_read_nodes_of's
// order, rpc_helper.rs:559-603), so a single move cannot hide a shard from it.  What is left for the retry below: a shard whose
// old holder was DOWN when it was asked and whose move completed in between, and a walk that met a corrupt copy at one holder
// while the other one was still on its way -- a block that comes back Missing (or Corrupt with too few shards) during a
// transition is asked for again, twice at most: moves only go forward, a later walk meets the shards where an earlier one's
// went to.  (The reference leaves the same cases to the client's retry.)
// What get_blocks_once says about the blocks it could not return (`worth_retry`, one flag per block of the call): 1 = its walk
// found SOME of the block (a shard, or a corrupt copy) but too little, or every holder it asked was DOWN while shards are being
// moved -- what a move in progress looks like, worth another walk; 0 = every holder of every version answered and none had
// anything of it (the block simply is not there), or its bytes were read and failed the end-to-end check: a retry cannot change
// either.

int get_blocks_impl(gbm_manager *mg, size_t nb, const uint8_t *hashes, const gbm_order_tag *tags, uint8_t *const *out,
		    const size_t *cap, size_t *len_out, int *rcs, bool raw, gbm_data_block_header *headers, const FanoutGate *gate)
{
	if (!mg || (nb && (!hashes || !out || !cap || !len_out || !rcs)))
		return fail(GBM_E_INVALID_ARG, "NULL argument");
	for (size_t b = 0; b < nb; ++b)
		if (!out[b] && cap[b])
			return fail(GBM_E_INVALID_ARG, "NULL output buffer with a capacity");
	DurationScope read_time(mg->bmx.read_duration);  // block.read_duration (metrics.rs:117-121): one observation per call
	std::vector<uint8_t> worth_retry;
	int rc = get_blocks_once(mg, nb, hashes, tags, out, cap, len_out, rcs, raw, headers, gate, worth_retry);
	// (twice more at most, a millisecond and five apart: a slow reader beside a fast mover can lose several shards of one block to
	// the window in one walk; the mover is done with a block in well under that)
	for (int attempt = 1; attempt <= 2 && rc == GBM_OK && mg->layout_cur.load() != mg->layout_oldest.load(); ++attempt) {
		bool any = false;
		for (size_t b = 0; b < nb && !any; ++b)
			any = (rcs[b] == GBM_E_MISSING_BLOCK || rcs[b] == GBM_E_CORRUPT_DATA) && b < worth_retry.size() && worth_retry[b];
		if (!any)  // (a block nobody holds and a block whose content does not match its name are final: no sleep, no second walk)
			break;
		if (attempt == 2)
			std::this_thread::sleep_for(std::chrono::milliseconds(5));
		else
			std::this_thread::sleep_for(std::chrono::milliseconds(1));
		rc = get_blocks_again(mg, nb, hashes, tags, out, cap, len_out, rcs, raw, headers, gate, worth_retry);
	}
	return rc;
}

// the blocks of a call that came back Missing (or Corrupt with too few shards found), asked for once more
static int get_blocks_again(gbm_manager *mg, size_t nb, const uint8_t *hashes, const gbm_order_tag *tags, uint8_t *const *out,
			    const size_t *cap, size_t *len_out, int *rcs, bool raw, gbm_data_block_header *headers, const FanoutGate *gate,
			    std::vector<uint8_t> &worth)
{
	int rc = GBM_OK;
	std::vector<size_t> again;
	for (size_t b = 0; b < nb; ++b)
		if ((rcs[b] == GBM_E_MISSING_BLOCK || rcs[b] == GBM_E_CORRUPT_DATA) && b < worth.size() && worth[b])  // (Corrupt: a bad shard was met AND too few others were found)
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
	std::vector<uint8_t> worth_sub, defer_sub(na, 0);
	FanoutGate sub_gate;
	if (gate) {
		sub_gate = *gate;
		if (gate->defer_block_hash)
			sub_gate.defer_block_hash = &defer_sub;  // (indexed by the sub-call's blocks: mapped back below)
	}
	rc = get_blocks_once(mg, na, hh.data(), tags ? tt.data() : nullptr, oo.data(), cc.data(), ll.data(), rr.data(), raw,
			     headers ? hd.data() : nullptr, gate ? &sub_gate : nullptr, worth_sub);
	if (rc != GBM_OK)
		return rc;
	if (gate && gate->defer_block_hash)
		for (size_t i = 0; i < na; ++i)
			if (again[i] < gate->defer_block_hash->size())
				(*gate->defer_block_hash)[again[i]] = defer_sub[i];
	std::vector<uint8_t> worth2(nb, 0);
	for (size_t i = 0; i < na; ++i) {
		rcs[again[i]] = rr[i];
		len_out[again[i]] = ll[i];
		if (headers)
			headers[again[i]] = hd[i];
		worth2[again[i]] = i < worth_sub.size() ? worth_sub[i] : 0;
	}
	worth = worth2;
	return GBM_OK;
}

static int get_blocks_once(gbm_manager *mg, size_t nb, const uint8_t *hashes, const gbm_order_tag *tags, uint8_t *const *out,
			   const size_t *cap, size_t *len_out, int *rcs, bool raw, gbm_data_block_header *headers, const FanoutGate *gate,
			   std::vector<uint8_t> &worth_retry)
{
	worth_retry.assign(nb, 0);
	const int k = mg->k;
	std::vector<Hash> hs(nb);
	for (size_t b = 0; b < nb; ++b)
		hs[b].assign((const char *)hashes + 32 * b, 32);
	std::vector<Gathered> g;
	std::vector<uint8_t> block_sums, changed, have_sum, early(nb, 0);
	std::vector<uint8_t> final_verdict(nb, 0);  // the block's bytes were read and failed a content check: no walk can change that
	// The requester's end-to-end check (gbm_set_verify_block_hash): every Plain block, only the blocks that went through a
	// decode, or -- the default, the reference's read path -- none: the shard checksums of the same trip are the serving
	// node's verify (read_block_from, manager.rs:577-609), and they are always checked.
	const int asked = mg->verify_mode.load();
	const bool defer = gate && gate->defer_block_hash && asked == GBM_VERIFY_ALWAYS && gate->defer_block_hash->size() >= nb;
	const int mode = defer ? GBM_VERIFY_REBUILT : asked;  // (the callers check the healthy blocks themselves: FanoutGate)
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
	const std::function<void(size_t)> assemble_one = [&](size_t b) {
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
		};
	auto assemble_early = [&] { mg->pool->parallel_for(nb, assemble_one); };
	// the blake2sum of blocks assemble_one has just laid down in the caller's buffers (the shared form of fetch_blocks): eight at a time
	const EarlyHashFn hash_early = [&](const size_t *bs, size_t n8, uint8_t *sums, uint8_t *ok) {
		const uint8_t *ptr[8] = {};
		size_t len[8] = {}, at[8] = {}, cnt = 0;
		for (size_t i = 0; i < n8 && i < 8; ++i) {
			ok[i] = 0;
			if (!early[bs[i]] || !missing_early[bs[i]].empty())
				continue;
			ptr[cnt] = out[bs[i]];
			len[cnt] = g[bs[i]].meta.orig_len;
			at[cnt++] = i;
		}
		if (!cnt)
			return;
		uint8_t tmp[8 * 32];
		b2host::blake2sum_many(ptr, len, cnt, tmp);
		for (size_t c = 0; c < cnt; ++c) {
			std::memcpy(sums + 32 * at[c], tmp + 32 * c, 32);
			ok[at[c]] = 1;
		}
	};
	Trace tr("get (whole call)");
	int frc = fetch_blocks(mg, hs, tags, g, rcs, verify && !cpu_hash ? (only_rebuilt ? 2 : 1) : 0, block_sums, assemble_early, &changed,
			       gate, &have_sum, &assemble_one, &hash_early);
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
		if (defer && !z && !check && !raw)
			(*gate->defer_block_hash)[b] = 1;  // (cleared below if the block turns out not to be deliverable)
		if (check && !cpu_hash && have_sum[b] && std::memcmp(block_sums.data() + 32 * b, hashes + 32 * b, 32) != 0) {
			rcs[b] = GBM_E_CORRUPT_DATA;
			final_verdict[b] = 1;
			return;
		}
		if (z && !raw) {
			std::vector<uint8_t> frame(L), plain;
			assemble(g[b], k, frame.data());
			if (!zstd().decode(frame.data(), L, kMaxDecompressed, plain)) {
				rcs[b] = GBM_E_CORRUPT_DATA;
				final_verdict[b] = 1;
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
		// ... but only when there are more blocks than threads: eight chains in lockstep take 2.3 times as long as one chain
		// alone (2.3 against 1.0 ms per MiB on a Zen 5 core), so a GetObject's few blocks -- the default mode over checksum v3
		// hashes every one -- are back sooner one per thread (3 blocks: 1.2 ms instead of 1.7; 8: 1.2 - 2.0 instead of 2.3).
		// Beyond one round the eight-lane form's 3.5x lower CPU cost per block wins:
		// the coalescing queue's batches of ~30 under 48 readers lost a fifth of their rate with one block per task.
		const size_t nthr = (size_t)mg->pool->workers() + 1;
		const size_t per = b2host::mb_available() && idx.size() > nthr ? 8 : 1;
		mg->pool->parallel_for((idx.size() + per - 1) / per, [&](size_t grp) {
			const size_t i0 = grp * per, cnt = std::min<size_t>(per, idx.size() - i0);
			const uint8_t *ptr[8] = {};
			size_t len[8] = {};
			uint8_t sums[8 * 32];
			for (size_t i = 0; i < cnt; ++i) {
				ptr[i] = out[idx[i0 + i]];
				len[i] = len_out[idx[i0 + i]];
			}
			b2host::blake2sum_many(ptr, len, cnt, sums);
			for (size_t i = 0; i < cnt; ++i) {
				const size_t b = idx[i0 + i];
				if (std::memcmp(sums + 32 * i, hashes + 32 * b, 32) != 0) {
					rcs[b] = GBM_E_CORRUPT_DATA;
					final_verdict[b] = 1;
				} else {
					mg->metrics[5]++;
				}
			}
		});
	}
	tr.lap("finish");
	if (defer)
		for (size_t b = 0; b < nb; ++b)
			if (rcs[b] != GBM_OK)
				(*gate->defer_block_hash)[b] = 0;
	for (size_t b = 0; b < nb; ++b)
		if ((rcs[b] == GBM_E_MISSING_BLOCK || rcs[b] == GBM_E_CORRUPT_DATA) && !final_verdict[b] && b < g.size())
			worth_retry[b] = g[b].count > 0 || g[b].corrupt_seen || g[b].have_meta || g[b].down_seen;
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
		try {
			for (int t = 1; t < kThreads; ++t)
				others.emplace_back([&run, mg] {
					lane_thread("gbm-put-slice", mg->codec);
					run();
				});
		} catch (...) {
			// no further thread to be had: the ones that started and this one share the slices (unwinding past joinable
			// threads would be std::terminate)
		}
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

}  // extern "C"
