// qos_bench -- foreground puts against background scrub on one device (the reference keeps repair off
// the request path with a bounded worker pool and a Tranquilizer, src/block/resync.rs:43-46,513-599,
// src/util/tranquilizer.rs:38-69).
//
// Two managers over ONE foreground codec: manager A takes the puts (C closed-loop callers through the coalescing
// batcher, 1 MiB blocks, RS(10,4), 16 in-memory nodes), manager B holds N blocks that gbm_scrub_all verifies over and
// over.  Four phases of `seconds` each: puts alone; scrub alone; both with maintenance on the BACKGROUND-class codec
// (gec_codec_background: low-priority CU-masked streams, small chunks that yield to foreground calls); both with
// maintenance on the request path's own codec (gbm_set_maintenance_class(m, 0): round 2's behaviour).
// Reports put p50 / p99 / rate and the scrub rate of each phase.
// usage: qos_bench [callers=3] [seconds=3] [scrub_blocks=512] [tranquility=0] [get_blocks=0] [gets_via_batcher=0] [nodes_down=0] [maintenance=scrub|resync]
// nodes_down = D > 0 (with get_blocks): D of manager A's 16 nodes are down -- the gets are DEGRADED reads, every one of them
// goes through a decode whose rebuilt shards travel home over the link;
// maintenance = resync: instead of a scrub, manager B keeps losing one shard of every block and gbm_resync_run rebuilds
// them (gather k, ONE reconstruct trip, the rebuilt shards written into host memory: the background class's WRITES).
// get_blocks = G > 0: the callers are readers instead -- each keeps fetching G of its blocks with one gbm_rpc_get_blocks (a
// GetObject with its prefetch), block-hash check on -- and the latencies reported are those of the gets;
// gets_via_batcher = 1 (with get_blocks = 1): every reader fetches one block at a time through gbm_batcher_get_block, the
// read side of the coalescing queue (48 callers = 16 GetObjects with three blocks in flight each).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "garage_block.h"

static const size_t L = 1u << 20;
using Clock = std::chrono::steady_clock;

static void fill(std::vector<uint8_t> &b, uint64_t seed)
{
	uint64_t x = seed * 0x9E3779B97F4A7C15ull + 88172645463325252ull;
	for (size_t o = 0; o + 8 <= b.size(); o += 8) {
		x ^= x << 13, x ^= x >> 7, x ^= x << 17;
		memcpy(&b[o], &x, 8);
	}
}

struct PutStats {
	std::vector<double> lat_ms;
	double secs = 0;
};

int main(int argc, char **argv)
{
	const int callers = argc > 1 ? atoi(argv[1]) : 3;
	const double seconds = argc > 2 ? atof(argv[2]) : 3.0;
	const size_t nscrub = argc > 3 ? (size_t)atol(argv[3]) : 512;
	const int tranq = argc > 4 ? atoi(argv[4]) : 0;
	const int get_blocks = argc > 5 ? atoi(argv[5]) : 0;
	const bool gets_via_batcher = argc > 6 && atoi(argv[6]) != 0 && get_blocks == 1;
	const int nodes_down = argc > 7 ? atoi(argv[7]) : 0;
	const bool resync_mode = argc > 8 && std::string(argv[8]) == "resync";
	gec_codec *c = nullptr;
	gbm_manager *ma = nullptr, *mb = nullptr;
	if (gec_codec_create(10, 4, GEC_BACKEND_AUTO, 0, &c) != GEC_OK || gbm_create(c, 16, NULL, 0, &ma) != GBM_OK ||
	    gbm_create(c, 16, NULL, 0, &mb) != GBM_OK) {
		fprintf(stderr, "setup failed: %s / %s\n", gec_last_error(), gbm_last_error());
		return 2;
	}
	gbm_set_tranquility(mb, tranq, -1);
	// ---- manager B: the blocks the scrub walks
	std::vector<uint8_t> bhashes(nscrub * 32);
	{
		std::vector<std::vector<uint8_t>> blk(64, std::vector<uint8_t>(L));
		std::vector<uint8_t> hashes(64 * 32);
		for (size_t b0 = 0; b0 < nscrub; b0 += 64) {
			const size_t nb = std::min<size_t>(64, nscrub - b0);
			std::vector<const uint8_t *> d(nb);
			std::vector<size_t> l(nb, L);
			for (size_t i = 0; i < nb; ++i) {
				fill(blk[i], 1000 + b0 + i);
				gbm_blake2sum(blk[i].data(), L, &hashes[32 * i]);
				d[i] = blk[i].data();
			}
			if (gbm_rpc_put_blocks(mb, nb, hashes.data(), d.data(), l.data(), NULL, NULL) != GBM_OK) {
				fprintf(stderr, "preload failed: %s\n", gbm_last_error());
				return 2;
			}
			memcpy(&bhashes[32 * b0], hashes.data(), 32 * nb);
			for (size_t i = 0; i < nb; ++i)
				gbm_block_incref(mb, &hashes[32 * i]);  // needed blocks: resync REBUILDS what is missing
		}
	}
	// ---- manager A: every caller rewrites its own ring of 8 blocks (memory stays bounded)
	const int RING = 8;
	std::vector<std::vector<uint8_t>> data((size_t)callers * RING, std::vector<uint8_t>(L));
	std::vector<uint8_t> hashes((size_t)callers * RING * 32);
	for (size_t i = 0; i < data.size(); ++i) {
		fill(data[i], i + 1);
		gbm_blake2sum(data[i].data(), L, &hashes[32 * i]);
	}
	gbm_batcher *bt = nullptr;
	if (gbm_batcher_create(ma, 128, 300, &bt) != GBM_OK)
		return 2;
	gbm_set_verify_block_hash(ma, GBM_VERIFY_OFF);
	std::atomic<bool> stop{false};
	const char *op = get_blocks > 0 ? "get" : "put";
	if (get_blocks > 0) {  // readers: their blocks have to be there first
		if (get_blocks > RING)
			return 2;
		std::vector<const uint8_t *> d(data.size());
		std::vector<size_t> l(data.size(), L);
		for (size_t i = 0; i < data.size(); ++i)
			d[i] = data[i].data();
		if (gbm_rpc_put_blocks(ma, data.size(), hashes.data(), d.data(), l.data(), NULL, NULL) != GBM_OK) {
			fprintf(stderr, "preload failed: %s\n", gbm_last_error());
			return 2;
		}
		for (int dn = 0; dn < nodes_down; ++dn)
			gbm_node_set_down(ma, 3 + 4 * dn, 1);
	}
	for (int d = 0; d < nodes_down && get_blocks > 0; ++d)  // (after the preload below would be too late to matter: done there)
		(void)d;
	auto run_gets = [&](PutStats &ps) {
		std::vector<std::vector<double>> lat(callers);
		std::vector<std::thread> th;
		const auto t0 = Clock::now();
		for (int t = 0; t < callers; ++t)
			th.emplace_back([&, t] {
				std::vector<std::vector<uint8_t>> out(get_blocks, std::vector<uint8_t>(L));
				std::vector<uint8_t *> op_(get_blocks);
				std::vector<size_t> cap(get_blocks, L), len(get_blocks);
				std::vector<int> rcs(get_blocks);
				for (int i = 0; i < get_blocks; ++i)
					op_[i] = out[i].data();
				for (size_t j = 0; !stop.load(); ++j) {
					const size_t i0 = (size_t)t * RING + (j % (RING / get_blocks)) * get_blocks;
					const auto a = Clock::now();
					int grc;
					if (gets_via_batcher) {
						grc = gbm_batcher_get_block(bt, &hashes[32 * i0], op_[0], L, &len[0]);
						rcs[0] = grc;
					} else {
						grc = gbm_rpc_get_blocks(ma, get_blocks, &hashes[32 * i0], NULL, op_.data(), cap.data(), len.data(), rcs.data());
					}
					if (grc != GBM_OK || rcs[0] != GBM_OK || len[0] != L) {
						fprintf(stderr, "get failed: %s\n", gbm_last_error());
						exit(1);
					}
					const double ms = std::chrono::duration<double, std::milli>(Clock::now() - a).count();
					for (int i = 0; i < get_blocks; ++i)
						lat[t].push_back(ms);  // one entry per block: the rate column counts blocks
				}
			});
		for (auto &x : th)
			x.join();
		ps.secs = std::chrono::duration<double>(Clock::now() - t0).count();
		for (auto &v : lat)
			ps.lat_ms.insert(ps.lat_ms.end(), v.begin(), v.end());
		std::sort(ps.lat_ms.begin(), ps.lat_ms.end());
	};
	auto run_puts_only = [&](PutStats &ps) {
		std::vector<std::vector<double>> lat(callers);
		std::vector<std::thread> th;
		const auto t0 = Clock::now();
		for (int t = 0; t < callers; ++t)
			th.emplace_back([&, t] {
				for (size_t j = 0; !stop.load(); ++j) {
					const size_t i = (size_t)t * RING + j % RING;
					const auto a = Clock::now();
					if (gbm_batcher_put_block(bt, &hashes[32 * i], data[i].data(), L, 0, NULL) != GBM_OK) {
						fprintf(stderr, "put failed: %s\n", gbm_last_error());
						exit(1);
					}
					lat[t].push_back(std::chrono::duration<double, std::milli>(Clock::now() - a).count());
				}
			});
		for (auto &x : th)
			x.join();
		ps.secs = std::chrono::duration<double>(Clock::now() - t0).count();
		for (auto &v : lat)
			ps.lat_ms.insert(ps.lat_ms.end(), v.begin(), v.end());
		std::sort(ps.lat_ms.begin(), ps.lat_ms.end());
	};
	auto run_puts = [&](PutStats &ps) {
		if (get_blocks > 0)
			run_gets(ps);
		else
			run_puts_only(ps);
	};
	struct ScrubStats {
		uint64_t blocks = 0, corruptions = 0;
		double secs = 0;
	};
	auto run_scrub = [&](ScrubStats &ss) {
		const auto t0 = Clock::now();
		for (int round = 0; resync_mode && !stop.load(); ++round) {
			// every block loses shard (round % 14), resync rebuilds it: gather k, one reconstruct trip, rebuilt shards home
			const int j = round % 14;
			int who[14];
			for (size_t b = 0; b < nscrub; ++b) {
				gbm_storage_nodes_of(mb, &bhashes[32 * b], who);
				gbm_node_delete_shard(mb, who[j], &bhashes[32 * b], j);
				gbm_put_to_resync(mb, &bhashes[32 * b], 0);
			}
			uint64_t st[8];
			if (gbm_resync_run(mb, 0, st) != GBM_OK) {
				fprintf(stderr, "resync failed: %s\n", gbm_last_error());
				exit(1);
			}
			ss.blocks += st[4];
		}
		while (!resync_mode && !stop.load()) {
			uint64_t st[4];
			if (gbm_scrub_all(mb, 512, st) != GBM_OK) {
				fprintf(stderr, "scrub failed: %s\n", gbm_last_error());
				exit(1);
			}
			ss.blocks += st[0];
			ss.corruptions += st[1];
		}
		ss.secs = std::chrono::duration<double>(Clock::now() - t0).count();
	};
	auto sleep_then_stop = [&] {
		std::this_thread::sleep_for(std::chrono::duration<double>(seconds));
		stop = true;
	};
	auto pct = [](const std::vector<double> &v, double p) { return v.empty() ? 0.0 : v[std::min(v.size() - 1, (size_t)(p * v.size()))]; };
	auto report = [&](const char *name, const PutStats *ps, const ScrubStats *ss) {
		printf("%-34s", name);
		if (ps)
			printf(" %s p50 %6.3f ms  p90 %6.3f  p99 %6.3f ms  max %6.3f  %6.2f GiB/s (%zu blocks)", op, pct(ps->lat_ms, 0.5), pct(ps->lat_ms, 0.9),
			       pct(ps->lat_ms, 0.99), pct(ps->lat_ms, 1.0), ps->lat_ms.size() / 1024.0 / ps->secs, ps->lat_ms.size());
		if (ss)
			printf("  scrub %6.2f GiB/s of blocks (%llu blocks, %llu corruptions)", ss->blocks / 1024.0 / ss->secs,
			       (unsigned long long)ss->blocks, (unsigned long long)ss->corruptions);
		printf("\n");
		fflush(stdout);
	};
	// warm both paths (pinned pools, staging slots, kernels)
	{
		PutStats ps;
		ScrubStats ss;
		stop = false;
		std::thread stopper([&] {
			std::this_thread::sleep_for(std::chrono::milliseconds(700));
			stop = true;
		});
		std::thread s([&] { run_scrub(ss); });
		run_puts(ps);
		s.join();
		stopper.join();
	}
	printf("qos_bench: backend %s, %d closed-loop callers (%s%s%s), %.1f s per phase, %s over %zu blocks, scrub tranquility %d\n",
	       gec_codec_backend(c) == GEC_BACKEND_CPU ? "cpu" : "hip", callers, get_blocks > 0 ? "gets of " : "puts through the batcher",
	       get_blocks > 0 ? (std::to_string(get_blocks) + " blocks").c_str() : "",
	       nodes_down ? (", " + std::to_string(nodes_down) + " of 16 nodes down").c_str() : "", seconds, resync_mode ? "resync (one lost shard per block per round)" : "scrub",
	       nscrub, tranq);
	PutStats solo_put, mixed_bg_put, mixed_fg_put;
	ScrubStats solo_scrub, mixed_bg_scrub, mixed_fg_scrub;
	{
		stop = false;
		std::thread stopper(sleep_then_stop);
		run_puts(solo_put);
		stopper.join();
		report(get_blocks > 0 ? "gets alone" : "puts alone", &solo_put, nullptr);
	}
	{
		stop = false;
		std::thread stopper(sleep_then_stop);
		run_scrub(solo_scrub);
		stopper.join();
		report("scrub alone (background class)", nullptr, &solo_scrub);
	}
	const uint64_t y0 = gec_qos_yields(0);
	{
		stop = false;
		std::thread stopper(sleep_then_stop);
		std::thread s([&] { run_scrub(mixed_bg_scrub); });
		run_puts(mixed_bg_put);
		s.join();
		stopper.join();
		report(get_blocks > 0 ? "gets + scrub, background class" : "puts + scrub, background class", &mixed_bg_put, &mixed_bg_scrub);
	}
	const uint64_t y1 = gec_qos_yields(0);
	gbm_set_maintenance_class(mb, 0);
	{
		stop = false;
		std::thread stopper(sleep_then_stop);
		std::thread s([&] { run_scrub(mixed_fg_scrub); });
		run_puts(mixed_fg_put);
		s.join();
		stopper.join();
		report(get_blocks > 0 ? "gets + scrub, no class (round 2)" : "puts + scrub, no class (round 2)", &mixed_fg_put, &mixed_fg_scrub);
	}
	const double p99_solo = pct(solo_put.lat_ms, 0.99), scrub_solo = solo_scrub.blocks / 1024.0 / solo_scrub.secs;
	printf("CU masks: %d (2 = link / checksum masks and the foreground-background partition, 1 = masks without the partition, 0 = refused by "
	       "this runtime: the classes then share every CU, -1 = not used)\n", gec_cu_masks_active());
	printf("with the class:    %s p99 %.2fx solo, scrub at %.0f %% of its solo rate (%llu chunk waits for foreground work)\n", op,
	       pct(mixed_bg_put.lat_ms, 0.99) / p99_solo, 100.0 * (mixed_bg_scrub.blocks / 1024.0 / mixed_bg_scrub.secs) / scrub_solo,
	       (unsigned long long)(y1 - y0));
	printf("without the class: %s p99 %.2fx solo, scrub at %.0f %% of its solo rate\n", op, pct(mixed_fg_put.lat_ms, 0.99) / p99_solo,
	       100.0 * (mixed_fg_scrub.blocks / 1024.0 / mixed_fg_scrub.secs) / scrub_solo);
	gbm_batcher_destroy(bt);
	gbm_destroy(ma);
	gbm_destroy(mb);
	gec_codec_destroy(c);
	return 0;
}
