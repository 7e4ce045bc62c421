// CPU-only exercise of the C++ BlockManager mirror (garage_amd/csrc/block_manager.cpp)
// against the oracle-backed gec stub, meant to run under ASan + UBSan
// (tests/test_sanitizers.py).  Scenarios follow tests/block_manager_cases.py.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/garage_block.h"

extern "C" gec_codec *stub_codec_create(int k, int m);
extern "C" void stub_codec_destroy(gec_codec *c);

#define CHECK(cond)                                                                               \
	do {                                                                                      \
		if (!(cond)) {                                                                    \
			fprintf(stderr, "FAIL %s:%d: %s (gbm: %s)\n", __FILE__, __LINE__, #cond, gbm_last_error()); \
			exit(1);                                                                  \
		}                                                                                 \
	} while (0)

static std::vector<uint8_t> pattern(size_t n, unsigned salt)
{
	std::vector<uint8_t> out;
	out.reserve(n + 1024);
	for (unsigned i = salt; out.size() < n; ++i)
		out.insert(out.end(), (i * 37u) % 1024u, (uint8_t)(i % 256u));
	out.resize(n);
	return out;
}

static void run(int k, int m, const char *dir_root)
{
	gec_codec *codec = stub_codec_create(k, m);
	const int n = k + m, nnodes = n + 2;
	std::vector<std::string> dirs;
	std::vector<const char *> dirp;
	if (dir_root)
		for (int i = 0; i < nnodes; ++i) {
			dirs.push_back(std::string(dir_root) + "/rs" + std::to_string(k) + "_" + std::to_string(m) + "/node" + std::to_string(i));
			dirp.push_back(dirs.back().c_str());
		}
	gbm_manager *mg = nullptr;
	CHECK(gbm_create(codec, n - 1, nullptr, 0, &mg) == GBM_E_INVALID_ARG);  // replication_factor == k+m
	CHECK(gbm_create(codec, nnodes, dir_root ? dirp.data() : nullptr, 0, &mg) == GBM_OK);
	CHECK(gbm_set_data_fsync(nullptr, 1) == GBM_E_INVALID_ARG);
	CHECK(gbm_set_data_fsync(mg, k == 10) == GBM_OK);  // Config.data_fsync on for one of the codes (dir nodes only)

	// put / get round trips, ragged sizes, batched put
	std::vector<std::vector<uint8_t>> blocks;
	for (size_t sz : {(size_t)3073, (size_t)65536, (size_t)500000, (size_t)(1 << 20)})
		blocks.push_back(pattern(sz, (unsigned)sz));
	std::vector<uint8_t> hashes(blocks.size() * 32);
	std::vector<const uint8_t *> ptrs;
	std::vector<size_t> lens;
	for (size_t b = 0; b < blocks.size(); ++b) {
		gbm_blake2sum(blocks[b].data(), blocks[b].size(), hashes.data() + 32 * b);
		ptrs.push_back(blocks[b].data());
		lens.push_back(blocks[b].size());
	}
	CHECK(gbm_rpc_put_blocks(mg, blocks.size(), hashes.data(), ptrs.data(), lens.data()) == GBM_OK);
	std::vector<uint8_t> out(1 << 20);
	size_t got = 0;
	for (size_t b = 0; b < blocks.size(); ++b) {
		CHECK(gbm_rpc_get_block(mg, hashes.data() + 32 * b, out.data(), out.size(), &got) == GBM_OK);
		CHECK(got == blocks[b].size() && std::memcmp(out.data(), blocks[b].data(), got) == 0);
		CHECK(gbm_block_incref(mg, hashes.data() + 32 * b) == GBM_OK);
	}
	CHECK(gbm_rpc_get_block(mg, hashes.data(), out.data(), 100, &got) == GBM_E_BUFFER_TOO_SMALL && got == 3073);

	// m nodes down (data shards first): still readable through a decode; one more: MissingBlock
	const uint8_t *h = hashes.data() + 32 * 2;
	std::vector<int> who(n);
	CHECK(gbm_storage_nodes_of(mg, h, who.data()) == GBM_OK);
	for (int j = 0; j < m; ++j)
		CHECK(gbm_node_set_down(mg, who[j], 1) == GBM_OK);
	CHECK(gbm_rpc_get_block(mg, h, out.data(), out.size(), &got) == GBM_OK && got == 500000);
	CHECK(std::memcmp(out.data(), blocks[2].data(), got) == 0);
	CHECK(gbm_node_set_down(mg, who[m], 1) == GBM_OK);
	CHECK(gbm_rpc_get_block(mg, h, out.data(), out.size(), &got) == GBM_E_MISSING_BLOCK);
	// write quorum
	CHECK(gbm_rpc_put_block(mg, h, blocks[2].data(), blocks[2].size()) == GBM_E_QUORUM);
	for (int j = 0; j <= m; ++j)
		gbm_node_set_down(mg, who[j], 0);

	// corrupt one shard (+ delete one if the code can take it): read repairs around it, resync rewrites
	CHECK(gbm_node_corrupt_shard(mg, who[1], h, 1, 1234, 0x55, 0) == GBM_OK);
	int lost = 1;
	if (m >= 2) {
		CHECK(gbm_node_delete_shard(mg, who[k], h, k) == GBM_OK);
		lost = 2;
	}
	CHECK(gbm_rpc_get_block(mg, h, out.data(), out.size(), &got) == GBM_OK);
	CHECK(std::memcmp(out.data(), blocks[2].data(), got) == 0);
	uint64_t met[6];
	CHECK(gbm_metrics(mg, met) == GBM_OK && met[2] == 1 && met[3] >= 2);
	int changed = -1;
	CHECK(gbm_resync_all(mg, &changed) == GBM_OK);
	CHECK(changed >= lost);
	for (int j = 0; j < n; ++j)
		CHECK(gbm_node_has_shard(mg, who[j], h, j));

	// scrub: clean, then silent corruption with a re-stamped checksum
	std::vector<uint8_t> bad(blocks.size());
	CHECK(gbm_scrub(mg, blocks.size(), hashes.data(), bad.data()) == GBM_OK);
	for (uint8_t x : bad)
		CHECK(x == 0);
	CHECK(gbm_node_corrupt_shard(mg, who[k], h, k, 77, 1, 1) == GBM_OK);
	CHECK(gbm_scrub(mg, blocks.size(), hashes.data(), bad.data()) == GBM_OK);
	CHECK(bad[2] == 1 && bad[0] == 0 && bad[1] == 0 && bad[3] == 0);

	// wrong content under a valid name -> CorruptData
	CHECK(gbm_rpc_put_block(mg, hashes.data(), blocks[1].data(), blocks[1].size()) == GBM_OK);
	CHECK(gbm_rpc_get_block(mg, hashes.data(), out.data(), out.size(), &got) == GBM_E_CORRUPT_DATA);

	// compression (zstd frame + checksum), if libzstd is there
	if (gbm_set_compression_level(mg, 1, 1) == GBM_OK) {
		std::vector<uint8_t> z = pattern(800000, 9);
		uint8_t hz[32];
		gbm_blake2sum(z.data(), z.size(), hz);
		CHECK(gbm_rpc_put_block(mg, hz, z.data(), z.size()) == GBM_OK);
		CHECK(gbm_rpc_get_block(mg, hz, out.data(), out.size(), &got) == GBM_OK && got == z.size());
		CHECK(std::memcmp(out.data(), z.data(), got) == 0);
		gbm_set_compression_level(mg, 0, 0);
	}

	// rc -> 0: resync deletes every shard
	CHECK(gbm_block_decref(mg, hashes.data() + 32 * 3) == GBM_OK);
	CHECK(gbm_resync_all(mg, &changed) == GBM_OK && changed >= n);
	CHECK(gbm_rpc_get_block(mg, hashes.data() + 32 * 3, out.data(), out.size(), &got) == GBM_E_MISSING_BLOCK);

	gbm_destroy(mg);
	stub_codec_destroy(codec);
	printf("RS(%d,%d) %s nodes: OK\n", k, m, dir_root ? "directory" : "memory");
}

// 8 caller threads x 6 blocks through the coalescing batcher while 2 reader threads get
// blocks that are already stored: results correct, and the worker really coalesced.
static void run_batcher(int k, int m)
{
	gec_codec *codec = stub_codec_create(k, m);
	gbm_manager *mg = nullptr;
	CHECK(gbm_create(codec, k + m + 1, nullptr, 0, &mg) == GBM_OK);
	gbm_batcher *bt = nullptr;
	CHECK(gbm_batcher_create(mg, 0, 100, &bt) == GBM_E_INVALID_ARG);
	CHECK(gbm_batcher_create(mg, 16, 20000, &bt) == GBM_OK);
	const int T = 8, PER = 6;
	std::vector<std::vector<uint8_t>> blocks(T * PER);
	std::vector<uint8_t> hashes(T * PER * 32);
	for (int i = 0; i < T * PER; ++i) {
		blocks[i] = pattern(40000 + 997 * i, 1000 + i);
		gbm_blake2sum(blocks[i].data(), blocks[i].size(), hashes.data() + 32 * i);
	}
	std::vector<int> rcs(T * PER, -999);
	std::vector<std::thread> th;
	for (int t = 0; t < T; ++t)
		th.emplace_back([&, t] {
			for (int j = 0; j < PER; ++j) {
				const int i = t * PER + j;
				rcs[i] = gbm_batcher_put_block(bt, hashes.data() + 32 * i, blocks[i].data(), blocks[i].size());
			}
		});
	std::vector<int> reader_ok(2, 1);
	for (int r = 0; r < 2; ++r)
		th.emplace_back([&, r] {
			std::vector<uint8_t> out(200000);
			for (int round = 0; round < 40; ++round)
				for (int i = r; i < T * PER; i += 7) {
					size_t got = 0;
					int rc = gbm_rpc_get_block(mg, hashes.data() + 32 * i, out.data(), out.size(), &got);
					if (rc == GBM_OK && (got != blocks[i].size() || std::memcmp(out.data(), blocks[i].data(), got)))
						reader_ok[r] = 0;  // a block is either not there yet or exactly right
					else if (rc != GBM_OK && rc != GBM_E_MISSING_BLOCK)
						reader_ok[r] = 0;
				}
		});
	for (auto &x : th)
		x.join();
	CHECK(reader_ok[0] && reader_ok[1]);
	std::vector<uint8_t> out(200000);
	for (int i = 0; i < T * PER; ++i) {
		CHECK(rcs[i] == GBM_OK);
		size_t got = 0;
		CHECK(gbm_rpc_get_block(mg, hashes.data() + 32 * i, out.data(), out.size(), &got) == GBM_OK);
		CHECK(got == blocks[i].size() && std::memcmp(out.data(), blocks[i].data(), got) == 0);
	}
	uint64_t st[3];
	CHECK(gbm_batcher_stats(bt, st) == GBM_OK);
	CHECK(st[1] == (uint64_t)T * PER && st[0] < st[1] && st[2] >= 2 && st[2] <= 16);
	// block_ram_buffer_max: with a budget of ~2 blocks the 8 callers still all get through (they wait for
	// permits), batches can no longer exceed the budget, and an oversized block is refused, not queued
	CHECK(gbm_batcher_set_ram_buffer_max(nullptr, 1 << 20) == GBM_E_INVALID_ARG);
	CHECK(gbm_batcher_set_ram_buffer_max(bt, 100 * 1024) == GBM_OK);
	uint64_t before[3], after[3];
	CHECK(gbm_batcher_stats(bt, before) == GBM_OK);
	{
		std::vector<std::thread> th2;
		std::vector<int> rc2(T, -999);
		for (int t = 0; t < T; ++t)
			th2.emplace_back([&, t] { rc2[t] = gbm_batcher_put_block(bt, hashes.data() + 32 * t, blocks[t].data(), blocks[t].size()); });
		for (auto &x : th2)
			x.join();
		for (int t = 0; t < T; ++t)
			CHECK(rc2[t] == GBM_OK);
	}
	CHECK(gbm_batcher_stats(bt, after) == GBM_OK);
	CHECK(after[1] == before[1] + T && after[0] - before[0] >= (uint64_t)T / 2);  // <= 2 blocks (~40-47 KB each) per batch
	std::vector<uint8_t> big(300 * 1024, 7);
	uint8_t bigh[32];
	gbm_blake2sum(big.data(), big.size(), bigh);
	CHECK(gbm_batcher_put_block(bt, bigh, big.data(), big.size()) == GBM_E_INVALID_ARG);
	CHECK(gbm_batcher_set_ram_buffer_max(bt, 256u << 20) == GBM_OK);
	// a quorum failure is reported to the caller whose block it was
	std::vector<int> who(k + m);
	CHECK(gbm_storage_nodes_of(mg, hashes.data(), who.data()) == GBM_OK);
	for (int j = 0; j < m; ++j)
		gbm_node_set_down(mg, who[j], 1);
	CHECK(gbm_batcher_put_block(bt, hashes.data(), blocks[0].data(), blocks[0].size()) == GBM_E_QUORUM);
	gbm_batcher_destroy(bt);
	gbm_destroy(mg);
	stub_codec_destroy(codec);
	printf("batcher RS(%d,%d): %llu blocks in %llu device batches (largest %llu): OK\n", k, m,
	       (unsigned long long)st[1], (unsigned long long)st[0], (unsigned long long)st[2]);
}

int main(int argc, char **argv)
{
	run(3, 1, nullptr);
	run(10, 4, nullptr);
	if (argc > 1)
		run(10, 4, argv[1]);
	run_batcher(10, 4);
	printf("block_manager_host_test: all scenarios OK\n");
	return 0;
}
