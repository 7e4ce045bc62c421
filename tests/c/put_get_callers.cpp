// put_get_callers.cpp -- SURVEY.md section 8 row a10 as a native test: the callers of the block path, the way the
// S3 layer drives them, through the C API of libgarage_block:
//
//   PutObject   put_block_and_meta keeps at most PUT_BLOCKS_MAX_PARALLEL = 3 block puts of one request in flight
//               (/root/reference/src/api/s3/put.rs:42,486-511), every block tagged OrderTag(stream of the request, index);
//               many requests run at once.  Here: R request threads, each submitting its blocks in order through
//               gbm_batcher_submit and keeping <= 3 tickets pending (gbm_batcher_wait = `.await` on the oldest).
//   GetObject   the blocks of an object are fetched with a 2-deep prefetch (src/api/s3/get.rs:429, `.buffered(2)`),
//               tagged the same way.  Here: G reader threads with 2 slots each, reading objects that were put
//               earlier while the writers are still writing theirs.
//   UploadPartCopy  the source object's blocks come through EncryptionParams::get_block = rpc_get_block_streaming with an
//               OrderTag, two in flight and consumed in order (src/api/s3/copy.rs:520-551, `.buffered(2)`;
//               src/api/s3/encryption.rs:269-280); each one is re-encrypted, named by the blake2sum of the NEW bytes and
//               put with prevent_compression and NO tag while the next source block is being read (try_join!, copy.rs:606-630).
//               Here: copier threads doing exactly that with an XOR key stream, then the copies are read back.
//
//   ranged GetObject  body_from_blocks_range (src/api/s3/get.rs:650-743): the blocks that intersect [begin, end) with their true
//               offsets in the object, fetched one after the other with an OrderTag, each cut to the range.  Here:
//               gbm_rpc_get_block_range_streaming per block -- only the data shards the range touches are read -- by a
//               thread per reader, over ranges that start and end inside blocks, while the writers are writing.
//
// Checked: every put is acknowledged; the batcher coalesced (fewer device batches than blocks, some batch > 1 block);
// no node ever saw a stream's PutShards out of `order` (gbm_node_order_violations == 0 everywhere) although blocks of
// one stream land in different batches on different workers; with a RAM budget of two blocks (block_ram_buffer_max,
// src/block/manager.rs:380-384) no batch ever holds more than two blocks and everything still completes; every
// byte of every object reads back.
//
// Two builds (tests/c/Makefile): `put_get_callers_tsan` links the host-only product sources (CPU backend) under
// ThreadSanitizer; `put_get_callers` links the real libraries -- GEC_BACKEND_AUTO picks the GPU on a GPU box.
// With [devices] > 1 the manager is a multi-device one (gbm_create_multi: one codec, one coalescing queue and one set of
// workers per device, blocks routed by gec_device_of_hash; on a one-GPU box every codec sits on device 0): the same
// assertions, plus per-device block counts that follow the hash exactly -- and the order guarantee now has to hold for
// streams whose blocks are encoded on DIFFERENT devices.
// usage: put_get_callers [requests] [blocks_per_object] [block_bytes] [readers] [devices]
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <future>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/garage_block.h"

#define CHECK(cond)                                                                                              \
	do {                                                                                                     \
		if (!(cond)) {                                                                                   \
			fprintf(stderr, "FAIL %s:%d: %s (gbm: %s)\n", __FILE__, __LINE__, #cond, gbm_last_error()); \
			exit(1);                                                                                 \
		}                                                                                                \
	} while (0)

namespace {

constexpr int K = 10, M = 4, NNODES = 16;
constexpr int PUT_BLOCKS_MAX_PARALLEL = 3;  // put.rs:42
constexpr int GET_PREFETCH = 2;             // get.rs:429

struct Object {
	uint64_t stream_id;
	std::vector<std::vector<uint8_t>> blocks;
	std::vector<uint8_t> hashes;  // 32 bytes per block
};

void fill_block(std::vector<uint8_t> &b, uint64_t seed)
{
	uint64_t x = seed * 0x9E3779B97F4A7C15ull + 0x6761726167650010ull;
	for (size_t i = 0; i + 8 <= b.size(); i += 8) {
		x ^= x << 13;
		x ^= x >> 7;
		x ^= x << 17;
		std::memcpy(&b[i], &x, 8);
	}
}

Object make_object(uint64_t stream_id, int nblocks, size_t block_bytes, bool short_last = true)
{
	Object o;
	o.stream_id = stream_id;
	o.blocks.resize(nblocks);
	o.hashes.resize((size_t)nblocks * 32);
	for (int i = 0; i < nblocks; ++i) {
		// the last block of an object is short, like a real upload's
		o.blocks[i].assign(short_last && i + 1 == nblocks ? block_bytes / 2 + 7 : block_bytes, 0);
		fill_block(o.blocks[i], stream_id * 1000 + i);
		gbm_blake2sum(o.blocks[i].data(), o.blocks[i].size(), &o.hashes[(size_t)i * 32]);
	}
	return o;
}

// PutObject: the request submits its blocks in order and keeps at most PUT_BLOCKS_MAX_PARALLEL of them pending --
// `buffered(PUT_BLOCKS_MAX_PARALLEL)` over futures created in block order (put.rs:486-511)
void put_object(gbm_batcher *bt, const Object &o, std::atomic<uint64_t> &put_ns, std::atomic<uint64_t> &nput)
{
	struct Pending {
		gbm_put_ticket *tk;
		std::chrono::steady_clock::time_point t0;
	};
	std::vector<Pending> pending;
	std::vector<gbm_order_tag> tags(o.blocks.size());
	auto finish_oldest = [&] {
		CHECK(gbm_batcher_wait(pending.front().tk) == GBM_OK);
		put_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - pending.front().t0).count();
		++nput;
		pending.erase(pending.begin());
	};
	for (size_t i = 0; i < o.blocks.size(); ++i) {
		if (pending.size() == (size_t)PUT_BLOCKS_MAX_PARALLEL)
			finish_oldest();
		tags[i] = gbm_order_tag{o.stream_id, (uint64_t)i};
		Pending p{nullptr, std::chrono::steady_clock::now()};
		CHECK(gbm_batcher_submit(bt, &o.hashes[i * 32], o.blocks[i].data(), o.blocks[i].size(), 0, &tags[i], &p.tk) == GBM_OK);
		pending.push_back(p);
	}
	while (!pending.empty())
		finish_oldest();
}

// GetObject: GET_PREFETCH slots fetch the object's blocks; every byte is compared
// (through the batcher's read side when `bt` is given: concurrent GetObjects share gather rounds and device trips)
void get_object(gbm_manager *mg, const Object &o, uint64_t read_stream, std::atomic<uint64_t> &nget, gbm_batcher *bt = nullptr)
{
	std::atomic<int> next{0};
	std::vector<std::thread> slots;
	for (int s = 0; s < GET_PREFETCH; ++s)
		slots.emplace_back([&] {
			std::vector<uint8_t> buf;
			for (;;) {
				const int i = next.fetch_add(1);
				if (i >= (int)o.blocks.size())
					return;
				buf.assign(o.blocks[i].size() + 64, 0xEE);
				size_t len = 0;
				const gbm_order_tag tag{read_stream, (uint64_t)i};
				if (bt)
					CHECK(gbm_batcher_get_block(bt, &o.hashes[(size_t)i * 32], buf.data(), buf.size(), &len) == GBM_OK);
				else
					CHECK(gbm_rpc_get_block(mg, &o.hashes[(size_t)i * 32], &tag, buf.data(), buf.size(), &len) == GBM_OK);
				CHECK(len == o.blocks[i].size() && std::memcmp(buf.data(), o.blocks[i].data(), len) == 0);
				++nget;
			}
		});
	for (auto &t : slots)
		t.join();
}

// A ranged GetObject (get.rs:650-743): blocks that intersect [begin, end), in order, each asked for its part of the range
void get_object_range(gbm_manager *mg, const Object &o, uint64_t begin, uint64_t end, uint64_t read_stream, std::atomic<uint64_t> &nranges)
{
	std::vector<uint8_t> want, got;
	uint64_t block_offset = 0;
	uint64_t order = 0;
	for (size_t i = 0; i < o.blocks.size() && block_offset < end; ++i) {
		const uint64_t size = o.blocks[i].size();
		if (block_offset + size > begin) {  // "keep only blocks that have an intersection with the requested range"
			const uint64_t b = begin > block_offset ? begin - block_offset : 0;
			const uint64_t e = std::min<uint64_t>(size, end - block_offset);
			want.insert(want.end(), o.blocks[i].begin() + (ptrdiff_t)b, o.blocks[i].begin() + (ptrdiff_t)e);
			const gbm_order_tag tag{read_stream, order++};
			auto sink = [](void *ctx, const uint8_t *chunk, size_t len) -> int {
				auto *v = static_cast<std::vector<uint8_t> *>(ctx);
				v->insert(v->end(), chunk, chunk + len);
				return 0;
			};
			// `end` is handed over unclipped for the blocks in the middle, as the reference's scan sees it
			CHECK(gbm_rpc_get_block_range_streaming(mg, &o.hashes[i * 32], &tag, (size_t)size, (size_t)b, (size_t)(end - block_offset), 16384, sink, &got) == GBM_OK);
		}
		block_offset += size;
	}
	CHECK(got == want);
	++nranges;
}

// UploadPartCopy (copy.rs:520-630): stream the source blocks in (2 in flight, in order), re-encrypt, put under the new name
// while the next block arrives.  Returns the destination object (what must read back).
Object copy_object(gbm_manager *mg, gbm_batcher *bt, const Object &src, uint64_t read_stream, uint8_t key, std::atomic<uint64_t> &ncopied)
{
	Object dst;
	dst.stream_id = read_stream;
	dst.blocks.resize(src.blocks.size());
	dst.hashes.resize(src.hashes.size());
	auto fetch = [&](size_t i) {  // EncryptionParams::get_block: the streaming get, chunks appended as they arrive
		std::vector<uint8_t> data;
		data.reserve(src.blocks[i].size());
		const gbm_order_tag tag{read_stream, (uint64_t)i};
		auto sink = [](void *ctx, const uint8_t *chunk, size_t len) -> int {
			auto *v = static_cast<std::vector<uint8_t> *>(ctx);
			v->insert(v->end(), chunk, chunk + len);
			return 0;
		};
		CHECK(gbm_rpc_get_block_streaming(mg, &src.hashes[i * 32], &tag, 16384, sink, &data) == GBM_OK);
		return data;
	};
	std::deque<std::future<std::vector<uint8_t>>> ahead;
	size_t next_fetch = 0;
	auto refill = [&] {
		while (ahead.size() < (size_t)GET_PREFETCH && next_fetch < src.blocks.size()) {
			const size_t i = next_fetch++;
			ahead.push_back(std::async(std::launch::async, fetch, i));
		}
	};
	refill();
	for (size_t i = 0; i < src.blocks.size(); ++i) {
		std::vector<uint8_t> data = ahead.front().get();
		ahead.pop_front();
		CHECK(data == src.blocks[i]);
		for (size_t j = 0; j < data.size(); ++j)  // "dest_encryption.encrypt_block": the bytes change, so does the name
			data[j] ^= (uint8_t)(key + j * 31);
		gbm_blake2sum(data.data(), data.size(), &dst.hashes[i * 32]);
		dst.blocks[i] = std::move(data);
		// try_join!(rpc_put_block(final_hash, final_data, is_encrypted, None), ..., defragmenter.next())
		gbm_put_ticket *tk = nullptr;
		CHECK(gbm_batcher_submit(bt, &dst.hashes[i * 32], dst.blocks[i].data(), dst.blocks[i].size(), /*prevent_compression=*/1, nullptr, &tk) == GBM_OK);
		refill();
		if (!ahead.empty())
			ahead.front().wait();
		CHECK(gbm_batcher_wait(tk) == GBM_OK);
		++ncopied;
	}
	return dst;
}

}  // namespace

int main(int argc, char **argv)
{
	const int requests = argc > 1 ? atoi(argv[1]) : 8;
	const int per_object = argc > 2 ? atoi(argv[2]) : 9;
	const size_t block_bytes = argc > 3 ? (size_t)atol(argv[3]) : 65536;
	const int readers = argc > 4 ? atoi(argv[4]) : 3;
	const int ndev = argc > 5 ? atoi(argv[5]) : 1;

	std::vector<gec_codec *> codecs((size_t)ndev, nullptr);
	for (auto &c : codecs)
		CHECK(gec_codec_create(K, M, GEC_BACKEND_AUTO, 0, &c) == GEC_OK);
	gec_codec *codec = codecs[0];
	gbm_manager *mg = nullptr;
	if (ndev > 1)
		CHECK(gbm_create_multi(codecs.data(), ndev, NNODES, nullptr, 0, &mg) == GBM_OK);
	else
		CHECK(gbm_create(codec, NNODES, nullptr, 0, &mg) == GBM_OK);
	CHECK(gbm_device_count(mg) == ndev);
	gbm_batcher *bt = nullptr;
	CHECK(gbm_batcher_create(mg, 64, 300, &bt) == GBM_OK);

	// objects the readers fetch while the writers are busy: put first, through the same batcher
	std::vector<Object> old_objs, new_objs;
	for (int r = 0; r < readers; ++r)
		old_objs.push_back(make_object(1000 + r, per_object, block_bytes));
	for (int r = 0; r < requests; ++r)
		new_objs.push_back(make_object(1 + r, per_object, block_bytes));
	std::atomic<uint64_t> put_ns{0}, nput{0}, nget{0};
	{
		std::vector<std::thread> th;
		for (const Object &o : old_objs)
			th.emplace_back([&] { put_object(bt, o, put_ns, nput); });
		for (auto &t : th)
			t.join();
	}
	uint64_t st0[3];
	CHECK(gbm_batcher_stats(bt, st0) == GBM_OK);

	// ---- the mixed phase: R PutObjects and G GetObjects at once
	put_ns = 0;
	nput = 0;
	const int ncopiers = readers > 0 ? 2 : 0;
	std::vector<Object> copies((size_t)ncopiers);
	std::atomic<uint64_t> ncopied{0}, nranges{0};
	const auto t0 = std::chrono::steady_clock::now();
	{
		std::vector<std::thread> th;
		for (const Object &o : new_objs)
			th.emplace_back([&] { put_object(bt, o, put_ns, nput); });
		for (int r = 0; r < readers; ++r)
			th.emplace_back([&, r] {
				for (int pass = 0; pass < 3; ++pass)
					get_object(mg, old_objs[r], 5000 + r * 10 + pass, nget, bt);
			});
		// ranged GetObjects of the same objects: inside one block, across two, from the middle of the first block to the object's end
		for (int r = 0; r < readers; ++r)
			th.emplace_back([&, r] {
				const Object &o = old_objs[(size_t)r];
				uint64_t total = 0;
				for (const auto &b : o.blocks)
					total += b.size();
				const uint64_t bb = block_bytes;
				const uint64_t ranges[][2] = {{bb / 3, bb / 3 + 1000}, {bb - 100, bb + 100}, {bb / 2, total}, {0, 1}, {total - 5, total + 50},
							      {2 * bb + 17, std::min<uint64_t>(total, 5 * bb - 3)}};
				for (size_t q = 0; q < sizeof ranges / sizeof ranges[0]; ++q)
					if (ranges[q][0] < total && ranges[q][0] < ranges[q][1])
						get_object_range(mg, o, ranges[q][0], ranges[q][1], 6000 + (uint64_t)r * 10 + q, nranges);
			});
		// two UploadPartCopy requests beside them (their puts are untagged and counted below)
		for (int c = 0; c < ncopiers; ++c)
			th.emplace_back([&, c] { copies[(size_t)c] = copy_object(mg, bt, old_objs[(size_t)c % old_objs.size()], 7000 + c, (uint8_t)(0x5C + c), ncopied); });
		for (auto &t : th)
			t.join();
	}
	const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	uint64_t st1[3];
	CHECK(gbm_batcher_stats(bt, st1) == GBM_OK);
	const uint64_t batches = st1[0] - st0[0], blocks = st1[1] - st0[1];
	const double mean_put_ms = blocks ? put_ns.load() / 1e6 / (double)blocks : 0.0;
	CHECK(blocks == (uint64_t)(requests + ncopiers) * per_object && nput.load() + ncopied.load() == blocks);
	CHECK(nget.load() == (uint64_t)readers * 3 * per_object);
	CHECK(readers == 0 || nranges.load() >= (uint64_t)readers * 4);
	// coalescing: concurrent requests share device batches
	if (requests >= 4)
		CHECK(batches < blocks && st1[2] >= 2);
	uint64_t gst[3];
	CHECK(gbm_batcher_get_stats(bt, gst) == GBM_OK);
	CHECK(gst[1] == (uint64_t)readers * 3 * per_object && gst[0] <= gst[1]);
	if (readers >= 3)  // six prefetch slots re-submit together whenever a batch completes
		CHECK(gst[2] >= 2);
	{  // every device's queue took exactly the blocks gec_device_of_hash gives it
		std::vector<uint64_t> want_put((size_t)ndev, 0), want_get((size_t)ndev, 0);
		for (const Object &o : old_objs)
			for (size_t i = 0; i < o.blocks.size(); ++i) {
				const int d = gec_device_of_hash(&o.hashes[i * 32], ndev);
				CHECK(d == gbm_device_of_hash(mg, &o.hashes[i * 32]));
				want_put[(size_t)d] += 1;
				want_get[(size_t)d] += 3;
			}
		for (const Object &o : new_objs)
			for (size_t i = 0; i < o.blocks.size(); ++i)
				want_put[(size_t)gec_device_of_hash(&o.hashes[i * 32], ndev)] += 1;
		for (const Object &o : copies)
			for (size_t i = 0; i < o.blocks.size(); ++i)
				want_put[(size_t)gec_device_of_hash(&o.hashes[i * 32], ndev)] += 1;
		uint64_t sum_put = 0;
		for (int d = 0; d < ndev; ++d) {
			uint64_t p[3], g[3], met[6];
			CHECK(gbm_batcher_device_stats(bt, d, p, g) == GBM_OK && gbm_device_metrics(mg, d, met) == GBM_OK);
			CHECK(p[1] == want_put[(size_t)d] && g[1] == want_get[(size_t)d] && met[4] == want_put[(size_t)d]);
			sum_put += p[1];
		}
		CHECK(sum_put == st1[1]);
	}
	{  // the read side reports a missing block and a short buffer like gbm_rpc_get_block does
		uint8_t nohash[32], small[8];
		std::memset(nohash, 0x5A, sizeof nohash);
		size_t len = 0;
		CHECK(gbm_batcher_get_block(bt, nohash, small, sizeof small, &len) == GBM_E_MISSING_BLOCK);
		CHECK(gbm_batcher_get_block(bt, &new_objs[0].hashes[0], small, sizeof small, &len) == GBM_E_BUFFER_TOO_SMALL);
	}
	// OrderTag: no node saw a stream's blocks out of order, although they crossed batches and batcher workers
	for (int nd = 0; nd < NNODES; ++nd)
		CHECK(gbm_node_order_violations(mg, nd) == 0);
	// every byte of what the writers put reads back (after the fact, whole objects)
	for (const Object &o : new_objs)
		get_object(mg, o, 9000 + o.stream_id, nget);
	for (const Object &o : copies)  // ... and every re-encrypted copy
		get_object(mg, o, 9200 + o.stream_id, nget);

	// ---- RAM permits: a budget of two blocks -- the queue can never hold more than two blocks, so no batch can
	gbm_batcher_destroy(bt);
	CHECK(gbm_batcher_create(mg, 64, 300, &bt) == GBM_OK);
	CHECK(gbm_batcher_set_ram_buffer_max(bt, 2 * block_bytes * (size_t)ndev) == GBM_OK);  // (the budget is shared out per device)
	std::vector<Object> tight;
	for (int r = 0; r < 4; ++r)
		tight.push_back(make_object(2000 + r, 4, block_bytes, /*short_last=*/false));
	{
		std::vector<std::thread> th;
		for (const Object &o : tight)
			th.emplace_back([&] { put_object(bt, o, put_ns, nput); });
		for (auto &t : th)
			t.join();
	}
	uint64_t st2[3];
	CHECK(gbm_batcher_stats(bt, st2) == GBM_OK);
	CHECK(st2[1] == 16 && st2[2] <= 2);
	for (const Object &o : tight)
		get_object(mg, o, 9500 + o.stream_id, nget);
	for (int nd = 0; nd < NNODES; ++nd)
		CHECK(gbm_node_order_violations(mg, nd) == 0);

	const double mib = (double)blocks * (double)block_bytes / (1 << 20);
	printf("put_get_callers: backend %s, %d device(s), %d PutObjects x %d blocks of %zu bytes (<=%d in flight each) beside %d GetObjects (prefetch %d), %llu ranged GetObjects and %d UploadPartCopies: "
	       "%llu blocks in %llu device batches (largest %llu), mean put %.3f ms, %.2f GiB/s put; %llu blocks read in %llu batches (largest %llu); "
	       "0 order violations; all bytes round-trip: OK\n",
	       gec_codec_backend(codec) == GEC_BACKEND_CPU ? "cpu" : "hip", ndev, requests, per_object, block_bytes, PUT_BLOCKS_MAX_PARALLEL, readers,
	       GET_PREFETCH, (unsigned long long)nranges.load(), ncopiers, (unsigned long long)blocks, (unsigned long long)batches, (unsigned long long)st1[2],
	       mean_put_ms, mib / 1024.0 / secs, (unsigned long long)gst[1], (unsigned long long)gst[0], (unsigned long long)gst[2]);
	gbm_batcher_destroy(bt);
	gbm_destroy(mg);
	for (gec_codec *c : codecs)
		gec_codec_destroy(c);
	return 0;
}
