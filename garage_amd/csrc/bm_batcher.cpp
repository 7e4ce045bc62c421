// bm_batcher.cpp -- the coalescing queue in front of the FFI: many callers with <= 3 puts in flight each
// (PUT_BLOCKS_MAX_PARALLEL, src/api/s3/put.rs:42,486-511) -> a few device batches, with RAM permits
// (buffer_kb_semaphore, src/block/manager.rs:380-384).
#include "bm_internal.hpp"

using namespace gbmimpl;

// ------------------------------------------------------------------ batcher
// The coalescing queue in front of the FFI.  Garage keeps <= 3 block puts in flight
// per PutObject (PUT_BLOCKS_MAX_PARALLEL, src/api/s3/put.rs:42,486-511) and serves
// many requests at once; each caller blocks in gbm_batcher_put_block (the way
// `rpc_put_block(...).await` suspends) while a worker thread turns whatever has
// queued up within max_wait_us (or max_blocks) into a single device batch.  GBM_BATCHER_WORKERS (default 2) batches
// are in flight at a time: while one is on the device the next one forms and starts.  Batches that carry order tags
// take a ticket when they are formed and hand their shards to the nodes in ticket order (their device trips still
// overlap), so the OrderTag guarantee -- requests of one stream reach a node in `order` order -- holds across
// batches as well as inside one.
struct gbm_batcher {
	struct Item {
		const uint8_t *hash, *data;
		size_t len;
		uint8_t prevent_compression = 0;
		bool has_tag = false;
		gbm_order_tag tag{0, 0};
		int rc = GBM_OK;
		bool done = false;
	};
	gbm_manager *mg = nullptr;
	size_t max_blocks = 64;
	unsigned max_wait_us = 200;
	// buffer_kb_semaphore (src/block/manager.rs:96,156,380-384): KiB permits for the bytes of blocks on
	// their way to the storage nodes, Config.block_ram_buffer_max (default 256 MiB, src/util/config.rs:276-278)
	size_t ram_permits_kb = 256 * 1024, ram_in_use_kb = 0;
	std::mutex mu;
	std::condition_variable cv_work, cv_done, cv_ram;
	std::deque<Item *> queue;
	bool stop = false, forming = false;
	uint64_t batches = 0, blocks = 0, max_batch = 0;
	// fan-out turnstile of the tagged batches
	uint64_t next_ticket = 0, serving = 0;
	std::condition_variable cv_turn;
	// two workers: while one batch is on the device the next one forms and starts (the device trip has a latency
	// floor -- the checksum chain -- that a single worker would pay serially)
	std::vector<std::thread> workers;

	// ---- the read side: GetObject's readers (a few blocks ahead each, src/api/s3/get.rs:429) coalesced the same way.
	// Sixteen readers fetching eight blocks each through gbm_rpc_get_blocks make sixteen device trips that queue up
	// behind one another and behind the host pool (4.3 GiB/s, 26 ms per get); through here they make a few.
	struct GetItem {
		const uint8_t *hash;
		uint8_t *out;
		size_t cap, len = 0;
		int rc = GBM_OK;
		bool done = false;
	};
	std::mutex gmu;
	std::condition_variable gcv_work, gcv_done;
	std::deque<GetItem *> gqueue;
	bool gforming = false;
	uint64_t gbatches = 0, gblocks = 0, gmax_batch = 0;
	std::vector<std::thread> gworkers;

	void run_gets()
	{
		std::unique_lock<std::mutex> lk(gmu);
		for (;;) {
			gcv_work.wait(lk, [&] { return stop_gets || (!gqueue.empty() && !gforming); });
			if (gqueue.empty()) {
				if (stop_gets)
					return;
				continue;
			}
			gforming = true;
			const auto deadline = std::chrono::system_clock::now() + std::chrono::microseconds(max_wait_us);
			const auto gap = std::chrono::microseconds(std::max(20u, max_wait_us / 6));
			size_t seen = gqueue.size();
			while (!stop_gets && gqueue.size() < max_blocks) {  // the put side's linger: it ends once arrivals stop
				const auto now = std::chrono::system_clock::now();
				if (now >= deadline)
					break;
				if (gcv_work.wait_until(lk, std::min(deadline, now + gap)) == std::cv_status::timeout && gqueue.size() == seen)
					break;
				seen = gqueue.size();
			}
			std::vector<GetItem *> batch;
			while (!gqueue.empty() && batch.size() < max_blocks) {
				batch.push_back(gqueue.front());
				gqueue.pop_front();
			}
			gforming = false;
			gcv_work.notify_all();
			lk.unlock();
			const size_t nb = batch.size();
			std::vector<uint8_t> hashes(nb * 32);
			std::vector<uint8_t *> outs(nb);
			std::vector<size_t> caps(nb), lens(nb, 0);
			std::vector<int> rcs(nb, GBM_E_MISSING_BLOCK);
			for (size_t i = 0; i < nb; ++i) {
				std::memcpy(hashes.data() + 32 * i, batch[i]->hash, 32);
				outs[i] = batch[i]->out;
				caps[i] = batch[i]->cap;
			}
			int rc;
			try {
				rc = get_blocks_impl(mg, nb, hashes.data(), nullptr, outs.data(), caps.data(), lens.data(), rcs.data(), false, nullptr);
			} catch (const std::exception &) {
				rc = GBM_E_IO;
			}
			if (rc != GBM_OK)  // the whole call failed (a device error, out of memory): nobody of this batch has a block
				std::fill(rcs.begin(), rcs.end(), rc);
			lk.lock();
			for (size_t i = 0; i < nb; ++i) {
				batch[i]->rc = rcs[i];
				batch[i]->len = lens[i];
				batch[i]->done = true;
			}
			++gbatches;
			gblocks += nb;
			gmax_batch = std::max<uint64_t>(gmax_batch, nb);
			gcv_done.notify_all();
		}
	}
	bool stop_gets = false;

	void run()
	{
		std::unique_lock<std::mutex> lk(mu);
		for (;;) {
			// one worker forms a batch at a time; the other one is either on the device or waits its turn
			cv_work.wait(lk, [&] { return stop || (!queue.empty() && !forming); });
			if (queue.empty()) {
				if (stop)
					return;
				continue;
			}
			forming = true;
			// linger a little so concurrent callers land in the same batch
			// system_clock: libstdc++ maps it to pthread_cond_timedwait, which ThreadSanitizer
			// understands (steady_clock -> pthread_cond_clockwait is not intercepted by gcc 11's
			// TSan and floods the report with false "double lock" findings)
			// The linger ends early once arrivals stop: callers come in bursts (the <= 3 parallel puts of a PutObject,
			// or everybody at once when a batch completes), and waiting out the full linger after the burst is pure
			// latency -- 3 callers: 0.80 -> 0.55 ms per put.
			const auto deadline = std::chrono::system_clock::now() + std::chrono::microseconds(max_wait_us);
			const auto gap = std::chrono::microseconds(std::max(20u, max_wait_us / 6));
			size_t seen = queue.size();
			while (!stop && queue.size() < max_blocks) {
				const auto now = std::chrono::system_clock::now();
				if (now >= deadline)
					break;
				if (cv_work.wait_until(lk, std::min(deadline, now + gap)) == std::cv_status::timeout && queue.size() == seen)
					break;  // nobody arrived during the gap
				seen = queue.size();
			}
			std::vector<Item *> batch;
			while (!queue.empty() && batch.size() < max_blocks) {
				batch.push_back(queue.front());
				queue.pop_front();
			}
			bool any_tag = false;
			for (Item *it : batch)
				any_tag = any_tag || it->has_tag;
			const uint64_t ticket = any_tag ? next_ticket++ : 0;  // taken in formation order, under the lock
			forming = false;
			cv_work.notify_all();
			lk.unlock();
			const size_t nb = batch.size();
			std::vector<uint8_t> hashes(nb * 32), pc(nb);
			std::vector<const uint8_t *> data(nb);
			std::vector<size_t> lens(nb);
			std::vector<gbm_order_tag> tags(nb);
			std::vector<int> rcs(nb, GBM_OK);
			static const uint8_t kEmpty = 0;  // a zero-length block may come with a NULL pointer
			for (size_t i = 0; i < nb; ++i) {
				std::memcpy(hashes.data() + 32 * i, batch[i]->hash, 32);
				data[i] = batch[i]->data ? batch[i]->data : &kEmpty;
				lens[i] = batch[i]->len;
				pc[i] = batch[i]->prevent_compression;
				// untagged blocks sort after tagged ones of the same batch; their relative order is free
				tags[i] = batch[i]->has_tag ? batch[i]->tag : gbm_order_tag{~0ull, i};
			}
			FanoutGate gate;
			bool passed = false;
			gate.before = [&] {
				std::unique_lock<std::mutex> g(mu);
				cv_turn.wait(g, [&] { return serving == ticket; });
			};
			gate.after = [&] {
				std::lock_guard<std::mutex> g(mu);
				++serving;
				passed = true;
				cv_turn.notify_all();
			};
			int rc;
			try {
				rc = put_blocks_impl(mg, nb, hashes.data(), data.data(), lens.data(), pc.data(), any_tag ? tags.data() : nullptr,
						     rcs.data(), any_tag ? &gate : nullptr);
			} catch (const std::exception &) {
				rc = GBM_E_IO;
				std::fill(rcs.begin(), rcs.end(), GBM_E_IO);
			}
			// per-block results are in rcs; put_blocks_impl marks every block on a whole-batch failure, and should it
			// ever return one without doing so, no caller of this batch is told its block was stored
			if (rc != GBM_OK && rc != GBM_E_QUORUM)
				for (int &r : rcs)
					if (r == GBM_OK)
						r = rc;
			lk.lock();
			if (any_tag && !passed) {  // the put failed before its fan-out: the turnstile must still move on
				cv_turn.wait(lk, [&] { return serving == ticket; });
				++serving;
				cv_turn.notify_all();
			}
			for (size_t i = 0; i < nb; ++i) {
				batch[i]->rc = rcs[i];
				batch[i]->done = true;
				ram_in_use_kb -= batch[i]->len / 1024;  // the permit is dropped once all sends finished
			}
			cv_ram.notify_all();
			++batches;
			blocks += nb;
			max_batch = std::max<uint64_t>(max_batch, nb);
			cv_done.notify_all();
		}
	}
};

extern "C" {

int gbm_batcher_create(gbm_manager *m, size_t max_blocks, unsigned max_wait_us, gbm_batcher **out)
{
	if (!m || !out || max_blocks == 0)
		return fail(GBM_E_INVALID_ARG, "bad batcher arguments");
	auto *b = new gbm_batcher();
	b->mg = m;
	b->max_blocks = max_blocks;
	b->max_wait_us = max_wait_us;
	const int nworkers = env().batcher_workers;
	for (int i = 0; i < nworkers; ++i)
		b->workers.emplace_back([b] { b->run(); });
	for (int i = 0; i < nworkers; ++i)
		b->gworkers.emplace_back([b] { b->run_gets(); });
	*out = b;
	return GBM_OK;
}

void gbm_batcher_destroy(gbm_batcher *b)
{
	if (!b)
		return;
	{
		std::lock_guard<std::mutex> g(b->mu);
		b->stop = true;
	}
	b->cv_work.notify_all();
	b->cv_ram.notify_all();
	{
		std::lock_guard<std::mutex> g(b->gmu);
		b->stop_gets = true;
	}
	b->gcv_work.notify_all();
	for (auto &t : b->workers)
		t.join();
	for (auto &t : b->gworkers)
		t.join();
	delete b;
}

// The asynchronous pair: submit queues the block and returns at once (it only waits for RAM permits), wait blocks until
// the batch that took the block has been fanned out.  This is the shape of `rpc_put_block(..)` as a future: a request
// creates its futures in block order and keeps <= 3 of them pending (put.rs:486-511), so the blocks of one stream enter
// the queue in `order` order -- which, with the fan-out turnstile, is what makes the OrderTag guarantee hold.
struct gbm_put_ticket {
	gbm_batcher *b;
	gbm_batcher::Item it;
};

int gbm_batcher_submit(gbm_batcher *b, const uint8_t hash[32], const uint8_t *data, size_t len, int prevent_compression,
		       const gbm_order_tag *order_tag, gbm_put_ticket **ticket_out)
{
	if (!b || !hash || (!data && len) || !ticket_out)
		return fail(GBM_E_INVALID_ARG, "NULL argument");
	*ticket_out = nullptr;
	std::unique_ptr<gbm_put_ticket> tk(new (std::nothrow) gbm_put_ticket());
	if (!tk)
		return fail(GBM_E_IO, "out of memory");
	tk->b = b;
	gbm_batcher::Item &it = tk->it;
	it.hash = hash;
	it.data = data;
	it.len = len;
	it.prevent_compression = prevent_compression ? 1 : 0;
	if (order_tag) {
		it.has_tag = true;
		it.tag = *order_tag;
	}
	std::unique_lock<std::mutex> lk(b->mu);
	// acquire len/1024 permits; a block larger than the whole budget could never be sent (Garage's
	// acquire_many would wait forever): refuse it instead
	const size_t need_kb = len / 1024;
	if (need_kb > b->ram_permits_kb)
		return fail(GBM_E_INVALID_ARG, "could not reserve space for buffer of data to send to remote nodes");
	b->cv_ram.wait(lk, [&] { return b->stop || b->ram_in_use_kb + need_kb <= b->ram_permits_kb; });
	if (b->stop)
		return fail(GBM_E_INVALID_ARG, "batcher is shutting down");
	b->ram_in_use_kb += need_kb;
	b->queue.push_back(&it);
	b->cv_work.notify_all();
	*ticket_out = tk.release();
	return GBM_OK;
}

int gbm_batcher_wait(gbm_put_ticket *ticket)
{
	if (!ticket)
		return fail(GBM_E_INVALID_ARG, "NULL ticket");
	std::unique_ptr<gbm_put_ticket> tk(ticket);
	int rc;
	{
		std::unique_lock<std::mutex> lk(tk->b->mu);
		tk->b->cv_done.wait(lk, [&] { return tk->it.done; });
		rc = tk->it.rc;
	}
	if (rc == GBM_E_QUORUM)
		return fail(rc, "Could not reach quorum");
	if (rc != GBM_OK)
		return fail(rc, "device batch failed");
	return GBM_OK;
}

int gbm_batcher_put_block(gbm_batcher *b, const uint8_t hash[32], const uint8_t *data, size_t len, int prevent_compression,
			  const gbm_order_tag *order_tag)
{
	gbm_put_ticket *tk = nullptr;
	int rc = gbm_batcher_submit(b, hash, data, len, prevent_compression, order_tag, &tk);
	return rc ? rc : gbm_batcher_wait(tk);
}

int gbm_batcher_get_block(gbm_batcher *b, const uint8_t hash[32], uint8_t *out, size_t cap, size_t *len_out)
{
	if (!b || !hash || (!out && cap) || !len_out)
		return fail(GBM_E_INVALID_ARG, "NULL argument");
	gbm_batcher::GetItem it;
	it.hash = hash;
	it.out = out;
	it.cap = cap;
	{
		std::unique_lock<std::mutex> lk(b->gmu);
		if (b->stop_gets)
			return fail(GBM_E_INVALID_ARG, "batcher is shutting down");
		b->gqueue.push_back(&it);
		b->gcv_work.notify_all();
		b->gcv_done.wait(lk, [&] { return it.done; });
	}
	*len_out = it.len;
	if (it.rc != GBM_OK)
		return one_block_rc(it.rc);
	return GBM_OK;
}

int gbm_batcher_get_stats(gbm_batcher *b, uint64_t out[3])
{
	if (!b || !out)
		return fail(GBM_E_INVALID_ARG, "NULL argument");
	std::lock_guard<std::mutex> g(b->gmu);
	out[0] = b->gbatches;
	out[1] = b->gblocks;
	out[2] = b->gmax_batch;
	return GBM_OK;
}

int gbm_batcher_set_ram_buffer_max(gbm_batcher *b, size_t bytes)
{
	if (!b || bytes < 1024)
		return fail(GBM_E_INVALID_ARG, "bad ram buffer size");
	std::lock_guard<std::mutex> g(b->mu);
	b->ram_permits_kb = bytes / 1024;
	b->cv_ram.notify_all();
	return GBM_OK;
}

int gbm_batcher_stats(gbm_batcher *b, uint64_t out[3])
{
	if (!b || !out)
		return fail(GBM_E_INVALID_ARG, "NULL argument");
	std::lock_guard<std::mutex> g(b->mu);
	out[0] = b->batches;
	out[1] = b->blocks;
	out[2] = b->max_batch;
	return GBM_OK;
}

}  // extern "C"
