// bm_batcher.cpp -- the coalescing queue in front of the FFI: many callers with <= 3 puts in flight each
// (PUT_BLOCKS_MAX_PARALLEL, src/api/s3/put.rs:42,486-511) -> a few device batches, with RAM permits
// (buffer_kb_semaphore, src/block/manager.rs:380-384); one such queue per device of a multi-device manager.
#include "bm_internal.hpp"

#ifdef __linux__
#include <sys/prctl.h>
#endif

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
//
// How a batch is cut (round 4).  A batch goes through three stages -- the copy into the shard buffers (host pool), the
// device trip (link-bound), the fan-out (host pool) -- and with two workers the stages of consecutive batches can
// overlap, IF the callers are not all in the same batch: a closed loop of 48 callers that land in one batch wait for
// it together, come back together and form the next one together, and the second worker never has anything to do
// (round 3: 960 puts in 22 batches, 21.6 GiB/s where a bulk put of the same blocks reaches 43).  So
//   - a worker that forms a batch while other workers are idle takes only its share of what is queued
//     (GBM_BATCHER_SPLIT_MIN blocks or more: a PutObject's three still go as one batch), and the next idle worker
//     takes the rest at once;
//   - one batch per device is on the link at a time (`dev_mu`): two trips that share the link only lengthen each
//     other and finish together, which is how the callers get back into lock step.
// The same holds for the read side (gather / device trip / assembly).
//
// Several devices (gbm_create_multi): the batcher the caller holds is a front with one complete queue -- workers, RAM
// budget share, statistics -- per device; a block goes to the queue of gec_device_of_hash(hash), and no lock is
// shared between the queues.  The exception is by necessity: an OrderTag stream's blocks are encoded on different
// devices, so tagged blocks take a sequence number from the front when they are submitted and reach the nodes one
// block at a time behind their stream's previous block (`StreamSeq`).
struct gbm_batcher {
	struct Item {
		const uint8_t *hash, *data;
		size_t len;
		uint8_t prevent_compression = 0;
		bool has_tag = false;
		gbm_order_tag tag{0, 0};
		uint64_t seq = 0, prev_seq = 0;  // multi-device fronts: submission sequence, and the stream's previous block's
		bool delivered = false;          // (its fan-out is over: the stream's next block may go)
		int rc = GBM_OK;
		std::string err;  // the worker's error text (thread-local there), re-published on the caller's thread
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
	int busy = 0;  // workers with a batch in hand
	static constexpr size_t kCrowdUnknown = ~(size_t)0;
	size_t last_size = 0, glast_size = 0;  // blocks of the batch formed last with nothing in flight (put side / get side)
	uint64_t batches = 0, blocks = 0, max_batch = 0;
	// fan-out turnstile of the tagged batches
	uint64_t next_ticket = 0, serving = 0;
	std::condition_variable cv_turn;
	std::mutex dev_mu, gdev_mu;  // one put batch / one get batch of this device on the link at a time
	// two workers: while one batch is on the device the next one forms and starts (the device trip has a latency
	// floor -- the checksum chain -- that a single worker would pay serially)
	std::vector<std::thread> workers;

	// ---- several devices
	std::vector<gbm_batcher *> lanes;  // non-empty: this is a front, it only routes
	gbm_batcher *front = nullptr;      // a lane's front
	// Tagged blocks of a multi-device front.  A block's `seq` is taken when it is queued (under its lane's lock, so a
	// lane's queue is in seq order) and a block waits for exactly one thing: its stream's previous block.  The block with
	// the smallest undelivered seq is always first in its batch's order, its batch is the oldest of its lane, and its
	// predecessor has a smaller seq -- so it can always go: no cycle.
	struct StreamSeq {
		std::mutex mu;
		std::condition_variable cv;
		uint64_t next = 1;
		struct St {
			uint64_t last_submitted = 0, delivered = 0;
		};
		std::unordered_map<uint64_t, St> streams;
		void submit(Item &it)
		{
			std::lock_guard<std::mutex> g(mu);
			St &st = streams[it.tag.stream_id];
			it.seq = next++;
			it.prev_seq = st.last_submitted;
			st.last_submitted = it.seq;
		}
		void wait_turn(const Item &it)
		{
			std::unique_lock<std::mutex> g(mu);
			cv.wait(g, [&] {
				auto f = streams.find(it.tag.stream_id);
				return f == streams.end() || f->second.delivered >= it.prev_seq;
			});
		}
		void deliver(Item &it)
		{
			{
				std::lock_guard<std::mutex> g(mu);
				auto f = streams.find(it.tag.stream_id);
				if (f != streams.end()) {
					f->second.delivered = std::max(f->second.delivered, it.seq);
					if (f->second.last_submitted == it.seq)
						streams.erase(f);  // nothing of this stream is pending any more
				}
				it.delivered = true;
			}
			cv.notify_all();
		}
	} sseq;
	bool sequenced() const { return front && front->lanes.size() > 1; }

	// ---- the read side: GetObject's readers (a few blocks ahead each, src/api/s3/get.rs:429) coalesced the same way.
	// Sixteen readers fetching eight blocks each through gbm_rpc_get_blocks make sixteen device trips that queue up
	// behind one another and behind the host pool (4.3 GiB/s, 26 ms per get); through here they make a few.
	struct GetItem {
		const uint8_t *hash;
		uint8_t *out;
		size_t cap, len = 0;
		int rc = GBM_OK;
		std::string err;
		bool done = false;
		bool needs_hash = false;  // delivered, but its end-to-end hash is the waiting caller's to do (FanoutGate::defer_block_hash)
	};
	std::mutex gmu;
	std::condition_variable gcv_work, gcv_done;
	std::deque<GetItem *> gqueue;
	bool gforming = false, stop_gets = false;
	int gbusy = 0;
	uint64_t gbatches = 0, gblocks = 0, gmax_batch = 0;
	std::vector<std::thread> gworkers;

	// Forms one batch out of `q` (the caller holds `lk`, `forming_flag` is this side's): lingers a little so concurrent
	// callers land in the same batch -- the linger ends early once arrivals stop: callers come in bursts (the <= 3
	// parallel puts of a PutObject, or everybody at once when a batch completes), and waiting out the full linger after
	// the burst is pure latency (3 callers: 0.80 -> 0.55 ms per put) -- then takes its share of what is queued.
	// (system_clock: libstdc++ maps it to pthread_cond_timedwait, which ThreadSanitizer understands; steady_clock ->
	// pthread_cond_clockwait is not intercepted by gcc 11's TSan and floods the report with false "double lock" findings)
	template <class T>
	std::vector<T *> form(std::unique_lock<std::mutex> &lk, std::condition_variable &cv, std::deque<T *> &q, const bool &stopping,
			      int busy_now, size_t nworkers, size_t &last_size, size_t split_min)
	{
		// Callers in a closed loop are not made to wait for company that never comes: when the previous batch was formed with
		// nothing in flight it held everybody there is (one block of a lone caller, the three of a PutObject), so with nothing
		// in flight again the batch goes the moment as many blocks are queued (a single put: 0.20 -> 0.17 ms; three: no 30 us
		// gap behind the third).  As soon as trips overlap -- the batch is formed while another is in flight -- the size of
		// the crowd is unknown and the linger is back.
		const size_t crowd = busy_now == 0 && last_size != kCrowdUnknown ? std::max<size_t>(last_size, 1) : 0;
		const auto deadline = std::chrono::system_clock::now() + std::chrono::microseconds(max_wait_us);
		const unsigned gap_us = std::min(100u, std::max(20u, max_wait_us / 10));  // (a tenth of the linger, within 20..100 us)
		const auto gap = std::chrono::microseconds(gap_us);
		size_t seen = q.size();
		while (!(crowd && q.size() >= crowd) && !stopping && q.size() < max_blocks) {
			const auto now = std::chrono::system_clock::now();
			if (now >= deadline)
				break;
			if (cv.wait_until(lk, std::min(deadline, now + gap)) == std::cv_status::timeout && q.size() == seen)
				break;  // nobody arrived during the gap
			seen = q.size();
		}
		size_t take = std::min(q.size(), max_blocks);
		const size_t idle = nworkers > (size_t)busy_now ? nworkers - (size_t)busy_now : 1;  // this worker included
		if (split_min && idle > 1 && q.size() >= split_min)
			take = std::min(take, (q.size() + idle - 1) / idle);
		else if (split_min && q.size() >= 2 * split_min)
			// the others are busy: half of a long queue stays for the worker that frees up next -- taking it all is how a
			// closed loop of callers falls back into one batch (one straggler's tiny batch in flight was enough)
			take = std::min(take, (q.size() + 1) / 2);
		std::vector<T *> batch;
		batch.reserve(take);
		while (batch.size() < take) {
			batch.push_back(q.front());
			q.pop_front();
		}
		last_size = busy_now == 0 ? batch.size() : kCrowdUnknown;
		return batch;
	}

	// A batch's turn on its device's link: taken right before the codec call, given back when the call's bulk transfers are
	// over (gec_thread_link_release: the codec calls back on this thread) -- the next batch's upload then runs beside this
	// one's last checksum kernels -- or, at the latest, when the call has returned.
	struct LinkTurn {
		std::mutex *mu;
		bool held = false;
		static void release(void *p)
		{
			LinkTurn *t = static_cast<LinkTurn *>(p);
			if (t->held) {
				t->held = false;
				t->mu->unlock();
			}
		}
		void enter()
		{
			mu->lock();
			held = true;
		}
		void exit() { release(this); }
	};

	// The linger is a timed wait of a few tens of microseconds; a thread's default timer slack (50 us) would double it.
	static void precise_timers()
	{
#ifdef __linux__
		(void)prctl(PR_SET_TIMERSLACK, 1000UL, 0UL, 0UL, 0UL);
#endif
	}

	void run_gets()
	{
		precise_timers();
		std::unique_lock<std::mutex> lk(gmu);
		for (;;) {
			gcv_work.wait(lk, [&] { return stop_gets || (!gqueue.empty() && !gforming); });
			if (gqueue.empty()) {
				if (stop_gets)
					return;
				continue;
			}
			gforming = true;
			std::vector<GetItem *> batch = form(lk, gcv_work, gqueue, stop_gets, gbusy, gworkers.size(), glast_size, env().batcher_get_split_min);
			gforming = false;
			++gbusy;
			gcv_work.notify_all();
			lk.unlock();
			const size_t nb = batch.size();
			std::vector<int> rcs;
			std::vector<size_t> lens;
			std::vector<std::string> errs(nb);
			std::vector<uint8_t> deferred;
			FanoutGate gate;
			LinkTurn turn{&gdev_mu};
			gate.device_enter = [&] { turn.enter(); };
			gate.device_exit = [&] { turn.exit(); };
			try {
				deferred.assign(nb, 0);
				gate.defer_block_hash = &deferred;
				std::vector<uint8_t> hashes(nb * 32);
				std::vector<uint8_t *> outs(nb);
				std::vector<size_t> caps(nb);
				lens.assign(nb, 0);
				rcs.assign(nb, GBM_E_MISSING_BLOCK);
				for (size_t i = 0; i < nb; ++i) {
					std::memcpy(hashes.data() + 32 * i, batch[i]->hash, 32);
					outs[i] = batch[i]->out;
					caps[i] = batch[i]->cap;
				}
				auto fetch = [&](size_t i0, size_t cnt) {
					int rc;
					try {
						// (the deferral flags are indexed by the call's blocks: a sub-call gets its own window of them)
						std::vector<uint8_t> win(cnt, 0);
						FanoutGate g2 = gate;
						g2.defer_block_hash = &win;
						rc = get_blocks_impl(mg, cnt, hashes.data() + 32 * i0, nullptr, outs.data() + i0, caps.data() + i0,
								     lens.data() + i0, rcs.data() + i0, false, nullptr, &g2);
						std::copy(win.begin(), win.end(), deferred.begin() + i0);
					} catch (const std::exception &e) {
						rc = fail(GBM_E_IO, std::string("batched get: ") + e.what());
					}
					if (rc != GBM_OK)  // the whole call failed (a device error, out of memory)
						for (size_t i = i0; i < i0 + cnt; ++i) {
							rcs[i] = rc;
							errs[i] = last_error();
						}
					return rc;
				};
				// a call that fails as a whole (out of pinned memory, a device error) fails every block it carries: the
				// blocks are then fetched one by one, so that one reader's trouble is not every reader's
				if (fetch(0, nb) != GBM_OK && nb > 1)
					for (size_t i = 0; i < nb; ++i)
						(void)fetch(i, 1);
			} catch (const std::exception &e) {  // the vectors above
				rcs.assign(nb, GBM_E_IO);
				lens.assign(nb, 0);
				deferred.clear();
				for (auto &s : errs)
					s = e.what();
			}
			lk.lock();
			for (size_t i = 0; i < nb; ++i) {
				batch[i]->rc = rcs[i];
				batch[i]->len = lens[i];
				batch[i]->err = std::move(errs[i]);
				batch[i]->needs_hash = rcs[i] == GBM_OK && i < deferred.size() && deferred[i];
				batch[i]->done = true;
			}
			--gbusy;
			++gbatches;
			gblocks += nb;
			gmax_batch = std::max<uint64_t>(gmax_batch, nb);
			gcv_done.notify_all();
		}
	}

	void run()
	{
		precise_timers();
		std::unique_lock<std::mutex> lk(mu);
		for (;;) {
			// one worker forms a batch at a time; the others are on the device, or wait their turn
			cv_work.wait(lk, [&] { return stop || (!queue.empty() && !forming); });
			if (queue.empty()) {
				if (stop)
					return;
				continue;
			}
			forming = true;
			std::vector<Item *> batch = form(lk, cv_work, queue, stop, busy, workers.size(), last_size, env().batcher_split_min);
			bool any_tag = false;
			for (Item *it : batch)
				any_tag = any_tag || it->has_tag;
			const bool seq = any_tag && sequenced();
			// (single device) taken in formation order, under the lock
			const uint64_t ticket = any_tag && !seq ? next_ticket++ : 0;
			forming = false;
			++busy;
			cv_work.notify_all();
			lk.unlock();
			const size_t nb = batch.size();
			std::vector<int> rcs;
			std::string err;
			bool passed = false;
			try {
				std::vector<uint8_t> hashes(nb * 32), pc(nb);
				std::vector<const uint8_t *> data(nb);
				std::vector<size_t> lens(nb), order;
				std::vector<gbm_order_tag> tags(nb);
				rcs.assign(nb, GBM_OK);
				static const uint8_t kEmpty = 0;  // a zero-length block may come with a NULL pointer
				for (size_t i = 0; i < nb; ++i) {
					std::memcpy(hashes.data() + 32 * i, batch[i]->hash, 32);
					data[i] = batch[i]->data ? batch[i]->data : &kEmpty;
					lens[i] = batch[i]->len;
					pc[i] = batch[i]->prevent_compression;
					// untagged blocks sort after tagged ones of the same batch; their relative order is free
					tags[i] = batch[i]->has_tag ? batch[i]->tag : gbm_order_tag{GBM_NO_STREAM, i};
				}
				FanoutGate gate;
				LinkTurn turn{&dev_mu};
				gate.device_enter = [&] { turn.enter(); };
				gate.device_exit = [&] { turn.exit(); };
				if (seq) {
					// tagged blocks in submission order (the queue's), untagged ones behind them
					for (size_t i = 0; i < nb; ++i)
						if (batch[i]->has_tag)
							order.push_back(i);
					for (size_t i = 0; i < nb; ++i)
						if (!batch[i]->has_tag)
							order.push_back(i);
					gate.block_order = &order;
					gate.before_block = [&](size_t b) {
						if (batch[b]->has_tag)
							front->sseq.wait_turn(*batch[b]);
					};
					gate.after_block = [&](size_t b) {
						if (batch[b]->has_tag)
							front->sseq.deliver(*batch[b]);
					};
				} else if (any_tag) {
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
				}
				int rc;
				try {
					rc = put_blocks_impl(mg, nb, hashes.data(), data.data(), lens.data(), pc.data(), any_tag ? tags.data() : nullptr,
							     rcs.data(), &gate);
				} catch (const std::exception &e) {
					rc = fail(GBM_E_IO, std::string("batched put: ") + e.what());
					std::fill(rcs.begin(), rcs.end(), GBM_E_IO);
				}
				if (rc != GBM_OK)
					err = last_error();
				// per-block results are in rcs; put_blocks_impl marks every block on a whole-batch failure, and should it
				// ever return one without doing so, no caller of this batch is told its block was stored
				if (rc != GBM_OK && rc != GBM_E_QUORUM)
					for (int &r : rcs)
						if (r == GBM_OK)
							r = rc;
			} catch (const std::exception &e) {  // the vectors above
				rcs.assign(nb, GBM_E_IO);
				err = e.what();
			}
			if (seq)  // a put that failed before (or during) its fan-out: the streams must still move on
				for (Item *it : batch)
					if (it->has_tag && !it->delivered) {
						front->sseq.wait_turn(*it);
						front->sseq.deliver(*it);
					}
			lk.lock();
			if (any_tag && !seq && !passed) {  // the put failed before its fan-out: the turnstile must still move on
				cv_turn.wait(lk, [&] { return serving == ticket; });
				++serving;
				cv_turn.notify_all();
			}
			for (size_t i = 0; i < nb; ++i) {
				batch[i]->rc = rcs[i];
				if (rcs[i] != GBM_OK)
					batch[i]->err = err;
				batch[i]->done = true;
				ram_in_use_kb -= batch[i]->len / 1024;  // the permit is dropped once all sends finished
			}
			--busy;
			cv_ram.notify_all();
			++batches;
			blocks += nb;
			max_batch = std::max<uint64_t>(max_batch, nb);
			cv_done.notify_all();
		}
	}
};

namespace {

gbm_batcher *make_lane(gbm_manager *m, size_t max_blocks, unsigned max_wait_us, gbm_batcher *front)
{
	auto *b = new gbm_batcher();
	b->mg = m;
	b->front = front;
	b->max_blocks = max_blocks;
	b->max_wait_us = max_wait_us;
	const int nworkers = env().batcher_workers;
	for (int i = 0; i < nworkers; ++i)
		b->workers.emplace_back([b] {
			lane_thread("gbm-batch-put", b->mg->is_front() ? nullptr : b->mg->codec);
			b->run();
		});
	for (int i = 0; i < nworkers; ++i)
		b->gworkers.emplace_back([b] {
			lane_thread("gbm-batch-get", b->mg->is_front() ? nullptr : b->mg->codec);
			b->run_gets();
		});
	return b;
}

void destroy_lane(gbm_batcher *b)
{
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

// the queue that serves `hash`
gbm_batcher *lane_for(gbm_batcher *b, const uint8_t *hash)
{
	return b->lanes.empty() ? b : b->lanes[(size_t)gec_device_of_hash(hash, (int)b->lanes.size())];
}

}  // namespace

void gbmimpl::batcher_snapshot(gbm_batcher *b, uint64_t out[5])
{
	std::fill(out, out + 5, 0);
	auto add = [&](gbm_batcher *q) {
		{
			std::lock_guard<std::mutex> g(q->mu);
			out[0] += q->ram_permits_kb > q->ram_in_use_kb ? q->ram_permits_kb - q->ram_in_use_kb : 0;
			out[1] += q->batches;
			out[2] += q->blocks;
		}
		std::lock_guard<std::mutex> g(q->gmu);
		out[3] += q->gbatches;
		out[4] += q->gblocks;
	};
	if (b->lanes.empty())
		add(b);
	for (gbm_batcher *l : b->lanes)
		add(l);
}

extern "C" {

int gbm_batcher_create(gbm_manager *m, size_t max_blocks, unsigned max_wait_us, gbm_batcher **out)
{
	if (!m || !out || max_blocks == 0)
		return fail(GBM_E_INVALID_ARG, "bad batcher arguments");
	*out = nullptr;
	try {
		if (!m->is_front()) {
			*out = make_lane(m, max_blocks, max_wait_us, nullptr);
			return GBM_OK;
		}
		// one queue, its workers and its share of the RAM budget per device
		auto *f = new gbm_batcher();
		f->mg = m;
		f->max_blocks = max_blocks;
		f->max_wait_us = max_wait_us;
		for (auto &lane : m->lanes) {
			f->lanes.push_back(make_lane(lane.get(), max_blocks, max_wait_us, f));
			f->lanes.back()->ram_permits_kb = std::max<size_t>(1, f->ram_permits_kb / m->lanes.size());
		}
		*out = f;
	} catch (const std::exception &e) {
		return fail(GBM_E_IO, std::string("gbm_batcher_create: ") + e.what());
	}
	return GBM_OK;
}

void gbm_batcher_destroy(gbm_batcher *b)
{
	if (!b)
		return;
	if (b->lanes.empty()) {
		destroy_lane(b);
		return;
	}
	for (gbm_batcher *l : b->lanes)
		destroy_lane(l);
	delete b;
}

// The asynchronous pair: submit queues the block and returns at once (it only waits for RAM permits), wait blocks until
// the batch that took the block has been fanned out.  This is the shape of `rpc_put_block(..)` as a future: a request
// creates its futures in block order and keeps <= 3 of them pending (put.rs:486-511), so the blocks of one stream enter
// the queue in `order` order -- which, with the fan-out turnstile, is what makes the OrderTag guarantee hold.
struct gbm_put_ticket {
	gbm_batcher *b;  // the queue that holds the item
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
	b = lane_for(b, hash);
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
	if (it.has_tag && b->sequenced())
		b->front->sseq.submit(it);  // under the queue's lock: the queue is in sequence order
	try {
		b->queue.push_back(&it);
	} catch (const std::bad_alloc &) {
		b->ram_in_use_kb -= need_kb;
		if (it.has_tag && b->sequenced())
			b->front->sseq.deliver(it);  // (waits for nobody: only marks the sequence number as gone)
		return fail(GBM_E_IO, "out of memory");
	}
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
		return fail(rc, tk->it.err.empty() ? "Could not reach quorum" : tk->it.err);
	if (rc != GBM_OK)
		return fail(rc, tk->it.err.empty() ? "device batch failed" : tk->it.err);
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
	b = lane_for(b, hash);
	gbm_batcher::GetItem it;
	it.hash = hash;
	it.out = out;
	it.cap = cap;
	{
		std::unique_lock<std::mutex> lk(b->gmu);
		if (b->stop_gets)
			return fail(GBM_E_INVALID_ARG, "batcher is shutting down");
		try {
			b->gqueue.push_back(&it);
		} catch (const std::bad_alloc &) {
			return fail(GBM_E_IO, "out of memory");
		}
		b->gcv_work.notify_all();
		b->gcv_done.wait(lk, [&] { return it.done; });
	}
	*len_out = it.len;
	if (it.rc == GBM_OK && it.needs_hash) {
		// DataBlock::verify (block.rs:69-77) by the reader itself: the batch delivered the block on its shards' checksums, the
		// name is checked here, on this caller's core, beside the other readers' (GBM_VERIFY_ALWAYS through the queue)
		uint8_t sum[32];
		blake2sum(out, it.len, sum);
		if (std::memcmp(sum, hash, 32) != 0)
			return one_block_rc(GBM_E_CORRUPT_DATA);
	}
	if (it.rc == GBM_OK)
		return GBM_OK;
	if (it.rc == GBM_E_MISSING_BLOCK || it.rc == GBM_E_CORRUPT_DATA || it.rc == GBM_E_BUFFER_TOO_SMALL)
		return one_block_rc(it.rc);
	return fail(it.rc, it.err.empty() ? "batched get failed" : it.err);  // the worker's text, on the caller's thread
}

static void sum_stats(gbm_batcher *b, bool gets, uint64_t out[3])
{
	out[0] = out[1] = out[2] = 0;
	auto add = [&](gbm_batcher *q) {
		std::lock_guard<std::mutex> g(gets ? q->gmu : q->mu);
		out[0] += gets ? q->gbatches : q->batches;
		out[1] += gets ? q->gblocks : q->blocks;
		out[2] = std::max(out[2], gets ? q->gmax_batch : q->max_batch);
	};
	if (b->lanes.empty())
		add(b);
	for (gbm_batcher *l : b->lanes)
		add(l);
}

int gbm_batcher_get_stats(gbm_batcher *b, uint64_t out[3])
{
	if (!b || !out)
		return fail(GBM_E_INVALID_ARG, "NULL argument");
	sum_stats(b, true, out);
	return GBM_OK;
}

int gbm_batcher_set_ram_buffer_max(gbm_batcher *b, size_t bytes)
{
	if (!b || bytes < 1024)
		return fail(GBM_E_INVALID_ARG, "bad ram buffer size");
	auto set = [](gbm_batcher *q, size_t kb) {
		std::lock_guard<std::mutex> g(q->mu);
		q->ram_permits_kb = std::max<size_t>(1, kb);
		q->cv_ram.notify_all();
	};
	set(b, bytes / 1024);
	for (gbm_batcher *l : b->lanes)  // the node's budget, shared out evenly: the queues share no lock
		set(l, bytes / 1024 / b->lanes.size());
	return GBM_OK;
}

int gbm_batcher_stats(gbm_batcher *b, uint64_t out[3])
{
	if (!b || !out)
		return fail(GBM_E_INVALID_ARG, "NULL argument");
	sum_stats(b, false, out);
	return GBM_OK;
}

int gbm_batcher_device_stats(gbm_batcher *b, int dev, uint64_t put_out[3], uint64_t get_out[3])
{
	if (!b || dev < 0 || dev >= (b->lanes.empty() ? 1 : (int)b->lanes.size()))
		return fail(GBM_E_INVALID_ARG, "bad device index");
	gbm_batcher *q = b->lanes.empty() ? b : b->lanes[(size_t)dev];
	if (put_out)
		sum_stats(q, false, put_out);
	if (get_out)
		sum_stats(q, true, get_out);
	return GBM_OK;
}

}  // extern "C"
