// bm_scrub_worker.cpp -- the ScrubWorker as the reference runs it (src/block/repair.rs:156-500): a continuously running task with
// a state machine (Finished / Running / Paused), four commands, a schedule (every 25 + 0..10 days), a persisted record with a
// checkpoint of its iterator, and a restart that carries on from it.  One step = one batch on the device: the steps themselves
// (read_scrub_batch / verify_scrub_batch) are bm_scrub.cpp's, shared with the blocking gbm_scrub_all.
#include "bm_internal.hpp"

#include <cstdio>
#include <random>

using namespace gbmimpl;

// ------------------------------------------------------------------ the continuously running ScrubWorker
// src/block/repair.rs:156-500.  State machine, commands, schedule and persisted record as the reference has them; the
// step is a batch on the device instead of one read_block.
namespace gbmimpl {

// BlockStoreIterator (repair.rs:196-233,634-752), reduced to what a hash-ordered walk needs: first-level directories
// (first hash byte) below `prefix` are done, and within `prefix` every hash <= `after`.  prefix == 256: the end.
struct ScrubCursor {
	int prefix = 0;
	Hash after;
	bool same(const ScrubCursor &o) const { return prefix == o.prefix && after == o.after; }
	// progress by the first two hash bytes -- the reference's directories (iterator.progress(), :664-674)
	double progress() const
	{
		if (prefix >= 256)
			return 1.0;
		return (prefix * 256.0 + (after.empty() ? 0.0 : (double)(unsigned char)after[1])) / 65536.0;
	}
};

struct ScrubWorker {
	gbm_manager *mg;
	std::string path;  // "" = not persisted
	size_t batch_blocks;
	uint64_t cp_interval_ms;
	std::mutex mu;
	std::condition_variable cv;
	std::thread th;
	bool stop = false;
	// ScrubWorkerPersisted (:185-194); the tranquility lives in mg->scrub_tranquility
	uint64_t t_last_complete = 0, t_next_run = 0, corruptions = 0;
	bool has_cp = false;
	ScrubCursor cp;
	// ScrubWorkerState (:272-286)
	int state = GBM_SCRUB_FINISHED;
	ScrubCursor it;  // Running / Paused: everything up to here is done
	uint64_t t_cp = 0, t_resume = 0;
	uint64_t generation = 0;  // moves with every command that drops the step in flight
	uint64_t blocks = 0, cps = 0, errors = 0;

	// the listing of a few consecutive first-level directories over all nodes (the worker thread's own)
	struct Listing {
		int lo = -1, hi = -1;  // directories lo <= h0 < hi
		std::vector<Hash> hashes;
		bool holds(int prefix) const { return lo <= prefix && prefix < hi; }
		void drop() { lo = hi = -1; }
	} listing;
	int width = 1;  // directories per listing: doubled while a listing holds less than a step, halved when it holds several

	// randomize_next_scrub_run_time (:245-256): SCRUB_INTERVAL plus a random 0..10 days, "to balance scrub load across
	// different cluster nodes"
	static uint64_t randomize_next_run(uint64_t ts)
	{
		static std::mutex rmu;
		static std::mt19937_64 rng{std::random_device{}()};
		std::lock_guard<std::mutex> g(rmu);
		return ts + GBM_SCRUB_INTERVAL_MS + (rng() % (3600ull * 24 * 10)) * 1000;
	}

	// ---- the state file
	static void put64(std::vector<uint8_t> &b, uint64_t v)
	{
		for (int i = 0; i < 8; ++i)
			b.push_back((uint8_t)(v >> (8 * i)));
	}
	static uint64_t get64(const uint8_t *p)
	{
		uint64_t v = 0;
		for (int i = 0; i < 8; ++i)
			v |= (uint64_t)p[i] << (8 * i);
		return v;
	}
	static constexpr size_t kRecord = 8 + 4 + 3 * 8 + 1 + 2 + 1 + 32;
	void save()  // (mu held)
	{
		if (path.empty())
			return;
		std::vector<uint8_t> b;
		b.insert(b.end(), {'G', 'B', 'M', 's', 'c', 'r', 'b', '1'});
		const uint32_t tq = mg->scrub_tranquility.load();
		for (int i = 0; i < 4; ++i)
			b.push_back((uint8_t)(tq >> (8 * i)));
		put64(b, t_last_complete);
		put64(b, t_next_run);
		put64(b, corruptions);
		b.push_back(has_cp ? 1 : 0);
		b.push_back((uint8_t)(cp.prefix & 0xff));
		b.push_back((uint8_t)(cp.prefix >> 8));
		b.push_back((uint8_t)cp.after.size());
		uint8_t h[32] = {};
		std::memcpy(h, cp.after.data(), std::min<size_t>(32, cp.after.size()));
		b.insert(b.end(), h, h + 32);
		const std::string tmp = path + ".tmp";
		bool ok = false;
		if (FILE *f = std::fopen(tmp.c_str(), "wb")) {
			ok = std::fwrite(b.data(), 1, b.size(), f) == b.size();
			ok = (std::fclose(f) == 0) && ok;
		}
		if (ok && std::rename(tmp.c_str(), path.c_str()) == 0) {
			if (has_cp)
				++cps;
		} else {
			std::remove(tmp.c_str());
			std::fprintf(stderr, "garage_block: could not save scrub checkpoint to %s\n", path.c_str());  // repair.rs:343-345
		}
	}
	// false: no file, or one that does not decode (PersisterShared::new falls back to Default, persister.rs:97-101)
	bool load()
	{
		if (path.empty())
			return false;
		uint8_t b[kRecord + 1];
		size_t n = 0;
		if (FILE *f = std::fopen(path.c_str(), "rb")) {
			n = std::fread(b, 1, sizeof(b), f);
			std::fclose(f);
		}
		if (n != kRecord || std::memcmp(b, "GBMscrb1", 8) != 0)
			return false;
		const uint32_t tq = (uint32_t)b[8] | (uint32_t)b[9] << 8 | (uint32_t)b[10] << 16 | (uint32_t)b[11] << 24;
		const int prefix = b[37] | b[38] << 8;
		const size_t alen = b[39];
		if (b[36] > 1 || prefix > 256 || (alen != 0 && alen != 32))
			return false;
		mg->scrub_tranquility = tq;
		t_last_complete = get64(b + 12);
		t_next_run = get64(b + 20);
		corruptions = get64(b + 28);
		has_cp = b[36] == 1;
		cp.prefix = prefix;
		cp.after.assign((const char *)b + 40, alen);
		return true;
	}

	// ---- the iterator
	void list_from(int prefix)
	{
		const int lo = prefix, hi = std::min(256, prefix + width);
		std::vector<std::set<Hash>> per(mg->nodes.size());
		mg->pool->parallel_for(mg->nodes.size(), [&](size_t i) {
			if (!mg->nodes[i]->down.load())
				mg->nodes[i]->list_prefix_range(lo, hi, per[i]);
		});
		std::set<Hash> all;
		for (auto &st : per)
			for (const Hash &h : st)
				if (mg->owns(h))
					all.insert(h);
		listing.lo = lo;
		listing.hi = hi;
		listing.hashes.assign(all.begin(), all.end());
		if (listing.hashes.size() < batch_blocks)
			width = std::min(256, width * 2);
		else if (listing.hashes.size() > 4 * batch_blocks)
			width = std::max(1, width / 2);
	}
	// the next (at most) n hashes behind c, and where the walk stands once they are done
	std::vector<Hash> take(ScrubCursor &c, size_t n)
	{
		std::vector<Hash> out;
		while (out.size() < n && c.prefix < 256) {
			if (!listing.holds(c.prefix))
				list_from(c.prefix);
			const auto &L = listing.hashes;
			// behind `after`, or at the first hash of directory `prefix` (a one-byte string sorts in front of them all)
			auto first = c.after.empty() ? std::lower_bound(L.begin(), L.end(), Hash(1, (char)c.prefix)) : std::upper_bound(L.begin(), L.end(), c.after);
			const size_t avail = (size_t)(L.end() - first), t = std::min(n - out.size(), avail);
			out.insert(out.end(), first, first + (ptrdiff_t)t);
			if (t == avail) {  // these directories are done
				c.prefix = listing.hi;
				c.after.clear();
			} else {
				c.after = out.back();
				c.prefix = (unsigned char)c.after[0];
			}
		}
		return out;
	}

	// ---- commands (ScrubWorker::handle_cmd, :328-400); mu held
	int command(int cmd, uint64_t pause_ms)
	{
		const uint64_t now = mg->now();
		switch (cmd) {
		case GBM_SCRUB_CMD_START:
			if (state != GBM_SCRUB_FINISHED)
				return fail(GBM_E_INVALID_ARG, "Cannot start scrub worker: already running!");
			it = ScrubCursor();
			cp = it;
			has_cp = true;
			save();
			state = GBM_SCRUB_RUNNING;
			t_cp = now;
			break;
		case GBM_SCRUB_CMD_PAUSE:
			if (state == GBM_SCRUB_FINISHED)
				return fail(GBM_E_INVALID_ARG, "Cannot pause scrub worker: not running!");
			cp = it;
			has_cp = true;
			save();
			state = GBM_SCRUB_PAUSED;
			t_resume = now + pause_ms;
			break;
		case GBM_SCRUB_CMD_RESUME:
			if (state != GBM_SCRUB_PAUSED)
				return fail(GBM_E_INVALID_ARG, "Cannot resume scrub worker: not paused!");
			state = GBM_SCRUB_RUNNING;
			t_cp = now;
			break;
		case GBM_SCRUB_CMD_CANCEL:
			if (state == GBM_SCRUB_FINISHED)
				return fail(GBM_E_INVALID_ARG, "Cannot cancel scrub worker: not running!");
			has_cp = false;
			save();
			state = GBM_SCRUB_FINISHED;
			break;
		default:
			return fail(GBM_E_INVALID_ARG, "unknown scrub worker command");
		}
		++generation;  // the step in flight (if any) belongs to the state before the command
		cv.notify_all();
		return GBM_OK;
	}

	void wait_ms(std::unique_lock<std::mutex> &lk, uint64_t ms)
	{
		cv.wait_until(lk, std::chrono::system_clock::now() + std::chrono::milliseconds(std::max<uint64_t>(1, std::min<uint64_t>(ms, 10000))));
	}

	// Worker::work / wait_for_work (:402-520)
	void run()
	{
		struct Ahead {  // the following step's shards, read from the nodes while this step is on the device
			bool valid = false;
			uint64_t gen = 0;
			ScrubCursor from, to;
			std::future<ScrubBatch> fut;
		} ahead;
		auto drop_ahead = [&] {
			if (ahead.valid)
				ahead.fut.wait();
			ahead.valid = false;
		};
		std::unique_lock<std::mutex> lk(mu);
		while (!stop) {
			const uint64_t now = mg->now();
			if (state == GBM_SCRUB_FINISHED) {
				if (now >= t_next_run)
					(void)command(GBM_SCRUB_CMD_START, 0);
				else
					wait_ms(lk, t_next_run - now);
				continue;
			}
			if (state == GBM_SCRUB_PAUSED) {
				if (now >= t_resume)
					(void)command(GBM_SCRUB_CMD_RESUME, 0);
				else
					wait_ms(lk, t_resume - now);
				continue;
			}
			// ---- Running: one step
			const uint64_t gen = generation;
			const ScrubCursor from = it;
			lk.unlock();
			ScrubBatch cur;
			ScrubCursor to = from;
			uint64_t st[4] = {0, 0, 0, 0};
			int rc = GBM_OK;
			std::chrono::nanoseconds device_time{0};
			try {
				if (ahead.valid && ahead.gen == gen && ahead.from.same(from)) {
					cur = ahead.fut.get();
					to = ahead.to;
					ahead.valid = false;
				} else {
					drop_ahead();
					listing.drop();  // the first step after a command or a restart: the directories are listed afresh
					cur = read_scrub_batch(mg, take(to, batch_blocks));
				}
				if (to.prefix < 256) {
					ScrubCursor to2 = to;
					std::vector<Hash> nxt = take(to2, batch_blocks);
					if (!nxt.empty()) {
						ahead.gen = gen;
						ahead.from = to;
						ahead.to = to2;
						gbm_manager *m = mg;
						ahead.fut = std::async(std::launch::async, [m, nxt]() { return read_scrub_batch(m, nxt); });
						ahead.valid = true;
					}
				}
				Trace tr("scrub step");
				rc = cur.rc ? fail(cur.rc, cur.err)
					    : verify_scrub_batch(mg, cur, st, tr, [&](std::chrono::nanoseconds t) { device_time += t; });
			} catch (const std::exception &e) {
				rc = fail(GBM_E_IO, std::string("scrub worker: ") + e.what());
			}
			lk.lock();
			if (rc != GBM_OK) {
				// Worker::work returned Err: the step is logged and tried again (util/background/worker.rs)
				++errors;
				std::fprintf(stderr, "garage_block: scrub worker: %s\n", gbm_last_error());
				listing.drop();
				lk.unlock();
				drop_ahead();
				lk.lock();
				if (!stop)
					wait_ms(lk, 10000);
				continue;
			}
			if (gen != generation) {  // a command came in meanwhile: this step is done again when (if) the pass goes on
				listing.drop();
				continue;
			}
			it = to;
			blocks += st[0];
			corruptions += st[1];
			mg->scrub_corruptions += st[1];
			const uint64_t t_now = mg->now();
			if (cur.batch.empty() && to.prefix >= 256) {
				// the pass is complete (:469-485)
				t_last_complete = t_now;
				t_next_run = randomize_next_run(t_now);
				has_cp = false;
				save();
				state = GBM_SCRUB_FINISHED;
				mg->scrub_last_complete_ms = t_now;
				listing = Listing();
				continue;
			}
			if (st[1])
				save();  // persister.set_with(|p| p.corruptions_detected += 1) (:455-458)
			if (t_now - t_cp > cp_interval_ms) {  // (:463-467)
				cp = it;
				has_cp = true;
				save();
				t_cp = t_now;
			}
			// tranquilizer.tranquilize_worker (:470-472): sleep tranquility x the time the step kept the device busy
			if (const uint32_t tranq = mg->scrub_tranquility.load()) {
				const auto pause = device_time * tranq;
				const auto until = std::chrono::system_clock::now() + std::chrono::duration_cast<std::chrono::microseconds>(pause);
				mg->tranquilized_ms += (uint64_t)std::chrono::duration_cast<std::chrono::milliseconds>(pause).count();
				while (!stop && gen == generation && std::chrono::system_clock::now() < until)
					cv.wait_until(lk, until);
			}
		}
		lk.unlock();
		drop_ahead();
		lk.lock();
	}
};

void scrub_worker_tranquility_changed(gbm_manager *mg)
{
	std::shared_ptr<ScrubWorker> w;
	{
		std::lock_guard<std::mutex> g(mg->scrub_worker_mu);
		w = mg->scrub_worker;
	}
	if (w) {
		std::lock_guard<std::mutex> g(w->mu);
		w->save();
	}
}

void scrub_worker_wake(gbm_manager *mg)
{
	std::shared_ptr<ScrubWorker> w;
	{
		std::lock_guard<std::mutex> g(mg->scrub_worker_mu);
		w = mg->scrub_worker;
	}
	if (w) {
		{
			std::lock_guard<std::mutex> g(w->mu);  // the worker is either before its look at the clock or already waiting
		}
		w->cv.notify_all();
	}
}

}  // namespace gbmimpl

extern "C" {

int gbm_scrub_worker_start(gbm_manager *m, const char *persist_path, size_t batch_blocks, uint64_t checkpoint_interval_ms)
{
	if (!m)
		return fail(GBM_E_INVALID_ARG, "NULL manager");
	if (m->is_front()) {  // one ScrubWorker and one state file per device
		for (size_t i = 0; i < m->lanes.size(); ++i) {
			const std::string p = persist_path ? std::string(persist_path) + ".dev" + std::to_string(i) : std::string();
			int rc = gbm_scrub_worker_start(m->lanes[i].get(), persist_path ? p.c_str() : nullptr, batch_blocks, checkpoint_interval_ms);
			if (rc)
				return rc;
		}
		return GBM_OK;
	}
	try {
		std::lock_guard<std::mutex> g(m->scrub_worker_mu);
		if (m->scrub_worker)
			return GBM_OK;
		auto w = std::make_shared<ScrubWorker>();
		w->mg = m;
		w->path = persist_path ? persist_path : "";
		w->batch_blocks = batch_blocks ? batch_blocks : 1024;
		w->cp_interval_ms = checkpoint_interval_ms ? checkpoint_interval_ms : 60000;
		const uint64_t now = m->now();
		if (!w->load()) {
			// ScrubWorkerPersisted::default (:258-268)
			w->t_next_run = ScrubWorker::randomize_next_run(now);
			if (!m->scrub_tranquility_set.load())
				m->scrub_tranquility = GBM_INITIAL_SCRUB_TRANQUILITY;
		}
		if (w->has_cp) {  // a checkpoint: the pass goes on where it was (ScrubWorker::new, :313-319)
			w->state = GBM_SCRUB_RUNNING;
			w->it = w->cp;
			w->t_cp = now;
		}
		if (w->t_last_complete > m->scrub_last_complete_ms.load())
			m->scrub_last_complete_ms = w->t_last_complete;
		ScrubWorker *raw = w.get();
		w->th = std::thread([raw, m] {
			lane_thread("gbm-scrub", m->codec);
			raw->run();
		});
		m->scrub_worker = std::move(w);
	} catch (const std::exception &e) {
		return fail(GBM_E_IO, std::string("gbm_scrub_worker_start: ") + e.what());
	}
	return GBM_OK;
}

int gbm_scrub_worker_stop(gbm_manager *m)
{
	if (!m)
		return fail(GBM_E_INVALID_ARG, "NULL manager");
	for (auto &l : m->lanes)
		(void)gbm_scrub_worker_stop(l.get());
	std::shared_ptr<ScrubWorker> w;
	{
		std::lock_guard<std::mutex> g(m->scrub_worker_mu);
		w = std::move(m->scrub_worker);
		m->scrub_worker.reset();
	}
	if (!w)
		return GBM_OK;
	{
		std::lock_guard<std::mutex> g(w->mu);
		w->stop = true;
	}
	w->cv.notify_all();
	if (w->th.joinable())
		w->th.join();
	std::lock_guard<std::mutex> g(w->mu);
	if (w->state != GBM_SCRUB_FINISHED) {  // what the next start carries on from
		w->cp = w->it;
		w->has_cp = true;
		w->save();
	}
	return GBM_OK;
}

int gbm_scrub_worker_command(gbm_manager *m, int cmd, uint64_t pause_ms)
try {
	if (!m)
		return fail(GBM_E_INVALID_ARG, "NULL manager");
	if (m->is_front()) {
		int rc = GBM_OK;
		std::string err;
		for (auto &l : m->lanes) {
			const int r = gbm_scrub_worker_command(l.get(), cmd, pause_ms);
			if (r && !rc) {
				rc = r;
				err = gbm_last_error();
			}
		}
		return rc ? fail(rc, err) : GBM_OK;
	}
	std::shared_ptr<ScrubWorker> w;
	{
		std::lock_guard<std::mutex> g(m->scrub_worker_mu);
		w = m->scrub_worker;
	}
	if (!w)
		return fail(GBM_E_INVALID_ARG, "no scrub worker: call gbm_scrub_worker_start first");
	std::lock_guard<std::mutex> g(w->mu);
	return w->command(cmd, pause_ms);
}
GBM_CATCH

int gbm_scrub_worker_status(const gbm_manager *cm, gbm_scrub_status *out)
{
	if (!cm || !out)
		return fail(GBM_E_INVALID_ARG, "NULL argument");
	gbm_manager *m = const_cast<gbm_manager *>(cm);
	if (m->is_front()) {
		gbm_scrub_status agg{};
		agg.state = GBM_SCRUB_NO_WORKER;
		bool first = true;
		for (auto &l : m->lanes) {
			gbm_scrub_status s;
			int rc = gbm_scrub_worker_status(l.get(), &s);
			if (rc)
				return rc;
			if (s.state == GBM_SCRUB_NO_WORKER)
				continue;
			if (first) {
				agg = s;
				agg.progress = 0;
			} else {
				if (s.state == GBM_SCRUB_RUNNING || (s.state == GBM_SCRUB_PAUSED && agg.state != GBM_SCRUB_RUNNING))
					agg.state = s.state;
				agg.corruptions_detected += s.corruptions_detected;
				agg.blocks_scrubbed += s.blocks_scrubbed;
				agg.checkpoints_saved += s.checkpoints_saved;
				agg.errors += s.errors;
				agg.time_last_complete_scrub_ms = std::min(agg.time_last_complete_scrub_ms, s.time_last_complete_scrub_ms);
				agg.time_next_run_scrub_ms = std::min(agg.time_next_run_scrub_ms, s.time_next_run_scrub_ms);
				agg.resume_at_ms = std::max(agg.resume_at_ms, s.resume_at_ms);
			}
			agg.progress += s.progress / (double)m->lanes.size();
			first = false;
		}
		*out = agg;
		return GBM_OK;
	}
	std::shared_ptr<ScrubWorker> w;
	{
		std::lock_guard<std::mutex> g(m->scrub_worker_mu);
		w = m->scrub_worker;
	}
	*out = gbm_scrub_status{};
	out->tranquility = m->scrub_tranquility.load();
	if (!w) {
		out->state = GBM_SCRUB_NO_WORKER;
		out->progress = 1.0;
		out->time_last_complete_scrub_ms = m->scrub_last_complete_ms.load();
		out->corruptions_detected = m->scrub_corruptions.load();
		return GBM_OK;
	}
	std::lock_guard<std::mutex> g(w->mu);
	out->state = w->state;
	out->progress = w->state == GBM_SCRUB_FINISHED ? 1.0 : w->it.progress();
	out->corruptions_detected = w->corruptions;
	out->time_last_complete_scrub_ms = w->t_last_complete;
	out->time_next_run_scrub_ms = w->t_next_run;
	out->resume_at_ms = w->state == GBM_SCRUB_PAUSED ? w->t_resume : 0;
	out->blocks_scrubbed = w->blocks;
	out->checkpoints_saved = w->cps;
	out->errors = w->errors;
	return GBM_OK;
}

}  // extern "C"
