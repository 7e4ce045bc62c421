// bm_core.cpp -- errors, zstd, the environment table, the manager's life cycle and settings, fault-injection hooks
// for tests, metrics.
#include "bm_internal.hpp"

#include <dlfcn.h>

namespace gbmimpl {

namespace {
thread_local std::string g_err;
}

int fail(int code, const std::string &msg)
{
	g_err = msg;
	return code;
}

const std::string &last_error() { return g_err; }

int ec_fail(int rc, const char *what)
{
	return fail(GBM_E_EC, std::string(what) + ": " + gec_strerror(rc) + " (" + gec_last_error() + ")");
}

// ------------------------------------------------------------------ environment: the one table of GBM_* switches
// (all optional, read once per process, none changes results; gbm_env_table() returns it as text)
namespace {
struct EnvRow {
	const char *name, *def, *doc;
};
const EnvRow kEnvRows[] = {
	{"GBM_TRACE", "0", "1 = stage timings of the batched put / get / resync / scrub on stderr"},
	{"GBM_PUT_SLICE", "64", "blocks per slice of a large untagged put"},
	{"GBM_PUT_THREADS", "4", "put slices in flight"},
	{"GBM_BATCHER_WORKERS", "2", "batches the coalescing batcher keeps in flight (per device)"},
	{"GBM_BATCHER_SPLIT_MIN", "12", "a batcher worker that finds this many blocks queued while other workers are idle takes only its share of them, and half of a queue twice this long even when none is idle (0 = never split)"},
	{"GBM_BATCHER_GET_SPLIT_MIN", "16", "the same rule for the queue's read side (a big read batch goes in pipelined pieces of its own: it is cut later than a put batch)"},
	{"GBM_PUT_SPOT_CHECK", "16", "every Nth put trip one device-computed shard checksum is re-computed on the host before anything is sent to a node (0 = never, 1 = every trip; gbm_set_put_spot_check overrides it per manager)"},
};
long env_long(const char *name, long def)
{
	const char *e = std::getenv(name);
	return e && *e ? std::atol(e) : def;
}
}  // namespace

const Env &env()
{
	static const Env e = [] {
		Env v;
		v.trace = env_long("GBM_TRACE", 0) == 1;
		const long sl = env_long("GBM_PUT_SLICE", 0);
		v.put_slice = (size_t)(sl > 0 ? sl : 64);
		const long pt = env_long("GBM_PUT_THREADS", 0);
		v.put_threads = (int)(pt > 0 && pt <= 8 ? pt : 4);
		const long bw = env_long("GBM_BATCHER_WORKERS", 0);
		v.batcher_workers = (int)(bw >= 1 && bw <= 16 ? bw : 2);
		const long sm = env_long("GBM_BATCHER_SPLIT_MIN", 12);
		v.batcher_split_min = (size_t)(sm >= 0 ? sm : 12);
		const long gsm = env_long("GBM_BATCHER_GET_SPLIT_MIN", 16);
		v.batcher_get_split_min = (size_t)(gsm >= 0 ? gsm : 16);
		const long sc = env_long("GBM_PUT_SPOT_CHECK", 16);
		v.put_spot_check = (unsigned)(sc >= 0 && sc <= 1000000 ? sc : 16);
		// the host forms of shard checksum v3 and of BLAKE2b are header-only code with a copy in each library: this library's
		// copies follow GEC_CPU_ISA as libgarage_ec's do (ec_env.cpp), so that one switch reaches the manager's shard checks and
		// block hashes as well (scalar / avx2: no AVX-512 forms)
		const char *isa = std::getenv("GEC_CPU_ISA");
		if (isa && std::string(isa) == "scalar")
			mlh::isa_cap().store(0);
		else if (isa && std::string(isa) == "avx2")
			mlh::isa_cap().store(1);
		if (isa && (std::string(isa) == "scalar" || std::string(isa) == "avx2"))
			b2host::mb_mode().store(0);
		return v;
	}();
	return e;
}

namespace {
const bool kEnvReadAtLoad = (env(), true);  // GEC_CPU_ISA acts on header-only code: in force from the first hash on
}  // namespace

const char *env_table_text()
{
	static const std::string text = [] {
		std::string s;
		for (const EnvRow &r : kEnvRows)
			s += std::string(r.name) + "\t" + r.def + "\t" + r.doc + "\n";
		return s;
	}();
	return text.c_str();
}

void Trace::lap(const char *stage)
{
	if (!env().trace)
		return;
	const auto t = std::chrono::steady_clock::now();
	char buf[64];
	std::snprintf(buf, sizeof buf, " %s %.2f ms", stage, std::chrono::duration<double, std::milli>(t - t0).count());
	line += buf;
	t0 = t;
}

Trace::~Trace()
{
	if (env().trace && !line.empty())
		std::fprintf(stderr, "[gbm] %s:%s\n", what, line.c_str());
}

Zstd::Zstd()
{
	void *h = dlopen("libzstd.so.1", RTLD_NOW | RTLD_LOCAL);
	if (!h)
		return;
#define GBM_SYM(field, name) field = reinterpret_cast<decltype(field)>(dlsym(h, name))
	GBM_SYM(createCCtx, "ZSTD_createCCtx");
	GBM_SYM(freeCCtx, "ZSTD_freeCCtx");
	GBM_SYM(setParameter, "ZSTD_CCtx_setParameter");
	GBM_SYM(compress2, "ZSTD_compress2");
	GBM_SYM(compressBound, "ZSTD_compressBound");
	GBM_SYM(decompress, "ZSTD_decompress");
	GBM_SYM(getFrameContentSize, "ZSTD_getFrameContentSize");
	GBM_SYM(isError, "ZSTD_isError");
	GBM_SYM(createDStream, "ZSTD_createDStream");  // optional: without them compressed blocks are decoded in one piece
	GBM_SYM(freeDStream, "ZSTD_freeDStream");
	GBM_SYM(initDStream, "ZSTD_initDStream");
	GBM_SYM(decompressStream, "ZSTD_decompressStream");
#undef GBM_SYM
	ok = createCCtx && freeCCtx && setParameter && compress2 && compressBound && decompress &&
	     getFrameContentSize && isError;
	streaming = ok && createDStream && freeDStream && initDStream && decompressStream;
}

Zstd::Stream::Stream(const Zstd &zz) : z(&zz)
{
	if (z->streaming && (ds = z->createDStream()) != nullptr && z->isError(z->initDStream(ds))) {
		z->freeDStream(ds);
		ds = nullptr;
	}
}
Zstd::Stream::~Stream()
{
	if (ds)
		z->freeDStream(ds);
}
bool Zstd::Stream::feed(const uint8_t *in, size_t len, const std::function<bool(const uint8_t *, size_t)> &emit)
{
	struct Buf {  // ZSTD_inBuffer / ZSTD_outBuffer
		const void *p;
		size_t size, pos;
	};
	if (!ds)
		return false;
	uint8_t out[1 << 16];
	Buf ib{in, len, 0};
	while (ib.pos < ib.size) {
		if (frame_done)
			return false;  // bytes behind the end of the frame: not a DataBlock
		Buf ob{out, sizeof out, 0};
		const size_t r = z->decompressStream(ds, &ob, &ib);
		if (z->isError(r))
			return false;
		if (ob.pos && !emit(out, ob.pos))
			return false;
		if (r == 0)
			frame_done = true;
	}
	// flush what the decoder still holds (it may have consumed all input with output pending)
	while (!frame_done) {
		Buf ob{out, sizeof out, 0};
		Buf none{in, 0, 0};
		const size_t r = z->decompressStream(ds, &ob, &none);
		if (z->isError(r))
			return false;
		if (ob.pos && !emit(out, ob.pos))
			return false;
		if (r == 0)
			frame_done = true;
		if (ob.pos < sizeof out)
			break;  // nothing more without more input
	}
	return true;
}
// false on any error: the caller then stores the block Plain (block.rs:88-93)
bool Zstd::encode(const uint8_t *data, size_t len, int level, std::vector<uint8_t> &out) const
{
	if (!ok)
		return false;
	void *c = createCCtx();
	if (!c)
		return false;
	bool good = !isError(setParameter(c, 100 /* ZSTD_c_compressionLevel */, level)) &&
		    !isError(setParameter(c, 201 /* ZSTD_c_checksumFlag */, 1));
	if (good) {
		out.resize(compressBound(len));
		size_t n = compress2(c, out.data(), out.size(), data, len);
		good = !isError(n);
		if (good)
			out.resize(n);
	}
	freeCCtx(c);
	return good;
}
// verifies the frame checksum; false = corrupt.  `max_out` bounds the allocation: a damaged or
// forged frame header must not be able to ask for terabytes (Garage blocks are <= block_size, and a
// zstd frame cannot expand by more than ~2^17 per byte; the caller passes a generous multiple of
// block_size).  Frames without a content-size field (streaming encoders write those) are decoded
// into a buffer that grows up to the same bound.
bool Zstd::decode(const uint8_t *data, size_t len, size_t max_out, std::vector<uint8_t> &out) const
{
	if (!ok)
		return false;
	const unsigned long long sz = getFrameContentSize(data, len);
	const unsigned long long UNKNOWN = 0ULL - 1, ERROR_ = 0ULL - 2;
	if (sz == ERROR_)
		return false;
	try {
		if (sz != UNKNOWN) {
			if (sz > max_out)
				return false;
			out.resize((size_t)sz);
			uint8_t dummy;
			size_t n = decompress(sz ? out.data() : &dummy, (size_t)sz, data, len);
			return !isError(n) && n == sz;
		}
		size_t cap = std::min<size_t>(std::max<size_t>(4 * len, 1 << 16), max_out);
		for (;;) {
			out.resize(cap);
			size_t n = decompress(out.data(), cap, data, len);
			if (!isError(n)) {
				out.resize(n);
				return true;
			}
			if (cap >= max_out)
				return false;  // corrupt, or larger than any block can be
			cap = std::min(cap * 4, max_out);
		}
	} catch (const std::bad_alloc &) {
		return false;
	}
}
const Zstd &zstd()
{
	static const Zstd z;
	return z;
}

uint64_t real_now_ms()
{
	return (uint64_t)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::system_clock::now().time_since_epoch()).count();
}

// shard checksums of many buffers: on the GPU (gec_shardsum_batch) once the batch is big enough to beat the
// CPU pool through PCIe, else on the pool's threads.  SURVEY.md section 8 row f4.
constexpr size_t kGpuHashMinMessages = 64;
constexpr size_t kGpuHashMinBytes = 8u << 20;

int hash_many(gbm_manager *mg, const std::vector<const uint8_t *> &ptrs, const std::vector<size_t> &lens,
	      std::vector<uint8_t> &sums)
{
	sums.resize(ptrs.size() * 32);
	size_t total = 0;
	for (size_t l : lens)
		total += l;
	// (on a CPU codec the library's own threads hash: no link to amortise, any batch may go)
	if (ptrs.size() >= kGpuHashMinMessages && total >= kGpuHashMinBytes) {
		int rc = gec_shardsum_batch(mg->codec, ptrs.size(), ptrs.data(), lens.data(), sums.data());
		if (rc)
			return ec_fail(rc, "gec_shardsum_batch");
		mg->gpu_hashed += ptrs.size();
		return GBM_OK;
	}
	// shards per pool task: their leaves go through the cores' vector lanes eight at a time -- where there are such
	// lanes; the one-at-a-time fallback keeps one shard per task, so a small batch still spreads over the pool
	const size_t kPerTask = b2host::mb_available() || mg->sumver == 3 ? 16 : 1;
	mg->pool->parallel_for((ptrs.size() + kPerTask - 1) / kPerTask, [&](size_t g) {
		const size_t i0 = g * kPerTask, cnt = std::min(kPerTask, ptrs.size() - i0);
		if (mg->sumver == 3)
			for (size_t i = i0; i < i0 + cnt; ++i)
				mlh::shardsum3(ptrs[i], lens[i], sums.data() + 32 * i);
		else
			b2host::shardsum_many(ptrs.data() + i0, lens.data() + i0, cnt, sums.data() + 32 * i0);
	});
	return GBM_OK;
}

// fn(lane) for every lane of a front, each on a thread of its own (the calling thread takes lane 0): the lanes are
// independent managers over different devices, so a batch call on the front is ndev batch calls side by side.
// Returns the last non-zero result and re-publishes its (thread-local) error text on the calling thread.
int for_lanes(gbm_manager *front, const std::function<int(gbm_manager *, size_t)> &fn)
{
	const size_t nl = front->lanes.size();
	std::vector<int> rcs(nl, GBM_OK);
	std::vector<std::string> errs(nl);
	auto one = [&](size_t i) {
		try {
			rcs[i] = fn(front->lanes[i].get(), i);
		} catch (const std::exception &e) {
			rcs[i] = fail(GBM_E_IO, e.what());
		}
		if (rcs[i])
			errs[i] = last_error();
	};
	// (a thread that cannot be started -- the process is at its limit -- must not unwind past joinable ones: std::terminate
	// inside a C entry point; its lane, and the ones behind it, then run here, one after the other)
	std::vector<std::thread> th;
	size_t started = 1;
	try {
		th.reserve(nl);
		for (; started < nl; ++started)
			th.emplace_back(one, started);
	} catch (const std::exception &) {
	}
	one(0);
	for (size_t i = started; i < nl; ++i)
		one(i);
	for (auto &t : th)
		t.join();
	for (size_t i = nl; i-- > 0;)
		if (rcs[i])
			return fail(rcs[i], errs[i]);
	return GBM_OK;
}

namespace {
// one complete manager over one codec; `nodes` non-empty: share these storage nodes (a lane of a front)
// Workers of a lane's pool (the calling thread works too): 16 threads in all, fewer on a small host -- what every path was tuned
// with.  Tried in round 6 (profiles/r06_experiments.txt, 5): 32 threads for a lone lane on a 256-CPU host lift the default mode's
// big gets (the pool hashes every block) from 35 to 42 GiB/s, and cost the coalescing queue a fifth of its rate (48 writers: 30 ->
// 18 - 24 GiB/s) and a lone put 20 - 80 us: not taken.  gbm_set_threads is there for a deployment that reads in bulk.
unsigned default_pool_workers(int)
{
	const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
	return std::min(15u, hw - 1);
}

int create_one(const gec_codec *codec, int nnodes, const char *const *node_dirs, int write_quorum,
	       const std::vector<std::shared_ptr<Node>> *nodes, std::unique_ptr<gbm_manager> &out)
{
	const int k = gec_codec_k(codec), m = gec_codec_m(codec);
	if (nnodes < k + m)
		return fail(GBM_E_INVALID_ARG, "RS(k,m) needs at least k+m storage nodes (replication_factor == k+m)");
	if (k + m > 255)
		return fail(GBM_E_INVALID_ARG, "shard index must fit a byte");
	auto mg = std::make_unique<gbm_manager>();
	mg->codec = codec;
	mg->sumver = gec_codec_shardsum(codec);  // the shard-header version it writes: its codec's checksum kind
	// the end-to-end default follows the shard checksum's strength (garage_block.h, gbm_set_verify_block_hash)
	mg->verify_mode = mg->sumver == 3 ? GBM_VERIFY_ALWAYS : GBM_VERIFY_REBUILT;
	mg->k = k;
	mg->m = m;
	mg->n = k + m;
	mg->write_quorum = write_quorum > 0 ? write_quorum : k + (m + 1) / 2;
	if (mg->write_quorum < k || mg->write_quorum > mg->n)
		return fail(GBM_E_INVALID_ARG, "write quorum must be in [k, k+m]");
	if (nodes) {
		mg->nodes = *nodes;
	} else {
		for (int i = 0; i < nnodes; ++i) {
			mg->nodes.push_back(node_dirs ? make_dir_node(node_dirs[i]) : make_memory_node());
			mg->nodes.back()->bufs = mg->bufs;
		}
	}
	mg->bufs->near = codec;  // shard buffers on the codec's memory node (numa.hpp; before the first buffer is drawn)
	mg->pool.reset(new Pool(default_pool_workers(1), codec));
	mg->cpu_block_hash_max = 6 * ((size_t)mg->pool->workers() + 1);
	mg->put_spot_every = env().put_spot_check;
	// maintenance gets a background-class sibling of the codec (its own staging slots, low-priority streams on a
	// subset of the CUs, small chunks that yield to the request path); without one it shares the request path's codec
	if (gec_codec_background(codec, &mg->bg_codec_owned) != GEC_OK)
		mg->bg_codec_owned = nullptr;
	out = std::move(mg);
	return GBM_OK;
}

void destroy_one(gbm_manager *m)
{
	gbm_resync_worker_stop(m);
	gbm_scrub_worker_stop(m);
	m->async.reset();  // drains: abandoned hedged requests still point at the nodes
	if (m->bg_codec_owned)
		gec_codec_destroy(m->bg_codec_owned);
	m->bg_codec_owned = nullptr;
}
}  // namespace

}  // namespace gbmimpl

using namespace gbmimpl;

bool gbmimpl::confirmed_corrupt(gbm_manager *mg, const uint8_t *data, size_t S, const uint8_t header_sum[32], const char *who)
{
	uint8_t sum[32];
	shardsum_v(mg->sumver, data, S, sum);
	if (std::memcmp(sum, header_sum, 32) != 0)
		return true;
	mg->bmx.unconfirmed_verdicts++;
	std::fprintf(stderr, "garage_block: %s reported a checksum mismatch that the host does not confirm (shard of %zu bytes): the shard stays\n", who, S);
	return false;
}

// `expr` on the manager itself, or on every lane of a front (and on the front, whose copy of the setting only
// answers the getters)
#define GBM_EACH(m, var, expr)                          \
	do {                                            \
		{                                       \
			gbm_manager *var = (m);         \
			expr;                           \
		}                                       \
		for (auto &lane_ : (m)->lanes) {        \
			gbm_manager *var = lane_.get(); \
			expr;                           \
		}                                       \
	} while (0)

extern "C" {

const char *gbm_last_error(void) { return last_error().c_str(); }

const char *gbm_env_table(void) { return env_table_text(); }

void gbm_blake2sum(const uint8_t *data, size_t len, uint8_t out[32])
{
	if (out && (data || len == 0))
		blake2sum(data, len, out);  // (no allocation: nothing to throw)
}

void gbm_shardsum(const uint8_t *data, size_t len, uint8_t out[32])
{
	if (!out || (!data && len))
		return;
	try {
		shardsum_v(2, data, len, out);
	} catch (const std::exception &) {  // the eight-lane form's scratch could not grow: one leaf at a time needs none
		b2host::shardsum(data, len, out);
	}
}

int gbm_shardsum_v(int version, const uint8_t *data, size_t len, uint8_t out[32])
{
	if (!out || (!data && len) || version < 1 || version > 3)
		return fail(GBM_E_INVALID_ARG, "gbm_shardsum_v: NULL argument or unknown shard-header version");
	try {  // (the v3 form allocates its leaf sums above 256 KiB: nothing may unwind across the C ABI)
		if (version == 2)
			gbm_shardsum(data, len, out);
		else
			shardsum_v(version, data, len, out);
	} catch (const std::exception &e) {
		return fail(GBM_E_IO, std::string("gbm_shardsum_v: ") + e.what());
	}
	return GBM_OK;
}

int gbm_shard_version(const gbm_manager *m) { return m ? (m->is_front() ? m->lanes[0]->sumver : m->sumver) : -1; }

int gbm_blake2sum_batch(size_t n, const uint8_t *const *data, const size_t *len, uint8_t *out)
{
	if (n == 0)
		return GBM_OK;
	if (!data || !len || !out)
		return fail(GBM_E_INVALID_ARG, "NULL argument");
	for (size_t i = 0; i < n; ++i)
		if (!data[i] && len[i])
			return fail(GBM_E_INVALID_ARG, "NULL message pointer");
	try {
		b2host::blake2sum_many(data, len, n, out);
	} catch (const std::exception &e) {  // nothing may unwind across the C ABI
		return fail(GBM_E_IO, std::string("gbm_blake2sum_batch: ") + e.what());
	}
	return GBM_OK;
}

int gbm_create(const gec_codec *codec, int nnodes, const char *const *node_dirs, int write_quorum, gbm_manager **out)
{
	if (!codec || !out)
		return fail(GBM_E_INVALID_ARG, "NULL argument");
	*out = nullptr;
	std::unique_ptr<gbm_manager> mg;
	int rc = create_one(codec, nnodes, node_dirs, write_quorum, nullptr, mg);
	if (rc)
		return rc;
	*out = mg.release();
	return GBM_OK;
}

int gbm_create_multi(const gec_codec *const *codecs, int ndev, int nnodes, const char *const *node_dirs, int write_quorum,
		     gbm_manager **out)
{
	if (!codecs || !out || ndev < 1 || ndev > 256)
		return fail(GBM_E_INVALID_ARG, "need 1 <= ndev <= 256 codecs");
	*out = nullptr;
	for (int d = 0; d < ndev; ++d) {
		if (!codecs[d])
			return fail(GBM_E_INVALID_ARG, "NULL codec");
		if (gec_codec_k(codecs[d]) != gec_codec_k(codecs[0]) || gec_codec_m(codecs[d]) != gec_codec_m(codecs[0]))
			return fail(GBM_E_INVALID_ARG, "every device's codec must be the same RS(k,m)");
		for (int e = 0; e < d; ++e)
			if (codecs[e] == codecs[d])
				return fail(GBM_E_INVALID_ARG, "one codec per device: the same codec was given twice");
	}
	try {
		auto front = std::make_unique<gbm_manager>();
		for (int d = 0; d < ndev; ++d) {
			std::unique_ptr<gbm_manager> lane;
			int rc = create_one(codecs[d], nnodes, node_dirs, write_quorum, d ? &front->nodes : nullptr, lane);
			if (rc) {
				for (auto &l : front->lanes)
					destroy_one(l.get());
				return rc;
			}
			lane->lane_idx = d;
			lane->lane_cnt = ndev;
			if (lane->pool && lane->pool->workers() != default_pool_workers(ndev)) {  // (the lanes share the host's CPUs)
				lane->pool->resize(default_pool_workers(ndev));
				lane->cpu_block_hash_max = 6 * ((size_t)lane->pool->workers() + 1);
			}
			if (d == 0) {
				front->nodes = lane->nodes;
				front->k = lane->k;
				front->m = lane->m;
				front->n = lane->n;
				front->write_quorum = lane->write_quorum;
			}
			front->lanes.push_back(std::move(lane));
		}
		*out = front.release();
	} catch (const std::exception &e) {
		return fail(GBM_E_IO, std::string("gbm_create_multi: ") + e.what());
	}
	return GBM_OK;
}

void gbm_destroy(gbm_manager *m)
{
	if (!m)
		return;
	for (auto &l : m->lanes)
		destroy_one(l.get());
	if (!m->is_front())
		destroy_one(m);
	delete m;
}

int gbm_device_count(const gbm_manager *m) { return !m ? 0 : m->is_front() ? (int)m->lanes.size() : 1; }

int gbm_device_of_hash(const gbm_manager *m, const uint8_t hash[32])
{
	if (!m || !hash)
		return -1;
	return m->is_front() ? gec_device_of_hash(hash, (int)m->lanes.size()) : 0;
}

static const gbm_manager *lane_of(const gbm_manager *m, int dev)
{
	if (!m || dev < 0 || dev >= gbm_device_count(m))
		return nullptr;
	return m->is_front() ? m->lanes[(size_t)dev].get() : m;
}

const gec_codec *gbm_device_codec(const gbm_manager *m, int dev)
{
	const gbm_manager *l = lane_of(m, dev);
	return l ? l->codec : nullptr;
}

const gec_codec *gbm_device_background_codec(const gbm_manager *m, int dev)
{
	const gbm_manager *l = lane_of(m, dev);
	return l ? l->bg_codec() : nullptr;
}

int gbm_device_metrics(const gbm_manager *m, int dev, uint64_t out[6])
{
	const gbm_manager *l = lane_of(m, dev);
	if (!l || !out)
		return fail(GBM_E_INVALID_ARG, "bad device index / NULL argument");
	for (int i = 0; i < 6; ++i)
		out[i] = l->metrics[i].load();
	return GBM_OK;
}

int gbm_set_threads(gbm_manager *m, int nthreads)
try {
	if (!m || nthreads < 1 || nthreads > 256)
		return fail(GBM_E_INVALID_ARG, "need 1 <= nthreads <= 256");
	GBM_EACH(m, x, {
		if (x->pool)
			x->pool->resize((unsigned)nthreads - 1);  // the calling thread works too
		x->cpu_block_hash_max = 6 * (size_t)nthreads;
	});
	return GBM_OK;
}
GBM_CATCH

int gbm_set_host_block_hash_max(gbm_manager *m, size_t nblocks)
{
	if (!m)
		return fail(GBM_E_INVALID_ARG, "NULL manager");
	GBM_EACH(m, x, x->cpu_block_hash_max = nblocks);
	return GBM_OK;
}

int gbm_set_data_fsync(gbm_manager *m, int enabled)
{
	if (!m)
		return fail(GBM_E_INVALID_ARG, "NULL manager");
	for (auto &nd : m->nodes)
		nd->set_fsync(enabled != 0);
	return GBM_OK;
}

int gbm_set_verify_block_hash(gbm_manager *m, int mode)
{
	if (!m || (mode != GBM_VERIFY_OFF && mode != GBM_VERIFY_ALWAYS && mode != GBM_VERIFY_REBUILT))
		return fail(GBM_E_INVALID_ARG, "mode must be GBM_VERIFY_OFF, GBM_VERIFY_ALWAYS or GBM_VERIFY_REBUILT");
	GBM_EACH(m, x, x->verify_mode = mode);
	return GBM_OK;
}

int gbm_get_verify_block_hash(const gbm_manager *m) { return m ? m->verify_mode.load() : GBM_E_INVALID_ARG; }

int gbm_set_migrate_on_read(gbm_manager *m, int enabled)
{
	if (!m)
		return fail(GBM_E_INVALID_ARG, "NULL manager");
	GBM_EACH(m, x, x->migrate_on_read = enabled != 0);
	return GBM_OK;
}

uint64_t gbm_shards_migrated(const gbm_manager *m)
{
	if (!m)
		return 0;
	if (!m->is_front())
		return m->shards_migrated.load();
	uint64_t t = 0;
	for (auto &l : m->lanes)
		t += l->shards_migrated.load();
	return t;
}

int gbm_set_tranquility(gbm_manager *m, int scrub_tranquility, int resync_tranquility)
{
	if (!m)
		return fail(GBM_E_INVALID_ARG, "NULL manager");
	GBM_EACH(m, x, {
		if (scrub_tranquility >= 0) {
			x->scrub_tranquility = (uint32_t)scrub_tranquility;
			x->scrub_tranquility_set = true;
			scrub_worker_tranquility_changed(x);
		}
		if (resync_tranquility >= 0) {
			x->resync_tranquility = (uint32_t)resync_tranquility;
			x->resync_tranquility_set = true;
			resync_config_changed(x);
		}
	});
	return GBM_OK;
}

int gbm_set_put_spot_check(gbm_manager *m, unsigned every_n)
{
	if (!m)
		return fail(GBM_E_INVALID_ARG, "NULL manager");
	GBM_EACH(m, x, x->put_spot_every = every_n);
	return GBM_OK;
}

int gbm_test_corrupt_put_sums(gbm_manager *m, int trips)
{
	if (!m || trips < 0)
		return fail(GBM_E_INVALID_ARG, "bad argument");
	GBM_EACH(m, x, x->test_bad_put_sums = trips);
	return GBM_OK;
}

int gbm_get_tranquility(const gbm_manager *m, uint32_t out[2])
{
	if (!m || !out)
		return fail(GBM_E_INVALID_ARG, "NULL argument");
	out[0] = m->scrub_tranquility.load();
	out[1] = m->resync_tranquility.load();
	return GBM_OK;
}

uint64_t gbm_tranquilized_ms(const gbm_manager *m)
{
	if (!m)
		return 0;
	uint64_t v = m->tranquilized_ms.load();
	for (auto &l : m->lanes)
		v += l->tranquilized_ms.load();
	return v;
}

int gbm_set_maintenance_class(gbm_manager *m, int background)
{
	if (!m)
		return fail(GBM_E_INVALID_ARG, "NULL manager");
	GBM_EACH(m, x, x->maintenance_on_bg = background != 0);
	return GBM_OK;
}

const gec_codec *gbm_background_codec(const gbm_manager *m) { return gbm_device_background_codec(m, 0); }

int gbm_set_read_hedge(gbm_manager *m, uint64_t hedge_us)
{
	if (!m)
		return fail(GBM_E_INVALID_ARG, "NULL manager");
	GBM_EACH(m, x, x->hedge_us = hedge_us);
	return GBM_OK;
}

uint64_t gbm_hedged_reads(const gbm_manager *m)
{
	if (!m)
		return 0;
	uint64_t v = m->hedged_reads.load();
	for (auto &l : m->lanes)
		v += l->hedged_reads.load();
	return v;
}

int gbm_node_set_latency(gbm_manager *m, int node, uint64_t latency_us)
{
	if (!m || node < 0 || node >= (int)m->nodes.size())
		return fail(GBM_E_INVALID_ARG, "bad node");
	m->nodes[node]->latency_us = latency_us;
	return GBM_OK;
}

int gbm_node_set_zone(gbm_manager *m, int node, int zone)
{
	if (!m || node < 0 || node >= (int)m->nodes.size())
		return fail(GBM_E_INVALID_ARG, "bad node");
	m->nodes[node]->zone = zone;
	GBM_EACH(m, x, x->locality_set = true);
	return GBM_OK;
}

int gbm_node_set_ping(gbm_manager *m, int node, uint64_t ping_us)
{
	if (!m || node < 0 || node >= (int)m->nodes.size())
		return fail(GBM_E_INVALID_ARG, "bad node");
	m->nodes[node]->ping_us = ping_us;
	GBM_EACH(m, x, x->locality_set = true);
	return GBM_OK;
}

int gbm_set_self_node(gbm_manager *m, int node, int zone)
{
	if (!m || node < -1 || node >= (int)m->nodes.size())
		return fail(GBM_E_INVALID_ARG, "bad node (-1 = the requester is not a storage node)");
	GBM_EACH(m, x, {
		x->self_node = node;
		x->self_zone = zone;
		x->locality_set = true;
	});
	return GBM_OK;
}

int gbm_block_read_order(const gbm_manager *m, const uint8_t hash[32], size_t cap, int *nodes_out, int *shards_out, int *versions_out, size_t *count)
try {
	if (!m || !hash || !count || (cap && (!nodes_out || !shards_out || !versions_out)))
		return fail(GBM_E_INVALID_ARG, "NULL argument");
	const gbm_manager *x = m->is_front() ? m->lanes[0].get() : m;
	const Hash h((const char *)hash, 32);
	const int vold = x->layout_oldest.load(), vcur = x->layout_cur.load();
	std::vector<uint32_t> order;
	read_candidate_order(x, h, vold, vcur, order);
	*count = order.size();
	std::vector<int> who;
	int who_v = -1;
	for (size_t i = 0; i < order.size() && i < cap; ++i) {
		const int v = vold + (int)(order[i] / (uint32_t)x->n), j = (int)(order[i] % (uint32_t)x->n);
		if (v != who_v) {
			x->nodes_of(h, v, who);
			who_v = v;
		}
		nodes_out[i] = who[j];
		shards_out[i] = j;
		versions_out[i] = v;
	}
	return GBM_OK;
}
GBM_CATCH

int gbm_set_timing(gbm_manager *m, int64_t gc_delay_ms, int64_t resync_retry_delay_ms, int64_t incref_check_delay_ms)
{
	if (!m)
		return fail(GBM_E_INVALID_ARG, "NULL manager");
	GBM_EACH(m, x, {
		if (gc_delay_ms >= 0)
			x->gc_delay_ms = (uint64_t)gc_delay_ms;
		if (resync_retry_delay_ms >= 0)
			x->retry_delay_ms = (uint64_t)resync_retry_delay_ms;
		if (incref_check_delay_ms >= 0)
			x->incref_delay_ms = (uint64_t)incref_check_delay_ms;
	});
	return GBM_OK;
}

int gbm_clock_advance(gbm_manager *m, uint64_t ms)
{
	if (!m)
		return fail(GBM_E_INVALID_ARG, "NULL manager");
	GBM_EACH(m, x, {
		x->clock_skew_ms += ms;
		x->rs_cv.notify_all();
		scrub_worker_wake(x);  // a pause may be over, the next run may be due
	});
	return GBM_OK;
}

int gbm_storage_nodes_of(const gbm_manager *m, const uint8_t hash[32], int *nodes_out)
try {
	if (!m || !hash || !nodes_out)
		return fail(GBM_E_INVALID_ARG, "NULL argument");
	std::vector<int> who;
	m->nodes_of(Hash((const char *)hash, 32), who);
	std::copy(who.begin(), who.end(), nodes_out);
	return GBM_OK;
}
GBM_CATCH

int gbm_layout_update(gbm_manager *m)
{
	if (!m)
		return fail(GBM_E_INVALID_ARG, "NULL manager");
	for (auto &l : m->lanes)  // the layout is the cluster's, not a device's: every lane follows
		++l->layout_cur;
	return ++m->layout_cur;
}

int gbm_layout_trim(gbm_manager *m)
{
	if (!m)
		return fail(GBM_E_INVALID_ARG, "NULL manager");
	GBM_EACH(m, x, x->layout_oldest = x->layout_cur.load());
	return GBM_OK;
}

int gbm_node_set_down(gbm_manager *m, int node, int down)
{
	if (!m || node < 0 || node >= (int)m->nodes.size())
		return fail(GBM_E_INVALID_ARG, "bad node index");
	m->nodes[node]->down = down != 0;
	return GBM_OK;
}

int gbm_node_has_shard(gbm_manager *m, int node, const uint8_t hash[32], int idx)
{
	if (!m || !hash || node < 0 || node >= (int)m->nodes.size())
		return 0;
	return m->nodes[node]->has(Hash((const char *)hash, 32), idx) ? 1 : 0;
}

int gbm_node_delete_shard(gbm_manager *m, int node, const uint8_t hash[32], int idx)
{
	if (!m || !hash || node < 0 || node >= (int)m->nodes.size())
		return fail(GBM_E_INVALID_ARG, "bad node index");
	m->nodes[node]->del(Hash((const char *)hash, 32), idx);
	return GBM_OK;
}

int gbm_node_corrupt_shard(gbm_manager *m, int node, const uint8_t hash[32], int idx, size_t offset, uint8_t mask,
			   int fix_checksum)
{
	if (!m || !hash || node < 0 || node >= (int)m->nodes.size())
		return fail(GBM_E_INVALID_ARG, "bad node index");
	Hash h((const char *)hash, 32);
	Shard s;
	if (!m->nodes[node]->get(h, idx, s) || s.data.n <= offset)
		return fail(GBM_E_IO, "no such shard / offset");
	try {
		// shard buffers are shared (a data shard is a slice of its block's buffer): corrupt a private copy
		Bytes copy = m->bufs->get(s.data.n);
		std::memcpy(copy.mut(), s.data.data(), s.data.n);
		copy.mut()[offset] ^= mask;
		s.data = copy;
	} catch (const std::bad_alloc &) {
		return fail(GBM_E_IO, "out of memory");
	}
	if (fix_checksum)
		shardsum_v(s.hd.version, s.data.data(), s.data.n, s.hd.checksum);
	return m->nodes[node]->put(h, idx, s) ? GBM_OK : fail(GBM_E_IO, "rewrite failed");
}

int gbm_node_shard_header(gbm_manager *m, int node, const uint8_t hash[32], int idx, uint8_t out[GBM_SHARD_HEADER_SIZE])
{
	if (!m || node < 0 || node >= (int)m->nodes.size() || !hash || !out)
		return fail(GBM_E_INVALID_ARG, "bad argument");
	Shard s;
	if (!m->nodes[node]->get(Hash((const char *)hash, 32), idx, s))
		return fail(GBM_E_IO, "no such shard");
	s.hd.pack(out);
	return GBM_OK;
}

uint64_t gbm_node_requests(gbm_manager *m, int node)
{
	if (!m || node < 0 || node >= (int)m->nodes.size())
		return 0;
	return m->nodes[node]->requests.load();
}

uint64_t gbm_node_order_violations(gbm_manager *m, int node)
{
	if (!m || node < 0 || node >= (int)m->nodes.size())
		return 0;
	return m->nodes[node]->order_violations.load();
}

int gbm_set_compression_level(gbm_manager *m, int enabled, int level)
{
	if (!m)
		return fail(GBM_E_INVALID_ARG, "NULL manager");
	if (enabled && !zstd().ok)
		return fail(GBM_E_IO, "libzstd.so.1 not available");
	GBM_EACH(m, x, {
		x->compression_level = level;
		x->compress = enabled != 0;
	});
	return GBM_OK;
}

uint64_t gbm_gpu_hashed(const gbm_manager *m)
{
	if (!m)
		return 0;
	uint64_t v = m->gpu_hashed.load();
	for (auto &l : m->lanes)
		v += l->gpu_hashed.load();
	return v;
}

int gbm_metrics(const gbm_manager *m, uint64_t out[6])
{
	if (!m || !out)
		return fail(GBM_E_INVALID_ARG, "NULL argument");
	for (int i = 0; i < 6; ++i) {
		out[i] = m->metrics[i].load();
		for (auto &l : m->lanes)  // a front: the sum over its devices (gbm_device_metrics has each one's)
			out[i] += l->metrics[i].load();
	}
	return GBM_OK;
}

}  // extern "C"

// ------------------------------------------------------------------ BlockManagerMetrics (src/block/metrics.rs)
namespace {

void add_hist(const gbmimpl::Histogram &h, gbm_histogram &o)
{
	o.sum_s += (double)h.sum_ns.load(std::memory_order_relaxed) * 1e-9;
	for (int i = 0; i <= GBM_HISTOGRAM_BUCKETS; ++i) {
		const uint64_t c = h.bucket[i].load(std::memory_order_relaxed);
		o.bucket[i] += c;  // per bucket here; made cumulative at the end
		o.count += c;      // readers and writers keep observing while this is read: count = what the buckets say, always
	}
}

void add_one(gbm_manager *x, gbm_block_metrics &o)
{
	for (auto &st : x->rc) {
		std::lock_guard<std::mutex> g(st.mu);
		o.rc_size += st.map.size();
	}
	{
		std::lock_guard<std::mutex> g(x->rs_mu);
		o.resync_queue_length += x->rs_queue.size();
		o.resync_errored_blocks += x->rs_errors.size();
	}
	o.resync_counter += x->bmx.resync_counter.load();
	o.resync_error_counter += x->bmx.resync_error_counter.load();
	o.resync_send_counter += x->bmx.resync_send_counter.load();
	o.resync_recv_counter += x->bmx.resync_recv_counter.load();
	o.delete_counter += x->bmx.delete_counter.load();
	o.unconfirmed_verdicts += x->bmx.unconfirmed_verdicts.load();
	o.put_spot_checks += x->bmx.put_spot_checks.load();
	o.put_spot_check_failures += x->bmx.put_spot_check_failures.load();
	o.bytes_written += x->metrics[0].load();
	o.bytes_read += x->metrics[1].load();
	o.corruption_counter += x->metrics[2].load();
	o.ec_reconstructs += x->metrics[3].load();
	o.blocks_put += x->metrics[4].load();
	o.blocks_get += x->metrics[5].load();
	add_hist(x->bmx.resync_duration, o.resync_duration);
	add_hist(x->bmx.read_duration, o.block_read_duration);
	add_hist(x->bmx.write_duration, o.block_write_duration);
	o.gpu_hashed += x->gpu_hashed.load();
	o.hedged_reads += x->hedged_reads.load();
	o.scrub_corruptions_detected += x->scrub_corruptions.load();
	o.scrub_time_last_complete_ms = std::max<uint64_t>(o.scrub_time_last_complete_ms, x->scrub_last_complete_ms.load());
	o.tranquilized_ms += x->tranquilized_ms.load();
}

void cumulate(gbm_histogram &h)
{
	for (int i = 1; i <= GBM_HISTOGRAM_BUCKETS; ++i)
		h.bucket[i] += h.bucket[i - 1];
}

void collect(const gbm_manager *cm, gbm_batcher *b, gbm_block_metrics &o, bool one_lane_only = false)
{
	gbm_manager *m = const_cast<gbm_manager *>(cm);
	o = gbm_block_metrics{};
	o.compression_level = m->compress.load() ? (uint64_t)std::max(0, m->compression_level.load()) : 0;
	o.devices = m->is_front() && !one_lane_only ? (uint32_t)m->lanes.size() : 1;
	add_one(m, o);
	if (!one_lane_only)
		for (auto &l : m->lanes)
			add_one(l.get(), o);
	cumulate(o.resync_duration);
	cumulate(o.block_read_duration);
	cumulate(o.block_write_duration);
	if (b) {
		uint64_t q[5];
		batcher_snapshot(b, q);
		o.ram_buffer_free_kb = q[0];
		o.batcher_put_batches = q[1];
		o.batcher_put_blocks = q[2];
		o.batcher_get_batches = q[3];
		o.batcher_get_blocks = q[4];
	}
}

void put_line(std::string &s, const char *name, const char *labels, double v)
{
	char buf[96];
	if (v == (double)(uint64_t)v && v < 1e18)
		std::snprintf(buf, sizeof(buf), "%llu", (unsigned long long)v);
	else
		std::snprintf(buf, sizeof(buf), "%.9g", v);
	s += name;
	s += labels;
	s += ' ';
	s += buf;
	s += '\n';
}

void put_metric(std::string &s, const char *name, const char *type, const char *help, double v)
{
	s += std::string("# HELP ") + name + " " + help + "\n# TYPE " + name + " " + type + "\n";
	put_line(s, name, "", v);
}

void put_hist(std::string &s, const char *name, const char *help, const gbm_histogram &h)
{
	s += std::string("# HELP ") + name + " " + help + "\n# TYPE " + name + " histogram\n";
	const double *b = gbmimpl::Histogram::bounds();
	const std::string bn = std::string(name) + "_bucket";
	char lab[48];
	for (int i = 0; i < GBM_HISTOGRAM_BUCKETS; ++i) {
		std::snprintf(lab, sizeof(lab), "{le=\"%g\"}", b[i]);
		put_line(s, bn.c_str(), lab, (double)h.bucket[i]);
	}
	put_line(s, bn.c_str(), "{le=\"+Inf\"}", (double)h.bucket[GBM_HISTOGRAM_BUCKETS]);
	put_line(s, (std::string(name) + "_sum").c_str(), "", h.sum_s);
	put_line(s, (std::string(name) + "_count").c_str(), "", (double)h.count);
}

}  // namespace

extern "C" {

int gbm_zstd_encode(const uint8_t *data, size_t len, int level, uint8_t *out, size_t cap, size_t *len_out)
{
	if ((!data && len) || (!out && cap) || !len_out)
		return fail(GBM_E_INVALID_ARG, "NULL argument");
	try {
		static const uint8_t kEmpty = 0;
		std::vector<uint8_t> frame;
		if (!zstd().encode(data ? data : &kEmpty, len, level, frame))
			return fail(GBM_E_IO, zstd().ok ? "zstd encoder error" : "libzstd.so.1 is not available");
		*len_out = frame.size();
		if (frame.size() > cap)
			return fail(GBM_E_BUFFER_TOO_SMALL, "the frame does not fit the buffer");
		std::memcpy(out, frame.data(), frame.size());
	} catch (const std::exception &e) {
		return fail(GBM_E_IO, std::string("gbm_zstd_encode: ") + e.what());
	}
	return GBM_OK;
}

int gbm_zstd_decode(const uint8_t *frame, size_t len, uint8_t *out, size_t cap, size_t *len_out)
{
	if (!frame || (!out && cap) || !len_out)
		return fail(GBM_E_INVALID_ARG, "NULL argument");
	try {
		if (!zstd().ok)
			return fail(GBM_E_IO, "libzstd.so.1 is not available");
		std::vector<uint8_t> plain;
		if (!zstd().decode(frame, len, cap, plain)) {
			// larger than the caller's buffer, or not a frame that decodes cleanly (block.rs:78-83)
			if (cap < kMaxDecompressed && zstd().decode(frame, len, kMaxDecompressed, plain)) {
				*len_out = plain.size();
				return fail(GBM_E_BUFFER_TOO_SMALL, "the decoded bytes do not fit the buffer");
			}
			return fail(GBM_E_CORRUPT_DATA, "zstd frame does not decode (content checksum or framing)");
		}
		*len_out = plain.size();
		if (!plain.empty())
			std::memcpy(out, plain.data(), plain.size());
	} catch (const std::exception &e) {
		return fail(GBM_E_IO, std::string("gbm_zstd_decode: ") + e.what());
	}
	return GBM_OK;
}

const double *gbm_histogram_bounds(void) { return gbmimpl::Histogram::bounds(); }

int gbm_block_metrics_get(const gbm_manager *m, gbm_batcher *b, gbm_block_metrics *out)
{
	if (!m || !out)
		return fail(GBM_E_INVALID_ARG, "NULL argument");
	collect(m, b, *out);
	return GBM_OK;
}

int gbm_metrics_prometheus(const gbm_manager *m, gbm_batcher *b, char *buf, size_t cap, size_t *len_out)
{
	if (!m || !len_out || (!buf && cap))
		return fail(GBM_E_INVALID_ARG, "NULL argument");
	try {
		gbm_block_metrics x;
		collect(m, b, x);
		std::string s;
		s.reserve(16384);
		// the descriptions are the reference's (src/block/metrics.rs:40-143)
		put_metric(s, "block_compression_level", "gauge", "Garage compression level for node", (double)x.compression_level);
		put_metric(s, "block_rc_size", "gauge", "Number of blocks known to the reference counter", (double)x.rc_size);
		put_metric(s, "block_resync_queue_length", "gauge", "Number of block hashes queued for local check and possible resync",
			   (double)x.resync_queue_length);
		put_metric(s, "block_resync_errored_blocks", "gauge", "Number of block hashes whose last resync resulted in an error",
			   (double)x.resync_errored_blocks);
		if (b)
			put_metric(s, "block_ram_buffer_free_kb", "gauge",
				   "Available RAM in KiB to use for buffering data blocks to be written to remote nodes", (double)x.ram_buffer_free_kb);
		put_metric(s, "block_resync_counter", "counter", "Number of calls to resync_block", (double)x.resync_counter);
		put_metric(s, "block_resync_error_counter", "counter", "Number of calls to resync_block that returned an error",
			   (double)x.resync_error_counter);
		put_hist(s, "block_resync_duration", "Duration of resync_block operations", x.resync_duration);
		put_metric(s, "block_resync_send_counter", "counter", "Number of blocks sent to another node in resync operations",
			   (double)x.resync_send_counter);
		put_metric(s, "block_resync_recv_counter", "counter", "Number of blocks received from other nodes in resync operations",
			   (double)x.resync_recv_counter);
		put_metric(s, "block_bytes_read", "counter", "Number of bytes read from disk", (double)x.bytes_read);
		put_hist(s, "block_read_duration", "Duration of block read operations", x.block_read_duration);
		put_metric(s, "block_bytes_written", "counter", "Number of bytes written to disk", (double)x.bytes_written);
		put_hist(s, "block_write_duration", "Duration of block write operations", x.block_write_duration);
		put_metric(s, "block_delete_counter", "counter", "Number of blocks deleted", (double)x.delete_counter);
		put_metric(s, "block_corruption_counter", "counter", "Data corruptions detected on block reads", (double)x.corruption_counter);
		// this engine's own
		put_metric(s, "block_ec_reconstructs", "counter", "Blocks that went through a decode, on a read or in a resync pass", (double)x.ec_reconstructs);
		put_metric(s, "block_ec_blocks_put", "counter", "Blocks stored through rpc_put_block", (double)x.blocks_put);
		put_metric(s, "block_ec_blocks_get", "counter", "Blocks read through rpc_get_block", (double)x.blocks_get);
		put_metric(s, "block_ec_device_hashed", "counter", "Messages (blocks, shards) whose checksum the device computed", (double)x.gpu_hashed);
		put_metric(s, "block_ec_unconfirmed_verdicts", "counter",
			   "Checksum mismatches reported by a device trip or the pool that the host's own check did not confirm (the shard stayed)",
			   (double)x.unconfirmed_verdicts);
		put_metric(s, "block_ec_put_spot_checks", "counter", "Put trips of which one device-computed shard checksum was re-computed on the host",
			   (double)x.put_spot_checks);
		put_metric(s, "block_ec_put_spot_check_failures", "counter", "Put trips refused because the host did not get the device's checksum",
			   (double)x.put_spot_check_failures);
		put_metric(s, "block_ec_hedged_reads", "counter", "Extra shard requests sent by hedged reads", (double)x.hedged_reads);
		put_metric(s, "block_ec_scrub_corruptions_detected", "counter", "Corrupt blocks found by the scrub", (double)x.scrub_corruptions_detected);
		put_metric(s, "block_ec_scrub_time_last_complete_ms", "gauge", "When the last complete scrub ended (ms since the epoch)",
			   (double)x.scrub_time_last_complete_ms);
		put_metric(s, "block_ec_tranquilized_ms", "counter", "Time the maintenance workers slept for the tranquilizer", (double)x.tranquilized_ms);
		if (b) {
			put_metric(s, "block_ec_batcher_put_batches", "counter", "Device batches the coalescing queue formed out of puts",
				   (double)x.batcher_put_batches);
			put_metric(s, "block_ec_batcher_put_blocks", "counter", "Blocks put through the coalescing queue", (double)x.batcher_put_blocks);
			put_metric(s, "block_ec_batcher_get_batches", "counter", "Device batches the coalescing queue formed out of gets",
				   (double)x.batcher_get_batches);
			put_metric(s, "block_ec_batcher_get_blocks", "counter", "Blocks read through the coalescing queue", (double)x.batcher_get_blocks);
		}
		put_metric(s, "block_ec_devices", "gauge", "Devices this manager encodes on", (double)x.devices);
		if (m->is_front()) {  // each device's share, under names of their own (a metric's samples must stay in one group)
			static const char *const names[6] = {"block_ec_device_bytes_written", "block_ec_device_bytes_read",
							     "block_ec_device_corruption_counter", "block_ec_device_reconstructs",
							     "block_ec_device_blocks_put", "block_ec_device_blocks_get"};
			for (int j = 0; j < 6; ++j) {
				s += std::string("# HELP ") + names[j] + " Per device (gec_device_of_hash): see the total of the same name\n# TYPE " +
				     names[j] + " counter\n";
				for (size_t d = 0; d < m->lanes.size(); ++d) {
					char lab[32];
					std::snprintf(lab, sizeof(lab), "{device=\"%zu\"}", d);
					put_line(s, names[j], lab, (double)m->lanes[d]->metrics[j].load());
				}
			}
		}
		*len_out = s.size();
		if (cap) {
			const size_t n = std::min(cap, s.size());
			std::memcpy(buf, s.data(), n);
			if (n < cap)
				buf[n] = 0;
		}
		if (s.size() > cap)
			return fail(GBM_E_BUFFER_TOO_SMALL, "metrics text does not fit the buffer");
		return GBM_OK;
	} catch (const std::exception &e) {
		return fail(GBM_E_IO, std::string("gbm_metrics_prometheus: ") + e.what());
	}
}

}  // extern "C"
