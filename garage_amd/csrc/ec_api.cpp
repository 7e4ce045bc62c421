// ec_api.cpp -- the C ABI of libgarage_ec.so (include/garage_ec.h): everything that does not depend on which
// backend moves the bytes -- errors, shard geometry, coding matrices, the decode-plan cache, codec life cycle,
// argument checking -- and the dispatch to the codec's backend (ec_cpu.cpp / ec_hip_*.cpp).  No HIP in this file.
#include "ec_internal.hpp"

#include <pthread.h>

#include <algorithm>
#include <new>

#include "ec_env.hpp"
#include "mlh64_host.hpp"
#include "numa.hpp"

namespace gecimpl {

namespace {
thread_local std::string g_last_error;
}

int fail(int code, const std::string &detail)
{
	g_last_error = detail;
	return code;
}

int on_exception() noexcept
{
	try {
		throw;
	} catch (const std::bad_alloc &) {
		try {
			return fail(GEC_E_NOMEM, "out of host memory");
		} catch (...) {
			return GEC_E_NOMEM;
		}
	} catch (const std::exception &e) {
		try {
			return fail(GEC_E_DEVICE, std::string("internal error: ") + e.what());
		} catch (...) {
			return GEC_E_DEVICE;
		}
	} catch (...) {
		return GEC_E_DEVICE;
	}
}

int check_km(int k, int m)
{
	if (k <= 0)
		return fail(GEC_E_TOO_FEW_DATA, "data shards must be >= 1");
	if (m <= 0)
		return fail(GEC_E_TOO_FEW_PARITY, "parity shards must be >= 1");
	if (k + m > GEC_MAX_SHARDS)
		return fail(GEC_E_TOO_MANY_SHARDS, "k + m must be <= 256 in GF(2^8)");
	return GEC_OK;
}

void *Backend::host_alloc(size_t bytes) const { return gec_host_alloc(bytes); }

// ------------------------------------------------------------------ ForkJoinPool
ForkJoinPool::ForkJoinPool(unsigned n, const std::vector<int> &cpus)
{
	for (unsigned i = 0; i < n; ++i)
		workers_.emplace_back([this, cpus] {
#ifdef __linux__
			(void)pthread_setname_np(pthread_self(), "gec-pool");  // top -H, perf, tools/cpu_where.py
#endif
			(void)gecnuma::bind_this_thread(cpus);
			run();
		});
}

ForkJoinPool::~ForkJoinPool()
{
	{
		std::lock_guard<std::mutex> g(mu_);
		stop_ = true;
	}
	cv_.notify_all();
	for (auto &t : workers_)
		t.join();
}

void ForkJoinPool::parallel_for(size_t n, const std::function<void(size_t)> &fn)
{
	if (n == 0)
		return;
	if (workers_.empty() || n == 1) {
		for (size_t i = 0; i < n; ++i)
			fn(i);
		return;
	}
	std::unique_lock<std::mutex> call_lock(call_mu_);  // one parallel_for at a time
	{
		std::lock_guard<std::mutex> g(mu_);
		fn_ = &fn;
		n_ = n;
		// items are handed out in runs: an item may be as small as a 16 KiB column range (~10 us), and a lock per
		// item would serialise the threads on the lock
		grab_ = std::max<size_t>(1, n / (4 * (workers_.size() + 1)));
		next_.store(0, std::memory_order_relaxed);
		pending_ = n;
		failed_.store(false, std::memory_order_relaxed);
		err_ = nullptr;
		++epoch_;
	}
	cv_.notify_all();
	work();
	std::unique_lock<std::mutex> g(mu_);
	done_cv_.wait(g, [this] { return pending_ == 0 && active_ == 0; });
	fn_ = nullptr;
	// an item threw (bad_alloc, as a rule): what had not started was skipped, every thread has left fn -- whose captures
	// live in the caller's frame -- and the first exception goes on from here, on the caller's thread, to the C ABI's catch
	if (err_) {
		std::exception_ptr e = err_;
		err_ = nullptr;
		g.unlock();
		std::rethrow_exception(e);
	}
}

void ForkJoinPool::work()
{
	const std::function<void(size_t)> *fn;
	size_t n, grab;
	{
		std::lock_guard<std::mutex> g(mu_);
		if (!fn_)
			return;
		fn = fn_;
		n = n_;
		grab = grab_;
		++active_;  // parallel_for does not return (and fn stays alive) while a registered thread is in here
	}
	size_t done = 0;
	std::exception_ptr err;
	for (;;) {
		const size_t i0 = next_.fetch_add(grab, std::memory_order_relaxed);
		if (i0 >= n)
			break;
		const size_t i1 = std::min(n, i0 + grab);
		if (!failed_.load(std::memory_order_relaxed)) {
			try {
				for (size_t i = i0; i < i1; ++i)
					(*fn)(i);
			} catch (...) {
				err = std::current_exception();
				failed_.store(true, std::memory_order_relaxed);
			}
		}
		done += i1 - i0;  // (items skipped after a failure count as handed out)
	}
	std::lock_guard<std::mutex> g(mu_);
	if (err && !err_)
		err_ = err;
	pending_ -= done;
	--active_;
	if (pending_ == 0 && active_ == 0)
		done_cv_.notify_all();
}

void ForkJoinPool::run()
{
	uint64_t seen = 0;
	for (;;) {
		{
			std::unique_lock<std::mutex> g(mu_);
			cv_.wait(g, [&] { return stop_ || epoch_ != seen; });
			if (stop_)
				return;
			seen = epoch_;
		}
		work();
	}
}

// ------------------------------------------------------------------ decode plans
int get_plan(const gec_codec *c, const uint8_t *present, bool data_only, std::shared_ptr<const Plan> &out)
{
	const int k = c->k, n = c->k + c->m;
	std::string key(reinterpret_cast<const char *>(present), n);
	for (auto &ch : key)
		ch = ch ? 1 : 0;
	key.push_back(data_only ? 1 : 0);
	{
		std::lock_guard<std::mutex> g(c->cache_mu);
		auto it = c->cache.find(key);
		if (it != c->cache.end()) {
			c->lru.splice(c->lru.begin(), c->lru, it->second.second);
			out = it->second.first;
			return GEC_OK;
		}
	}
	auto plan = std::make_shared<Plan>();
	for (int j = 0; j < n && (int)plan->valid.size() < k; ++j)
		if (present[j])
			plan->valid.push_back(j);
	if ((int)plan->valid.size() < k)
		return fail(GEC_E_TOO_FEW_PRESENT, "fewer than k shards present");
	for (int j = 0; j < n; ++j)
		if (!present[j] && !(data_only && j >= k))
			plan->missing.push_back(j);
	if (!plan->missing.empty()) {
		gec::Matrix sub(k, k), dec;
		for (int t = 0; t < k; ++t)
			std::memcpy(&sub.at(t, 0), c->enc.row(plan->valid[t]), k);
		if (!gec::invert(sub, dec))
			return fail(GEC_E_INVALID_ARG, "decode sub-matrix singular (cannot happen for an MDS code)");
		// missing data j: row j of dec.  missing parity p: enc[p] * dec, which
		// equals re-encoding p from the completed data (crate order) because GF
		// arithmetic is exact.
		plan->rows = gec::Matrix((int)plan->missing.size(), k);
		for (size_t r = 0; r < plan->missing.size(); ++r) {
			int j = plan->missing[r];
			if (j < k) {
				std::memcpy(&plan->rows.at((int)r, 0), dec.row(j), k);
			} else {
				gec::Matrix prow(1, k);
				std::memcpy(&prow.at(0, 0), c->enc.row(j), k);
				gec::Matrix comp = gec::matmul(prow, dec);
				std::memcpy(&plan->rows.at((int)r, 0), comp.row(0), k);
			}
		}
	}
	{
		std::lock_guard<std::mutex> g(c->cache_mu);
		++c->inversions;
		if (c->cache.find(key) == c->cache.end()) {
			c->lru.push_front(key);
			c->cache[key] = {plan, c->lru.begin()};
			if (c->cache.size() > gec_codec::kCacheCap) {
				c->cache.erase(c->lru.back());
				c->lru.pop_back();
			}
		}
	}
	out = plan;
	return GEC_OK;
}

// ------------------------------------------------------------------ Backend defaults: no device entry points
namespace {
int no_dev() { return fail(GEC_E_DEVICE, "this codec runs on the host cores (GEC_BACKEND_CPU): it has no device-resident entry points"); }
}
int Backend::encode_batch_dev(size_t, const void *, size_t, size_t, void *, size_t, void *) { return no_dev(); }
int Backend::verify_batch_dev(size_t, const void *, size_t, size_t, uint32_t *, void *) { return no_dev(); }
int Backend::reconstruct_dev(size_t, void *, size_t, const size_t *, size_t, const uint8_t *, int, size_t, size_t, void *) { return no_dev(); }
int Backend::reconstruct_dev_ex(size_t, void *, size_t, size_t, const uint8_t *, int, void *) { return no_dev(); }
int Backend::hash_batch_dev(size_t, const void *, size_t, size_t, void *, void *, bool) { return no_dev(); }
int Backend::encode_hash_batch_dev(size_t, void *, size_t, size_t, void *, void *) { return no_dev(); }

namespace {

int build_matrix_kind(int k, int m, int matrix, gec::Matrix &enc)
{
	if (matrix == GEC_MATRIX_VANDERMONDE) {
		if (!gec::build_encoding_matrix(k, m, enc))
			return fail(GEC_E_INVALID_ARG, "vandermonde top block singular");
	} else if (matrix == GEC_MATRIX_CAUCHY) {
		gec::build_cauchy_matrix(k, m, enc);
	} else {
		return fail(GEC_E_INVALID_ARG, "unknown matrix family");
	}
	return GEC_OK;
}

int check_shard_size(size_t S)
{
	if (S == 0)
		return fail(GEC_E_EMPTY_SHARD, "shard length is 0");
	if (S % 64)
		return fail(GEC_E_INCORRECT_SHARD_SIZE, "S must be a multiple of 64");
	return GEC_OK;
}

int check_dev_layout(const void *p, size_t stride, size_t S, size_t need)
{
	if (int rc = check_shard_size(S))
		return rc;
	if (!p)
		return fail(GEC_E_INVALID_ARG, "NULL device pointer");
	if (reinterpret_cast<uintptr_t>(p) % 16 != 0 || stride % 16 != 0)
		return fail(GEC_E_INVALID_ARG, "device pointer/stride must be 16-byte aligned");
	if (stride < need)
		return fail(GEC_E_INCORRECT_SHARD_SIZE, "stride smaller than the shards it must hold");
	return GEC_OK;
}

int create_codec(int k, int m, int backend, int device, int matrix, int qos_class, gec_codec **out, int sumkind = GEC_SHARDSUM_DEFAULT)
{
	if (!out)
		return fail(GEC_E_INVALID_ARG, "NULL out");
	*out = nullptr;
	int rc = check_km(k, m);  // argument errors are reported before any device is touched
	if (rc)
		return rc;
	if (sumkind == GEC_SHARDSUM_DEFAULT)
		sumkind = GEC_SHARDSUM_MLH64;
	if (sumkind != GEC_SHARDSUM_BLAKE2B_TREE && sumkind != GEC_SHARDSUM_MLH64)
		return fail(GEC_E_INVALID_ARG, "unknown shard checksum kind");
	if (matrix != GEC_MATRIX_VANDERMONDE && matrix != GEC_MATRIX_CAUCHY)
		return fail(GEC_E_INVALID_ARG, "unknown matrix family");
	if (backend != GEC_BACKEND_CPU && backend != GEC_BACKEND_HIP && backend != GEC_BACKEND_AUTO)
		return fail(GEC_E_INVALID_ARG, "unknown backend");
	if (backend == GEC_BACKEND_AUTO)  // the GPU when there is one, the host cores when there is none (or it is gone)
		backend = hip_device_count() > 0 ? GEC_BACKEND_HIP : GEC_BACKEND_CPU;
	std::unique_ptr<gec_codec> c(new (std::nothrow) gec_codec());
	if (!c)
		return fail(GEC_E_NOMEM, "alloc codec");
	c->k = k;
	c->m = m;
	c->backend = backend;
	c->matrix = matrix;
	c->qos_class = qos_class;
	c->sumkind = sumkind;
	rc = build_matrix_kind(k, m, matrix, c->enc);
	if (rc)
		return rc;
	if (backend == GEC_BACKEND_CPU) {
		c->device = -1;
		rc = make_cpu_backend(c.get(), c->be);
	} else {
		c->device = device;
		rc = make_hip_backend(c.get(), device, c->be);
	}
	if (rc)
		return rc;
	*out = c.release();
	return GEC_OK;
}

}  // namespace
}  // namespace gecimpl

using namespace gecimpl;

// ---------------------------------------------------------------------------
static thread_local gec_link_release_fn t_release_fn = nullptr;
static thread_local void *t_release_arg = nullptr;
namespace gecimpl {
void link_release_fire()
{
	if (gec_link_release_fn fn = t_release_fn) {
		t_release_fn = nullptr;
		fn(t_release_arg);
	}
}
bool link_release_armed() { return t_release_fn != nullptr; }
}  // namespace gecimpl
namespace {
struct LinkReleaseScope {
	~LinkReleaseScope() { gecimpl::link_release_fire(); }
};
}  // namespace
extern "C" {

uint32_t gec_version(void) { return GEC_VERSION; }

int gec_device_count(void) { return hip_device_count(); }

int gec_device_of_hash(const uint8_t hash[32], int ndev) { return hash && ndev >= 1 ? hash[4] % ndev : -1; }

const char *gec_strerror(int code)
{
	switch (code) {
	case GEC_OK: return "ok";
	case GEC_E_TOO_FEW_SHARDS: return "too few shards";
	case GEC_E_TOO_MANY_SHARDS: return "too many shards";
	case GEC_E_TOO_FEW_DATA: return "too few data shards";
	case GEC_E_TOO_MANY_DATA: return "too many data shards";
	case GEC_E_TOO_FEW_PARITY: return "too few parity shards";
	case GEC_E_TOO_MANY_PARITY: return "too many parity shards";
	case GEC_E_INCORRECT_SHARD_SIZE: return "incorrect shard size";
	case GEC_E_TOO_FEW_PRESENT: return "too few shards present";
	case GEC_E_EMPTY_SHARD: return "empty shard";
	case GEC_E_INVALID_INDEX: return "invalid index";
	case GEC_E_DEVICE: return "device (HIP) error";
	case GEC_E_NOMEM: return "out of memory";
	case GEC_E_INVALID_ARG: return "invalid argument";
	default: return "unknown error";
	}
}

const char *gec_last_error(void) { return g_last_error.c_str(); }

void gec_thread_link_release(gec_link_release_fn fn, void *arg)
{
	t_release_fn = fn;
	t_release_arg = arg;
}

const char *gec_env_table(void) { return env_table_text(); }

size_t gec_shard_len(int k, size_t block_len)
{
	if (k <= 0)
		return 0;
	size_t per = (std::max<size_t>(block_len, 1) + (size_t)k - 1) / (size_t)k;
	return (per + 63) / 64 * 64;
}

int gec_build_matrix_ex(int k, int m, int matrix, uint8_t *out)
try {
	int rc = check_km(k, m);
	if (rc)
		return rc;
	if (!out)
		return fail(GEC_E_INVALID_ARG, "NULL output");
	gec::Matrix enc;
	rc = build_matrix_kind(k, m, matrix, enc);
	if (rc)
		return rc;
	std::memcpy(out, enc.v.data(), enc.v.size());
	return GEC_OK;
}
GEC_CATCH

int gec_build_matrix(int k, int m, uint8_t *out) { return gec_build_matrix_ex(k, m, GEC_MATRIX_VANDERMONDE, out); }

int gec_build_decode_matrix(int k, int m, const uint8_t *present, int32_t *valid_out, uint8_t *out)
try {
	int rc = check_km(k, m);
	if (rc)
		return rc;
	if (!present || !valid_out || !out)
		return fail(GEC_E_INVALID_ARG, "NULL argument");
	gec::Matrix enc;
	if (!gec::build_encoding_matrix(k, m, enc))
		return fail(GEC_E_INVALID_ARG, "vandermonde top block singular");
	int nv = 0;
	for (int j = 0; j < k + m && nv < k; ++j)
		if (present[j])
			valid_out[nv++] = j;
	if (nv < k)
		return fail(GEC_E_TOO_FEW_PRESENT, "fewer than k shards present");
	gec::Matrix sub(k, k), dec;
	for (int t = 0; t < k; ++t)
		std::memcpy(&sub.at(t, 0), enc.row(valid_out[t]), k);
	if (!gec::invert(sub, dec))
		return fail(GEC_E_INVALID_ARG, "decode sub-matrix singular");
	std::memcpy(out, dec.v.data(), dec.v.size());
	return GEC_OK;
}
GEC_CATCH

// ------------------------------------------------------------------ codec
int gec_codec_create(int k, int m, int backend, int device, gec_codec **out)
try {
	return create_codec(k, m, backend, device, GEC_MATRIX_VANDERMONDE, GEC_CLASS_FOREGROUND, out);
}
GEC_CATCH

int gec_codec_create_ex(int k, int m, int backend, int device, int matrix, gec_codec **out)
try {
	return create_codec(k, m, backend, device, matrix, GEC_CLASS_FOREGROUND, out);
}
GEC_CATCH

int gec_codec_create_ex2(int k, int m, int backend, int device, int matrix, int shardsum, gec_codec **out)
try {
	return create_codec(k, m, backend, device, matrix, GEC_CLASS_FOREGROUND, out, shardsum);
}
GEC_CATCH

int gec_codec_background(const gec_codec *c, gec_codec **out)
try {
	if (!c)
		return fail(GEC_E_INVALID_ARG, "NULL codec");
	return create_codec(c->k, c->m, c->backend, c->device, c->matrix, GEC_CLASS_BACKGROUND, out, c->sumkind);
}
GEC_CATCH

// a sibling of `c` (same code, backend, device and class) that produces the other kind of shard checksum
int gec_codec_with_shardsum(const gec_codec *c, int shardsum, gec_codec **out)
try {
	if (!c)
		return fail(GEC_E_INVALID_ARG, "NULL codec");
	return create_codec(c->k, c->m, c->backend, c->device, c->matrix, c->qos_class, out, shardsum);
}
GEC_CATCH

int gec_codec_shardsum(const gec_codec *c) { return c ? c->sumkind : -1; }

int gec_shardsum_host(int shardsum, const uint8_t *data, size_t len, uint8_t out[32])
try {
	if ((!data && len) || !out)
		return fail(GEC_E_INVALID_ARG, "NULL argument");
	if (shardsum == GEC_SHARDSUM_MLH64)
		mlh::shardsum3(data, len, out);
	else if (shardsum == GEC_SHARDSUM_BLAKE2B_TREE)
		b2host::shardsum(data, len, out);
	else
		return fail(GEC_E_INVALID_ARG, "unknown shard checksum kind");
	return GEC_OK;
}
GEC_CATCH

void gec_codec_destroy(gec_codec *c) { delete c; }

int gec_codec_k(const gec_codec *c) { return c ? c->k : 0; }
int gec_codec_m(const gec_codec *c) { return c ? c->m : 0; }
int gec_codec_device(const gec_codec *c) { return c ? c->device : -1; }
int gec_codec_backend(const gec_codec *c) { return c ? c->backend : -1; }
int gec_codec_class(const gec_codec *c) { return c ? c->qos_class : -1; }

int gec_codec_numa_node(const gec_codec *c) { return c && c->be ? c->be->numa_node() : -1; }

int gec_codec_numa_cpus(const gec_codec *c, size_t cap, int *cpus, size_t *count)
{
	if (!c || !count || (cap && !cpus))
		return fail(GEC_E_INVALID_ARG, "NULL argument");
	const std::vector<int> *v = c->be ? c->be->numa_cpus() : nullptr;
	*count = v ? v->size() : 0;
	for (size_t i = 0; v && i < v->size() && i < cap; ++i)
		cpus[i] = (*v)[i];
	return GEC_OK;
}

void *gec_host_alloc_near(const gec_codec *c, size_t bytes)
{
	if (!c || !c->be)
		return gec_host_alloc(bytes);
	try {
		return c->be->host_alloc(bytes);
	} catch (...) {
		(void)on_exception();
		return nullptr;
	}
}

int gec_numa_node_of(const void *p) { return p ? gecnuma::node_of_address(p) : -1; }

int gec_numa_bind_thread(const gec_codec *c)
{
	const std::vector<int> *v = c && c->be ? c->be->numa_cpus() : nullptr;
	return v && gecnuma::bind_this_thread(*v) ? 1 : 0;
}

int gec_parity_matrix(const gec_codec *c, uint8_t *out)
try {
	if (!c || !out)
		return fail(GEC_E_INVALID_ARG, "NULL argument");
	std::memcpy(out, c->enc.row(c->k), (size_t)c->m * c->k);
	return GEC_OK;
}
GEC_CATCH

int gec_codec_cache_stats(const gec_codec *c, uint64_t *cached, uint64_t *inversions)
try {
	if (!c)
		return fail(GEC_E_INVALID_ARG, "NULL codec");
	std::lock_guard<std::mutex> g(c->cache_mu);
	if (cached)
		*cached = c->cache.size();
	if (inversions)
		*inversions = c->inversions;
	return GEC_OK;
}
GEC_CATCH

// ------------------------------------------------------------------ host-pointer entry points
static int encode_common(const gec_codec *c, size_t nblocks, const uint8_t *const *blocks, const size_t *block_len, size_t S,
			 uint8_t *const *parity, uint8_t *shard_sums)
{
	if (!c)
		return fail(GEC_E_INVALID_ARG, "NULL codec");
	if (nblocks == 0)
		return GEC_OK;
	if (!blocks || !block_len || !parity)
		return fail(GEC_E_INVALID_ARG, "NULL argument");
	if (int rc = check_shard_size(S))
		return rc;
	for (size_t b = 0; b < nblocks; ++b) {
		if (!blocks[b] || !parity[b])
			return fail(GEC_E_INVALID_ARG, "NULL block/parity pointer");
		if (block_len[b] > (size_t)c->k * S)
			return fail(GEC_E_INCORRECT_SHARD_SIZE, "block longer than k*S");
	}
	return c->be->encode_batch(nblocks, blocks, block_len, S, parity, shard_sums);
}

int gec_encode_batch(const gec_codec *c, size_t nblocks, const uint8_t *const *blocks, const size_t *block_len, size_t S,
		     uint8_t *const *parity)
try {
	return encode_common(c, nblocks, blocks, block_len, S, parity, nullptr);
}
GEC_CATCH

int gec_encode_hash_batch(const gec_codec *c, size_t nblocks, const uint8_t *const *blocks, const size_t *block_len, size_t S,
			  uint8_t *const *parity, uint8_t *shard_sums)
try {
	LinkReleaseScope release;
	if (!shard_sums)
		return fail(GEC_E_INVALID_ARG, "NULL shard_sums");
	return encode_common(c, nblocks, blocks, block_len, S, parity, shard_sums);
}
GEC_CATCH

static int verify_common(const gec_codec *c, size_t nblocks, const uint8_t *const *shards, size_t S, uint8_t *ok, uint8_t *shard_sums,
			 bool want_sums)
{
	if (!c)
		return fail(GEC_E_INVALID_ARG, "NULL codec");
	if (nblocks == 0)
		return GEC_OK;
	if (!shards || !ok || (want_sums && !shard_sums))
		return fail(GEC_E_INVALID_ARG, "NULL argument");
	if (int rc = check_shard_size(S))
		return rc;
	const size_t n = (size_t)c->k + c->m;
	for (size_t i = 0; i < nblocks * n; ++i)
		if (!shards[i])
			return fail(GEC_E_TOO_FEW_SHARDS, "verify needs all k+m shards");
	return want_sums ? c->be->verify_hash_batch(nblocks, shards, S, ok, shard_sums) : c->be->verify_batch(nblocks, shards, S, ok);
}

int gec_verify_batch(const gec_codec *c, size_t nblocks, const uint8_t *const *shards, size_t S, uint8_t *ok)
try {
	return verify_common(c, nblocks, shards, S, ok, nullptr, false);
}
GEC_CATCH

int gec_verify_hash_batch(const gec_codec *c, size_t nblocks, const uint8_t *const *shards, size_t S, uint8_t *ok,
			  uint8_t *shard_sums)
try {
	return verify_common(c, nblocks, shards, S, ok, shard_sums, true);
}
GEC_CATCH

static int reconstruct_common(const gec_codec *c, size_t nblocks, const uint8_t *const *shards, uint8_t *const *out, size_t S,
			      int data_only, uint8_t *in_sums, uint8_t *out_sums)
{
	if (!c)
		return fail(GEC_E_INVALID_ARG, "NULL codec");
	if (nblocks == 0)
		return GEC_OK;
	if (!shards || !out)
		return fail(GEC_E_INVALID_ARG, "NULL argument");
	if (int rc = check_shard_size(S))
		return rc;
	const size_t k = c->k, n = (size_t)c->k + c->m;
	for (size_t b = 0; b < nblocks; ++b) {
		size_t np = 0;
		for (size_t j = 0; j < n; ++j)
			np += shards[b * n + j] != nullptr;
		if (np < k)
			return fail(GEC_E_TOO_FEW_PRESENT, "fewer than k shards present");
	}
	return c->be->reconstruct_batch(nblocks, shards, out, S, data_only, in_sums, out_sums);
}

int gec_reconstruct_batch(const gec_codec *c, size_t nblocks, const uint8_t *const *shards, uint8_t *const *out, size_t S,
			  int data_only)
try {
	return reconstruct_common(c, nblocks, shards, out, S, data_only, nullptr, nullptr);
}
GEC_CATCH

int gec_reconstruct_hash_batch(const gec_codec *c, size_t nblocks, const uint8_t *const *shards, uint8_t *const *out, size_t S,
			       int data_only, uint8_t *in_sums, uint8_t *out_sums)
try {
	if (!in_sums || !out_sums)
		return fail(GEC_E_INVALID_ARG, "NULL checksum output");
	return reconstruct_common(c, nblocks, shards, out, S, data_only, in_sums, out_sums);
}
GEC_CATCH

int gec_decode_verify_batch(const gec_codec *c, size_t nblocks, const uint8_t *const *shards, size_t S, const size_t *block_len,
			    uint8_t *const *rebuilt, uint8_t *shard_sums, uint8_t *block_sums)
try {
	LinkReleaseScope release;
	if (!c)
		return fail(GEC_E_INVALID_ARG, "NULL codec");
	if (nblocks == 0)
		return GEC_OK;
	if (!shards || !shard_sums || (block_sums && !block_len))
		return fail(GEC_E_INVALID_ARG, "NULL argument");
	if (int rc = check_shard_size(S))
		return rc;
	const size_t k = c->k, n = (size_t)c->k + c->m;
	for (size_t b = 0; b < nblocks; ++b) {
		size_t np = 0;
		for (size_t j = 0; j < n; ++j)
			np += shards[b * n + j] != nullptr;
		if (np < k)
			return fail(GEC_E_TOO_FEW_PRESENT, "fewer than k shards present");
		if (block_len && block_len[b] > k * S)
			return fail(GEC_E_INCORRECT_SHARD_SIZE, "block longer than k*S");
		// the data shards that will be rebuilt: missing ones (a missing data shard is always among the erased)
		for (size_t j = 0; j < k; ++j)
			if (!shards[b * n + j] && (!rebuilt || !rebuilt[b * n + j]))
				return fail(GEC_E_INVALID_ARG, "NULL output for a missing data shard");
	}
	return c->be->decode_verify_batch(nblocks, shards, S, block_len, rebuilt, shard_sums, block_sums);
}
GEC_CATCH

static int hash_common(const gec_codec *c, size_t n, const uint8_t *const *msgs, const size_t *lens, uint8_t *out, bool tree)
{
	if (!c)
		return fail(GEC_E_INVALID_ARG, "NULL codec");
	if (n == 0)
		return GEC_OK;
	if (!msgs || !lens || !out)
		return fail(GEC_E_INVALID_ARG, "NULL argument");
	for (size_t i = 0; i < n; ++i)
		if (!msgs[i] && lens[i])
			return fail(GEC_E_INVALID_ARG, "NULL message pointer");
	return c->be->hash_batch(n, msgs, lens, out, tree);
}

int gec_blake2sum_batch(const gec_codec *c, size_t n, const uint8_t *const *msgs, const size_t *lens, uint8_t *out)
try {
	return hash_common(c, n, msgs, lens, out, false);
}
GEC_CATCH

int gec_shardsum_batch(const gec_codec *c, size_t n, const uint8_t *const *msgs, const size_t *lens, uint8_t *out)
try {
	return hash_common(c, n, msgs, lens, out, true);
}
GEC_CATCH

// ------------------------------------------------------------------ device-resident entry points
int gec_encode_batch_dev(const gec_codec *c, size_t nblocks, const void *d_data, size_t data_stride, size_t S, void *d_parity,
			 size_t parity_stride, void *hip_stream)
try {
	if (!c)
		return fail(GEC_E_INVALID_ARG, "NULL codec");
	if (nblocks == 0)
		return GEC_OK;
	int rc = check_dev_layout(d_data, data_stride, S, (size_t)c->k * S);
	if (rc)
		return rc;
	rc = check_dev_layout(d_parity, parity_stride, S, (size_t)c->m * S);
	if (rc)
		return rc;
	return c->be->encode_batch_dev(nblocks, d_data, data_stride, S, d_parity, parity_stride, hip_stream);
}
GEC_CATCH

int gec_verify_batch_dev(const gec_codec *c, size_t nblocks, const void *d_stripes, size_t stride, size_t S, uint32_t *d_bad,
			 void *hip_stream)
try {
	if (!c || !d_bad)
		return fail(GEC_E_INVALID_ARG, "NULL argument");
	if (nblocks == 0)
		return GEC_OK;
	int rc = check_dev_layout(d_stripes, stride, S, (size_t)(c->k + c->m) * S);
	if (rc)
		return rc;
	return c->be->verify_batch_dev(nblocks, d_stripes, stride, S, d_bad, hip_stream);
}
GEC_CATCH

int gec_reconstruct_range_dev(const gec_codec *c, size_t nblocks, void *d_stripes, size_t stride, size_t S, const uint8_t *present,
			      int data_only, size_t byte_off, size_t byte_len, void *hip_stream)
try {
	if (!c || !present)
		return fail(GEC_E_INVALID_ARG, "NULL argument");
	if (nblocks == 0)
		return GEC_OK;
	int rc = check_dev_layout(d_stripes, stride, S, (size_t)(c->k + c->m) * S);
	if (rc)
		return rc;
	if (byte_off % 16 || byte_len % 16 || byte_off > S || byte_len > S - byte_off)
		return fail(GEC_E_INVALID_ARG, "byte range must be 16-byte aligned and inside the shard");
	return c->be->reconstruct_dev(nblocks, d_stripes, stride, nullptr, S, present, data_only, byte_off, byte_len, hip_stream);
}
GEC_CATCH

int gec_reconstruct_scattered_dev(const gec_codec *c, size_t nblocks, void *d_base, size_t block_stride, const size_t *shard_off,
				  size_t S, const uint8_t *present, int data_only, size_t byte_off, size_t byte_len, void *hip_stream)
try {
	if (!c || !present || !shard_off)
		return fail(GEC_E_INVALID_ARG, "NULL argument");
	if (nblocks == 0)
		return GEC_OK;
	int rc = check_dev_layout(d_base, block_stride, S, S);
	if (rc)
		return rc;
	for (int j = 0; j < c->k + c->m; ++j)
		if (shard_off[j] % 16)
			return fail(GEC_E_INVALID_ARG, "shard offsets must be multiples of 16");
	if (byte_off % 16 || byte_len % 16 || byte_off > S || byte_len > S - byte_off)
		return fail(GEC_E_INVALID_ARG, "byte range must be 16-byte aligned and inside the shard");
	return c->be->reconstruct_dev(nblocks, d_base, block_stride, shard_off, S, present, data_only, byte_off, byte_len, hip_stream);
}
GEC_CATCH

int gec_reconstruct_batch_dev(const gec_codec *c, size_t nblocks, void *d_stripes, size_t stride, size_t S, const uint8_t *present,
			      int data_only, void *hip_stream)
try {
	return gec_reconstruct_range_dev(c, nblocks, d_stripes, stride, S, present, data_only, 0, S, hip_stream);
}
GEC_CATCH

int gec_reconstruct_batch_dev_ex(const gec_codec *c, size_t nblocks, void *d_stripes, size_t stride, size_t S, const uint8_t *present,
				 int data_only, void *hip_stream)
try {
	if (!c || !present)
		return fail(GEC_E_INVALID_ARG, "NULL argument");
	if (nblocks == 0)
		return GEC_OK;
	int rc = check_dev_layout(d_stripes, stride, S, (size_t)(c->k + c->m) * S);
	if (rc)
		return rc;
	// every block's pattern is checked before anything is enqueued: a batch is rebuilt whole or not at all
	const size_t n = (size_t)c->k + c->m;
	for (size_t b = 0; b < nblocks; ++b) {
		int have = 0;
		for (size_t j = 0; j < n; ++j)
			have += present[b * n + j] ? 1 : 0;
		if (have < c->k)
			return fail(GEC_E_TOO_FEW_PRESENT, "block " + std::to_string(b) + ": fewer than k shards present");
	}
	return c->be->reconstruct_dev_ex(nblocks, d_stripes, stride, S, present, data_only, hip_stream);
}
GEC_CATCH

static int hash_dev_common(const gec_codec *c, size_t n, const void *d_base, size_t stride, size_t len, void *d_out, void *hip_stream,
			   bool tree)
{
	if (!c || (n && (!d_base || !d_out)))
		return fail(GEC_E_INVALID_ARG, "NULL argument");
	if (n == 0)
		return GEC_OK;
	if (reinterpret_cast<uintptr_t>(d_base) % 16 || stride % 16 || reinterpret_cast<uintptr_t>(d_out) % 16)
		return fail(GEC_E_INVALID_ARG, "device pointers/stride must be 16-byte aligned");
	if (n > 1 && stride < len)
		return fail(GEC_E_INVALID_ARG, "stride smaller than the message length");
	return c->be->hash_batch_dev(n, d_base, stride, len, d_out, hip_stream, tree);
}

int gec_blake2sum_batch_dev(const gec_codec *c, size_t n, const void *d_base, size_t stride, size_t len, void *d_out, void *hip_stream)
try {
	return hash_dev_common(c, n, d_base, stride, len, d_out, hip_stream, false);
}
GEC_CATCH

int gec_shardsum_batch_dev(const gec_codec *c, size_t n, const void *d_base, size_t stride, size_t len, void *d_out, void *hip_stream)
try {
	return hash_dev_common(c, n, d_base, stride, len, d_out, hip_stream, true);
}
GEC_CATCH

int gec_encode_hash_batch_dev(const gec_codec *c, size_t nblocks, void *d_stripes, size_t stride, size_t S, void *d_sums,
			      void *hip_stream)
try {
	if (!c || !d_sums)
		return fail(GEC_E_INVALID_ARG, "NULL argument");
	if (nblocks == 0)
		return GEC_OK;
	int rc = check_dev_layout(d_stripes, stride, S, (size_t)(c->k + c->m) * S);
	if (rc)
		return rc;
	if (reinterpret_cast<uintptr_t>(d_sums) % 16)
		return fail(GEC_E_INVALID_ARG, "d_sums must be 16-byte aligned");
	return c->be->encode_hash_batch_dev(nblocks, d_stripes, stride, S, d_sums, hip_stream);
}
GEC_CATCH

}  // extern "C"
