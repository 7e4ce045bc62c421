// ec_hip_staging.cpp -- the HIP backend's host-side resources: staging slots (streams, events, pinned + device
// buffers), the registry of pinned caller memory (gec_host_*), the quality-of-service gate between foreground and
// background codecs.  Host code only.
#include "ec_hip.hpp"
#include "kernel_args.hpp"

#include <algorithm>
#include <chrono>
#include <cstdlib>

namespace gecimpl {

// ------------------------------------------------------------------ pinned caller memory
void PinnedRanges::add(const void *p, size_t n, bool owned, intptr_t dev_delta, bool plain)
{
	std::lock_guard<std::mutex> g(mu_);
	ranges_[reinterpret_cast<uintptr_t>(p)] = {n, owned, plain, dev_delta};
}

bool PinnedRanges::remove(const void *p, bool &owned, bool &plain)
{
	std::lock_guard<std::mutex> g(mu_);
	auto it = ranges_.find(reinterpret_cast<uintptr_t>(p));
	if (it == ranges_.end())
		return false;
	owned = it->second.owned;
	plain = it->second.plain;
	ranges_.erase(it);
	return true;
}

bool PinnedRanges::contains(const void *p, size_t n, intptr_t *dev_delta) const
{
	if (!p)
		return false;
	const uintptr_t a = reinterpret_cast<uintptr_t>(p);
	std::lock_guard<std::mutex> g(mu_);
	if (ranges_.empty())
		return false;
	auto it = ranges_.upper_bound(a);
	if (it == ranges_.begin())
		return false;
	--it;
	if (it->second.plain || !(a >= it->first && a + n <= it->first + it->second.len))
		return false;
	if (dev_delta)
		*dev_delta = it->second.dev_delta;
	return true;
}

PinnedRanges &pinned()
{
	static PinnedRanges r;
	return r;
}

namespace {
std::atomic<int> g_cu_masks{-1};  // gec_cu_masks_active: -1 not tried yet, 0 refused by the runtime, 1 masks, 2 masks + class partition
// BACKGROUND-class codecs alive per device: the class partition of the CUs is only applied while there is one -- a
// process that never scrubs or resyncs keeps the whole chip for the request path's checksum kernels
std::atomic<int> g_bg_codecs[64];
}

void background_codec_born(int device)
{
	if (device >= 0 && device < 64)
		g_bg_codecs[device].fetch_add(1);
}
void background_codec_gone(int device)
{
	if (device >= 0 && device < 64)
		g_bg_codecs[device].fetch_sub(1);
}
static bool background_class_present(int device) { return device >= 0 && device < 64 && g_bg_codecs[device].load() > 0; }

// ------------------------------------------------------------------ staging slots
// Every stream of a slot is created here, so that the codec's class decides priority and CU mask in one place: a
// background codec's streams are confined to qos.compute_cus CUs (the LAST ones of the chip -- the link kernels'
// GEC_UPLOAD_CUS are the first) or, where the runtime has no CU masks, at least run at the lowest priority.
int Staging::make_stream(hipStream_t *s)
{
	if (qos.background && qos.compute_cus > 0 && qos.num_cu > qos.compute_cus) {
		const int words = (qos.num_cu + 31) / 32;
		std::vector<uint32_t> mask(words, 0);
		for (int i = qos.num_cu - qos.compute_cus; i < qos.num_cu; ++i)
			mask[i / 32] |= 1u << (i % 32);
		if (hipExtStreamCreateWithCUMask(s, (uint32_t)words, mask.data()) == hipSuccess)
			return GEC_OK;
		(void)hipGetLastError();  // no CU masks in this runtime / partition mode: priority alone
		*s = nullptr;
	}
	if (qos.background) {
		if (hipStreamCreateWithPriority(s, hipStreamNonBlocking, qos.stream_priority) == hipSuccess)
			return GEC_OK;
		(void)hipGetLastError();
	}
	HIP_TRY(hipStreamCreateWithFlags(s, hipStreamNonBlocking));
	return GEC_OK;
}

int Staging::ensure_segments(int num_cu)
{
	// the masks follow the device's class situation: created for "no background class here" (the request path's checksum
	// kernels get every CU the link kernels do not use), they are made again the first time the slot is used after a
	// BACKGROUND codec appeared on the device (the slot is on loan to one call: nothing of it is in flight here)
	const bool want_split = qos.background || background_class_present(qos.device);
	if (stream3 && seg_split == want_split)
		return GEC_OK;
	if (stream3) {
		for (hipStream_t *s : {&stream_up, &stream_chain, &stream_down})
			if (*s) {
				(void)hipStreamSynchronize(*s);
				(void)hipStreamDestroy(*s);
				*s = nullptr;
			}
		cus_up = cus_chain = cus_down = 0;
	} else {
		for (int i = 0; i < kMaxSeg; ++i) {
			HIP_TRY(hipEventCreateWithFlags(&ev_seg[i], hipEventDisableTiming));
			HIP_TRY(hipEventCreateWithFlags(&ev_dec[i], hipEventDisableTiming));
		}
	}
	seg_split = want_split;
	const int up_cus = env().upload_cus;  // 0 = no CU masks (A/B)
	if (up_cus > 0 && up_cus < num_cu) {
		// The chip is partitioned once, the same way for every codec of the process (CuPlan), so that the classes
		// never share a CU:
		//   [0, U)            foreground kernels that READ host memory   (the link needs very little in flight)
		//   [U, 2U)           foreground kernels that WRITE host memory  (sixteen CUs cannot do both at link rate)
		//   [2U, 2U + U/2)    background kernels that touch host memory, either way
		//   [2U + U/2, N - B) foreground checksum kernels
		//   [N - B, N)        everything else a background codec launches (B = GEC_BG_CUS)
		// A kernel whose loads share a CU with microsecond-long host accesses crawls, and a foreground kernel is only
		// done when its slowest workgroup is: a scrub's long checksum launch on the CUs a PutObject's small one lands on
		// was what its p99 was made of (profiles/r03_qos.txt).  Without a usable partition (tiny device, B = 0) the
		// background class falls back to sharing the foreground's CUs at the lowest stream priority.
		const int words = (num_cu + 31) / 32;
		const int U = up_cus, B = want_split ? qos.compute_cus_plan : 0;
		const bool split = B > 0 && 2 * U + U / 2 + B + 16 <= num_cu && U >= 2;
		auto mask = [&](int lo, int hi) {
			std::vector<uint32_t> m(words, 0);
			for (int i = lo; i < hi && i < num_cu; ++i)
				m[i / 32] |= 1u << (i % 32);
			return m;
		};
		std::vector<uint32_t> up, down, rest;
		if (split && qos.background) {
			up = mask(2 * U, 2 * U + U / 2);
			rest = mask(num_cu - B, num_cu);
		} else if (split) {
			up = mask(0, U);
			down = mask(U, 2 * U);
			rest = mask(2 * U + U / 2, num_cu - B);
		} else {
			const int down_hi = 2 * U < num_cu ? 2 * U : U;
			up = mask(0, U);
			if (down_hi > U)
				down = mask(U, down_hi);
			rest = mask(down_hi, num_cu);
			// (no partition: a background codec's checksum kernels still keep to their own CUs, the chip's last ones)
			if (qos.background && qos.compute_cus > 0 && qos.compute_cus < num_cu - down_hi)
				rest = mask(num_cu - qos.compute_cus, num_cu);
		}
		// (a runtime or partition mode without CU masks is not an error: the paths then share all CUs, slower)
		if (hipExtStreamCreateWithCUMask(&stream_up, (uint32_t)words, up.data()) != hipSuccess)
			stream_up = nullptr;
		if (!stream_up || hipExtStreamCreateWithCUMask(&stream_chain, (uint32_t)words, rest.data()) != hipSuccess) {
			if (stream_up)
				(void)hipStreamDestroy(stream_up);
			stream_up = stream_chain = nullptr;
			(void)hipGetLastError();
		}
		g_cu_masks.store(stream_up ? (split ? 2 : 1) : 0);
		auto bits = [](const std::vector<uint32_t> &m) {
			int n = 0;
			for (uint32_t w : m)
				n += __builtin_popcount(w);
			return n;
		};
		if (stream_up) {
			cus_up = bits(up);
			cus_chain = bits(rest);
		}
		if (stream_up && !down.empty() && hipExtStreamCreateWithCUMask(&stream_down, (uint32_t)words, down.data()) != hipSuccess) {
			stream_down = nullptr;
			(void)hipGetLastError();
		}
		if (stream_down)
			cus_down = bits(down);
	}
	return stream3 ? GEC_OK : make_stream(&stream3);
}

int Staging::cus_of(hipStream_t s) const
{
	if (s && s == stream_up)
		return cus_up;
	if (s && s == stream_chain)
		return cus_chain;
	if (s && s == stream_down)
		return cus_down;
	const int all = qos.num_cu > 0 ? qos.num_cu : 256;
	return qos.background && qos.compute_cus > 0 && qos.compute_cus < all ? qos.compute_cus : all;
}

int Staging::ensure_big(size_t bytes)
{
	if (bytes <= big_cap)
		return GEC_OK;
	if (d_big)
		(void)hipFree(d_big);
	d_big = nullptr;
	big_cap = 0;
	HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d_big), bytes));
	big_cap = bytes;
	return GEC_OK;
}

int Staging::ensure_tab(size_t entries)
{
	tab_used = 0;
	if (entries <= tab_cap)
		return GEC_OK;
	if (h_tab)
		(void)hipHostFree(h_tab);
	h_tab = nullptr;
	tab_cap = 0;
	HIP_TRY(host_malloc_on_node(reinterpret_cast<void **>(&h_tab), entries * sizeof(gec::CopyEntry), hipHostMallocDefault, qos.numa_node));
	tab_cap = entries;
	return GEC_OK;
}

int Staging::ensure(size_t bytes, size_t nbad)
{
	if (!stream)
		if (int rc = make_stream(&stream))
			return rc;
	if (!stream2) {
		if (int rc = make_stream(&stream2))
			return rc;
		HIP_TRY(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
		HIP_TRY(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
		HIP_TRY(hipEventCreateWithFlags(&ev_in, hipEventDisableTiming));
		HIP_TRY(hipEventCreateWithFlags(&ev_out, hipEventDisableTiming));
	}
	if (bytes > cap) {
		if (h_buf)
			(void)hipHostFree(h_buf);
		if (d_buf)
			(void)hipFree(d_buf);
		h_buf = d_buf = nullptr;
		cap = 0;
		HIP_TRY(host_malloc_on_node(reinterpret_cast<void **>(&h_buf), bytes, hipHostMallocDefault, qos.numa_node));
		HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d_buf), bytes));
		cap = bytes;
	}
	if (nbad > bad_cap) {
		if (h_bad)
			(void)hipHostFree(h_bad);
		if (d_bad)
			(void)hipFree(d_bad);
		h_bad = d_bad = nullptr;
		bad_cap = 0;
		HIP_TRY(host_malloc_on_node(reinterpret_cast<void **>(&h_bad), nbad * sizeof(uint32_t), hipHostMallocDefault, qos.numa_node));
		HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d_bad), nbad * sizeof(uint32_t)));
		bad_cap = nbad;
	}
	return GEC_OK;
}
void Staging::release()
{
	if (h_buf)
		(void)hipHostFree(h_buf);
	if (d_buf)
		(void)hipFree(d_buf);
	if (h_bad)
		(void)hipHostFree(h_bad);
	if (d_bad)
		(void)hipFree(d_bad);
	if (h_tab)
		(void)hipHostFree(h_tab);
	if (d_big)
		(void)hipFree(d_big);
	if (stream)
		(void)hipStreamDestroy(stream);
	if (stream2)
		(void)hipStreamDestroy(stream2);
	if (ev_fork)
		(void)hipEventDestroy(ev_fork);
	if (ev_join)
		(void)hipEventDestroy(ev_join);
	if (ev_in)
		(void)hipEventDestroy(ev_in);
	if (ev_out)
		(void)hipEventDestroy(ev_out);
	for (hipEvent_t e : ev_seg)
		if (e)
			(void)hipEventDestroy(e);
	if (stream3)
		(void)hipStreamDestroy(stream3);
	if (stream_up)
		(void)hipStreamDestroy(stream_up);
	if (stream_chain)
		(void)hipStreamDestroy(stream_chain);
	if (stream_down)
		(void)hipStreamDestroy(stream_down);
	for (hipEvent_t e : ev_dec)
		if (e)
			(void)hipEventDestroy(e);
	const QosPolicy keep = qos;
	*this = Staging();
	qos = keep;
}


StagingLease::StagingLease(const gec_codec *cc) : c(cc), unwinding_at_entry(std::uncaught_exceptions())
{
	HipBackend &hb = hip_of(c);
	{
		std::lock_guard<std::mutex> g(hb.pool_mu);
		if (!hb.pool.empty()) {
			st = hb.pool.back();
			hb.pool.pop_back();
		}
	}
	st.qos = hb.qos;
}

StagingLease::~StagingLease()
{
	if (std::uncaught_exceptions() > unwinding_at_entry) {
		// the call is being unwound (bad_alloc in its host code, as a rule) with work possibly still queued on this slot's
		// streams: that work reads the slot's tables and reads / writes the caller's buffers, which the caller is free to
		// release the moment the error code is back -- nothing of it may outlive the call
		int prev = -1;
		(void)hipGetDevice(&prev);
		if (hipSetDevice(c->device) == hipSuccess)
			(void)hipDeviceSynchronize();
		if (prev >= 0)
			(void)hipSetDevice(prev);
	}
	HipBackend &hb = hip_of(c);
	std::lock_guard<std::mutex> g(hb.pool_mu);
	hb.pool.push_back(st);
}

// ------------------------------------------------------------------ who has the link
uint32_t *link_busy_counter(int device)
{
	static std::mutex mu;
	static uint32_t *ctr[64] = {};
	static bool tried[64] = {};
	const unsigned d = (unsigned)device % 64;
	std::lock_guard<std::mutex> g(mu);
	if (!tried[d]) {
		tried[d] = true;
		void *p = nullptr;
		if (hipMalloc(&p, 256) == hipSuccess && hipMemset(p, 0, 256) == hipSuccess)
			ctr[d] = static_cast<uint32_t *>(p);
		else
			(void)hipGetLastError();
	}
	return ctr[d];
}

// Foreground link kernels announce themselves, background ones give way (GEC_BG_LINK_WAIT_US per workgroup and launch);
// a paced copy (rebuilt shards on their way home) is in no hurry and does neither.
void link_role_of(const Staging &st, unsigned pace_ns, uint32_t **busy, uint32_t *role, uint32_t *wait_ticks)
{
	*busy = nullptr;
	*role = gec::LINK_NONE;
	*wait_ticks = 0;
	const unsigned wait_us = env().bg_link_wait_us;
	if (wait_us == 0 || (pace_ns != 0 && !st.qos.background))  // (a background kernel gives way even when it is paced)
		return;
	uint32_t *c = link_busy_counter(st.qos.device);
	if (!c)
		return;
	*busy = c;
	*role = st.qos.background ? gec::LINK_YIELD : gec::LINK_SIGNAL;
	*wait_ticks = wait_us * 100u;
}

// ------------------------------------------------------------------ foreground / background
QosGate &QosGate::of(int device)
{
	static QosGate gates[64];
	return gates[(unsigned)device % 64];
}

void QosGate::leave()
{
	if (fg_.fetch_sub(1, std::memory_order_acq_rel) == 1) {
		std::lock_guard<std::mutex> g(mu_);
		cv_.notify_all();
	}
}

uint64_t QosGate::yield_to_foreground(unsigned max_wait_us)
{
	if (max_wait_us == 0 || fg_.load(std::memory_order_acquire) == 0)
		return 0;
	const auto t0 = std::chrono::steady_clock::now();
	std::unique_lock<std::mutex> lk(mu_);
	yields_.fetch_add(1);
	cv_.wait_for(lk, std::chrono::microseconds(max_wait_us), [&] { return fg_.load(std::memory_order_acquire) == 0; });
	return (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
}

namespace {
thread_local int t_call_depth = 0;
}

ForegroundScope::ForegroundScope(const gec_codec *c)
{
	const unsigned cap = env().max_calls;
	if (cap > 0 && t_call_depth++ == 0) {
		const HipBackend &hb = hip_of(c);
		std::unique_lock<std::mutex> lk(hb.calls_mu);
		hb.calls_cv.wait(lk, [&] { return hb.calls_in_flight < cap; });
		++hb.calls_in_flight;
		permit_of = &hb;
	}
	if (c->qos_class == GEC_CLASS_FOREGROUND) {
		gate = &QosGate::of(c->device);
		gate->enter();
	}
}

ForegroundScope::~ForegroundScope()
{
	if (gate)
		gate->leave();
	if (env().max_calls > 0)
		--t_call_depth;
	if (permit_of) {
		{
			std::lock_guard<std::mutex> g(permit_of->calls_mu);
			--permit_of->calls_in_flight;
		}
		permit_of->calls_cv.notify_one();
	}
}

// Called by a background codec's chunk loops before every chunk: the chunk waits (at most GEC_BG_YIELD_US) for the
// foreground calls in flight to drain, so that a PutObject's encode finds the link and the copy threads free within
// one background chunk.  (Pausing the background class in proportion to the foreground's load was tried and dropped:
// under a saturating PutObject load it slowed the scrub and left the puts where they were, profiles/r03_qos.txt.)
void background_yield(const gec_codec *c)
{
	if (c->qos_class == GEC_CLASS_BACKGROUND)
		(void)QosGate::of(c->device).yield_to_foreground(env().bg_yield_us);
}

size_t trip_chunk_bytes(const gec_codec *c)
{
	return c->qos_class == GEC_CLASS_BACKGROUND ? (env().bg_chunk_mb << 20) : 8 * kChunkBytes;
}

// sweep on MI355X: 16..64 MiB 33-37 GiB/s, 128 MiB 45, 256 MiB 39
size_t pinned_chunk_bytes(const gec_codec *c)
{
	return c->qos_class == GEC_CLASS_BACKGROUND ? std::min(env().bg_chunk_mb, env().pinned_chunk_mb) << 20 : env().pinned_chunk_mb << 20;
}

}  // namespace gecimpl

using namespace gecimpl;

// gec_host_alloc_near: pinned like gec_host_alloc's, with the pages on this codec's memory node
void *HipBackend::host_alloc(size_t bytes) const
{
	void *p = nullptr;
	DeviceGuard g(device);
	hipError_t e = host_malloc_on_node(&p, std::max<size_t>(bytes, 1), hipHostMallocPortable, numa_node_);
	if (e != hipSuccess) {
		fail(e == hipErrorOutOfMemory ? GEC_E_NOMEM : GEC_E_DEVICE, std::string("hipHostMalloc: ") + hipGetErrorString(e));
		return nullptr;
	}
	try {
		pinned().add(p, std::max<size_t>(bytes, 1), true);
	} catch (...) {
		(void)hipHostFree(p);
		throw;
	}
	return p;
}

extern "C" {

// ------------------------------------------------------------ pinned host memory
void *gec_host_alloc(size_t bytes)
{
	void *p = nullptr;
	// portable: usable by every device's DMA engines (one process may drive several codecs)
	if (hip_device_count() == 0) {
		// a host without a (working) device: page-aligned heap memory, so that callers that draw their buffers from
		// here -- libgarage_block's pool -- keep working over a GEC_BACKEND_CPU codec.  Not device-addressable.
		const size_t n = (std::max<size_t>(bytes, 1) + 4095) / 4096 * 4096;
		p = std::aligned_alloc(4096, n);
		if (!p) {
			fail(GEC_E_NOMEM, "aligned_alloc failed");
			return nullptr;
		}
		try {
			pinned().add(p, n, true, 0, /*plain=*/true);
		} catch (...) {  // the registry could not grow
			std::free(p);
			(void)on_exception();
			return nullptr;
		}
		return p;
	}
	hipError_t e = hipHostMalloc(&p, std::max<size_t>(bytes, 1), hipHostMallocPortable);
	if (e != hipSuccess) {
		fail(e == hipErrorOutOfMemory ? GEC_E_NOMEM : GEC_E_DEVICE, std::string("hipHostMalloc: ") + hipGetErrorString(e));
		return nullptr;
	}
	try {
		pinned().add(p, std::max<size_t>(bytes, 1), true);
	} catch (...) {
		(void)hipHostFree(p);
		(void)on_exception();
		return nullptr;
	}
	return p;
}

void gec_host_free(void *p)
{
	bool owned = false, plain = false;
	if (p && pinned().remove(p, owned, plain) && owned) {
		if (plain)
			std::free(p);
		else
			(void)hipHostFree(p);
	}
}

int gec_host_register(void *p, size_t bytes)
try {
	if (!p || bytes == 0)
		return fail(GEC_E_INVALID_ARG, "NULL / empty range");
	HIP_TRY(hipHostRegister(p, bytes, hipHostRegisterPortable | hipHostRegisterMapped));
	void *dptr = p;
	if (hipHostGetDevicePointer(&dptr, p, 0) != hipSuccess || !dptr)
		dptr = p;
	try {
		pinned().add(p, bytes, false, reinterpret_cast<intptr_t>(dptr) - reinterpret_cast<intptr_t>(p));
	} catch (...) {  // the registry could not grow: the range must not stay mapped behind a call that failed
		(void)hipHostUnregister(p);
		throw;
	}
	return GEC_OK;
}
GEC_CATCH

int gec_host_unregister(void *p)
try {
	bool owned = false, plain = false;
	if (!p || !pinned().remove(p, owned, plain))
		return fail(GEC_E_INVALID_ARG, "not a registered range");
	if (owned) {  // it came from gec_host_alloc: treat like gec_host_free
		if (plain)
			std::free(p);
		else
			(void)hipHostFree(p);
		return GEC_OK;
	}
	HIP_TRY(hipHostUnregister(p));
	return GEC_OK;
}
GEC_CATCH

int gec_host_is_pinned(const void *p, size_t bytes) { return pinned().contains(p, bytes) ? 1 : 0; }

uint64_t gec_qos_yields(int device) { return QosGate::of(device).yields(); }

int gec_cu_masks_active(void) { return g_cu_masks.load(); }

}  // extern "C"
