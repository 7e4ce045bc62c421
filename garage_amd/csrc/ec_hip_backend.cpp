// ec_hip_backend.cpp -- GEC_BACKEND_HIP: device probe, the per-codec device state, the compositions of launches
// that several entry points share, and the device-resident (*_dev) entry points.  Host code only.
#include "ec_hip.hpp"

#include <algorithm>

namespace gecimpl {

int encode_dev(const gec_codec *c, size_t nblocks, const uint8_t *d_data, size_t data_stride, size_t S,
	       uint8_t *d_parity, size_t parity_stride, hipStream_t stream)
{
	const int k = c->k, m = c->m;
	std::vector<size_t> in_off(k), out_off(m);
	for (int t = 0; t < k; ++t)
		in_off[t] = (size_t)t * S;
	for (int r = 0; r < m; ++r)
		out_off[r] = (size_t)r * S;
	return launch_apply(c, d_data, data_stride, d_parity, parity_stride, nullptr, 0, S, nblocks, in_off.data(),
			    out_off.data(), m, c->enc.row(k), gec::MODE_STORE, stream);
}

int verify_dev(const gec_codec *c, size_t nblocks, const uint8_t *d_stripes, size_t stride, size_t S,
	       uint32_t *d_bad, hipStream_t stream)
{
	const int k = c->k, m = c->m;
	std::vector<size_t> in_off(k), out_off(m);
	for (int t = 0; t < k; ++t)
		in_off[t] = (size_t)t * S;
	for (int r = 0; r < m; ++r)
		out_off[r] = (size_t)(k + r) * S;
	if (int rc = launch_clear_flags(d_bad, nblocks, stream))
		return rc;
	return launch_apply(c, d_stripes, stride, const_cast<uint8_t *>(d_stripes), stride, d_bad, 0, S, nblocks,
			    in_off.data(), out_off.data(), m, c->enc.row(k), gec::MODE_COMPARE, stream);
}

int leaf_scratch(const gec_codec *c, hipStream_t stream, size_t bytes, uint8_t **out)
{
	HipBackend &hb = hip_of(c);
	std::lock_guard<std::mutex> g(hb.leaf_mu);
	HipBackend::LeafScratch &ls = hb.leaf_scratch[stream];
	if (bytes > ls.cap) {
		if (ls.p) {
			HIP_TRY(hipStreamSynchronize(stream));  // earlier launches on this stream may still read the old one
			(void)hipFree(ls.p);
			ls.p = nullptr;
			ls.cap = 0;
		}
		const size_t want = std::max<size_t>(bytes + bytes / 4, 1 << 20);
		HIP_TRY(hipMalloc(reinterpret_cast<void **>(&ls.p), want));
		ls.cap = want;
	}
	*out = ls.p;
	return GEC_OK;
}

int done_counters(const gec_codec *c, hipStream_t stream, size_t nblocks, uint32_t **out)
{
	HipBackend &hb = hip_of(c);
	std::lock_guard<std::mutex> g(hb.leaf_mu);
	HipBackend::LeafScratch &ls = hb.done_counters[stream];
	const size_t bytes = nblocks * sizeof(uint32_t);
	if (bytes > ls.cap) {
		if (ls.p) {
			HIP_TRY(hipStreamSynchronize(stream));
			(void)hipFree(ls.p);
			ls.p = nullptr;
			ls.cap = 0;
		}
		const size_t want = std::max<size_t>(2 * bytes, 4096);
		HIP_TRY(hipMalloc(reinterpret_cast<void **>(&ls.p), want));
		HIP_TRY(hipMemset(ls.p, 0, want));
		ls.cap = want;
	}
	*out = reinterpret_cast<uint32_t *>(ls.p);
	return GEC_OK;
}

int reconstruct_dev(const gec_codec *c, size_t nblocks, uint8_t *d_base, size_t stride, const size_t *shard_off,
		    const uint8_t *present, bool data_only, size_t byte_off, size_t byte_len, hipStream_t stream)
{
	std::shared_ptr<const Plan> plan;
	int rc = get_plan(c, present, data_only, plan);
	if (rc)
		return rc;
	if (plan->missing.empty())
		return GEC_OK;
	const int k = c->k;
	std::vector<size_t> in_off(k), out_off(plan->missing.size());
	for (int t = 0; t < k; ++t)
		in_off[t] = shard_off[plan->valid[t]];
	for (size_t r = 0; r < plan->missing.size(); ++r)
		out_off[r] = shard_off[plan->missing[r]];
	return launch_apply(c, d_base, stride, d_base, stride, nullptr, byte_off, byte_len, nblocks, in_off.data(),
			    out_off.data(), (int)plan->missing.size(), plan->rows.v.data(), gec::MODE_STORE, stream);
}

// Encode + the blake2sum of all k+m shards of every stripe (d_stripes: shard j of block b at b*stride + j*S),
// everything enqueued behind whatever `stream` already holds.  The checksums of the k data shards do not depend
// on the encode, so they are computed on a second stream BESIDE the RS kernel (HBM-bound, it leaves the VALUs
// mostly idle; the hash is a pure VALU dependency chain); only the m parity checksums follow the encode.
// `aux` provides the partner stream and the fork/join events.
int encode_hash_dev(const gec_codec *c, size_t nblocks, uint8_t *d_stripes, size_t stride, size_t S, uint8_t *d_sums,
		    hipStream_t stream, Staging &aux)
{
	const size_t k = c->k, m = c->m, n = k + m;
	if (c->sumkind == GEC_SHARDSUM_MLH64) {
		// checksum v3: ONE pass -- the encode kernel leaves the leaf sums of the k shards it reads and the m rows it writes,
		// from the registers that hold them (no second read of the stripe); then one lane per shard for the roots
		const uint32_t nleaf_max = (uint32_t)((S + gec::SHARDSUM_LEAF - 1) / gec::SHARDSUM_LEAF);
		uint8_t *scratch = nullptr;
		int rc = leaf_scratch(c, stream, nblocks * n * nleaf_max * 8, &scratch);
		if (rc)
			return rc;
		const SumOut so{reinterpret_cast<uint64_t *>(scratch), nleaf_max, (uint32_t)n, 0u, true};
		std::vector<size_t> in_off(k), out_off(m);
		for (size_t t = 0; t < k; ++t)
			in_off[t] = t * S;
		for (size_t r = 0; r < m; ++r)
			out_off[r] = (k + r) * S;
		rc = launch_apply(c, d_stripes, stride, d_stripes, stride, nullptr, 0, S, nblocks, in_off.data(), out_off.data(), (int)m,
				  c->enc.row((int)k), gec::MODE_STORE, stream, &so);
		if (rc)
			return rc;
		return mlh_roots_dev(c, nblocks * n, so.lsum, nleaf_max, nullptr, S, d_sums, stream);
	}
	const bool fork = true;  // (one stream for everything lost its A/B by 2x: the data shards' chains hide behind the RS kernel)
	if (fork) {
		HIP_TRY(hipEventRecord(aux.ev_fork, stream));
		HIP_TRY(hipStreamWaitEvent(aux.stream2, aux.ev_fork, 0));
	}
	int rc = blake2_dev(c, nblocks * k, d_stripes, nullptr, nullptr, S, S, d_sums, fork ? aux.stream2 : stream, (uint32_t)k, stride, (uint32_t)n, true);
	if (rc)
		return rc;
	if (fork)
		HIP_TRY(hipEventRecord(aux.ev_join, aux.stream2));
	rc = encode_dev(c, nblocks, d_stripes, stride, S, d_stripes + k * S, stride, stream);
	if (rc)
		return rc;
	rc = blake2_dev(c, nblocks * m, d_stripes + k * S, nullptr, nullptr, S, S, d_sums + 32 * k, stream, (uint32_t)m, stride, (uint32_t)n, true);
	if (rc)
		return rc;
	if (fork)
		HIP_TRY(hipStreamWaitEvent(stream, aux.ev_join, 0));
	return GEC_OK;
}

// ------------------------------------------------------------------ per-codec device state
int hip_device_count()
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) {
		(void)hipGetLastError();
		return 0;
	}
	return n;
}

ForkJoinPool &HipBackend::copy_pool() const
{
	std::call_once(copy_once, [this] {
		// a background codec keeps to a couple of copy threads: the staging copies of a scrub must not take the
		// memory bandwidth a PutObject's copies need
		const unsigned n = env().copy_threads;
		copy_threads.reset(new ForkJoinPool(qos.background ? std::min(n, 2u) : n, numa_cpus_));  // (on the device's node: numa.hpp)
	});
	return *copy_threads;
}

HipBackend::~HipBackend()
{
	if (qos.background)
		background_codec_gone(device);
	DeviceGuard g(device);
	for (auto &s : pool)
		s.release();
	for (auto &kv : leaf_scratch)
		if (kv.second.p)
			(void)hipFree(kv.second.p);
	for (auto &kv : done_counters)
		if (kv.second.p)
			(void)hipFree(kv.second.p);
	if (d_logexp)
		(void)hipFree(d_logexp);
}

int make_hip_backend(gec_codec *c, int device, std::unique_ptr<Backend> &out)
{
	int ndev = 0;
	hipError_t e = hipGetDeviceCount(&ndev);
	if (e != hipSuccess || ndev <= 0)
		return fail(GEC_E_DEVICE, std::string("no HIP device available (GEC_BACKEND_CPU or GEC_BACKEND_AUTO run on the host cores): ") +
						  (e != hipSuccess ? hipGetErrorString(e) : "device count is 0"));
	if (device < 0 || device >= ndev)
		return fail(GEC_E_INVALID_ARG, "device index out of range");
	std::unique_ptr<HipBackend> hb(new (std::nothrow) HipBackend());
	if (!hb)
		return fail(GEC_E_NOMEM, "alloc backend");
	hb->c = c;
	hb->device = device;
	DeviceGuard g(device);
	if (!g.ok)
		return fail(GEC_E_DEVICE, "hipSetDevice failed");
	hipDeviceProp_t prop;
	HIP_TRY(hipGetDeviceProperties(&prop, device));
	hb->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
	hb->qos.num_cu = hb->num_cu;
	hb->qos.device = device;
	// the device's memory node: PCI address -> /sys/bus/pci/devices/<bdf>/numa_node (numa.hpp; GEC_NUMA)
	char bdf[64] = {0};
	if (hipDeviceGetPCIBusId(bdf, (int)sizeof bdf, device) == hipSuccess)
		hb->numa_node_of_device = gecnuma::node_of_pci(bdf);
	else
		(void)hipGetLastError();
	const int nnodes = gecnuma::node_count();
	if (env().numa != 0 && hb->numa_node_of_device >= 0 && nnodes > 1) {
		hb->numa_node_ = env().numa == 2 ? (hb->numa_node_of_device + 1) % nnodes : hb->numa_node_of_device;
		hb->numa_cpus_ = gecnuma::cpus_of_node(hb->numa_node_);
		if (hb->numa_cpus_.empty())
			hb->numa_node_ = -1;
	}
	hb->qos.numa_node = hb->numa_node_;
	hb->qos.compute_cus_plan = env().bg_cus > 0 && env().bg_cus < hb->num_cu ? env().bg_cus : 0;
	if (c->qos_class == GEC_CLASS_BACKGROUND) {
		// low-priority streams (the hardware scheduler hands free CUs to the foreground queues first) and a CU mask
		// so that a long checksum kernel of a scrub cannot occupy the whole chip when a PutObject's encode arrives
		hb->qos.background = true;
		int least = 0, greatest = 0;
		if (hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess)
			hb->qos.stream_priority = least;
		else
			(void)hipGetLastError();
		hb->qos.compute_cus = env().bg_cus > 0 && env().bg_cus < hb->num_cu ? env().bg_cus : 0;
		background_codec_born(device);
	}
	gec::LogExp le;
	const gec::Field &f = gec::field();
	std::memcpy(le.exp, f.exp.data(), 512);
	std::memcpy(le.log, f.log.data(), 256);
	HIP_TRY(hipMalloc(reinterpret_cast<void **>(&hb->d_logexp), sizeof(le)));
	HIP_TRY(hipMemcpy(hb->d_logexp, &le, sizeof(le), hipMemcpyHostToDevice));
	out = std::move(hb);
	return GEC_OK;
}

// ------------------------------------------------------------------ device-resident entry points
// (ec_api.cpp has checked pointers, alignment and strides)
int HipBackend::encode_batch_dev(size_t nblocks, const void *d_data, size_t data_stride, size_t S, void *d_parity,
				 size_t parity_stride, void *hip_stream)
{
	DeviceGuard g(device);
	if (!g.ok)
		return fail(GEC_E_DEVICE, "hipSetDevice failed");
	return encode_dev(c, nblocks, static_cast<const uint8_t *>(d_data), data_stride, S, static_cast<uint8_t *>(d_parity),
			  parity_stride, static_cast<hipStream_t>(hip_stream));
}

int HipBackend::verify_batch_dev(size_t nblocks, const void *d_stripes, size_t stride, size_t S, uint32_t *d_bad, void *hip_stream)
{
	DeviceGuard g(device);
	if (!g.ok)
		return fail(GEC_E_DEVICE, "hipSetDevice failed");
	return verify_dev(c, nblocks, static_cast<const uint8_t *>(d_stripes), stride, S, d_bad, static_cast<hipStream_t>(hip_stream));
}

int HipBackend::reconstruct_dev(size_t nblocks, void *d_base, size_t block_stride, const size_t *shard_off, size_t S,
				const uint8_t *present, int data_only, size_t byte_off, size_t byte_len, void *hip_stream)
{
	DeviceGuard g(device);
	if (!g.ok)
		return fail(GEC_E_DEVICE, "hipSetDevice failed");
	std::vector<size_t> contiguous;
	if (!shard_off) {
		contiguous = stripe_offsets(c, S);
		shard_off = contiguous.data();
	}
	return gecimpl::reconstruct_dev(c, nblocks, static_cast<uint8_t *>(d_base), block_stride, shard_off, present, data_only != 0,
					byte_off, byte_len, static_cast<hipStream_t>(hip_stream));
}

// One erasure pattern PER BLOCK (the crate's reconstruct is per call; a device-resident batch gathered from a degraded
// cluster has a mix): the blocks are grouped by pattern on the host -- one decode plan per distinct pattern, cached like any
// other -- and rebuilt by ONE launch in which every workgroup expands the table of the block it is at (gf_apply_nibble_pat).
int HipBackend::reconstruct_dev_ex(size_t nblocks, void *d_stripes, size_t stride, size_t S, const uint8_t *present, int data_only,
				   void *hip_stream)
{
	DeviceGuard g(device);
	if (!g.ok)
		return fail(GEC_E_DEVICE, "hipSetDevice failed");
	const size_t n = (size_t)c->k + c->m;
	hipStream_t stream = static_cast<hipStream_t>(hip_stream);
	std::unordered_map<std::string, uint16_t> seen;
	std::vector<std::shared_ptr<const Plan>> plans;
	std::vector<uint16_t> pat(nblocks);
	std::vector<size_t> wide;  // blocks whose pattern has more than RMAX rows (codes with m > 8): one at a time
	for (size_t b = 0; b < nblocks; ++b) {
		std::string key(reinterpret_cast<const char *>(present + b * n), n);
		for (char &ch : key)
			ch = ch ? 1 : 0;
		auto it = seen.find(key);
		if (it == seen.end()) {
			if (plans.size() >= 0xffff)
				return fail(GEC_E_INVALID_ARG, "more than 65535 distinct erasure patterns in one call");
			std::shared_ptr<const Plan> plan;
			int rc = get_plan(c, present + b * n, data_only != 0, plan);
			if (rc)
				return rc;
			it = seen.emplace(key, (uint16_t)plans.size()).first;
			plans.push_back(plan);
		}
		pat[b] = it->second;
		if (plans[pat[b]]->missing.size() > (size_t)gec::RMAX)
			wide.push_back(b);
	}
	if (!wide.empty()) {
		// rare (m > 8 and more than 8 shards of a block gone): those blocks alone, the uniform way -- and their patterns
		// become "nothing to do" for the launch below, which takes the rest
		const std::vector<size_t> off = stripe_offsets(c, S);
		for (size_t b : wide) {
			int rc = gecimpl::reconstruct_dev(c, 1, static_cast<uint8_t *>(d_stripes) + b * stride, stride, off.data(), present + b * n, data_only != 0, 0, S, stream);
			if (rc)
				return rc;
		}
		for (auto &pl : plans)
			if (pl->missing.size() > (size_t)gec::RMAX) {
				std::shared_ptr<Plan> none(new Plan(*pl));
				none->missing.clear();
				pl = none;
			}
	}
	return launch_apply_pat(c, static_cast<uint8_t *>(d_stripes), stride, S, nblocks, plans, pat, stream);
}

int HipBackend::hash_batch_dev(size_t n, const void *d_base, size_t stride, size_t len, void *d_out, void *hip_stream, bool tree)
{
	DeviceGuard g(device);
	if (!g.ok)
		return fail(GEC_E_DEVICE, "hipSetDevice failed");
	return blake2_dev(c, n, static_cast<const uint8_t *>(d_base), nullptr, nullptr, stride, len, static_cast<uint8_t *>(d_out),
			  static_cast<hipStream_t>(hip_stream), 0, 0, 0, tree, len);
}

int HipBackend::encode_hash_batch_dev(size_t nblocks, void *d_stripes, size_t stride, size_t S, void *d_sums, void *hip_stream)
{
	DeviceGuard g(device);
	if (!g.ok)
		return fail(GEC_E_DEVICE, "hipSetDevice failed");
	StagingLease aux(c);  // only its partner stream and events are used; enqueued work may outlive the lease
	int rc = aux.st.ensure(0, 0);
	if (rc)
		return rc;
	return encode_hash_dev(c, nblocks, static_cast<uint8_t *>(d_stripes), stride, S, static_cast<uint8_t *>(d_sums),
			       static_cast<hipStream_t>(hip_stream), aux.st);
}

}  // namespace gecimpl
