// ec_internal.hpp -- what every translation unit of libgarage_ec shares (no HIP types in here).
//
// Layout of the library (include/garage_ec.h is the contract):
//   ec_api.cpp          the C ABI: argument checks common to both backends, codec life cycle, dispatch
//   ec_env.cpp          every GEC_* environment switch, read once, in one table
//   ec_cpu.cpp          GEC_BACKEND_CPU: the host cores' data path (GFNI / AVX2 / scalar), host-pointer calls only
//   ec_hip_backend.cpp  GEC_BACKEND_HIP: device probe, per-codec device state, the *_dev entry points
//   ec_hip_launch.hip   the ONLY unit with device code (kernels.hpp, blake2b.hpp): launch geometry and launches
//   ec_hip_staging.cpp  staging slots, pinned host memory (gec_host_*), the chunk pipeline
//   ec_hip_host.cpp     the host-pointer entry points of the HIP backend (staged and zero-copy paths)
//   ec_hip_group.cpp    gec_group_*: the striped-object decode over RCCL or a caller transport
// The first three have no HIP dependency at all: tests/c links them (plus a "no device in this build" stand-in for
// the HIP factory) with libgarage_block's sources under ASan / UBSan / TSan.
#pragma once

#include "../../include/garage_ec.h"

#include <atomic>
#include <condition_variable>
#include <cstring>
#include <exception>
#include <functional>
#include <list>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "gf256.hpp"

namespace gecimpl {

// Sets the calling thread's gec_last_error() text and returns `code`.
int fail(int code, const std::string &detail);

// Nothing may unwind across the C ABI (the caller is Rust, C or ctypes): every extern "C" entry point that can allocate is a
// function-try-block ending in GEC_CATCH, which turns whatever was thrown into a code (bad_alloc -> GEC_E_NOMEM, anything
// else -> GEC_E_DEVICE with the exception's text in gec_last_error()).
int on_exception() noexcept;
#define GEC_CATCH                                  \
	catch (...)                                \
	{                                          \
		return ::gecimpl::on_exception();  \
	}

// same order of checks as ReedSolomon::new [EXT]
int check_km(int k, int m);

// What to compute for one erasure pattern: out[missing[r]] = rows[r] . in[valid[*]]
struct Plan {
	std::vector<int> valid;    // k input shard indices
	std::vector<int> missing;  // output shard indices
	gec::Matrix rows;          // missing.size() x k
};

// Tiny fork-join pool: fn(i) for i in [0, n) on the workers and the calling thread.  The HIP backend uses one per
// codec for the staging copies (pageable user memory <-> pinned buffers: one memcpy thread tops out near 10 GB/s,
// well below PCIe Gen5), the CPU backend one per codec for the arithmetic itself.
class ForkJoinPool {
public:
	// `cpus` (optional): the workers are restricted to these CPUs (a HIP codec's copy threads run on its device's memory node)
	explicit ForkJoinPool(unsigned nworkers, const std::vector<int> &cpus = {});
	~ForkJoinPool();
	void parallel_for(size_t n, const std::function<void(size_t)> &fn);
	unsigned workers() const { return (unsigned)workers_.size(); }

private:
	void work();
	void run();
	std::vector<std::thread> workers_;
	std::mutex mu_, call_mu_;
	std::condition_variable cv_, done_cv_;
	const std::function<void(size_t)> *fn_ = nullptr;
	size_t n_ = 0, grab_ = 1, pending_ = 0, active_ = 0;
	std::atomic<size_t> next_{0};
	std::atomic<bool> failed_{false};
	std::exception_ptr err_;  // the first exception of the call in flight (under mu_)
	uint64_t epoch_ = 0;
	bool stop_ = false;
};

// What a backend implements.  ec_api.cpp has validated the arguments that do not depend on the backend (NULL
// pointers, S, shard counts) before any of these is called; nblocks / n is > 0.
struct Backend {
	virtual ~Backend() = default;
	// where the codec's host side lives (numa.hpp): -1 / NULL for a CPU codec, a one-node box, or GEC_NUMA=0
	virtual int numa_node() const { return -1; }
	virtual const std::vector<int> *numa_cpus() const { return nullptr; }
	// pinned memory near the codec's device (gec_host_alloc_near); a CPU codec hands out what gec_host_alloc does
	virtual void *host_alloc(size_t bytes) const;

	// ---- host-pointer entry points (both backends)
	// shard_sums == NULL: gec_encode_batch, else gec_encode_hash_batch
	virtual int encode_batch(size_t nblocks, const uint8_t *const *blocks, const size_t *block_len, size_t S,
				 uint8_t *const *parity, uint8_t *shard_sums) = 0;
	virtual int verify_batch(size_t nblocks, const uint8_t *const *shards, size_t S, uint8_t *ok) = 0;
	virtual int verify_hash_batch(size_t nblocks, const uint8_t *const *shards, size_t S, uint8_t *ok, uint8_t *shard_sums) = 0;
	// in_sums / out_sums: both NULL (gec_reconstruct_batch) or both set (gec_reconstruct_hash_batch)
	virtual int reconstruct_batch(size_t nblocks, const uint8_t *const *shards, uint8_t *const *out, size_t S, int data_only,
				      uint8_t *in_sums, uint8_t *out_sums) = 0;
	virtual int decode_verify_batch(size_t nblocks, const uint8_t *const *shards, size_t S, const size_t *block_len,
					uint8_t *const *rebuilt, uint8_t *shard_sums, uint8_t *block_sums) = 0;
	// tree: the shard checksum (BLAKE2b tree mode) instead of plain blake2sum
	virtual int hash_batch(size_t n, const uint8_t *const *msgs, const size_t *lens, uint8_t *out, bool tree) = 0;

	// ---- device-resident entry points (HIP backend only; a CPU codec answers GEC_E_DEVICE)
	virtual int encode_batch_dev(size_t nblocks, const void *d_data, size_t data_stride, size_t S, void *d_parity,
				     size_t parity_stride, void *hip_stream);
	virtual int verify_batch_dev(size_t nblocks, const void *d_stripes, size_t stride, size_t S, uint32_t *d_bad, void *hip_stream);
	// shard_off == NULL: contiguous stripes (shard j at j*S)
	virtual int reconstruct_dev(size_t nblocks, void *d_base, size_t block_stride, const size_t *shard_off, size_t S,
				    const uint8_t *present, int data_only, size_t byte_off, size_t byte_len, void *hip_stream);
	// one erasure pattern per block: present[b * (k+m) + j]
	virtual int reconstruct_dev_ex(size_t nblocks, void *d_stripes, size_t stride, size_t S, const uint8_t *present, int data_only,
				       void *hip_stream);
	virtual int hash_batch_dev(size_t n, const void *d_base, size_t stride, size_t len, void *d_out, void *hip_stream, bool tree);
	virtual int encode_hash_batch_dev(size_t nblocks, void *d_stripes, size_t stride, size_t S, void *d_sums, void *hip_stream);
};

}  // namespace gecimpl

// The codec: the code itself (k, m, matrix, decode plans) plus the backend that moves the bytes.
struct gec_codec {
	int k = 0, m = 0;
	int backend = GEC_BACKEND_HIP;  // GEC_BACKEND_CPU / GEC_BACKEND_HIP (never AUTO: resolved at creation)
	int device = -1;                // HIP device index; -1 for a CPU codec
	int matrix = GEC_MATRIX_VANDERMONDE;
	int qos_class = GEC_CLASS_FOREGROUND;
	int sumkind = GEC_SHARDSUM_MLH64;  // which shard checksum the *_hash / shardsum entry points produce
	gec::Matrix enc;                // (k+m) x k

	// decode-plan LRU, keyed by present bitmap + data_only (the crate keeps an LRU of decode matrices [EXT])
	static constexpr size_t kCacheCap = 254;
	mutable std::mutex cache_mu;
	mutable std::list<std::string> lru;
	mutable std::unordered_map<std::string, std::pair<std::shared_ptr<const gecimpl::Plan>, std::list<std::string>::iterator>> cache;
	mutable uint64_t inversions = 0;

	std::unique_ptr<gecimpl::Backend> be;
};

namespace gecimpl {

// the decode plan for an erasure pattern (cached per codec)
int get_plan(const gec_codec *c, const uint8_t *present, bool data_only, std::shared_ptr<const Plan> &out);

// Backend factories.  make_hip_backend is defined in ec_hip_backend.cpp (or, in the sanitizer builds of tests/c,
// by a stand-in that reports "no device in this build"); hip_device_count() is what gec_device_count returns.
int make_cpu_backend(gec_codec *c, std::unique_ptr<Backend> &out);
int make_hip_backend(gec_codec *c, int device, std::unique_ptr<Backend> &out);
int hip_device_count();

// gec_thread_link_release: calls the hook this thread armed (once; no-op when none is armed).  A backend calls it when a
// host-pointer call's bulk transfers are over; the entry points call it again on the way out.
void link_release_fire();
bool link_release_armed();

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace gecimpl
