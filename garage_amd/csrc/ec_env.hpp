// ec_env.hpp -- every environment switch of libgarage_ec, in one place.  All are optional, read once per process,
// and none changes results: they select between equivalent paths (A/B measurements), size pools, or name a library.
// ec_env.cpp holds the table (name, default, meaning) that gec_env_table() prints; include/garage_ec.h and
// INTEGRATION.md quote it.
#pragma once

#include <cstddef>
#include <cstdint>
#include <string>

namespace gecimpl {

struct Env {
	// ---- both backends
	int cpu_threads;        // GEC_CPU_THREADS: worker threads of a CPU codec (0 = the calling thread only)
	std::string cpu_isa;    // GEC_CPU_ISA: auto | gfni | avx2 | scalar
	// ---- HIP backend: host-pointer paths
	unsigned copy_threads;  // GEC_COPY_THREADS
	unsigned max_calls;     // GEC_MAX_CALLS
	int upload_cus;         // GEC_UPLOAD_CUS
	unsigned bg_link_wait_us; // GEC_BG_LINK_WAIT_US
	unsigned home_rate_gbps; // GEC_HOME_RATE_GBPS
	size_t fused_max_leaves;  // GEC_FUSED_MAX_LEAVES
	size_t fused_get_max_leaves;  // GEC_FUSED_GET_MAX_LEAVES
	unsigned bg_home_rate_gbps;  // GEC_BG_HOME_RATE_GBPS
	size_t pinned_chunk_mb; // GEC_PINNED_CHUNK_MB
	// ---- HIP backend: background class
	int bg_cus;             // GEC_BG_CUS
	size_t bg_chunk_mb;     // GEC_BG_CHUNK_MB
	unsigned bg_yield_us;   // GEC_BG_YIELD_US
	uint64_t max_cols_per_launch;  // GEC_MAX_COLS_PER_LAUNCH (0 = no cap)
	int numa;               // GEC_NUMA: 1 near (default), 0 off, 2 far (test hook)
	// ---- multi-GPU
	std::string rccl_lib;   // GEC_RCCL_LIB ("" = librccl.so.1, then librccl.so)
};

const Env &env();

// the table as text: one "NAME  default  meaning" line per switch
const char *env_table_text();

}  // namespace gecimpl
