// ec_env.cpp -- the one table of libgarage_ec's environment switches (see ec_env.hpp).
#include "ec_env.hpp"
#include "blake2b_mb.hpp"
#include "mlh64_host.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <thread>

namespace gecimpl {
namespace {

struct Row {
	const char *name, *def, *doc;
};

// Keep in step with the parser below and with the table in include/garage_ec.h / INTEGRATION.md
// (tests/test_cabi_host.py checks that every GEC_* name in the sources appears here).
const Row kRows[] = {
	{"GEC_CPU_THREADS", "min(cores, 16)", "threads a CPU codec spreads one call over (1 = the calling thread only)"},
	{"GEC_CPU_ISA", "auto", "CPU backend kernel: auto, gfni (AVX-512 + GFNI), avx2 (split-nibble vpshufb) or scalar; the host forms of shard checksum v3 (AVX-512 / AVX2 / scalar) and of BLAKE2b (eight messages at a time with AVX-512, else one) follow it, in libgarage_block as well"},
	{"GEC_MAX_CALLS", "4", "host-pointer calls in flight per HIP codec; further callers wait (0 = no limit: every concurrent call gets staging slots and device queues of its own)"},
	{"GEC_COPY_THREADS", "7", "staging-copy threads per HIP codec for pageable caller memory (0 = copy on the calling thread)"},
	{"GEC_UPLOAD_CUS", "16", "CUs reserved for kernels that read / write host memory (0 = no CU masks)"},
	{"GEC_PINNED_CHUNK_MB", "128", "chunk size of the staged path for pinned memory"},
	{"GEC_BG_CUS", "64", "CUs a background-class codec's kernels may occupy (0 = no mask; link kernels stay on GEC_UPLOAD_CUS)"},
	{"GEC_BG_CHUNK_MB", "32", "chunk size of a background-class codec's host-pointer trips (a foreground call waits for at most one)"},
	{"GEC_BG_YIELD_US", "2000", "a background chunk waits up to this long for foreground calls on the same device to drain (0 = never waits)"},
	{"GEC_BG_LINK_WAIT_US", "200", "a background link kernel's workgroups sleep while foreground link kernels run on the device, at most this long per launch (0 = the classes share the link as it comes)"},
	{"GEC_HOME_RATE_GBPS", "25", "the read path sends rebuilt shards home no faster than this while checksum chains run (0 = unpaced, one workgroup per tile): a link saturated with writes backs up into the fabric and every other kernel's loads wait"},
	{"GEC_FUSED_MAX_LEAVES", "3300", "a put trip (gec_encode_hash_batch) with fewer 4 KiB leaves to hash than this takes the one-launch kernel (a 1 MiB RS(10,4) block has 364: up to 9 such blocks; from there on the link kernel + the one-lane-per-leaf checksum kernels are faster per trip, profiles/r04_trip_bench.txt)"},
	{"GEC_FUSED_GET_MAX_LEAVES", "4400", "the same for a read trip (gec_decode_verify_batch without block checksums: k leaves per tile, 260 per 1 MiB RS(10,4) block: up to 16 such blocks)"},
	{"GEC_BG_HOME_RATE_GBPS", "20", "a background-class codec writes rebuilt shards into host memory (resync's rebuilds on their way home) no faster than this (0 = unpaced)"},
	{"GEC_MAX_COLS_PER_LAUNCH", "0", "test hook: cap on the 16-byte columns one launch covers, to exercise the multi-launch split on small inputs"},
	{"GEC_NUMA", "1", "a HIP codec keeps its host side on its device's memory node: copy threads run on that node's CPUs, pinned staging slots and gec_host_alloc_near memory are bound to it (libgarage_block does the same with a lane's pool / batcher threads and shard buffers); 0 = off: threads and pages go where the scheduler and the HIP runtime put them; far = test hook: the node the device is NOT on (the forced-far leg of profiles/r06_numa.txt)"},
	{"GEC_RCCL_LIB", "librccl.so.1", "RCCL to dlopen for gec_group_* (when set: that library or GEC_E_DEVICE, no fallback)"},
};

const char *get(const char *name) { return std::getenv(name); }

long get_long(const char *name, long def)
{
	const char *e = get(name);
	return e && *e ? std::strtol(e, nullptr, 0) : def;
}

}  // namespace

const Env &env()
{
	static const Env e = [] {
		Env v;
		const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
		v.cpu_threads = (int)std::min<long>(std::max<long>(get_long("GEC_CPU_THREADS", std::min(hw, 16u)), 1), 256);
		v.cpu_isa = get("GEC_CPU_ISA") ? get("GEC_CPU_ISA") : "auto";
		if (v.cpu_isa == "scalar")  // the host form of shard checksum v3 follows the same switch (mlh64_host.hpp)
			mlh::isa_cap().store(0);
		else if (v.cpu_isa == "avx2")
			mlh::isa_cap().store(1);
		if (v.cpu_isa == "scalar" || v.cpu_isa == "avx2")  // ... and so does the host-side BLAKE2b: one message at a time without AVX-512
			b2host::mb_mode().store(0);
		v.max_calls = (unsigned)std::max<long>(get_long("GEC_MAX_CALLS", 4), 0);
		v.copy_threads = get("GEC_COPY_THREADS") ? (unsigned)std::min<unsigned long>(std::strtoul(get("GEC_COPY_THREADS"), nullptr, 0), 64ul)
							 : std::min(7u, hw - 1);
		v.upload_cus = (int)get_long("GEC_UPLOAD_CUS", 16);
		v.pinned_chunk_mb = (size_t)std::max<long>(get_long("GEC_PINNED_CHUNK_MB", 128), 1);
		v.bg_cus = (int)get_long("GEC_BG_CUS", 64);
		v.bg_chunk_mb = (size_t)std::max<long>(get_long("GEC_BG_CHUNK_MB", 32), 1);
		v.bg_yield_us = (unsigned)std::max<long>(get_long("GEC_BG_YIELD_US", 2000), 0);
		v.bg_link_wait_us = (unsigned)std::min<long>(std::max<long>(get_long("GEC_BG_LINK_WAIT_US", 200), 0), 1000000);
		v.home_rate_gbps = (unsigned)std::max<long>(get_long("GEC_HOME_RATE_GBPS", 25), 0);
		v.fused_max_leaves = (size_t)std::max<long>(get_long("GEC_FUSED_MAX_LEAVES", 3300), 0);
		v.fused_get_max_leaves = (size_t)std::max<long>(get_long("GEC_FUSED_GET_MAX_LEAVES", 4400), 0);
		v.bg_home_rate_gbps = (unsigned)std::max<long>(get_long("GEC_BG_HOME_RATE_GBPS", 20), 0);
		v.max_cols_per_launch = get("GEC_MAX_COLS_PER_LAUNCH") ? std::strtoull(get("GEC_MAX_COLS_PER_LAUNCH"), nullptr, 0) : 0ull;
		v.rccl_lib = get("GEC_RCCL_LIB") ? get("GEC_RCCL_LIB") : "";
		const char *nu = get("GEC_NUMA");
		v.numa = !nu || !*nu ? 1 : (nu[0] == '0' ? 0 : (nu[0] == 'f' ? 2 : 1));
		return v;
	}();
	return e;
}

const char *env_table_text()
{
	static const std::string text = [] {
		std::string s;
		for (const Row &r : kRows) {
			s += r.name;
			s += "\t";
			s += r.def;
			s += "\t";
			s += r.doc;
			s += "\n";
		}
		return s;
	}();
	return text.c_str();
}

}  // namespace gecimpl

namespace {
// the switches that act on header-only code shared with libgarage_block (blake2b_mb.hpp) take effect when the library is loaded
const bool kEnvReadAtLoad = (gecimpl::env(), true);
}  // namespace
