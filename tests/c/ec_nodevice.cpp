// TEST-ONLY link partner for the sanitizer builds (tests/c/Makefile): libgarage_ec's host-only sources -- ec_api.cpp,
// ec_env.cpp, ec_cpu.cpp, i.e. the C ABI and the product's own CPU backend -- are linked WITHOUT the HIP backend, so
// that ASan / UBSan / TSan see the whole path libgarage_block -> C ABI -> CPU data path and nothing of the ROCm runtime.
// This file supplies the three things ec_hip_*.cpp would: "there is no device in this build".
#include <cstdlib>

#include "../../garage_amd/csrc/ec_internal.hpp"

namespace gecimpl {

int hip_device_count() { return 0; }

int make_hip_backend(gec_codec *, int, std::unique_ptr<Backend> &)
{
	return fail(GEC_E_DEVICE, "no HIP backend in this build (host-only sanitizer link of libgarage_ec)");
}

}  // namespace gecimpl

extern "C" {

// same contract as the real library on a host without a device: page-aligned ordinary memory
void *gec_host_alloc(size_t bytes)
{
	const size_t n = ((bytes ? bytes : 1) + 4095) / 4096 * 4096;
	return std::aligned_alloc(4096, n);
}
void gec_host_free(void *p) { std::free(p); }
int gec_host_register(void *, size_t) { return gecimpl::fail(GEC_E_DEVICE, "no device"); }
int gec_host_unregister(void *) { return gecimpl::fail(GEC_E_DEVICE, "no device"); }
int gec_host_is_pinned(const void *, size_t) { return 0; }
uint64_t gec_qos_yields(int) { return 0; }
int gec_cu_masks_active(void) { return -1; }

}  // extern "C"
