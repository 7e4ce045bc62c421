// Does hipMemcpyAsync from PAGEABLE host memory return before the source has been read?  (ec_hip_launch.hip uploads two small
// tables from std::vector storage that dies when the launcher returns.)  For each size: time to return, time to stream idle, and
// whether bytes the host overwrote right after the call reached the device.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main()
{
	hipStream_t s;
	CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
	const size_t sizes[] = {256, 4096, 65536, 262144, 1 << 20, 4 << 20, 16 << 20, 64 << 20, 200 << 20};
	for (size_t n : sizes) {
		uint8_t *h = (uint8_t *)malloc(n), *back = (uint8_t *)malloc(n);
		uint8_t *d;
		CK(hipMalloc(&d, n));
		int overwritten_seen = 0;
		double ret_us = 0, idle_us = 0;
		for (int rep = 0; rep < 5; ++rep) {
			memset(h, 0x11, n);
			CK(hipStreamSynchronize(s));
			auto t0 = std::chrono::steady_clock::now();
			CK(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, s));
			auto t1 = std::chrono::steady_clock::now();
			memset(h, 0xEE, n);  // what a freed vector's next owner would do
			CK(hipStreamSynchronize(s));
			auto t2 = std::chrono::steady_clock::now();
			CK(hipMemcpy(back, d, n, hipMemcpyDeviceToHost));
			for (size_t i = 0; i < n; i += 64)
				if (back[i] != 0x11) { overwritten_seen = 1; break; }
			ret_us = std::chrono::duration<double, std::micro>(t1 - t0).count();
			idle_us = std::chrono::duration<double, std::micro>(t2 - t0).count();
		}
		printf("%10zu B: returns after %9.1f us, stream idle after %9.1f us, device saw the host's later writes: %s\n", n, ret_us, idle_us, overwritten_seen ? "YES (the copy is asynchronous to the host)" : "no");
		CK(hipFree(d));
		free(h);
		free(back);
	}
	return 0;
}
