// loopback_transport.cpp -- a gec_allgather_fn for N LOGICAL ranks that live as threads
// of one process on ONE device (test infrastructure: RCCL refuses two ranks on a device,
// and the driver's 8-GPU node is not available to the tests).  It lets the tests run
// gec_group_allgather_decode's full flow -- exchange, per-rank range reconstruct, range
// pack / second exchange / unpack -- for world sizes 2..8 on the single GPU of a gpurun box.
//
// all-gather = host barrier, then every rank copies every rank's send buffer into its own
// receive buffer (device-to-device on its stream), then a second barrier so that nobody
// reuses a send buffer another rank is still reading.
#include <hip/hip_runtime.h>

#include <condition_variable>
#include <cstddef>
#include <mutex>
#include <vector>

namespace {
struct Loopback {
	int n;
	std::mutex mu;
	std::condition_variable cv;
	int waiting = 0;
	unsigned long generation = 0;
	std::vector<const void *> send;
	struct RankCtx {
		Loopback *lb;
		int rank;
	};
	std::vector<RankCtx> ctx;

	void barrier()
	{
		std::unique_lock<std::mutex> g(mu);
		const unsigned long gen = generation;
		if (++waiting == n) {
			waiting = 0;
			++generation;
			cv.notify_all();
		} else {
			cv.wait(g, [&] { return generation != gen; });
		}
	}
};
}  // namespace

extern "C" {

void *lb_create(int nranks)
{
	Loopback *lb = new Loopback();
	lb->n = nranks;
	lb->send.assign(nranks, nullptr);
	lb->ctx.resize(nranks);
	for (int r = 0; r < nranks; ++r)
		lb->ctx[r] = {lb, r};
	return lb;
}

void lb_destroy(void *p) { delete static_cast<Loopback *>(p); }

void *lb_rank_ctx(void *p, int rank) { return &static_cast<Loopback *>(p)->ctx[rank]; }

// matches gec_allgather_fn (include/garage_ec.h)
int lb_all_gather(void *ctx, const void *d_send, void *d_recv, size_t bytes, void *hip_stream)
{
	Loopback::RankCtx *rc = static_cast<Loopback::RankCtx *>(ctx);
	Loopback *lb = rc->lb;
	hipStream_t s = static_cast<hipStream_t>(hip_stream);
	if (hipStreamSynchronize(s) != hipSuccess)  // my send buffer is final
		return 1;
	lb->send[rc->rank] = d_send;
	lb->barrier();
	int err = 0;
	for (int q = 0; q < lb->n; ++q)
		if (hipMemcpyAsync(static_cast<char *>(d_recv) + (size_t)q * bytes, lb->send[q], bytes,
				   hipMemcpyDeviceToDevice, s) != hipSuccess)
			err = 1;
	if (hipStreamSynchronize(s) != hipSuccess)
		err = 1;
	lb->barrier();
	return err;
}

void *lb_all_gather_ptr(void) { return reinterpret_cast<void *>(&lb_all_gather); }

// matches gec_alltoall_fn: rank r's piece q goes to rank q's slot r
int lb_all_to_all(void *ctx, const void *d_send, void *d_recv, size_t bytes, void *hip_stream)
{
	Loopback::RankCtx *rc = static_cast<Loopback::RankCtx *>(ctx);
	Loopback *lb = rc->lb;
	hipStream_t s = static_cast<hipStream_t>(hip_stream);
	if (hipStreamSynchronize(s) != hipSuccess)
		return 1;
	lb->send[rc->rank] = d_send;
	lb->barrier();
	int err = 0;
	for (int q = 0; q < lb->n; ++q)
		if (hipMemcpyAsync(static_cast<char *>(d_recv) + (size_t)q * bytes,
				   static_cast<const char *>(lb->send[q]) + (size_t)rc->rank * bytes, bytes,
				   hipMemcpyDeviceToDevice, s) != hipSuccess)
			err = 1;
	if (hipStreamSynchronize(s) != hipSuccess)
		err = 1;
	lb->barrier();
	return err;
}

void *lb_all_to_all_ptr(void) { return reinterpret_cast<void *>(&lb_all_to_all); }

}  // extern "C"
