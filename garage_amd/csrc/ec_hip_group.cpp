// ec_hip_group.cpp -- gec_group_*: decode of an object whose shards are striped over several GPUs (BASELINE config 5),
// the path's one real exchange step: all-gather (or all-to-all) of the survivors over RCCL / a caller transport, every
// rank rebuilds its byte range of the missing shards, the rebuilt ranges are exchanged.  Host code only; the pack /
// unpack kernels are launched through ec_hip_launch.hip.
// A group over a GEC_BACKEND_CPU codec and a caller transport runs the SAME steps on host buffers (host_* below: the
// range split, the packs and unpacks restated as loops, the rebuild on the codec's own host data path): ranks that
// have lost their GPU stay in the group, and the CPU suite drives this file's exchange logic with N > 1 real
// processes (tests/test_group_multiprocess.py).
#include "ec_hip.hpp"

#include <rccl/rccl.h>  // types only: RCCL itself is resolved with dlopen

#include <dlfcn.h>

#include <algorithm>

using namespace gecimpl;

// RCCL entry points, resolved on first use (the library must load on hosts without RCCL,
// and inside a PyTorch process it must bind to the RCCL torch already loaded).
namespace {
struct Rccl {
	void *handle = nullptr;
	decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
	decltype(&ncclCommInitRank) CommInitRank = nullptr;
	decltype(&ncclCommDestroy) CommDestroy = nullptr;
	decltype(&ncclAllGather) AllGather = nullptr;
	decltype(&ncclSend) Send = nullptr;
	decltype(&ncclRecv) Recv = nullptr;
	decltype(&ncclGroupStart) GroupStart = nullptr;
	decltype(&ncclGroupEnd) GroupEnd = nullptr;
	decltype(&ncclGetErrorString) GetErrorString = nullptr;
	std::string error;
};

const Rccl &rccl()
{
	static const Rccl r = [] {
		Rccl x;
		std::vector<std::string> names;
		if (!env().rccl_lib.empty())
			names.push_back(env().rccl_lib);  // an explicit GEC_RCCL_LIB is authoritative: no fallback to another RCCL
		else
			names.insert(names.end(), {"librccl.so.1", "librccl.so"});
		for (const std::string &n : names) {
			x.handle = dlopen(n.c_str(), RTLD_NOW | RTLD_GLOBAL);
			if (x.handle)
				break;
			const char *de = dlerror();
			x.error += n + ": " + (de ? de : "?") + "; ";
		}
		if (!x.handle)
			return x;
		x.GetUniqueId = reinterpret_cast<decltype(x.GetUniqueId)>(dlsym(x.handle, "ncclGetUniqueId"));
		x.CommInitRank = reinterpret_cast<decltype(x.CommInitRank)>(dlsym(x.handle, "ncclCommInitRank"));
		x.CommDestroy = reinterpret_cast<decltype(x.CommDestroy)>(dlsym(x.handle, "ncclCommDestroy"));
		x.AllGather = reinterpret_cast<decltype(x.AllGather)>(dlsym(x.handle, "ncclAllGather"));
		x.GetErrorString = reinterpret_cast<decltype(x.GetErrorString)>(dlsym(x.handle, "ncclGetErrorString"));
		x.Send = reinterpret_cast<decltype(x.Send)>(dlsym(x.handle, "ncclSend"));
		x.Recv = reinterpret_cast<decltype(x.Recv)>(dlsym(x.handle, "ncclRecv"));
		x.GroupStart = reinterpret_cast<decltype(x.GroupStart)>(dlsym(x.handle, "ncclGroupStart"));
		x.GroupEnd = reinterpret_cast<decltype(x.GroupEnd)>(dlsym(x.handle, "ncclGroupEnd"));
		if (!x.GetUniqueId || !x.CommInitRank || !x.CommDestroy || !x.AllGather || !x.GetErrorString || !x.Send || !x.Recv ||
		    !x.GroupStart || !x.GroupEnd) {
			x.error = "RCCL library lacks a required symbol";
			x.handle = nullptr;
		}
		return x;
	}();
	return r;
}
}  // namespace

extern "C" size_t gec_group_slots(const gec_group *g);

struct gec_group {
	const gec_codec *c = nullptr;
	int rank = 0, nranks = 1;
	gec_allgather_fn all_gather = nullptr;
	gec_alltoall_fn all_to_all = nullptr;
	void *ctx = nullptr;
	ncclComm_t comm = nullptr;  // RCCL transport only
	// all-to-all exchange scratch
	uint8_t *d_a2a_send = nullptr, *d_a2a_recv = nullptr;
	size_t a2a_cap = 0;
	uint64_t bytes_exchanged = 0;  // bytes this rank RECEIVED from other ranks in the last decode call
	// scratch for the exchange of the rebuilt ranges (step 3)
	uint8_t *d_send = nullptr, *d_recv = nullptr;
	size_t send_cap = 0, recv_cap = 0;
	// peer-pointer exchange: pointer tables of the one decode launch, and the tokens of its barriers
	uint8_t *d_tab = nullptr, *d_tok = nullptr;
	size_t tab_cap = 0;
	std::vector<int> peer_enabled;  // devices hipDeviceEnablePeerAccess has been called for
	// what d_tab holds: the pointer tables of the last peer decode.  A caller that decodes batch after batch out of the same slot
	// buffers (same peers, pattern, geometry, destination) finds them in place: the call is then the launch and the barriers.
	struct PeerKey {
		std::vector<const void *> slots;
		std::vector<uint8_t> present;
		size_t nobjects = 0, S = 0;
		int data_only = 0, complete = 0;
		const void *rebuilt = nullptr, *stream = nullptr;  // (the stream: the tables' upload is ordered on the stream of the call that made them)
		bool operator==(const PeerKey &o) const
		{
			return slots == o.slots && present == o.present && nobjects == o.nobjects && S == o.S && data_only == o.data_only &&
			       complete == o.complete && rebuilt == o.rebuilt && stream == o.stream;
		}
	};
	PeerKey peer_key;
	bool peer_key_valid = false;
	uint64_t peer_bytes_cached = 0;  // bytes_exchanged of the cached table's launch
};

namespace {

int rccl_all_gather(void *ctx, const void *d_send, void *d_recv, size_t bytes, void *hip_stream)
{
	gec_group *g = static_cast<gec_group *>(ctx);
	ncclResult_t r = rccl().AllGather(d_send, d_recv, bytes, ncclUint8, g->comm, static_cast<hipStream_t>(hip_stream));
	if (r != ncclSuccess)
		return fail(GEC_E_DEVICE, std::string("ncclAllGather: ") + rccl().GetErrorString(r));
	return GEC_OK;
}

// all-to-all over RCCL: one grouped ncclSend/ncclRecv pair per peer (xGMI is a full mesh: every pair has its own link)
int rccl_all_to_all(void *ctx, const void *d_send, void *d_recv, size_t bytes, void *hip_stream)
{
	gec_group *g = static_cast<gec_group *>(ctx);
	const Rccl &R = rccl();
	hipStream_t s = static_cast<hipStream_t>(hip_stream);
	// One send / receive pair per peer and per piece of at most 512 MiB: a single ncclSend / ncclRecv of 2^30 bytes or more
	// delivers garbage (RCCL 2.26 on one rank, found by bench.py's check of the timed batch in round 6: 250 objects of config 5
	// -- 1 048 640 000 bytes to "every" peer of a group of one -- decode, 256 -- 1 073 807 360 -- do not; profiles/r06_striped.txt).
	// At N = 8 a peer's share of config 5 is 134 MB: one piece, as before.
	constexpr size_t kPiece = 512ull << 20;
	ncclResult_t r = R.GroupStart();
	for (int q = 0; q < g->nranks && r == ncclSuccess; ++q)
		for (size_t off = 0; off < bytes && r == ncclSuccess; off += kPiece) {
			const size_t n = std::min(kPiece, bytes - off);
			r = R.Send(static_cast<const uint8_t *>(d_send) + (size_t)q * bytes + off, n, ncclUint8, q, g->comm, s);
			if (r == ncclSuccess)
				r = R.Recv(static_cast<uint8_t *>(d_recv) + (size_t)q * bytes + off, n, ncclUint8, q, g->comm, s);
		}
	ncclResult_t e = R.GroupEnd();
	if (r == ncclSuccess)
		r = e;
	if (r != ncclSuccess)
		return fail(GEC_E_DEVICE, std::string("ncclSend/ncclRecv: ") + R.GetErrorString(r));
	return GEC_OK;
}

}  // namespace


// ---------------------------------------------------------------- the same steps on host buffers (CPU codec)
namespace {

int check_host_layout(const void *p, size_t S)
{
	if (S == 0)
		return fail(GEC_E_EMPTY_SHARD, "shard length is 0");
	if (S % 64 != 0)
		return fail(GEC_E_INCORRECT_SHARD_SIZE, "S must be a multiple of 64");
	if (!p)
		return fail(GEC_E_INVALID_ARG, "NULL buffer");
	return GEC_OK;
}

int host_scratch(uint8_t **p, size_t *cap, size_t bytes)
{
	if (bytes <= *cap)
		return GEC_OK;
	std::free(*p);
	*p = static_cast<uint8_t *>(std::calloc(bytes ? bytes : 1, 1));  // zeroed: pad columns are defined bytes on the wire
	*cap = *p ? bytes : 0;
	return *p ? GEC_OK : fail(GEC_E_NOMEM, "group scratch");
}

inline size_t range_lo(size_t cols, size_t r, size_t N) { return cols * r / N; }

int host_allgather_decode(gec_group *g, size_t nobjects, const uint8_t *local, size_t S, const uint8_t *present, int data_only, int complete,
			  uint8_t *gathered)
{
	const gec_codec *c = g->c;
	const size_t n = (size_t)c->k + c->m, N = (size_t)g->nranks, slots = gec_group_slots(g);
	int rc = check_host_layout(local, S);
	if (!rc)
		rc = check_host_layout(gathered, S);
	if (rc)
		return rc;
	std::shared_ptr<const Plan> plan;
	rc = get_plan(c, present, data_only != 0, plan);
	if (rc)
		return rc;
	// (1) every rank's slot buffer to everybody
	const size_t per_rank = nobjects * slots * S;
	g->bytes_exchanged = per_rank * (N - 1);
	rc = g->all_gather(g->ctx, local, gathered, per_rank, nullptr);
	if (rc)
		return rc > 0 ? fail(GEC_E_DEVICE, "all_gather transport failed") : rc;
	if (plan->missing.empty())
		return GEC_OK;
	// (2) my byte range of every missing shard, in place in the gathered buffer, on the codec's own host data path
	std::vector<size_t> shard_off(n);
	for (size_t j = 0; j < n; ++j)
		shard_off[j] = (j % N) * per_rank + (j / N) * S;
	const size_t cols = S / 16, lo = range_lo(cols, g->rank, N), my_cols = range_lo(cols, g->rank + 1, N) - lo;
	const size_t nmiss = plan->missing.size();
	if (my_cols) {
		std::vector<const uint8_t *> sp(nobjects * n, nullptr);
		std::vector<uint8_t *> op(nobjects * n, nullptr);
		for (size_t o = 0; o < nobjects; ++o) {
			for (size_t j = 0; j < n; ++j)
				if (present[j])
					sp[o * n + j] = gathered + shard_off[j] + o * slots * S + lo * 16;
			for (int j : plan->missing)
				op[o * n + j] = gathered + shard_off[j] + o * slots * S + lo * 16;
		}
		rc = c->be->reconstruct_batch(nobjects, sp.data(), op.data(), my_cols * 16, data_only, nullptr, nullptr);
		if (rc)
			return rc;
	}
	if (!complete || N == 1)
		return GEC_OK;
	// (3) exchange the rebuilt ranges: [nmiss][nobj][max_cols], ranges differ by at most one column
	size_t max_cols = 0;
	for (size_t r = 0; r < N; ++r)
		max_cols = std::max(max_cols, range_lo(cols, r + 1, N) - range_lo(cols, r, N));
	const size_t send_bytes = nmiss * nobjects * max_cols * 16;
	rc = host_scratch(&g->d_send, &g->send_cap, send_bytes);
	if (!rc)
		rc = host_scratch(&g->d_recv, &g->recv_cap, send_bytes * N);
	if (rc)
		return rc;
	for (size_t i = 0; i < nmiss; ++i)  // range_pack
		for (size_t o = 0; o < nobjects; ++o)
			std::memcpy(g->d_send + (i * nobjects + o) * max_cols * 16, gathered + shard_off[plan->missing[i]] + o * slots * S + lo * 16,
				    my_cols * 16);
	g->bytes_exchanged += send_bytes * (N - 1);
	rc = g->all_gather(g->ctx, g->d_send, g->d_recv, send_bytes, nullptr);
	if (rc)
		return rc > 0 ? fail(GEC_E_DEVICE, "all_gather transport failed") : rc;
	for (size_t r = 0; r < N; ++r) {  // range_unpack
		if (r == (size_t)g->rank)
			continue;
		const size_t rlo = range_lo(cols, r, N), rn = range_lo(cols, r + 1, N) - rlo;
		for (size_t i = 0; i < nmiss; ++i)
			for (size_t o = 0; o < nobjects; ++o)
				std::memcpy(gathered + shard_off[plan->missing[i]] + o * slots * S + rlo * 16,
					    g->d_recv + ((r * nmiss + i) * nobjects + o) * max_cols * 16, rn * 16);
	}
	return GEC_OK;
}

int host_alltoall_decode(gec_group *g, size_t nobjects, const uint8_t *local, size_t S, const uint8_t *present, int data_only, int complete,
			 uint8_t *rebuilt)
{
	const gec_codec *c = g->c;
	const size_t k = c->k, n = (size_t)c->k + c->m, N = (size_t)g->nranks, slots = gec_group_slots(g);
	int rc = check_host_layout(local, S);
	if (rc)
		return rc;
	std::shared_ptr<const Plan> plan;
	rc = get_plan(c, present, data_only != 0, plan);
	if (rc)
		return rc;
	g->bytes_exchanged = 0;
	const size_t nmiss = plan->missing.size();
	if (nmiss == 0)
		return GEC_OK;
	std::vector<std::vector<int>> valid_of(N);
	for (size_t t = 0; t < k; ++t)
		valid_of[plan->valid[t] % N].push_back(plan->valid[t]);
	size_t nvs_max = 0;
	for (auto &v : valid_of)
		nvs_max = std::max(nvs_max, v.size());
	const size_t cols = S / 16;
	size_t max_cols = 0;
	for (size_t r = 0; r < N; ++r)
		max_cols = std::max(max_cols, range_lo(cols, r + 1, N) - range_lo(cols, r, N));
	const size_t my_lo = range_lo(cols, g->rank, N), my_cols = range_lo(cols, g->rank + 1, N) - my_lo;
	const size_t per_peer = nvs_max * nobjects * max_cols * 16, packed_bytes = nmiss * nobjects * max_cols * 16;
	size_t cap2 = g->a2a_cap;
	rc = host_scratch(&g->d_a2a_send, &g->a2a_cap, N * per_peer);
	if (!rc)
		rc = host_scratch(&g->d_a2a_recv, &cap2, N * per_peer);
	if (!rc)
		rc = host_scratch(&g->d_send, &g->send_cap, packed_bytes);
	if (!rc)
		rc = host_scratch(&g->d_recv, &g->recv_cap, packed_bytes * N);
	if (rc)
		return rc;
	// (1) a2a_pack: for every peer, that peer's byte range of my valid shards
	const std::vector<int> &mine = valid_of[g->rank];
	for (size_t peer = 0; peer < N; ++peer) {
		const size_t plo = range_lo(cols, peer, N), pn = range_lo(cols, peer + 1, N) - plo;
		for (size_t vs = 0; vs < mine.size(); ++vs)
			for (size_t o = 0; o < nobjects; ++o)
				std::memcpy(g->d_a2a_send + ((peer * nvs_max + vs) * nobjects + o) * max_cols * 16,
					    local + o * slots * S + (size_t)(mine[vs] / N) * S + plo * 16, pn * 16);
	}
	// (2) the exchange step
	g->bytes_exchanged = per_peer * (N - 1);
	rc = g->all_to_all(g->ctx, g->d_a2a_send, g->d_a2a_recv, per_peer, nullptr);
	if (rc)
		return rc > 0 ? fail(GEC_E_DEVICE, "all_to_all transport failed") : rc;
	// (3) my byte range of every missing shard from the received ranges
	if (my_cols) {
		std::vector<const uint8_t *> sp(nobjects * n, nullptr);
		std::vector<uint8_t *> op(nobjects * n, nullptr);
		for (size_t o = 0; o < nobjects; ++o) {
			for (size_t t = 0; t < k; ++t) {
				const int v = plan->valid[t];
				const size_t owner = v % N;
				const size_t vs = std::find(valid_of[owner].begin(), valid_of[owner].end(), v) - valid_of[owner].begin();
				sp[o * n + v] = g->d_a2a_recv + ((owner * nvs_max + vs) * nobjects + o) * max_cols * 16;
			}
			for (size_t i = 0; i < nmiss; ++i)
				op[o * n + plan->missing[i]] = g->d_send + (i * nobjects + o) * max_cols * 16;
		}
		rc = c->be->reconstruct_batch(nobjects, sp.data(), op.data(), my_cols * 16, data_only, nullptr, nullptr);
		if (rc)
			return rc;
	}
	// (4) the rebuilt ranges: mine only, or everybody's after a (small) all-gather
	auto unpack = [&](const uint8_t *packed, size_t r) {
		const size_t rlo = range_lo(cols, r, N), rn = range_lo(cols, r + 1, N) - rlo;
		for (size_t i = 0; i < nmiss; ++i)
			for (size_t o = 0; o < nobjects; ++o)
				std::memcpy(rebuilt + (i * nobjects + o) * S + rlo * 16, packed + (i * nobjects + o) * max_cols * 16, rn * 16);
	};
	if (complete && N > 1) {
		g->bytes_exchanged += packed_bytes * (N - 1);
		rc = g->all_gather(g->ctx, g->d_send, g->d_recv, packed_bytes, nullptr);
		if (rc)
			return rc > 0 ? fail(GEC_E_DEVICE, "all_gather transport failed") : rc;
		for (size_t r = 0; r < N; ++r)
			unpack(g->d_recv + r * packed_bytes, r);
	} else {
		unpack(g->d_send, g->rank);
	}
	return GEC_OK;
}

// gec_group_peer_decode on host buffers (CPU codec): the "peers' slot buffers" are plain pointers of this address space
// (ranks that are threads of one process, or shared memory the caller mapped)
int host_peer_decode(gec_group *g, size_t nobjects, const uint8_t *const *peer_slots, size_t S, const uint8_t *present, int data_only,
		     int complete, uint8_t *rebuilt)
{
	const gec_codec *c = g->c;
	const size_t k = c->k, n = (size_t)c->k + c->m, N = (size_t)g->nranks, slots = gec_group_slots(g);
	for (size_t q = 0; q < N; ++q) {
		int rc = check_host_layout(peer_slots[q], S);
		if (rc)
			return rc;
	}
	std::shared_ptr<const Plan> plan;
	int rc = get_plan(c, present, data_only != 0, plan);
	if (rc)
		return rc;
	g->bytes_exchanged = 0;
	const size_t nmiss = plan->missing.size();
	if (nmiss == 0)
		return GEC_OK;
	const size_t cols = S / 16;
	size_t max_cols = 0;
	for (size_t r = 0; r < N; ++r)
		max_cols = std::max(max_cols, range_lo(cols, r + 1, N) - range_lo(cols, r, N));
	const size_t my_lo = range_lo(cols, g->rank, N), my_cols = range_lo(cols, g->rank + 1, N) - my_lo;
	const size_t packed_bytes = nmiss * nobjects * max_cols * 16;
	size_t tok_cap = g->tab_cap;
	rc = host_scratch(&g->d_send, &g->send_cap, packed_bytes);
	if (!rc)
		rc = host_scratch(&g->d_recv, &g->recv_cap, packed_bytes * N);
	if (!rc)
		rc = host_scratch(&g->d_tok, &tok_cap, 16 * (N + 1));
	if (rc)
		return rc;
	g->tab_cap = tok_cap;
	// every rank's slot buffer is final: the barrier is a 16-byte all-gather through the group's transport
	rc = g->all_gather(g->ctx, g->d_tok, g->d_tok + 16, 16, nullptr);
	if (rc)
		return rc > 0 ? fail(GEC_E_DEVICE, "all_gather transport failed") : rc;
	if (my_cols) {
		std::vector<const uint8_t *> sp(nobjects * n, nullptr);
		std::vector<uint8_t *> op(nobjects * n, nullptr);
		for (size_t o = 0; o < nobjects; ++o) {
			for (size_t t = 0; t < k; ++t) {
				const size_t v = (size_t)plan->valid[t];
				sp[o * n + v] = peer_slots[v % N] + o * slots * S + (v / N) * S + my_lo * 16;
				if (v % N != (size_t)g->rank)
					g->bytes_exchanged += my_cols * 16;
			}
			for (size_t i = 0; i < nmiss; ++i)
				op[o * n + plan->missing[i]] = g->d_send + (i * nobjects + o) * max_cols * 16;
		}
		rc = c->be->reconstruct_batch(nobjects, sp.data(), op.data(), my_cols * 16, data_only, nullptr, nullptr);
		if (rc)
			return rc;
	}
	auto unpack = [&](const uint8_t *packed, size_t r) {
		const size_t rlo = range_lo(cols, r, N), rn = range_lo(cols, r + 1, N) - rlo;
		for (size_t i = 0; i < nmiss; ++i)
			for (size_t o = 0; o < nobjects; ++o)
				std::memcpy(rebuilt + (i * nobjects + o) * S + rlo * 16, packed + (i * nobjects + o) * max_cols * 16, rn * 16);
	};
	if (complete && N > 1) {
		g->bytes_exchanged += packed_bytes * (N - 1);
		rc = g->all_gather(g->ctx, g->d_send, g->d_recv, packed_bytes, nullptr);  // (also: everybody is done reading)
		if (rc)
			return rc > 0 ? fail(GEC_E_DEVICE, "all_gather transport failed") : rc;
		for (size_t r = 0; r < N; ++r)
			unpack(g->d_recv + r * packed_bytes, r);
	} else {
		unpack(g->d_send, g->rank);
		rc = g->all_gather(g->ctx, g->d_tok, g->d_tok + 16, 16, nullptr);  // nobody's slot buffer changes while a peer still reads it
		if (rc)
			return rc > 0 ? fail(GEC_E_DEVICE, "all_gather transport failed") : rc;
	}
	return GEC_OK;
}

}  // namespace

extern "C" {

int gec_group_unique_id(uint8_t id[GEC_GROUP_ID_BYTES])
try {
	static_assert(sizeof(ncclUniqueId) == GEC_GROUP_ID_BYTES, "GEC_GROUP_ID_BYTES must equal sizeof(ncclUniqueId)");
	if (!id)
		return fail(GEC_E_INVALID_ARG, "NULL id");
	const Rccl &R = rccl();
	if (!R.handle)
		return fail(GEC_E_DEVICE, "RCCL is not available: " + R.error);
	ncclUniqueId u;
	ncclResult_t r = R.GetUniqueId(&u);
	if (r != ncclSuccess)
		return fail(GEC_E_DEVICE, std::string("ncclGetUniqueId: ") + R.GetErrorString(r));
	std::memcpy(id, &u, sizeof(u));
	return GEC_OK;
}
GEC_CATCH

static int check_dev_layout(const void *p, size_t stride, size_t S, size_t need)
{
	if (S == 0)
		return fail(GEC_E_EMPTY_SHARD, "shard length is 0");
	if (S % 64 != 0)
		return fail(GEC_E_INCORRECT_SHARD_SIZE, "S must be a multiple of 64");
	if (!p)
		return fail(GEC_E_INVALID_ARG, "NULL device pointer");
	if (reinterpret_cast<uintptr_t>(p) % 16 != 0 || stride % 16 != 0)
		return fail(GEC_E_INVALID_ARG, "device pointer/stride must be 16-byte aligned");
	if (stride < need)
		return fail(GEC_E_INCORRECT_SHARD_SIZE, "stride smaller than the shards it must hold");
	return GEC_OK;
}

static int group_new(const gec_codec *c, int rank, int nranks, gec_group **out, std::unique_ptr<gec_group> &g)
{
	if (!out)
		return fail(GEC_E_INVALID_ARG, "NULL out");
	*out = nullptr;
	if (!c)
		return fail(GEC_E_INVALID_ARG, "NULL codec");
	if (nranks < 1 || rank < 0 || rank >= nranks)
		return fail(GEC_E_INVALID_ARG, "need 0 <= rank < nranks");
	g.reset(new (std::nothrow) gec_group());
	if (!g)
		return fail(GEC_E_NOMEM, "alloc group");
	g->c = c;
	g->rank = rank;
	g->nranks = nranks;
	return GEC_OK;
}

int gec_group_create(const gec_codec *c, int rank, int nranks, const uint8_t id[GEC_GROUP_ID_BYTES], gec_group **out)
try {
	std::unique_ptr<gec_group> g;
	int rc = group_new(c, rank, nranks, out, g);
	if (rc)
		return rc;
	if (!id)
		return fail(GEC_E_INVALID_ARG, "NULL id");
	if (c->backend != GEC_BACKEND_HIP)
		return fail(GEC_E_DEVICE, "an RCCL group needs a GEC_BACKEND_HIP codec (RCCL moves device memory); a CPU codec takes a caller transport");
	const Rccl &R = rccl();
	if (!R.handle)
		return fail(GEC_E_DEVICE, "RCCL is not available: " + R.error);
	DeviceGuard dg(c->device);
	if (!dg.ok)
		return fail(GEC_E_DEVICE, "hipSetDevice failed");
	ncclUniqueId u;
	std::memcpy(&u, id, sizeof(u));
	ncclResult_t r = R.CommInitRank(&g->comm, nranks, u, rank);
	if (r != ncclSuccess)
		return fail(GEC_E_DEVICE, std::string("ncclCommInitRank: ") + R.GetErrorString(r));
	g->all_gather = rccl_all_gather;
	g->all_to_all = rccl_all_to_all;
	g->ctx = g.get();
	*out = g.release();
	return GEC_OK;
}
GEC_CATCH

int gec_group_create_with_transport2(const gec_codec *c, int rank, int nranks, gec_allgather_fn all_gather,
				     gec_alltoall_fn all_to_all, void *ctx, gec_group **out)
try {
	std::unique_ptr<gec_group> g;
	int rc = group_new(c, rank, nranks, out, g);
	if (rc)
		return rc;
	if (!all_gather)
		return fail(GEC_E_INVALID_ARG, "NULL all_gather");
	g->all_gather = all_gather;
	g->all_to_all = all_to_all;
	g->ctx = ctx;
	*out = g.release();
	return GEC_OK;
}
GEC_CATCH

int gec_group_create_with_transport(const gec_codec *c, int rank, int nranks, gec_allgather_fn all_gather, void *ctx,
				    gec_group **out)
try {
	return gec_group_create_with_transport2(c, rank, nranks, all_gather, nullptr, ctx, out);
}
GEC_CATCH

void gec_group_destroy(gec_group *g)
{
	if (!g)
		return;
	if (g->c->backend != GEC_BACKEND_HIP) {  // host scratch
		std::free(g->d_send);
		std::free(g->d_recv);
		std::free(g->d_a2a_send);
		std::free(g->d_a2a_recv);
		std::free(g->d_tok);
		delete g;
		return;
	}
	{
		DeviceGuard dg(g->c->device);
		if (g->d_send)
			(void)hipFree(g->d_send);
		if (g->d_recv)
			(void)hipFree(g->d_recv);
		if (g->d_a2a_send)
			(void)hipFree(g->d_a2a_send);
		if (g->d_a2a_recv)
			(void)hipFree(g->d_a2a_recv);
		if (g->d_tab)
			(void)hipFree(g->d_tab);
		if (g->d_tok)
			(void)hipFree(g->d_tok);
		if (g->comm)
			(void)rccl().CommDestroy(g->comm);
	}
	delete g;
}

int gec_group_rank(const gec_group *g) { return g ? g->rank : -1; }
int gec_group_size(const gec_group *g) { return g ? g->nranks : 0; }
size_t gec_group_slots(const gec_group *g)
{
	return g ? ((size_t)(g->c->k + g->c->m) + g->nranks - 1) / g->nranks : 0;
}

int gec_group_allgather_decode(gec_group *g, size_t nobjects, const void *d_local_slots, size_t S,
			       const uint8_t *present, int data_only, int complete, void *d_gathered, void *hip_stream)
try {
	if (!g || !present)
		return fail(GEC_E_INVALID_ARG, "NULL argument");
	if (nobjects == 0)
		return GEC_OK;
	const gec_codec *c = g->c;
	if (c->backend != GEC_BACKEND_HIP)  // host buffers, same steps (hip_stream is ignored)
		return host_allgather_decode(g, nobjects, static_cast<const uint8_t *>(d_local_slots), S, present, data_only, complete,
					     static_cast<uint8_t *>(d_gathered));
	const size_t n = (size_t)c->k + c->m, N = (size_t)g->nranks, slots = gec_group_slots(g);
	int rc = check_dev_layout(d_local_slots, slots * S, S, slots * S);
	if (rc)
		return rc;
	rc = check_dev_layout(d_gathered, slots * S, S, slots * S);
	if (rc)
		return rc;
	hipStream_t stream = static_cast<hipStream_t>(hip_stream);
	DeviceGuard dg(c->device);
	if (!dg.ok)
		return fail(GEC_E_DEVICE, "hipSetDevice failed");
	std::shared_ptr<const Plan> plan;  // before the exchange: a bad pattern fails on every rank alike, no rank hangs
	rc = get_plan(c, present, data_only != 0, plan);
	if (rc)
		return rc;
	// (1) the exchange step: every rank's slot buffer to everybody
	const size_t per_rank = nobjects * slots * S;
	g->bytes_exchanged = per_rank * (N - 1);
	rc = g->all_gather(g->ctx, d_local_slots, d_gathered, per_rank, hip_stream);
	if (rc)
		return rc > 0 ? fail(GEC_E_DEVICE, "all_gather transport failed") : rc;
	if (plan->missing.empty())
		return GEC_OK;
	// (2) my byte range of every missing shard, in place in the gathered buffer
	std::vector<size_t> shard_off(n);
	for (size_t j = 0; j < n; ++j)
		shard_off[j] = (j % N) * per_rank + (j / N) * S;
	const size_t cols = S / 16;
	auto range_lo = [&](size_t r) { return cols * r / N; };
	const size_t lo = range_lo(g->rank), my_cols = range_lo(g->rank + 1) - lo;
	if (my_cols) {
		rc = reconstruct_dev(c, nobjects, static_cast<uint8_t *>(d_gathered), slots * S, shard_off.data(), present,
				     data_only != 0, lo * 16, my_cols * 16, stream);
		if (rc)
			return rc;
	}
	if (!complete || N == 1)
		return GEC_OK;
	// (3) exchange the rebuilt ranges (ranges differ by at most one column: pad to the longest)
	size_t max_cols = 0;
	for (size_t r = 0; r < N; ++r)
		max_cols = std::max(max_cols, range_lo(r + 1) - range_lo(r));
	const size_t nmiss = plan->missing.size();
	const size_t send_bytes = nmiss * nobjects * max_cols * 16;
	if (nobjects > 0xffffffffull || cols > 0xffffffffull)
		return fail(GEC_E_INVALID_ARG, "batch too large for one call");
	if (send_bytes > g->send_cap || send_bytes * N > g->recv_cap) {
		HIP_TRY(hipStreamSynchronize(stream));  // earlier calls may still use the old buffers
		if (g->d_send)
			(void)hipFree(g->d_send);
		if (g->d_recv)
			(void)hipFree(g->d_recv);
		g->d_send = g->d_recv = nullptr;
		g->send_cap = g->recv_cap = 0;
		HIP_TRY(hipMalloc(reinterpret_cast<void **>(&g->d_send), send_bytes));
		HIP_TRY(hipMalloc(reinterpret_cast<void **>(&g->d_recv), send_bytes * N));
		HIP_TRY(hipMemsetAsync(g->d_send, 0, send_bytes, stream));  // pad columns: defined bytes on the wire
		g->send_cap = send_bytes;
		g->recv_cap = send_bytes * N;
	}
	gec::RangeArgs ra;
	std::memset(&ra, 0, sizeof(ra));
	ra.gathered = static_cast<uint8_t *>(d_gathered);
	ra.obj_stride = slots * S;
	ra.nobj = (uint32_t)nobjects;
	ra.nmiss = (uint32_t)nmiss;
	ra.cols = (uint32_t)cols;
	ra.max_cols = (uint32_t)max_cols;
	ra.world = (uint32_t)N;
	ra.rank = (uint32_t)g->rank;
	for (size_t i = 0; i < nmiss; ++i)
		ra.shard_off[i] = shard_off[plan->missing[i]];
	if (my_cols) {
		ra.packed = g->d_send;
		rc = launch_range_pack(ra, nmiss * nobjects * my_cols, stream);
		if (rc)
			return rc;
	}
	g->bytes_exchanged += send_bytes * (N - 1);
	rc = g->all_gather(g->ctx, g->d_send, g->d_recv, send_bytes, hip_stream);
	if (rc)
		return rc > 0 ? fail(GEC_E_DEVICE, "all_gather transport failed") : rc;
	ra.packed = g->d_recv;
	return launch_range_unpack(ra, send_bytes / 16 * N, stream);
}
GEC_CATCH

uint64_t gec_group_bytes_exchanged(const gec_group *g) { return g ? g->bytes_exchanged : 0; }

int gec_group_alltoall_decode(gec_group *g, size_t nobjects, const void *d_local_slots, size_t S, const uint8_t *present,
			      int data_only, int complete, void *d_rebuilt, void *hip_stream)
try {
	if (!g || !present || !d_rebuilt)
		return fail(GEC_E_INVALID_ARG, "NULL argument");
	if (!g->all_to_all)
		return fail(GEC_E_INVALID_ARG, "this group's transport has no all-to-all");
	if (nobjects == 0)
		return GEC_OK;
	const gec_codec *c = g->c;
	if (c->backend != GEC_BACKEND_HIP)
		return host_alltoall_decode(g, nobjects, static_cast<const uint8_t *>(d_local_slots), S, present, data_only, complete,
					    static_cast<uint8_t *>(d_rebuilt));
	const size_t k = c->k, N = (size_t)g->nranks, slots = gec_group_slots(g);
	int rc = check_dev_layout(d_local_slots, slots * S, S, slots * S);
	if (rc)
		return rc;
	if (reinterpret_cast<uintptr_t>(d_rebuilt) % 16)
		return fail(GEC_E_INVALID_ARG, "d_rebuilt must be 16-byte aligned");
	if (nobjects > 0xffffffffull || S / 16 > 0xffffffffull)
		return fail(GEC_E_INVALID_ARG, "batch too large for one call");
	hipStream_t stream = static_cast<hipStream_t>(hip_stream);
	DeviceGuard dg(c->device);
	if (!dg.ok)
		return fail(GEC_E_DEVICE, "hipSetDevice failed");
	std::shared_ptr<const Plan> plan;  // before the exchange: a bad pattern fails on every rank alike, no rank hangs
	rc = get_plan(c, present, data_only != 0, plan);
	if (rc)
		return rc;
	g->bytes_exchanged = 0;
	const size_t nmiss = plan->missing.size();
	if (nmiss == 0)
		return GEC_OK;
	// which of the k shards the decode reads live on which rank: shard v on rank v % N, local slot v / N;
	// vs index = position among that rank's valid shards
	std::vector<std::vector<int>> valid_of(N);
	for (size_t t = 0; t < k; ++t)
		valid_of[plan->valid[t] % N].push_back(plan->valid[t]);
	size_t nvs_max = 0;
	for (auto &v : valid_of)
		nvs_max = std::max(nvs_max, v.size());
	const size_t cols = S / 16;
	auto range_lo = [&](size_t r) { return cols * r / N; };
	size_t max_cols = 0;
	for (size_t r = 0; r < N; ++r)
		max_cols = std::max(max_cols, range_lo(r + 1) - range_lo(r));
	const size_t my_cols = range_lo(g->rank + 1) - range_lo(g->rank);
	const size_t per_peer = nvs_max * nobjects * max_cols * 16;
	const size_t packed_bytes = nmiss * nobjects * max_cols * 16;
	// scratch: [send N*per_peer][recv N*per_peer]; the rebuilt ranges reuse the group's d_send / d_recv
	if (N * per_peer > g->a2a_cap || packed_bytes > g->send_cap || packed_bytes * N > g->recv_cap) {
		HIP_TRY(hipStreamSynchronize(stream));
		for (uint8_t **p : {&g->d_a2a_send, &g->d_a2a_recv, &g->d_send, &g->d_recv})
			if (*p) {
				(void)hipFree(*p);
				*p = nullptr;
			}
		g->a2a_cap = g->send_cap = g->recv_cap = 0;
		HIP_TRY(hipMalloc(reinterpret_cast<void **>(&g->d_a2a_send), N * per_peer));
		HIP_TRY(hipMalloc(reinterpret_cast<void **>(&g->d_a2a_recv), N * per_peer));
		HIP_TRY(hipMalloc(reinterpret_cast<void **>(&g->d_send), packed_bytes));
		HIP_TRY(hipMalloc(reinterpret_cast<void **>(&g->d_recv), packed_bytes * N));
		HIP_TRY(hipMemsetAsync(g->d_a2a_send, 0, N * per_peer, stream));  // pad columns / unused vs slots: defined bytes on the wire
		HIP_TRY(hipMemsetAsync(g->d_send, 0, packed_bytes, stream));
		g->a2a_cap = N * per_peer;
		g->send_cap = packed_bytes;
		g->recv_cap = packed_bytes * N;
	}
	// (1) pack: for every peer, that peer's byte range of my valid shards
	const std::vector<int> &mine = valid_of[g->rank];
	if (!mine.empty()) {
		gec::A2aArgs pa;
		std::memset(&pa, 0, sizeof(pa));
		pa.local = static_cast<const uint8_t *>(d_local_slots);
		pa.send = g->d_a2a_send;
		pa.obj_stride = slots * S;
		pa.nobj = (uint32_t)nobjects;
		pa.nvs = (uint32_t)mine.size();
		pa.nvs_max = (uint32_t)nvs_max;
		pa.cols = (uint32_t)cols;
		pa.max_cols = (uint32_t)max_cols;
		pa.world = (uint32_t)N;
		for (size_t i = 0; i < mine.size(); ++i)
			pa.slot_of[i] = (uint32_t)(mine[i] / N);
		rc = launch_a2a_pack(pa, N * mine.size() * nobjects * max_cols, stream);
		if (rc)
			return rc;
	}
	// (2) the exchange step: 1/N of the all-gather's bytes
	g->bytes_exchanged = per_peer * (N - 1);
	rc = g->all_to_all(g->ctx, g->d_a2a_send, g->d_a2a_recv, per_peer, hip_stream);
	if (rc)
		return rc > 0 ? fail(GEC_E_DEVICE, "all_to_all transport failed") : rc;
	// (3) my byte range of every missing shard, from the received ranges: input shard valid[t] sits at
	//     recv[(owner*nvs_max + vs)*nobj + obj][max_cols]; output i at d_send[(i*nobj + obj)][max_cols]
	if (my_cols) {
		std::vector<size_t> in_off(k), out_off(nmiss);
		for (size_t t = 0; t < k; ++t) {
			const int v = plan->valid[t];
			const size_t owner = v % N;
			const size_t vs = std::find(valid_of[owner].begin(), valid_of[owner].end(), v) - valid_of[owner].begin();
			in_off[t] = (owner * nvs_max + vs) * nobjects * max_cols * 16;
		}
		for (size_t i = 0; i < nmiss; ++i)
			out_off[i] = i * nobjects * max_cols * 16;
		rc = launch_apply(c, g->d_a2a_recv, max_cols * 16, g->d_send, max_cols * 16, nullptr, 0, my_cols * 16, nobjects,
				  in_off.data(), out_off.data(), (int)nmiss, plan->rows.v.data(), gec::MODE_STORE, stream);
		if (rc)
			return rc;
	}
	// (4) the rebuilt ranges: mine only, or everybody's after a (small) all-gather
	gec::RebuiltArgs ua;
	std::memset(&ua, 0, sizeof(ua));
	ua.rebuilt = static_cast<uint8_t *>(d_rebuilt);
	ua.nobj = (uint32_t)nobjects;
	ua.nmiss = (uint32_t)nmiss;
	ua.cols = (uint32_t)cols;
	ua.max_cols = (uint32_t)max_cols;
	ua.world = (uint32_t)N;
	if (complete && N > 1) {
		g->bytes_exchanged += packed_bytes * (N - 1);
		rc = g->all_gather(g->ctx, g->d_send, g->d_recv, packed_bytes, hip_stream);
		if (rc)
			return rc > 0 ? fail(GEC_E_DEVICE, "all_gather transport failed") : rc;
		ua.packed = g->d_recv;
		ua.first_rank = 0;
		ua.nranks_in = (uint32_t)N;
	} else {
		ua.packed = g->d_send;
		ua.first_rank = (uint32_t)g->rank;
		ua.nranks_in = 1;
	}
	return launch_rebuilt_unpack(ua, (size_t)ua.nranks_in * nmiss * nobjects * max_cols, stream);
}
GEC_CATCH

// ---------------------------------------------------------------- peer-pointer exchange
// The handle is hipIpcMemHandle_t of the ALLOCATION d_ptr lies in, followed by d_ptr's offset inside it: a tensor of a caching
// allocator (torch's, a Rust arena) rarely starts its allocation, and hipIpcOpenMemHandle maps allocations.
namespace {
std::mutex g_ipc_mu;
std::unordered_map<void *, void *> g_ipc_base;  // what gec_ipc_open returned -> the mapping's base (what must be closed)
}  // namespace

int gec_ipc_export(const void *d_ptr, uint8_t handle[GEC_IPC_HANDLE_BYTES])
try {
	static_assert(sizeof(hipIpcMemHandle_t) + 8 == GEC_IPC_HANDLE_BYTES, "GEC_IPC_HANDLE_BYTES must equal sizeof(hipIpcMemHandle_t) + 8");
	if (!d_ptr || !handle)
		return fail(GEC_E_INVALID_ARG, "NULL argument");
	hipDeviceptr_t base = nullptr;
	size_t size = 0;
	HIP_TRY(hipMemGetAddressRange(&base, &size, const_cast<void *>(d_ptr)));
	hipIpcMemHandle_t h;
	HIP_TRY(hipIpcGetMemHandle(&h, base));
	const uint64_t off = (uint64_t)(static_cast<const uint8_t *>(d_ptr) - static_cast<const uint8_t *>(base));
	std::memcpy(handle, &h, sizeof(h));
	std::memcpy(handle + sizeof(h), &off, 8);
	return GEC_OK;
}
GEC_CATCH

int gec_ipc_open(const uint8_t handle[GEC_IPC_HANDLE_BYTES], int device, void **d_ptr)
try {
	if (!handle || !d_ptr)
		return fail(GEC_E_INVALID_ARG, "NULL argument");
	*d_ptr = nullptr;
	DeviceGuard dg(device);
	if (!dg.ok)
		return fail(GEC_E_DEVICE, "hipSetDevice failed");
	hipIpcMemHandle_t h;
	uint64_t off = 0;
	std::memcpy(&h, handle, sizeof(h));
	std::memcpy(&off, handle + sizeof(h), 8);
	void *base = nullptr;
	HIP_TRY(hipIpcOpenMemHandle(&base, h, hipIpcMemLazyEnablePeerAccess));
	*d_ptr = static_cast<uint8_t *>(base) + off;
	std::lock_guard<std::mutex> lk(g_ipc_mu);
	g_ipc_base[*d_ptr] = base;
	return GEC_OK;
}
GEC_CATCH

int gec_ipc_close(void *d_ptr)
try {
	if (!d_ptr)
		return GEC_OK;
	void *base = nullptr;
	{
		std::lock_guard<std::mutex> lk(g_ipc_mu);
		auto it = g_ipc_base.find(d_ptr);
		if (it == g_ipc_base.end())
			return fail(GEC_E_INVALID_ARG, "not a pointer gec_ipc_open returned");
		base = it->second;
		g_ipc_base.erase(it);
	}
	HIP_TRY(hipIpcCloseMemHandle(base));
	return GEC_OK;
}
GEC_CATCH

int gec_group_peer_decode(gec_group *g, size_t nobjects, const void *const *d_peer_slots, size_t S, const uint8_t *present,
			  int data_only, int complete, void *d_rebuilt, void *hip_stream)
try {
	if (!g || !present || !d_rebuilt || !d_peer_slots)
		return fail(GEC_E_INVALID_ARG, "NULL argument");
	if (nobjects == 0)
		return GEC_OK;
	const gec_codec *c = g->c;
	const size_t k = c->k, N = (size_t)g->nranks, slots = gec_group_slots(g);
	for (size_t q = 0; q < N; ++q)
		if (!d_peer_slots[q])
			return fail(GEC_E_INVALID_ARG, "NULL peer slot buffer");
	if (c->backend != GEC_BACKEND_HIP)
		return host_peer_decode(g, nobjects, reinterpret_cast<const uint8_t *const *>(d_peer_slots), S, present, data_only, complete,
					static_cast<uint8_t *>(d_rebuilt));
	for (size_t q = 0; q < N; ++q) {
		int rc = check_dev_layout(d_peer_slots[q], slots * S, S, slots * S);
		if (rc)
			return rc;
	}
	if (reinterpret_cast<uintptr_t>(d_rebuilt) % 16)
		return fail(GEC_E_INVALID_ARG, "d_rebuilt must be 16-byte aligned");
	if (nobjects > 0xffffffffull || S / 16 > 0xffffffffull)
		return fail(GEC_E_INVALID_ARG, "batch too large for one call");
	hipStream_t stream = static_cast<hipStream_t>(hip_stream);
	DeviceGuard dg(c->device);
	if (!dg.ok)
		return fail(GEC_E_DEVICE, "hipSetDevice failed");
	std::shared_ptr<const Plan> plan;  // before anything is exchanged: a bad pattern fails on every rank alike, no rank hangs
	int rc = get_plan(c, present, data_only != 0, plan);
	if (rc)
		return rc;
	g->bytes_exchanged = 0;
	const size_t nmiss = plan->missing.size();
	if (nmiss == 0)
		return GEC_OK;
	gec_group::PeerKey key;
	key.slots.assign(d_peer_slots, d_peer_slots + N);
	key.present.assign(present, present + (size_t)c->k + c->m);
	key.nobjects = nobjects;
	key.S = S;
	key.data_only = data_only != 0;
	key.complete = complete != 0;
	key.rebuilt = d_rebuilt;
	key.stream = hip_stream;
	const bool cached = g->peer_key_valid && key == g->peer_key;
	// a peer's buffer that lives on another device of this process: make it addressable from the codec's device
	// (pointers opened with gec_ipc_open already are)
	for (size_t q = 0; q < N && !cached; ++q) {
		hipPointerAttribute_t at;
		if (hipPointerGetAttributes(&at, d_peer_slots[q]) != hipSuccess) {
			(void)hipGetLastError();
			continue;
		}
		const int dev = at.device;
		if (dev == c->device || std::find(g->peer_enabled.begin(), g->peer_enabled.end(), dev) != g->peer_enabled.end())
			continue;
		const hipError_t e = hipDeviceEnablePeerAccess(dev, 0);
		if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled)
			return fail(GEC_E_DEVICE, std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(e));
		(void)hipGetLastError();
		g->peer_enabled.push_back(dev);
	}
	const size_t cols = S / 16;
	auto range_lo = [&](size_t r) { return cols * r / N; };
	size_t max_cols = 0;
	for (size_t r = 0; r < N; ++r)
		max_cols = std::max(max_cols, range_lo(r + 1) - range_lo(r));
	const size_t my_lo = range_lo(g->rank), my_cols = range_lo(g->rank + 1) - my_lo;
	const size_t packed_bytes = nmiss * nobjects * max_cols * 16;
	const size_t tab_bytes = ptrs_dev_scratch_bytes(nobjects, k, (int)nmiss);
	if (packed_bytes > g->send_cap || packed_bytes * N > g->recv_cap || tab_bytes > g->tab_cap || !g->d_tok) {
		HIP_TRY(hipStreamSynchronize(stream));  // earlier calls may still use the old buffers
		g->peer_key_valid = false;
		for (uint8_t **p : {&g->d_send, &g->d_recv, &g->d_tab, &g->d_tok})
			if (*p) {
				(void)hipFree(*p);
				*p = nullptr;
			}
		g->send_cap = g->recv_cap = g->tab_cap = 0;
		HIP_TRY(hipMalloc(reinterpret_cast<void **>(&g->d_send), packed_bytes));
		HIP_TRY(hipMalloc(reinterpret_cast<void **>(&g->d_recv), packed_bytes * N));
		HIP_TRY(hipMalloc(reinterpret_cast<void **>(&g->d_tab), tab_bytes));
		HIP_TRY(hipMalloc(reinterpret_cast<void **>(&g->d_tok), 16 * (N + 1)));
		HIP_TRY(hipMemsetAsync(g->d_send, 0, packed_bytes, stream));  // pad columns: defined bytes on the wire
		HIP_TRY(hipMemsetAsync(g->d_tok, 0, 16 * (N + 1), stream));
		g->send_cap = packed_bytes;
		g->recv_cap = packed_bytes * N;
		g->tab_cap = tab_bytes;
	}
	// (0) every rank's slot buffer is final: a 16-byte all-gather through the group's transport is the barrier -- stream-ordered
	//     behind whatever filled the slots on each rank, ahead of the kernel that reads them here
	rc = g->all_gather(g->ctx, g->d_tok, g->d_tok + 16, 16, hip_stream);
	if (rc)
		return rc > 0 ? fail(GEC_E_DEVICE, "all_gather transport failed") : rc;
	// (1) ONE launch: my byte range of the k shards the decode reads -- wherever they live -- in, my range of every missing shard
	//     out.  No pack, no staging buffer, 1/N of the all-gather's bytes over each link.  When the rebuilt ranges are not
	//     exchanged afterwards (complete == 0, or a group of one) the launch stores straight into d_rebuilt: no unpack pass.
	const bool direct = !(complete && N > 1);
	// The k inputs of an object sit at the same place in every rank's slot buffer, objects slots*S apart: when the k addresses of
	// object 0 lie within 64 GiB of each other (always in a group of one; mappings of peers' buffers often do) the STRIDED kernel can
	// reach them all from one base with its 32-bit shard offsets -- no pointer tables, no dependent pointer load per tile: the same
	// loads over xGMI, 0.27 instead of 0.36 ms for 256 x 4 MiB at world 1.  Otherwise (or gec_set_kernel_variant(5), the tests'
	// route to it) the pointer-table kernel below.
	bool strided = gec_get_kernel_variant() != 5;
	std::vector<size_t> s_in(k), s_out(nmiss);
	const uint8_t *s_base = nullptr;
	if (strided) {
		uintptr_t lo = UINTPTR_MAX, hi = 0;
		for (size_t t = 0; t < k; ++t) {
			const size_t v = (size_t)plan->valid[t];
			const uintptr_t p = reinterpret_cast<uintptr_t>(d_peer_slots[v % N]) + (v / N) * S;
			lo = std::min(lo, p);
			hi = std::max(hi, p);
		}
		strided = (hi - lo) / 16 <= 0xffffffffull;
		s_base = reinterpret_cast<const uint8_t *>(lo);
		for (size_t t = 0; t < k && strided; ++t) {
			const size_t v = (size_t)plan->valid[t];
			s_in[t] = reinterpret_cast<uintptr_t>(d_peer_slots[v % N]) + (v / N) * S - lo;
		}
	}
	if (my_cols && strided) {
		for (size_t t = 0; t < k; ++t)
			if ((size_t)plan->valid[t] % N != (size_t)g->rank)
				g->bytes_exchanged += nobjects * my_cols * 16;
		// rows: straight into d_rebuilt (column range [my_lo, my_lo + my_cols) of every missing shard), or packed into d_send at
		// column 0 -- the launch shifts inputs AND outputs by its first column, so the packed base is moved back by as much
		uint8_t *out_base = direct ? static_cast<uint8_t *>(d_rebuilt) : g->d_send - my_lo * 16;
		const size_t out_stride = direct ? S : max_cols * 16;
		for (size_t i = 0; i < nmiss; ++i)
			s_out[i] = i * nobjects * out_stride;
		rc = launch_apply(c, s_base, slots * S, out_base, out_stride, nullptr, my_lo * 16, my_cols * 16, nobjects, s_in.data(), s_out.data(),
				  (int)nmiss, plan->rows.v.data(), gec::MODE_STORE, stream);
		if (rc)
			return rc;
	} else if (my_cols) {
		if (cached) {
			g->bytes_exchanged = g->peer_bytes_cached;
			rc = launch_apply_ptrs_dev(c, g->d_tab, nobjects, nullptr, nullptr, (int)nmiss, (uint32_t)my_cols, plan->rows.v.data(), stream);
		} else {
			g->peer_key_valid = false;
			std::vector<const uint8_t *> in(nobjects * k);
			std::vector<uint8_t *> out(nobjects * nmiss);
			for (size_t o = 0; o < nobjects; ++o) {
				for (size_t t = 0; t < k; ++t) {
					const size_t v = (size_t)plan->valid[t];
					in[o * k + t] = static_cast<const uint8_t *>(d_peer_slots[v % N]) + o * slots * S + (v / N) * S + my_lo * 16;
					if (v % N != (size_t)g->rank)
						g->bytes_exchanged += my_cols * 16;
				}
				for (size_t i = 0; i < nmiss; ++i)
					out[o * nmiss + i] = direct ? static_cast<uint8_t *>(d_rebuilt) + (i * nobjects + o) * S + my_lo * 16
								    : g->d_send + (i * nobjects + o) * max_cols * 16;
			}
			rc = launch_apply_ptrs_dev(c, g->d_tab, nobjects, in.data(), out.data(), (int)nmiss, (uint32_t)my_cols, plan->rows.v.data(), stream);
			if (rc == GEC_OK) {
				g->peer_key = std::move(key);
				g->peer_key_valid = true;
				g->peer_bytes_cached = g->bytes_exchanged;
			}
		}
		if (rc)
			return rc;
	}
	// (2) the rebuilt ranges: mine only (already in place), or everybody's after a (small) all-gather -- which is also the "done
	//     reading" barrier
	if (direct) {
		if (N > 1) {
			rc = g->all_gather(g->ctx, g->d_tok, g->d_tok + 16, 16, hip_stream);  // nobody's slots change while a peer still reads them
			if (rc)
				return rc > 0 ? fail(GEC_E_DEVICE, "all_gather transport failed") : rc;
		}
		return GEC_OK;
	}
	gec::RebuiltArgs ua;
	std::memset(&ua, 0, sizeof(ua));
	ua.rebuilt = static_cast<uint8_t *>(d_rebuilt);
	ua.nobj = (uint32_t)nobjects;
	ua.nmiss = (uint32_t)nmiss;
	ua.cols = (uint32_t)cols;
	ua.max_cols = (uint32_t)max_cols;
	ua.world = (uint32_t)N;
	g->bytes_exchanged += packed_bytes * (N - 1);
	rc = g->all_gather(g->ctx, g->d_send, g->d_recv, packed_bytes, hip_stream);
	if (rc)
		return rc > 0 ? fail(GEC_E_DEVICE, "all_gather transport failed") : rc;
	ua.packed = g->d_recv;
	ua.first_rank = 0;
	ua.nranks_in = (uint32_t)N;
	return launch_rebuilt_unpack(ua, (size_t)ua.nranks_in * nmiss * nobjects * max_cols, stream);
}
GEC_CATCH

}  // extern "C"
