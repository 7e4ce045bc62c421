// TEST-ONLY stand-in for the subset of libgarage_ec's C ABI that
// libgarage_block uses, with the arithmetic done by the CPU oracle
// (oracle/rs_oracle.c).  It exists so that the C++ BlockManager host logic can
// run on a box without a GPU and under ASan/UBSan.  Never linked into a product
// library: the real libgarage_ec has no CPU path.
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/garage_block.h"
#include "../../oracle/rs_oracle.h"

struct gec_codec {
	int k, m;
};

extern "C" {

gec_codec *stub_codec_create(int k, int m) { return new gec_codec{k, m}; }
void stub_codec_destroy(gec_codec *c) { delete c; }

int gec_codec_k(const gec_codec *c) { return c->k; }
int gec_codec_m(const gec_codec *c) { return c->m; }
const char *gec_strerror(int) { return "stub error"; }
const char *gec_last_error(void) { return ""; }

size_t gec_shard_len(int k, size_t block_len)
{
	if (k <= 0)
		return 0;
	size_t per = ((block_len ? block_len : 1) + k - 1) / k;
	return (per + 63) / 64 * 64;
}

int gec_encode_hash_batch(const gec_codec *c, size_t nb, const uint8_t *const *blocks, const size_t *len, size_t S,
			  uint8_t *const *parity, uint8_t *sums)
{
	const int k = c->k, m = c->m;
	for (size_t b = 0; b < nb; ++b) {
		std::vector<uint8_t> pad((size_t)k * S, 0);
		std::memcpy(pad.data(), blocks[b], len[b]);
		std::vector<const uint8_t *> d(k);
		std::vector<uint8_t *> p(m);
		for (int i = 0; i < k; ++i)
			d[i] = pad.data() + (size_t)i * S;
		for (int r = 0; r < m; ++r)
			p[r] = parity[b] + (size_t)r * S;
		if (rso_encode(k, m, S, d.data(), p.data(), RSO_SCALAR) != RSO_OK)
			return GEC_E_INVALID_ARG;
		for (int j = 0; j < k + m; ++j)
			gbm_shardsum(j < k ? d[j] : p[j - k], S, sums + (b * (k + m) + j) * 32);
	}
	return GEC_OK;
}

// pinned host memory does not exist without a device: plain heap
void *gec_host_alloc(size_t bytes) { return std::malloc(bytes ? bytes : 1); }
void gec_host_free(void *p) { std::free(p); }

static std::atomic<unsigned long long> g_reconstruct_calls{0};
unsigned long long stub_reconstruct_calls(void) { return g_reconstruct_calls.load(); }

int gec_reconstruct_batch(const gec_codec *c, size_t nb, const uint8_t *const *shards, uint8_t *const *out, size_t S,
			  int data_only)
{
	++g_reconstruct_calls;
	const int k = c->k, n = c->k + c->m;
	for (size_t b = 0; b < nb; ++b) {
		std::vector<std::vector<uint8_t>> buf(n, std::vector<uint8_t>(S));
		std::vector<uint8_t *> ptr(n);
		std::vector<uint8_t> present(n);
		for (int j = 0; j < n; ++j) {
			present[j] = shards[b * n + j] != nullptr;
			if (present[j])
				std::memcpy(buf[j].data(), shards[b * n + j], S);
			ptr[j] = buf[j].data();
		}
		int rc = rso_reconstruct(k, c->m, S, ptr.data(), present.data(), data_only);
		if (rc == RSO_TOO_FEW_PRESENT)
			return GEC_E_TOO_FEW_PRESENT;
		if (rc != RSO_OK)
			return GEC_E_INVALID_ARG;
		for (int j = 0; j < n; ++j)
			if (!present[j] && out[b * n + j])
				std::memcpy(out[b * n + j], buf[j].data(), S);
	}
	return GEC_OK;
}

int gec_reconstruct_hash_batch(const gec_codec *c, size_t nb, const uint8_t *const *shards, uint8_t *const *out, size_t S,
			       int data_only, uint8_t *in_sums, uint8_t *out_sums)
{
	int rc = gec_reconstruct_batch(c, nb, shards, out, S, data_only);
	if (rc)
		return rc;
	const int k = c->k, n = c->k + c->m;
	for (size_t b = 0; b < nb; ++b) {
		int seen = 0;
		for (int j = 0; j < n; ++j) {
			if (shards[b * n + j]) {
				if (seen++ < k)
					gbm_shardsum(shards[b * n + j], S, in_sums + 32 * (b * n + j));
			} else if (out[b * n + j] && !(data_only && j >= k)) {
				gbm_shardsum(out[b * n + j], S, out_sums + 32 * (b * n + j));
			}
		}
	}
	return GEC_OK;
}

// read path in one trip: checksums of the first k present shards, rebuild of missing data shards, block checksum
int gec_decode_verify_batch(const gec_codec *c, size_t nb, const uint8_t *const *shards, size_t S, const size_t *block_len,
			    uint8_t *const *rebuilt, uint8_t *shard_sums, uint8_t *block_sums)
{
	const int k = c->k, n = c->k + c->m;
	for (size_t b = 0; b < nb; ++b) {
		std::vector<std::vector<uint8_t>> buf(n, std::vector<uint8_t>(S));
		std::vector<uint8_t *> ptr(n);
		std::vector<uint8_t> present(n, 0);
		int seen = 0;
		for (int j = 0; j < n; ++j) {
			ptr[j] = buf[j].data();
			if (shards[b * n + j] && seen < k) {  // the crate's rule: only the first k present shards are read
				present[j] = 1;
				++seen;
				std::memcpy(buf[j].data(), shards[b * n + j], S);
				gbm_shardsum(shards[b * n + j], S, shard_sums + (b * n + j) * 32);
			}
		}
		if (seen < k)
			return GEC_E_TOO_FEW_PRESENT;
		if (rso_reconstruct(k, c->m, S, ptr.data(), present.data(), 1) != RSO_OK)
			return GEC_E_INVALID_ARG;
		for (int j = 0; j < k; ++j)
			if (!shards[b * n + j]) {
				if (!rebuilt || !rebuilt[b * n + j])
					return GEC_E_INVALID_ARG;
				std::memcpy(rebuilt[b * n + j], buf[j].data(), S);
			}
		if (block_sums) {
			std::vector<uint8_t> blk;
			for (int j = 0; j < k; ++j)
				blk.insert(blk.end(), buf[j].begin(), buf[j].end());
			gbm_blake2sum(blk.data(), block_len[b], block_sums + 32 * b);
		}
	}
	return GEC_OK;
}

int gec_verify_batch(const gec_codec *c, size_t nb, const uint8_t *const *shards, size_t S, uint8_t *ok)
{
	const int n = c->k + c->m;
	for (size_t b = 0; b < nb; ++b) {
		int good = 0;
		if (rso_verify(c->k, c->m, S, shards + b * n, &good) != RSO_OK)
			return GEC_E_INVALID_ARG;
		ok[b] = (uint8_t)good;
	}
	return GEC_OK;
}

int gec_verify_hash_batch(const gec_codec *c, size_t nb, const uint8_t *const *shards, size_t S, uint8_t *ok, uint8_t *sums)
{
	const int n = c->k + c->m;
	int rc = gec_verify_batch(c, nb, shards, S, ok);
	for (size_t i = 0; !rc && i < nb * (size_t)n; ++i)
		gbm_shardsum(shards[i], S, sums + 32 * i);
	return rc;
}

int gec_shardsum_batch(const gec_codec *, size_t n, const uint8_t *const *msgs, const size_t *lens, uint8_t *out)
{
	for (size_t i = 0; i < n; ++i)
		gbm_shardsum(msgs[i], lens[i], out + 32 * i);
	return GEC_OK;
}

int gec_blake2sum_batch(const gec_codec *, size_t n, const uint8_t *const *msgs, const size_t *lens, uint8_t *out)
{
	for (size_t i = 0; i < n; ++i)
		gbm_blake2sum(msgs[i], lens[i], out + 32 * i);
	return GEC_OK;
}

}  // extern "C"
