// blake2b_host.hpp -- BLAKE2b-512 on the host cores (RFC 7693, unkeyed), for the two checksums of this project:
//   blake2sum  Garage's content hash = the first 32 bytes of blake2b-512 (src/util/data.rs:130-138), sequential;
//   shardsum   the shard checksum = BLAKE2b in its standard tree mode (include/garage_ec.h has the definition).
// Header-only, no dependencies: libgarage_block (gbm_blake2sum / gbm_shardsum, the few shards the manager hashes
// itself) and libgarage_ec's CPU backend (ec_cpu.cpp) share it.  The device versions are in blake2b.hpp; the oracle
// for all of them is CPython's hashlib.
#pragma once

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

namespace b2host {

constexpr size_t kShardsumLeaf = 4096;  // == GEC_SHARDSUM_LEAF

inline const uint64_t *iv()
{
	static const uint64_t v[8] = {0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL, 0xa54ff53a5f1d36f1ULL,
				      0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
	return v;
}

inline uint64_t rotr64(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }

inline void compress(uint64_t h[8], const uint8_t block[128], uint64_t t, bool last, bool last_node = false)
{
	static const uint8_t SIGMA[12][16] = {
		{0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
		{11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
		{9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
		{12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
		{6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0},
		{0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3}};
	uint64_t m[16], v[16];
	std::memcpy(m, block, 128);  // little-endian host
	const uint64_t *IV = iv();
	for (int i = 0; i < 8; ++i) {
		v[i] = h[i];
		v[i + 8] = IV[i];
	}
	v[12] ^= t;  // t fits 64 bits here
	if (last)
		v[14] = ~v[14];
	if (last && last_node)
		v[15] = ~v[15];  // f1, tree mode
#define B2H_G(a, b, c, d, x, y)                 \
	v[a] = v[a] + v[b] + (x);               \
	v[d] = rotr64(v[d] ^ v[a], 32);         \
	v[c] = v[c] + v[d];                     \
	v[b] = rotr64(v[b] ^ v[c], 24);         \
	v[a] = v[a] + v[b] + (y);               \
	v[d] = rotr64(v[d] ^ v[a], 16);         \
	v[c] = v[c] + v[d];                     \
	v[b] = rotr64(v[b] ^ v[c], 63);
	for (int r = 0; r < 12; ++r) {
		const uint8_t *s = SIGMA[r];
		B2H_G(0, 4, 8, 12, m[s[0]], m[s[1]]);
		B2H_G(1, 5, 9, 13, m[s[2]], m[s[3]]);
		B2H_G(2, 6, 10, 14, m[s[4]], m[s[5]]);
		B2H_G(3, 7, 11, 15, m[s[6]], m[s[7]]);
		B2H_G(0, 5, 10, 15, m[s[8]], m[s[9]]);
		B2H_G(1, 6, 11, 12, m[s[10]], m[s[11]]);
		B2H_G(2, 7, 8, 13, m[s[12]], m[s[13]]);
		B2H_G(3, 4, 9, 14, m[s[14]], m[s[15]]);
	}
#undef B2H_G
	for (int i = 0; i < 8; ++i)
		h[i] ^= v[i] ^ v[i + 8];
}

// Streaming BLAKE2b-512 with an explicit parameter block (words 0..2): a block whose data shards sit in separate
// buffers (some of them rebuilt) is hashed piece by piece.
struct State {
	uint64_t h[8];
	uint64_t t = 0;
	uint8_t buf[128];
	size_t buflen = 0;

	explicit State(uint64_t p0 = 0x01010000ULL ^ 64 /* digest length 64, no key, fanout 1, depth 1 */, uint64_t p1 = 0, uint64_t p2 = 0)
	{
		const uint64_t *IV = iv();
		for (int i = 0; i < 8; ++i)
			h[i] = IV[i];
		h[0] ^= p0;
		h[1] ^= p1;
		h[2] ^= p2;
	}
	void update(const uint8_t *data, size_t len)
	{
		if (len == 0)
			return;
		// the buffered block is only compressed once more data follows it: the LAST block needs the final flag
		if (buflen == 128) {
			t += 128;
			compress(h, buf, t, false);
			buflen = 0;
		}
		if (buflen) {
			const size_t take = std::min(len, 128 - buflen);
			std::memcpy(buf + buflen, data, take);
			buflen += take;
			data += take;
			len -= take;
			if (len == 0)
				return;
			t += 128;
			compress(h, buf, t, false);
			buflen = 0;
		}
		while (len > 128) {
			t += 128;
			compress(h, data, t, false);
			data += 128;
			len -= 128;
		}
		std::memcpy(buf, data, len);
		buflen = len;
	}
	void final(uint8_t out[64], bool last_node = false)
	{
		t += buflen;
		std::memset(buf + buflen, 0, 128 - buflen);
		compress(h, buf, t, true, last_node);
		std::memcpy(out, h, 64);
	}
};

inline void blake2b_params(const uint8_t *data, size_t len, uint64_t p0, uint64_t p1, uint64_t p2, bool last_node, uint8_t out[64])
{
	State s(p0, p1, p2);
	s.update(data, len);  // data may be NULL for the empty message (len == 0)
	s.final(out, last_node);
}

inline void blake2sum(const uint8_t *data, size_t len, uint8_t out[32])
{
	uint8_t full[64];
	State s;
	s.update(data, len);
	s.final(full);
	std::memcpy(out, full, 32);
}

// The shard checksum: BLAKE2b tree mode, kShardsumLeaf-byte leaves, unlimited fanout, depth 2, 64-byte inner
// digests, root truncated to 32 bytes.
inline void shardsum(const uint8_t *data, size_t len, uint8_t out[32])
{
	const uint64_t P0 = 64ull | (2ull << 24) | ((uint64_t)kShardsumLeaf << 32);
	const size_t nleaf = len ? (len + kShardsumLeaf - 1) / kShardsumLeaf : 1;
	State root(P0, 0, 1ull | (64ull << 8));
	uint8_t dig[64];
	for (size_t i = 0; i < nleaf; ++i) {
		const size_t lo = i * kShardsumLeaf, n = len > lo ? std::min<size_t>(kShardsumLeaf, len - lo) : 0;
		blake2b_params(n ? data + lo : nullptr, n, P0, i, 64ull << 8, i + 1 == nleaf, dig);
		root.update(dig, 64);
	}
	uint8_t full[64];
	root.final(full, true);
	std::memcpy(out, full, 32);
}

}  // namespace b2host
