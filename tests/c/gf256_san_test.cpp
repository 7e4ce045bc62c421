// Product host logic (garage_amd/csrc/gf256.hpp) under ASan + UBSan: encoding
// matrices for a sweep of (k, m), systematic top, M[valid]^-1 * M[valid] == I for
// random erasure patterns, and the Appendix-A known answers.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <random>

#include "../../garage_amd/csrc/gf256.hpp"

#define CHECK(c)                                                            \
	do {                                                                \
		if (!(c)) {                                                 \
			fprintf(stderr, "FAIL %s:%d: %s\n", __FILE__, __LINE__, #c); \
			exit(1);                                            \
		}                                                           \
	} while (0)

int main()
{
	const gec::Field &f = gec::field();
	CHECK(f.mul(3, 4) == 12 && f.mul(7, 7) == 21 && f.mul(23, 45) == 41);
	CHECK(f.pow(2, 2) == 4 && f.pow(5, 20) == 235 && f.pow(13, 7) == 43);
	gec::Matrix a(3, 3), inv;
	const uint8_t av[9] = {56, 23, 98, 3, 100, 200, 45, 201, 123}, want[9] = {175, 133, 33, 130, 13, 245, 112, 35, 126};
	std::memcpy(a.v.data(), av, 9);
	CHECK(gec::invert(a, inv) && std::memcmp(inv.v.data(), want, 9) == 0);
	{  // upstream constant-times-slice vectors (klauspost/reedsolomon TestGalois; recalled, matched first time)
		const uint8_t in[18] = {0, 1, 2, 3, 4, 5, 6, 10, 50, 100, 150, 174, 201, 255, 99, 32, 67, 85};
		const uint8_t w25[18] = {0x0, 0x19, 0x32, 0x2b, 0x64, 0x7d, 0x56, 0xfa, 0xb8, 0x6d, 0xc7, 0x85, 0xc3, 0x1f, 0x22, 0x7, 0x25, 0xfe};
		const uint8_t w177[18] = {0x0, 0xb1, 0x7f, 0xce, 0xfe, 0x4f, 0x81, 0x9e, 0x3, 0x6, 0xe8, 0x75, 0xbd, 0x40, 0x36, 0xa3, 0x95, 0xcb};
		for (int i = 0; i < 18; ++i)
			CHECK(f.mul(25, in[i]) == w25[i] && f.mul(177, in[i]) == w177[i]);
	}
	{  // upstream MatrixTest vectors (recalled; matched first time): 2x2 product, 5x5 inverse with row swaps
		gec::Matrix p(2, 2), q(2, 2);
		p.v = {1, 2, 3, 4};
		q.v = {5, 6, 7, 8};
		const uint8_t pq[4] = {11, 22, 19, 42};
		CHECK(std::memcmp(gec::matmul(p, q).v.data(), pq, 4) == 0);
		gec::Matrix m5(5, 5), i5;
		m5.v = {1, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 1, 7, 7, 6, 6, 1};
		const uint8_t w5[25] = {1, 0, 0, 0, 0, 0, 1, 0, 0, 0, 123, 123, 1, 122, 122, 0, 0, 1, 0, 0, 0, 0, 0, 1, 0};
		CHECK(gec::invert(m5, i5) && std::memcmp(i5.v.data(), w5, 25) == 0);
	}
	gec::Matrix sing(2, 2);
	sing.v = {1, 1, 1, 1};
	CHECK(!gec::invert(sing, inv));
	std::mt19937 rng(7);
	const int ks[] = {1, 2, 3, 5, 10, 17, 20, 40, 100, 200, 255};
	for (int k : ks)
		for (int m : {1, 2, 4, 8, 56}) {
			if (k + m > 256)
				continue;
			gec::Matrix enc;
			CHECK(gec::build_encoding_matrix(k, m, enc));
			for (int r = 0; r < k; ++r)
				for (int c = 0; c < k; ++c)
					CHECK(enc.at(r, c) == (r == c ? 1 : 0));
			for (int trial = 0; trial < 3; ++trial) {
				std::vector<int> idx(k + m);
				for (int i = 0; i < k + m; ++i)
					idx[i] = i;
				std::shuffle(idx.begin(), idx.end(), rng);
				std::sort(idx.begin(), idx.begin() + k);
				gec::Matrix sub(k, k), dec;
				for (int t = 0; t < k; ++t)
					std::memcpy(&sub.at(t, 0), enc.row(idx[t]), k);
				CHECK(gec::invert(sub, dec));  // MDS: every k rows are independent
				gec::Matrix id = gec::matmul(dec, sub);
				for (int r = 0; r < k; ++r)
					for (int c = 0; c < k; ++c)
						CHECK(id.at(r, c) == (r == c ? 1 : 0));
			}
		}
	gec::Matrix enc;
	CHECK(gec::build_encoding_matrix(10, 4, enc));
	const uint8_t row0[10] = {129, 150, 175, 184, 210, 196, 254, 232, 3, 2};
	CHECK(std::memcmp(enc.row(10), row0, 10) == 0);
	printf("gf256_san_test: OK\n");
	return 0;
}
