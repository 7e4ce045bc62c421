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
