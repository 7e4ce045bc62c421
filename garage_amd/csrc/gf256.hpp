// gf256.hpp -- host-side GF(2^8) field and coding-matrix logic of libgarage_ec.
//
// This is product host logic (tiny k x k matrices), not the data path: shard
// bytes are only ever touched by the HIP kernels in kernels.hip.
//
// Conventions are those of reed-solomon-erasure::galois_8 [EXT, not vendored;
// SURVEY.md Appendix A]: polynomial 0x11D, generator 2, encoding matrix =
// vandermonde(n, k) * inverse(top k rows), reconstruct from the first k present
// shards.  Get these three right and results are bit-exact with the crate.
#pragma once

#include <array>
#include <cstdint>
#include <vector>

namespace gec {

struct Field {
	std::array<uint8_t, 512> exp{};  // doubled so exp[log a + log b] needs no mod
	std::array<uint8_t, 256> log{};

	Field()
	{
		unsigned x = 1;
		for (int i = 0; i < 255; ++i) {
			exp[i] = static_cast<uint8_t>(x);
			log[x] = static_cast<uint8_t>(i);
			x <<= 1;
			if (x & 0x100u)
				x ^= 0x11Du;
		}
		for (int i = 255; i < 512; ++i)
			exp[i] = exp[i - 255];
	}

	uint8_t mul(uint8_t a, uint8_t b) const
	{
		return (a && b) ? exp[log[a] + log[b]] : 0;
	}
	uint8_t inv(uint8_t a) const { return exp[255 - log[a]]; }  // a != 0
	uint8_t pow(uint8_t a, unsigned n) const
	{
		if (n == 0)
			return 1;
		if (a == 0)
			return 0;
		return exp[(static_cast<unsigned>(log[a]) * n) % 255u];
	}
};

inline const Field &field()
{
	static const Field f;
	return f;
}

// Row-major byte matrix.
struct Matrix {
	int rows = 0, cols = 0;
	std::vector<uint8_t> v;

	Matrix() = default;
	Matrix(int r, int c) : rows(r), cols(c), v(static_cast<size_t>(r) * c, 0) {}
	uint8_t &at(int r, int c) { return v[static_cast<size_t>(r) * cols + c]; }
	uint8_t at(int r, int c) const { return v[static_cast<size_t>(r) * cols + c]; }
	const uint8_t *row(int r) const { return v.data() + static_cast<size_t>(r) * cols; }
};

inline Matrix matmul(const Matrix &a, const Matrix &b)
{
	const Field &f = field();
	Matrix o(a.rows, b.cols);
	for (int r = 0; r < a.rows; ++r)
		for (int t = 0; t < a.cols; ++t) {
			uint8_t x = a.at(r, t);
			if (!x)
				continue;
			for (int c = 0; c < b.cols; ++c)
				o.at(r, c) ^= f.mul(x, b.at(t, c));
		}
	return o;
}

// Gauss-Jordan on [A | I]; false if singular.  The inverse is unique, so the
// pivoting order cannot change the bytes.
inline bool invert(const Matrix &a, Matrix &out)
{
	const Field &f = field();
	const int n = a.rows;
	Matrix w(n, 2 * n);
	for (int r = 0; r < n; ++r) {
		for (int c = 0; c < n; ++c)
			w.at(r, c) = a.at(r, c);
		w.at(r, n + r) = 1;
	}
	for (int col = 0; col < n; ++col) {
		int piv = col;
		while (piv < n && w.at(piv, col) == 0)
			++piv;
		if (piv == n)
			return false;
		if (piv != col)
			for (int c = 0; c < 2 * n; ++c)
				std::swap(w.at(piv, c), w.at(col, c));
		uint8_t s = f.inv(w.at(col, col));
		if (s != 1)
			for (int c = 0; c < 2 * n; ++c)
				w.at(col, c) = f.mul(w.at(col, c), s);
		for (int r = 0; r < n; ++r) {
			if (r == col)
				continue;
			uint8_t q = w.at(r, col);
			if (!q)
				continue;
			for (int c = 0; c < 2 * n; ++c)
				w.at(r, c) ^= f.mul(q, w.at(col, c));
		}
	}
	out = Matrix(n, n);
	for (int r = 0; r < n; ++r)
		for (int c = 0; c < n; ++c)
			out.at(r, c) = w.at(r, n + c);
	return true;
}

// (k+m) x k systematic encoding matrix.
inline bool build_encoding_matrix(int k, int m, Matrix &out)
{
	const Field &f = field();
	const int n = k + m;
	Matrix vm(n, k);
	for (int r = 0; r < n; ++r)
		for (int c = 0; c < k; ++c)
			vm.at(r, c) = f.pow(static_cast<uint8_t>(r), static_cast<unsigned>(c));
	Matrix top(k, k), topinv;
	for (int r = 0; r < k; ++r)
		for (int c = 0; c < k; ++c)
			top.at(r, c) = vm.at(r, c);
	if (!invert(top, topinv))
		return false;
	out = matmul(vm, topinv);
	return true;
}

// Systematic Cauchy matrix: identity on top, parity row r / column c = 1 / ((k+r) ^ c).
// x_r = k + r and y_c = c are disjoint sets of field elements, so every square
// sub-matrix of the parity block is invertible and the code is MDS.
inline bool build_cauchy_matrix(int k, int m, Matrix &out)
{
	const Field &f = field();
	out = Matrix(k + m, k);
	for (int r = 0; r < k; ++r)
		out.at(r, r) = 1;
	for (int r = 0; r < m; ++r)
		for (int c = 0; c < k; ++c)
			out.at(k + r, c) = f.inv(static_cast<uint8_t>((k + r) ^ c));  // (k+r)^c != 0 since c < k <= k+r
	return true;
}

}  // namespace gec
