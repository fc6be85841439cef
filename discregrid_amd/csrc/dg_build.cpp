// dg_build.cpp -- see dg_build.h.  Host code; compile with -ffp-contract=off.
#include "dg_build.h"
#include "dg_force.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <future>
#include <thread>
#include <limits>
#include <numeric>

namespace dg
{
namespace
{

#ifndef DG_FLAT_PATCH
#define DG_FLAT_PATCH 0.8
#endif
const double kFlatPatch = DG_FLAT_PATCH; // |mean normal| / area above which a patch gets an oriented box
const int kMinRectPrims = 24;            // patches of at most this many triangles: minimum-area rectangle instead of principal axes

struct D3
{
	double x, y, z;
};
inline D3 sub(D3 a, D3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline D3 add(D3 a, D3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline double dot3(D3 a, D3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline D3 cross3(D3 a, D3 b) { return {a.y * b.z - a.z * b.y, -a.x * b.z + a.z * b.x, a.x * b.y - a.y * b.x}; }
inline D3 over(D3 a, double s) { return {a.x / s, a.y / s, a.z / s}; }
inline D3 times(double s, D3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline D3 unit3(D3 a) { return over(a, std::sqrt(dot3(a, a))); }

// [k0, k1) in up to `max_threads` contiguous pieces, each on its own thread (one piece: on the caller's)
template <class F>
void parallel_ranges(size_t n, size_t min_per_thread, size_t max_threads, F body)
{
	const size_t hw = std::max(1u, std::thread::hardware_concurrency());
	const size_t threads = std::max<size_t>(1, std::min(std::min(max_threads, hw), n / std::max<size_t>(1, min_per_thread)));
	if (threads <= 1)
	{
		body((size_t)0, n);
		return;
	}
	std::vector<std::thread> workers;
	const size_t per = (n + threads - 1) / threads;
	for (size_t k0 = 0; k0 < n; k0 += per)
		workers.emplace_back(body, k0, std::min(n, k0 + per));
	for (auto& w : workers)
		w.join();
}

// Pseudonormals in the caller's triangle order: out[t*8 + slot], slots per dg_geom.h.
// Accumulation order = triangle index order, as in the reference's single loop: the per-triangle terms (face normal, the
// three angles: square roots and arc cosines) are computed on many threads, the SUMS -- per vertex, per edge -- are formed
// by one thread in triangle order, the edges in an open-addressing table (a node-based map was 60 % of this function).
void pseudonormals(const std::vector<D3>& V, const uint32_t* T, size_t nt, std::vector<D3>& out, uint32_t& flags)
{
	const uint64_t nv = V.size();
	std::vector<D3> vsum(V.size(), D3{0, 0, 0});
	auto key = [nv](uint32_t i, uint32_t j) { return (uint64_t)std::min(i, j) * nv + (uint64_t)std::max(i, j); };
	out.assign(nt * kPnSlots, D3{0, 0, 0});
	std::vector<double> angle(3 * nt);
	parallel_ranges(nt, 8192, 64, [&](size_t t0, size_t t1) {
		for (size_t t = t0; t < t1; ++t)
		{
			const D3 a = V[T[3 * t]], b = V[T[3 * t + 1]], c = V[T[3 * t + 2]];
			out[t * kPnSlots + kFace] = unit3(cross3(sub(b, a), sub(c, a)));
			angle[3 * t] = std::acos(std::abs(dot3(unit3(sub(b, a)), unit3(sub(c, a)))));
			angle[3 * t + 1] = std::acos(std::abs(dot3(unit3(sub(a, b)), unit3(sub(c, b)))));
			angle[3 * t + 2] = std::acos(std::abs(dot3(unit3(sub(b, c)), unit3(sub(a, c)))));
		}
	});
	// edge table: open addressing, linear probing, at most 3 nt entries in >= 4 nt slots
	size_t cap = 16;
	while (cap < 4 * nt)
		cap <<= 1;
	const uint64_t kEmpty = ~(uint64_t)0; // (no key: min * nv + max < nv^2 <= 2^64 - 2^33)
	std::vector<uint64_t> ekey(cap, kEmpty);
	std::vector<D3> esum(cap);
	std::vector<uint32_t> ecount(cap, 0u);
	const int shift = 64 - (int)std::lround(std::log2((double)cap));
	auto slot_of = [&](uint64_t k) {
		size_t h = (size_t)((k * 0x9E3779B97F4A7C15ull) >> shift);
		while (ekey[h] != kEmpty && ekey[h] != k)
			h = (h + 1) & (cap - 1);
		return h;
	};
	auto add_edge = [&](uint32_t i, uint32_t j, D3 n) {
		const uint64_t k = key(i, j);
		const size_t h = slot_of(k);
		if (ekey[h] == kEmpty)
		{
			ekey[h] = k;
			esum[h] = n;
			ecount[h] = 1;
		}
		else
		{
			esum[h] = add(esum[h], n);
			ecount[h] += 1;
		}
	};
	for (size_t t = 0; t < nt; ++t)
	{
		const uint32_t i0 = T[3 * t], i1 = T[3 * t + 1], i2 = T[3 * t + 2];
		const D3 n = out[t * kPnSlots + kFace];
		vsum[i0] = add(vsum[i0], times(angle[3 * t], n));
		vsum[i1] = add(vsum[i1], times(angle[3 * t + 1], n));
		vsum[i2] = add(vsum[i2], times(angle[3 * t + 2], n));
		add_edge(i0, i1, n);
		add_edge(i1, i2, n);
		add_edge(i0, i2, n);
	}
	for (auto& n : vsum)
	{
		const double l = std::sqrt(dot3(n, n));
		n = {n.x / l, n.y / l, n.z / l};
	}
	flags = 0;
	for (size_t h = 0; h < cap; ++h)
	{
		if (ecount[h] == 1)
			flags |= 1u;
		else if (ecount[h] > 2)
			flags |= 2u;
	}
	parallel_ranges(nt, 8192, 64, [&](size_t t0, size_t t1) {
		for (size_t t = t0; t < t1; ++t)
		{
			const uint32_t i0 = T[3 * t], i1 = T[3 * t + 1], i2 = T[3 * t + 2];
			out[t * kPnSlots + kV0] = vsum[i0];
			out[t * kPnSlots + kV1] = vsum[i1];
			out[t * kPnSlots + kV2] = vsum[i2];
			out[t * kPnSlots + kE01] = unit3(esum[slot_of(key(i0, i1))]);
			out[t * kPnSlots + kE12] = unit3(esum[slot_of(key(i1, i2))]);
			out[t * kPnSlots + kE02] = unit3(esum[slot_of(key(i0, i2))]);
		}
	});
}

inline float round_down(double v)
{
	float f = (float)v;
	if ((double)f > v)
		f = std::nextafterf(f, -std::numeric_limits<float>::infinity());
	return f;
}
inline float round_up(double v)
{
	float f = (float)v;
	if ((double)f < v)
		f = std::nextafterf(f, std::numeric_limits<float>::infinity());
	return f;
}

struct Prim
{
	double lo[3], hi[3], c[3];
	double v[3][3]; // vertices
	double an[3];   // area-weighted normal (cross product of the edges)
	uint32_t tri;
};

inline float down(double v) { return std::nextafterf(round_down(v), -std::numeric_limits<float>::infinity()); }
inline float up(double v) { return std::nextafterf(round_up(v), std::numeric_limits<float>::infinity()); }

// oriented box of a set of primitives (dg_geom.h: PairRec), float, relative to origin, rounded outward
struct Bounds
{
	float c[3], u[3][3], half[3];
};

Bounds bounds_of(const Prim* p, size_t n, const double origin[3])
{
	// mean normal and how flat the patch is: |sum of area normals| / sum of areas
	double m[3] = {0, 0, 0}, area = 0;
	for (size_t i = 0; i < n; ++i)
	{
		for (int d = 0; d < 3; ++d)
			m[d] += p[i].an[d];
		area += std::sqrt(p[i].an[0] * p[i].an[0] + p[i].an[1] * p[i].an[1] + p[i].an[2] * p[i].an[2]);
	}
	const double len = std::sqrt(m[0] * m[0] + m[1] * m[1] + m[2] * m[2]);
	double ax[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}; // curved or degenerate patch: an ordinary box
	if (std::isfinite(len) && std::isfinite(area) && area > 0 && len > kFlatPatch * area)
	{
		const D3 nrm = {m[0] / len, m[1] / len, m[2] / len};
		// any tangent frame (t1, t2), then rotate it to the principal directions of the vertices
		const double an[3] = {std::fabs(nrm.x), std::fabs(nrm.y), std::fabs(nrm.z)};
		D3 e = {1, 0, 0};
		if (an[1] <= an[0] && an[1] <= an[2]) e = {0, 1, 0};
		else if (an[2] <= an[0] && an[2] <= an[1]) e = {0, 0, 1};
		const D3 t1 = unit3(cross3(nrm, e));
		const D3 t2 = cross3(nrm, t1);
		double c1 = 0, c2 = 0;
		for (size_t i = 0; i < n; ++i)
			for (int k = 0; k < 3; ++k)
			{
				c1 += dot3(t1, D3{p[i].v[k][0], p[i].v[k][1], p[i].v[k][2]});
				c2 += dot3(t2, D3{p[i].v[k][0], p[i].v[k][1], p[i].v[k][2]});
			}
		c1 /= (double)(3 * n);
		c2 /= (double)(3 * n);
		double s11 = 0, s12 = 0, s22 = 0;
		for (size_t i = 0; i < n; ++i)
			for (int k = 0; k < 3; ++k)
			{
				const D3 v = {p[i].v[k][0], p[i].v[k][1], p[i].v[k][2]};
				const double a = dot3(t1, v) - c1, b = dot3(t2, v) - c2;
				s11 += a * a;
				s12 += a * b;
				s22 += b * b;
			}
		double phi = 0.5 * std::atan2(2.0 * s12, s11 - s22);
		// small patches: the in-plane rotation whose bounding rectangle has the smallest area (the principal axes
		// leave empty corners around an irregular patch; the box only prunes, any rotation is valid)
		if (n <= (size_t)kMinRectPrims)
		{
			double best_area = std::numeric_limits<double>::max();
			for (int step = 0; step < 30; ++step)
			{
				const double a = (double)step * (3.14159265358979323846 / 60.0); // 0 .. 87 degrees in steps of 3
				const double ca = std::cos(a), sa = std::sin(a);
				double lo1 = std::numeric_limits<double>::max(), hi1 = -lo1, lo2 = lo1, hi2 = -lo1;
				for (size_t i = 0; i < n; ++i)
					for (int k = 0; k < 3; ++k)
					{
						const D3 v = {p[i].v[k][0], p[i].v[k][1], p[i].v[k][2]};
						const double x1 = dot3(t1, v), x2 = dot3(t2, v);
						const double q1 = ca * x1 + sa * x2, q2 = -sa * x1 + ca * x2;
						lo1 = std::min(lo1, q1);
						hi1 = std::max(hi1, q1);
						lo2 = std::min(lo2, q2);
						hi2 = std::max(hi2, q2);
					}
				const double area = (hi1 - lo1) * (hi2 - lo2);
				if (area < best_area)
				{
					best_area = area;
					phi = a;
				}
			}
		}
		const double cs = std::cos(phi), sn = std::sin(phi);
		const D3 u1 = add(times(cs, t1), times(sn, t2));
		const D3 u2 = cross3(nrm, u1);
		const D3 fr[3] = {nrm, u1, u2};
		bool ok = true;
		for (int a = 0; a < 3; ++a)
			ok = ok && std::isfinite(fr[a].x) && std::isfinite(fr[a].y) && std::isfinite(fr[a].z);
		if (ok)
			for (int a = 0; a < 3; ++a)
			{
				ax[a][0] = fr[a].x;
				ax[a][1] = fr[a].y;
				ax[a][2] = fr[a].z;
			}
	}
	Bounds B;
	// directions shrunk by 1e-6: after rounding to float the Gram matrix of the three still has no
	// eigenvalue above 1 (the sum of squared projections never exceeds the squared length)
	for (int a = 0; a < 3; ++a)
		for (int d = 0; d < 3; ++d)
			B.u[a][d] = (float)(ax[a][d] * (1.0 - 1.0e-6));
	// centre = midpoint of the projection ranges, mapped back (any point would do: the half widths
	// below are measured from the float centre actually stored)
	double mid[3];
	for (int a = 0; a < 3; ++a)
	{
		double plo = std::numeric_limits<double>::max(), phi = std::numeric_limits<double>::lowest();
		for (size_t i = 0; i < n; ++i)
			for (int k = 0; k < 3; ++k)
			{
				const double pr = ax[a][0] * (p[i].v[k][0] - origin[0]) + ax[a][1] * (p[i].v[k][1] - origin[1]) +
								  ax[a][2] * (p[i].v[k][2] - origin[2]);
				plo = std::min(plo, pr);
				phi = std::max(phi, pr);
			}
		mid[a] = 0.5 * (plo + phi);
	}
	for (int d = 0; d < 3; ++d)
		B.c[d] = (float)(mid[0] * ax[0][d] + mid[1] * ax[1][d] + mid[2] * ax[2][d]);
	for (int a = 0; a < 3; ++a)
	{
		double h = 0.0;
		for (size_t i = 0; i < n; ++i)
			for (int k = 0; k < 3; ++k)
				h = std::max(h, std::fabs((double)B.u[a][0] * (p[i].v[k][0] - origin[0] - (double)B.c[0]) +
										  (double)B.u[a][1] * (p[i].v[k][1] - origin[1] - (double)B.c[1]) +
										  (double)B.u[a][2] * (p[i].v[k][2] - origin[2] - (double)B.c[2])));
		B.half[a] = up(h);
	}
	return B;
}

void put_side(PairRec& r, int side, const Bounds& B)
{
	for (int d = 0; d < 3; ++d)
		r.f[d][side] = B.c[d];
	for (int a = 0; a < 3; ++a)
	{
		for (int d = 0; d < 3; ++d)
			r.f[3 + 3 * a + d][side] = B.u[a][d];
		r.f[12 + a][side] = B.half[a];
	}
}
// a side that can never be hit: slabs of half width -FLT_MAX (distance^2 = inf)
void put_empty(PairRec& r, int side)
{
	for (int k = 0; k < 12; ++k)
		r.f[k][side] = 0.0f;
	for (int a = 0; a < 3; ++a)
		r.f[12 + a][side] = -std::numeric_limits<float>::max();
}
void clear_rec(PairRec& r)
{
	std::memset(&r, 0, sizeof(r));
	put_empty(r, 0);
	put_empty(r, 1);
}

// The tree over n primitives has a SHAPE that depends on n alone (object-median splits with the left
// half rounded up to whole leaves), so the record index of every node and the position of every leaf
// are known before the primitives are partitioned: subtrees write into disjoint, precomputed ranges,
// the top levels of the recursion run on separate threads, and the arrays come out exactly as a
// sequential depth-first build would number them.
struct Builder
{
	std::vector<Prim> prims;
	std::vector<PairRec> pairs;
	std::vector<int64_t> order; // position (leaf order, padded to even per leaf) -> prim index or -1
	const double* origin;
	int max_leaf;
	std::atomic<uint32_t> depth{0};
	// levels whose left halves get a thread of their own: 2^levels threads at the bottom of the spawning part
	// (4 on a small host, 6 from 64 hardware threads up: 0.086 -> 0.05 s for 100 820 triangles on 256 cores)
	uint32_t spawn_levels = std::thread::hardware_concurrency() >= 64 ? 6u : 4u;

	size_t left_count(size_t n) const
	{
		size_t half = n / 2;
		if (n > (size_t)(2 * max_leaf))
			half = ((half + max_leaf - 1) / max_leaf) * max_leaf; // keep leaves full
		return half;
	}
	// pair records / leaf positions of the subtree over n primitives
	void shape(size_t n, size_t& n_pairs, size_t& n_positions) const
	{
		if (n <= (size_t)max_leaf)
		{
			n_pairs = 0;
			n_positions = n + (n & 1u);
			return;
		}
		const size_t half = left_count(n);
		size_t pl, ql, pr, qr;
		shape(half, pl, ql);
		shape(n - half, pr, qr);
		n_pairs = 1 + pl + pr;
		n_positions = ql + qr;
	}
	void allocate()
	{
		size_t np, nq;
		shape(prims.size(), np, nq);
		pairs.resize(np);
		for (auto& r : pairs)
			clear_rec(r);
		order.assign(nq, -1);
	}

	// Returns the info word of the subtree over prims [b, e), whose records start at `rec` and whose
	// leaves start at position `pos`.  Split: object median along the largest extent of the centroid
	// bounds (balanced tree, depth = ceil(log2(n / max_leaf))).
	int32_t build(size_t b, size_t e, uint32_t level, size_t rec, size_t pos)
	{
		uint32_t seen = depth.load(std::memory_order_relaxed);
		while (level > seen && !depth.compare_exchange_weak(seen, level, std::memory_order_relaxed))
		{
		}
		if (e - b <= (size_t)max_leaf)
		{
			for (size_t i = b; i < e; ++i)
				order[pos + (i - b)] = (int64_t)i; // an odd leaf keeps its padding slot (-1) at the end
			const uint32_t positions = (uint32_t)((e - b) + ((e - b) & 1u));
			return ~(int32_t)(((uint32_t)pos << kLeafBits) | (positions - 1));
		}
		double clo[3], chi[3];
		for (int d = 0; d < 3; ++d)
		{
			clo[d] = std::numeric_limits<double>::max();
			chi[d] = std::numeric_limits<double>::lowest();
		}
		for (size_t i = b; i < e; ++i)
			for (int d = 0; d < 3; ++d)
			{
				clo[d] = std::min(clo[d], prims[i].c[d]);
				chi[d] = std::max(chi[d], prims[i].c[d]);
			}
		int axis = 0;
		for (int d = 1; d < 3; ++d)
			if (chi[d] - clo[d] > chi[axis] - clo[axis])
				axis = d;
		const size_t half = left_count(e - b);
		const size_t mid = b + half;
		std::nth_element(prims.begin() + b, prims.begin() + mid, prims.begin() + e,
						 [axis](const Prim& p, const Prim& q) {
							 return p.c[axis] < q.c[axis] || (p.c[axis] == q.c[axis] && p.tri < q.tri);
						 });
		size_t pl, ql;
		shape(half, pl, ql);
		int32_t il, ir;
		if (level < spawn_levels && e - b > 8192) // the top of a big tree: left half on its own thread
		{
			auto left = std::async(std::launch::async, [&]() { return build(b, mid, level + 1, rec + 1, pos); });
			ir = build(mid, e, level + 1, rec + 1 + pl, pos + ql);
			il = left.get();
		}
		else
		{
			il = build(b, mid, level + 1, rec + 1, pos);
			ir = build(mid, e, level + 1, rec + 1 + pl, pos + ql);
		}
		put_side(pairs[rec], 0, bounds_of(&prims[b], mid - b, origin));
		put_side(pairs[rec], 1, bounds_of(&prims[mid], e - mid, origin));
		pairs[rec].info[0] = il;
		pairs[rec].info[1] = ir;
		return (int32_t)rec;
	}
};

} // namespace

bool build_mesh(const double* verts, size_t n_vertices, const uint32_t* tris, size_t n_triangles, int max_leaf,
				MeshBuild& out)
{
	if (!verts || !tris || n_triangles == 0 || n_vertices == 0 || n_triangles >= (1u << 27))
		return false;
	max_leaf = std::max(1, std::min(kMaxLeaf, max_leaf));
	for (size_t i = 0; i < 3 * n_triangles; ++i)
		if (tris[i] >= n_vertices)
			return false;
	std::vector<D3> V(n_vertices);
	for (size_t i = 0; i < n_vertices; ++i)
		V[i] = {verts[3 * i], verts[3 * i + 1], verts[3 * i + 2]};

	const bool timing = dg::force_set("build_timing"); // (DG_FORCE=build_timing=1: the phases of the build on stderr)
	auto t_last = std::chrono::steady_clock::now();
	auto tick = [&](const char* what) {
		const auto now = std::chrono::steady_clock::now();
		if (timing)
			std::fprintf(stderr, "build_mesh: %s %.3f s\n", what, std::chrono::duration<double>(now - t_last).count());
		t_last = now;
	};
	std::vector<D3> pn;
	pseudonormals(V, tris, n_triangles, pn, out.not_watertight);
	tick("pseudonormals");

	Builder B;
	B.max_leaf = max_leaf;
	B.prims.resize(n_triangles);
	double lo[3], hi[3];
	for (int d = 0; d < 3; ++d)
	{
		lo[d] = std::numeric_limits<double>::max();
		hi[d] = std::numeric_limits<double>::lowest();
	}
	for (size_t t = 0; t < n_triangles; ++t)
	{
		Prim& p = B.prims[t];
		p.tri = (uint32_t)t;
		for (int d = 0; d < 3; ++d)
		{
			const double a = verts[3 * tris[3 * t] + d], b = verts[3 * tris[3 * t + 1] + d],
						 c = verts[3 * tris[3 * t + 2] + d];
			p.v[0][d] = a;
			p.v[1][d] = b;
			p.v[2][d] = c;
			p.lo[d] = std::min(a, std::min(b, c));
			p.hi[d] = std::max(a, std::max(b, c));
			p.c[d] = 0.5 * (p.lo[d] + p.hi[d]);
			lo[d] = std::min(lo[d], p.lo[d]);
			hi[d] = std::max(hi[d], p.hi[d]);
		}
	}
	for (size_t t = 0; t < n_triangles; ++t)
	{
		Prim& p = B.prims[t];
		const D3 e0 = {p.v[1][0] - p.v[0][0], p.v[1][1] - p.v[0][1], p.v[1][2] - p.v[0][2]};
		const D3 e1 = {p.v[2][0] - p.v[0][0], p.v[2][1] - p.v[0][1], p.v[2][2] - p.v[0][2]};
		const D3 n = cross3(e0, e1);
		p.an[0] = std::isfinite(n.x) ? n.x : 0.0;
		p.an[1] = std::isfinite(n.y) ? n.y : 0.0;
		p.an[2] = std::isfinite(n.z) ? n.z : 0.0;
	}
	for (int d = 0; d < 3; ++d)
		out.origin[d] = 0.5 * (lo[d] + hi[d]);
	{
		double area = 0.0;
		for (size_t t = 0; t < n_triangles; ++t)
		{
			const Prim& p = B.prims[t];
			const double a = 0.5 * std::sqrt(p.an[0] * p.an[0] + p.an[1] * p.an[1] + p.an[2] * p.an[2]);
			area += std::isfinite(a) ? a : 0.0;
		}
		out.mean_edge = std::sqrt(area / (double)n_triangles * (4.0 / 1.7320508075688772));
	}
	B.origin = out.origin;
	tick("primitives");
	B.allocate();
	out.root_info = B.build(0, n_triangles, 0, 0, 0);
	tick("tree");

	out.pairs.swap(B.pairs);
	out.depth = B.depth.load();
	// level-order cut of the tree into <= kSubtrees disjoint subtrees that cover it (heavy bricks are split
	// over them): split the oldest inner node of the queue until kSubtrees pieces exist or only leaves remain
	{
		std::deque<int32_t> open(1, out.root_info);
		out.sub_roots.clear();
		while (!open.empty() && out.sub_roots.size() + open.size() < (size_t)kSubtrees)
		{
			const int32_t info = open.front();
			open.pop_front();
			if (info < 0)
			{
				out.sub_roots.push_back(info);
				continue;
			}
			open.push_back(out.pairs[(size_t)info].info[0]);
			open.push_back(out.pairs[(size_t)info].info[1]);
		}
		out.sub_roots.insert(out.sub_roots.end(), open.begin(), open.end());
	}
	out.n_vertices = n_vertices;
	out.n_triangles = n_triangles;
	const size_t npos = B.order.size(); // even
	out.tris.resize(npos);
	out.pn.assign(npos * kPnSlots * 3, 0.0);
	out.tri_pairs.resize(npos / 2);
	for (auto& r : out.tri_pairs)
		clear_rec(r);
	out.tri_approx.resize(npos / 2);
	std::memset(out.tri_approx.data(), 0, out.tri_approx.size() * sizeof(TriApproxPair)); // padding slots: valid = 0
	double l1 = 0.0;
	for (size_t i = 0; i < n_vertices; ++i)
		l1 = std::max(l1, std::fabs(V[i].x - out.origin[0]) + std::fabs(V[i].y - out.origin[1]) +
							  std::fabs(V[i].z - out.origin[2]));
	out.mesh_l1 = std::nextafterf(round_up(l1), std::numeric_limits<float>::infinity());
	// packets, per-triangle boxes and pseudonormals in leaf order: independent per position pair
	// (a padding slot copies its left neighbour, which is in the same pair)
	auto fill = [&](size_t k0, size_t k1) {
		for (size_t k = k0; k < k1; ++k)
		{
			if (B.order[k] < 0)
			{
				// padding slot: a copy of its left neighbour's packet that no bound test can select
				out.tris[k] = out.tris[k - 1];
				out.tris[k].tri_id = -1;
				continue;
			}
			const Prim& P = B.prims[(size_t)B.order[k]];
			const uint32_t t = P.tri;
			make_packet(verts + 3 * tris[3 * t], verts + 3 * tris[3 * t + 1], verts + 3 * tris[3 * t + 2], (int32_t)t,
						out.tris[k]);
			put_side(out.tri_pairs[k / 2], (int)(k & 1), bounds_of(&P, 1, out.origin));
			make_tri_approx(verts + 3 * tris[3 * t], verts + 3 * tris[3 * t + 1], verts + 3 * tris[3 * t + 2], out.origin,
							out.tri_approx[k / 2], (int)(k & 1));
			for (int s = 0; s < kPnSlots; ++s)
			{
				out.pn[(k * kPnSlots + s) * 3 + 0] = pn[t * kPnSlots + s].x;
				out.pn[(k * kPnSlots + s) * 3 + 1] = pn[t * kPnSlots + s].y;
				out.pn[(k * kPnSlots + s) * 3 + 2] = pn[t * kPnSlots + s].z;
			}
		}
	};
	const size_t n_threads = npos >= (1u << 16) ? std::min<size_t>(64, std::max(1u, std::thread::hardware_concurrency())) : 1;
	if (n_threads <= 1)
		fill(0, npos);
	else
	{
		std::vector<std::thread> workers;
		const size_t per = (((npos + n_threads - 1) / n_threads) + 1) & ~(size_t)1; // even: a pair stays in one chunk
		for (size_t k0 = 0; k0 < npos; k0 += per)
			workers.emplace_back(fill, k0, std::min(npos, k0 + per));
		for (auto& w : workers)
			w.join();
	}
	tick("packets");
	return true;
}

} // namespace dg
