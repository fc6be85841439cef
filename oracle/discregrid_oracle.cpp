// oracle/discregrid_oracle.cpp -- CPU RESTATEMENT OF THE REFERENCE HOT PATH.
//
// *** TEST INFRASTRUCTURE, NOT THE PRODUCT. ***  Only tests/, __graft_entry__.smoke()
// and bench.py's `cpu_baseline` leg may load this library.  The product
// (include/discregrid_hip.h, discregrid_amd/csrc) never links, loads or calls it and
// has no CPU fallback.
//
// Parity status: PINNED.  tests/test_oracle.py checks this restatement
//   (1) byte-for-byte against the reference's only golden file
//       cmd/generate_sdf/resources/box.cdf (committed copy: tests/golden/box.cdf),
//   (2) bit-for-bit against the UNMODIFIED reference compiled by `make -C oracle ref`
//       (oracle/_ref/libdiscregrid_ref.so) on box / bunny / icosphere inputs whenever
//       that library is present, and
//   (3) against golden vectors under tests/golden/ generated from that reference build
//       by tests/golden/make_golden.py.
//
// Every function cites the reference lines it restates (paths relative to
// /root/reference).  The arithmetic order of every floating-point expression is the
// reference's (compile with -ffp-contract=off: the result is FMA-sensitive, SURVEY.md
// fact 4); the code structure is this repository's own.
#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <numeric>
#include <random>
#include <unordered_map>
#include <vector>

namespace
{

// ----------------------------------------------------------------------------------
// 3-vector with the reference's association order.
// discregrid/include/Discregrid/geometry/TriangleMeshDistance.h:40-70
// ----------------------------------------------------------------------------------
struct V3
{
	double x, y, z;
	double operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 scale(double s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }  // :59, :70
inline V3 divide(V3 a, double s) { return {a.x / s, a.y / s, a.z / s}; } // :61
inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; } // :53 (left-assoc)
inline V3 cross(V3 a, V3 b)                                                // :54
{
	return {a.y * b.z - a.z * b.y, -a.x * b.z + a.z * b.x, a.x * b.y - a.y * b.x};
}
inline double norm(V3 a) { return std::sqrt(dot(a, a)); }  // :64-65
inline V3 unit(V3 a) { return divide(a, norm(a)); }        // :66

enum Entity : int { V0 = 0, V1 = 1, V2 = 2, E01 = 3, E12 = 4, E02 = 5, FACE = 6 }; // :75

struct Closest
{
	double d2;
	double s, t;
	int entity;
};

// ----------------------------------------------------------------------------------
// Squared distance point <-> triangle, 7-region test.
// TriangleMeshDistance.h:564-820
// ----------------------------------------------------------------------------------
struct TriTerms
{
	double a00, a01, a11, b0, b1, c;
};
// closest point restricted to the edge v0-v1 (t = 0): :587-601, :651-669, :759-777
inline Closest on_edge01(const TriTerms& q, bool test_v0)
{
	if (test_v0 && q.b0 >= 0)
		return {q.c, 0.0, 0.0, V0};
	if (-q.b0 >= q.a00)
		return {q.a00 + 2 * q.b0 + q.c, 1.0, 0.0, V1};
	double s = -q.b0 / q.a00;
	return {q.b0 * s + q.c, s, 0.0, E01};
}
// closest point restricted to the edge v0-v2 (s = 0): :605-623, :628-646, :712-730
inline Closest on_edge02(const TriTerms& q)
{
	if (q.b1 >= 0)
		return {q.c, 0.0, 0.0, V0};
	if (-q.b1 >= q.a11)
		return {q.a11 + 2 * q.b1 + q.c, 0.0, 1.0, V2};
	double t = -q.b1 / q.a11;
	return {q.b1 * t + q.c, 0.0, t, E02};
}
// the full quadratic Q(s,t): :678-679, :706-707, :753-754, :805-806
inline double quadratic(const TriTerms& q, double s, double t)
{
	return s * (q.a00 * s + q.a01 * t + 2 * q.b0) + t * (q.a01 * s + q.a11 * t + 2 * q.b1) + q.c;
}

inline Closest closest_on_triangle(V3 p, V3 v0, V3 v1, V3 v2, V3* nearest_point)
{
	const V3 diff = v0 - p;
	const V3 e0 = v1 - v0;
	const V3 e1 = v2 - v0;
	TriTerms q;
	q.a00 = dot(e0, e0);
	q.a01 = dot(e0, e1);
	q.a11 = dot(e1, e1);
	q.b0 = dot(diff, e0);
	q.b1 = dot(diff, e1);
	q.c = dot(diff, diff);
	const double det = std::abs(q.a00 * q.a11 - q.a01 * q.a01);
	double s = q.a01 * q.b1 - q.a11 * q.b0;
	double t = q.a01 * q.b0 - q.a00 * q.b1;

	Closest r;
	if (s + t <= det)
	{
		if (s < 0)
		{
			if (t < 0) // region 4 :585-625
				r = (q.b0 < 0) ? on_edge01(q, false) : on_edge02(q);
			else // region 3 :626-647
				r = on_edge02(q);
		}
		else if (t < 0) // region 5 :649-670
			r = on_edge01(q, true);
		else // region 0 :671-680
		{
			const double inv_det = 1 / det;
			s *= inv_det;
			t *= inv_det;
			r = {quadratic(q, s, t), s, t, FACE};
		}
	}
	else if (s < 0) // region 2 :686-732
	{
		const double tmp0 = q.a01 + q.b0;
		const double tmp1 = q.a11 + q.b1;
		if (tmp1 > tmp0)
		{
			const double numer = tmp1 - tmp0;
			const double denom = q.a00 - 2 * q.a01 + q.a11;
			if (numer >= denom)
				r = {q.a00 + 2 * q.b0 + q.c, 1.0, 0.0, V1};
			else
			{
				s = numer / denom;
				t = 1 - s;
				r = {quadratic(q, s, t), s, t, E12};
			}
		}
		else if (tmp1 <= 0)
			r = {q.a11 + 2 * q.b1 + q.c, 0.0, 1.0, V2};
		else if (q.b1 >= 0)
			r = {q.c, 0.0, 0.0, V0};
		else
		{
			t = -q.b1 / q.a11;
			r = {q.b1 * t + q.c, 0.0, t, E02};
		}
	}
	else if (t < 0) // region 6 :733-779
	{
		const double tmp0 = q.a01 + q.b1;
		const double tmp1 = q.a00 + q.b0;
		if (tmp1 > tmp0)
		{
			const double numer = tmp1 - tmp0;
			const double denom = q.a00 - 2 * q.a01 + q.a11;
			if (numer >= denom)
				r = {q.a11 + 2 * q.b1 + q.c, 0.0, 1.0, V2};
			else
			{
				t = numer / denom;
				s = 1 - t;
				r = {quadratic(q, s, t), s, t, E12};
			}
		}
		else if (tmp1 <= 0)
			r = {q.a00 + 2 * q.b0 + q.c, 1.0, 0.0, V1};
		else if (q.b0 >= 0)
			r = {q.c, 0.0, 0.0, V0};
		else
		{
			s = -q.b0 / q.a00;
			r = {q.b0 * s + q.c, s, 0.0, E01};
		}
	}
	else // region 1 :780-809
	{
		const double numer = q.a11 + q.b1 - q.a01 - q.b0;
		if (numer <= 0)
			r = {q.a11 + 2 * q.b1 + q.c, 0.0, 1.0, V2};
		else
		{
			const double denom = q.a00 - 2 * q.a01 + q.a11;
			if (numer >= denom)
				r = {q.a00 + 2 * q.b0 + q.c, 1.0, 0.0, V1};
			else
			{
				s = numer / denom;
				t = 1 - s;
				r = {quadratic(q, s, t), s, t, E12};
			}
		}
	}
	if (r.d2 < 0) // :813-816
		r.d2 = 0;
	*nearest_point = (v0 + scale(r.s, e0)) + scale(r.t, e1); // :818
	return r;
}

// ----------------------------------------------------------------------------------
// The mesh-distance object: sphere BVH + pseudonormals.
// TriangleMeshDistance.h:93-133 (fields), :336-441 (_construct), :443-512 (_build_tree)
// ----------------------------------------------------------------------------------
struct Sphere
{
	V3 c;
	double r;
};
struct BvhNode
{
	Sphere left_bv, right_bv;
	int left = -1; // -1 => leaf, `right` is the triangle id (:107)
	int right = -1;
};
struct BuildTri
{
	V3 v[3];
	int id;
};

struct MeshDistance
{
	std::vector<V3> verts;
	std::vector<std::array<int, 3>> tris;
	std::vector<BvhNode> nodes;
	std::vector<V3> pn_tri;
	std::vector<std::array<V3, 3>> pn_edge;
	std::vector<V3> pn_vert;
	Sphere root_bv;
	bool single_edge = false, triple_edge = false;

	// :443-512.  Recursion writes the child's sphere into a caller-provided slot; slots
	// are addressed by (node, side) instead of by reference so that growing `nodes`
	// cannot invalidate them (the reference's by-reference variant is only safe by
	// accident).
	void build(int node_id, int slot_node, int slot_side, std::vector<BuildTri>& t, int begin, int end)
	{
		auto put_sphere = [&](Sphere s) {
			if (slot_node < 0)
				root_bv = s;
			else if (slot_side == 0)
				nodes[slot_node].left_bv = s;
			else
				nodes[slot_node].right_bv = s;
		};
		const int n = end - begin;
		if (n == 1)
		{
			nodes[node_id].left = -1;
			nodes[node_id].right = t[begin].id;
			const BuildTri& tri = t[begin];
			const V3 c = divide((tri.v[0] + tri.v[1]) + tri.v[2], 3.0);
			const double r = std::max(std::max(norm(tri.v[0] - c), norm(tri.v[1] - c)), norm(tri.v[2] - c));
			put_sphere({c, r});
			return;
		}
		double top[3], bot[3];
		for (int d = 0; d < 3; ++d)
		{
			top[d] = std::numeric_limits<double>::lowest();
			bot[d] = std::numeric_limits<double>::max();
		}
		V3 c = {0, 0, 0};
		for (int i = begin; i < end; ++i)
			for (int k = 0; k < 3; ++k)
			{
				const V3& p = t[i].v[k];
				c = c + p;
				for (int d = 0; d < 3; ++d)
				{
					top[d] = std::max(top[d], p[d]);
					bot[d] = std::min(bot[d], p[d]);
				}
			}
		c = divide(c, (double)(3 * n)); // `center /= 3*n_triangles` (int -> double) :479
		const double diag[3] = {top[0] - bot[0], top[1] - bot[1], top[2] - bot[2]};
		const int split = (int)(std::max_element(diag, diag + 3) - diag); // first maximum :481
		double r2 = 0.0;
		for (int i = begin; i < end; ++i)
			for (int k = 0; k < 3; ++k)
			{
				const V3 d = c - t[i].v[k];
				r2 = std::max(r2, dot(d, d));
			}
		put_sphere({c, std::sqrt(r2)});
		// same comparator, same algorithm (libstdc++ std::sort) => same permutation :494-499
		std::sort(t.begin() + begin, t.begin() + end,
				  [split](const BuildTri& a, const BuildTri& b) { return a.v[0][split] < b.v[0][split]; });
		const int mid = (int)(0.5 * (begin + end)); // :502
		const int l = (int)nodes.size();
		nodes[node_id].left = l;
		nodes.push_back(BvhNode());
		build(l, node_id, 0, t, begin, mid);
		const int r = (int)nodes.size();
		nodes[node_id].right = r;
		nodes.push_back(BvhNode());
		build(r, node_id, 1, t, mid, end);
	}

	void construct() // :336-441
	{
		const int nt = (int)tris.size();
		std::vector<BuildTri> t(nt);
		for (int i = 0; i < nt; ++i)
		{
			t[i].id = i;
			for (int k = 0; k < 3; ++k)
				t[i].v[k] = verts[tris[i][k]];
		}
		nodes.clear();
		nodes.reserve(2 * (size_t)nt);
		nodes.push_back(BvhNode());
		build(0, -1, 0, t, 0, nt);

		// pseudonormals :359-420
		std::unordered_map<uint64_t, V3> edge_sum;
		std::unordered_map<uint64_t, int> edge_cnt;
		const uint64_t nv = (uint64_t)verts.size();
		auto key = [nv](int i, int j) { return (uint64_t)std::min(i, j) * nv + (uint64_t)std::max(i, j); };
		auto add_edge = [&](int i, int j, V3 n) {
			auto k = key(i, j);
			auto it = edge_sum.find(k);
			if (it == edge_sum.end())
			{
				edge_sum[k] = n;
				edge_cnt[k] = 1;
			}
			else
			{
				it->second = it->second + n;
				edge_cnt[k] += 1;
			}
		};
		pn_tri.assign(nt, V3{0, 0, 0});
		pn_edge.resize(nt);
		pn_vert.assign(verts.size(), V3{0, 0, 0});
		for (int i = 0; i < nt; ++i)
		{
			const V3 a = verts[tris[i][0]], b = verts[tris[i][1]], c = verts[tris[i][2]];
			const V3 n = unit(cross(b - a, c - a)); // :394
			pn_tri[i] = n;
			const double al0 = std::acos(std::abs(dot(unit(b - a), unit(c - a)))); // :398
			const double al1 = std::acos(std::abs(dot(unit(a - b), unit(c - b)))); // :399
			const double al2 = std::acos(std::abs(dot(unit(b - c), unit(a - c)))); // :400
			pn_vert[tris[i][0]] = pn_vert[tris[i][0]] + scale(al0, n);
			pn_vert[tris[i][1]] = pn_vert[tris[i][1]] + scale(al1, n);
			pn_vert[tris[i][2]] = pn_vert[tris[i][2]] + scale(al2, n);
			add_edge(tris[i][0], tris[i][1], n);
			add_edge(tris[i][1], tris[i][2], n);
			add_edge(tris[i][0], tris[i][2], n);
		}
		for (auto& n : pn_vert) // normalize(): component / norm :67, :411-413
		{
			const double l = norm(n);
			n = {n.x / l, n.y / l, n.z / l};
		}
		for (int i = 0; i < nt; ++i)
		{
			pn_edge[i][0] = unit(edge_sum[key(tris[i][0], tris[i][1])]);
			pn_edge[i][1] = unit(edge_sum[key(tris[i][1], tris[i][2])]);
			pn_edge[i][2] = unit(edge_sum[key(tris[i][0], tris[i][2])]);
		}
		for (auto const& kv : edge_cnt) // watertightness diagnostics :422-438
		{
			if (kv.second == 1)
				single_edge = true;
			else if (kv.second > 2)
				triple_edge = true;
		}
	}
};

struct QueryResult // TriangleMeshDistance.h:80-86
{
	double distance = std::numeric_limits<double>::max();
	V3 nearest{0, 0, 0};
	int entity = 0;
	int tri = -1;
};
struct Counters
{
	uint64_t inner = 0, leaf = 0;
};

// Recursive nearest-first walk, TriangleMeshDistance.h:514-562.
template <bool COUNT>
void query(const MeshDistance& m, QueryResult& res, const BvhNode& node, V3 p, Counters* cnt)
{
	if (node.left == -1)
	{
		if (COUNT)
			cnt->leaf++;
		const auto& tri = m.tris[node.right];
		V3 np;
		const Closest c = closest_on_triangle(p, m.verts[tri[0]], m.verts[tri[1]], m.verts[tri[2]], &np);
		if (c.d2 < res.distance * res.distance) // compares against the re-squared sqrt :528
		{
			res.nearest = np;
			res.entity = c.entity;
			res.distance = std::sqrt(c.d2);
			res.tri = node.right;
		}
		return;
	}
	if (COUNT)
		cnt->inner++;
	const double dl = norm(p - node.left_bv.c) - node.left_bv.r;
	const double dr = norm(p - node.right_bv.c) - node.right_bv.r;
	if (dl < dr)
	{
		if (dl < res.distance)
			query<COUNT>(m, res, m.nodes[node.left], p, cnt);
		if (dr < res.distance)
			query<COUNT>(m, res, m.nodes[node.right], p, cnt);
	}
	else
	{
		if (dr < res.distance)
			query<COUNT>(m, res, m.nodes[node.right], p, cnt);
		if (dl < res.distance)
			query<COUNT>(m, res, m.nodes[node.left], p, cnt);
	}
}

// signed_distance, TriangleMeshDistance.h:269-308 (+ unsigned_distance :316-328)
template <bool COUNT>
QueryResult signed_distance(const MeshDistance& m, V3 p, Counters* cnt)
{
	QueryResult r;
	query<COUNT>(m, r, m.nodes[0], p, cnt);
	const auto& tri = m.tris[r.tri];
	V3 n;
	switch (r.entity)
	{
	case V0: n = m.pn_vert[tri[0]]; break;
	case V1: n = m.pn_vert[tri[1]]; break;
	case V2: n = m.pn_vert[tri[2]]; break;
	case E01: n = m.pn_edge[r.tri][0]; break;
	case E12: n = m.pn_edge[r.tri][1]; break;
	case E02: n = m.pn_edge[r.tri][2]; break;
	default: n = m.pn_tri[r.tri]; break;
	}
	const V3 u = p - r.nearest;
	r.distance *= (dot(u, n) >= 0.0) ? 1.0 : -1.0;
	return r;
}

// ----------------------------------------------------------------------------------
// Grid bookkeeping.  discregrid/include/Discregrid/discrete_grid.hpp:22-29,
// discregrid/src/discrete_grid.cpp:9-38
// ----------------------------------------------------------------------------------
struct Grid
{
	double dmin[3], dmax[3];
	unsigned res[3];
	double cell[3], inv_cell[3];
	uint64_t n_cells;

	void init(const double domain[6], const unsigned r[3])
	{
		for (int d = 0; d < 3; ++d)
		{
			dmin[d] = domain[d];
			dmax[d] = domain[3 + d];
			res[d] = r[d];
			cell[d] = (dmax[d] - dmin[d]) / (double)r[d]; // diagonal ./ n  :26
			inv_cell[d] = 1.0 / cell[d];                   // cwiseInverse :27
		}
		n_cells = (uint64_t)(r[0] * (r[1] * r[2])); // Eigen prod() of unsigned: x*(y*z) :28
	}
	unsigned nv() const { return (res[0] + 1) * (res[1] + 1) * (res[2] + 1); }
	unsigned nex() const { return res[0] * (res[1] + 1) * (res[2] + 1); }
	unsigned ney() const { return (res[0] + 1) * res[1] * (res[2] + 1); }
	unsigned nez() const { return (res[0] + 1) * (res[1] + 1) * res[2]; }
	unsigned n_nodes() const { return nv() + 2 * (nex() + ney() + nez()); } // cubic_lagrange_discrete_grid.cpp:790-796

	// cubic_lagrange_discrete_grid.cpp:604-665
	void node_position(unsigned l, double x[3]) const
	{
		const unsigned nx = res[0], ny = res[1], nz = res[2];
		unsigned ijk[3];
		int axis = -1;
		unsigned odd = 0;
		if (l < nv())
		{
			ijk[2] = l / ((ny + 1) * (nx + 1));
			unsigned t = l % ((ny + 1) * (nx + 1));
			ijk[1] = t / (nx + 1);
			ijk[0] = t % (nx + 1);
		}
		else if (l < nv() + 2 * nex())
		{
			l -= nv();
			odd = l % 2;
			unsigned e = l / 2;
			ijk[2] = e / ((ny + 1) * nx);
			unsigned t = e % ((ny + 1) * nx);
			ijk[1] = t / nx;
			ijk[0] = t % nx;
			axis = 0;
		}
		else if (l < nv() + 2 * (nex() + ney()))
		{
			l -= nv() + 2 * nex();
			odd = l % 2;
			unsigned e = l / 2;
			ijk[0] = e / ((nz + 1) * ny);
			unsigned t = e % ((nz + 1) * ny);
			ijk[2] = t / ny;
			ijk[1] = t % ny;
			axis = 1;
		}
		else
		{
			l -= nv() + 2 * (nex() + ney());
			odd = l % 2;
			unsigned e = l / 2;
			ijk[1] = e / ((nx + 1) * nz);
			unsigned t = e % ((nx + 1) * nz);
			ijk[0] = t / nz;
			ijk[2] = t % nz;
			axis = 2;
		}
		for (int d = 0; d < 3; ++d)
			x[d] = dmin[d] + cell[d] * (double)ijk[d];
		if (axis >= 0)
			x[axis] += (1.0 + (double)odd) / 3.0 * cell[axis];
	}

	// 32 node indices of cell (i,j,k): cubic_lagrange_discrete_grid.cpp:836-886
	void cell_nodes(unsigned l, unsigned out[32]) const
	{
		const unsigned nx = res[0], ny = res[1], nz = res[2];
		const unsigned k = l / (ny * nx);
		const unsigned t = l % (ny * nx);
		const unsigned j = t / nx;
		const unsigned i = t % nx;
		int o = 0;
		for (unsigned dk = 0; dk < 2; ++dk)
			for (unsigned dj = 0; dj < 2; ++dj)
				for (unsigned di = 0; di < 2; ++di)
					out[o++] = (nx + 1) * (ny + 1) * (k + dk) + (nx + 1) * (j + dj) + (i + di);
		unsigned off = nv();
		const unsigned xjk[4][2] = {{0, 0}, {0, 1}, {1, 0}, {1, 1}}; // (dj, dk) :858-865
		for (auto& a : xjk)
		{
			out[o] = off + 2 * (nx * (ny + 1) * (k + a[1]) + nx * (j + a[0]) + i);
			out[o + 1] = out[o] + 1;
			o += 2;
		}
		off += 2 * nex();
		const unsigned yik[4][2] = {{0, 0}, {1, 0}, {0, 1}, {1, 1}}; // (di, dk) :868-875
		for (auto& a : yik)
		{
			out[o] = off + 2 * (ny * (nz + 1) * (i + a[0]) + ny * (k + a[1]) + j);
			out[o + 1] = out[o] + 1;
			o += 2;
		}
		off += 2 * ney();
		const unsigned zji[4][2] = {{0, 0}, {1, 0}, {0, 1}, {1, 1}}; // (dj, di) :878-885
		for (auto& a : zji)
		{
			out[o] = off + 2 * (nz * (nx + 1) * (j + a[0]) + nz * (i + a[1]) + k);
			out[o + 1] = out[o] + 1;
			o += 2;
		}
	}
};

// ----------------------------------------------------------------------------------
// 32 serendipity-cubic shape functions and derivatives.
// cubic_lagrange_discrete_grid.cpp:339-580 (same products, same order; the sign
// pattern tables below replace the 96 hand-written derivative lines).
// ----------------------------------------------------------------------------------
void shape_functions(const double xi[3], double N[32], double* dN /* [32][3] row-major or null */)
{
	const double x = xi[0], y = xi[1], z = xi[2];
	const double x2 = x * x, y2 = y * y, z2 = z * z;
	const double mx = 1.0 - x, my = 1.0 - y, mz = 1.0 - z;
	const double px = 1.0 + x, py = 1.0 + y, pz = 1.0 + z;
	const double m3x = 1.0 - 3.0 * x, m3y = 1.0 - 3.0 * y, m3z = 1.0 - 3.0 * z;
	const double p3x = 1.0 + 3.0 * x, p3y = 1.0 + 3.0 * y, p3z = 1.0 + 3.0 * z;
	// pair products, index = (second sign)*2 + (first sign), sign 0 = minus, 1 = plus
	const double xy[4] = {mx * my, px * my, mx * py, px * py}; // _1mxt1my,_1pxt1my,_1mxt1py,_1pxt1py
	const double xz[4] = {mx * mz, px * mz, mx * pz, px * pz};
	const double yz[4] = {my * mz, py * mz, my * pz, py * pz};
	const double omx2 = 1.0 - x2, omy2 = 1.0 - y2, omz2 = 1.0 - z2;
	const double sx[2] = {mx, px}, sy[2] = {my, py}, sz[2] = {mz, pz};

	double fac = 1.0 / 64.0 * (9.0 * (x2 + y2 + z2) - 19.0);
	for (int c = 0; c < 8; ++c) // corner c: x sign = bit0, y sign = bit1, z sign = bit2 :389-396
		N[c] = fac * xy[c & 3] * sz[c >> 2];

	fac = 9.0 / 64.0 * omx2;
	const double fx[2] = {fac * m3x, fac * p3x};
	for (int e = 0; e < 4; ++e) // x-edges: (y,z) signs (-,-),(-,+),(+,-),(+,+) :403-410
	{
		const int ys = e >> 1, zs = e & 1;
		N[8 + 2 * e] = fx[0] * yz[zs * 2 + ys];
		N[9 + 2 * e] = fx[1] * yz[zs * 2 + ys];
	}
	fac = 9.0 / 64.0 * omy2;
	const double fy[2] = {fac * m3y, fac * p3y};
	for (int e = 0; e < 4; ++e) // y-edges: (x,z) signs (-,-),(+,-),(-,+),(+,+) :415-422
	{
		const int xs = e & 1, zs = e >> 1;
		N[16 + 2 * e] = fy[0] * xz[zs * 2 + xs];
		N[17 + 2 * e] = fy[1] * xz[zs * 2 + xs];
	}
	fac = 9.0 / 64.0 * omz2;
	const double fz[2] = {fac * m3z, fac * p3z};
	for (int e = 0; e < 4; ++e) // z-edges: (x,y) signs (-,-),(-,+),(+,-),(+,+) :427-434
	{
		const int xs = e >> 1, ys = e & 1;
		N[24 + 2 * e] = fz[0] * xy[ys * 2 + xs];
		N[25 + 2 * e] = fz[1] * xy[ys * 2 + xs];
	}
	if (!dN)
		return;

	auto D = [dN](int r, int c) -> double& { return dN[3 * r + c]; };
	const double gx = 9.0 * (3.0 * x2 + y2 + z2) - 19.0; // :440-442
	const double gy = 9.0 * (x2 + 3.0 * y2 + z2) - 19.0;
	const double gz = 9.0 * (x2 + y2 + 3.0 * z2) - 19.0;
	const double x18 = 18.0 * x, y18 = 18.0 * y, z18 = 18.0 * z;
	const double hx[2] = {x18 - gx, x18 + gx}; // :455-460
	const double hy[2] = {y18 - gy, y18 + gy};
	const double hz[2] = {z18 - gz, z18 + gz};
	for (int c = 0; c < 8; ++c) // :462-487
	{
		const int xs = c & 1, ys = (c >> 1) & 1, zs = c >> 2;
		D(c, 0) = hx[xs] * yz[zs * 2 + ys];
		D(c, 1) = xz[zs * 2 + xs] * hy[ys];
		D(c, 2) = xy[ys * 2 + xs] * hz[zs];
		for (int d = 0; d < 3; ++d)
			D(c, d) /= 64.0;
	}
	const double t3x = 3.0 - 9.0 * x2, t3y = 3.0 - 9.0 * y2, t3z = 3.0 - 9.0 * z2; // :447-449
	const double x2_ = 2.0 * x, y2_ = 2.0 * y, z2_ = 2.0 * z;
	const double qx[2] = {-t3x - x2_, t3x - x2_}; // _m3m9x2m2x, _p3m9x2m2x :489-490
	const double qy[2] = {-t3y - y2_, t3y - y2_};
	const double qz[2] = {-t3z - z2_, t3z - z2_};
	const double wx[2] = {omx2 * m3x, omx2 * p3x}; // :491-492
	const double wy[2] = {omy2 * m3y, omy2 * p3y};
	const double wz[2] = {omz2 * m3z, omz2 * p3z};
	for (int e = 0; e < 4; ++e) // x-edges :493-516
	{
		const int ys = e >> 1, zs = e & 1;
		for (int h = 0; h < 2; ++h)
		{
			const int r = 8 + 2 * e + h;
			D(r, 0) = qx[h] * yz[zs * 2 + ys];
			D(r, 1) = (ys ? wx[h] : -wx[h]) * sz[zs];
			D(r, 2) = (zs ? wx[h] : -wx[h]) * sy[ys];
		}
	}
	for (int e = 0; e < 4; ++e) // y-edges :522-545
	{
		const int xs = e & 1, zs = e >> 1;
		for (int h = 0; h < 2; ++h)
		{
			const int r = 16 + 2 * e + h;
			D(r, 0) = (xs ? wy[h] : -wy[h]) * sz[zs];
			D(r, 1) = qy[h] * xz[zs * 2 + xs];
			D(r, 2) = (zs ? wy[h] : -wy[h]) * sx[xs];
		}
	}
	for (int e = 0; e < 4; ++e) // z-edges :551-574
	{
		const int xs = e >> 1, ys = e & 1;
		for (int h = 0; h < 2; ++h)
		{
			const int r = 24 + 2 * e + h;
			D(r, 0) = (xs ? wz[h] : -wz[h]) * sy[ys];
			D(r, 1) = (ys ? wz[h] : -wz[h]) * sx[xs];
			D(r, 2) = qz[h] * xy[ys * 2 + xs];
		}
	}
	for (int r = 8; r < 32; ++r) // bottomRows(24) *= 9/64 :576
		for (int d = 0; d < 3; ++d)
			D(r, d) *= 9.0 / 64.0;
}

const double kNoValue = std::numeric_limits<double>::max();

// interpolate(field, x, grad*), cubic_lagrange_discrete_grid.cpp:977-1063.
// cells/cell_map may be null => identity map + closed-form cell rows (what the
// reference's own table holds before any reduceField, :833-891).
double interpolate_one(const Grid& g, const double* coeffs, const unsigned* cells, const unsigned* cell_map,
					   const double x[3], double* grad)
{
	for (int d = 0; d < 3; ++d) // AlignedBox::contains, inclusive :981
		if (!(g.dmin[d] <= x[d] && x[d] <= g.dmax[d]))
			return kNoValue;
	unsigned mi[3];
	for (int d = 0; d < 3; ++d)
	{
		mi[d] = (unsigned)((x[d] - g.dmin[d]) * g.inv_cell[d]); // :984
		if (mi[d] >= g.res[d])
			mi[d] = g.res[d] - 1;
	}
	unsigned ci = g.res[1] * g.res[0] * mi[2] + g.res[0] * mi[1] + mi[0]; // discrete_grid.cpp:21-24
	unsigned cm = cell_map ? cell_map[ci] : ci;
	if (cm == std::numeric_limits<unsigned>::max())
		return kNoValue;
	double c0[3], xi[3];
	for (int d = 0; d < 3; ++d)
	{
		const double lo = g.dmin[d] + (double)mi[d] * g.cell[d]; // subdomain origin discrete_grid.cpp:29-31
		const double hi = lo + g.cell[d];
		const double denom = hi - lo;  // :1000
		c0[d] = 2.0 / denom;           // :1001
		const double c1 = (hi + lo) / denom; // :1002
		xi[d] = c0[d] * x[d] - c1;     // :1003
	}
	unsigned row[32];
	const unsigned* cell = row;
	if (cells)
		cell = cells + 32 * (size_t)cm;
	else
		g.cell_nodes(cm, row);
	double N[32], dN[96];
	shape_functions(xi, N, grad ? dN : nullptr);
	double phi = 0.0;
	if (!grad)
	{
		for (int j = 0; j < 32; ++j) // :1011-1020
		{
			const double c = coeffs[cell[j]];
			if (c == kNoValue)
				return kNoValue;
			phi += c * N[j];
		}
		return phi;
	}
	grad[0] = grad[1] = grad[2] = 0.0;
	for (int j = 0; j < 32; ++j) // :1046-1059
	{
		const double c = coeffs[cell[j]];
		if (c == kNoValue)
		{
			grad[0] = grad[1] = grad[2] = 0.0;
			return kNoValue;
		}
		phi += c * N[j];
		grad[0] += c * dN[3 * j + 0];
		grad[1] += c * dN[3 * j + 1];
		grad[2] += c * dN[3 * j + 2];
	}
	for (int d = 0; d < 3; ++d) // :1060
		grad[d] *= c0[d];
	return phi;
}

// ----------------------------------------------------------------------------------
// Density map node function (cmd/generate_density_map): CubicKernel::W sph_kernel.hpp:22-42
// (r.norm() with Eigen's 3-vector association), gamma main.cpp:86-93, density_func :96-112,
// tensor Gauss-Legendre rule gauss_quadrature.cpp:5927-5960, node predicate main.cpp:119-133.
// ----------------------------------------------------------------------------------
struct Density
{
	double h, rho0, k, cell_diag;
	int band;
	const double* gx; // 16 abscissae (ascending) and weights of the reference's p = 30 rule
	const double* gw;
	double W(const double r[3]) const
	{
		double res = 0.0;
		const double rl = std::sqrt(r[0] * r[0] + (r[1] * r[1] + r[2] * r[2]));
		const double q = rl / h;
		if (q <= 1.0)
		{
			if (q <= 0.5)
			{
				const double q2 = q * q, q3 = q2 * q;
				res = k * (6.0 * q3 - 6.0 * q2 + 1.0);
			}
			else
			{
				const double m = 1.0 - q;
				res = k * (2.0 * m * m * m);
			}
		}
		return res;
	}
};

double density_node(const Grid& g, const double* coeffs, const unsigned* cells, const unsigned* map, const Density& D,
					const double x[3])
{
	if (D.band) // main.cpp:119-133
	{
		double xc[3];
		for (int d = 0; d < 3; ++d)
			xc[d] = std::min(std::max(x[d], g.dmin[d]), g.dmax[d]);
		const double dist = interpolate_one(g, coeffs, cells, map, xc, nullptr);
		if (dist == kNoValue)
			return kNoValue;
		if (!(-6.0 * D.h < dist + D.cell_diag && dist - D.cell_diag < 2.0 * D.h))
			return kNoValue;
	}
	const double dist = interpolate_one(g, coeffs, cells, map, x, nullptr);
	if (dist > 2.0 * D.h)
		return 0.0;
	const double c0 = 0.5 * (D.h - (-D.h));   // 0.5 * diagonal
	const double c1 = 0.5 * (-D.h + D.h);     // 0.5 * (min + max)
	double res = 0.0;
	double xi[3];
	for (int i = 0; i < 16; ++i)
	{
		const double wi = D.gw[i];
		xi[0] = D.gx[i];
		for (int j = 0; j < 16; ++j)
		{
			const double wij = wi * D.gw[j];
			xi[1] = D.gx[j];
			for (int k = 0; k < 16; ++k)
			{
				const double wijk = wij * D.gw[k];
				xi[2] = D.gx[k];
				const double r[3] = {c0 * xi[0] + c1, c0 * xi[1] + c1, c0 * xi[2] + c1};
				const double y[3] = {x[0] + r[0], x[1] + r[1], x[2] + r[2]};
				const double d = interpolate_one(g, coeffs, cells, map, y, nullptr);
				const double gamma = (d > D.h) ? 0.0 : 1.0 - d / D.h;
				res += wijk * (gamma * D.W(r));
			}
		}
	}
	res *= c0 * (c0 * c0); // Eigen prod() of a 3-vector
	return D.rho0 * res;
}

template <class T>
void put(std::vector<unsigned char>& b, const T& v)
{
	const unsigned char* p = reinterpret_cast<const unsigned char*>(&v);
	b.insert(b.end(), p, p + sizeof(T));
}

} // namespace

// ======================================================================================
// C entry points (ctypes-friendly).
// ======================================================================================
extern "C"
{

// cmd/generate_sdf/main.cpp:83-91 with Eigen's 3-vector norm association x^2+(y^2+z^2).
void dgo_default_domain(const double* verts, size_t nv, double out[6])
{
	double lo[3], hi[3];
	for (int d = 0; d < 3; ++d)
	{
		lo[d] = std::numeric_limits<double>::max();
		hi[d] = std::numeric_limits<double>::lowest();
	}
	for (size_t i = 0; i < nv; ++i)
		for (int d = 0; d < 3; ++d)
		{
			lo[d] = std::min(lo[d], verts[3 * i + d]);
			hi[d] = std::max(hi[d], verts[3 * i + d]);
		}
	auto diag_norm = [&]() {
		const double dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
		return std::sqrt(dx * dx + (dy * dy + dz * dz));
	};
	const double grow_max = 1.0e-3 * diag_norm();
	for (int d = 0; d < 3; ++d)
		hi[d] += grow_max * 1.0;
	const double grow_min = 1.0e-3 * diag_norm(); // uses the already grown max (asymmetric)
	for (int d = 0; d < 3; ++d)
		lo[d] -= grow_min * 1.0;
	for (int d = 0; d < 3; ++d)
	{
		out[d] = lo[d];
		out[3 + d] = hi[d];
	}
}

// TriangleMeshDistance(TriangleMesh const&) path: TriangleMeshDistance.h:227-230 -> :252-267
void* dgo_mesh_create(const double* verts, size_t nv, const unsigned* faces, size_t nf)
{
	if (!nf)
		return nullptr; // reference prints + exit(-1) :338-341
	auto m = new MeshDistance;
	m->verts.resize(nv);
	for (size_t i = 0; i < nv; ++i)
		m->verts[i] = {verts[3 * i], verts[3 * i + 1], verts[3 * i + 2]};
	m->tris.resize(nf);
	for (size_t i = 0; i < nf; ++i)
		m->tris[i] = {{(int)faces[3 * i], (int)faces[3 * i + 1], (int)faces[3 * i + 2]}};
	m->construct();
	return m;
}
void dgo_mesh_free(void* h) { delete static_cast<MeshDistance*>(h); }
int dgo_mesh_watertight_flags(void* h)
{
	auto m = static_cast<MeshDistance*>(h);
	return (m->single_edge ? 1 : 0) | (m->triple_edge ? 2 : 0);
}
void dgo_mesh_sizes(void* h, size_t* n_nodes, size_t* n_tris, size_t* n_verts)
{
	auto m = static_cast<MeshDistance*>(h);
	*n_nodes = m->nodes.size();
	*n_tris = m->tris.size();
	*n_verts = m->verts.size();
}
void dgo_mesh_get(void* h, double* pn_tri, double* pn_edge, double* pn_vert, double* node_spheres, int* node_children)
{
	auto& m = *static_cast<MeshDistance*>(h);
	for (size_t i = 0; i < m.tris.size(); ++i)
		for (int d = 0; d < 3; ++d)
		{
			pn_tri[3 * i + d] = m.pn_tri[i][d];
			for (int e = 0; e < 3; ++e)
				pn_edge[9 * i + 3 * e + d] = m.pn_edge[i][e][d];
		}
	for (size_t i = 0; i < m.verts.size(); ++i)
		for (int d = 0; d < 3; ++d)
			pn_vert[3 * i + d] = m.pn_vert[i][d];
	for (size_t i = 0; i < m.nodes.size(); ++i)
	{
		const auto& n = m.nodes[i];
		for (int d = 0; d < 3; ++d)
		{
			node_spheres[8 * i + d] = n.left_bv.c[d];
			node_spheres[8 * i + 4 + d] = n.right_bv.c[d];
		}
		node_spheres[8 * i + 3] = n.left_bv.r;
		node_spheres[8 * i + 7] = n.right_bv.r;
		node_children[2 * i] = n.left;
		node_children[2 * i + 1] = n.right;
	}
}

// Batch signed_distance.  visits (nullable) receives {sum inner, sum leaf} node visits
// of the reference traversal -- the V-bar / L-bar of SURVEY.md section 8(d).
int dgo_signed_distance(void* h, const double* xyz, size_t n, double* dist, int* tri, int* entity,
						double* nearest, uint64_t* visits)
{
	auto& m = *static_cast<MeshDistance*>(h);
	uint64_t vi = 0, vl = 0;
#pragma omp parallel for schedule(dynamic, 256) reduction(+ : vi, vl)
	for (long long q = 0; q < (long long)n; ++q)
	{
		Counters c;
		const V3 p = {xyz[3 * q], xyz[3 * q + 1], xyz[3 * q + 2]};
		QueryResult r = visits ? signed_distance<true>(m, p, &c) : signed_distance<false>(m, p, nullptr);
		dist[q] = r.distance;
		if (tri)
			tri[q] = r.tri;
		if (entity)
			entity[q] = r.entity;
		if (nearest)
		{
			nearest[3 * q] = r.nearest.x;
			nearest[3 * q + 1] = r.nearest.y;
			nearest[3 * q + 2] = r.nearest.z;
		}
		vi += c.inner;
		vl += c.leaf;
	}
	if (visits)
	{
		visits[0] = vi;
		visits[1] = vl;
	}
	return 0;
}

// Squared distance + closest feature for explicit (point, triangle) pairs.
void dgo_point_triangle(const double* p, const double* v0, const double* v1, const double* v2, size_t n, double* d2,
						double* st, int* entity, double* nearest)
{
	for (size_t i = 0; i < n; ++i)
	{
		V3 np;
		Closest c = closest_on_triangle({p[3 * i], p[3 * i + 1], p[3 * i + 2]}, {v0[3 * i], v0[3 * i + 1], v0[3 * i + 2]},
										{v1[3 * i], v1[3 * i + 1], v1[3 * i + 2]},
										{v2[3 * i], v2[3 * i + 1], v2[3 * i + 2]}, &np);
		d2[i] = c.d2;
		if (st)
		{
			st[2 * i] = c.s;
			st[2 * i + 1] = c.t;
		}
		if (entity)
			entity[i] = c.entity;
		if (nearest)
		{
			nearest[3 * i] = np.x;
			nearest[3 * i + 1] = np.y;
			nearest[3 * i + 2] = np.z;
		}
	}
}

unsigned dgo_n_nodes(const unsigned res[3])
{
	Grid g;
	const double dom[6] = {0, 0, 0, 1, 1, 1};
	g.init(dom, res);
	return g.n_nodes();
}

void dgo_grid_header(const double domain[6], const unsigned res[3], double cell[3], double inv_cell[3])
{
	Grid g;
	g.init(domain, res);
	for (int d = 0; d < 3; ++d)
	{
		cell[d] = g.cell[d];
		inv_cell[d] = g.inv_cell[d];
	}
}

void dgo_node_positions(const double domain[6], const unsigned res[3], unsigned begin, unsigned end, double* xyz)
{
	Grid g;
	g.init(domain, res);
	for (unsigned l = begin; l < end; ++l)
		g.node_position(l, xyz + 3 * (size_t)(l - begin));
}

// The node-sampling loop of addFunction, cubic_lagrange_discrete_grid.cpp:806-831, for
// func = +-signed_distance (cmd/generate_sdf/main.cpp:95-101).  Same `omp for
// schedule(static)`.  Returns wall seconds.  visits as in dgo_signed_distance.
double dgo_sample_nodes(void* h, const double domain[6], const unsigned res[3], int invert, unsigned begin,
						unsigned end, double* out, uint64_t* visits)
{
	auto& m = *static_cast<MeshDistance*>(h);
	Grid g;
	g.init(domain, res);
	uint64_t vi = 0, vl = 0;
	auto t0 = std::chrono::high_resolution_clock::now();
#pragma omp parallel for schedule(static) reduction(+ : vi, vl)
	for (long long l = begin; l < (long long)end; ++l)
	{
		double x[3];
		g.node_position((unsigned)l, x);
		Counters c;
		QueryResult r = visits ? signed_distance<true>(m, {x[0], x[1], x[2]}, &c)
							   : signed_distance<false>(m, {x[0], x[1], x[2]}, nullptr);
		out[l - begin] = invert ? -1.0 * r.distance : r.distance;
		vi += c.inner;
		vl += c.leaf;
	}
	auto t1 = std::chrono::high_resolution_clock::now();
	if (visits)
	{
		visits[0] = vi;
		visits[1] = vl;
	}
	return std::chrono::duration<double>(t1 - t0).count();
}

void dgo_cell_table(const unsigned res[3], unsigned cell_begin, unsigned cell_end, unsigned* out /* 32 per cell */)
{
	Grid g;
	const double dom[6] = {0, 0, 0, 1, 1, 1};
	g.init(dom, res);
	for (unsigned l = cell_begin; l < cell_end; ++l)
		g.cell_nodes(l, out + 32 * (size_t)(l - cell_begin));
}

void dgo_shape_functions(const double* xi, size_t n, double* N /*32n*/, double* dN /*96n or null*/)
{
	for (size_t i = 0; i < n; ++i)
		shape_functions(xi + 3 * i, N + 32 * i, dN ? dN + 96 * i : nullptr);
}

double dgo_interpolate(const double domain[6], const unsigned res[3], const double* coeffs, const unsigned* cells,
					   const unsigned* cell_map, const double* xyz, size_t n, double* phi, double* grad)
{
	Grid g;
	g.init(domain, res);
	auto t0 = std::chrono::high_resolution_clock::now();
#pragma omp parallel for schedule(static)
	for (long long q = 0; q < (long long)n; ++q)
		phi[q] = interpolate_one(g, coeffs, cells, cell_map, xyz + 3 * q, grad ? grad + 3 * q : nullptr);
	auto t1 = std::chrono::high_resolution_clock::now();
	return std::chrono::duration<double>(t1 - t0).count();
}

double dgo_density_map_nodes(const double domain[6], const unsigned res[3], const double* coeffs, const unsigned* cells,
							 const unsigned* cell_map, double h, double rho0, int band, const double* gx, const double* gw,
							 unsigned begin, unsigned end, double* out)
{
	Grid g;
	g.init(domain, res);
	Density D;
	D.h = h;
	D.rho0 = rho0;
	D.band = band;
	D.gx = gx;
	D.gw = gw;
	const double pi = 3.14159265358979323846;
	D.k = 8.0 / (pi * (h * h * h)); // sph_kernel.hpp:14-17
	D.cell_diag = std::sqrt(g.cell[0] * g.cell[0] + (g.cell[1] * g.cell[1] + g.cell[2] * g.cell[2]));
	auto t0 = std::chrono::high_resolution_clock::now();
#pragma omp parallel for schedule(dynamic, 16)
	for (long long l = begin; l < (long long)end; ++l)
	{
		double x[3];
		g.node_position((unsigned)l, x);
		out[l - begin] = density_node(g, coeffs, cells, cell_map, D, x);
	}
	auto t1 = std::chrono::high_resolution_clock::now();
	return std::chrono::duration<double>(t1 - t0).count();
}

// Serialised grid with n_fields fields, all unreduced (identity cell map), in the
// reference's packed little-endian layout: cubic_lagrange_discrete_grid.cpp:678-719,
// utility/serialize.hpp:11-37.  Returns bytes written, or 0 on error.
size_t dgo_write_cdf(const char* path, const double domain[6], const unsigned res[3], const double* const* fields,
					 size_t n_fields)
{
	Grid g;
	g.init(domain, res);
	std::vector<unsigned char> b;
	for (int d = 0; d < 6; ++d)
		put(b, domain[d]);
	for (int d = 0; d < 3; ++d)
		put(b, res[d]);
	for (int d = 0; d < 3; ++d)
		put(b, g.cell[d]);
	for (int d = 0; d < 3; ++d)
		put(b, g.inv_cell[d]);
	put(b, (uint64_t)g.n_cells);
	put(b, (uint64_t)n_fields);
	const uint64_t nn = g.n_nodes();
	put(b, (uint64_t)n_fields);
	for (size_t f = 0; f < n_fields; ++f)
	{
		put(b, nn);
		const unsigned char* p = reinterpret_cast<const unsigned char*>(fields[f]);
		b.insert(b.end(), p, p + nn * sizeof(double));
	}
	put(b, (uint64_t)n_fields);
	std::vector<unsigned> row(32);
	for (size_t f = 0; f < n_fields; ++f)
	{
		put(b, (uint64_t)g.n_cells);
		for (unsigned l = 0; l < g.n_cells; ++l)
		{
			g.cell_nodes(l, row.data());
			const unsigned char* p = reinterpret_cast<const unsigned char*>(row.data());
			b.insert(b.end(), p, p + 32 * sizeof(unsigned));
		}
	}
	put(b, (uint64_t)n_fields);
	for (size_t f = 0; f < n_fields; ++f)
	{
		put(b, (uint64_t)g.n_cells);
		for (unsigned l = 0; l < g.n_cells; ++l)
			put(b, l);
	}
	FILE* fp = std::fopen(path, "wb");
	if (!fp)
		return 0;
	size_t w = std::fwrite(b.data(), 1, b.size(), fp);
	std::fclose(fp);
	return w;
}

// Query points of BASELINE config 5: n points uniform in the box [lo, hi], drawn with
// std::mt19937_64 + std::uniform_real_distribution<double> (x, y, z per point, in that order).  The
// same libstdc++ runs on the build container and on the GPU box, so both sides see the same points.
void dgo_uniform_points(uint64_t seed, size_t n, const double lo[3], const double hi[3], double* out)
{
	std::mt19937_64 gen(seed);
	std::uniform_real_distribution<double> ux(lo[0], hi[0]), uy(lo[1], hi[1]), uz(lo[2], hi[2]);
	for (size_t i = 0; i < n; ++i)
	{
		out[3 * i] = ux(gen);
		out[3 * i + 1] = uy(gen);
		out[3 * i + 2] = uz(gen);
	}
}

} // extern "C"
