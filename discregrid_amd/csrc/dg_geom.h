// dg_geom.h -- per-lane arithmetic shared by the HIP kernels (device) and the host-side
// structure builders.  Everything here is written for bit parity with the reference's CPU
// path: every floating-point expression keeps the reference's association order and must be
// compiled with -ffp-contract=off (an FMA changes d^2 by up to 8e-9 relative near the zero
// level set).  No function in this header touches memory other than its
// arguments.
//
// This header: the mesh side -- bound records, triangle packets, the point/triangle test and the
// per-lane query state (reference: discregrid/include/Discregrid/geometry/TriangleMeshDistance.h:564-820).
// The grid side is in dg_lattice.h (node positions, shape functions, interpolation) and
// dg_density.h (the density-map integrand).
#pragma once
#include <stddef.h>
#include <stdint.h>

#if defined(__HIP__)
#define DG_HD __host__ __device__ __attribute__((always_inline)) inline
#define DG_NOUNROLL _Pragma("nounroll")
#else
#define DG_HD inline
#define DG_NOUNROLL
#endif

namespace dg
{

// ---- data layouts (shared by host builder and device kernels) ------------------------------
// Bounds come in PAIRS: one 128-byte record holds the bounds of two sibling BVH nodes (or of two
// neighbouring triangles of a leaf), interleaved field by field, so that (a) one scalar load
// fetches both and (b) the wave tests both with packed two-wide float instructions
// (v_pk_add/mul/fma_f32) -- the kernel is VALU-issue bound, every packed instruction saved is
// time saved.  All bounds are float, relative to the mesh origin, rounded OUTWARD: they are only
// ever used to prune, so their arithmetic is free.  (The first design -- an axis-aligned box AND one slab along the
// mean normal, tight on the concave side of a curved surface but unable to separate neighbouring facets seen from far
// away -- took 45 ms against 22 ms per 256^3 launch; docs/DESIGN_history_r1_r3.md.)
// Bounds of an item (subtree or triangle) = an oriented box: centre c, directions u_a, half widths
//     |u_a . (y - c)| <= half_a   for every point y of the item,  a = 0, 1, 2,
// with (nearly) orthonormal directions u_a whose Gram matrix has no eigenvalue above 1, so that
//     dist(p, item)^2 >= sum_a max(|u_a.(p - c)| - half_a, 0)^2.
// For a flat patch u_0 is its mean normal and u_1, u_2 the principal tangent directions; for a
// curved one the three directions are the coordinate axes (an ordinary box).
struct alignas(128) PairRec
{
	float f[15][2];  // [k][side], k: 0..2 centre, 3 + 3*a + d component d of u_a, 12 + a half_a (an empty item has half_a = -FLT_MAX)
	int32_t info[2]; // node pairs: what is below each side (see below); triangle pairs: unused
};
static const int kPairFloats = 30;
static_assert(sizeof(PairRec) == 128, "PairRec must be 128 bytes");
// info word of a subtree:  >= 0: index of the PairRec holding its two children;
//                          <  0: leaf, ~info = (first_position << kLeafBits) | (positions - 1),
// where positions (<= 16, even) counts triangle slots in leaf order including one padding slot
// for leaves with an odd triangle count (padding slots have empty bounds and are never tested).
static const int kLeafBits = 4;
static const int kMaxLeaf = 16;
#ifndef DG_SUBTREES
#define DG_SUBTREES 256
#endif
static const int kSubtrees = DG_SUBTREES; // disjoint subtrees the builder cuts the tree into (dg_kernels.h, "Heavy bricks")

// One triangle packet = 128 bytes, in BVH leaf order.  The point-independent terms of the
// Eberly test are precomputed on the host WITH THE REFERENCE'S OWN OPERATIONS (same inputs,
// same order => same bits as recomputing them per query, TriangleMeshDistance.h:566-575,
// 675, 693).
struct alignas(128) TriPacket
{
	double v0[3];
	double e0[3]; // v1 - v0
	double e1[3]; // v2 - v0
	double a00, a01, a11;
	double det;     // |a00*a11 - a01*a01|
	double inv_det; // 1 / det
	double denom;   // a00 - 2*a01 + a11
	int32_t tri_id; // index in the caller's triangle array
	int32_t pad_;
};
static_assert(sizeof(TriPacket) == 128, "TriPacket must be 128 bytes");

// Pseudonormals, 8 slots of 3 doubles per triangle (leaf order), slot = nearest entity:
// 0..2 vertex normals of v0,v1,v2; 3..5 edge normals E01,E12,E02; 6 face normal; 7 unused.
static const int kPnSlots = 8;

enum Entity : int { kV0 = 0, kV1 = 1, kV2 = 2, kE01 = 3, kE12 = 4, kE02 = 5, kFace = 6 };

DG_HD void make_packet(const double v0[3], const double v1[3], const double v2[3], int32_t id, TriPacket& t)
{
	for (int d = 0; d < 3; ++d)
	{
		t.v0[d] = v0[d];
		t.e0[d] = v1[d] - v0[d];
		t.e1[d] = v2[d] - v0[d];
	}
	t.a00 = t.e0[0] * t.e0[0] + t.e0[1] * t.e0[1] + t.e0[2] * t.e0[2];
	t.a01 = t.e0[0] * t.e1[0] + t.e0[1] * t.e1[1] + t.e0[2] * t.e1[2];
	t.a11 = t.e1[0] * t.e1[0] + t.e1[1] * t.e1[1] + t.e1[2] * t.e1[2];
	const double dd = t.a00 * t.a11 - t.a01 * t.a01;
	t.det = __builtin_fabs(dd); // std::abs
	t.inv_det = 1 / t.det;
	t.denom = t.a00 - 2 * t.a01 + t.a11;
	t.tri_id = id;
	t.pad_ = 0;
}

// ---- squared distance point <-> triangle ------------------------------------------------------
// FULL = false: only d^2 (the traversal inner loop).  FULL = true: also the barycentric
// parameters and the nearest entity (run once per query on the winning triangle).
struct Hit
{
	double d2, s, t;
	int entity;
};

template <bool FULL>
DG_HD Hit tri_closest(double v0x, double v0y, double v0z, double e0x, double e0y, double e0z, double e1x, double e1y,
					  double e1z, double a00, double a01, double a11, double det, double inv_det, double denom,
					  double px, double py, double pz)
{
	const double dx = v0x - px, dy = v0y - py, dz = v0z - pz;
	const double b0 = dx * e0x + dy * e0y + dz * e0z;
	const double b1 = dx * e1x + dy * e1y + dz * e1z;
	const double c = dx * dx + dy * dy + dz * dz;
	double s = a01 * b1 - a11 * b0;
	double t = a01 * b0 - a00 * b1;
	// vertex / edge candidates shared by several regions
	const double d2_v1 = a00 + 2 * b0 + c;
	const double d2_v2 = a11 + 2 * b1 + c;
	Hit h;
	h.s = 0;
	h.t = 0;
	h.entity = kV0;
	h.d2 = c;
	if (s + t <= det)
	{
		if (s < 0 || t < 0)
		{
			// regions 3, 4, 5: the minimum lies on edge v0-v1 (t = 0) or v0-v2 (s = 0)
			const bool use01 = (s < 0) ? (t < 0 && b0 < 0) : true;
			if (use01)
			{
				// region 4 enters with b0 < 0 already known; region 5 tests it (same outcome)
				if (b0 >= 0)
				{ /* V0 */
				}
				else if (-b0 >= a00)
				{
					h.d2 = d2_v1;
					h.s = 1;
					h.entity = kV1;
				}
				else
				{
					h.s = -b0 / a00;
					h.d2 = b0 * h.s + c;
					h.entity = kE01;
				}
			}
			else
			{
				if (b1 >= 0)
				{ /* V0 */
				}
				else if (-b1 >= a11)
				{
					h.d2 = d2_v2;
					h.t = 1;
					h.entity = kV2;
				}
				else
				{
					h.t = -b1 / a11;
					h.d2 = b1 * h.t + c;
					h.entity = kE02;
				}
			}
		}
		else
		{
			// region 0: interior
			s *= inv_det;
			t *= inv_det;
			h.d2 = s * (a00 * s + a01 * t + 2 * b0) + t * (a01 * s + a11 * t + 2 * b1) + c;
			h.s = s;
			h.t = t;
			h.entity = kFace;
		}
	}
	else if (s < 0)
	{
		// region 2
		const double tmp0 = a01 + b0;
		const double tmp1 = a11 + b1;
		if (tmp1 > tmp0)
		{
			const double numer = tmp1 - tmp0;
			if (numer >= denom)
			{
				h.d2 = d2_v1;
				h.s = 1;
				h.entity = kV1;
			}
			else
			{
				s = numer / denom;
				t = 1 - s;
				h.d2 = s * (a00 * s + a01 * t + 2 * b0) + t * (a01 * s + a11 * t + 2 * b1) + c;
				h.s = s;
				h.t = t;
				h.entity = kE12;
			}
		}
		else if (tmp1 <= 0)
		{
			h.d2 = d2_v2;
			h.t = 1;
			h.entity = kV2;
		}
		else if (b1 >= 0)
		{ /* V0 */
		}
		else
		{
			h.t = -b1 / a11;
			h.d2 = b1 * h.t + c;
			h.entity = kE02;
		}
	}
	else if (t < 0)
	{
		// region 6
		const double tmp0 = a01 + b1;
		const double tmp1 = a00 + b0;
		if (tmp1 > tmp0)
		{
			const double numer = tmp1 - tmp0;
			if (numer >= denom)
			{
				h.d2 = d2_v2;
				h.t = 1;
				h.entity = kV2;
			}
			else
			{
				t = numer / denom;
				s = 1 - t;
				h.d2 = s * (a00 * s + a01 * t + 2 * b0) + t * (a01 * s + a11 * t + 2 * b1) + c;
				h.s = s;
				h.t = t;
				h.entity = kE12;
			}
		}
		else if (tmp1 <= 0)
		{
			h.d2 = d2_v1;
			h.s = 1;
			h.entity = kV1;
		}
		else if (b0 >= 0)
		{ /* V0 */
		}
		else
		{
			h.s = -b0 / a00;
			h.d2 = b0 * h.s + c;
			h.entity = kE01;
		}
	}
	else
	{
		// region 1
		const double numer = a11 + b1 - a01 - b0;
		if (numer <= 0)
		{
			h.d2 = d2_v2;
			h.t = 1;
			h.entity = kV2;
		}
		else if (numer >= denom)
		{
			h.d2 = d2_v1;
			h.s = 1;
			h.entity = kV1;
		}
		else
		{
			s = numer / denom;
			t = 1 - s;
			h.d2 = s * (a00 * s + a01 * t + 2 * b0) + t * (a01 * s + a11 * t + 2 * b1) + c;
			h.s = s;
			h.t = t;
			h.entity = kE12;
		}
	}
	if (h.d2 < 0) // "account for numerical round-off error"
		h.d2 = 0;
	return h;
}

template <bool FULL>
DG_HD Hit tri_closest(const TriPacket& T, double px, double py, double pz)
{
	return tri_closest<FULL>(T.v0[0], T.v0[1], T.v0[2], T.e0[0], T.e0[1], T.e0[2], T.e1[0], T.e1[1], T.e1[2], T.a00,
							 T.a01, T.a11, T.det, T.inv_det, T.denom, px, py, pz);
}

// ---- conservative float box test ------------------------------------------------------------------
// The query point, relative to the mesh origin, is carried as an interval [plo, phi] in float
// that strictly contains it; node boxes are rounded outward.  The returned value is a lower
// bound of the squared distance to anything inside the box up to a relative 2^-22, which the
// caller absorbs by inflating its float copy of the running best (best_as_float()).
struct FPoint
{
	float lo[3], hi[3];
	float x[3]; // the rounded point itself (slab test)
	float es;   // absolute error bound of a slab projection u.x, incl. the subtractions against lo/hi
};
// r = p - origin (double); mesh_l1 = max L1 norm of (vertex - origin) over the mesh
DG_HD FPoint make_fpoint(double rx, double ry, double rz, float mesh_l1)
{
	FPoint f;
	const float x = (float)rx, y = (float)ry, z = (float)rz;
	const float ax = x < 0 ? -x : x, ay = y < 0 ? -y : y, az = z < 0 ? -z : z;
	float m = ax > ay ? ax : ay;
	m = m > az ? m : az;
	const float e = m * 4.76837158203125e-07f + 1.0e-37f; // 2^-21 * max|coord|  (>= 4 float ulps)
	f.x[0] = x;
	f.x[1] = y;
	f.x[2] = z;
	f.es = ((ax + ay) + az + mesh_l1) * 9.5367431640625e-07f + 1.0e-37f; // 2^-20 * (|p|_1 + |mesh|_1)
	f.lo[0] = x - e;
	f.lo[1] = y - e;
	f.lo[2] = z - e;
	f.hi[0] = x + e;
	f.hi[1] = y + e;
	f.hi[2] = z + e;
	return f;
}
// The bound arithmetic only prunes, so it may use fused multiply-adds (smaller rounding error,
// fewer VALU instructions) and the hardware max (v_max_f32 / v_max3_f32; a NaN operand yields
// the other operand, i.e. bound 0 = never prune).
#if defined(__clang__)
typedef float f2 __attribute__((ext_vector_type(2))); // two-wide: v_pk_*_f32 on gfx950
DG_HD f2 f2_make(float a, float b) { f2 r; r.x = a; r.y = b; return r; }
DG_HD f2 f2_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
#else
struct f2
{
	float x, y;
};
DG_HD f2 f2_make(float a, float b) { return f2{a, b}; }
DG_HD f2 operator-(f2 a, f2 b) { return f2{a.x - b.x, a.y - b.y}; }
DG_HD f2 operator+(f2 a, f2 b) { return f2{a.x + b.x, a.y + b.y}; }
DG_HD f2 operator*(f2 a, f2 b) { return f2{a.x * b.x, a.y * b.y}; }
DG_HD f2 f2_fma(f2 a, f2 b, f2 c) { return f2{__builtin_fmaf(a.x, b.x, c.x), __builtin_fmaf(a.y, b.y, c.y)}; }
#endif
DG_HD f2 f2_splat(float a) { return f2_make(a, a); }
DG_HD float fmax2(float a, float b) { return __builtin_fmaxf(a, b); }
DG_HD float fmax3(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }

// Squared lower bounds of BOTH items of a pair record for one query: distance to the oriented box,
// r = the record's 30 interleaved floats (PairRec::f).  d = p - centre, then per axis
// excess = |u.d| - (half + es), clamped at 0 and squared: 30 VALU instructions for the two items
// (everything but the abs/clamp is packed two-wide; every instruction reads one SGPR pair).
// *centre2 (optional): squared distances to the two box centres, what the traversals order children by.
DG_HD f2 pair_lb2(const float* r, const FPoint& p, f2* centre2 = nullptr)
{
	const f2 dx = f2_splat(p.x[0]) - f2_make(r[0], r[1]);
	const f2 dy = f2_splat(p.x[1]) - f2_make(r[2], r[3]);
	const f2 dz = f2_splat(p.x[2]) - f2_make(r[4], r[5]);
	if (centre2)
		*centre2 = f2_fma(dx, dx, f2_fma(dy, dy, dz * dz));
	const f2 es = f2_splat(p.es);
	f2 acc = f2_splat(0.0f);
	for (int a = 0; a < 3; ++a)
	{
		const float* q = r + 6 + 6 * a;
		f2 t = f2_make(q[4], q[5]) * dz;
		t = f2_fma(f2_make(q[2], q[3]), dy, t);
		t = f2_fma(f2_make(q[0], q[1]), dx, t);
		const f2 hes = f2_make(r[24 + 2 * a], r[25 + 2 * a]) + es;
		const f2 d = f2_make(fmax2(__builtin_fabsf(t.x) - hes.x, 0.0f), fmax2(__builtin_fabsf(t.y) - hes.y, 0.0f));
		acc = f2_fma(d, d, acc);
	}
	return acc;
}
// float upper bound of the running best d^2 (strictly above it unless it is 0 or inf)
DG_HD float best_as_float(double d2)
{
	const float b = (float)d2;
	return b * 1.0000019073486328125f; // * (1 + 2^-19)
}

// ---- float filter for the point/triangle test --------------------------------------------------------------
// The traversal needs, per lane, only (a) an UPPER bound of its minimum squared distance to prune
// with and (b) the set of triangles that can attain the minimum.  Both follow from an evaluation of
// the point/triangle distance in float whose error is bounded rigorously: the exact double test with
// the reference's operation order (tri_closest) then runs only on each lane's own short list of
// candidates, once, after the traversal -- instead of on every triangle any lane of the wave is
// interested in (dg_kernels_k1.hip: k_sample_fast).
//
// Formulation (chosen for its error analysis, not for minimal arithmetic): the triangle carries an
// orthonormal frame (u along v0->v1, w in the plane towards v2, n the normal).  With d = p - v0:
// x = u.d, y = w.d, h = n.d, and dist^2 = h^2 + r^2 with r the distance of (x, y) to the 2-D
// triangle A = (0,0), B = (l0,0), C: r = 0 inside, otherwise the minimum over the three sides of the
// point/segment distance sqrt(perp_i^2 + excess_i^2) (perp_i: signed distance to the side's line,
// positive inside; excess_i: how far the foot of the perpendicular lies beyond the segment's ends).
//
// Error bound.  eps = 2^-24.  Inputs: p - origin and v0 - origin rounded to float (componentwise
// relative eps); the frame vectors, side directions and lengths rounded to float.  With
// R = |p - origin|_1 + mesh_l1 (>= |p-origin|_inf + |v0-origin|_inf, >= |d|_inf, >= l/2):
//   * each of x, y, h carries an absolute error <= 11 eps R (input rounding 2 sqrt(3) eps R, frame
//     rounding and the three roundings of the dot product 4 eps |d|_2 <= 4 sqrt(3) eps R);
//   * every side's segment distance is 1-Lipschitz in the point and in its end points, its float
//     evaluation adds <= 5 eps l + 3 eps |X|: absolute error of r <= 29 eps R; the clamped parameter
//     need not be accurate (any parameter in [0,1] yields a point of the segment; an error du costs
//     3 l^2 du^2, second order);
//   * a point counts as inside (r = 0) only if min(perp_i) >= E, E >= 23 eps R the error of a perp_i: then
//     it IS inside.  A point that is inside by less gets the distance to its nearest side, which is below
//     2 E: still an upper value, and as a lower value too high by at most 4 E^2 (carried by kappa below).
//     A needle-shaped triangle, where a point beyond the sharp corner is within E of two side lines,
//     therefore gets a valid (if loose) interval.
// Hence |dist_float - dist| <= sqrt(29^2 + 11^2) eps R < E := 2^-19 R, and in squares
//   dist^2 in [q - err(q), q + err(q)],  err(q) >= 2 sqrt(q) E + 5 E^2 + 2^-18 q   (approx_err_terms).
//
// What has to lie in that interval is the DOUBLE value the reference computes (the minimum is taken over those), not
// the true distance, so the reference's own rounding error must fit into the slack of err(q) (5 E^2 = 1.8e-11 R^2, of
// which the float side uses a small part).  Bound, with u = 2^-53, D = |p - v0|, l the longest side, rho = area2 / l^2
// the shape ratio (area2 = |e0 x e1|):
//   * the reference evaluates Q(s, t) = |v0 - p + s e0 + t e1|^2 = a00 s^2 + 2 a01 s t + a11 t^2 + 2 b0 s + 2 b1 t + c
//     at the (s, t) it computed; the dozen roundings of that evaluation (and of a00 .. c) each cost at most u times
//     the sum of the absolute terms, (D + 2 l)^2: a first-order term c1 u (D + l)^2, c1 <= 16;
//   * the computed (s, t) is not the minimiser.  On the sides and at the corners the parameter comes from a
//     one-dimensional, well-conditioned quotient (-b / a); in the interior from the 2x2 system with det = a00 a11 -
//     a01^2 = rho^2 l^4, whose cancellation leaves |ds|, |dt| <= 2 u D l^3 / det = 2 u D / (rho^2 l).  Q is a convex
//     quadratic whose gradient vanishes at an interior minimiser, so the value rises by |ds e0 + dt e1|^2 <=
//     (4 u D / rho^2)^2 only: a second-order term c2 u^2 (D + l)^2 / rho^4, c2 <= 16.  A point that changes region
//     because of (ds, dt) is evaluated at a feasible point within (ds, dt) of the minimiser: the same bound.
// |d2_reference - d2_true| <= (c1 u + c2 u^2 / rho^4) (D + l)^2, D + l <= 3 R.  Measured (tests/perf/filter_campaign.py
// -> profiles/r03_filter_campaign.json, seeded, 80-bit reference values): c1 = 3.8, c2 = 0.54; the error is
// 4.2e-16 (D + l)^2 for every rho >= 1e-4 and climbs as rho^-4 below (1.6e-13 at 1e-5, 1.7e-9 at 1e-6, 2.4e-6 at 1e-7).
// With the proven constants the slack is used up at rho = 1.7e-5 (16 u^2 9 R^2 / rho^4 = 1.8e-11 R^2).  Triangles with
// area2 below 1e-4 of the longest side squared -- a factor 6 in rho, 1300 in the error term, above that point --, with
// a zero-length side or with non-finite data are therefore DEGENERATE for the filter (valid = 2): a wave that meets
// one gives all its lanes the exact traversal.  At rho >= 1e-4 the bound is 16 u (1 + 1e-16) 9 R^2 = 1.6e-14 R^2,
// 1100 times below the slack (measured: 4.0e-17 R^2).  The first version of the threshold, 1e-7, was caught by
// tests/test_emu.py::test_float_filter_interval_contains_the_double_value with 36 violations.  The campaign checks
// the interval on 33.5 M seeded pairs -- random and adversarial, half of them slivers within two decades of the
// threshold with points 1e-6 .. 1 side lengths off the plane and beyond the sharp corners --: no violation, largest
// |q - d2| / err = 0.14; tests/test_gpu_edge_cases.py::test_sliver_band_around_the_filter_threshold runs the band
// through both kernels on the GPU.
struct alignas(64) TriApproxPair
{
	// [k][side]: 0..2 v0 - origin; 3..5 u; 6..8 w; 9..11 n; 12 l0; 13 xlo; 14,15 direction of side B->C;
	// 16 l1; 17 xhi; 18,19 direction of side A->C; 20 l2; 21 cy; 22 unused.  In the frame (u, w, n) the triangle is
	// A = (0,0), B = (l0,0), C = (cx,cy): [xlo, xhi] x [0, cy] is its bounding rectangle (xlo = min(0,cx), xhi = max(l0,cx))
	float f[23][2];
	int32_t valid[2]; // 1: triangle; 0: padding slot of an odd leaf (never tested); 2: degenerate triangle (floats all 0):
	                  // a wave that meets one hands its brick to the exact kernel
};
static_assert(sizeof(TriApproxPair) == 192, "TriApproxPair must be 192 bytes");
static const int kApproxFloats = 46;

inline float round_down_f(double v)
{
	float f = (float)v;
	if ((double)f > v)
		f = __builtin_nextafterf(f, -__builtin_inff());
	return f;
}
inline float round_up_f(double v)
{
	float f = (float)v;
	if ((double)f < v)
		f = __builtin_nextafterf(f, __builtin_inff());
	return f;
}
// host: fill one side of a record from the triangle's vertices (double, absolute coordinates)
inline void make_tri_approx(const double v0[3], const double v1[3], const double v2[3], const double origin[3],
							TriApproxPair& rec, int side)
{
	float* f = &rec.f[0][0];
	auto put = [&](int k, double v) { f[2 * k + side] = (float)v; };
	for (int k = 0; k < 23; ++k)
		put(k, 0.0);
	rec.valid[side] = 1;
	double e0[3], e1[3], e2[3];
	for (int d = 0; d < 3; ++d)
	{
		e0[d] = v1[d] - v0[d];
		e1[d] = v2[d] - v0[d];
		e2[d] = v2[d] - v1[d];
	}
	const double n[3] = {e0[1] * e1[2] - e0[2] * e1[1], e0[2] * e1[0] - e0[0] * e1[2], e0[0] * e1[1] - e0[1] * e1[0]};
	const double l0 = __builtin_sqrt(e0[0] * e0[0] + e0[1] * e0[1] + e0[2] * e0[2]);
	const double l2 = __builtin_sqrt(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2]);
	const double l1 = __builtin_sqrt(e2[0] * e2[0] + e2[1] * e2[1] + e2[2] * e2[2]);
	const double area2 = __builtin_sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
	const double lmax = l0 > l1 ? (l0 > l2 ? l0 : l2) : (l1 > l2 ? l1 : l2);
	const double lmin = l0 < l1 ? (l0 < l2 ? l0 : l2) : (l1 < l2 ? l1 : l2);
	bool ok = lmin > 1.0e-15 && lmax < 1.0e15 && area2 > 1.0e-4 * lmax * lmax; // (false for NaN)
	for (int d = 0; d < 3; ++d)
		ok = ok && __builtin_fabs(v0[d] - origin[d]) < 1.0e15;
	if (!ok)
	{
		rec.valid[side] = 2; // degenerate: the float filter says nothing about it
		return;
	}
	double u[3], w[3], nn[3];
	for (int d = 0; d < 3; ++d)
	{
		u[d] = e0[d] / l0;
		nn[d] = n[d] / area2;
	}
	w[0] = nn[1] * u[2] - nn[2] * u[1];
	w[1] = nn[2] * u[0] - nn[0] * u[2];
	w[2] = nn[0] * u[1] - nn[1] * u[0];
	const double cx = e1[0] * u[0] + e1[1] * u[1] + e1[2] * u[2];
	const double cy = e1[0] * w[0] + e1[1] * w[1] + e1[2] * w[2]; // > 0
	for (int d = 0; d < 3; ++d)
	{
		put(d, v0[d] - origin[d]);
		put(3 + d, u[d]);
		put(6 + d, w[d]);
		put(9 + d, nn[d]);
	}
	put(12, l0);
	f[2 * 13 + side] = round_down_f(cx < 0.0 ? cx : 0.0); // rectangle rounded outward
	put(14, (cx - l0) / l1);
	put(15, cy / l1);
	put(16, l1);
	f[2 * 17 + side] = round_up_f(cx > l0 ? cx : l0);
	put(18, cx / l2);
	put(19, cy / l2);
	put(20, l2);
	f[2 * 21 + side] = round_up_f(cy);
}

// per-lane constants of the filter
struct ApproxLane
{
	float x[3]; // p - origin, rounded
	float E;    // 2^-19 (|p - origin|_1 + mesh_l1): bound of |dist_float - dist|; +inf => the lane cannot use the filter
};
DG_HD ApproxLane make_approx_lane(double rx, double ry, double rz, float mesh_l1)
{
	ApproxLane a;
	a.x[0] = (float)rx;
	a.x[1] = (float)ry;
	a.x[2] = (float)rz;
	const float s = (__builtin_fabsf(a.x[0]) + __builtin_fabsf(a.x[1])) + __builtin_fabsf(a.x[2]) + mesh_l1;
	// 2^-19 s, rounded up a little; coordinates beyond 1e15 (or NaN) leave the range in which the
	// squares stay finite floats
	a.E = (s < 1.0e15f) ? s * 1.9092579e-06f + 1.0e-30f : __builtin_inff(); // 2^-19 x 1.001
	return a;
}
// err(q) = q theta + kappa: the float filter's value q of a triangle brackets dist^2 as
// [q - err(q), q + err(q)].  From err(q) >= 2 sqrt(q) E + 5 E^2 + 2^-18 q (5 E^2 instead of E^2: 4 E^2 for
// points within the margin of a side, see tri_approx_pair) and 2 sqrt(q) <= q a + b for ANY a, b > 0 with
// a b >= 1:  theta = E a + 2^-18,  kappa = E b + 5 E^2.  The bound is tight where q = b / a, so b should be
// near the lane's distance: b = sqrt(d0sq) and a = 1 / b to a few percent, from integer arithmetic on the
// float bits (halved exponent; magic-constant reciprocal, whose product with b lies in [0.9494, 1.0506] for
// every float b -- checked exhaustively over a binade -- hence a b >= 1.006 after the factor 1.06).
// The same two numbers turn an upper bound U of dist^2 into the threshold the BOUND tests of the traversal
// may prune with, U (1 + theta) + kappa >= (sqrt(U) + sqrt(3) es)^2 (1 + 2^-19): es <= E / 2 is the error of
// one slab projection (FPoint::es), which the fast bound test leaves out of its three slabs, and 2^-19
// covers the float rounding of the bound's own arithmetic.
// d0sq: any float (a negative, infinite or NaN value only makes the terms loose, b >= E always).
DG_HD void approx_err_terms(float E, float d0sq, float* theta, float* kappa)
{
	union { float f; int32_t i; } u;
	u.f = d0sq;
	u.i = (u.i >> 1) + 0x1fbd1df5; // ~ sqrt(d0sq) within 4.5 %
	const float b = __builtin_fmaxf(u.f, E); // (NaN -> E)
	u.f = b;
	u.i = 0x7ef311c7 - u.i; // ~ 1 / b within 5.1 %
	const float a = u.f * 1.06f;
	*theta = __builtin_fmaf(E, a, 3.814697265625e-06f);
	*kappa = __builtin_fmaf(E, __builtin_fmaf(5.0f, E, b), 1.0e-30f);
}
DG_HD float fmin2(float a, float b) { return __builtin_fminf(a, b); }
// min of two non-NaN floats in ONE instruction (fminf() costs two more to quiet signalling NaNs)
#if defined(__HIP_DEVICE_COMPILE__)
DG_HD float fmin_sel(float a, float b)
{
	float r;
	__asm__("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
	return r;
}
#else
DG_HD float fmin_sel(float a, float b) { return b < a ? b : a; }
#endif
DG_HD float fmin3(float a, float b, float c) { return __builtin_fminf(__builtin_fminf(a, b), c); }
#if defined(__HIP_DEVICE_COMPILE__)
DG_HD float sat01(float a) { return __builtin_amdgcn_fmed3f(a, 0.0f, 1.0f); }
DG_HD float fmed3(float a, float lo, float hi) { return __builtin_amdgcn_fmed3f(a, lo, hi); } // lo <= hi, no NaN
#else
DG_HD float sat01(float a) { return __builtin_fminf(__builtin_fmaxf(a, 0.0f), 1.0f); }
DG_HD float fmed3(float a, float lo, float hi) { return __builtin_fmaxf(__builtin_fminf(a, hi), lo); }
#endif
DG_HD f2 f2_sat01(f2 a) { return f2_make(sat01(a.x), sat01(a.y)); }
DG_HD f2 f2_neg(f2 a) { return f2_make(-a.x, -a.y); }
// The float filter of both triangles of a record for one lane, in two steps.
// Step 1 (tri_approx_frame): the point in the triangles' frames and a cheap LOWER value qlb of dist^2 -- plane distance
// plus the excess over the triangle's bounding rectangle in its own frame.  If qlb - err(qlb) is above every lane's
// upper bound for both triangles, the wave skips step 2 (45 % of the pairs of visited leaves on the judged workload).
// Step 2 (tri_approx_rest): the value q of dist^2 (BEFORE the error terms; the caller forms q -+ err(q)).  A point
// counts as inside (r = 0) only if it is inside by the margin p.E >= 23 eps R; a point that is inside by less gets the
// distance to the nearest side, which is then below 2 E: an upper value all the same, and as a lower value too high
// by at most 4 E^2 (in kappa).  The same error bound covers qlb: the rectangle contains the triangle, its float
// corners are rounded outward, and x, y, h carry the errors analysed above.
struct TriFrame
{
	f2 x, y, h;
};
DG_HD f2 tri_approx_frame(const float* r, const ApproxLane& p, TriFrame* fr)
{
#define DG_R(k) f2_make(r[2 * (k)], r[2 * (k) + 1])
	const f2 dx = f2_splat(p.x[0]) - DG_R(0);
	const f2 dy = f2_splat(p.x[1]) - DG_R(1);
	const f2 dz = f2_splat(p.x[2]) - DG_R(2);
	const f2 x = f2_fma(DG_R(3), dx, f2_fma(DG_R(4), dy, DG_R(5) * dz));
	const f2 y = f2_fma(DG_R(6), dx, f2_fma(DG_R(7), dy, DG_R(8) * dz));
	const f2 h = f2_fma(DG_R(9), dx, f2_fma(DG_R(10), dy, DG_R(11) * dz));
	fr->x = x;
	fr->y = y;
	fr->h = h;
	const f2 bx = x - f2_make(fmed3(x.x, r[26], r[34]), fmed3(x.y, r[27], r[35]));
	const f2 by = y - f2_make(fmed3(y.x, 0.0f, r[42]), fmed3(y.y, 0.0f, r[43]));
	return f2_fma(h, h, f2_fma(bx, bx, by * by));
}
DG_HD f2 tri_approx_rest(const float* r, const ApproxLane& p, const TriFrame& fr)
{
	const f2 x = fr.x, y = fr.y, h = fr.h;
	// side A->B: along = x, perp = y
	const f2 l0 = DG_R(12);
	const f2 ex0 = x - f2_make(fmed3(x.x, 0.0f, l0.x), fmed3(x.y, 0.0f, l0.y));
	const f2 t0 = f2_fma(ex0, ex0, y * y);
	// side B->C: relative to B; interior on the left
	const f2 xm = x - l0;
	const f2 t1x = DG_R(14), t1y = DG_R(15), l1 = DG_R(16);
	const f2 al1 = f2_fma(t1x, xm, t1y * y);
	const f2 pp1 = f2_fma(t1x, y, f2_neg(t1y * xm));
	const f2 ex1 = al1 - f2_make(fmed3(al1.x, 0.0f, l1.x), fmed3(al1.y, 0.0f, l1.y));
	const f2 t1 = f2_fma(ex1, ex1, pp1 * pp1);
	// side A->C: relative to A; interior on the right
	const f2 t2x = DG_R(18), t2y = DG_R(19), l2 = DG_R(20);
	const f2 al2 = f2_fma(t2x, x, t2y * y);
	const f2 pp2 = f2_fma(t2y, x, f2_neg(t2x * y));
	const f2 ex2 = al2 - f2_make(fmed3(al2.x, 0.0f, l2.x), fmed3(al2.y, 0.0f, l2.y));
	const f2 t2 = f2_fma(ex2, ex2, pp2 * pp2);
#undef DG_R
	const float m0 = fmin3(y.x, pp1.x, pp2.x), m1 = fmin3(y.y, pp1.y, pp2.y);
	const float tm0 = fmin3(t0.x, t1.x, t2.x), tm1 = fmin3(t0.y, t1.y, t2.y);
	return f2_fma(h, h, f2_make(m0 >= p.E ? 0.0f : tm0, m1 >= p.E ? 0.0f : tm1));
}
DG_HD f2 tri_approx_pair(const float* r, const ApproxLane& p) // both steps (interval checks, emulator)
{
	TriFrame fr;
	(void)tri_approx_frame(r, p, &fr);
	return tri_approx_rest(r, p, fr);
}

// The bound test of the filtered traversal: pair_lb2 without the per-slab error term (the caller's
// threshold carries it, approx_err_terms) and with the slab excess as t - median(t, -half, half).
// Node pairs only (no empty sides: an inner node has two children).
// *centre2: squared distances of the point to the two box centres -- what the traversal orders the children by when
// it needs both (the lower bounds of two large, curved patches are both 0 or nearly equal for most points; the
// centre distance sends the wave towards the right part of the surface first, so that the upper bounds are tight
// before the other subtrees are looked at: 14.1 -> 11.5 leaf visits per brick on the judged workload).
DG_HD f2 pair_lb2_fast(const float* r, const float* x, f2* centre2)
{
	const f2 dx = f2_splat(x[0]) - f2_make(r[0], r[1]);
	const f2 dy = f2_splat(x[1]) - f2_make(r[2], r[3]);
	const f2 dz = f2_splat(x[2]) - f2_make(r[4], r[5]);
	*centre2 = f2_fma(dx, dx, f2_fma(dy, dy, dz * dz));
	f2 acc = f2_splat(0.0f);
	for (int a = 0; a < 3; ++a)
	{
		const float* q = r + 6 + 6 * a;
		f2 t = f2_make(q[4], q[5]) * dz;
		t = f2_fma(f2_make(q[2], q[3]), dy, t);
		t = f2_fma(f2_make(q[0], q[1]), dx, t);
		const float h0 = r[24 + 2 * a], h1 = r[25 + 2 * a];
		const f2 e = t - f2_make(fmed3(t.x, -h0, h0), fmed3(t.y, -h1, h1));
		acc = f2_fma(e, e, acc);
	}
	return acc;
}

// ---- per-lane query state and epilogue ------------------------------------------------------------------
struct LaneQuery
{
	double px, py, pz; // query point
	FPoint fp;         // float interval of (p - origin)
	double best_d2;
	float bestf;       // float upper bound of best_d2; < 0 => lane inactive (never hits a box)
	int best_tri;      // packet index of the best triangle so far
};
DG_HD void init_query(const double origin[3], float mesh_l1, bool active, double px, double py, double pz,
					  LaneQuery& q)
{
	q.px = px;
	q.py = py;
	q.pz = pz;
	q.fp = make_fpoint(px - origin[0], py - origin[1], pz - origin[2], mesh_l1);
	q.best_d2 = active ? 1.7976931348623157e308 : 0.0;
	q.bestf = active ? __builtin_inff() : -1.0f;
	q.best_tri = -1;
}
// strict '<': of several exactly tied triangles the first one visited wins
DG_HD void offer(LaneQuery& q, double d2, int packet_index)
{
	if (d2 < q.best_d2)
	{
		q.best_d2 = d2;
		q.best_tri = packet_index;
		q.bestf = fmin2(q.bestf, best_as_float(d2)); // (min: bestf may have been seeded with a tighter upper bound)
	}
}

struct LaneResult
{
	double signed_dist;
	double nearest[3];
	int tri_id;
	int entity;
};
// Once per query: re-run the test on the winning triangle to recover (s, t, entity), then the
// sign from the angle-weighted pseudonormal (TriangleMeshDistance.h:274-305, :818).
// `sqrt_fn` is the correctly rounded double sqrt of the platform.
template <class Sqrt>
DG_HD LaneResult finish_query(const TriPacket* tris, const double* pn_all, const LaneQuery& q, Sqrt sqrt_fn)
{
	LaneResult r;
	const TriPacket T = tris[q.best_tri];
	const Hit h = tri_closest<true>(T, q.px, q.py, q.pz);
	const double nx = T.v0[0] + h.s * T.e0[0] + h.t * T.e1[0];
	const double ny = T.v0[1] + h.s * T.e0[1] + h.t * T.e1[1];
	const double nz = T.v0[2] + h.s * T.e0[2] + h.t * T.e1[2];
	const double* pn = pn_all + ((size_t)q.best_tri * kPnSlots + (size_t)h.entity) * 3;
	const double ux = q.px - nx, uy = q.py - ny, uz = q.pz - nz;
	const double dotp = ux * pn[0] + uy * pn[1] + uz * pn[2];
	const double dist = sqrt_fn(h.d2);
	r.signed_dist = dist * ((dotp >= 0.0) ? 1.0 : -1.0);
	r.nearest[0] = nx;
	r.nearest[1] = ny;
	r.nearest[2] = nz;
	r.tri_id = T.tri_id;
	r.entity = h.entity;
	return r;
}

} // namespace dg
