// dg_geom.h -- per-lane arithmetic shared by the HIP kernels (device) and the host-side
// structure builders.  Everything here is written for bit parity with the reference's CPU
// path: every floating-point expression keeps the reference's association order and must be
// compiled with -ffp-contract=off (an FMA changes d^2 by up to 8e-9 relative near the zero
// level set).  No function in this header touches memory other than its
// arguments.
//
// Reference lines restated (paths relative to the Discregrid tree):
//   point/triangle distance  discregrid/include/Discregrid/geometry/TriangleMeshDistance.h:564-820
//   node positions           discregrid/src/cubic_lagrange_discrete_grid.cpp:604-665
//   shape functions          discregrid/src/cubic_lagrange_discrete_grid.cpp:339-580
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
// ever used to prune, so their arithmetic is free.  DG_OBB=0 selects the first design, kept for
// A/B measurements: an axis-aligned box AND one slab along the mean normal, which is tight on the
// concave side of a curved surface but cannot separate neighbouring facets seen from far away
// (their boxes are fat along a tilted normal): 45 ms against 22 ms per 256^3 launch.
#ifndef DG_OBB
#define DG_OBB 1
#endif
#if DG_OBB
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
#else
struct alignas(128) PairRec
{
	float f[11][2];  // [k][side], k: 0..2 box lo xyz, 3..5 box hi xyz, 6..8 slab direction, 9 slab lo, 10 slab hi
	int32_t info[2]; // node pairs: what is below each side (see below); triangle pairs: unused
	float pad_[8];
};
static const int kPairFloats = 22;
#endif
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

#if DG_OBB
// Squared lower bounds of BOTH items of a pair record for one query: distance to the oriented box,
// r = the record's 30 interleaved floats (PairRec::f).  d = p - centre, then per axis
// excess = |u.d| - (half + es), clamped at 0 and squared: 30 VALU instructions for the two items
// (everything but the abs/clamp is packed two-wide; every instruction reads one SGPR pair).
DG_HD f2 pair_lb2(const float* r, const FPoint& p)
{
	const f2 dx = f2_splat(p.x[0]) - f2_make(r[0], r[1]);
	const f2 dy = f2_splat(p.x[1]) - f2_make(r[2], r[3]);
	const f2 dz = f2_splat(p.x[2]) - f2_make(r[4], r[5]);
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
#else
// Squared lower bounds (box and slab combined) of BOTH items of a pair record for one query.
// r = the record's 22 interleaved floats (PairRec::f).
DG_HD f2 pair_lb2(const float* r, const FPoint& p)
{
	f2 acc = f2_splat(0.0f);
	for (int d = 0; d < 3; ++d)
	{
		const f2 lo = f2_make(r[2 * d], r[2 * d + 1]);
		const f2 hi = f2_make(r[6 + 2 * d], r[6 + 2 * d + 1]);
		const f2 a = lo - f2_splat(p.hi[d]);
		const f2 b = f2_splat(p.lo[d]) - hi;
		const f2 m = f2_make(fmax3(a.x, b.x, 0.0f), fmax3(a.y, b.y, 0.0f));
		acc = f2_fma(m, m, acc);
	}
	f2 t = f2_make(r[16], r[17]) * f2_splat(p.x[2]);
	t = f2_fma(f2_make(r[14], r[15]), f2_splat(p.x[1]), t);
	t = f2_fma(f2_make(r[12], r[13]), f2_splat(p.x[0]), t);
	const f2 es = f2_splat(p.es);
	const f2 q1 = t - es - f2_make(r[20], r[21]);
	const f2 q2 = f2_make(r[18], r[19]) - t - es;
	const f2 ds = f2_make(fmax3(q1.x, q2.x, 0.0f), fmax3(q1.y, q2.y, 0.0f));
	const f2 s2 = ds * ds;
	return f2_make(fmax2(acc.x, s2.x), fmax2(acc.y, s2.y));
}
#endif
// float upper bound of the running best d^2 (strictly above it unless it is 0 or inf)
DG_HD float best_as_float(double d2)
{
	const float b = (float)d2;
	return b * 1.0000019073486328125f; // * (1 + 2^-19)
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
		q.bestf = best_as_float(d2);
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

// ---- lattice node positions ---------------------------------------------------------------------------
// Node class c in {0:V, 1:X, 2:Y, 3:Z}; (a, b, s) are the class' own (fastest, middle, slowest)
// lattice coordinates, i.e. class-local flat index = (s*D1 + b)*D0 + a with
//   V: (i, j, k)         D = (nx+1, ny+1, nz+1)
//   X: (2i+h, j, k)      D = (2nx,  ny+1, nz+1)      h = 0: node at 1/3, h = 1: at 2/3 of the edge
//   Y: (2j+h, k, i)      D = (2ny,  nz+1, nx+1)
//   Z: (2k+h, i, j)      D = (2nz,  nx+1, ny+1)
// which is exactly the order indexToNodePosition() enumerates (cubic_lagrange_discrete_grid.cpp:618-662).
DG_HD void class_dims(int c, const uint32_t res[3], uint32_t D[3])
{
	const uint32_t nx = res[0], ny = res[1], nz = res[2];
	if (c == 0) { D[0] = nx + 1; D[1] = ny + 1; D[2] = nz + 1; }
	else if (c == 1) { D[0] = 2 * nx; D[1] = ny + 1; D[2] = nz + 1; }
	else if (c == 2) { D[0] = 2 * ny; D[1] = nz + 1; D[2] = nx + 1; }
	else { D[0] = 2 * nz; D[1] = nx + 1; D[2] = ny + 1; }
}
DG_HD void node_position(int c, uint32_t a, uint32_t b, uint32_t s, const double dmin[3], const double cell[3],
						 double x[3])
{
	uint32_t i, j, k, h = a & 1u;
	if (c == 0) { i = a; j = b; k = s; }
	else if (c == 1) { i = a >> 1; j = b; k = s; }
	else if (c == 2) { j = a >> 1; k = b; i = s; }
	else { k = a >> 1; i = b; j = s; }
	x[0] = dmin[0] + cell[0] * (double)i;
	x[1] = dmin[1] + cell[1] * (double)j;
	x[2] = dmin[2] + cell[2] * (double)k;
	if (c > 0)
		x[c - 1] += (1.0 + (double)h) / 3.0 * cell[c - 1];
}

// ---- 32 serendipity-cubic shape functions (+ derivatives) -------------------------------------------------
// N[j], j = 0..31 in the reference's node order; dN (if GRAD) as dNx[32], dNy[32], dNz[32].
// Same products in the same order as shape_function_() (cubic_lagrange_discrete_grid.cpp:339-580);
// everything is fully unrolled so the arrays live in registers.
template <bool GRAD>
DG_HD void shape_functions(double x, double y, double z, double N[32], double dNx[32], double dNy[32], double dNz[32])
{
	const double x2 = x * x, y2 = y * y, z2 = z * z;
	const double mx = 1.0 - x, my = 1.0 - y, mz = 1.0 - z;
	const double px = 1.0 + x, py = 1.0 + y, pz = 1.0 + z;
	const double m3x = 1.0 - 3.0 * x, m3y = 1.0 - 3.0 * y, m3z = 1.0 - 3.0 * z;
	const double p3x = 1.0 + 3.0 * x, p3y = 1.0 + 3.0 * y, p3z = 1.0 + 3.0 * z;
	const double mxmy = mx * my, mxpy = mx * py, pxmy = px * my, pxpy = px * py;
	const double mxmz = mx * mz, mxpz = mx * pz, pxmz = px * mz, pxpz = px * pz;
	const double mymz = my * mz, mypz = my * pz, pymz = py * mz, pypz = py * pz;
	const double omx2 = 1.0 - x2, omy2 = 1.0 - y2, omz2 = 1.0 - z2;

	double fac = 1.0 / 64.0 * (9.0 * (x2 + y2 + z2) - 19.0);
	N[0] = fac * mxmy * mz;
	N[1] = fac * pxmy * mz;
	N[2] = fac * mxpy * mz;
	N[3] = fac * pxpy * mz;
	N[4] = fac * mxmy * pz;
	N[5] = fac * pxmy * pz;
	N[6] = fac * mxpy * pz;
	N[7] = fac * pxpy * pz;

	fac = 9.0 / 64.0 * omx2;
	const double fm3x = fac * m3x, fp3x = fac * p3x;
	N[8] = fm3x * mymz;
	N[9] = fp3x * mymz;
	N[10] = fm3x * mypz;
	N[11] = fp3x * mypz;
	N[12] = fm3x * pymz;
	N[13] = fp3x * pymz;
	N[14] = fm3x * pypz;
	N[15] = fp3x * pypz;

	fac = 9.0 / 64.0 * omy2;
	const double fm3y = fac * m3y, fp3y = fac * p3y;
	N[16] = fm3y * mxmz;
	N[17] = fp3y * mxmz;
	N[18] = fm3y * pxmz;
	N[19] = fp3y * pxmz;
	N[20] = fm3y * mxpz;
	N[21] = fp3y * mxpz;
	N[22] = fm3y * pxpz;
	N[23] = fp3y * pxpz;

	fac = 9.0 / 64.0 * omz2;
	const double fm3z = fac * m3z, fp3z = fac * p3z;
	N[24] = fm3z * mxmy;
	N[25] = fp3z * mxmy;
	N[26] = fm3z * mxpy;
	N[27] = fp3z * mxpy;
	N[28] = fm3z * pxmy;
	N[29] = fp3z * pxmy;
	N[30] = fm3z * pxpy;
	N[31] = fp3z * pxpy;

	if (!GRAD)
		return;

	const double gx = 9.0 * (3.0 * x2 + y2 + z2) - 19.0;
	const double gy = 9.0 * (x2 + 3.0 * y2 + z2) - 19.0;
	const double gz = 9.0 * (x2 + y2 + 3.0 * z2) - 19.0;
	const double x18 = 18.0 * x, y18 = 18.0 * y, z18 = 18.0 * z;
	const double hxm = x18 - gx, hxp = x18 + gx;
	const double hym = y18 - gy, hyp = y18 + gy;
	const double hzm = z18 - gz, hzp = z18 + gz;
	// corners: value / 64 (topRows(8) /= 64)
	dNx[0] = hxm * mymz / 64.0; dNy[0] = mxmz * hym / 64.0; dNz[0] = mxmy * hzm / 64.0;
	dNx[1] = hxp * mymz / 64.0; dNy[1] = pxmz * hym / 64.0; dNz[1] = pxmy * hzm / 64.0;
	dNx[2] = hxm * pymz / 64.0; dNy[2] = mxmz * hyp / 64.0; dNz[2] = mxpy * hzm / 64.0;
	dNx[3] = hxp * pymz / 64.0; dNy[3] = pxmz * hyp / 64.0; dNz[3] = pxpy * hzm / 64.0;
	dNx[4] = hxm * mypz / 64.0; dNy[4] = mxpz * hym / 64.0; dNz[4] = mxmy * hzp / 64.0;
	dNx[5] = hxp * mypz / 64.0; dNy[5] = pxpz * hym / 64.0; dNz[5] = pxmy * hzp / 64.0;
	dNx[6] = hxm * pypz / 64.0; dNy[6] = mxpz * hyp / 64.0; dNz[6] = mxpy * hzp / 64.0;
	dNx[7] = hxp * pypz / 64.0; dNy[7] = pxpz * hyp / 64.0; dNz[7] = pxpy * hzp / 64.0;

	const double k = 9.0 / 64.0; // bottomRows(24) *= 9/64
	const double t3x = 3.0 - 9.0 * x2, t3y = 3.0 - 9.0 * y2, t3z = 3.0 - 9.0 * z2;
	const double tx = 2.0 * x, ty = 2.0 * y, tz = 2.0 * z;
	const double qxm = -t3x - tx, qxp = t3x - tx;
	const double qym = -t3y - ty, qyp = t3y - ty;
	const double qzm = -t3z - tz, qzp = t3z - tz;
	const double wxm = omx2 * m3x, wxp = omx2 * p3x;
	const double wym = omy2 * m3y, wyp = omy2 * p3y;
	const double wzm = omz2 * m3z, wzp = omz2 * p3z;
	// x-edges
	dNx[8] = qxm * mymz * k;  dNy[8] = -wxm * mz * k;  dNz[8] = -wxm * my * k;
	dNx[9] = qxp * mymz * k;  dNy[9] = -wxp * mz * k;  dNz[9] = -wxp * my * k;
	dNx[10] = qxm * mypz * k; dNy[10] = -wxm * pz * k; dNz[10] = wxm * my * k;
	dNx[11] = qxp * mypz * k; dNy[11] = -wxp * pz * k; dNz[11] = wxp * my * k;
	dNx[12] = qxm * pymz * k; dNy[12] = wxm * mz * k;  dNz[12] = -wxm * py * k;
	dNx[13] = qxp * pymz * k; dNy[13] = wxp * mz * k;  dNz[13] = -wxp * py * k;
	dNx[14] = qxm * pypz * k; dNy[14] = wxm * pz * k;  dNz[14] = wxm * py * k;
	dNx[15] = qxp * pypz * k; dNy[15] = wxp * pz * k;  dNz[15] = wxp * py * k;
	// y-edges
	dNx[16] = -wym * mz * k; dNy[16] = qym * mxmz * k; dNz[16] = -wym * mx * k;
	dNx[17] = -wyp * mz * k; dNy[17] = qyp * mxmz * k; dNz[17] = -wyp * mx * k;
	dNx[18] = wym * mz * k;  dNy[18] = qym * pxmz * k; dNz[18] = -wym * px * k;
	dNx[19] = wyp * mz * k;  dNy[19] = qyp * pxmz * k; dNz[19] = -wyp * px * k;
	dNx[20] = -wym * pz * k; dNy[20] = qym * mxpz * k; dNz[20] = wym * mx * k;
	dNx[21] = -wyp * pz * k; dNy[21] = qyp * mxpz * k; dNz[21] = wyp * mx * k;
	dNx[22] = wym * pz * k;  dNy[22] = qym * pxpz * k; dNz[22] = wym * px * k;
	dNx[23] = wyp * pz * k;  dNy[23] = qyp * pxpz * k; dNz[23] = wyp * px * k;
	// z-edges
	dNx[24] = -wzm * my * k; dNy[24] = -wzm * mx * k; dNz[24] = qzm * mxmy * k;
	dNx[25] = -wzp * my * k; dNy[25] = -wzp * mx * k; dNz[25] = qzp * mxmy * k;
	dNx[26] = -wzm * py * k; dNy[26] = wzm * mx * k;  dNz[26] = qzm * mxpy * k;
	dNx[27] = -wzp * py * k; dNy[27] = wzp * mx * k;  dNz[27] = qzp * mxpy * k;
	dNx[28] = wzm * my * k;  dNy[28] = -wzm * px * k; dNz[28] = qzm * pxmy * k;
	dNx[29] = wzp * my * k;  dNy[29] = -wzp * px * k; dNz[29] = qzp * pxmy * k;
	dNx[30] = wzm * py * k;  dNy[30] = wzm * px * k;  dNz[30] = qzm * pxpy * k;
	dNx[31] = wzp * py * k;  dNy[31] = wzp * px * k;  dNz[31] = qzp * pxpy * k;
}

// 32 node indices of grid cell (i,j,k) for an unreduced field -- the rows the reference's
// serial loop materialises (cubic_lagrange_discrete_grid.cpp:836-886).  Entries come in
// adjacent pairs (2m, 2m+1) for m >= 4, and corner pairs (0,1),(2,3),(4,5),(6,7) are adjacent
// too: the evaluator fetches 16 x 16-byte segments.
DG_HD void cell_node_indices(uint32_t i, uint32_t j, uint32_t k, const uint32_t res[3], uint32_t out[32])
{
	const uint32_t nx = res[0], ny = res[1], nz = res[2];
	const uint32_t nv = (nx + 1) * (ny + 1) * (nz + 1);
	const uint32_t nex = nx * (ny + 1) * (nz + 1);
	const uint32_t ney = (nx + 1) * ny * (nz + 1);
	const uint32_t r0 = (nx + 1) * (ny + 1) * k + (nx + 1) * j + i;
	out[0] = r0;
	out[1] = r0 + 1;
	out[2] = r0 + (nx + 1);
	out[3] = r0 + (nx + 1) + 1;
	const uint32_t r1 = r0 + (nx + 1) * (ny + 1);
	out[4] = r1;
	out[5] = r1 + 1;
	out[6] = r1 + (nx + 1);
	out[7] = r1 + (nx + 1) + 1;
	uint32_t off = nv;
	out[8] = off + 2 * (nx * (ny + 1) * k + nx * j + i);
	out[10] = off + 2 * (nx * (ny + 1) * (k + 1) + nx * j + i);
	out[12] = off + 2 * (nx * (ny + 1) * k + nx * (j + 1) + i);
	out[14] = off + 2 * (nx * (ny + 1) * (k + 1) + nx * (j + 1) + i);
	off += 2 * nex;
	out[16] = off + 2 * (ny * (nz + 1) * i + ny * k + j);
	out[18] = off + 2 * (ny * (nz + 1) * (i + 1) + ny * k + j);
	out[20] = off + 2 * (ny * (nz + 1) * i + ny * (k + 1) + j);
	out[22] = off + 2 * (ny * (nz + 1) * (i + 1) + ny * (k + 1) + j);
	off += 2 * ney;
	out[24] = off + 2 * (nz * (nx + 1) * j + nz * i + k);
	out[26] = off + 2 * (nz * (nx + 1) * (j + 1) + nz * i + k);
	out[28] = off + 2 * (nz * (nx + 1) * j + nz * (i + 1) + k);
	out[30] = off + 2 * (nz * (nx + 1) * (j + 1) + nz * (i + 1) + k);
	for (int m = 8; m < 32; m += 2)
		out[m + 1] = out[m] + 1;
}

// ---- K2 per-query body -------------------------------------------------------------------------------
// Host or device arrays, same code (the C++ host API evaluates single points with it).
struct FieldDev
{
	double dmin[3], dmax[3];
	double cell[3], inv_cell[3];
	uint32_t res[3];
	const double* coeffs;
	const uint32_t* cells;    // nullable => closed-form rows
	const uint32_t* cell_map; // nullable => identity
	// Optional cell-major copy of the field: 32 doubles (256 B, 2 cache lines) per cell row, in
	// the row's node order.  Trades 4.6x the memory (288 GB of HBM3E is the point of this chip)
	// for a gather-free evaluator: one query reads 256 contiguous bytes instead of 16 scattered
	// 16-byte segments in 16 different lines.
	const double* cell_major;
};

// Per-query body of K2 = CubicLagrangeDiscreteGrid::interpolate(field, x, gradient*)
// (discregrid/src/cubic_lagrange_discrete_grid.cpp:977-1063).  The 32-term sum runs in j order
// (parity), the 32 coefficients are fetched as 16 adjacent pairs for unreduced fields.  Returns
// DBL_MAX ("no value") outside the domain, in removed cells, or if a coefficient is DBL_MAX;
// the gradient is zero in those cases.
template <bool GRAD>
DG_HD double interpolate_point(const FieldDev& F, const double x[3], double g[3])
{
	const double NOVAL = 1.7976931348623157e308;
	g[0] = g[1] = g[2] = 0.0;
	for (int d = 0; d < 3; ++d)
		if (!((F.dmin[d] <= x[d]) && (x[d] <= F.dmax[d]))) // AlignedBox::contains, inclusive (:981)
			return NOVAL;
	uint32_t mi[3];
	for (int d = 0; d < 3; ++d)
	{
		mi[d] = (uint32_t)((x[d] - F.dmin[d]) * F.inv_cell[d]); // :984
		if (mi[d] >= F.res[d])
			mi[d] = F.res[d] - 1;
	}
	const uint32_t ci = F.res[1] * F.res[0] * mi[2] + F.res[0] * mi[1] + mi[0];
	const uint32_t cm = F.cell_map ? F.cell_map[ci] : ci;
	if (cm == 0xffffffffu)
		return NOVAL;
	double c0[3], xi[3];
	for (int d = 0; d < 3; ++d)
	{
		const double lo = F.dmin[d] + (double)mi[d] * F.cell[d]; // subdomain(), discrete_grid.cpp:26-32
		const double hi = lo + F.cell[d];
		const double den = hi - lo; // :1000
		c0[d] = 2.0 / den;
		const double c1 = (hi + lo) / den;
		xi[d] = c0[d] * x[d] - c1;
	}
	double cf[32];
	if (F.cell_major)
	{
		const double* row = F.cell_major + 32 * (size_t)cm;
#if defined(__HIP__)
#pragma unroll
#endif
		for (int j = 0; j < 32; ++j)
			cf[j] = row[j];
	}
	else if (F.cells)
	{
		const uint32_t* row = F.cells + 32 * (size_t)cm;
#if defined(__HIP__)
#pragma unroll
#endif
		for (int j = 0; j < 32; ++j)
			cf[j] = F.coeffs[row[j]];
	}
	else
	{
		uint32_t idx[32];
		cell_node_indices(mi[0], mi[1], mi[2], F.res, idx);
#if defined(__HIP__)
#pragma unroll
#endif
		for (int m = 0; m < 32; m += 2)
		{
			const double* pr = F.coeffs + idx[m]; // adjacent pair: one 16-byte load
			cf[m] = pr[0];
			cf[m + 1] = pr[1];
		}
	}
	double N[32], dNx[32], dNy[32], dNz[32];
	shape_functions<GRAD>(xi[0], xi[1], xi[2], N, dNx, dNy, dNz);
	bool ok = true;
	double phi = 0.0, gx = 0.0, gy = 0.0, gz = 0.0;
#if defined(__HIP__)
#pragma unroll
#endif
	for (int j = 0; j < 32; ++j)
	{
		ok = ok && (cf[j] != NOVAL);
		phi += cf[j] * N[j];
		if (GRAD)
		{
			gx += cf[j] * dNx[j];
			gy += cf[j] * dNy[j];
			gz += cf[j] * dNz[j];
		}
	}
	if (!ok)
		return NOVAL;
	if (GRAD)
	{
		g[0] = gx * c0[0];
		g[1] = gy * c0[1];
		g[2] = gz * c0[2];
	}
	return phi;
}

// ---- K3: SPH boundary density map (GenerateDensityMap) ------------------------------------------------
// Restates cmd/generate_density_map/main.cpp:86-112 (gamma, density_func), :119-133 (node
// predicate), sph_kernel.hpp:11-42 (CubicKernel::W) and gauss_quadrature.cpp:5927-5960 (the
// 16^3-point tensor Gauss-Legendre rule for p = 30), with the reference's operation order:
// the 4096-term sum runs i, j, k sequentially per node.
struct DensityParams
{
	double h;           // kernel support radius ("ar")
	double rho0;
	double c0prod;      // (0.5*diag).prod() = h*(h*h)
	double cell_diag;   // cellSize().norm(), Eigen association x^2 + (y^2 + z^2)
	int band_predicate; // apply the node predicate of main.cpp:119-133
	double xi[16];      // quadrature offsets  c0*abscissa + c1 = h*a + 0.0
	double w[16];       // weights
	const double* wtab; // 4096 values W(xi_i, xi_j, xi_k), index (i*16 + j)*16 + k
	// Quadrature points outside the kernel's support (|xi| > h: 3088 of the 4096 points) contribute
	// w * (gamma * 0.0) = +0.0 to a sum of non-negative terms, i.e. nothing -- provided gamma is
	// finite, which holds whenever every coefficient other than DBL_MAX is finite and below 1e290.
	// kmask[i*16 + j] has bit k set where W(xi_i, xi_j, xi_k) != 0.
	// skip_mode 0: evaluate every point; 1: skip the zero-weight points; 2 (device): skip them unless
	// *unsafe != 0 (set by k_field_check when the field holds NaN / Inf / huge values).
	uint16_t kmask[256];
	int32_t skip_mode;
	const uint32_t* unsafe;
};

// CubicKernel::setRadius / W (sph_kernel.hpp:11-42); r.norm() as Eigen evaluates it for a 3-vector
DG_HD double cubic_kernel_k(double radius)
{
	const double pi = 3.14159265358979323846; // M_PI
	const double h3 = radius * radius * radius;
	return 8.0 / (pi * h3);
}
template <class Sqrt>
DG_HD double cubic_kernel_W(double rx, double ry, double rz, double radius, double k, Sqrt sqrt_fn)
{
	double res = 0.0;
	const double rl = sqrt_fn(rx * rx + (ry * ry + rz * rz));
	const double q = rl / radius;
	if (q <= 1.0)
	{
		if (q <= 0.5)
		{
			const double q2 = q * q;
			const double q3 = q2 * q;
			res = k * (6.0 * q3 - 6.0 * q2 + 1.0);
		}
		else
		{
			const double omq = 1.0 - q;
			res = k * (2.0 * omq * omq * omq);
		}
	}
	return res;
}

// Stage 1 (cheap): node predicate (main.cpp:119-133) and the early-out of density_func (:98-102).
// Returns true if the node needs the quadrature; otherwise *value is the final field value
// (DBL_MAX for predicate-rejected nodes, 0.0 for nodes farther than 2h from the surface).
DG_HD bool density_prefilter(const FieldDev& F, const DensityParams& P, const double x[3], double* value)
{
	const double NOVAL = 1.7976931348623157e308;
	double g[3];
	if (P.band_predicate)
	{
		double xc[3];
		for (int d = 0; d < 3; ++d) // x.cwiseMax(domain.min()).cwiseMin(domain.max())
		{
			const double a = x[d] < F.dmin[d] ? F.dmin[d] : x[d];
			xc[d] = a < F.dmax[d] ? a : F.dmax[d];
		}
		const double dist = interpolate_point<false>(F, xc, g);
		if (dist == NOVAL || !(-6.0 * P.h < dist + P.cell_diag && dist - P.cell_diag < 2.0 * P.h))
		{
			*value = NOVAL;
			return false;
		}
	}
	const double dist = interpolate_point<false>(F, x, g);
	if (dist > 2.0 * P.h)
	{
		*value = 0.0;
		return false;
	}
	return true;
}

// One coordinate axis of interpolate_point(): everything that depends on a single coordinate of
// the evaluation point.  Same expressions as in interpolate_point()/shape_functions(), so staging
// them per axis changes no bits -- it only avoids recomputing the x- and y-dependent parts (cell
// lookup, affine map with its two divisions, polynomial factors) 16 and 256 times.
struct Axis1D
{
	double t;              // local coordinate in [-1, 1]
	double t2, m, p;       // t^2, 1 - t, 1 + t
	double fm3, fp3;       // 9/64 * (1 - t^2) * (1 -+ 3t)
	uint32_t mi;           // cell index along the axis
	bool inside;
};
DG_HD Axis1D axis_eval(const FieldDev& F, int d, double y)
{
	Axis1D a;
	a.inside = (F.dmin[d] <= y) && (y <= F.dmax[d]);
	uint32_t mi = (uint32_t)((y - F.dmin[d]) * F.inv_cell[d]);
	if (mi >= F.res[d])
		mi = F.res[d] - 1;
	if (!a.inside)
		mi = 0;
	a.mi = mi;
	const double lo = F.dmin[d] + (double)mi * F.cell[d];
	const double hi = lo + F.cell[d];
	const double den = hi - lo;
	const double c0 = 2.0 / den;
	const double c1 = (hi + lo) / den;
	a.t = c0 * y - c1;
	a.t2 = a.t * a.t;
	a.m = 1.0 - a.t;
	a.p = 1.0 + a.t;
	const double fac = 9.0 / 64.0 * (1.0 - a.t2);
	a.fm3 = fac * (1.0 - 3.0 * a.t);
	a.fp3 = fac * (1.0 + 3.0 * a.t);
	return a;
}

// Stage 2: rho0 * integral over [-h,h]^3 of gamma(x + xi) W(xi), 16^3 Gauss points, summed in
// the reference's i, j, k order (gauss_quadrature.cpp:5941-5958).  Unreduced fields take the
// staged path (per-axis work hoisted out of the inner loops); reduced fields go through
// interpolate_point().  Both produce the same bits (tests/test_density_map.py).
template <bool STAGED>
DG_HD double density_integral_t(const FieldDev& F, const DensityParams& P, const double x[3])
{
	const double NOVAL = 1.7976931348623157e308;
	double g[3];
	double res = 0.0;
	const bool staged = STAGED;
	const bool skip = P.skip_mode == 1 || (P.skip_mode == 2 && P.unsafe[0] == 0u);
	DG_NOUNROLL
	for (int i = 0; i < 16; ++i)
	{
		const double wi = P.w[i];
		const double yx = x[0] + P.xi[i];
		const Axis1D ax = axis_eval(F, 0, yx);
		DG_NOUNROLL
		for (int j = 0; j < 16; ++j)
		{
			const uint32_t kmask = skip ? (uint32_t)P.kmask[i * 16 + j] : 0xffffu;
			if (kmask == 0u)
				continue; // the whole column lies outside the kernel's support
			const double wij = wi * P.w[j];
			const double yy = x[1] + P.xi[j];
			const Axis1D ay = axis_eval(F, 1, yy);
			const double mxmy = ax.m * ay.m, mxpy = ax.m * ay.p, pxmy = ax.p * ay.m, pxpy = ax.p * ay.p;
			const double x2y2 = ax.t2 + ay.t2;
			DG_NOUNROLL
			for (int k = 0; k < 16; ++k)
			{
				if (((kmask >> k) & 1u) == 0u)
					continue;
				const double wijk = wij * P.w[k];
				const double yz = x[2] + P.xi[k];
				double d;
				if (staged)
				{
					const Axis1D az = axis_eval(F, 2, yz);
					if (ax.inside && ay.inside && az.inside)
					{
						const uint32_t ci = F.res[1] * F.res[0] * az.mi + F.res[0] * ay.mi + ax.mi;
						double cf[32];
						if (F.cell_major)
						{
							const double* row = F.cell_major + 32 * (size_t)ci;
#if defined(__HIP__)
#pragma unroll
#endif
							for (int q = 0; q < 32; ++q)
								cf[q] = row[q];
						}
						else
						{
							uint32_t idx[32];
							cell_node_indices(ax.mi, ay.mi, az.mi, F.res, idx);
#if defined(__HIP__)
#pragma unroll
#endif
							for (int q = 0; q < 32; q += 2)
							{
								const double* pr = F.coeffs + idx[q]; // adjacent pair: one 16-byte load
								cf[q] = pr[0];
								cf[q + 1] = pr[1];
							}
						}
						const double mz = az.m, pz = az.p;
						const double fac = 1.0 / 64.0 * (9.0 * (x2y2 + az.t2) - 19.0);
						// phi = sum_q cf[q] * N[q] in q order; every N[q] is formed right where it is
						// consumed (same products as shape_functions(), no 32-entry array kept live)
						bool ok = true;
						double phi = 0.0;
#define DG_ACC(q, n)                  \
	ok = ok && (cf[q] != NOVAL); \
	phi += cf[q] * (n);
						DG_ACC(0, fac * mxmy * mz)
						DG_ACC(1, fac * pxmy * mz)
						DG_ACC(2, fac * mxpy * mz)
						DG_ACC(3, fac * pxpy * mz)
						DG_ACC(4, fac * mxmy * pz)
						DG_ACC(5, fac * pxmy * pz)
						DG_ACC(6, fac * mxpy * pz)
						DG_ACC(7, fac * pxpy * pz)
						{
							const double mymz = ay.m * mz, mypz = ay.m * pz, pymz = ay.p * mz, pypz = ay.p * pz;
							DG_ACC(8, ax.fm3 * mymz)
							DG_ACC(9, ax.fp3 * mymz)
							DG_ACC(10, ax.fm3 * mypz)
							DG_ACC(11, ax.fp3 * mypz)
							DG_ACC(12, ax.fm3 * pymz)
							DG_ACC(13, ax.fp3 * pymz)
							DG_ACC(14, ax.fm3 * pypz)
							DG_ACC(15, ax.fp3 * pypz)
						}
						{
							const double mxmz = ax.m * mz, mxpz = ax.m * pz, pxmz = ax.p * mz, pxpz = ax.p * pz;
							DG_ACC(16, ay.fm3 * mxmz)
							DG_ACC(17, ay.fp3 * mxmz)
							DG_ACC(18, ay.fm3 * pxmz)
							DG_ACC(19, ay.fp3 * pxmz)
							DG_ACC(20, ay.fm3 * mxpz)
							DG_ACC(21, ay.fp3 * mxpz)
							DG_ACC(22, ay.fm3 * pxpz)
							DG_ACC(23, ay.fp3 * pxpz)
						}
						DG_ACC(24, az.fm3 * mxmy)
						DG_ACC(25, az.fp3 * mxmy)
						DG_ACC(26, az.fm3 * mxpy)
						DG_ACC(27, az.fp3 * mxpy)
						DG_ACC(28, az.fm3 * pxmy)
						DG_ACC(29, az.fp3 * pxmy)
						DG_ACC(30, az.fm3 * pxpy)
						DG_ACC(31, az.fp3 * pxpy)
#undef DG_ACC
						d = ok ? phi : NOVAL;
					}
					else
						d = NOVAL;
				}
				else
				{
					const double y[3] = {yx, yy, yz};
					d = interpolate_point<false>(F, y, g);
				}
				const double gamma = (d > P.h) ? 0.0 : 1.0 - d / P.h;
				res += wijk * (gamma * P.wtab[(i * 16 + j) * 16 + k]);
			}
		}
	}
	res *= P.c0prod;
	return P.rho0 * res;
}
DG_HD double density_integral(const FieldDev& F, const DensityParams& P, const double x[3])
{
	if ((F.cells == nullptr) && (F.cell_map == nullptr))
		return density_integral_t<true>(F, P, x);
	return density_integral_t<false>(F, P, x);
}

// flat node index -> position (the inverse of the class decomposition; used where nodes are
// addressed individually rather than as bricks)
DG_HD void node_position_flat(uint64_t l, const uint32_t res[3], const double dmin[3], const double cell[3], double x[3])
{
	uint32_t D[3];
	int c = 0;
	uint64_t off = 0;
	for (; c < 4; ++c)
	{
		class_dims(c, res, D);
		const uint64_t size = (uint64_t)D[0] * D[1] * D[2];
		if (l < off + size || c == 3)
			break;
		off += size;
	}
	const uint64_t lc = l - off;
	const uint32_t a = (uint32_t)(lc % D[0]);
	const uint32_t b = (uint32_t)((lc / D[0]) % D[1]);
	const uint32_t s = (uint32_t)(lc / ((uint64_t)D[0] * D[1]));
	node_position(c, a, b, s, dmin, cell, x);
}

} // namespace dg
