// dg_density_cells.h -- K3 with one lane per lattice POINT (k_density_cells), shared by the kernel and the host-side wave
// emulator of the CPU tests.  Same rules as dg_geom.h / dg_density.h: the reference's operations in the reference's
// order (cmd/generate_density_map/main.cpp:86-133, gauss_quadrature.cpp:5927-5960), -ffp-contract=off.
//
// Why this shape.  A kernel with one node (or one edge's node pair) per lane -- round 3's row-block kernel, removed in round 5 --
// sits on the texture-data path: every lane pulls the 256 bytes of its cell through L1 at every quadrature point, 0.71 fetches
// per node and point once a pair shares a fetch two times out of three (profiles/r03_pmc_summary.txt: TD busy 0.98, VALU 0.81,
// and it sweeps the field four times, once per node class).  Here a lane owns ALL SEVEN nodes that hang on lattice point (i, j, k): the vertex
// and the two nodes of each of the three cell edges that start there, at 1/3 and 2/3 of the edge.  Shifted by the same
// quadrature offset the seven evaluation points lie within 2/3 of a cell of each other along one axis each, so they
// fall into the vertex point's cell c0 or into c0 + ex / c0 + ey / c0 + ez: 3.0 fetches per 7 nodes and point on average
// instead of 5.0, every class of the lattice in ONE sweep, and everything that depends on one coordinate only is shared:
// the lane keeps the axis states of the vertex point (X0 per i, Y0 per (i, j), Z0 per point) in registers and reads the
// states of the six shifted coordinates from small LDS tables the wave fills once per i (X), per (i, j) (Y) or per
// launch (Z: all sixteen k).  Every node still receives its 4096 (or 1008) terms in the reference's i, j, k order from
// the same operations on the same values, so the bits do not change (tests/test_emu.py, tests/test_gpu_density_map.py,
// the reference digests of tests/test_gpu_digests.py).
//
// A wave is a row block of 16 x 2 x 2 lattice points (lanes side by side along x: the sixteen
// coefficient pairs of a cell are 256-byte runs along x in the field's V / X classes and in the x-major copy of the
// Y / Z classes, dg_lattice.h).
#pragma once
#include "dg_density.h"

namespace dg
{

static const int kK3cLx = 16, kK3cLy = 2, kK3cLz = 2; // lattice points of a wave along x, y, z

// what k3c_value() needs of one coordinate of an evaluation point: the fields of axis_eval()'s result
struct K3Axis
{
	double m, p, fm3, fp3, t2;
	uint32_t mi;
	uint32_t inside;
};
DG_HD K3Axis k3_axis_entry(const FieldDev& F, int d, double y)
{
	const Axis1D a = axis_eval(F, d, y);
	K3Axis e;
	e.m = a.m;
	e.p = a.p;
	e.fm3 = a.fm3;
	e.fp3 = a.fp3;
	e.t2 = a.t2;
	e.mi = a.mi;
	e.inside = a.inside ? 1u : 0u;
	return e;
}

// Launch constants: the twelve row bases a cell's sixteen pair loads start from and, per class, the byte strides of a
// step of one cell along x, y, z.  A cell's offsets are LINEAR in its indices: (oV, oX, oY, oZ) = sum_d s[d] * c_d, 32-bit
// (every class below 4 GB, k3c_geometry_fits()), so that a load is "uniform base + 32-bit lane offset (+ 16)".
struct K3CellsGeom
{
	const char* bV[4]; // vertex rows (j, k), (j + 1, k), (j, k + 1), (j + 1, k + 1): coefficients 0..7
	const char* bX[4]; // X-edge rows (j, k), (j, k + 1), (j + 1, k), (j + 1, k + 1): coefficients 8..15
	const char* bY[2]; // x-major Y rows k, k + 1 (pairs at i and i + 1): coefficients 16..23
	const char* bZ[2]; // x-major Z rows j, j + 1 (pairs at i and i + 1): coefficients 24..31
	uint32_t sV[3], sX[3], sY[3], sZ[3];
};
inline bool k3c_geometry_fits(const uint32_t res[3])
{
	const uint64_t nx = res[0], ny = res[1], nz = res[2];
	const uint64_t lim = 0xffffffffull - 64;
	return 8 * (nx + 1) * (ny + 1) * (nz + 1) < lim && 16 * nx * (ny + 1) * (nz + 1) < lim && 16 * (nx + 1) * ny * (nz + 1) < lim &&
		   16 * (nx + 1) * (ny + 1) * nz < lim && 16 * (nx + 1) * (ny + 1) < (1u << 23) && nx < (1u << 20) && ny < (1u << 20) && nz < (1u << 20);
}
inline K3CellsGeom k3c_geometry(const FieldDev& F)
{
	K3CellsGeom G;
	const size_t nx = F.res[0], ny = F.res[1], nz = F.res[2];
	const char* v = reinterpret_cast<const char*>(F.coeffs);
	const size_t vj = 8 * (nx + 1), vk = 8 * (nx + 1) * (ny + 1);
	G.bV[0] = v;
	G.bV[1] = v + vj;
	G.bV[2] = v + vk;
	G.bV[3] = v + vk + vj;
	const char* ex = v + 8 * (nx + 1) * (ny + 1) * (nz + 1);
	const size_t xj = 16 * nx, xk = 16 * nx * (ny + 1);
	G.bX[0] = ex;
	G.bX[1] = ex + xk;
	G.bX[2] = ex + xj;
	G.bX[3] = ex + xk + xj;
	const char* ey = reinterpret_cast<const char*>(F.xmajor);
	G.bY[0] = ey;
	G.bY[1] = ey + 16 * ny * (nx + 1);
	const char* ez = ey + 16 * xmajor_y_pairs(F.res);
	G.bZ[0] = ez;
	G.bZ[1] = ez + 16 * (nx + 1);
	G.sV[0] = 8u;
	G.sV[1] = (uint32_t)vj;
	G.sV[2] = (uint32_t)vk;
	G.sX[0] = 16u;
	G.sX[1] = (uint32_t)xj;
	G.sX[2] = (uint32_t)xk;
	G.sY[0] = 16u;
	G.sY[1] = (uint32_t)(16 * (nx + 1));
	G.sY[2] = (uint32_t)(16 * ny * (nx + 1));
	G.sZ[0] = 16u;
	G.sZ[1] = (uint32_t)(16 * (nx + 1));
	G.sZ[2] = (uint32_t)(16 * (ny + 1) * (nx + 1));
	return G;
}

#if defined(__HIP_DEVICE_COMPILE__)
#define DG_K3C_OPAQUE(v) asm volatile("" : "+v"(v))
#define DG_K3C_OPAQUE_D(v) asm volatile("" : "+v"(v))
// a wave-uniform address of launch-constant data: through the scalar cache into scalar registers
#define DG_K3C_ULOAD(p) (*(const __attribute__((address_space(4))) double*)(uintptr_t)(p))
#else
#define DG_K3C_OPAQUE(v) (void)(v)
#define DG_K3C_OPAQUE_D(v) (void)(v)
#define DG_K3C_ULOAD(p) (*(p))
#endif
// 24-bit multiply-add in 32-bit offsets (one VALU instruction on the device; k3c_geometry_fits() keeps the operands in range)
DG_HD uint32_t k3c_mad(uint32_t a, uint32_t b, uint32_t c)
{
#if defined(__HIP_DEVICE_COMPILE__)
	return __umul24(a, b) + c;
#else
	return a * b + c;
#endif
}
struct K3Offsets
{
	uint32_t v, x, y, z;
};
// the 32 coefficients of the cell whose offsets are o: fetch_cell<kFieldXMajor>() with the addresses of K3CellsGeom
DG_HD void k3c_fetch(const K3CellsGeom& G, const K3Offsets& o, double cf[32])
{
#define DG_K3C_LD(q, base, off)                                                   \
	{                                                                             \
		const double* p_ = reinterpret_cast<const double*>((base) + (size_t)(off)); \
		cf[q] = p_[0];                                                            \
		cf[(q) + 1] = p_[1];                                                      \
	}
	DG_K3C_LD(0, G.bV[0], o.v)
	DG_K3C_LD(2, G.bV[1], o.v)
	DG_K3C_LD(4, G.bV[2], o.v)
	DG_K3C_LD(6, G.bV[3], o.v)
	DG_K3C_LD(8, G.bX[0], o.x)
	DG_K3C_LD(10, G.bX[1], o.x)
	DG_K3C_LD(12, G.bX[2], o.x)
	DG_K3C_LD(14, G.bX[3], o.x)
	DG_K3C_LD(16, G.bY[0], o.y)
	DG_K3C_LD(18, G.bY[0] + 16, o.y)
	DG_K3C_LD(20, G.bY[1], o.y)
	DG_K3C_LD(22, G.bY[1] + 16, o.y)
	DG_K3C_LD(24, G.bZ[0], o.z)
	DG_K3C_LD(26, G.bZ[1], o.z)
	DG_K3C_LD(28, G.bZ[0] + 16, o.z)
	DG_K3C_LD(30, G.bZ[1] + 16, o.z)
#undef DG_K3C_LD
}

// phi = sum_q cf[q] N_q in q order, every N_q formed where it is consumed: the statements of density_integral_t<true, .>
// (dg_density.h) on K3Axis states
DG_HD double k3c_value(const double cf[32], bool ok, const K3Axis& ax, const K3Axis& ay, const K3Axis& az)
{
	const double NOVAL = 1.7976931348623157e308;
	const double mxmy = ax.m * ay.m, mxpy = ax.m * ay.p, pxmy = ax.p * ay.m, pxpy = ax.p * ay.p;
	const double x2y2 = ax.t2 + ay.t2;
	const double mz = az.m, pz = az.p;
	const double fac = 1.0 / 64.0 * (9.0 * (x2y2 + az.t2) - 19.0);
	double phi = 0.0;
#define DG_ACC(q, n) phi += cf[q] * (n);
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
	return ok ? phi : NOVAL;
}
// d / h, correctly rounded, without the division: q0 = RN(d y), y = RN(1 / h); one residual correction makes q1 faithful
// (q0 is within 2 ulps, so q0 + (d - h q0) y differs from d / h by < 2^-104 relative), and a second one is Markstein's
// theorem (a faithful quotient corrected with the exact residual and the correctly rounded reciprocal IS the correctly rounded
// quotient; divisors with an all-ones significand excluded: DensityParams::fast_div).  The residuals are exact as long as
// nothing under- or overflows: the caller guarantees d == 0 or 1e-280 <= |d| <= 1e300 and 1e-12 <= h <= 1e12
// (tests/test_emu.py checks 2 x 10^8 quotients, half of them built to sit next to rounding boundaries, against the division).
DG_HD double k3c_div_h(double d, double h, double y)
{
#if defined(__HIP_DEVICE_COMPILE__)
	const double q0 = d * y;
	const double r0 = __builtin_fma(-h, q0, d);
	const double q1 = __builtin_fma(r0, y, q0);
	const double r1 = __builtin_fma(-h, q1, d);
	return __builtin_fma(r1, y, q1);
#else
	const double q0 = d * y;
	const double r0 = std::fma(-h, q0, d);
	const double q1 = std::fma(r0, y, q0);
	const double r1 = std::fma(-h, q1, d);
	return std::fma(r1, y, q1);
#endif
}
// (|d| <= 1e300 holds for every value of a field k_field_check passed; "no value" sums are beyond h and take gamma = 0)
DG_HD bool k3c_div_h_unsafe(double d) { return __builtin_fabs(d) < 1.0e-280 && d != 0.0; }
// one term of the quadrature sum: wijk * (gamma(d) * W)  (main.cpp:104-110, gauss_quadrature.cpp:5953); FAST: the launch
// may use k3c_div_h() (a compile-time choice: as a run-time one it makes the compiler carry some twenty lane masks
// around the k loop)
template <bool FAST>
DG_HD double k3c_term(const DensityParams& P, double d, double wijk, double wv)
{
	double gamma;
	if (FAST)
	{
		double q = k3c_div_h(d, P.h, P.rcp_h);
		if (k3c_div_h_unsafe(d))
		{
			double dd = d;
			DG_K3C_OPAQUE_D(dd); // (keeps the division out of line: speculated, it would cost more than it saves)
			q = dd / P.h;
		}
		gamma = (d > P.h) ? 0.0 : 1.0 - q;
	}
	else
		gamma = (d > P.h) ? 0.0 : 1.0 - d / P.h;
	return wijk * (gamma * wv);
}

// The seven nodes of a lattice point: 0 the vertex, 1 / 2 the X edge's nodes (A at 1/3, B at 2/3), 3 / 4 the Y edge's,
// 5 / 6 the Z edge's.  Global node indices (reference order [V | X | Y | Z], dg_lattice.h) and validity.
struct K3PointNodes
{
	bool valid[4];    // the point itself / its X / Y / Z edge exist (and the lane is no overhang)
	uint64_t node[4]; // index of the vertex and of node A of each edge (B = A + 1)
};
DG_HD K3PointNodes k3c_point_nodes(const uint32_t res[3], uint32_t i, uint32_t j, uint32_t k, bool lane_valid)
{
	const uint64_t nx = res[0], ny = res[1], nz = res[2];
	const uint64_t nv = (nx + 1) * (ny + 1) * (nz + 1), nex = 2 * nx * (ny + 1) * (nz + 1), ney = 2 * (nx + 1) * ny * (nz + 1);
	K3PointNodes n;
	n.valid[0] = lane_valid;
	n.valid[1] = lane_valid && i < nx;
	n.valid[2] = lane_valid && j < ny;
	n.valid[3] = lane_valid && k < nz;
	n.node[0] = ((uint64_t)k * (ny + 1) + j) * (nx + 1) + i;
	n.node[1] = nv + ((uint64_t)k * (ny + 1) + j) * (2 * nx) + 2 * (uint64_t)i;       // X: (2i + h, j, k), D = (2nx, ny + 1, nz + 1)
	n.node[2] = nv + nex + ((uint64_t)i * (nz + 1) + k) * (2 * ny) + 2 * (uint64_t)j; // Y: (2j + h, k, i), D = (2ny, nz + 1, nx + 1)
	n.node[3] = nv + nex + ney + ((uint64_t)j * (nx + 1) + i) * (2 * nz) + 2 * (uint64_t)k; // Z: (2k + h, i, j), D = (2nz, nx + 1, ny + 1)
	return n;
}

// the coordinates of lattice index `idx` along axis d and of the two edge nodes behind it: node_position()
DG_HD void k3c_coords(const SampleParams& L, int d, uint32_t idx, double* x0, double* xa, double* xb)
{
	*x0 = L.dmin[d] + L.cell[d] * (double)idx;
	*xa = *x0 + 1.0 / 3.0 * L.cell[d];
	*xb = *x0 + 2.0 / 3.0 * L.cell[d];
}
// Which of a lane's seven nodes depend on which coordinate: node bits 0 V, 1 / 2 X edge A / B, 3 / 4 Y edge, 5 / 6 Z edge.
// Along x the nodes V, YA, YB, ZA, ZB sit at the lattice point's own coordinate (mask 0x79), XA / XB at the shifted ones.
static const uint32_t kK3cX0 = 0x79u, kK3cXA = 0x02u, kK3cXB = 0x04u;
static const uint32_t kK3cY0 = 0x67u, kK3cYA = 0x08u, kK3cYB = 0x10u;
static const uint32_t kK3cZ0 = 0x1fu, kK3cZA = 0x20u, kK3cZB = 0x40u;
// One axis' verdict on the seven nodes for the current quadrature offset, in one register: bits 0..6 "the node's
// coordinate along this axis lies inside the domain", bits 8..14 "... and in the same cell as the lattice point's own".
// (Per-lane booleans would each take a pair of scalar registers; the kernel has none to spare.)
DG_HD uint32_t k3c_axis_bits(uint32_t m0, uint32_t ma, uint32_t mb, uint32_t mi0, bool in0, uint32_t mia, bool ina, uint32_t mib, bool inb)
{
	const uint32_t in = (in0 ? m0 : 0u) | (ina ? ma : 0u) | (inb ? mb : 0u);
	const uint32_t same = m0 | (mia == mi0 ? ma : 0u) | (mib == mi0 ? mb : 0u);
	return in | (same << 8);
}
DG_HD bool k3c_cell_ok(const FieldDev& F, uint32_t i, uint32_t j, uint32_t k)
{
	const uint32_t word = (k * F.res[1] + j) * xmajor_flag_words(F.res) + (i >> 6);
	const uint64_t w = F.xmajor_flags[word];
	return ((w >> (i & 63u)) & 1ull) == 0ull;
}
// The quadrature of one lane.  W is the wave context: any(b) (device: ballot), the tables of the shifted coordinates'
// axis states (set_x / x_var, set_y / y_var, z_var) -- LDS on the device, computed on demand by the host emulator from
// the same pure function k3_axis_entry() on the same coordinates.
//   li, lj          the lattice point's indices along x and y (its coordinates are recomputed where they are needed: the
//                   kernel is short of registers, and three instructions per i / (i, j) level are nothing)
//   need            bit n: node n needs the quadrature
//   res[n]          the sums (before * c0prod, * rho0)
template <bool FAST, class W>
DG_HD void k3c_quadrature(W& w, const SampleParams& L, const FieldDev& F, const DensityParams& P, const K3CellsGeom& G, bool has_noval,
						  bool skip, uint32_t li, uint32_t lj, uint32_t need, double res[7])
{
	w.acc_init();
	DG_NOUNROLL
	for (int i = 0; i < 16; ++i)
	{
		double cx0, cxa, cxb;
		{
			uint32_t t = li;
			DG_K3C_OPAQUE(t);
			k3c_coords(L, 0, t, &cx0, &cxa, &cxb);
		}
		const K3Axis X0 = k3_axis_entry(F, 0, cx0 + P.xi[i]);
		w.set_x(F, cxa + P.xi[i], cxb + P.xi[i]);
		uint32_t xbits;
		{
			uint32_t a_mi, b_mi;
			bool a_in, b_in;
			w.x_id(F, 0, &a_mi, &a_in);
			w.x_id(F, 1, &b_mi, &b_in);
			xbits = k3c_axis_bits(kK3cX0, kK3cXA, kK3cXB, X0.mi, X0.inside != 0u, a_mi, a_in, b_mi, b_in);
		}
		DG_NOUNROLL
		for (int j = 0; j < 16; ++j)
		{
			const uint32_t kmask = skip ? (uint32_t)P.kmask[i * 16 + j] : 0xffffu;
			if (kmask == 0u)
				continue; // the whole column lies outside the kernel's support
			double cy0, cya, cyb;
			{
				uint32_t t = lj;
				DG_K3C_OPAQUE(t);
				k3c_coords(L, 1, t, &cy0, &cya, &cyb);
			}
			const K3Axis Y0 = k3_axis_entry(F, 1, cy0 + P.xi[j]);
			w.set_y(F, cya + P.xi[j], cyb + P.xi[j]);
			uint32_t xyb; // need & x & y: bits 0..6 "has a point here so far", bits 8..14 "in the lattice point's cell so far"
			{
				uint32_t a_mi, b_mi;
				bool a_in, b_in;
				w.y_id(F, 0, &a_mi, &a_in);
				w.y_id(F, 1, &b_mi, &b_in);
				const uint32_t ybits = k3c_axis_bits(kK3cY0, kK3cYA, kK3cYB, Y0.mi, Y0.inside != 0u, a_mi, a_in, b_mi, b_in);
				xyb = xbits & ybits & (need | 0xff00u);
			}
			// offsets of cell (X0.mi, Y0.mi, 0)
			K3Offsets oxy;
			oxy.v = k3c_mad(Y0.mi, G.sV[1], X0.mi * G.sV[0]);
			oxy.x = k3c_mad(Y0.mi, G.sX[1], X0.mi * G.sX[0]);
			oxy.y = k3c_mad(Y0.mi, G.sY[1], X0.mi * G.sY[0]);
			oxy.z = k3c_mad(Y0.mi, G.sZ[1], X0.mi * G.sZ[0]);
			DG_NOUNROLL
			for (int k = 0; k < 16; ++k)
			{
				if (((kmask >> k) & 1u) == 0u)
					continue;
				// (w_i w_j) w_k and W(xi_i, xi_j, xi_k) from the launch's table: scalars
				const double wijk = DG_K3C_ULOAD(P.wtab + 4096 + (i * 16 + j) * 16 + k);
				const double wv = DG_K3C_ULOAD(P.wtab + (i * 16 + j) * 16 + k);
				const K3Axis Z0 = w.z_var(F, P, k, 0);
				// which nodes have a point to evaluate here (a point outside the domain contributes wijk * (0 * W) = +0.0 to a
				// sum that starts at +0.0 and never turns -0.0: nothing), and which of them share the lattice point's cell
				const uint32_t t = xyb & w.z_bits(F, P, k);
				const uint32_t act = t & 0x7fu;
				const uint32_t g = act & (t >> 8);
				K3Offsets o0;
				o0.v = k3c_mad(Z0.mi, G.sV[2], oxy.v);
				o0.x = k3c_mad(Z0.mi, G.sX[2], oxy.x);
				o0.y = k3c_mad(Z0.mi, G.sY[2], oxy.y);
				o0.z = k3c_mad(Z0.mi, G.sZ[2], oxy.z);
				double cf[32];
				bool ok = true;
#define DG_K3C_EVAL(mask, n, AX, AY, AZ)                                   \
	if (mask)                                                             \
	{                                                                     \
		const double d_ = k3c_value(cf, ok, AX, AY, AZ);                  \
		w.acc_add(n, k3c_term<FAST>(P, d_, wijk, wv));                 \
	}
#define DG_K3C_IF(mask) if (w.any(mask)) if (mask)
				// ---- the cell of the lattice point ----
				if (w.any(g != 0u))
				{
					if (g != 0u)
					{
						k3c_fetch(G, o0, cf);
						if (has_noval)
							ok = k3c_cell_ok(F, X0.mi, Y0.mi, Z0.mi);
					}
					DG_K3C_IF((g & 1u) != 0u)
					{
						DG_K3C_EVAL(true, 0, X0, Y0, Z0)
					}
					DG_K3C_IF((g & kK3cXA) != 0u)
					{
						const K3Axis XA = w.x_var(F, 0);
						DG_K3C_EVAL(true, 1, XA, Y0, Z0)
					}
					DG_K3C_IF((g & kK3cXB) != 0u)
					{
						const K3Axis XB = w.x_var(F, 1);
						DG_K3C_EVAL(true, 2, XB, Y0, Z0)
					}
					DG_K3C_IF((g & kK3cYA) != 0u)
					{
						const K3Axis YA = w.y_var(F, 0);
						DG_K3C_EVAL(true, 3, X0, YA, Z0)
					}
					DG_K3C_IF((g & kK3cYB) != 0u)
					{
						const K3Axis YB = w.y_var(F, 1);
						DG_K3C_EVAL(true, 4, X0, YB, Z0)
					}
					DG_K3C_IF((g & kK3cZA) != 0u)
					{
						const K3Axis ZA = w.z_var(F, P, k, 1);
						DG_K3C_EVAL(true, 5, X0, Y0, ZA)
					}
					DG_K3C_IF((g & kK3cZB) != 0u)
					{
						const K3Axis ZB = w.z_var(F, P, k, 2);
						DG_K3C_EVAL(true, 6, X0, Y0, ZB)
					}
				}
				const uint32_t rest = act & ~g;
				// ---- the edge nodes beyond a cell face: normally ONE more cell per axis (c0 + ex for both X nodes, ...); if
				// the two nodes of an edge ever sit in two further cells, a second round takes node A.  Straight-line code: a
				// loop would carry the 32 coefficients around its back edge in a second set of registers.
#define DG_K3C_NEIGHBOUR(r_, MA_, MB_, ID_A, ID_B, C0MI, SV, SX, SY, SZ, OKCELL, VAR_A, VAR_B, NA, NB, AXA, AYA, AZA, AXB, AYB, AZB)     \
	if (w.any(r_ != 0u))                                                                                                           \
	{                                                                                                                              \
		uint32_t a_mi, b_mi;                                                                                                       \
		bool a_in, b_in;                                                                                                           \
		ID_A;                                                                                                                      \
		ID_B;                                                                                                                      \
		(void)a_in;                                                                                                                \
		(void)b_in;                                                                                                                \
		const uint32_t cc = (r_ & MB_) != 0u ? b_mi : a_mi;                                                                        \
		const bool mB = (r_ & MB_) != 0u, mA = (r_ & MA_) != 0u && a_mi == cc;                                                     \
		if (mA || mB)                                                                                                              \
		{                                                                                                                          \
			uint32_t dd = cc - (C0MI);                                                                                             \
			DG_K3C_OPAQUE(dd); /* (loop-invariant products of the stride are not worth a register each) */                         \
			K3Offsets o;                                                                                                           \
			o.v = k3c_mad(dd, SV, o0.v);                                                                                           \
			o.x = k3c_mad(dd, SX, o0.x);                                                                                           \
			o.y = k3c_mad(dd, SY, o0.y);                                                                                           \
			o.z = k3c_mad(dd, SZ, o0.z);                                                                                           \
			k3c_fetch(G, o, cf);                                                                                                   \
			ok = true;                                                                                                             \
			if (has_noval)                                                                                                         \
				ok = OKCELL;                                                                                                       \
		}                                                                                                                          \
		DG_K3C_IF(mB)                                                                                                              \
		{                                                                                                                          \
			const K3Axis VB = VAR_B;                                                                                               \
			DG_K3C_EVAL(true, NB, AXB, AYB, AZB)                                                                                   \
		}                                                                                                                          \
		DG_K3C_IF(mA)                                                                                                              \
		{                                                                                                                          \
			const K3Axis VA = VAR_A;                                                                                               \
			DG_K3C_EVAL(true, NA, AXA, AYA, AZA)                                                                                   \
		}                                                                                                                          \
		r_ = mA ? 0u : (r_ & MA_);                                                                                                 \
	}
				{
					uint32_t r = rest & (kK3cXA | kK3cXB);
					DG_K3C_NEIGHBOUR(r, kK3cXA, kK3cXB, w.x_id(F, 0, &a_mi, &a_in), w.x_id(F, 1, &b_mi, &b_in), X0.mi, G.sV[0], G.sX[0], G.sY[0],
									 G.sZ[0], k3c_cell_ok(F, cc, Y0.mi, Z0.mi), w.x_var(F, 0), w.x_var(F, 1), 1, 2, VA, Y0, Z0, VB, Y0, Z0)
					DG_K3C_NEIGHBOUR(r, kK3cXA, kK3cXB, w.x_id(F, 0, &a_mi, &a_in), w.x_id(F, 1, &b_mi, &b_in), X0.mi, G.sV[0], G.sX[0], G.sY[0],
									 G.sZ[0], k3c_cell_ok(F, cc, Y0.mi, Z0.mi), w.x_var(F, 0), w.x_var(F, 1), 1, 2, VA, Y0, Z0, VB, Y0, Z0)
				}
				{
					uint32_t r = rest & (kK3cYA | kK3cYB);
					DG_K3C_NEIGHBOUR(r, kK3cYA, kK3cYB, w.y_id(F, 0, &a_mi, &a_in), w.y_id(F, 1, &b_mi, &b_in), Y0.mi, G.sV[1], G.sX[1], G.sY[1],
									 G.sZ[1], k3c_cell_ok(F, X0.mi, cc, Z0.mi), w.y_var(F, 0), w.y_var(F, 1), 3, 4, X0, VA, Z0, X0, VB, Z0)
					DG_K3C_NEIGHBOUR(r, kK3cYA, kK3cYB, w.y_id(F, 0, &a_mi, &a_in), w.y_id(F, 1, &b_mi, &b_in), Y0.mi, G.sV[1], G.sX[1], G.sY[1],
									 G.sZ[1], k3c_cell_ok(F, X0.mi, cc, Z0.mi), w.y_var(F, 0), w.y_var(F, 1), 3, 4, X0, VA, Z0, X0, VB, Z0)
				}
				{
					uint32_t r = rest & (kK3cZA | kK3cZB);
					DG_K3C_NEIGHBOUR(r, kK3cZA, kK3cZB, w.z_id(F, P, k, 1, &a_mi, &a_in), w.z_id(F, P, k, 2, &b_mi, &b_in), Z0.mi, G.sV[2], G.sX[2],
									 G.sY[2], G.sZ[2], k3c_cell_ok(F, X0.mi, Y0.mi, cc), w.z_var(F, P, k, 1), w.z_var(F, P, k, 2), 5, 6, X0, Y0, VA, X0, Y0, VB)
					DG_K3C_NEIGHBOUR(r, kK3cZA, kK3cZB, w.z_id(F, P, k, 1, &a_mi, &a_in), w.z_id(F, P, k, 2, &b_mi, &b_in), Z0.mi, G.sV[2], G.sX[2],
									 G.sY[2], G.sZ[2], k3c_cell_ok(F, X0.mi, Y0.mi, cc), w.z_var(F, P, k, 1), w.z_var(F, P, k, 2), 5, 6, X0, Y0, VA, X0, Y0, VB)
				}
#undef DG_K3C_NEIGHBOUR
#undef DG_K3C_EVAL
#undef DG_K3C_IF
			}
		}
	}
	for (int n = 0; n < 7; ++n)
		res[n] = w.acc_get(n);
}

// Everything one lane of a wave of k_density_cells does.  m: the wave's coordinates in the grid of row blocks
// (row_wave_map() of DensityParams: a single "class" of ceil((n + 1) / 16) x ceil((n + 1) / 2) x ceil((n + 1) / 2) waves).
template <class W>
DG_HD void k3c_lane(W& w, const SampleParams& L, const FieldDev& F, const DensityParams& P, const K3CellsGeom& G, const RowWave& m, int lane)
{
	const double NOVAL = 1.7976931348623157e308;
	const uint32_t nx = F.res[0], ny = F.res[1], nz = F.res[2];
	uint32_t i = m.w[0] * (uint32_t)kK3cLx + ((uint32_t)lane & 15u), j = m.w[1] * (uint32_t)kK3cLy + (((uint32_t)lane >> 4) & 1u),
			 k = m.w[2] * (uint32_t)kK3cLz + ((uint32_t)lane >> 5);
	const bool lane_valid = i <= nx && j <= ny && k <= nz;
	i = i <= nx ? i : nx;
	j = j <= ny ? j : ny;
	k = k <= nz ? k : nz;
	// node_position(): dmin + cell * index, and += (1/3 | 2/3) * cell along the edge's axis
	double x0[3], xa[3], xb[3];
	k3c_coords(L, 0, i, &x0[0], &xa[0], &xb[0]);
	k3c_coords(L, 1, j, &x0[1], &xa[1], &xb[1]);
	k3c_coords(L, 2, k, &x0[2], &xa[2], &xb[2]);
	const K3PointNodes pn = k3c_point_nodes(F.res, i, j, k, lane_valid);
	// stage 1 per node: range, mask, node predicate and the early-out of density_func; what needs no quadrature is final
	uint32_t need = 0u;
	DG_NOUNROLL
	for (int n = 0; n < 7; ++n)
	{
		const int e = (n + 1) >> 1; // 0: vertex, 1..3: the X / Y / Z edge
		const bool second = n != 0 && (n & 1) == 0;
		const bool valid = e == 0 ? pn.valid[0] : (e == 1 ? pn.valid[1] : (e == 2 ? pn.valid[2] : pn.valid[3]));
		const uint64_t l = (e == 0 ? pn.node[0] : (e == 1 ? pn.node[1] : (e == 2 ? pn.node[2] : pn.node[3]))) + (second ? 1u : 0u);
		double x[3] = {x0[0], x0[1], x0[2]};
		if (e == 1)
			x[0] = second ? xb[0] : xa[0];
		else if (e == 2)
			x[1] = second ? xb[1] : xa[1];
		else if (e == 3)
			x[2] = second ? xb[2] : xa[2];
		const bool mine = valid && l >= P.row_node_begin && l < P.row_node_end;
		if (mine)
		{
			const uint64_t o = l - P.row_node_begin;
			double v = NOVAL;
			bool q = false;
			if (L.mask == nullptr || L.mask[o] != 0)
				q = density_prefilter(F, P, x, &v);
			if (q)
				need |= 1u << n;
			else
				L.out[o] = v;
		}
	}
	if (!w.any(need != 0u))
		return;
	// bit 0: NaN / Inf / huge values, bit 1: "no value" coefficients.  One word for the whole launch: told to the compiler
	// (uniform()), or every branch on it becomes a lane mask carried around the loops
	const uint32_t flags = w.uniform(P.unsafe ? P.unsafe[0] : 2u);
	const bool has_noval = (flags & 2u) != 0u; // (answered by the x-major copy's one bit per cell: the launch supplies F.xmajor_flags)
	const bool skip = P.skip_mode == 1 || (P.skip_mode == 2 && (flags & 1u) == 0u);
	w.set_z(F, P, L, m, x0[2], xa[2], xb[2]);
	double res[7];
	// (skip: the field holds no NaN / Inf / huge value, k_field_check -- what k3c_div_h() needs as well)
	if (skip && P.fast_div != 0)
		k3c_quadrature<true>(w, L, F, P, G, has_noval, skip, i, j, need, res);
	else
		k3c_quadrature<false>(w, L, F, P, G, has_noval, skip, i, j, need, res);
	// (the node indices again: not kept across the quadrature)
	uint32_t i2 = i, j2 = j, k2 = k;
	DG_K3C_OPAQUE(i2);
	DG_K3C_OPAQUE(j2);
	DG_K3C_OPAQUE(k2);
	const K3PointNodes pn2 = k3c_point_nodes(F.res, i2, j2, k2, lane_valid);
#define DG_K3C_OUT(n, e, second)                                                        \
	if ((need >> (n)) & 1u)                                                             \
	{                                                                                   \
		double r_ = res[n];                                                             \
		r_ *= P.c0prod;                                                                 \
		L.out[pn2.node[e] + (second) - P.row_node_begin] = P.rho0 * r_;                 \
	}
	DG_K3C_OUT(0, 0, 0u)
	DG_K3C_OUT(1, 1, 0u)
	DG_K3C_OUT(2, 1, 1u)
	DG_K3C_OUT(3, 2, 0u)
	DG_K3C_OUT(4, 2, 1u)
	DG_K3C_OUT(5, 3, 0u)
	DG_K3C_OUT(6, 3, 1u)
#undef DG_K3C_OUT
}

// The host's wave context: one lane at a time, every table entry computed where it is read -- k3_axis_entry() of the
// same coordinate the device's producer lane uses (the coordinates depend on the lattice index along one axis only).
struct K3HostWave
{
	double xa_ = 0.0, xb_ = 0.0, ya_ = 0.0, yb_ = 0.0;
	double z_[3] = {0.0, 0.0, 0.0};
	double acc_[7];
	bool any(bool b) const { return b; }
	uint32_t uniform(uint32_t v) const { return v; }
	void acc_init()
	{
		for (int n = 0; n < 7; ++n)
			acc_[n] = 0.0;
	}
	void acc_add(int n, double t) { acc_[n] += t; }
	double acc_get(int n) const { return acc_[n]; }
	void set_x(const FieldDev&, double a, double b)
	{
		xa_ = a;
		xb_ = b;
	}
	void set_y(const FieldDev&, double a, double b)
	{
		ya_ = a;
		yb_ = b;
	}
	K3Axis x_var(const FieldDev& F, int v) const { return k3_axis_entry(F, 0, v ? xb_ : xa_); }
	K3Axis y_var(const FieldDev& F, int v) const { return k3_axis_entry(F, 1, v ? yb_ : ya_); }
	void x_id(const FieldDev& F, int v, uint32_t* mi, bool* in) const
	{
		const K3Axis e = x_var(F, v);
		*mi = e.mi;
		*in = e.inside != 0u;
	}
	void y_id(const FieldDev& F, int v, uint32_t* mi, bool* in) const
	{
		const K3Axis e = y_var(F, v);
		*mi = e.mi;
		*in = e.inside != 0u;
	}
	// the lane's own z coordinates: lattice point, node A, node B of its Z edge (the device fills the wave's table for
	// all sixteen k from the wave's coordinates instead)
	void set_z(const FieldDev&, const DensityParams&, const SampleParams&, const RowWave&, double z0, double za, double zb)
	{
		z_[0] = z0;
		z_[1] = za;
		z_[2] = zb;
	}
	K3Axis z_var(const FieldDev& F, const DensityParams& P, int k, int v) const { return k3_axis_entry(F, 2, z_[v] + P.xi[k]); }
	void z_id(const FieldDev& F, const DensityParams& P, int k, int v, uint32_t* mi, bool* in) const
	{
		const K3Axis e = z_var(F, P, k, v);
		*mi = e.mi;
		*in = e.inside != 0u;
	}
	uint32_t z_bits(const FieldDev& F, const DensityParams& P, int k) const
	{
		const K3Axis z0 = z_var(F, P, k, 0), za = z_var(F, P, k, 1), zb = z_var(F, P, k, 2);
		return k3c_axis_bits(kK3cZ0, kK3cZA, kK3cZB, z0.mi, z0.inside != 0u, za.mi, za.inside != 0u, zb.mi, zb.inside != 0u);
	}
};

} // namespace dg
