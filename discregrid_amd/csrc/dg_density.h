// dg_density.h -- K3's per-node arithmetic shared by the kernel and the host-side test mirror: the
// SPH boundary density map of the reference's GenerateDensityMap tool
// (cmd/generate_density_map/main.cpp:86-133, sph_kernel.hpp:11-42, gauss_quadrature.cpp:5927-5960).
// Same rules as dg_geom.h.
#pragma once
#include "dg_lattice.h"

namespace dg
{

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
	const double* wtab; // 4096 values W(xi_i, xi_j, xi_k), index (i*16 + j)*16 + k (+ 4096 weight products, dg_layout.h init_density_params())
	double rcp_h;       // RN(1 / h): the correctly rounded reciprocal, for k3c_div_h()
	int32_t fast_div;   // h allows the division-free d / h (k_density_cells, fields without NaN / Inf / huge values only)
	// Quadrature points outside the kernel's support (|xi| > h: 3088 of the 4096 points) contribute
	// w * (gamma * 0.0) = +0.0 to a sum of non-negative terms, i.e. nothing -- provided gamma is
	// finite, which holds whenever every coefficient other than DBL_MAX is finite and below 1e290.
	// kmask[i*16 + j] has bit k set where W(xi_i, xi_j, xi_k) != 0.
	// skip_mode 0: evaluate every point; 1: skip the zero-weight points; 2 (device): skip them unless
	// bit 0 of *unsafe is set (by k_field_check when the field holds NaN / Inf / huge values; bit 1 of the same
	// word: the field holds "no value" coefficients).
	uint16_t kmask[256];
	int32_t skip_mode;
	// device, x-major copy: kRowShapeCells = k_density_cells (a lane owns a lattice POINT with its seven nodes, waves of 16 x 2 x 2
	// points, dg_density_cells.h), 0: the brick kernel; waves along x / y / z of the blocks consecutive wave ids fill, and the wave
	// counts along the three axes with the first wave id
	int32_t row_shape;
	uint64_t row_node_begin, row_node_end; // k_density_cells: the launch's node range (out[l - row_node_begin]); lanes outside idle
	uint32_t row_block[3];
	uint32_t row_waves[4][3];
	uint32_t row_prefix[5];
	const uint32_t* unsafe;
};

static const int kRowShapeCells = 6;
// k_density_cells: wave id -> (class, wave coordinates along x, y, z).  The waves of a class fill blocks of
// row_block[0] x [1] x [2] waves (truncated at the upper faces), blocks in row-major order: the waves an XCD has in
// flight -- consecutive ids -- integrate over (nearly) the same part of the field at the same time.  A bijection of
// [row_prefix[c], row_prefix[c + 1]) onto the class's waves, all in wave-uniform integers.
struct RowWave
{
	int cls;
	uint32_t w[3];
};
DG_HD RowWave row_wave_map(const DensityParams& P, uint32_t id)
{
	int c = 0;
	if (id >= P.row_prefix[1]) c = 1;
	if (id >= P.row_prefix[2]) c = 2;
	if (id >= P.row_prefix[3]) c = 3;
	const uint32_t local = id - P.row_prefix[c];
	const uint32_t n0 = P.row_waves[c][0], n1 = P.row_waves[c][1], n2 = P.row_waves[c][2];
	const uint32_t B0 = P.row_block[0], B1 = P.row_block[1], B2 = P.row_block[2];
	const uint32_t slab = n0 * n1 * B2;
	const uint32_t i2 = local / slab, r = local - i2 * slab;
	const uint32_t s2 = (n2 - i2 * B2) < B2 ? (n2 - i2 * B2) : B2;
	const uint32_t row = n0 * B1 * s2;
	const uint32_t i1 = r / row, r2 = r - i1 * row;
	const uint32_t s1 = (n1 - i1 * B1) < B1 ? (n1 - i1 * B1) : B1;
	const uint32_t blk = B0 * s1 * s2;
	const uint32_t i0 = r2 / blk, r3 = r2 - i0 * blk;
	const uint32_t s0 = (n0 - i0 * B0) < B0 ? (n0 - i0 * B0) : B0;
	RowWave m;
	m.cls = c;
	m.w[0] = i0 * B0 + r3 % s0;
	m.w[1] = i1 * B1 + (r3 / s0) % s1;
	m.w[2] = i2 * B2 + r3 / (s0 * s1);
	return m;
}

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
template <bool STAGED, int MODE>
DG_HD double density_integral_t(const FieldDev& F, const DensityParams& P, const double x[3])
{
	const double NOVAL = 1.7976931348623157e308;
	double g[3];
	double res = 0.0;
	const bool staged = STAGED;
	const bool skip = P.skip_mode == 1 || (P.skip_mode == 2 && (P.unsafe[0] & 1u) == 0u);
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
						fetch_cell<MODE>(F, ax.mi, ay.mi, az.mi, ci, cf);
						const double mz = az.m, pz = az.p;
						const double fac = 1.0 / 64.0 * (9.0 * (x2y2 + az.t2) - 19.0);
						// phi = sum_q cf[q] * N[q] in q order; every N[q] is formed right where it is
						// consumed (same products as shape_functions(), no 32-entry array kept live)
						// "no value" coefficients: one flag bit per cell in the tile-major copy, else 32 compares
						// (the x-major copy: one bit per cell if its producer supplied them)
						const bool cell_flags = MODE == kFieldTileMajor || (MODE == kFieldXMajor && F.xmajor_flags != nullptr);
						bool ok = (MODE == kFieldTileMajor) ? !tile_cell_has_novalue(F.tile_major, F.ntile, ax.mi, ay.mi, az.mi) : true;
						if (MODE == kFieldXMajor && cell_flags)
							ok = !xmajor_cell_has_novalue(F, ax.mi, ay.mi, az.mi);
						double phi = 0.0;
#define DG_ACC(q, n)                                   \
	if (!cell_flags)                                   \
		ok = ok && (cf[q] != NOVAL);                   \
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
					d = interpolate_point_mode<false, MODE>(F, y, g);
				}
				const double gamma = (d > P.h) ? 0.0 : 1.0 - d / P.h;
				res += wijk * (gamma * P.wtab[(i * 16 + j) * 16 + k]);
			}
		}
	}
	res *= P.c0prod;
	return P.rho0 * res;
}
// runtime dispatch (emulator); the kernels are instantiated per (staged, mode)
DG_HD double density_integral(const FieldDev& F, const DensityParams& P, const double x[3])
{
	const bool unreduced = (F.cells == nullptr) && (F.cell_map == nullptr);
	switch (field_mode(F))
	{
	case kFieldXMajor: return density_integral_t<true, kFieldXMajor>(F, P, x);       // (unreduced fields only)
	case kFieldTileMajor: return density_integral_t<true, kFieldTileMajor>(F, P, x); // (unreduced fields only)
	case kFieldCellMajor: return unreduced ? density_integral_t<true, kFieldCellMajor>(F, P, x) : density_integral_t<false, kFieldCellMajor>(F, P, x);
	case kFieldTable: return density_integral_t<false, kFieldTable>(F, P, x);
	default: return unreduced ? density_integral_t<true, kFieldClosed>(F, P, x) : density_integral_t<false, kFieldClosed>(F, P, x);
	}
}

} // namespace dg
