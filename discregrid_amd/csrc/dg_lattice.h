// dg_lattice.h -- the grid side of the per-lane arithmetic shared by kernels and host code: node
// classes and positions, the 32 serendipity-cubic shape functions, the closed-form cell -> node
// table and the per-query body of K2.  Same rules as dg_geom.h: the reference's association order,
// -ffp-contract=off, no memory touched other than the arguments.
//
// Reference lines restated (paths relative to the Discregrid tree):
//   node positions           discregrid/src/cubic_lagrange_discrete_grid.cpp:604-665
//   shape functions          discregrid/src/cubic_lagrange_discrete_grid.cpp:339-580
//   interpolate              discregrid/src/cubic_lagrange_discrete_grid.cpp:977-1063
#pragma once
#include "dg_geom.h"

namespace dg
{

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
	// the reference's (1.0 + h) / 3.0 is one of two correctly rounded constants; the axis is picked with
	// (wave-uniform) branches rather than a dynamic index, which costs a chain of selects per coordinate
	const double third = h == 0u ? 1.0 / 3.0 : 2.0 / 3.0;
	if (c == 1)
		x[0] += third * cell[0];
	else if (c == 2)
		x[1] += third * cell[1];
	else if (c == 3)
		x[2] += third * cell[2];
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

// ---- tile-major copy of an unreduced field ------------------------------------------------------------------------
// The reference's coefficient vector is [V | X | Y | Z] with a different fastest axis per class: the 32
// coefficients of a cell sit in 16 different cache lines, megabytes apart, and an evaluator that gathers from
// it pulls a whole 128-byte line into L1 for every 16 bytes it uses (measured: K2 on cell-SORTED queries ran
// at 0.78 ms per 10 M, bound by L1 fills, not by HBM).  The tile-major copy stores, for every tile of 4x4x4
// cells, ALL nodes the tile's cells reference -- 5^3 vertex nodes and 3 x 4 x 5 x 5 edges of two nodes = 725
// doubles, padded to 736 = 46 lines -- contiguously: the queries of one wave, which lie in one or two tiles
// once a batch is in tile order, hit the same few dozen lines.  Nodes on tile faces are stored once per
// adjacent tile (1.64 x the field's memory; the cell-major layout costs 4.6 x).  Same values, same
// arithmetic: bit-identical results.
static const uint32_t kTmCells = 4;     // tile edge in cells
static const uint32_t kTmNodes = 736;   // doubles per tile (725 used)
// first slot of the X / Y / Z edge nodes of a tile: multiples of 8 (three pad slots after the 125 vertices).  The four
// edges a row of four cells needs along the edges' own axis are 8 doubles = one 64-byte sector, and with the blocks on
// sector boundaries the four lanes of such a row read ONE sector (K3's roof is the texture-data path, whose work is the
// number of sectors an instruction touches: 31 per 64-lane load with the vertices packed to 125 and rows straddling)
static const uint32_t kTmX = 128, kTmY = 328, kTmZ = 528;
// Slot kTmFlags of a tile holds 64 bits (stored in the double's place): bit (lk 4 + lj) 4 + li is set iff one of the
// 32 coefficients of the tile's cell (li, lj, lk) is DBL_MAX ("no value").  K3 tests that bit instead of comparing
// all 32 coefficients at each of its thousand quadrature points per node.
static const uint32_t kTmFlags = 728;
// slots (tile-local) of the 32 nodes of the cell with tile-local coordinates (li, lj, lk), in the order of
// cell_node_indices(); pairs (2m, 2m+1) are adjacent as there
DG_HD void tile_node_slots(uint32_t li, uint32_t lj, uint32_t lk, uint32_t out[32])
{
	const uint32_t v0 = (lk * 5 + lj) * 5 + li;
	out[0] = v0;
	out[2] = v0 + 5;
	out[4] = v0 + 25;
	out[6] = v0 + 30;
	for (int m = 0; m < 8; m += 2)
		out[m + 1] = out[m] + 1;
	// X edges (along x, at (j, k)): slot kTmX + 2 ((k 5 + j) 4 + i)
	out[8] = kTmX + 2 * ((lk * 5 + lj) * 4 + li);
	out[10] = kTmX + 2 * (((lk + 1) * 5 + lj) * 4 + li);
	out[12] = kTmX + 2 * ((lk * 5 + lj + 1) * 4 + li);
	out[14] = kTmX + 2 * (((lk + 1) * 5 + lj + 1) * 4 + li);
	// Y edges (along y, at (i, k)): slot kTmY + 2 ((i 5 + k) 4 + j)
	out[16] = kTmY + 2 * ((li * 5 + lk) * 4 + lj);
	out[18] = kTmY + 2 * (((li + 1) * 5 + lk) * 4 + lj);
	out[20] = kTmY + 2 * ((li * 5 + lk + 1) * 4 + lj);
	out[22] = kTmY + 2 * (((li + 1) * 5 + lk + 1) * 4 + lj);
	// Z edges (along z, at (j, i)): slot kTmZ + 2 ((j 5 + i) 4 + k)
	out[24] = kTmZ + 2 * ((lj * 5 + li) * 4 + lk);
	out[26] = kTmZ + 2 * (((lj + 1) * 5 + li) * 4 + lk);
	out[28] = kTmZ + 2 * ((lj * 5 + li + 1) * 4 + lk);
	out[30] = kTmZ + 2 * (((lj + 1) * 5 + li + 1) * 4 + lk);
	for (int m = 8; m < 32; m += 2)
		out[m + 1] = out[m] + 1;
}
// does cell (i, j, k) of an unreduced field with a tile-major copy hold a "no value" coefficient?
struct FieldDev;
DG_HD bool tile_cell_has_novalue(const double* tile_major, const uint32_t ntile[3], uint32_t i, uint32_t j, uint32_t k)
{
	const double* tile = tile_major + (size_t)kTmNodes * ((size_t)((k >> 2) * ntile[1] + (j >> 2)) * ntile[0] + (i >> 2));
	const uint64_t flags = *(const uint64_t*)(tile + kTmFlags);
	return ((flags >> (((k & 3u) * 4u + (j & 3u)) * 4u + (i & 3u))) & 1ull) != 0ull;
}
// global node index (reference order) stored in slot `slot` of tile (ti, tj, tk); 0xffffffff for padding slots
// and for nodes beyond the lattice (tiles that stick out of a resolution that is no multiple of 4)
DG_HD uint32_t tile_slot_node(uint32_t slot, uint32_t ti, uint32_t tj, uint32_t tk, const uint32_t res[3])
{
	const uint32_t nx = res[0], ny = res[1], nz = res[2];
	const uint32_t i0 = ti * kTmCells, j0 = tj * kTmCells, k0 = tk * kTmCells;
	if (slot >= 125 && slot < kTmX)
		return 0xffffffffu; // the pad slot behind the vertices
	if (slot < kTmX)
	{
		const uint32_t a = slot % 5, b = (slot / 5) % 5, c = slot / 25;
		const uint32_t i = i0 + a, j = j0 + b, k = k0 + c;
		if (i > nx || j > ny || k > nz)
			return 0xffffffffu;
		return (nx + 1) * (ny + 1) * k + (nx + 1) * j + i;
	}
	if (slot >= kTmZ + 200)
		return 0xffffffffu;
	const uint32_t nv = (nx + 1) * (ny + 1) * (nz + 1);
	const uint32_t nex = nx * (ny + 1) * (nz + 1);
	const uint32_t ney = (nx + 1) * ny * (nz + 1);
	const uint32_t cls = slot < kTmY ? 0u : (slot < kTmZ ? 1u : 2u);
	const uint32_t e = (slot - (cls == 0 ? kTmX : (cls == 1 ? kTmY : kTmZ)));
	const uint32_t sub = e & 1u, q = e >> 1; // q = (s2 5 + s1) 4 + s0
	const uint32_t s0 = q % 4, s1 = (q / 4) % 5, s2 = q / 20;
	if (cls == 0) // X: s0 = i, s1 = j, s2 = k
	{
		const uint32_t i = i0 + s0, j = j0 + s1, k = k0 + s2;
		if (i >= nx || j > ny || k > nz)
			return 0xffffffffu;
		return nv + 2 * (nx * (ny + 1) * k + nx * j + i) + sub;
	}
	if (cls == 1) // Y: s0 = j, s1 = k, s2 = i
	{
		const uint32_t j = j0 + s0, k = k0 + s1, i = i0 + s2;
		if (i > nx || j >= ny || k > nz)
			return 0xffffffffu;
		return nv + 2 * nex + 2 * (ny * (nz + 1) * i + ny * k + j) + sub;
	}
	// Z: s0 = k, s1 = i, s2 = j
	const uint32_t k = k0 + s0, i = i0 + s1, j = j0 + s2;
	if (i > nx || j > ny || k >= nz)
		return 0xffffffffu;
	return nv + 2 * nex + 2 * ney + 2 * (nz * (nx + 1) * j + nz * i + k) + sub;
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
	// Optional tile-major copy of an UNREDUCED field (see above): kTmNodes doubles per tile of 4^3 cells,
	// tiles in x-fastest order, ntile[d] = ceil(res[d] / 4).
	const double* tile_major;
	uint32_t ntile[3];
	// Optional x-major copy of the Y and Z edge classes of an UNREDUCED field (see "x-major copy" below): with it
	// all 16 coefficient pairs of a cell lie in rows that run along x.  xmajor_flags: one "no value" bit per cell
	// (nullable; valid only where the producer says so).
	const double* xmajor = nullptr;
	const uint64_t* xmajor_flags = nullptr;
	// Optional BAND-LIMITED cell-major copy (round 4): 256-byte rows like cell_major, but only for the cell rows whose
	// coefficients reach into a value band [lo, hi] (SPH boundary handling and GenerateDensityMap query the shell
	// |phi| < 2h around the surface: 10-20 % of the cells, i.e. less than 1 x the field instead of 4.6 x).
	// Which cell rows have one: one bit per cell row (band_bits, 64 rows per word) and the number of rows before each word
	// (band_rank) -- 12 bytes per 64 cells, 3 MB at 256^3: the look-up of a query stays in the L2s instead of costing a sector
	// of HBM traffic per query as a 4-byte-per-cell map would.  Queries into cells without a row gather from the field.
	const double* band_rows = nullptr;
	const uint64_t* band_bits = nullptr;
	const uint32_t* band_rank = nullptr;
};
// index of cell row `row` in the band copy, 0xffffffff if it has none
DG_HD uint32_t band_row_of(const FieldDev& F, uint32_t row)
{
	const uint64_t bits = F.band_bits[row >> 6];
	const uint32_t b = row & 63u;
	if (((bits >> b) & 1ull) == 0ull)
		return 0xffffffffu;
	const uint64_t below = bits & ((1ull << b) - 1ull);
#if defined(__HIP_DEVICE_COMPILE__)
	return F.band_rank[row >> 6] + (uint32_t)__popcll(below);
#else
	return F.band_rank[row >> 6] + (uint32_t)__builtin_popcountll(below);
#endif
}
// Where the 32 coefficients of a cell come from (FieldDev): the kernels are instantiated per mode, so that each
// has ONE load sequence (a runtime switch makes the compiler merge the variants into 32 separate 8-byte loads).
enum FieldMode : int
{
	kFieldClosed = 0,    // unreduced field, closed-form node indices into the reference layout: 16 x 16-byte pairs
	kFieldTable = 1,     // cell table (reduced / loaded fields): 32 indexed 8-byte loads
	kFieldCellMajor = 2, // cell-major copy: 256 contiguous bytes
	kFieldTileMajor = 3, // tile-major copy (unreduced fields): 16 x 16-byte pairs within one 5.9 KB tile
	kFieldXMajor = 4,    // V and X classes from the reference layout, Y and Z from the x-major copy: 16 pairs in 16 rows along x
};
DG_HD int field_mode(const FieldDev& F)
{
	if (F.xmajor)
		return kFieldXMajor;
	return F.tile_major ? kFieldTileMajor : (F.cell_major ? kFieldCellMajor : (F.cells ? kFieldTable : kFieldClosed));
}

// ---- x-major copy of the Y and Z edge classes (K3) -------------------------------------------------------------
// The reference numbers the Y edges y-fastest ((2j+h, k, i)) and the Z edges z-fastest ((2k+h, i, j)): lanes that sit
// side by side along x read them a plane apart.  K3's lanes do sit side by side along x (k_density_cells: a wave is a
// row block of 16 x 2 x 2 lattice points), so that the vertex and X-edge pairs of a wave -- x-fastest in the reference layout
// already -- come in 256-byte runs; the x-major copy gives the other two classes the same property at 1.0 x their
// memory: Y'[k][j][i][h] (k <= nz, j < ny, i <= nx) followed by Z'[k][j][i][h] (k < nz, j <= ny, i <= nx).
// Same values, same arithmetic: bit-identical results.
DG_HD uint64_t xmajor_y_pairs(const uint32_t res[3]) { return (uint64_t)(res[2] + 1) * res[1] * (res[0] + 1); }
DG_HD uint64_t xmajor_z_pairs(const uint32_t res[3]) { return (uint64_t)res[2] * (res[1] + 1) * (res[0] + 1); }
DG_HD uint64_t xmajor_doubles(const uint32_t res[3]) { return 2 * (xmajor_y_pairs(res) + xmajor_z_pairs(res)); }
// global node index (reference order) of the FIRST node (h = 0) of pair `pair` of the copy
DG_HD uint32_t xmajor_pair_node(uint64_t pair, const uint32_t res[3])
{
	const uint32_t nx = res[0], ny = res[1], nz = res[2];
	const uint32_t nv = (nx + 1) * (ny + 1) * (nz + 1);
	const uint32_t nex = nx * (ny + 1) * (nz + 1);
	const uint32_t ney = (nx + 1) * ny * (nz + 1);
	const uint64_t ny_pairs = xmajor_y_pairs(res);
	if (pair < ny_pairs)
	{
		const uint32_t i = (uint32_t)(pair % (nx + 1)), j = (uint32_t)((pair / (nx + 1)) % ny), k = (uint32_t)(pair / ((uint64_t)(nx + 1) * ny));
		return nv + 2 * nex + 2 * (ny * (nz + 1) * i + ny * k + j);
	}
	pair -= ny_pairs;
	const uint32_t i = (uint32_t)(pair % (nx + 1)), j = (uint32_t)((pair / (nx + 1)) % (ny + 1)), k = (uint32_t)(pair / ((uint64_t)(nx + 1) * (ny + 1)));
	return nv + 2 * nex + 2 * ney + 2 * (nz * (nx + 1) * j + nz * i + k);
}
// words per row of the per-cell "no value" bits: bit i & 63 of word (k ny + j) xmajor_flag_words(res) + (i >> 6)
DG_HD uint32_t xmajor_flag_words(const uint32_t res[3]) { return (res[0] + 63u) / 64u; }
DG_HD bool xmajor_cell_has_novalue(const FieldDev& F, uint32_t i, uint32_t j, uint32_t k)
{
	const uint64_t w = F.xmajor_flags[((size_t)k * F.res[1] + j) * xmajor_flag_words(F.res) + (i >> 6)];
	return ((w >> (i & 63u)) & 1ull) != 0ull;
}
template <int MODE>
DG_HD void fetch_cell(const FieldDev& F, uint32_t i, uint32_t j, uint32_t k, uint32_t row, double cf[32])
{
	if (MODE == kFieldCellMajor)
	{
		const double* r = F.cell_major + 32 * (size_t)row;
#if defined(__HIP__)
#pragma unroll
#endif
		for (int q = 0; q < 32; ++q)
			cf[q] = r[q];
	}
	else if (MODE == kFieldTable)
	{
		const uint32_t* r = F.cells + 32 * (size_t)row;
#if defined(__HIP__)
#pragma unroll
#endif
		for (int q = 0; q < 32; ++q)
			cf[q] = F.coeffs[r[q]];
	}
	else if (MODE == kFieldXMajor)
	{
		// rows along x: 4 vertex rows and 4 X-edge rows of the reference layout, 2 + 2 rows of the copy read at i and i + 1
		const uint32_t nx = F.res[0], ny = F.res[1], nz = F.res[2];
		const double* v = F.coeffs + ((size_t)(nx + 1) * (ny + 1) * k + (size_t)(nx + 1) * j + i);
		const size_t vj = nx + 1, vk = (size_t)(nx + 1) * (ny + 1);
		const double* ex = F.coeffs + (size_t)(nx + 1) * (ny + 1) * (nz + 1) + 2 * ((size_t)nx * (ny + 1) * k + (size_t)nx * j + i);
		const size_t xj = 2 * (size_t)nx, xk = 2 * (size_t)nx * (ny + 1);
		const double* ey = F.xmajor + 2 * (((size_t)k * ny + j) * (nx + 1) + i);
		const size_t yk = 2 * (size_t)ny * (nx + 1);
		const double* ez = F.xmajor + 2 * xmajor_y_pairs(F.res) + 2 * (((size_t)k * (ny + 1) + j) * (nx + 1) + i);
		const size_t zj = 2 * (size_t)(nx + 1);
		const double* src[16] = {v, v + vj, v + vk, v + vk + vj, ex, ex + xk, ex + xj, ex + xk + xj,
								 ey, ey + 2, ey + yk, ey + yk + 2, ez, ez + zj, ez + 2, ez + zj + 2};
#if defined(__HIP__)
#pragma unroll
#endif
		for (int m = 0; m < 16; ++m)
		{
			cf[2 * m] = src[m][0];
			cf[2 * m + 1] = src[m][1];
		}
	}
	else
	{
		uint32_t idx[32];
		const double* base = F.coeffs;
		if (MODE == kFieldTileMajor)
		{
			tile_node_slots(i & 3u, j & 3u, k & 3u, idx);
			base = F.tile_major + (size_t)kTmNodes * ((size_t)((k >> 2) * F.ntile[1] + (j >> 2)) * F.ntile[0] + (i >> 2));
		}
		else
			cell_node_indices(i, j, k, F.res, idx);
#if defined(__HIP__)
#pragma unroll
#endif
		for (int m = 0; m < 32; m += 2)
		{
			const double* pr = base + idx[m]; // adjacent pair: one 16-byte load
			cf[m] = pr[0];
			cf[m + 1] = pr[1];
		}
	}
}

// Per-query body of K2 = CubicLagrangeDiscreteGrid::interpolate(field, x, gradient*)
// (discregrid/src/cubic_lagrange_discrete_grid.cpp:977-1063).  The 32-term sum runs in j order
// (parity), the 32 coefficients are fetched as 16 adjacent pairs for unreduced fields.  Returns
// DBL_MAX ("no value") outside the domain, in removed cells, or if a coefficient is DBL_MAX;
// the gradient is zero in those cases.
template <bool GRAD, int MODE>
DG_HD double interpolate_point_mode(const FieldDev& F, const double x[3], double g[3])
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
	fetch_cell<MODE>(F, mi[0], mi[1], mi[2], cm, cf);
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
// interpolate_point_mode() in two halves, for kernels that fetch the coefficients themselves (k_interpolate_rows): the
// same statements in the same order (the K2 kernels keep the one-piece form above -- split, the compiler allocates 18
// more registers for them).  locate_query: the query's cell and its local coordinates, everything up to the fetch.
struct CellQuery
{
	bool valid;       // false: outside the domain or in a removed cell -> DBL_MAX, zero gradient
	uint32_t mi[3];   // cell multi-index
	uint32_t row;     // cell row (through cell_map for reduced fields)
	double c0[3], xi[3];
};
template <bool MAP = true> // MAP = false: the caller knows that the field has no cell map (unreduced): no look-up in the way
DG_HD CellQuery locate_query(const FieldDev& F, const double x[3])
{
	CellQuery q;
	q.valid = false;
	q.row = 0;
	for (int d = 0; d < 3; ++d)
	{
		q.mi[d] = 0;
		q.c0[d] = q.xi[d] = 0.0;
	}
	for (int d = 0; d < 3; ++d)
		if (!((F.dmin[d] <= x[d]) && (x[d] <= F.dmax[d]))) // AlignedBox::contains, inclusive (:981)
			return q;
	for (int d = 0; d < 3; ++d)
	{
		q.mi[d] = (uint32_t)((x[d] - F.dmin[d]) * F.inv_cell[d]); // :984
		if (q.mi[d] >= F.res[d])
			q.mi[d] = F.res[d] - 1;
	}
	const uint32_t ci = F.res[1] * F.res[0] * q.mi[2] + F.res[0] * q.mi[1] + q.mi[0];
	const uint32_t cm = (MAP && F.cell_map) ? F.cell_map[ci] : ci;
	if (cm == 0xffffffffu)
		return q;
	q.row = cm;
	for (int d = 0; d < 3; ++d)
	{
		const double lo = F.dmin[d] + (double)q.mi[d] * F.cell[d]; // subdomain(), discrete_grid.cpp:26-32
		const double hi = lo + F.cell[d];
		const double den = hi - lo; // :1000
		q.c0[d] = 2.0 / den;
		const double c1 = (hi + lo) / den;
		q.xi[d] = q.c0[d] * x[d] - c1;
	}
	q.valid = true;
	return q;
}
// The 32-term sum in j order over the cell's coefficients cf (parity); DBL_MAX if one of them is DBL_MAX.
template <bool GRAD>
DG_HD double evaluate_cell(const double cf[32], const double xi[3], const double c0[3], double g[3])
{
	const double NOVAL = 1.7976931348623157e308;
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
// runtime dispatch (host scalar API, emulator)
template <bool GRAD>
DG_HD double interpolate_point(const FieldDev& F, const double x[3], double g[3])
{
	switch (field_mode(F))
	{
	case kFieldXMajor: return interpolate_point_mode<GRAD, kFieldXMajor>(F, x, g);
	case kFieldTileMajor: return interpolate_point_mode<GRAD, kFieldTileMajor>(F, x, g);
	case kFieldCellMajor: return interpolate_point_mode<GRAD, kFieldCellMajor>(F, x, g);
	case kFieldTable: return interpolate_point_mode<GRAD, kFieldTable>(F, x, g);
	default: return interpolate_point_mode<GRAD, kFieldClosed>(F, x, g);
	}
}

// Morton key of the reference's zValue() / morton_lut() for a node position
// (cubic_lagrange_discrete_grid.cpp:583-601, src/data/z_sort_table.hpp:119-134).  The reference shifts its
// partial result by 48 and then by 24 bits, which pushes the contribution of the top byte out of the
// 64-bit word: only the low 16 bits of each biased coordinate end up in the key (48 significant bits).
// Reproduced as is: the node order of reduced fields, and with it the .cdm files, depend on it.
DG_HD uint64_t morton_spread3(uint32_t byte)
{
	uint64_t r = 0;
	for (int b = 0; b < 8; ++b)
		r |= (uint64_t)((byte >> b) & 1u) << (3 * b);
	return r;
}
DG_HD uint64_t reference_z_value(const double x[3], double inv_cell)
{
	uint32_t p[3];
	for (int d = 0; d < 3; ++d)
	{
		const int key = (x[d] >= 0.0) ? (int)(inv_cell * x[d]) : (int)(inv_cell * x[d]) - 1;
		p[d] = (uint32_t)((int64_t)key - (int64_t)(-2147483647)); // key - (INT_MIN + 1)
	}
	uint64_t lv[3];
	for (int s = 0; s < 3; ++s)
	{
		const int shift = 8 * s;
		lv[s] = morton_spread3((p[0] >> shift) & 0xFFu) | (morton_spread3((p[1] >> shift) & 0xFFu) << 1) |
				(morton_spread3((p[2] >> shift) & 0xFFu) << 2);
	}
	uint64_t a = lv[2];
	a = (a << 48) | lv[1];
	a = (a << 24) | lv[0];
	return a;
}
static const int kReferenceZBits = 48;

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
