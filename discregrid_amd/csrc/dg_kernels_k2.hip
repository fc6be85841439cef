// dg_kernels_k2.hip -- K2: batched CubicLagrangeDiscreteGrid::interpolate (cubic_lagrange_discrete_grid.cpp:977-1063): k_interpolate*,
// the device-side binning of unordered query batches (also used by K1p) and the optional copies of a field (cell-major, band-limited,
// tile-major).
// Compile with -ffp-contract=off (parity) -- see discregrid_amd/build.py.
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_select.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <stdint.h>
#include <algorithm>
#include "dg_kernels.h"
#include "dg_device.h"

namespace dg
{
namespace
{

// ------------------------------------------------------------------------------------------------
// K2: one thread per query point.  The 32-term sum must run in j order for parity, so the
// evaluation is per-lane; the 32 coefficients are fetched as 16 x 16-byte pairs (closed-form
// rows) with all loads issued before the first use.
// ------------------------------------------------------------------------------------------------
// XCD-aware block order for K2: the hardware deals consecutive workgroups to the 8 XCDs in turn, so eight
// neighbouring blocks of queries -- which, in tile order, gather from the same coefficient lines -- would
// each pull those lines into a different L2.  Chunks of kK2XcdChunk consecutive LOGICAL blocks (one
// chunk = what an XCD holds in flight) go to one XCD instead.  Returns false for padding blocks.
#ifndef DG_K2_XCD_CHUNK
#define DG_K2_XCD_CHUNK 256
#endif
static const uint32_t kK2XcdChunk = DG_K2_XCD_CHUNK;
__device__ __forceinline__ bool k2_logical_block(uint32_t block_idx, uint32_t n_blocks, uint32_t* blk)
{
	if (kK2XcdChunk == 0)
	{
		*blk = block_idx;
		return block_idx < n_blocks;
	}
	const uint32_t xcd = block_idx & 7u, within = block_idx >> 3;
	const uint32_t b = ((within / kK2XcdChunk) * 8u + xcd) * kK2XcdChunk + within % kK2XcdChunk;
	*blk = b;
	return b < n_blocks;
}
static uint32_t k2_grid(uint64_t n)
{
	const uint32_t blocks = (uint32_t)((n + 255) / 256);
	if (kK2XcdChunk == 0)
		return blocks;
	const uint32_t round = 8u * kK2XcdChunk;
	return (blocks + round - 1) / round * round;
}

// Occupancy matters more than anything else for this gather-latency bound kernel: left alone the compiler keeps all
// 32 coefficients AND all 32 shape functions in registers (132 / 154 VGPRs, 3 waves per SIMD); asked for more
// waves it forms the shape functions where they are consumed.
#ifndef DG_K2_WAVES
#define DG_K2_WAVES 3
#endif
template <bool GRAD, int MODE>
__global__ __launch_bounds__(256, DG_K2_WAVES) void k_interpolate(const FieldDev F, const double* __restrict__ xyz, uint64_t n,
													  double* __restrict__ phi_out, double* __restrict__ grad_out)
{
	uint32_t blk;
	if (!k2_logical_block(blockIdx.x, (uint32_t)((n + 255) / 256), &blk))
		return;
	const uint64_t gid = (uint64_t)blk * blockDim.x + threadIdx.x;
	if (gid >= n)
		return;
	const double x[3] = {xyz[3 * gid], xyz[3 * gid + 1], xyz[3 * gid + 2]};
	double g[3];
	phi_out[gid] = interpolate_point_mode<GRAD, MODE>(F, x, g);
	if (GRAD)
	{
		grad_out[3 * gid] = g[0];
		grad_out[3 * gid + 1] = g[1];
		grad_out[3 * gid + 2] = g[2];
	}
}

// ---- K2 query binning (dg_kernels.h: BinScratch) ---------------------------------------------------------
__device__ __forceinline__ uint32_t tile_of(const TileGrid& G, const double* __restrict__ xyz, uint64_t i)
{
	uint32_t t[3];
#pragma unroll
	for (int d = 0; d < 3; ++d)
	{
		const double u = (xyz[3 * i + d] - G.origin[d]) * G.inv_size[d];
		uint32_t c = u > 0.0 ? (uint32_t)(u < 4.0e9 ? u : 4.0e9) : 0u; // NaN -> 0
		t[d] = c < G.dims[d] ? c : G.dims[d] - 1;
	}
	return tile_key(G.dims, t);
}
// one block: how often do consecutive queries (among the first 4096) change tile?
__global__ __launch_bounds__(1024) void k_bin_probe(const TileGrid F, const double* __restrict__ xyz, uint64_t n, BinScratch S, uint32_t one_in)
{
	__shared__ uint32_t changes;
	if (threadIdx.x == 0)
		changes = 0;
	__syncthreads();
	const uint64_t m = n < 4096 ? n : 4096;
	uint32_t mine = 0;
	for (uint64_t i = threadIdx.x; i + 1 < m; i += blockDim.x)
		mine += tile_of(F, xyz, i) != tile_of(F, xyz, i + 1);
	atomicAdd(&changes, mine);
	__syncthreads();
	if (threadIdx.x == 0)
	{
		const uint32_t unordered = ((uint64_t)one_in * changes > m) ? 1u : 0u; // more than one change of tile in `one_in` steps
		S.flag[0] = unordered;
		*(volatile uint32_t*)S.flag_host = unordered; // prediction for the handle's next batch
	}
}
// sort keys: tile of every point, values: the point indices
__global__ __launch_bounds__(256) void k_bin_keys(const TileGrid F, const double* __restrict__ xyz, uint64_t n, BinScratch S)
{
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
	{
		S.keys[i] = tile_of(F, xyz, i);
		S.vals[i] = (uint32_t)i;
	}
}
template <bool GRAD, int MODE>
__global__ __launch_bounds__(256, DG_K2_WAVES) void k_interpolate_binned(const FieldDev F, const double* __restrict__ xyz, uint64_t n,
															 double* __restrict__ phi_out, double* __restrict__ grad_out, BinScratch S)
{
	uint32_t blk;
	if (!k2_logical_block(blockIdx.x, (uint32_t)((n + 255) / 256), &blk))
		return;
	uint64_t gid = (uint64_t)blk * blockDim.x + threadIdx.x;
	if (gid >= n)
		return;
	if (S.sort_launched != 0 && S.flag[0] != 0)
		gid = S.perm[gid];
	const double x[3] = {xyz[3 * gid], xyz[3 * gid + 1], xyz[3 * gid + 2]};
	double g[3];
	phi_out[gid] = interpolate_point_mode<GRAD, MODE>(F, x, g);
	if (GRAD)
	{
		grad_out[3 * gid] = g[0];
		grad_out[3 * gid + 1] = g[1];
		grad_out[3 * gid + 2] = g[2];
	}
}

// ---- K2 on the plain layout, staged through LDS (dg_kernels.h: TileBin) ----------------------------------------------------
// tile key of a query, from its CELL as locate_query computes it (same expression, same rounding); queries outside the domain
// (and NaNs) land in some tile and are answered DBL_MAX by locate_query like everywhere else
__device__ __forceinline__ uint32_t stage_key(const FieldDev& F, const uint32_t tdims[3], const uint32_t tlog[3], const double* __restrict__ xyz, uint64_t i)
{
	uint32_t t[3];
#pragma unroll
	for (int d = 0; d < 3; ++d)
	{
		const double u = (xyz[3 * i + d] - F.dmin[d]) * F.inv_cell[d];
		uint32_t c = u > 0.0 ? (u < 4.0e9 ? (uint32_t)u : 0xffffffffu) : 0u; // NaN -> 0
		c = c < F.res[d] ? c : F.res[d] - 1u;
		t[d] = c >> tlog[d];
	}
	return tile_key(tdims, t);
}
// keys of all queries; on the side (nothing here depends on them, and a launch of its own costs 5-16 us of an otherwise idle device):
// the tile tables cleared for k_tile_bounds, and -- block 0 -- the probe "is this batch ordered?" for the handle's next batch
__global__ __launch_bounds__(256) void k_tile_keys(const FieldDev F, const TileGrid probe_tiles, const double* __restrict__ xyz, uint64_t n, TileBin B)
{
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n)
		B.keys[i] = stage_key(F, B.tdims, B.tlog, xyz, i);
	for (uint64_t j = i; j < 2ull * B.key_space; j += (uint64_t)gridDim.x * blockDim.x)
		B.begin[j] = 0u; // (begin and end are one array)
	if (blockIdx.x == 0)
	{
		__shared__ uint32_t changes;
		if (threadIdx.x == 0)
			changes = 0;
		__syncthreads();
		const uint64_t m = n < 4096 ? n : 4096;
		uint32_t mine = 0;
		for (uint64_t k = threadIdx.x; k + 1 < m; k += blockDim.x)
			mine += tile_of(probe_tiles, xyz, k) != tile_of(probe_tiles, xyz, k + 1);
		atomicAdd(&changes, mine);
		__syncthreads();
		if (threadIdx.x == 0)
		{
			const uint32_t unordered = (4ull * changes > m) ? 1u : 0u; // (K2: more than one change of tile in four steps)
			B.flag[0] = unordered;
			*(volatile uint32_t*)B.flag_host = unordered;
		}
	}
}
// where the run of every tile begins and ends in the sorted keys (begin / end cleared before: tiles without a query keep 0, 0)
__global__ __launch_bounds__(256) void k_tile_bounds(uint64_t n, TileBin B)
{
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n)
		return;
	const uint32_t k = B.keys_out[i];
	if (i == 0 || B.keys_out[i - 1] != k)
		B.begin[k] = (uint32_t)i;
	if (i + 1 == n || B.keys_out[i + 1] != k)
		B.end[k] = (uint32_t)i + 1u;
}
// The work items (tile, chunk of kStageChunk queries) of the non-empty tiles, in key order: rows of 1024 consecutive tiles, one row
// per block, one tile per thread.  k_tile_row_items: items per row; k_tile_items: every block adds up the rows before its own (a few
// hundred numbers at most), scans its row and writes its items.  (One block for the whole table -- whichever way its 32 768 loads
// were arranged -- took 65-78 us on the one CU that ran it.)
__device__ __forceinline__ uint32_t block_scan_1024(uint32_t mine, uint32_t* wave_sum, uint32_t* total)
{
	uint32_t incl = mine;
#pragma unroll
	for (int off = 1; off < 64; off <<= 1)
	{
		const uint32_t up = (uint32_t)__shfl_up((int)incl, off);
		if ((int)(threadIdx.x & 63u) >= off)
			incl += up;
	}
	if ((threadIdx.x & 63u) == 63u)
		wave_sum[threadIdx.x >> 6] = incl;
	__syncthreads();
	uint32_t before = 0, all = 0;
#pragma unroll
	for (uint32_t wv = 0; wv < 16; ++wv)
	{
		const uint32_t v = wave_sum[wv];
		before += wv < (threadIdx.x >> 6) ? v : 0u;
		all += v;
	}
	*total = all;
	return before + incl - mine; // exclusive
}
__global__ __launch_bounds__(1024) void k_tile_row_items(TileBin B, uint32_t* __restrict__ row_items)
{
	__shared__ uint32_t wave_sum[16];
	const uint32_t t = blockIdx.x * 1024u + threadIdx.x;
	const uint32_t mine = t < B.key_space ? (B.end[t] - B.begin[t] + kStageChunk - 1u) / kStageChunk : 0u;
	uint32_t total;
	(void)block_scan_1024(mine, wave_sum, &total);
	if (threadIdx.x == 0)
		row_items[blockIdx.x] = total;
}
__global__ __launch_bounds__(1024) void k_tile_items(TileBin B, const uint32_t* __restrict__ row_items)
{
	__shared__ uint32_t wave_sum[16], wave_sum2[16];
	const uint32_t t = blockIdx.x * 1024u + threadIdx.x;
	const uint32_t b = t < B.key_space ? B.begin[t] : 0u, e = t < B.key_space ? B.end[t] : 0u;
	uint32_t rows_before = 0;
	for (uint32_t r = threadIdx.x; r < blockIdx.x; r += 1024u)
		rows_before += row_items[r];
	uint32_t base;
	{
		uint32_t tot;
		(void)block_scan_1024(rows_before, wave_sum2, &tot);
		base = tot;
	}
	uint32_t total;
	uint32_t it = base + block_scan_1024((e - b + kStageChunk - 1u) / kStageChunk, wave_sum, &total);
	uint32_t tc[3];
	tile_from_key(B.tdims, t, tc);
	const uint32_t tijk = tc[0] | (tc[1] << 10) | (tc[2] << 20);
	for (uint32_t q = b; q < e; q += kStageChunk)
	{
		StageItem item;
		item.key = t;
		item.q0 = q;
		item.q1 = e - q > kStageChunk ? q + kStageChunk : e;
		item.tijk = tijk;
		B.items[it++] = item;
	}
	if (blockIdx.x == gridDim.x - 1u && threadIdx.x == 0)
		*B.n_items = base + total;
}
// blockIdx -> logical work item with `chunk` consecutive items per XCD (the hardware deals consecutive workgroups to the 8 XCDs in
// turn; neighbouring tiles share the lines their rows straddle); grid = a multiple of 8 chunk
__device__ __forceinline__ uint32_t xcd_logical(uint32_t block_idx, uint32_t chunk)
{
	if (chunk == 0)
		return block_idx;
	const uint32_t xcd = block_idx & 7u, within = block_idx >> 3;
	return ((within / chunk) * 8u + xcd) * chunk + within % chunk;
}
// LDS image of a tile of TX x TY x TZ cells: V[k][j][i], X[k][j][2 i + h], Y[i][k][2 j + h], Z[j][i][2 k + h]
template <int LX, int LY, int LZ>
struct StageShape
{
	static constexpr uint32_t TX = 1u << LX, TY = 1u << LY, TZ = 1u << LZ;
	static constexpr uint32_t kV = 0, nV = (TX + 1) * (TY + 1) * (TZ + 1);
	static constexpr uint32_t kX = (nV + 1u) & ~1u, nX = 2 * TX * (TY + 1) * (TZ + 1);
	static constexpr uint32_t kY = kX + nX, nY = 2 * TY * (TX + 1) * (TZ + 1);
	static constexpr uint32_t kZ = kY + nY, nZ = 2 * TZ * (TX + 1) * (TY + 1);
	static constexpr uint32_t kDoubles = kZ + nZ;
	// threads per block: about one per query of a tile at 0.6 queries per cell (config 5), at least a wave
	static constexpr uint32_t NT = TX * TY * TZ >= 512 ? 256u : (TX * TY * TZ >= 256 ? 128u : 64u);
};
// one class of rows of the tile image: ROWS_A x ROWS_B rows (a fast) of WIDTH doubles in LDS, L lanes per row with 16 bytes each.
// Every load is UNCONDITIONAL (rows and pieces a clipped tile does not have read a clamped, valid address and land in slots no cell of
// the tile refers to): a load behind a branch costs the compiler its count of the loads in flight, and every wait becomes "all of them".
template <uint32_t NT, uint32_t ROWS_A, uint32_t ROWS_B, uint32_t WIDTH>
struct StageClass
{
	static constexpr uint32_t L = WIDTH > 8 ? 8u : (WIDTH > 4 ? 4u : 2u); // lanes per row (two doubles each)
	static constexpr uint32_t kRowsPerStep = NT / L;
	static constexpr uint32_t kRows = ROWS_A * ROWS_B;
	static constexpr uint32_t kSteps = (kRows + kRowsPerStep - 1) / kRowsPerStep;
	double v0[kSteps], v1[kSteps];
	// first: index of the tile's first element of the class; last: the last index a 16-byte load may start at (n_coeffs - 2)
	__device__ __forceinline__ void load(const double* __restrict__ coeffs, uint64_t first, uint64_t last, uint32_t sa, uint32_t sb, uint32_t na, uint32_t nb)
	{
		const uint32_t piece = threadIdx.x % L, rslot = threadIdx.x / L;
#pragma unroll
		for (uint32_t s = 0; s < kSteps; ++s)
		{
			const uint32_t r = s * kRowsPerStep + rslot;
			uint32_t b = r / ROWS_A, a = r - ROWS_A * b;
			a = a < na ? a : na - 1u;
			b = b < nb ? b : nb - 1u;
			uint64_t e = first + ((uint64_t)a * sa + (uint64_t)b * sb + 2u * piece);
			e = e < last ? e : last;
#if DG_K2_ABLATE & 2
			v0[s] = v1[s] = 0.0;
			(void)coeffs;
			(void)e;
#else
			v0[s] = coeffs[e];
			v1[s] = coeffs[e + 1];
#endif
		}
	}
	__device__ __forceinline__ void store(double* __restrict__ lds) const
	{
		const uint32_t piece = threadIdx.x % L, rslot = threadIdx.x / L;
#pragma unroll
		for (uint32_t s = 0; s < kSteps; ++s)
		{
			const uint32_t r = s * kRowsPerStep + rslot;
			if (r < kRows)
			{
				double* dst = lds + r * WIDTH + 2u * piece;
				if (2u * piece < WIDTH)
					dst[0] = v0[s];
				if (2u * piece + 1u < WIDTH)
					dst[1] = v1[s];
			}
		}
	}
};
// One block = one work item (a tile's queries, at most kStageChunk of them): the item, the first round's query indices, then the
// query points and the tile's rows together, one barrier, then the evaluation from LDS, round by round.  (Also built and measured on the same boxes, round 6: PERSISTENT blocks that run the chains of the
// next items -- descriptor three items ahead, query indices two, points and rows one -- under the evaluation of the current one.
// With the rows prefetched into registers the loop needs 200-240 VGPRs, two waves per SIMD: 0.73-0.75 ms per 10 M queries; with only
// the query chains prefetched, three waves: 0.82 ms; this form: 0.55-0.65 ms.  What the pipelining gains in hidden latency it loses in
// occupancy: the evaluation is f64 dependency chains and LDS round trips that want four waves per SIMD.  The shapes 8 x 8 x 4 and
// 8 x 4 x 4 -- half and a quarter of the LDS image, twice and four times the blocks -- measured 1.00 and 0.84 of this one's rate.)
// (measurement builds only, tools/gpu_k2_ablate.sh: -DDG_K2_ABLATE=<bits> removes one phase at a time -- 1: the query points come from
// the tile's own cells instead of xyz[perm[q]]; 2: no row loads (the LDS image holds zeros); 4: no evaluation; 8: results stored in tile
// order instead of scattered in query order.  Results are then wrong; the product is built without it.)
#ifndef DG_K2_ABLATE
#define DG_K2_ABLATE 0
#endif
template <bool GRAD, int LX, int LY, int LZ>
__global__ __launch_bounds__((StageShape<LX, LY, LZ>::NT), GRAD ? 3 : 4)
void k_interpolate_tiles(const FieldDev F, const double* __restrict__ xyz, double* __restrict__ phi_out, double* __restrict__ grad_out, TileBin B,
						 uint32_t chunk, double* __restrict__ packed)
{
	typedef StageShape<LX, LY, LZ> S;
	constexpr uint32_t TX = S::TX, TY = S::TY, TZ = S::TZ, NT = S::NT;
	__shared__ double tile[S::kDoubles];
	const uint32_t w = xcd_logical(blockIdx.x, chunk);
	if (w >= *B.n_items)
		return;
	const StageItem it = B.items[w];
	const uint32_t nx = F.res[0], ny = F.res[1], nz = F.res[2];
	const uint64_t nv = (uint64_t)(nx + 1) * (ny + 1) * (nz + 1), nex = (uint64_t)nx * (ny + 1) * (nz + 1), ney = (uint64_t)(nx + 1) * ny * (nz + 1);
	const uint64_t last = nv + 2 * (nex + ney) + 2 * (uint64_t)(nx + 1) * (ny + 1) * nz - 2;
	const uint32_t i0 = (it.tijk & 1023u) << LX, j0 = ((it.tijk >> 10) & 1023u) << LY, k0 = (it.tijk >> 20) << LZ;
	// the first round's query, fetched under the staging (lanes without one issue nothing: a clamped, unconditional load here was
	// measured -- every lane of the block fetching a point for the second round that one wave in four has)
	uint32_t q = it.q0 + threadIdx.x;
	uint32_t gid = 0;
	double x[3] = {0.0, 0.0, 0.0};
	if (q < it.q1)
		gid = B.perm[q];
	{
		StageClass<NT, TY + 1, TZ + 1, TX + 1> cv;  // rows a = j, b = k
		StageClass<NT, TY + 1, TZ + 1, 2 * TX> cx;  // a = j, b = k
		StageClass<NT, TZ + 1, TX + 1, 2 * TY> cy;  // a = k, b = i
		StageClass<NT, TX + 1, TY + 1, 2 * TZ> cz;  // a = i, b = j
		// cells of this tile per axis (tiles at the far faces of a resolution that is no multiple of the tile are smaller)
		const uint32_t ex = nx - i0 < TX ? nx - i0 : TX, ey = ny - j0 < TY ? ny - j0 : TY, ez = nz - k0 < TZ ? nz - k0 : TZ;
		cv.load(F.coeffs, (uint64_t)(nx + 1) * (ny + 1) * k0 + (uint64_t)(nx + 1) * j0 + i0, last, nx + 1, (nx + 1) * (ny + 1), ey + 1, ez + 1);
		cx.load(F.coeffs, nv + 2 * ((uint64_t)nx * (ny + 1) * k0 + (uint64_t)nx * j0 + i0), last, 2 * nx, 2 * nx * (ny + 1), ey + 1, ez + 1);
		cy.load(F.coeffs, nv + 2 * nex + 2 * ((uint64_t)ny * (nz + 1) * i0 + (uint64_t)ny * k0 + j0), last, 2 * ny, 2 * ny * (nz + 1), ez + 1, ex + 1);
		cz.load(F.coeffs, nv + 2 * nex + 2 * ney + 2 * ((uint64_t)nz * (nx + 1) * j0 + (uint64_t)nz * i0 + k0), last, 2 * nz, 2 * nz * (nx + 1), ex + 1, ey + 1);
		if (q < it.q1)
		{
#if DG_K2_ABLATE & 1
			x[0] = F.dmin[0] + ((double)i0 + 0.37 + (double)(q % TX)) * F.cell[0];
			x[1] = F.dmin[1] + ((double)j0 + 0.21 + (double)((q / TX) % TY)) * F.cell[1];
			x[2] = F.dmin[2] + ((double)k0 + 0.63 + (double)((q / (TX * TY)) % TZ)) * F.cell[2];
#else
			x[0] = xyz[3 * (uint64_t)gid];
			x[1] = xyz[3 * (uint64_t)gid + 1];
			x[2] = xyz[3 * (uint64_t)gid + 2];
#endif
		}
		cv.store(tile + S::kV);
		cx.store(tile + S::kX);
		cy.store(tile + S::kY);
		cz.store(tile + S::kZ);
	}
	__syncthreads();
	for (; q < it.q1; q += NT)
	{
		const CellQuery cq = locate_query<false>(F, x);
		double cf[32];
		double g[3] = {0.0, 0.0, 0.0};
		double phi = 1.7976931348623157e308;
		if (cq.valid)
		{
			// the query's cell is a cell of this tile: the key the queries were sorted by is (mi >> log2 tile), the very words
			// locate_query computes (clamped all the same: an LDS address must not depend on the caller leaving xyz alone)
			uint32_t li = cq.mi[0] - i0, lj = cq.mi[1] - j0, lk = cq.mi[2] - k0;
			li = li < TX ? li : TX - 1u;
			lj = lj < TY ? lj : TY - 1u;
			lk = lk < TZ ? lk : TZ - 1u;
			constexpr uint32_t vj = TX + 1, vk = (TX + 1) * (TY + 1);
			const double* V = tile + S::kV + lk * vk + lj * vj + li;
			cf[0] = V[0]; cf[1] = V[1]; cf[2] = V[vj]; cf[3] = V[vj + 1];
			cf[4] = V[vk]; cf[5] = V[vk + 1]; cf[6] = V[vk + vj]; cf[7] = V[vk + vj + 1];
			constexpr uint32_t xj = 2 * TX, xk = 2 * TX * (TY + 1);
			const double* X = tile + S::kX + lk * xk + lj * xj + 2u * li; // x edges at (j, k), (j, k + 1), (j + 1, k), (j + 1, k + 1)
			cf[8] = X[0]; cf[9] = X[1]; cf[10] = X[xk]; cf[11] = X[xk + 1];
			cf[12] = X[xj]; cf[13] = X[xj + 1]; cf[14] = X[xk + xj]; cf[15] = X[xk + xj + 1];
			constexpr uint32_t yk = 2 * TY, yi = 2 * TY * (TZ + 1);
			const double* Y = tile + S::kY + li * yi + lk * yk + 2u * lj; // y edges at (i, k), (i + 1, k), (i, k + 1), (i + 1, k + 1)
			cf[16] = Y[0]; cf[17] = Y[1]; cf[18] = Y[yi]; cf[19] = Y[yi + 1];
			cf[20] = Y[yk]; cf[21] = Y[yk + 1]; cf[22] = Y[yi + yk]; cf[23] = Y[yi + yk + 1];
			constexpr uint32_t zi = 2 * TZ, zj = 2 * TZ * (TX + 1);
			const double* Z = tile + S::kZ + lj * zj + li * zi + 2u * lk; // z edges at (j, i), (j + 1, i), (j, i + 1), (j + 1, i + 1)
			cf[24] = Z[0]; cf[25] = Z[1]; cf[26] = Z[zj]; cf[27] = Z[zj + 1];
			cf[28] = Z[zi]; cf[29] = Z[zi + 1]; cf[30] = Z[zj + zi]; cf[31] = Z[zj + zi + 1];
#if DG_K2_ABLATE & 4
			phi = cq.xi[0] + cq.xi[1] + cq.xi[2];
			for (int m = 0; m < 32; ++m)
				phi += cf[m];
#else
			phi = evaluate_cell<GRAD>(cf, cq.xi, cq.c0, g);
#endif
		}
#if DG_K2_ABLATE & 8
		gid = q; // (tile order: coalesced)
#endif
		if (GRAD && packed != nullptr)
		{
			// value and gradient as ONE aligned 32-byte store into the query's slot of a scratch array (k_unpack_results splits it into the
			// caller's two arrays): scattered in query order a result is then one whole sector instead of an 8-byte and a 24-byte piece
			// of three, each of which the memory system has to merge into a sector it holds only in part
			double4 r4;
			r4.x = phi;
			r4.y = g[0];
			r4.z = g[1];
			r4.w = g[2];
			*reinterpret_cast<double4*>(packed + 4 * (uint64_t)gid) = r4;
		}
		else
		{
			phi_out[gid] = phi;
			if (GRAD)
			{
				grad_out[3 * (uint64_t)gid] = g[0];
				grad_out[3 * (uint64_t)gid + 1] = g[1];
				grad_out[3 * (uint64_t)gid + 2] = g[2];
			}
		}
		if (q + NT < it.q1)
		{
			gid = B.perm[q + NT];
#if !(DG_K2_ABLATE & 1)
			x[0] = xyz[3 * (uint64_t)gid];
			x[1] = xyz[3 * (uint64_t)gid + 1];
			x[2] = xyz[3 * (uint64_t)gid + 2];
#else
			x[0] += 0.03 * F.cell[0];
#endif
		}
	}
}

// packed (phi, gx, gy, gz) per query -> the caller's phi[n] and grad[3 n]: streaming, everything coalesced
__global__ __launch_bounds__(256) void k_unpack_results(const double* __restrict__ packed, uint64_t n, double* __restrict__ phi_out, double* __restrict__ grad_out)
{
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n)
		return;
	const double4 r4 = *reinterpret_cast<const double4*>(packed + 4 * i);
	phi_out[i] = r4.x;
	grad_out[3 * i] = r4.y;
	grad_out[3 * i + 1] = r4.z;
	grad_out[3 * i + 2] = r4.w;
}

// K2 over a field with a CELL-MAJOR copy, queries in ANY order, no binning: one wave = 64 queries.  A query's 32
// coefficients are one contiguous 256-byte row of the copy; left to itself every lane would read its own row with
// sixteen 16-byte loads, 64 different lines per instruction.  Here the wave fetches the 64 rows TOGETHER -- sixteen
// loads, each covering four whole rows (lane = (row in the group, 16-byte piece)), i.e. four fully used 256-byte
// segments per instruction --, hands them to their owners through LDS (rows padded to 33 doubles: the owners' reads are
// conflict-free) and every lane then evaluates its own query exactly as interpolate_point_mode does (locate_query /
// evaluate_cell: the same statements).  HBM moves 256 B + 24 B + 8 B per query whatever the order of the queries.
static const int kRowStride = 33; // doubles per staged row
template <bool GRAD>
__global__ __launch_bounds__(64) void k_interpolate_rows(const FieldDev F, const double* __restrict__ xyz, uint64_t n,
														   double* __restrict__ phi_out, double* __restrict__ grad_out)
{
	__shared__ double rows[64 * kRowStride];
	const int lane = (int)threadIdx.x;
	const int sub = lane & 15, grp = lane >> 4;
	for (uint64_t base = (uint64_t)blockIdx.x * 64u; base < n; base += (uint64_t)gridDim.x * 64u)
	{
		const uint64_t gid = base + (uint64_t)lane;
		const bool have = gid < n;
		double x[3] = {0.0, 0.0, 0.0};
		if (have)
		{
			x[0] = xyz[3 * gid];
			x[1] = xyz[3 * gid + 1];
			x[2] = xyz[3 * gid + 2];
		}
		CellQuery q = locate_query(F, x);
		q.valid = q.valid && have;
		const uint32_t my_row = q.valid ? q.row : 0u; // (row 0 exists: a field has at least one cell row)
		__syncthreads(); // the previous round's rows have been read
#pragma unroll
		for (int k = 0; k < 16; ++k)
		{
			const int owner = 4 * k + grp;
			const uint32_t r = (uint32_t)__shfl((int)my_row, owner);
			const double2 v = *reinterpret_cast<const double2*>(F.cell_major + 32 * (size_t)r + 2 * sub);
			rows[owner * kRowStride + 2 * sub] = v.x;
			rows[owner * kRowStride + 2 * sub + 1] = v.y;
		}
		__syncthreads();
		double cf[32];
#pragma unroll
		for (int j = 0; j < 32; ++j)
			cf[j] = rows[lane * kRowStride + j];
		double g[3] = {0.0, 0.0, 0.0};
		double phi = 1.7976931348623157e308;
		if (q.valid)
			phi = evaluate_cell<GRAD>(cf, q.xi, q.c0, g);
		if (have)
		{
			phi_out[gid] = phi;
			if (GRAD)
			{
				grad_out[3 * gid] = g[0];
				grad_out[3 * gid + 1] = g[1];
				grad_out[3 * gid + 2] = g[2];
			}
		}
	}
}

// K2 over a field with a BAND-LIMITED cell-major copy (FieldDev::band_rows / band_map): k_interpolate_rows for the queries
// whose cell has a row in the copy -- the wave fetches those rows together, four whole rows per load instruction, owners
// read them from LDS --, and in the same launch the plain gather (closed-form indices or the cell table) for the lanes
// whose cell has none.  A load group whose owner has no row (or no query) is switched off, so a batch that lives in the
// band moves 256 B per query and a batch far from it moves what the plain kernel moves plus 4 B of map.  Same
// locate_query / evaluate_cell statements as every other K2 path: same bits.
// (one lane's query of a round: located, and looked up in the band copy)
struct BandQuery
{
	CellQuery q;
	uint32_t row; // in the band copy, 0xffffffff: none (or no query)
	bool have;
};
__device__ __forceinline__ BandQuery band_locate(const FieldDev& F, const double* __restrict__ xyz, uint64_t gid, uint64_t n)
{
	BandQuery b;
	b.have = gid < n;
	double x[3] = {0.0, 0.0, 0.0};
	if (b.have)
	{
		x[0] = xyz[3 * gid];
		x[1] = xyz[3 * gid + 1];
		x[2] = xyz[3 * gid + 2];
	}
	b.q = locate_query(F, x);
	b.q.valid = b.q.valid && b.have;
	b.row = b.q.valid ? band_row_of(F, b.q.row) : 0xffffffffu;
	return b;
}
template <bool GRAD, int MODE>
__global__ __launch_bounds__(64) void k_interpolate_band(const FieldDev F, const double* __restrict__ xyz, uint64_t n,
														   double* __restrict__ phi_out, double* __restrict__ grad_out)
{
	__shared__ double rows[64 * kRowStride];
	const int lane = (int)threadIdx.x;
	const int sub = lane & 15, grp = lane >> 4;
	const uint64_t stride = (uint64_t)gridDim.x * 64u;
	uint64_t base = (uint64_t)blockIdx.x * 64u;
	if (base >= n)
		return;
	// The look-up (query -> cell -> bit / rank words -> row) is a chain of two dependent memory round trips in front of the
	// row fetch; the wave therefore locates the queries of its NEXT round while the rows of the current one are in flight.
	BandQuery cur = band_locate(F, xyz, base + (uint64_t)lane, n);
	for (; base < n; base += stride)
	{
		const uint64_t gid = base + (uint64_t)lane;
		const bool mapped = cur.row != 0xffffffffu;
		__syncthreads(); // the previous round's rows have been read
#pragma unroll
		for (int k = 0; k < 16; ++k)
		{
			const int owner = 4 * k + grp;
			// (owners without a row read row 0 -- it exists, and it is the same cached line for all of them --: unconditional
			// loads let the sixteen fetches be in flight together; behind a branch each they ran one after the other, 11 instead
			// of 18 Gq/s on a batch that lives in the band)
			uint32_t r = (uint32_t)__shfl((int)cur.row, owner);
			r = r != 0xffffffffu ? r : 0u;
			const double2 v = *reinterpret_cast<const double2*>(F.band_rows + 32 * (size_t)r + 2 * sub);
			rows[owner * kRowStride + 2 * sub] = v.x;
			rows[owner * kRowStride + 2 * sub + 1] = v.y;
		}
		BandQuery nxt;
		nxt.have = false;
		nxt.row = 0xffffffffu;
		nxt.q.valid = false;
		if (base + stride < n) // (wave-uniform)
			nxt = band_locate(F, xyz, base + stride + (uint64_t)lane, n);
		__syncthreads();
		double cf[32];
		if (mapped)
		{
#pragma unroll
			for (int j = 0; j < 32; ++j)
				cf[j] = rows[lane * kRowStride + j];
		}
		else if (cur.q.valid)
			fetch_cell<MODE>(F, cur.q.mi[0], cur.q.mi[1], cur.q.mi[2], cur.q.row, cf);
		double g[3] = {0.0, 0.0, 0.0};
		double phi = 1.7976931348623157e308;
		if (cur.q.valid)
			phi = evaluate_cell<GRAD>(cf, cur.q.xi, cur.q.c0, g);
		if (cur.have)
		{
			phi_out[gid] = phi;
			if (GRAD)
			{
				grad_out[3 * gid] = g[0];
				grad_out[3 * gid + 1] = g[1];
				grad_out[3 * gid + 2] = g[2];
			}
		}
		cur = nxt;
	}
}
// Routing probe of the band path: of (up to) 1024 queries spread evenly over the batch, how many have a cell and how many of
// those have a row in the band copy?  One block; the two counts go to pinned host words -- the prediction the HOST routes the
// field's NEXT large batch with (a batch that mostly misses the band is faster through the sorted gather of the binned path
// than through this kernel's unsorted one: 8.3 against 5.5 Gq/s with 57 % of the queries mapped).
__global__ __launch_bounds__(1024) void k_band_probe(const FieldDev F, const double* __restrict__ xyz, uint64_t n, uint32_t* host_counts)
{
	// ONE sample per thread (1024 of them): the look-up is a chain of dependent loads, and a thread that walked several chains one
	// after the other made the probe cost 80 us in front of a 550 us launch
	__shared__ uint32_t valid, mapped;
	if (threadIdx.x == 0)
		valid = mapped = 0;
	__syncthreads();
	const uint64_t m = n < 1024 ? n : 1024;
	const uint64_t step = n / m;
	if (threadIdx.x < m)
	{
		const BandQuery b = band_locate(F, xyz, (uint64_t)threadIdx.x * step, n);
		const unsigned long long v = __ballot(b.q.valid), h = __ballot(b.row != 0xffffffffu);
		if ((threadIdx.x & 63u) == 0)
		{
			atomicAdd(&valid, (uint32_t)__popcll(v));
			atomicAdd(&mapped, (uint32_t)__popcll(h));
		}
	}
	__syncthreads();
	if (threadIdx.x == 0)
	{
		*(volatile uint32_t*)(host_counts + 1) = mapped;
		*(volatile uint32_t*)host_counts = valid;
	}
}
// the band copy's builders: (1) per cell row, does any value the cell's 32 coefficients span reach into [lo, hi]?
// (min <= hi and max >= lo: a cell that straddles a thin band counts); (2) after a scan of the flags: rows and map
__device__ __forceinline__ void band_cell_indices(const FieldDev& F, uint64_t row, uint32_t idx[32])
{
	if (F.cells)
	{
#pragma unroll
		for (int j = 0; j < 32; ++j)
			idx[j] = F.cells[32 * row + j];
	}
	else
	{
		const uint32_t n01 = F.res[0] * F.res[1];
		const uint32_t k = (uint32_t)(row / n01), r = (uint32_t)(row % n01);
		cell_node_indices(r % F.res[0], r / F.res[0], k, F.res, idx);
	}
}
__global__ __launch_bounds__(256) void k_band_flags(const FieldDev F, uint64_t n_rows, double lo, double hi, uint32_t* __restrict__ flag)
{
	const uint64_t row = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (row >= n_rows)
		return;
	uint32_t idx[32];
	band_cell_indices(F, row, idx);
	double mn = F.coeffs[idx[0]], mx = mn;
#pragma unroll
	for (int j = 1; j < 32; ++j)
	{
		const double v = F.coeffs[idx[j]];
		mn = v < mn ? v : mn;
		mx = v > mx ? v : mx;
	}
	flag[row] = (mn <= hi && mx >= lo) ? 1u : 0u;
}
__global__ __launch_bounds__(256) void k_band_expand(const FieldDev F, uint64_t n_rows, const uint32_t* __restrict__ flag,
													   const uint32_t* __restrict__ pos, uint64_t* __restrict__ bits, uint32_t* __restrict__ rank,
													   double* __restrict__ out)
{
	const uint64_t row = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; // (64 consecutive rows per wave: one word of bits)
	const bool keep = row < n_rows && flag[row] != 0u;
	const unsigned long long word = __ballot(keep);
	if ((threadIdx.x & 63u) == 0u && row < n_rows)
	{
		bits[row >> 6] = word;
		rank[row >> 6] = pos[row]; // rows of the copy before this word (exclusive scan of the flags)
	}
	if (!keep)
		return;
	const uint32_t r = pos[row];
	uint32_t idx[32];
	band_cell_indices(F, row, idx);
	double* o = out + 32 * (size_t)r;
#pragma unroll
	for (int j = 0; j < 32; ++j)
		o[j] = F.coeffs[idx[j]];
}

// Builds the cell-major copy of a field (FieldDev::cell_major): one thread per cell row.
__global__ __launch_bounds__(256) void k_expand_cells(const FieldDev F, uint64_t n_rows, double* __restrict__ out)
{
	const uint64_t row = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (row >= n_rows)
		return;
	uint32_t idx[32];
	if (F.cells)
	{
#pragma unroll
		for (int j = 0; j < 32; ++j)
			idx[j] = F.cells[32 * row + j];
	}
	else
	{
		const uint32_t n01 = F.res[0] * F.res[1];
		const uint32_t k = (uint32_t)(row / n01), r = (uint32_t)(row % n01);
		cell_node_indices(r % F.res[0], r / F.res[0], k, F.res, idx);
	}
	double* o = out + 32 * row;
#pragma unroll
	for (int j = 0; j < 32; ++j)
		o[j] = F.coeffs[idx[j]];
}

// Builds the tile-major copy of an unreduced field (dg_lattice.h): one thread per slot, one block row per tile;
// reads are gathers from the reference layout (each node is read by at most 8 tiles), writes are contiguous.
__global__ __launch_bounds__(256) void k_expand_tiles(const FieldDev F, uint64_t n_tiles, double* __restrict__ out)
{
	const uint64_t total = n_tiles * kTmNodes;
	for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (uint64_t)gridDim.x * blockDim.x)
	{
		const uint64_t tile = e / kTmNodes;
		const uint32_t slot = (uint32_t)(e - tile * kTmNodes);
		const uint32_t ti = (uint32_t)(tile % F.ntile[0]);
		const uint32_t tj = (uint32_t)((tile / F.ntile[0]) % F.ntile[1]);
		const uint32_t tk = (uint32_t)(tile / ((uint64_t)F.ntile[0] * F.ntile[1]));
		const uint32_t node = tile_slot_node(slot, ti, tj, tk, F.res);
		out[e] = node == 0xffffffffu ? 0.0 : F.coeffs[node];
	}
}
// second pass: the "no value" flags of the 64 cells of every tile (dg_lattice.h: kTmFlags); one wave per tile
__global__ __launch_bounds__(256) void k_tile_flags(uint64_t n_tiles, double* __restrict__ tiles)
{
	const uint64_t tile = (uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6);
	if (tile >= n_tiles)
		return;
	const uint32_t c = threadIdx.x & 63u;
	uint32_t slots[32];
	tile_node_slots(c & 3u, (c >> 2) & 3u, c >> 4, slots);
	const double* t = tiles + tile * kTmNodes;
	bool nov = false;
#pragma unroll
	for (int q = 0; q < 32; ++q)
		nov = nov || (t[slots[q]] == 1.7976931348623157e308);
	const unsigned long long flags = __ballot(nov);
	if (c == 0)
		*(unsigned long long*)(tiles + tile * kTmNodes + kTmFlags) = flags;
}

} // namespace

hipError_t launch_expand_tiles(const FieldDev& f, uint64_t n_tiles, double* d_out, hipStream_t stream)
{
	if (n_tiles == 0)
		return hipSuccess;
	const uint64_t total = n_tiles * kTmNodes;
	const uint32_t blocks = (uint32_t)std::min<uint64_t>((total + 255) / 256, 256ull * 64ull);
	hipLaunchKernelGGL(k_expand_tiles, dim3(blocks), dim3(256), 0, stream, f, n_tiles, d_out);
	hipLaunchKernelGGL(k_tile_flags, dim3((uint32_t)((n_tiles + 3) / 4)), dim3(256), 0, stream, n_tiles, d_out);
	return hipGetLastError();
}

hipError_t launch_interpolate_band(const FieldDev& f, const double* d_xyz, uint64_t n, double* d_phi, double* d_grad, hipStream_t stream)
{
	if (n == 0)
		return hipSuccess;
	const uint32_t blocks = (uint32_t)std::min<uint64_t>((n + 63) / 64, 256ull * 64ull);
	const bool table = f.cells != nullptr;
#define DG_K2_BAND(G, M) hipLaunchKernelGGL((k_interpolate_band<G, M>), dim3(blocks), dim3(64), 0, stream, f, d_xyz, n, d_phi, d_grad)
	if (d_grad && table) DG_K2_BAND(true, kFieldTable);
	else if (d_grad) DG_K2_BAND(true, kFieldClosed);
	else if (table) DG_K2_BAND(false, kFieldTable);
	else DG_K2_BAND(false, kFieldClosed);
#undef DG_K2_BAND
	return hipGetLastError();
}
hipError_t launch_band_probe(const FieldDev& f, const double* d_xyz, uint64_t n, uint32_t* host_counts, hipStream_t stream)
{
	if (n == 0 || !host_counts)
		return hipSuccess;
	hipLaunchKernelGGL(k_band_probe, dim3(1), dim3(1024), 0, stream, f, d_xyz, n, host_counts);
	return hipGetLastError();
}
hipError_t launch_band_flags(const FieldDev& f, uint64_t n_rows, double lo, double hi, uint32_t* d_flag, hipStream_t stream)
{
	if (n_rows == 0)
		return hipSuccess;
	hipLaunchKernelGGL(k_band_flags, dim3((uint32_t)((n_rows + 255) / 256)), dim3(256), 0, stream, f, n_rows, lo, hi, d_flag);
	return hipGetLastError();
}
// exclusive scan of the flags (rocPRIM); tmp: scratch of *tmp_bytes (query with d_tmp == nullptr)
hipError_t band_scan(const uint32_t* d_flag, uint32_t* d_pos, uint64_t n_rows, void* d_tmp, size_t* tmp_bytes, hipStream_t stream)
{
	return rocprim::exclusive_scan(d_tmp, *tmp_bytes, d_flag, d_pos, 0u, (size_t)n_rows, rocprim::plus<uint32_t>(), stream);
}
hipError_t launch_band_expand(const FieldDev& f, uint64_t n_rows, const uint32_t* d_flag, const uint32_t* d_pos, uint64_t* d_bits, uint32_t* d_rank,
							  double* d_rows, hipStream_t stream)
{
	if (n_rows == 0)
		return hipSuccess;
	hipLaunchKernelGGL(k_band_expand, dim3((uint32_t)((n_rows + 255) / 256)), dim3(256), 0, stream, f, n_rows, d_flag, d_pos, d_bits, d_rank, d_rows);
	return hipGetLastError();
}

hipError_t launch_expand_cells(const FieldDev& f, uint64_t n_rows, double* d_out, hipStream_t stream)
{
	if (n_rows == 0)
		return hipSuccess;
	hipLaunchKernelGGL(k_expand_cells, dim3((uint32_t)((n_rows + 255) / 256)), dim3(256), 0, stream, f, n_rows, d_out);
	return hipGetLastError();
}

// the binning passes shared by K2 and K1p: S.flag / S.perm describe the order to process the points in
static uint32_t key_bits(uint32_t n_tiles)
{
	uint32_t bits = 1;
	while (bits < 32 && (1u << bits) < n_tiles)
		++bits;
	return bits;
}
size_t bin_sort_tmp_bytes(uint64_t n, uint32_t n_tiles)
{
	size_t bytes = 0;
	(void)rocprim::radix_sort_pairs(nullptr, bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const uint32_t*)nullptr,
									(uint32_t*)nullptr, (size_t)n, 0u, 32u, (hipStream_t) nullptr); // (all 32 bits: an upper bound for any key width)
	return bytes;
}
// the binning passes shared by K2 and K1p: probe (always), and -- if the host predicts an unordered batch
// (S.sort_launched) -- tile keys + radix sort, which leaves the processing order in S.perm
hipError_t launch_binning(const TileGrid& probe_tiles, const TileGrid& tiles, const double* d_xyz, uint64_t n, const BinScratch& S, uint32_t one_in, hipStream_t stream)
{
	hipLaunchKernelGGL(k_bin_probe, dim3(1), dim3(1024), 0, stream, probe_tiles, d_xyz, n, S, one_in);
	if (S.sort_launched == 0)
		return hipGetLastError();
	const uint32_t wide = (uint32_t)std::min<uint64_t>((n + 255) / 256, 256ull * 64ull);
	hipLaunchKernelGGL(k_bin_keys, dim3(wide), dim3(256), 0, stream, tiles, d_xyz, n, S);
	size_t bytes = S.sort_tmp_bytes;
	const hipError_t e = rocprim::radix_sort_pairs(S.sort_tmp, bytes, (const uint32_t*)S.keys, S.keys_out, (const uint32_t*)S.vals, S.perm,
												  (size_t)n, 0u, tile_key_bits(tiles), stream);
	return e != hipSuccess ? e : hipGetLastError();
}

hipError_t launch_interpolate_tiles(const FieldDev& f, const double* d_xyz, uint64_t n, double* d_phi, double* d_grad, const TileBin& B,
									uint32_t xcd_chunk, hipStream_t stream)
{
	if (n == 0)
		return hipSuccess;
	if (B.sort_launched == 0)
	{
		// predicted ordered: only the probe (the prediction for the next batch: row-ordered queries are coherent enough; bin only if more
		// than a quarter of the steps change tile); the caller runs the queries as they came
		BinScratch P{};
		P.flag = B.flag;
		P.flag_host = B.flag_host;
		hipLaunchKernelGGL(k_bin_probe, dim3(1), dim3(1024), 0, stream, field_tiles(f), d_xyz, n, P, 4u);
		return hipGetLastError();
	}
	const uint32_t per_query = (uint32_t)((n + 255) / 256);
	hipLaunchKernelGGL(k_tile_keys, dim3(per_query), dim3(256), 0, stream, f, field_tiles(f), d_xyz, n, B);
	size_t bytes = B.sort_tmp_bytes;
	hipError_t e = rocprim::radix_sort_pairs(B.sort_tmp, bytes, (const uint32_t*)B.keys, B.keys_out, rocprim::counting_iterator<uint32_t>(0u), B.perm, (size_t)n,
											 0u, B.key_bits, stream);
	if (e != hipSuccess)
		return e;
	hipLaunchKernelGGL(k_tile_bounds, dim3(per_query), dim3(256), 0, stream, n, B);
	const uint32_t rows = (B.key_space + 1023u) / 1024u;
	hipLaunchKernelGGL(k_tile_row_items, dim3(rows), dim3(1024), 0, stream, B, B.row_items);
	hipLaunchKernelGGL(k_tile_items, dim3(rows), dim3(1024), 0, stream, B, (const uint32_t*)B.row_items);
	uint32_t grid = B.max_items;
	if (xcd_chunk != 0)
	{
		const uint32_t round = 8u * xcd_chunk;
		grid = (grid + round - 1) / round * round;
	}
#define DG_K2_TILES(LX, LY, LZ)                                                                                                                          \
	if (d_grad)                                                                                                                                          \
		hipLaunchKernelGGL((k_interpolate_tiles<true, LX, LY, LZ>), dim3(grid), dim3(StageShape<LX, LY, LZ>::NT), 0, stream, f, d_xyz, d_phi, d_grad, B, \
						   xcd_chunk, B.packed);                                                                                                         \
	else                                                                                                                                                 \
		hipLaunchKernelGGL((k_interpolate_tiles<false, LX, LY, LZ>), dim3(grid), dim3(StageShape<LX, LY, LZ>::NT), 0, stream, f, d_xyz, d_phi, d_grad, B, \
						   xcd_chunk, (double*)nullptr);
	DG_K2_TILES(3, 3, 3)
#undef DG_K2_TILES
	if (d_grad && B.packed != nullptr)
		hipLaunchKernelGGL(k_unpack_results, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, stream, (const double*)B.packed, n, d_phi, d_grad);
	return hipGetLastError();
}

hipError_t launch_interpolate(const FieldDev& f, const double* d_xyz, uint64_t n, double* d_phi, double* d_grad,
							  hipStream_t stream)
{
	if (n == 0)
		return hipSuccess;
	const uint32_t grid = k2_grid(n);
#define DG_K2_LAUNCH(MODE)                                                                                                  \
	if (d_grad)                                                                                                             \
		hipLaunchKernelGGL((k_interpolate<true, MODE>), dim3(grid), dim3(256), 0, stream, f, d_xyz, n, d_phi, d_grad);      \
	else                                                                                                                    \
		hipLaunchKernelGGL((k_interpolate<false, MODE>), dim3(grid), dim3(256), 0, stream, f, d_xyz, n, d_phi, d_grad);
	switch (field_mode(f))
	{
	case kFieldTileMajor: DG_K2_LAUNCH(kFieldTileMajor) break;
	case kFieldCellMajor: DG_K2_LAUNCH(kFieldCellMajor) break;
	case kFieldTable: DG_K2_LAUNCH(kFieldTable) break;
	default: DG_K2_LAUNCH(kFieldClosed)
	}
#undef DG_K2_LAUNCH
	return hipGetLastError();
}

hipError_t launch_interpolate_rows(const FieldDev& f, const double* d_xyz, uint64_t n, double* d_phi, double* d_grad, hipStream_t stream)
{
	if (n == 0)
		return hipSuccess;
	if (f.cell_major == nullptr)
		return hipErrorInvalidValue;
	const uint64_t waves = (n + 63) / 64;
	const uint32_t grid = (uint32_t)std::min<uint64_t>(waves, 256ull * 64ull); // grid-stride beyond 64 waves per CU
	if (d_grad)
		hipLaunchKernelGGL((k_interpolate_rows<true>), dim3(grid), dim3(64), 0, stream, f, d_xyz, n, d_phi, d_grad);
	else
		hipLaunchKernelGGL((k_interpolate_rows<false>), dim3(grid), dim3(64), 0, stream, f, d_xyz, n, d_phi, d_grad);
	return hipGetLastError();
}

hipError_t launch_interpolate_binned(const FieldDev& f, const double* d_xyz, uint64_t n, double* d_phi, double* d_grad,
									 const BinScratch& S, hipStream_t stream)
{
	if (n == 0)
		return hipSuccess;
	// K2: row-ordered queries are coherent enough; bin only if more than a quarter of the steps change tile
	const hipError_t e = launch_binning(field_tiles(f), field_tiles(f, kSortCells), d_xyz, n, S, 4u, stream);
	if (e != hipSuccess)
		return e;
	const uint32_t grid = k2_grid(n);
#define DG_K2_LAUNCH(MODE)                                                                                                        \
	if (d_grad)                                                                                                                   \
		hipLaunchKernelGGL((k_interpolate_binned<true, MODE>), dim3(grid), dim3(256), 0, stream, f, d_xyz, n, d_phi, d_grad, S);  \
	else                                                                                                                          \
		hipLaunchKernelGGL((k_interpolate_binned<false, MODE>), dim3(grid), dim3(256), 0, stream, f, d_xyz, n, d_phi, d_grad, S);
	switch (field_mode(f))
	{
	case kFieldTileMajor: DG_K2_LAUNCH(kFieldTileMajor) break;
	case kFieldCellMajor: DG_K2_LAUNCH(kFieldCellMajor) break;
	case kFieldTable: DG_K2_LAUNCH(kFieldTable) break;
	default: DG_K2_LAUNCH(kFieldClosed)
	}
#undef DG_K2_LAUNCH
	return hipGetLastError();
}

} // namespace dg
