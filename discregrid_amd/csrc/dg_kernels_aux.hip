// dg_kernels_aux.hip -- U (packed all-gather buffer -> reference node order, multi-GPU) and the device side of reduceField
// (cubic_lagrange_discrete_grid.cpp:1065-1174).
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
// U: gathered packed shards -> reference node order.  One thread per node, coalesced stores,
// reads are contiguous runs of one plane row.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_unpack_shards(const UnpackParams P)
{
	const uint64_t total = P.class_off[4];
	for (uint64_t l = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; l < total;
		 l += (uint64_t)gridDim.x * blockDim.x)
	{
		int c = 0;
		if (l >= P.class_off[1]) c = 1;
		if (l >= P.class_off[2]) c = 2;
		if (l >= P.class_off[3]) c = 3;
		const uint64_t lc = l - P.class_off[c];
		const uint64_t plane = (uint64_t)P.D0[c] * P.D1[c];
		const uint32_t s = (uint32_t)(lc / plane);
		const uint64_t inplane = lc - (uint64_t)s * plane;
		const uint32_t slab = s / kSlabPlanes;
		const uint32_t r = slab % (uint32_t)P.nranks;
		const uint32_t q = (slab / (uint32_t)P.nranks) * kSlabPlanes + (s % kSlabPlanes);
		P.field[l] = P.gathered[(uint64_t)r * P.stride + P.pack_off[c][r] + (uint64_t)q * plane + inplane];
	}
}

// U, one piece: the slots [rank_begin, rank_end) of the gathered buffer -> reference node order.
// One thread per gathered value: contiguous reads, writes in contiguous runs of one plane.
__global__ __launch_bounds__(256) void k_unpack_ranks(const UnpackParams P)
{
	const uint64_t total = (uint64_t)(P.rank_end - P.rank_begin) * P.stride;
	for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (uint64_t)gridDim.x * blockDim.x)
	{
		const uint32_t r = (uint32_t)P.rank_begin + (uint32_t)(e / P.stride);
		const uint64_t off = e % P.stride;
		if (off >= P.count[r])
			continue; // padding of the slot
		P.field[unpack_dest(P, r, off)] = P.gathered[(uint64_t)r * P.stride + off];
	}
}

// K3 pre-pass: does the field hold values for which skipping the zero-weight quadrature points would
// change the result (NaN, Inf, |c| >= 1e290)?  DBL_MAX is the regular "no value" marker.
// ---- reduceField (dg_kernels.h: reduce_field_device) --------------------------------------------------------------
__global__ __launch_bounds__(256) void k_reduce_keep(const double* __restrict__ v, uint64_t n, const ReducePredicate P, uint8_t* __restrict__ keep,
													  uint8_t* __restrict__ used)
{
	for (uint64_t l = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; l < n; l += (uint64_t)gridDim.x * blockDim.x)
	{
		const double x = v[l];
		const bool ok = P.closed ? (P.lo <= x && x <= P.hi) : (P.lo < x + P.offset && x - P.offset < P.hi);
		keep[l] = (ok && x != 1.7976931348623157e308) ? 1 : 0;
		used[l] = 0;
	}
}
// a cell survives if any of its 32 nodes is kept (:1091-1098)
__global__ __launch_bounds__(256) void k_reduce_cell_flags(const uint32_t rx, const uint32_t ry, const uint32_t rz, const uint8_t* __restrict__ keep,
															uint32_t* __restrict__ flag)
{
	const uint64_t n_cells = (uint64_t)rx * ry * rz;
	const uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (c >= n_cells)
		return;
	const uint32_t res[3] = {rx, ry, rz};
	const uint32_t n01 = rx * ry;
	const uint32_t k = (uint32_t)(c / n01), r = (uint32_t)(c % n01);
	uint32_t idx[32];
	cell_node_indices(r % rx, r / rx, k, res, idx);
	bool any = false;
#pragma unroll
	for (int j = 0; j < 32; ++j)
		any = any || keep[idx[j]] != 0;
	flag[c] = any ? 1u : 0u;
}
// cell map + the nodes the surviving cells reference (:1099-1128)
__global__ __launch_bounds__(256) void k_reduce_cell_map(const uint32_t rx, const uint32_t ry, const uint32_t rz, const uint32_t* __restrict__ flag,
														  const uint32_t* __restrict__ row, uint32_t* __restrict__ cell_map, uint8_t* __restrict__ used)
{
	const uint64_t n_cells = (uint64_t)rx * ry * rz;
	const uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (c >= n_cells)
		return;
	if (flag[c] == 0u)
	{
		cell_map[c] = 0xffffffffu;
		return;
	}
	cell_map[c] = row[c];
	const uint32_t res[3] = {rx, ry, rz};
	const uint32_t n01 = rx * ry;
	const uint32_t k = (uint32_t)(c / n01), r = (uint32_t)(c % n01);
	uint32_t idx[32];
	cell_node_indices(r % rx, r / rx, k, res, idx);
#pragma unroll
	for (int j = 0; j < 32; ++j)
		used[idx[j]] = 1; // same value from every writer
}
struct ReduceGeom
{
	uint32_t res[3];
	double dmin[3], cell[3];
	double zscale;
};
// Morton keys of the surviving nodes, with the reference's arithmetic (:1110-1115)
__global__ __launch_bounds__(256) void k_reduce_keys(const ReduceGeom G, const uint32_t* __restrict__ node, uint64_t m, uint64_t* __restrict__ keys)
{
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (uint64_t)gridDim.x * blockDim.x)
	{
		double x[3];
		node_position_flat(node[i], G.res, G.dmin, G.cell, x);
		keys[i] = reference_z_value(x, G.zscale);
	}
}
// after the sort: new numbering, coefficients in the new order, and whether two survivors share a key
__global__ __launch_bounds__(256) void k_reduce_renumber(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ node, uint64_t m,
														  const double* __restrict__ v, uint32_t* __restrict__ new_id, double* __restrict__ out,
														  uint32_t* __restrict__ tied)
{
	bool tie = false;
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (uint64_t)gridDim.x * blockDim.x)
	{
		const uint32_t l = node[i];
		new_id[l] = (uint32_t)i;
		out[i] = v[l];
		tie = tie || (i + 1 < m && keys[i] == keys[i + 1]);
	}
	if (__ballot(tie) != 0ull && (threadIdx.x & 63u) == 0u)
		atomicOr(tied, 1u);
}
// rows of the surviving cells with the new node numbers (:1161-1173)
__global__ __launch_bounds__(256) void k_reduce_rows(const uint32_t rx, const uint32_t ry, const uint32_t rz, const uint32_t* __restrict__ cell_map,
													  const uint32_t* __restrict__ new_id, uint32_t* __restrict__ rows)
{
	const uint64_t n_cells = (uint64_t)rx * ry * rz;
	const uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (c >= n_cells)
		return;
	const uint32_t row = cell_map[c];
	if (row == 0xffffffffu)
		return;
	const uint32_t res[3] = {rx, ry, rz};
	const uint32_t n01 = rx * ry;
	const uint32_t k = (uint32_t)(c / n01), r = (uint32_t)(c % n01);
	uint32_t idx[32];
	cell_node_indices(r % rx, r / rx, k, res, idx);
	uint32_t* o = rows + 32 * (size_t)row;
#pragma unroll
	for (int j = 0; j < 32; ++j)
		o[j] = new_id[idx[j]];
}

} // namespace

hipError_t reduce_field_device(const uint32_t res[3], const double dmin[3], const double cell[3], const double inv_cell[3],
							   const double* d_coeffs, uint64_t n, const ReducePredicate& pred, ReduceResult& out, hipStream_t stream)
{
	const uint64_t n_cells = (uint64_t)res[0] * res[1] * res[2];
	struct Scratch
	{
		std::vector<void*> p;
		~Scratch()
		{
			for (void* q : p)
				(void)hipFree(q);
		}
		hipError_t get(void** q, size_t bytes)
		{
			const hipError_t e = hipMalloc(q, bytes ? bytes : 8);
			if (e == hipSuccess)
				p.push_back(*q);
			return e;
		}
	} S;
#define DG_TRY(x)                \
	do                           \
	{                            \
		const hipError_t e_ = (x); \
		if (e_ != hipSuccess)    \
			return e_;           \
	} while (0)
	uint8_t *keep = nullptr, *used = nullptr;
	uint32_t *flag = nullptr, *row = nullptr, *cell_map = nullptr, *node = nullptr, *node_sorted = nullptr, *new_id = nullptr, *counts = nullptr;
	DG_TRY(S.get((void**)&keep, n));
	DG_TRY(S.get((void**)&used, n));
	DG_TRY(S.get((void**)&flag, n_cells * 4));
	DG_TRY(S.get((void**)&row, n_cells * 4));
	DG_TRY(S.get((void**)&counts, 16));
	DG_TRY(hipMalloc((void**)&cell_map, n_cells * 4));
	out.d_cell_map = cell_map; // results are freed by the caller (also on failure: it owns `out`)
	DG_TRY(hipMemsetAsync(counts, 0, 16, stream));
	const uint32_t wide = (uint32_t)std::min<uint64_t>((n + 255) / 256, 256ull * 64ull);
	const uint32_t cgrid = (uint32_t)((n_cells + 255) / 256);
	hipLaunchKernelGGL(k_reduce_keep, dim3(wide), dim3(256), 0, stream, d_coeffs, n, pred, keep, used);
	hipLaunchKernelGGL(k_reduce_cell_flags, dim3(cgrid), dim3(256), 0, stream, res[0], res[1], res[2], keep, flag);
	size_t tmp_bytes = 0, b2 = 0;
	DG_TRY(rocprim::exclusive_scan(nullptr, tmp_bytes, flag, row, 0u, (size_t)n_cells, rocprim::plus<uint32_t>(), stream));
	DG_TRY(rocprim::select(nullptr, b2, rocprim::counting_iterator<uint32_t>(0u), used, (uint32_t*)nullptr, (uint32_t*)nullptr, (size_t)n, stream));
	tmp_bytes = std::max(tmp_bytes, b2);
	void* tmp = nullptr;
	DG_TRY(S.get(&tmp, tmp_bytes));
	size_t tb = tmp_bytes;
	DG_TRY(rocprim::exclusive_scan(tmp, tb, flag, row, 0u, (size_t)n_cells, rocprim::plus<uint32_t>(), stream));
	hipLaunchKernelGGL(k_reduce_cell_map, dim3(cgrid), dim3(256), 0, stream, res[0], res[1], res[2], flag, row, cell_map, used);
	DG_TRY(S.get((void**)&node, n * 4)); // survivors in node order (at most n)
	tb = tmp_bytes;
	DG_TRY(rocprim::select(tmp, tb, rocprim::counting_iterator<uint32_t>(0u), used, node, counts, (size_t)n, stream));
	DG_TRY(hipGetLastError());
	// sizes: surviving nodes, surviving cells
	uint32_t h_m = 0, h_last_row = 0, h_last_flag = 0;
	DG_TRY(hipMemcpyAsync(&h_m, counts, 4, hipMemcpyDeviceToHost, stream));
	DG_TRY(hipMemcpyAsync(&h_last_row, row + (n_cells - 1), 4, hipMemcpyDeviceToHost, stream));
	DG_TRY(hipMemcpyAsync(&h_last_flag, flag + (n_cells - 1), 4, hipMemcpyDeviceToHost, stream));
	DG_TRY(hipStreamSynchronize(stream));
	const uint64_t m = h_m, rows = (uint64_t)h_last_row + h_last_flag;
	out.n_nodes_out = m;
	out.n_rows = rows;
	DG_TRY(hipMalloc(&out.d_coeffs, std::max<uint64_t>(m, 1) * sizeof(double)));
	DG_TRY(hipMalloc(&out.d_cells, std::max<uint64_t>(rows, 1) * 32 * sizeof(uint32_t)));
	if (m == 0)
		return hipSuccess;
	uint64_t *keys = nullptr, *keys_sorted = nullptr;
	DG_TRY(S.get((void**)&keys, m * 8));
	DG_TRY(S.get((void**)&keys_sorted, m * 8));
	DG_TRY(S.get((void**)&node_sorted, m * 4));
	DG_TRY(S.get((void**)&new_id, n * 4));
	ReduceGeom G;
	for (int d = 0; d < 3; ++d)
	{
		G.res[d] = res[d];
		G.dmin[d] = dmin[d];
		G.cell[d] = cell[d];
	}
	G.zscale = 4.0 * std::min(std::min(inv_cell[0], inv_cell[1]), inv_cell[2]); // :1112
	const uint32_t mwide = (uint32_t)std::min<uint64_t>((m + 255) / 256, 256ull * 64ull);
	hipLaunchKernelGGL(k_reduce_keys, dim3(mwide), dim3(256), 0, stream, G, node, m, keys);
	size_t sort_bytes = 0;
	DG_TRY(rocprim::radix_sort_pairs(nullptr, sort_bytes, keys, keys_sorted, node, node_sorted, (size_t)m, 0u, (unsigned)kReferenceZBits, stream));
	void* sort_tmp = nullptr;
	DG_TRY(S.get(&sort_tmp, sort_bytes));
	DG_TRY(rocprim::radix_sort_pairs(sort_tmp, sort_bytes, keys, keys_sorted, node, node_sorted, (size_t)m, 0u, (unsigned)kReferenceZBits, stream));
	hipLaunchKernelGGL(k_reduce_renumber, dim3(mwide), dim3(256), 0, stream, keys_sorted, node_sorted, m, d_coeffs, new_id,
					   static_cast<double*>(out.d_coeffs), counts + 1);
	hipLaunchKernelGGL(k_reduce_rows, dim3(cgrid), dim3(256), 0, stream, res[0], res[1], res[2], cell_map, new_id, static_cast<uint32_t*>(out.d_cells));
	DG_TRY(hipGetLastError());
	uint32_t h_tied = 0;
	DG_TRY(hipMemcpyAsync(&h_tied, counts + 1, 4, hipMemcpyDeviceToHost, stream));
	DG_TRY(hipStreamSynchronize(stream));
	out.tied_keys = (int)h_tied;
#undef DG_TRY
	return hipSuccess;
}
hipError_t launch_unpack(const UnpackParams& p, hipStream_t stream)
{
	const uint64_t total = p.class_off[4];
	if (total == 0)
		return hipSuccess;
	uint64_t blocks = (total + 255) / 256;
	if (blocks > 256ull * 32ull)
		blocks = 256ull * 32ull;
	hipLaunchKernelGGL(k_unpack_shards, dim3((uint32_t)blocks), dim3(256), 0, stream, p);
	return hipGetLastError();
}

hipError_t launch_unpack_ranks(const UnpackParams& p, hipStream_t stream)
{
	const uint64_t total = (uint64_t)(p.rank_end - p.rank_begin) * p.stride;
	if (total == 0)
		return hipSuccess;
	uint64_t blocks = (total + 255) / 256;
	if (blocks > 256ull * 32ull)
		blocks = 256ull * 32ull;
	hipLaunchKernelGGL(k_unpack_ranks, dim3((uint32_t)blocks), dim3(256), 0, stream, p);
	return hipGetLastError();
}

} // namespace dg
