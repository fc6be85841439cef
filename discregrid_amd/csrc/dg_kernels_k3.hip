// dg_kernels_k3.hip -- K3: SPH boundary density map (cmd/generate_density_map/main.cpp:86-133): k_density_cells (one lane per lattice
// point with its seven nodes, dg_density_cells.h) and the brick kernels k_density_bricks (reduced fields, short node ranges, lattices whose
// classes exceed 32-bit offsets).
// Compile with -ffp-contract=off (parity) -- see discregrid_amd/build.py.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include "dg_kernels.h"
#include "dg_device.h"

namespace dg
{
namespace
{

#ifndef DG_K3_WAVES
#define DG_K3_WAVES 1 // min waves per SIMD requested for K3 (register budget)
#endif

// ------------------------------------------------------------------------------------------------
// K3: SPH boundary density map (GenerateDensityMap).  One wave = one 4x4x4 brick of lattice nodes
// (K1's decomposition), so the 64 lanes evaluate the SDF in a 3x3x3-cell neighbourhood at every
// quadrature step and the 256-byte coefficient rows they read stay in L1.  Lanes whose node is
// rejected or beyond 2h idle; waves without an active lane exit at once.
template <bool STAGED, int MODE>
__global__ __launch_bounds__(64, DG_K3_WAVES) void k_density_bricks(const SampleParams L, const FieldDev F, const DensityParams P)
{
	uint32_t blk;
	if (!logical_block(L, blockIdx.x, &blk))
		return;
	const uint64_t brick = (uint64_t)blk; // one wave per block
	if (brick >= L.total_bricks)
		return;
	const LaneNode ln = map_lane(L, brick, (int)(threadIdx.x & 63u));
	if (!ln.valid)
		return;
	double v = 1.7976931348623157e308;
	if (L.mask == nullptr || L.mask[ln.out_idx] != 0)
	{
		double x[3];
		node_position(ln.cls, ln.a, ln.b, ln.s, L.dmin, L.cell, x);
		if (density_prefilter(F, P, x, &v))
			v = density_integral_t<STAGED, MODE>(F, P, x);
	}
	L.out[ln.out_idx] = v;
}

__device__ __forceinline__ int k3_cell_guess(const FieldDev& F, int d, double y)
{
	double t = (y - F.dmin[d]) * F.inv_cell[d];
	t = fmin(fmax(t, -8.0), (double)F.res[d] + 8.0);
	return (int)floor(t);
}

// ---- K3, one lane per lattice point (dg_density_cells.h) ----------------------------------------------------------------
// The wave context of k3c_lane() on the device: ballots, and the LDS tables of the axis states of the shifted coordinates.
// Producers: the X states of (variant v, lane x) come from the lane (x, y = v, z = 0), the Y states of (v, lane y) from
// the lane (x = v, y, z = 0), the Z states from lanes 0..5 for one k after the other -- each the pure function
// k3_axis_entry() of a coordinate that depends on the lattice index along that axis only, i.e. the value the consuming
// lane would compute itself.  One wave per workgroup: the barriers cost nothing and fence the compiler.
struct K3WaveDev
{
	K3Axis* sX; // [2][16]    (variant, lane x)
	K3Axis* sY; // [2][2]     (variant, lane y)
	K3Axis* sZ; // [16][3][2] (k, variant: lattice point / node A / node B, lane z)
	uint32_t* sZb; // [16][2] (k, lane z): k3c_axis_bits() of the z axis
	double* sR; // [7][64]    the seven sums of every lane (registers are what this kernel is short of)
	int lane, ix, iy, iz;
	__device__ __forceinline__ bool any(bool b) const { return __builtin_amdgcn_ballot_w64(b) != 0ull; }
	__device__ __forceinline__ uint32_t uniform(uint32_t v) const { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
	__device__ __forceinline__ void acc_init()
	{
#pragma unroll
		for (int n = 0; n < 7; ++n)
			sR[n * 64 + lane] = 0.0;
	}
	__device__ __forceinline__ void acc_add(int n, double t) { sR[n * 64 + lane] += t; }
	__device__ __forceinline__ double acc_get(int n) const { return sR[n * 64 + lane]; }
	__device__ __forceinline__ void set_x(const FieldDev& F, double a, double b)
	{
		const K3Axis e = k3_axis_entry(F, 0, iy == 0 ? a : b);
		__syncthreads();
		if (iz == 0)
			sX[iy * 16 + ix] = e;
		__syncthreads();
	}
	__device__ __forceinline__ void set_y(const FieldDev& F, double a, double b)
	{
		const K3Axis e = k3_axis_entry(F, 1, ix == 0 ? a : b);
		__syncthreads();
		if (iz == 0 && ix < 2)
			sY[ix * 2 + iy] = e;
		__syncthreads();
	}
	__device__ __forceinline__ void set_z(const FieldDev& F, const DensityParams& P, const SampleParams& L, const RowWave& m, double, double, double)
	{
		const int v = lane >> 1, s = lane & 1; // lanes 0..5
		uint32_t kz = m.w[2] * (uint32_t)kK3cLz + (uint32_t)s;
		kz = kz <= F.res[2] ? kz : F.res[2];
		double z = L.dmin[2] + L.cell[2] * (double)kz;
		if (v == 1)
			z = z + 1.0 / 3.0 * L.cell[2];
		else if (v == 2)
			z = z + 2.0 / 3.0 * L.cell[2];
		DG_NOUNROLL
		for (int k = 0; k < 16; ++k)
		{
			const K3Axis e = k3_axis_entry(F, 2, z + P.xi[k]);
			if (lane < 6)
				sZ[k * 6 + lane] = e; // (k * 3 + v) * 2 + s
		}
		__syncthreads();
		// the z axis' verdict on the seven nodes (k3c_axis_bits()) for every k and lane z, once
		if (lane < 32)
		{
			const int k = lane >> 1, sl = lane & 1;
			const K3Axis* z = sZ + k * 6 + sl;
			sZb[lane] = k3c_axis_bits(kK3cZ0, kK3cZA, kK3cZB, z[0].mi, z[0].inside != 0u, z[2].mi, z[2].inside != 0u, z[4].mi, z[4].inside != 0u);
		}
		__syncthreads();
	}
	__device__ __forceinline__ uint32_t z_bits(const FieldDev&, const DensityParams&, int k) const { return sZb[k * 2 + iz]; }
	__device__ __forceinline__ K3Axis x_var(const FieldDev&, int v) const { return sX[v * 16 + ix]; }
	__device__ __forceinline__ K3Axis y_var(const FieldDev&, int v) const { return sY[v * 2 + iy]; }
	__device__ __forceinline__ K3Axis z_var(const FieldDev&, const DensityParams&, int k, int v) const { return sZ[(k * 3 + v) * 2 + iz]; }
	__device__ __forceinline__ void x_id(const FieldDev&, int v, uint32_t* mi, bool* in) const
	{
		*mi = sX[v * 16 + ix].mi;
		*in = sX[v * 16 + ix].inside != 0u;
	}
	__device__ __forceinline__ void y_id(const FieldDev&, int v, uint32_t* mi, bool* in) const
	{
		*mi = sY[v * 2 + iy].mi;
		*in = sY[v * 2 + iy].inside != 0u;
	}
	__device__ __forceinline__ void z_id(const FieldDev&, const DensityParams&, int k, int v, uint32_t* mi, bool* in) const
	{
		*mi = sZ[(k * 3 + v) * 2 + iz].mi;
		*in = sZ[(k * 3 + v) * 2 + iz].inside != 0u;
	}
};
template <int WAVES>
__global__ __launch_bounds__(64, WAVES) void k_density_cells(const SampleParams L, const FieldDev F, const DensityParams P, const K3CellsGeom G)
{
	__shared__ K3Axis sX[2 * 16], sY[2 * 2], sZ[16 * 3 * 2];
	__shared__ double sR[7 * 64];
	__shared__ uint32_t sZb[16 * 2];
	uint32_t blk;
	if (!logical_block(L, blockIdx.x, &blk))
		return;
	if (blk >= P.row_prefix[4])
		return;
	const RowWave m = row_wave_map(P, blk);
	K3WaveDev w;
	w.sX = sX;
	w.sY = sY;
	w.sZ = sZ;
	w.sR = sR;
	w.sZb = sZb;
	w.lane = (int)(threadIdx.x & 63u);
	w.ix = w.lane & 15;
	w.iy = (w.lane >> 4) & 1;
	w.iz = w.lane >> 5;
	k3c_lane(w, L, F, P, G, m, w.lane);
}

// the x-major copy (dg_lattice.h): one thread per pair of the copy, contiguous 16-byte writes, reads a plane apart
__global__ __launch_bounds__(256) void k_xmajor_copy(const FieldDev F, uint32_t n_pairs, double* __restrict__ out)
{
	for (uint32_t e = blockIdx.x * 256u + threadIdx.x; e < n_pairs; e += gridDim.x * 256u)
	{
		const double* src = F.coeffs + xmajor_pair_node(e, F.res);
		double2 v;
		v.x = src[0];
		v.y = src[1];
		*(double2*)(out + 2 * (size_t)e) = v;
	}
}
// the "no value" bit of every cell, one wave per 64 cells of a row; only if k_field_check found such a value at all
// (flag bit 1) -- the kernel does not read the bits otherwise
__global__ __launch_bounds__(256) void k_xmajor_flags(const FieldDev F, const uint32_t* __restrict__ field_flags, uint64_t* __restrict__ out)
{
	if ((field_flags[0] & 2u) == 0u)
		return;
	const uint32_t words = xmajor_flag_words(F.res);
	const uint64_t total = (uint64_t)F.res[2] * F.res[1] * words;
	const int lane = (int)(threadIdx.x & 63u);
	for (uint64_t w = (uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6); w < total; w += (uint64_t)gridDim.x * 4u)
	{
		const uint32_t wi = (uint32_t)(w % words), row = (uint32_t)(w / words);
		const uint32_t j = row % F.res[1], k = row / F.res[1];
		const uint32_t i = wi * 64u + (uint32_t)lane;
		bool nov = false;
		if (i < F.res[0])
		{
			double cf[32];
			fetch_cell<kFieldXMajor>(F, i, j, k, 0u, cf);
#pragma unroll
			for (int q = 0; q < 32; ++q)
				nov = nov || (cf[q] == 1.7976931348623157e308);
		}
		const unsigned long long bits = __ballot(nov);
		if (lane == 0)
			out[w] = bits;
	}
}

__global__ __launch_bounds__(256) void k_field_check(const double* __restrict__ coeffs, uint64_t n, uint32_t* __restrict__ unsafe)
{
	bool bad = false, nov = false;
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
	{
		const double c = coeffs[i];
		nov = nov || c == 1.7976931348623157e308;
		bad = bad || (c != 1.7976931348623157e308 && !(fabs(c) < 1.0e290));
	}
	// bit 0: values that forbid skipping the zero-weight points; bit 1: the field holds "no value" coefficients at all
	const uint32_t bits = (__ballot(bad) != 0ull ? 1u : 0u) | (__ballot(nov) != 0ull ? 2u : 0u);
	if (bits != 0u && (threadIdx.x & 63u) == 0u)
		atomicOr(unsafe, bits);
}

} // namespace

hipError_t launch_density_bricks(const SampleParams& layout, const FieldDev& f, uint64_t n_coeffs, const DensityParams& p,
								 hipStream_t stream)
{
	if (layout.total_bricks == 0)
		return hipSuccess;
	if (p.unsafe != nullptr)
	{
		const hipError_t e = hipMemsetAsync(const_cast<uint32_t*>(p.unsafe), 0, sizeof(uint32_t), stream);
		if (e != hipSuccess)
			return e;
		const uint32_t blocks = (uint32_t)std::min<uint64_t>((n_coeffs + 255) / 256, 256ull * 16ull);
		hipLaunchKernelGGL(k_field_check, dim3(blocks), dim3(256), 0, stream, f.coeffs, n_coeffs, const_cast<uint32_t*>(p.unsafe));
	}
	static_assert(kWavesPerBlock == 1, "k_density_bricks assumes one brick per block");
	const dim3 grid(layout.blocks_per_xcd * 8u), block(64);
	const bool unreduced = f.cells == nullptr && f.cell_map == nullptr; // staged evaluator
	switch (field_mode(f))
	{
	case kFieldXMajor:
	{
		// one lane per lattice point with its seven nodes (dg_density_cells.h) over the whole lattice, on the x-major copy of the
		// Y / Z classes; the per-cell "no value" bits first
		if (p.row_shape != kRowShapeCells || !unreduced || f.xmajor_flags == nullptr || !k3c_geometry_fits(f.res))
			return hipErrorInvalidValue;
		if (p.unsafe != nullptr)
			hipLaunchKernelGGL(k_xmajor_flags, dim3(2048), dim3(256), 0, stream, f, p.unsafe, const_cast<uint64_t*>(f.xmajor_flags));
		const K3CellsGeom g = k3c_geometry(f);
		hipLaunchKernelGGL((k_density_cells<3>), grid, block, 0, stream, layout, f, p, g);
		break;
	}
	// everything else -- reduced fields (cell table), node ranges below an eighth of the lattice, lattices whose classes exceed
	// 32-bit offsets -- is the brick kernel: one node per lane, one wave per 4 x 4 x 4 brick of K1's decomposition
	case kFieldTileMajor: hipLaunchKernelGGL((k_density_bricks<true, kFieldTileMajor>), grid, block, 0, stream, layout, f, p); break;
	case kFieldCellMajor:
		if (unreduced) hipLaunchKernelGGL((k_density_bricks<true, kFieldCellMajor>), grid, block, 0, stream, layout, f, p);
		else hipLaunchKernelGGL((k_density_bricks<false, kFieldCellMajor>), grid, block, 0, stream, layout, f, p);
		break;
	case kFieldTable: hipLaunchKernelGGL((k_density_bricks<false, kFieldTable>), grid, block, 0, stream, layout, f, p); break;
	default:
		if (unreduced) hipLaunchKernelGGL((k_density_bricks<true, kFieldClosed>), grid, block, 0, stream, layout, f, p);
		else hipLaunchKernelGGL((k_density_bricks<false, kFieldClosed>), grid, block, 0, stream, layout, f, p);
	}
	return hipGetLastError();
}

hipError_t launch_xmajor_copy(const FieldDev& f, double* d_out, hipStream_t stream)
{
	const uint64_t n_pairs = xmajor_doubles(f.res) / 2;
	if (n_pairs == 0 || n_pairs > 0xffffffffull)
		return n_pairs == 0 ? hipSuccess : hipErrorInvalidValue;
	const uint32_t blocks = (uint32_t)std::min<uint64_t>((n_pairs + 255) / 256, 256ull * 64ull);
	hipLaunchKernelGGL(k_xmajor_copy, dim3(blocks), dim3(256), 0, stream, f, (uint32_t)n_pairs, d_out);
	return hipGetLastError();
}

} // namespace dg
