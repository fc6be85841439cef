// dg_layout.h -- host-side decomposition of the node lattice into the 4x4x4 bricks the K1
// kernel consumes, for (a) a flat node range [node_begin, node_end) and (b) one rank's shard
// of a multi-GPU run.  Pure index arithmetic, no device work.  Shared by the C ABI
// (dg_capi*.cpp) and the wave emulator of the CPU tests.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>
#include "dg_gauss16.h"
#include "dg_kernels.h"

namespace dg
{

struct ClassGeom
{
	uint32_t D[3];
	uint64_t off;  // global node offset of the class: [V | X | Y | Z]
	uint64_t size;
};

inline uint64_t class_geometry(const uint32_t res[3], ClassGeom cg[4])
{
	uint64_t off = 0;
	for (int c = 0; c < 4; ++c)
	{
		class_dims(c, res, cg[c].D);
		cg[c].off = off;
		cg[c].size = (uint64_t)cg[c].D[0] * cg[c].D[1] * cg[c].D[2];
		off += cg[c].size;
	}
	return off;
}

// number of planes s in [0, D2) owned by `rank` when slabs of kSlabPlanes planes are dealt
// round-robin to `nranks` ranks
inline uint32_t owned_planes(uint32_t D2, int rank, int nranks)
{
	const uint32_t T = kSlabPlanes;
	const uint32_t n_slabs = (D2 + T - 1) / T;
	uint32_t n = 0;
	for (uint32_t slab = (uint32_t)rank; slab < n_slabs; slab += (uint32_t)nranks)
		n += std::min(T, D2 - slab * T);
	return n;
}

inline uint64_t shard_count(const uint32_t res[3], int rank, int nranks)
{
	ClassGeom cg[4];
	class_geometry(res, cg);
	uint64_t cnt = 0;
	for (int c = 0; c < 4; ++c)
		cnt += (uint64_t)owned_planes(cg[c].D[2], rank, nranks) * cg[c].D[0] * cg[c].D[1];
	return cnt;
}

// blocks -> XCD chunks (dg_kernels.h: logical_block()).  Default chunk: kXcdChunk blocks, but at least
// 8 chunks per XCD so that small launches are still spread evenly (a 64^3 lattice has 29 k bricks:
// 29 chunks of 1024 would leave some XCDs with 4 chunks and others with 3).
inline void finish_blocks(SampleParams& P)
{
	if (P.xcd_chunk == 0)
		P.xcd_chunk = std::max(16u, std::min(kXcdChunk, P.n_blocks / 64u));
	if (P.xcd_chunk == 0xffffffffu) // one chunk per XCD
		P.xcd_chunk = std::max(1u, (P.n_blocks + 7) / 8);
	P.rcp_xcd_chunk = udiv_magic(P.xcd_chunk);
	const uint32_t per_group = 8u * P.xcd_chunk;
	P.blocks_per_xcd = ((P.n_blocks + per_group - 1) / per_group) * P.xcd_chunk;
}

inline void finish_bricks(SampleParams& P)
{
	uint64_t prefix = 0;
	for (int c = 0; c < 4; ++c)
	{
		ClassDesc& C = P.cls[c];
		const uint32_t nq = C.q_end > C.q_begin ? C.q_end - C.q_begin : 0;
		C.nb0 = (C.D0 + 3) / 4;
		C.nb1 = (C.D1 + 3) / 4;
		C.nbq = (nq + 3) / 4;
		C.rcp_nb0 = udiv_magic(C.nb0);
		C.rcp_nb01 = udiv_magic(C.nb0 * C.nb1);
		C.brick_prefix = prefix; // an empty class shares its prefix with the next one; the
		                         // kernel picks the LAST class whose prefix <= brick
		prefix += (uint64_t)C.nb0 * C.nb1 * C.nbq;
	}
	P.total_bricks = prefix;
	P.n_blocks = (uint32_t)((prefix + kWavesPerBlock - 1) / kWavesPerBlock);
	finish_blocks(P);
}


// two nodes per cell edge run along); the vertex class keeps its 4 x 4 x 4 bricks.  Call after layout_range().

inline void init_params(SampleParams& P, const MeshDev& mesh, const double dmin[3], const double cell[3], int invert)
{
	std::memset(&P, 0, sizeof(P));
	P.mesh = mesh;
	for (int d = 0; d < 3; ++d)
	{
		P.dmin[d] = dmin[d];
		P.cell[d] = cell[d];
	}
	P.invert = invert ? 1 : 0;
	P.shard_rank = 0;
	P.shard_n = 1;
}

// out[l - node_begin] for l in [node_begin, node_end)
inline void layout_range(SampleParams& P, const uint32_t res[3], uint64_t node_begin, uint64_t node_end)
{
	ClassGeom cg[4];
	class_geometry(res, cg);
	for (int c = 0; c < 4; ++c)
	{
		ClassDesc& C = P.cls[c];
		C.D0 = cg[c].D[0];
		C.D1 = cg[c].D[1];
		C.D2 = cg[c].D[2];
		const uint64_t lo = std::max(node_begin, cg[c].off), hi = std::min(node_end, cg[c].off + cg[c].size);
		if (lo < hi)
		{
			const uint64_t plane = (uint64_t)C.D0 * C.D1;
			C.l_begin = lo - cg[c].off;
			C.l_end = hi - cg[c].off;
			C.q_begin = (uint32_t)(C.l_begin / plane);
			C.q_end = (uint32_t)((C.l_end + plane - 1) / plane);
		}
		else
		{
			C.l_begin = C.l_end = 0;
			C.q_begin = C.q_end = 0;
		}
		C.out_base = (int64_t)cg[c].off - (int64_t)node_begin;
	}
	P.shard_rank = 0;
	P.shard_n = 1;
	finish_bricks(P);
}

// Contiguous chunks: every class's planes [0, D2) are cut into `nchunks` runs of whole planes, chunk v of class c =
// planes [cuts[c][v], cuts[c][v + 1]) -- one contiguous run of the coefficient vector per class (the slowest index of
// each class is k, k, i, j).  The cuts equalise the cumulated plane cost (cost[c][s]: relative cost of plane s of
// class c; null: every plane costs the same) and fall on multiples of the brick depth where the class has at least
// one brick layer per chunk.
inline void chunk_planes(const uint32_t res[3], int nchunks, const float* const cost[4], uint32_t cuts[4][kMaxRanks + 1])
{
	ClassGeom cg[4];
	class_geometry(res, cg);
	for (int c = 0; c < 4; ++c)
	{
		const uint32_t D2 = cg[c].D[2];
		const uint32_t step = D2 >= (uint32_t)nchunks * (uint32_t)kSlabPlanes ? (uint32_t)kSlabPlanes : 1u;
		std::vector<double> cum(D2 + 1, 0.0);
		for (uint32_t sl = 0; sl < D2; ++sl)
		{
			const double w = cost && cost[c] ? (double)cost[c][sl] : 1.0;
			cum[sl + 1] = cum[sl] + (w > 0.0 ? w : 0.0);
		}
		if (!(cum[D2] > 0.0))
			for (uint32_t sl = 0; sl <= D2; ++sl)
				cum[sl] = (double)sl;
		cuts[c][0] = 0;
		for (int v = 1; v < nchunks; ++v)
		{
			const double want = cum[D2] * (double)v / (double)nchunks;
			uint32_t at = (uint32_t)(std::lower_bound(cum.begin(), cum.end(), want) - cum.begin());
			at = std::min(D2, (at + step / 2) / step * step);
			cuts[c][v] = std::max(at, cuts[c][v - 1]);
		}
		cuts[c][nchunks] = D2;
	}
}

// planes [q_begin[c], q_end[c]) of every class, written to their places in the whole coefficient vector
inline void layout_class_planes(SampleParams& P, const uint32_t res[3], const uint32_t q_begin[4], const uint32_t q_end[4])
{
	ClassGeom cg[4];
	class_geometry(res, cg);
	for (int c = 0; c < 4; ++c)
	{
		ClassDesc& C = P.cls[c];
		C.D0 = cg[c].D[0];
		C.D1 = cg[c].D[1];
		C.D2 = cg[c].D[2];
		const uint64_t plane = (uint64_t)C.D0 * C.D1;
		C.q_begin = std::min(q_begin[c], C.D2);
		C.q_end = std::max(C.q_begin, std::min(q_end[c], C.D2));
		C.l_begin = (uint64_t)C.q_begin * plane;
		C.l_end = (uint64_t)C.q_end * plane;
		C.out_base = (int64_t)cg[c].off;
	}
	P.shard_rank = 0;
	P.shard_n = 1;
	finish_bricks(P);
}

// packed buffer of rank `rank`: [V planes | X planes | Y planes | Z planes] it owns
inline void layout_shard(SampleParams& P, const uint32_t res[3], int rank, int nranks)
{
	ClassGeom cg[4];
	class_geometry(res, cg);
	uint64_t pack = 0;
	for (int c = 0; c < 4; ++c)
	{
		ClassDesc& C = P.cls[c];
		C.D0 = cg[c].D[0];
		C.D1 = cg[c].D[1];
		C.D2 = cg[c].D[2];
		C.l_begin = 0;
		C.l_end = cg[c].size;
		C.q_begin = 0;
		C.q_end = owned_planes(C.D2, rank, nranks);
		C.out_base = (int64_t)pack;
		pack += (uint64_t)C.q_end * C.D0 * C.D1;
	}
	P.shard_rank = rank;
	P.shard_n = nranks;
	finish_bricks(P);
}

// K1p: n points, 64 per wave in processing order (P.pts filled by the caller)
inline void layout_points(SampleParams& P, uint64_t n)
{
	for (int c = 0; c < 4; ++c)
		std::memset(&P.cls[c], 0, sizeof(ClassDesc));
	P.shard_rank = 0;
	P.shard_n = 1;
	P.total_bricks = (n + 63) / 64;
	P.n_blocks = (uint32_t)((P.total_bricks + kWavesPerBlock - 1) / kWavesPerBlock);
	finish_blocks(P);
}

inline void layout_unpack(UnpackParams& U, const uint32_t res[3], int nranks)
{
	std::memset(&U, 0, sizeof(U));
	ClassGeom cg[4];
	const uint64_t total = class_geometry(res, cg);
	for (int c = 0; c < 4; ++c)
	{
		U.D0[c] = cg[c].D[0];
		U.D1[c] = cg[c].D[1];
		U.D2[c] = cg[c].D[2];
		U.class_off[c] = cg[c].off;
	}
	U.class_off[4] = total;
	for (int r = 0; r < nranks; ++r)
	{
		uint64_t pack = 0;
		for (int c = 0; c < 4; ++c)
		{
			U.pack_off[c][r] = pack;
			pack += (uint64_t)owned_planes(cg[c].D[2], r, nranks) * cg[c].D[0] * cg[c].D[1];
		}
		U.count[r] = pack;
	}
	U.nranks = nranks;
	U.rank_begin = 0;
	U.rank_end = nranks;
}

// K3 launch constants: quadrature offsets/weights and the table of kernel values W(xi) for
// support radius h (host arithmetic, same operations as the reference performs per call).
inline void init_density_params(DensityParams& P, double h, double rho0, const double cell_size[3], int band,
								std::vector<double>& wtab_host)
{
	struct HostSqrt
	{
		double operator()(double x) const { return std::sqrt(x); }
	};
	P.h = h;
	P.rho0 = rho0;
	P.c0prod = h * (h * h);
	P.cell_diag = std::sqrt(cell_size[0] * cell_size[0] + (cell_size[1] * cell_size[1] + cell_size[2] * cell_size[2]));
	P.band_predicate = band ? 1 : 0;
	for (int i = 0; i < 16; ++i)
	{
		P.xi[i] = h * kGaussX[i] + 0.0; // c0 * abscissa + c1 with c0 = 0.5 * (h - (-h)) = h, c1 = 0.5 * (-h + h) = 0
		P.w[i] = kGaussW[i];
	}
	const double k = cubic_kernel_k(h);
	// [0, 4096): W(xi_i, xi_j, xi_k); [4096, 8192): the weight products (w_i * w_j) * w_k as the kernels' loops form them
	// (k_density_cells reads them as scalars instead of keeping w_i w_j in vector registers)
	wtab_host.resize(8192);
	for (int i = 0; i < 16; ++i)
		for (int j = 0; j < 16; ++j)
			for (int kk = 0; kk < 16; ++kk)
			{
				wtab_host[(i * 16 + j) * 16 + kk] = cubic_kernel_W(P.xi[i], P.xi[j], P.xi[kk], h, k, HostSqrt());
				const double wi = P.w[i];
				const double wij = wi * P.w[j];
				wtab_host[4096 + (i * 16 + j) * 16 + kk] = wij * P.w[kk];
			}
	P.wtab = nullptr;
	for (int i = 0; i < 16; ++i)
		for (int j = 0; j < 16; ++j)
		{
			uint32_t m = 0;
			for (int kk = 0; kk < 16; ++kk)
				if (wtab_host[(i * 16 + j) * 16 + kk] != 0.0)
					m |= 1u << kk;
			P.kmask[i * 16 + j] = (uint16_t)m;
		}
	P.rcp_h = 1.0 / h;
	{
		// k3c_div_h(): Markstein's correction needs RN(1 / h) (the line above) and excludes divisors whose significand is all ones
		uint64_t bits;
		std::memcpy(&bits, &h, sizeof(bits));
		P.fast_div = (h >= 1.0e-12 && h <= 1.0e12 && (bits & 0xfffffffffffffull) != 0xfffffffffffffull) ? 1 : 0;
	}
	P.skip_mode = 0;
	P.row_shape = 0;
	P.unsafe = nullptr;
}
// K3 with one lane per lattice point (k_density_cells): waves of 16 x 2 x 2 points over the (n + 1)^3 point lattice, ids in
// blocks of `block` waves along x / y / z (one "class" holds every wave; DensityParams::row_block); returns the number of waves
inline uint64_t layout_density_cells(DensityParams& P, SampleParams& L, const uint32_t res[3], const uint32_t block[3])
{
	const uint32_t l[3] = {(uint32_t)kK3cLx, (uint32_t)kK3cLy, (uint32_t)kK3cLz};
	P.row_shape = kRowShapeCells;
	uint64_t n = 1;
	for (int d = 0; d < 3; ++d)
	{
		P.row_waves[0][d] = (res[d] + 1 + l[d] - 1) / l[d];
		n *= P.row_waves[0][d];
		for (int c = 1; c < 4; ++c)
			P.row_waves[c][d] = 0;
	}
	P.row_prefix[0] = 0;
	for (int c = 1; c <= 4; ++c)
		P.row_prefix[c] = (uint32_t)n;
	P.row_node_begin = 0;
	P.row_node_end = ~0ull;
	for (int d = 0; d < 3; ++d)
		P.row_block[d] = std::max(1u, block[d]);
	L.total_bricks = n;
	L.n_blocks = (uint32_t)n;
	L.xcd_chunk = 0;
	finish_blocks(L);
	return n;
}
// may the zero-weight points be skipped for this coefficient? (host mirror of k_field_check)
inline bool density_value_unsafe(double c) { return c != 1.7976931348623157e308 && !(std::fabs(c) < 1.0e290); }

// the source index k_unpack_shards reads for global node l (host mirror, used by tests)
inline uint64_t unpack_source(const UnpackParams& U, uint64_t l)
{
	int c = 0;
	if (l >= U.class_off[1]) c = 1;
	if (l >= U.class_off[2]) c = 2;
	if (l >= U.class_off[3]) c = 3;
	const uint64_t lc = l - U.class_off[c];
	const uint64_t plane = (uint64_t)U.D0[c] * U.D1[c];
	const uint32_t s = (uint32_t)(lc / plane);
	const uint64_t inplane = lc - (uint64_t)s * plane;
	const uint32_t slab = s / kSlabPlanes;
	const uint32_t r = slab % (uint32_t)U.nranks;
	const uint32_t q = (slab / (uint32_t)U.nranks) * kSlabPlanes + (s % kSlabPlanes);
	return (uint64_t)r * U.stride + U.pack_off[c][r] + (uint64_t)q * plane + inplane;
}

} // namespace dg
