// dg_capi_field.cpp -- the field handle of include/discregrid_hip.h (coefficient vector, optional cell
// table / cell map / cell-major copy) and the device entry points of K2 (batched interpolate, with
// query binning) and K3 (density map).
#include "dg_capi_internal.h"

extern "C"
{

// ---- field + K2 ---------------------------------------------------------------------------------------------
static void fill_field(dg::FieldDev& F, const dg_grid_desc* g)
{
	for (int d = 0; d < 3; ++d)
	{
		F.dmin[d] = g->domain_min[d];
		F.dmax[d] = g->domain_max[d];
		F.cell[d] = g->cell_size[d];
		F.inv_cell[d] = g->inv_cell_size[d];
		F.res[d] = g->resolution[d];
	}
}

dg_status dg_field_attach_device(const dg_grid_desc* grid, const double* d_coeffs, uint64_t n_coeffs,
								 const uint32_t* d_cells, uint64_t n_cell_rows, const uint32_t* d_cell_map,
								 dg_field** out)
{
	if (!out)
		return fail(DG_ERR_INVALID, "out is null");
	*out = nullptr;
	if (!grid || !d_coeffs)
		return fail(DG_ERR_INVALID, "null argument");
	if (!valid_grid(grid))
		return fail(DG_ERR_INVALID, "invalid grid");
	if ((d_cells == nullptr) != (d_cell_map == nullptr))
		return fail(DG_ERR_INVALID, "cells and cell_map must be given together");
	if (!d_cells && n_coeffs != dg_grid_n_nodes(grid))
		return fail(DG_ERR_INVALID, "an unreduced field needs %llu coefficients, got %llu",
					(unsigned long long)dg_grid_n_nodes(grid), (unsigned long long)n_coeffs);
	(void)n_cell_rows;
	dg_field* f = new (std::nothrow) dg_field;
	if (!f)
		return fail(DG_ERR_ALLOC, "host allocation failed");
	fill_field(f->dev, grid);
	f->dev.coeffs = d_coeffs;
	f->dev.cells = d_cells;
	f->dev.cell_map = d_cell_map;
	f->dev.cell_major = nullptr;
	f->grid = *grid;
	f->n_coeffs = n_coeffs;
	f->n_rows = d_cells ? n_cell_rows : dg_grid_n_cells(grid);
	(void)hipGetDevice(&f->device);
	*out = f;
	return DG_OK;
}

dg_status dg_field_create(const dg_grid_desc* grid, const double* coeffs, uint64_t n_coeffs, const uint32_t* cells,
						  uint64_t n_cell_rows, const uint32_t* cell_map, dg_field** out)
{
	if (!out)
		return fail(DG_ERR_INVALID, "out is null");
	*out = nullptr;
	if (!grid || !coeffs)
		return fail(DG_ERR_INVALID, "null argument");
	if ((cells == nullptr) != (cell_map == nullptr))
		return fail(DG_ERR_INVALID, "cells and cell_map must be given together");
	dg_status s = require_device();
	if (s != DG_OK)
		return s;
	void *d_c = nullptr, *d_cells = nullptr, *d_map = nullptr;
	const uint64_t ncell = dg_grid_n_cells(grid);
	hipError_t e = hipMalloc(&d_c, n_coeffs * sizeof(double));
	if (e == hipSuccess) e = hipMemcpy(d_c, coeffs, n_coeffs * sizeof(double), hipMemcpyHostToDevice);
	if (e == hipSuccess && cells)
	{
		e = hipMalloc(&d_cells, std::max<uint64_t>(n_cell_rows, 1) * 32 * sizeof(uint32_t));
		if (e == hipSuccess && n_cell_rows)
			e = hipMemcpy(d_cells, cells, n_cell_rows * 32 * sizeof(uint32_t), hipMemcpyHostToDevice);
		if (e == hipSuccess) e = hipMalloc(&d_map, ncell * sizeof(uint32_t));
		if (e == hipSuccess) e = hipMemcpy(d_map, cell_map, ncell * sizeof(uint32_t), hipMemcpyHostToDevice);
	}
	dg_status st = DG_OK;
	if (e == hipSuccess)
		st = dg_field_attach_device(grid, (const double*)d_c, n_coeffs, (const uint32_t*)d_cells, n_cell_rows,
									(const uint32_t*)d_map, out);
	if (e != hipSuccess || st != DG_OK)
	{
		if (d_c) (void)hipFree(d_c);
		if (d_cells) (void)hipFree(d_cells);
		if (d_map) (void)hipFree(d_map);
		if (st != DG_OK)
			return st;
		return fail(e == hipErrorOutOfMemory ? DG_ERR_ALLOC : DG_ERR_HIP, "dg_field_create: %s", hipGetErrorString(e));
	}
	(*out)->owned[0] = d_c;
	(*out)->owned[1] = d_cells;
	(*out)->owned[2] = d_map;
	return DG_OK;
}

void dg_field_destroy(dg_field* f)
{
	if (!f)
		return;
	for (void* p : f->owned)
		if (p)
			(void)hipFree(p);
	if (f->d_cell_major)
		(void)hipFree(f->d_cell_major);
	if (f->d_wtab)
		(void)hipFree(f->d_wtab);
	if (f->d_unsafe)
		(void)hipFree(f->d_unsafe);
	f->scratch.destroy();
	if (f->bin_flag_host) (void)hipHostFree(f->bin_flag_host);
	delete f;
}

dg_status dg_field_build_cell_major(dg_field* field, void* stream)
{
	if (!field)
		return fail(DG_ERR_INVALID, "null argument");
	if (field->d_cell_major)
		return DG_OK;
	if (field->n_rows == 0)
		return DG_OK;
	void* p = nullptr;
	hipError_t e = hipMalloc(&p, field->n_rows * 32 * sizeof(double));
	if (e != hipSuccess)
		return fail(e == hipErrorOutOfMemory ? DG_ERR_ALLOC : DG_ERR_HIP, "cell-major allocation of %llu bytes: %s",
					(unsigned long long)(field->n_rows * 256), hipGetErrorString(e));
	e = dg::launch_expand_cells(field->dev, field->n_rows, static_cast<double*>(p), static_cast<hipStream_t>(stream));
	if (e != hipSuccess)
	{
		(void)hipFree(p);
		return fail(DG_ERR_HIP, "k_expand_cells: %s", hipGetErrorString(e));
	}
	field->d_cell_major = p;
	field->dev.cell_major = static_cast<const double*>(p);
	return DG_OK;
}

dg_status dg_field_drop_cell_major(dg_field* field)
{
	if (!field)
		return fail(DG_ERR_INVALID, "null argument");
	if (field->d_cell_major)
	{
		DG_HIP(hipDeviceSynchronize());
		(void)hipFree(field->d_cell_major);
		field->d_cell_major = nullptr;
		field->dev.cell_major = nullptr;
	}
	return DG_OK;
}

// ---- K3 ---------------------------------------------------------------------------------------------------
dg_status dg_density_map_nodes_device(dg_field* sdf, double support_radius, double rho0, int band_predicate,
									  uint64_t node_begin, uint64_t node_end, const uint8_t* d_pred_mask,
									  double* d_out, void* stream)
{
	if (!sdf || !d_out)
		return fail(DG_ERR_INVALID, "null argument");
	if (!(support_radius > 0.0))
		return fail(DG_ERR_INVALID, "support radius must be positive");
	const uint64_t total = dg_grid_n_nodes(&sdf->grid);
	if (node_begin > node_end || node_end > total)
		return fail(DG_ERR_INVALID, "node range outside the lattice");
	if (node_begin == node_end)
		return DG_OK;
	hipStream_t st = static_cast<hipStream_t>(stream);
	dg::DensityParams P;
	std::vector<double> w;
	dg::init_density_params(P, support_radius, rho0, sdf->grid.cell_size, band_predicate, w);
	if (sdf->wtab_h != support_radius)
	{
		if (!sdf->d_wtab)
			DG_HIP(hipMalloc(&sdf->d_wtab, 4096 * sizeof(double)));
		DG_HIP(hipMemcpy(sdf->d_wtab, w.data(), 4096 * sizeof(double), hipMemcpyHostToDevice));
		sdf->wtab_h = support_radius;
	}
	P.wtab = static_cast<const double*>(sdf->d_wtab);
	// K1's lattice decomposition: one wave per 4x4x4 brick of nodes
	dg::SampleParams L;
	dg::MeshDev none;
	std::memset(&none, 0, sizeof(none));
	dg::init_params(L, none, sdf->grid.domain_min, sdf->grid.cell_size, 0);
	dg::layout_range(L, sdf->grid.resolution, node_begin, node_end);
	L.mask = d_pred_mask;
	L.out = d_out;
	P.wtab = static_cast<const double*>(sdf->d_wtab);
	// zero-weight quadrature points are skipped unless the field holds non-finite / huge values (checked
	// on the device before every launch: an attached device array may have changed); DG_K3_SKIP=0: never
	if (env_int("DG_K3_SKIP", 1, 0, 1) != 0 && support_radius >= 1.0e-12)
	{
		if (!sdf->d_unsafe)
			DG_HIP(hipMalloc(&sdf->d_unsafe, sizeof(uint32_t)));
		P.skip_mode = 2;
		P.unsafe = static_cast<const uint32_t*>(sdf->d_unsafe);
	}
	DG_HIP(dg::launch_density_bricks(L, sdf->dev, sdf->n_coeffs, P, st));
	return DG_OK;
}

dg_status dg_interpolate_batch_device(const dg_field* field, const double* d_xyz, uint64_t n, double* d_phi,
									  double* d_grad, void* stream)
{
	if (!field || (n && (!d_xyz || !d_phi)))
		return fail(DG_ERR_INVALID, "null argument");
	hipStream_t st = static_cast<hipStream_t>(stream);
	// Large batches against a field that does not fit the L2s go through the binned path (queries in
	// arbitrary order are then processed tile by tile; ordered inputs are detected on the device and
	// run as they are).  DG_K2_BINNING=0 switches it off, =2 forces it for any size.
	const int binning = env_int("DG_K2_BINNING", 1, 0, 2);
	const bool big = n >= (1u << 18) && field->n_coeffs * sizeof(double) >= (32u << 20);
	if (binning != 0 && (big || binning == 2) && n < 0xffffffffull)
	{
		dg::BinScratch S;
		const int idx = acquire_bin_scratch(field->scratch, &field->bin_flag_host, dg::field_tiles(field->dev, dg::kSortCells), n, st, S);
		if (idx >= 0)
		{
			const hipError_t e = dg::launch_interpolate_binned(field->dev, d_xyz, n, d_phi, d_grad, S, st);
			field->scratch.release(idx, st);
			DG_HIP(e);
			return DG_OK;
		}
	}
	DG_HIP(dg::launch_interpolate(field->dev, d_xyz, n, d_phi, d_grad, st));
	return DG_OK;
}

} // extern "C"
