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
	if (!grid || (!d_coeffs && n_coeffs != 0))
		return fail(DG_ERR_INVALID, "null argument");
	if (!valid_grid(grid))
		return fail(DG_ERR_INVALID, "invalid grid");
	// a field reduced to nothing (reduceField removed every cell) has an empty cell table and a cell map
	// that marks every cell as removed: every query then returns DG_NO_VALUE, as in the reference (:992-994)
	const bool reduced_to_nothing = d_cells == nullptr && d_cell_map != nullptr && n_cell_rows == 0;
	if ((d_cells == nullptr) != (d_cell_map == nullptr) && !reduced_to_nothing)
		return fail(DG_ERR_INVALID, "cells and cell_map must be given together");
	if (!d_cells && !d_cell_map && n_coeffs != dg_grid_n_nodes(grid))
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
	f->dev.tile_major = nullptr;
	for (int d = 0; d < 3; ++d)
		f->dev.ntile[d] = (f->dev.res[d] + dg::kTmCells - 1) / dg::kTmCells;
	f->grid = *grid;
	f->n_coeffs = n_coeffs;
	f->n_rows = (d_cells || d_cell_map) ? n_cell_rows : dg_grid_n_cells(grid);
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
	if (!grid || (!coeffs && n_coeffs != 0))
		return fail(DG_ERR_INVALID, "null argument");
	if (!valid_grid(grid))
		return fail(DG_ERR_INVALID, "invalid grid");
	const uint64_t ncell = dg_grid_n_cells(grid);
	if (n_cell_rows == 0 && cell_map != nullptr)
	{
		// reduced to nothing: the (possibly null) table has no rows, so the map may only hold "removed"
		for (uint64_t i = 0; i < ncell; ++i)
			if (cell_map[i] != 0xffffffffu)
				return fail(DG_ERR_INVALID, "cell_map[%llu] refers to a row of an empty cell table", (unsigned long long)i);
		cells = nullptr;
	}
	else if ((cells == nullptr) != (cell_map == nullptr))
		return fail(DG_ERR_INVALID, "cells and cell_map must be given together");
	dg_status s = require_device();
	if (s != DG_OK)
		return s;
	void *d_c = nullptr, *d_cells = nullptr, *d_map = nullptr;
	hipError_t e = hipMalloc(&d_c, std::max<uint64_t>(n_coeffs, 1) * sizeof(double));
	if (e == hipSuccess && n_coeffs) e = hipMemcpy(d_c, coeffs, n_coeffs * sizeof(double), hipMemcpyHostToDevice);
	if (e == hipSuccess && cells)
	{
		e = hipMalloc(&d_cells, n_cell_rows * 32 * sizeof(uint32_t));
		if (e == hipSuccess)
			e = hipMemcpy(d_cells, cells, n_cell_rows * 32 * sizeof(uint32_t), hipMemcpyHostToDevice);
	}
	if (e == hipSuccess && cell_map)
	{
		e = hipMalloc(&d_map, ncell * sizeof(uint32_t));
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
	DeviceGuard guard(f->device);
	(void)finish_host_job(f); // a copy into the caller's array that is still running
	if (f->producer_stream)
	{
		(void)hipStreamSynchronize(f->producer_stream);
		recycle_stream(f->device, f->producer_stream);
	}
	if (f->produced)
		(void)hipEventDestroy(f->produced);
	if (f->d_producer_mask)
		(void)hipFree(f->d_producer_mask);
	if (f->owned[0] && f->recyclable_bytes)
	{
		(void)hipDeviceSynchronize(); // nothing may still read the array when the next field is sampled into it
		if (recycle_field_buffer(f->owned[0], f->recyclable_bytes, f->device))
			f->owned[0] = nullptr;
	}
	for (void* p : f->owned)
		if (p)
			(void)hipFree(p);
	if (f->d_cell_major)
		(void)hipFree(f->d_cell_major);
	if (f->d_band_rows)
		(void)hipFree(f->d_band_rows);
	if (f->d_band_map)
		(void)hipFree(f->d_band_map);
	if (f->band_ready)
		(void)hipEventDestroy(f->band_ready);
	if (f->d_tile_major)
		(void)hipFree(f->d_tile_major);
	if (f->cell_major_ready)
		(void)hipEventDestroy(f->cell_major_ready);
	if (f->tile_major_ready)
		(void)hipEventDestroy(f->tile_major_ready);
	for (auto& kv : f->wtabs)
		(void)hipFree(kv.second);
	f->scratch.destroy();
	f->flag_scratch.destroy();
	f->tile_scratch.destroy();
	if (f->bin_flag_host) (void)hipHostFree(f->bin_flag_host);
	if (f->band_probe_host) (void)hipHostFree(f->band_probe_host);
	delete f;
}

static dg_status build_cell_major_locked(dg_field* field, void* stream);
dg_status dg_field_build_cell_major(dg_field* field, void* stream)
{
	if (!field)
		return fail(DG_ERR_INVALID, "null argument");
	std::lock_guard<std::mutex> lock(field->copy_mutex);
	return build_cell_major_locked(field, stream);
}
static dg_status build_cell_major_locked(dg_field* field, void* stream) // field->copy_mutex held
{
	if (field->d_cell_major)
		return DG_OK;
	if (field->n_rows == 0)
		return DG_OK;
	DG_ON_DEVICE_OF(field);
	void* p = nullptr;
	hipError_t e = hipMalloc(&p, field->n_rows * 32 * sizeof(double));
	if (e != hipSuccess)
		return fail(e == hipErrorOutOfMemory ? DG_ERR_ALLOC : DG_ERR_HIP, "cell-major allocation of %llu bytes: %s",
					(unsigned long long)(field->n_rows * 256), hipGetErrorString(e));
	if (!field->cell_major_ready)
		e = hipEventCreateWithFlags(&field->cell_major_ready, hipEventDisableTiming);
	if (e == hipSuccess)
		e = wait_produced(field, static_cast<hipStream_t>(stream));
	if (e == hipSuccess)
		e = dg::launch_expand_cells(field->dev, field->n_rows, static_cast<double*>(p), static_cast<hipStream_t>(stream));
	if (e == hipSuccess)
		e = hipEventRecord(field->cell_major_ready, static_cast<hipStream_t>(stream));
	if (e != hipSuccess)
	{
		(void)hipFree(p);
		return fail(DG_ERR_HIP, "k_expand_cells: %s", hipGetErrorString(e));
	}
	field->d_cell_major = p;
	field->dev.cell_major = static_cast<const double*>(p);
	return DG_OK;
}

dg_status dg_field_build_tile_major(dg_field* field, void* stream)
{
	if (!field)
		return fail(DG_ERR_INVALID, "null argument");
	std::lock_guard<std::mutex> lock(field->copy_mutex);
	if (field->d_tile_major)
		return DG_OK;
	if (field->dev.cells != nullptr || field->dev.cell_map != nullptr)
		return fail(DG_ERR_INVALID, "the tile-major copy exists for unreduced fields only");
	DG_ON_DEVICE_OF(field);
	const uint64_t n_tiles = (uint64_t)field->dev.ntile[0] * field->dev.ntile[1] * field->dev.ntile[2];
	void* p = nullptr;
	hipError_t e = hipMalloc(&p, n_tiles * dg::kTmNodes * sizeof(double));
	if (e != hipSuccess)
		return fail(e == hipErrorOutOfMemory ? DG_ERR_ALLOC : DG_ERR_HIP, "tile-major allocation of %llu bytes: %s",
					(unsigned long long)(n_tiles * dg::kTmNodes * sizeof(double)), hipGetErrorString(e));
	if (!field->tile_major_ready)
		e = hipEventCreateWithFlags(&field->tile_major_ready, hipEventDisableTiming);
	if (e == hipSuccess)
		e = wait_produced(field, static_cast<hipStream_t>(stream));
	if (e == hipSuccess)
		e = dg::launch_expand_tiles(field->dev, n_tiles, static_cast<double*>(p), static_cast<hipStream_t>(stream));
	if (e == hipSuccess)
		e = hipEventRecord(field->tile_major_ready, static_cast<hipStream_t>(stream));
	if (e != hipSuccess)
	{
		(void)hipFree(p);
		return fail(DG_ERR_HIP, "k_expand_tiles: %s", hipGetErrorString(e));
	}
	field->d_tile_major = p;
	field->dev.tile_major = static_cast<const double*>(p);
	return DG_OK;
}

dg_status dg_field_drop_tile_major(dg_field* field)
{
	if (!field)
		return fail(DG_ERR_INVALID, "null argument");
	std::lock_guard<std::mutex> lock(field->copy_mutex);
	if (field->d_tile_major)
	{
		DG_ON_DEVICE_OF(field);
		DG_HIP(hipDeviceSynchronize());
		(void)hipFree(field->d_tile_major);
		field->d_tile_major = nullptr;
		field->dev.tile_major = nullptr;
	}
	return DG_OK;
}

dg_status dg_field_drop_cell_major(dg_field* field)
{
	if (!field)
		return fail(DG_ERR_INVALID, "null argument");
	std::lock_guard<std::mutex> lock(field->copy_mutex);
	field->auto_copy_tried = true; // the caller decided against the copy: K2 does not bring it back by itself
	if (field->d_cell_major || field->d_band_rows)
	{
		DG_ON_DEVICE_OF(field);
		DG_HIP(hipDeviceSynchronize());
		if (field->d_cell_major) (void)hipFree(field->d_cell_major);
		field->d_cell_major = nullptr;
		field->dev.cell_major = nullptr;
		if (field->d_band_rows) (void)hipFree(field->d_band_rows);
		if (field->d_band_map) (void)hipFree(field->d_band_map);
		field->d_band_rows = field->d_band_map = nullptr;
		field->dev.band_rows = nullptr;
		field->dev.band_bits = nullptr;
		field->dev.band_rank = nullptr;
		field->band_rows = 0;
	}
	return DG_OK;
}

// The band-limited cell-major copy: rows only for the cells whose coefficients reach into [lo, hi].
dg_status dg_field_build_cell_major_band(dg_field* field, double lo, double hi, void* stream, uint64_t* rows_out)
{
	if (!field || !(lo <= hi))
		return fail(DG_ERR_INVALID, "null field or an empty band");
	std::lock_guard<std::mutex> lock(field->copy_mutex);
	if (field->d_band_rows || field->n_rows == 0)
	{
		if (rows_out)
			*rows_out = field->band_rows;
		return DG_OK;
	}
	if (field->n_rows >= 0xffffffffull)
		return fail(DG_ERR_INVALID, "too many cell rows for a 32-bit row map");
	DG_ON_DEVICE_OF(field);
	hipStream_t st = static_cast<hipStream_t>(stream);
	DG_HIP(wait_produced(field, st));
	const uint64_t n = field->n_rows;
	void *d_flag = nullptr, *d_pos = nullptr, *d_tmp = nullptr, *d_map = nullptr, *d_rows = nullptr; // d_map: bit words, then rank words
	const uint64_t words = (n + 63) / 64;
	size_t tmp_bytes = 0;
	hipError_t e = hipMalloc(&d_flag, n * sizeof(uint32_t));
	if (e == hipSuccess) e = hipMalloc(&d_pos, (n + 1) * sizeof(uint32_t));
	if (e == hipSuccess) e = hipMalloc(&d_map, words * (sizeof(uint64_t) + sizeof(uint32_t)));
	if (e == hipSuccess) e = dg::band_scan(static_cast<uint32_t*>(d_flag), static_cast<uint32_t*>(d_pos), n, nullptr, &tmp_bytes, st);
	if (e == hipSuccess) e = hipMalloc(&d_tmp, std::max<size_t>(tmp_bytes, 256));
	if (e == hipSuccess) e = dg::launch_band_flags(field->dev, n, lo, hi, static_cast<uint32_t*>(d_flag), st);
	if (e == hipSuccess) e = dg::band_scan(static_cast<uint32_t*>(d_flag), static_cast<uint32_t*>(d_pos), n, d_tmp, &tmp_bytes, st);
	uint32_t last_pos = 0, last_flag = 0;
	if (e == hipSuccess) e = hipMemcpyAsync(&last_pos, static_cast<uint32_t*>(d_pos) + (n - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, st);
	if (e == hipSuccess) e = hipMemcpyAsync(&last_flag, static_cast<uint32_t*>(d_flag) + (n - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, st);
	if (e == hipSuccess) e = hipStreamSynchronize(st); // (building a copy is a one-off: the row count has to reach the host)
	const uint64_t rows = (uint64_t)last_pos + last_flag;
	if (e == hipSuccess) e = hipMalloc(&d_rows, std::max<uint64_t>(rows, 1) * 32 * sizeof(double));
	if (e == hipSuccess && !field->band_ready) e = hipEventCreateWithFlags(&field->band_ready, hipEventDisableTiming);
	if (e == hipSuccess)
		e = dg::launch_band_expand(field->dev, n, static_cast<uint32_t*>(d_flag), static_cast<uint32_t*>(d_pos), static_cast<uint64_t*>(d_map),
								   reinterpret_cast<uint32_t*>(static_cast<uint64_t*>(d_map) + words), static_cast<double*>(d_rows), st);
	if (e == hipSuccess) e = hipEventRecord(field->band_ready, st);
	if (e == hipSuccess) e = hipStreamSynchronize(st); // the scratch below is freed right away
	(void)hipFree(d_flag);
	(void)hipFree(d_pos);
	(void)hipFree(d_tmp);
	if (e != hipSuccess)
	{
		(void)hipFree(d_map);
		(void)hipFree(d_rows);
		return fail(e == hipErrorOutOfMemory ? DG_ERR_ALLOC : DG_ERR_HIP, "band-limited cell-major copy: %s", hipGetErrorString(e));
	}
	field->d_band_rows = d_rows;
	field->d_band_map = d_map;
	field->band_rows = rows;
	field->dev.band_rows = static_cast<const double*>(d_rows);
	field->dev.band_bits = static_cast<const uint64_t*>(d_map);
	field->dev.band_rank = reinterpret_cast<const uint32_t*>(static_cast<const uint64_t*>(d_map) + words);
	if (rows_out)
		*rows_out = rows;
	return DG_OK;
}

dg_status dg_field_set_immutable(dg_field* field, int immutable)
{
	if (!field)
		return fail(DG_ERR_INVALID, "null argument");
	std::lock_guard<std::mutex> lock(field->copy_mutex);
	field->immutable = immutable != 0;
	return DG_OK;
}

dg_status dg_field_get_info(const dg_field* field, dg_field_info* info)
{
	if (!field || !info)
		return fail(DG_ERR_INVALID, "null argument");
	std::memset(info, 0, sizeof(*info));
	info->n_coeffs = field->n_coeffs;
	info->n_cell_rows = field->n_rows;
	info->device = field->device;
	info->owns_coefficients = field->owned[0] != nullptr;
	info->d_coeffs = field->dev.coeffs;
	{
		std::lock_guard<std::mutex> lock(field->copy_mutex);
		info->has_cell_major = field->d_cell_major != nullptr;
		info->band_rows = field->band_rows;
		info->has_tile_major = field->d_tile_major != nullptr;
		info->immutable = field->immutable;
	}
	{
		std::lock_guard<std::mutex> lock(field->host_mutex);
		info->host_copy_pending = field->host_job != nullptr;
	}
	uint64_t bytes = field->owned[0] ? field->n_coeffs * sizeof(double) : 0;
	if (field->owned[1]) bytes += field->n_rows * 32 * sizeof(uint32_t);
	if (field->owned[2]) bytes += dg_grid_n_cells(&field->grid) * sizeof(uint32_t);
	if (info->has_cell_major) bytes += field->n_rows * 256ull;
	if (field->d_band_rows) bytes += info->band_rows * 256ull + (field->n_rows + 63) / 64 * 12ull;
	if (info->has_tile_major)
		bytes += (uint64_t)field->dev.ntile[0] * field->dev.ntile[1] * field->dev.ntile[2] * dg::kTmNodes * sizeof(double);
	info->device_bytes = bytes;
	return DG_OK;
}

// ---- reduceField -------------------------------------------------------------------------------------------
} // extern "C"
struct dg_reduction
{
	dg::ReduceResult r;
	uint64_t n_cells = 0;
	int device = -1;
	dg_grid_desc grid;
};
extern "C"
{

dg_status dg_reduce_field(const dg_grid_desc* grid, const double* coeffs, uint64_t n_coeffs, int closed, double lo,
						  double hi, double offset, dg_reduction** out)
{
	if (!out)
		return fail(DG_ERR_INVALID, "out is null");
	*out = nullptr;
	if (!grid || !coeffs)
		return fail(DG_ERR_INVALID, "null argument");
	if (!valid_grid(grid))
		return fail(DG_ERR_INVALID, "invalid grid");
	if (n_coeffs != dg_grid_n_nodes(grid))
		return fail(DG_ERR_INVALID, "dg_reduce_field needs an unreduced field: %llu coefficients, got %llu",
					(unsigned long long)dg_grid_n_nodes(grid), (unsigned long long)n_coeffs);
	dg_status s = require_device();
	if (s != DG_OK)
		return s;
	dg_reduction* red = new (std::nothrow) dg_reduction;
	if (!red)
		return fail(DG_ERR_ALLOC, "host allocation failed");
	red->n_cells = dg_grid_n_cells(grid);
	red->grid = *grid;
	(void)hipGetDevice(&red->device);
	void* d_c = nullptr;
	hipError_t e = hipMalloc(&d_c, n_coeffs * sizeof(double));
	if (e == hipSuccess) e = hipMemcpy(d_c, coeffs, n_coeffs * sizeof(double), hipMemcpyHostToDevice);
	if (e == hipSuccess)
	{
		dg::ReducePredicate P;
		P.lo = lo;
		P.hi = hi;
		P.offset = offset;
		P.closed = closed ? 1 : 0;
		e = dg::reduce_field_device(grid->resolution, grid->domain_min, grid->cell_size, grid->inv_cell_size,
									static_cast<const double*>(d_c), n_coeffs, P, red->r, nullptr);
	}
	if (d_c) (void)hipFree(d_c);
	if (e != hipSuccess)
	{
		dg_reduction_destroy(red);
		return fail(e == hipErrorOutOfMemory ? DG_ERR_ALLOC : DG_ERR_HIP, "dg_reduce_field: %s", hipGetErrorString(e));
	}
	*out = red;
	return DG_OK;
}

// The same on the coefficients a field handle already holds on the device: no upload.
dg_status dg_reduce_field_device(const dg_field* field, int closed, double lo, double hi, double offset, dg_reduction** out)
{
	if (!out)
		return fail(DG_ERR_INVALID, "out is null");
	*out = nullptr;
	if (!field)
		return fail(DG_ERR_INVALID, "null argument");
	if (field->dev.cells || field->dev.cell_map || field->n_coeffs != dg_grid_n_nodes(&field->grid))
		return fail(DG_ERR_INVALID, "dg_reduce_field_device needs an unreduced field");
	DG_ON_DEVICE_OF(field);
	dg_reduction* red = new (std::nothrow) dg_reduction;
	if (!red)
		return fail(DG_ERR_ALLOC, "host allocation failed");
	red->n_cells = dg_grid_n_cells(&field->grid);
	red->device = field->device;
	red->grid = field->grid;
	hipError_t e = wait_produced(field, nullptr);
	if (e == hipSuccess)
	{
		dg::ReducePredicate P;
		P.lo = lo;
		P.hi = hi;
		P.offset = offset;
		P.closed = closed ? 1 : 0;
		e = dg::reduce_field_device(field->grid.resolution, field->grid.domain_min, field->grid.cell_size, field->grid.inv_cell_size,
									field->dev.coeffs, field->n_coeffs, P, red->r, nullptr);
	}
	if (e != hipSuccess)
	{
		dg_reduction_destroy(red);
		return fail(e == hipErrorOutOfMemory ? DG_ERR_ALLOC : DG_ERR_HIP, "dg_reduce_field_device: %s", hipGetErrorString(e));
	}
	*out = red;
	return DG_OK;
}

// The reduced field as a handle of its own: the reduction's device arrays (coefficients, cell rows, cell map) change
// owner, nothing is copied; the reduction can still be destroyed (and no longer fetched) afterwards.
dg_status dg_reduction_to_field(dg_reduction* r, dg_field** out)
{
	if (!out)
		return fail(DG_ERR_INVALID, "out is null");
	*out = nullptr;
	if (!r)
		return fail(DG_ERR_INVALID, "null argument");
	if (r->r.tied_keys)
		return fail(DG_ERR_INVALID, "tied Morton keys: the node order is not unique, run the host algorithm");
	if (!r->r.d_cell_map)
		return fail(DG_ERR_INVALID, "the reduction's arrays were already moved into a field");
	DG_ON_DEVICE_OF(r);
	const dg_status st = dg_field_attach_device(&r->grid, static_cast<const double*>(r->r.d_coeffs), r->r.n_nodes_out,
												 r->r.n_rows ? static_cast<const uint32_t*>(r->r.d_cells) : nullptr, r->r.n_rows,
												 static_cast<const uint32_t*>(r->r.d_cell_map), out);
	if (st != DG_OK)
		return st;
	(*out)->owned[0] = r->r.d_coeffs;
	(*out)->owned[1] = r->r.d_cells;
	(*out)->owned[2] = r->r.d_cell_map;
	r->r.d_coeffs = nullptr;
	r->r.d_cells = nullptr;
	r->r.d_cell_map = nullptr;
	return DG_OK;
}

dg_status dg_reduction_info(const dg_reduction* r, uint64_t* n_coeffs_out, uint64_t* n_cell_rows, int* tied_keys)
{
	if (!r)
		return fail(DG_ERR_INVALID, "null argument");
	if (n_coeffs_out) *n_coeffs_out = r->r.n_nodes_out;
	if (n_cell_rows) *n_cell_rows = r->r.n_rows;
	if (tied_keys) *tied_keys = r->r.tied_keys;
	return DG_OK;
}

dg_status dg_reduction_fetch(const dg_reduction* r, double* coeffs, uint32_t* cells, uint32_t* cell_map)
{
	if (!r || !cell_map || (r->r.n_nodes_out && !coeffs) || (r->r.n_rows && !cells))
		return fail(DG_ERR_INVALID, "null argument");
	if (r->r.tied_keys)
		return fail(DG_ERR_INVALID, "tied Morton keys: the node order is not unique, run the host algorithm");
	if (!r->r.d_cell_map)
		return fail(DG_ERR_INVALID, "the reduction's arrays were moved into a field (dg_reduction_to_field): fetch first");
	DG_ON_DEVICE_OF(r);
	if (r->r.n_nodes_out)
		DG_HIP(hipMemcpy(coeffs, r->r.d_coeffs, r->r.n_nodes_out * sizeof(double), hipMemcpyDeviceToHost));
	if (r->r.n_rows)
		DG_HIP(hipMemcpy(cells, r->r.d_cells, r->r.n_rows * 32 * sizeof(uint32_t), hipMemcpyDeviceToHost));
	DG_HIP(hipMemcpy(cell_map, r->r.d_cell_map, r->n_cells * sizeof(uint32_t), hipMemcpyDeviceToHost));
	return DG_OK;
}

void dg_reduction_destroy(dg_reduction* r)
{
	if (!r)
		return;
	DeviceGuard guard(r->device);
	if (r->r.d_coeffs) (void)hipFree(r->r.d_coeffs);
	if (r->r.d_cells) (void)hipFree(r->r.d_cells);
	if (r->r.d_cell_map) (void)hipFree(r->r.d_cell_map);
	delete r;
}

// ---- K3 ---------------------------------------------------------------------------------------------------
dg_status dg_density_map_nodes_device(dg_field* sdf, double support_radius, double rho0, int band_predicate,
									  uint64_t node_begin, uint64_t node_end, const uint8_t* d_pred_mask,
									  double* d_out, void* stream)
{
	TraceRange trace_range_("dg K3 density_map");
	if (!sdf || !d_out)
		return fail(DG_ERR_INVALID, "null argument");
	if (!(support_radius > 0.0))
		return fail(DG_ERR_INVALID, "support radius must be positive");
	const uint64_t total = dg_grid_n_nodes(&sdf->grid);
	if (node_begin > node_end || node_end > total)
		return fail(DG_ERR_INVALID, "node range outside the lattice");
	if (node_begin == node_end)
		return DG_OK;
	DG_ON_DEVICE_OF(sdf);
	hipStream_t st = static_cast<hipStream_t>(stream);
	// One handle serves K2 and K3: K2 may build a cell-major copy on this handle at any moment (from another thread,
	// on its first large batch), so the device view is snapshotted under the lock -- and the cell-major copy is NOT
	// taken along: in brick order neighbouring cells share most of their nodes, cell-major rows share none (K3 is
	// 1.8 x slower through it than through its tile copy).
	dg::FieldDev dev;
	hipEvent_t tiles_ready = nullptr;
	{
		std::lock_guard<std::mutex> lock(sdf->copy_mutex);
		dev = sdf->dev;
		dev.cell_major = nullptr;
		if (dev.tile_major)
			tiles_ready = sdf->tile_major_ready;
	}
	if (tiles_ready)
		DG_HIP(hipStreamWaitEvent(st, tiles_ready, 0));
	DG_HIP(wait_produced(sdf, st));
	dg::DensityParams P;
	std::vector<double> w;
	dg::init_density_params(P, support_radius, rho0, sdf->grid.cell_size, band_predicate, w);
	// One immutable device table of kernel values per support radius: uploaded (blocking) before the
	// first launch that uses it and never written again, so launches on other streams or threads that
	// are still reading a table cannot see it change.
	{
		std::lock_guard<std::mutex> lock(sdf->wtab_mutex);
		auto it = sdf->wtabs.find(support_radius);
		if (it == sdf->wtabs.end())
		{
			void* d_w = nullptr;
			DG_HIP(hipMalloc(&d_w, w.size() * sizeof(double)));
			const hipError_t e = hipMemcpy(d_w, w.data(), w.size() * sizeof(double), hipMemcpyHostToDevice);
			if (e != hipSuccess)
			{
				(void)hipFree(d_w);
				return fail(DG_ERR_HIP, "kernel table upload: %s", hipGetErrorString(e));
			}
			it = sdf->wtabs.emplace(support_radius, d_w).first;
		}
		P.wtab = static_cast<const double*>(it->second);
	}
	// K1's lattice decomposition: one wave per 4x4x4 brick of nodes
	dg::SampleParams L;
	dg::MeshDev none;
	std::memset(&none, 0, sizeof(none));
	dg::init_params(L, none, sdf->grid.domain_min, sdf->grid.cell_size, 0);
	dg::layout_range(L, sdf->grid.resolution, node_begin, node_end);
	L.mask = d_pred_mask;
	L.out = d_out;
	L.brick_blocking = force_int("k3_blocked", 1, 0, 1);
	const bool unreduced_field = dev.cells == nullptr && dev.cell_map == nullptr;
	// An eighth of the lattice or more of an unreduced field: k_density_cells -- one lane per lattice POINT with its seven nodes
	// (dg_density_cells.h: 3 cell fetches per 7 nodes and quadrature point, one sweep of the field for all node classes) on the
	// x-major copy of the Y / Z classes (x fastest, 0.57 x the field).  The copy and the per-cell "no value" bits are
	// stream-ordered scratch of this launch.  Everything else (reduced fields, short node ranges, classes beyond 32-bit
	// offsets; DG_FORCE=k3_cells=0) is the brick kernel.
	// (scratch slots go back to their pools on EVERY way out of this function: a slot left busy is never handed out again)
	struct SlotGuard
	{
		ScratchPool& pool;
		int& idx;
		hipStream_t st;
		~SlotGuard() { pool.release(idx, st); }
	};
	int rows_idx = -1;
	SlotGuard rows_guard{sdf->tile_scratch, rows_idx, st};
	if (unreduced_field && (node_end - node_begin) * 8 >= total && dg::k3c_geometry_fits(dev.res) && force_int("k3_cells", 1, 0, 1) != 0)
	{
		const size_t copy_bytes = ((size_t)dg::xmajor_doubles(dev.res) * sizeof(double) + 255) & ~(size_t)255;
		const size_t flag_bytes = (size_t)dev.res[2] * dev.res[1] * dg::xmajor_flag_words(dev.res) * sizeof(uint64_t);
		void* d_rows = nullptr;
		rows_idx = sdf->tile_scratch.acquire(copy_bytes + flag_bytes, st, &d_rows);
		if (rows_idx >= 0 && dg::launch_xmajor_copy(dev, static_cast<double*>(d_rows), st) == hipSuccess)
		{
			dev.xmajor = static_cast<const double*>(d_rows);
			dev.xmajor_flags = reinterpret_cast<const uint64_t*>(static_cast<const char*>(d_rows) + copy_bytes);
			// waves along x / y / z of the blocks consecutive wave ids fill, fitted by counter (profiles/r04_k3_blocks_pmc.txt, 256^3,
			// kernel ms / L2 hit rate): 1 x 16 x 8 472 / 0.890, 1 x 12 x 8 466 / 0.894, 1 x 8 x 8 468 / 0.853, 1 x 6 x 22 462 / 0.899,
			// 1 x 3 x 43 467 / 0.864, 1 x 2 x 64 465 / 0.816, 1 x 8 x 16 489 / 0.911 -- moderately tall in z: the quadrature's
			// innermost loop sweeps z, so the waves of a block read what their z-neighbours read a step ago
			const uint32_t block[3] = {(uint32_t)force_int("k3_rb0", 1, 1, 256), (uint32_t)force_int("k3_rb1", 6, 1, 256),
									   (uint32_t)force_int("k3_rb2", 22, 1, 256)};
			dg::layout_density_cells(P, L, sdf->grid.resolution, block);
			P.row_node_begin = node_begin;
			P.row_node_end = node_end;
		}
		else
		{
			(void)hipGetLastError(); // without the copy then
			sdf->tile_scratch.release(rows_idx, st);
			rows_idx = -1;
		}
	}
	// zero-weight quadrature points are skipped unless the field holds non-finite / huge values (checked
	// on the device before every launch: an attached device array may have changed); DG_FORCE=k3_skip=0: never
	int flag_idx = -1; // the flag k_field_check writes belongs to this launch (stream-ordered scratch)
	SlotGuard flag_guard{sdf->flag_scratch, flag_idx, st};
	const bool skip_points = force_int("k3_skip", 1, 0, 1) != 0 && support_radius >= 1.0e-12;
	// (always: besides the values that forbid the skip, k_field_check reports whether the field holds "no value" coefficients at all)
	{
		void* d_flag = nullptr;
		flag_idx = sdf->flag_scratch.acquire(256, st, &d_flag);
		if (flag_idx < 0)
			return fail(DG_ERR_ALLOC, "device allocation failed");
		P.skip_mode = skip_points ? 2 : 0;
		P.unsafe = static_cast<const uint32_t*>(d_flag);
	}
	// Unreduced field without a tile-major copy: build one for this launch (stream-ordered scratch, built from
	// the coefficients as they are NOW -- an attached device array may have changed since the last call).  One
	// pass over the field against 1 + 4096 interpolations per integrated node: 128^3 161 -> 135 ms
	// (DG_FORCE=k3_tiles=0: off).  Launches over a small part of the lattice are not worth the pass.
	int tile_idx = -1;
	SlotGuard tile_guard{sdf->tile_scratch, tile_idx, st};
	if (rows_idx < 0 && dev.tile_major == nullptr && dev.cell_major == nullptr && dev.cells == nullptr && dev.cell_map == nullptr &&
		force_int("k3_tiles", 1, 0, 1) != 0 && (node_end - node_begin) * 8 >= total)
	{
		const uint64_t n_tiles = (uint64_t)dev.ntile[0] * dev.ntile[1] * dev.ntile[2];
		void* d_tiles = nullptr;
		tile_idx = sdf->tile_scratch.acquire(n_tiles * dg::kTmNodes * sizeof(double), st, &d_tiles);
		if (tile_idx >= 0)
		{
			const hipError_t te = dg::launch_expand_tiles(dev, n_tiles, static_cast<double*>(d_tiles), st);
			if (te == hipSuccess)
				dev.tile_major = static_cast<const double*>(d_tiles);
			else
				(void)hipGetLastError(); // without the copy then
		}
	}
	DG_HIP(dg::launch_density_bricks(L, dev, sdf->n_coeffs, P, st));
	return DG_OK;
}

dg_status dg_interpolate_batch_device(const dg_field* field, const double* d_xyz, uint64_t n, double* d_phi,
									  double* d_grad, void* stream)
{
	TraceRange trace_range_("dg K2 interpolate");
	if (!field || (n && (!d_xyz || !d_phi)))
		return fail(DG_ERR_INVALID, "null argument");
	if (n == 0)
		return DG_OK;
	DG_ON_DEVICE_OF(field);
	hipStream_t st = static_cast<hipStream_t>(stream);
	const bool big = n >= (1u << 18) && field->n_coeffs * sizeof(double) >= (32u << 20);
	dg::FieldDev dev;
	hipEvent_t copy_ready = nullptr, band_ready = nullptr;
	{
		std::lock_guard<std::mutex> lock(field->copy_mutex);
		// A field whose coefficients this library owns (dg_field_create: they cannot change) that gets a large batch and
		// has no copy yet: build the cell-major copy now, on this stream, once (5 ms and 4.3 GB at 256^3 against 0.7 ms
		// saved on every 10 M queries from then on).  Not for attached device arrays (they may change between calls: their
		// owner calls dg_field_build_cell_major), not beyond DG_K2_AUTO_CELL_MAJOR_MB (default 16384; 0: never) or a
		// quarter of the free device memory, not after dg_field_drop_cell_major.
		if (big && (field->owned[0] != nullptr || field->immutable) && !field->d_cell_major && !field->d_tile_major && !field->auto_copy_tried)
		{
			field->auto_copy_tried = true;
			const uint64_t need = field->n_rows * 256ull;
			const uint64_t cap = (uint64_t)env_int("DG_K2_AUTO_CELL_MAJOR_MB", 16384, 0, 1 << 20) << 20;
			size_t free_b = 0, total_b = 0;
			if (need <= cap && hipMemGetInfo(&free_b, &total_b) == hipSuccess && need <= free_b / 4)
			{
				if (build_cell_major_locked(const_cast<dg_field*>(field), stream) != DG_OK)
					(void)hipGetLastError(); // without the copy then
			}
			else
				(void)hipGetLastError();
		}
		dev = field->dev;
		// the event of the copy the kernels below will actually read (field_mode(): tiles before cells)
		if (field->d_tile_major)
			copy_ready = field->tile_major_ready;
		else if (field->d_cell_major)
			copy_ready = field->cell_major_ready;
		band_ready = field->d_band_rows ? field->band_ready : nullptr;
	}
	DG_HIP(wait_produced(field, st));
	if (copy_ready) // the copy may still be being built on another stream
		DG_HIP(hipStreamWaitEvent(st, copy_ready, 0));
	// A field with a cell-major copy: one contiguous row per query, fetched cooperatively -- the order of the queries
	// does not matter, nothing is sorted (DG_FORCE=k2_rows=0: the binned / per-lane kernels on the copy, as in round 1).
	if (dev.cell_major != nullptr && dev.tile_major == nullptr && force_int("k2_rows", 1, 0, 1) != 0)
	{
		DG_HIP(dg::launch_interpolate_rows(dev, d_xyz, n, d_phi, d_grad, st));
		return DG_OK;
	}
	// A field with a band-limited cell-major copy (and no full one): rows for the queries inside the band, the plain gather
	// for the others, in one launch, queries in any order (DG_FORCE=k2_band=0: ignore the copy)
	// Routed by measurement: every large batch is probed (1024 of its queries: how many have a row in the copy?), and the verdict
	// of the field's PREVIOUS large batch decides -- a batch of which less than three quarters map into the band is faster through
	// the sorted gather of the binned path below (uniform queries over a shell copy, 57 % mapped: 8.3 against 5.5 Gq/s; the band
	// kernel wins from ~78 % mapped on: 0.055 ns per mapped query, 0.35 ns per unmapped one, against 0.12 ns binned).
	// DG_FORCE=k2_band=2: the band kernel whatever the probe said.
	const int band_mode = force_int("k2_band", 1, 0, 2);
	if (dev.band_rows != nullptr && dev.cell_major == nullptr && dev.tile_major == nullptr && band_mode != 0)
	{
		if (band_ready)
			DG_HIP(hipStreamWaitEvent(st, band_ready, 0));
		bool through_band = true;
		if (big)
		{
			if (field->band_probe_host == nullptr)
			{
				void* p = nullptr;
				if (hipHostMalloc(&p, 2 * sizeof(uint32_t), hipHostMallocDefault) == hipSuccess)
				{
					field->band_probe_host = static_cast<uint32_t*>(p);
					field->band_probe_host[0] = field->band_probe_host[1] = 0u; // nothing known yet: the band kernel
				}
				else
					(void)hipGetLastError();
			}
			if (field->band_probe_host != nullptr)
			{
				const uint32_t valid = reinterpret_cast<volatile uint32_t*>(field->band_probe_host)[0];
				const uint32_t mapped = reinterpret_cast<volatile uint32_t*>(field->band_probe_host)[1];
				through_band = band_mode == 2 || valid == 0u || 4ull * mapped >= 3ull * valid;
				DG_HIP(dg::launch_band_probe(dev, d_xyz, n, field->band_probe_host, st));
			}
		}
		if (through_band)
		{
			DG_HIP(dg::launch_interpolate_band(dev, d_xyz, n, d_phi, d_grad, st));
			return DG_OK;
		}
	}
	// Large batches against a field that does not fit the L2s go through the binned path (queries in
	// arbitrary order are then processed tile by tile; ordered inputs are detected on the device and
	// run as they are).  DG_FORCE=k2_binning=0 switches it off, =2 forces it for any size.
	const int binning = force_int("k2_binning", 1, 0, 2);
	// Unreduced field in the reference layout, no copy of it (the default for every attached device array): counting sort by tile
	// of 8^3 cells + the gather that stages a tile in LDS once for all its queries (round 6; dg_kernels.h: TileBin).
	// DG_FORCE=k2_tiles=0: the per-lane gather behind the radix sort of rounds 1-5.
	if (binning != 0 && (big || binning == 2) && n < 0xffffffffull && dg::field_mode(dev) == dg::kFieldClosed && dev.cell_map == nullptr &&
		force_int("k2_tiles", 1, 0, 2) != 0)
	{
		dg::TileBin B;
		std::memset(&B, 0, sizeof(B));
		B.shape = 0;
		const uint32_t key_bits = dg::stage_key_bits(dev.res, B.shape, B.tdims, B.tlog);
		// (k2_tiles=2: whatever the batch's size)
		const bool dense = n >= (uint64_t)(d_grad ? dg::kStageMinPerTileGrad : dg::kStageMinPerTile) * B.tdims[0] * B.tdims[1] * B.tdims[2] ||
						   force_int("k2_tiles", 1, 0, 2) == 2;
		if (dense && key_bits <= dg::kStageMaxBits && B.tdims[0] <= 1024u && B.tdims[1] <= 1024u && B.tdims[2] <= 1024u)
		{
			if (field->bin_flag_host == nullptr)
			{
				void* p = nullptr;
				if (hipHostMalloc(&p, sizeof(uint32_t), hipHostMallocDefault) == hipSuccess)
				{
					field->bin_flag_host = static_cast<uint32_t*>(p);
					*field->bin_flag_host = 1u; // nothing known yet: assume the first batch is unordered
				}
				else
					(void)hipGetLastError();
			}
			size_t off[8];
			void* mem = nullptr;
			// gradient batches: value + gradient leave the gather as one aligned 32-byte store per query into a scratch array and a
			// streaming pass splits it into the caller's arrays (DG_FORCE=k2_grad_packed=0: the 8 + 24 byte stores of rounds 1-5)
			const bool packed = d_grad != nullptr && force_int("k2_grad_packed", 1, 0, 1) != 0;
			const int idx = field->bin_flag_host ? field->scratch.acquire(dg::tile_bin_bytes(1u << key_bits, n, off, packed), st, &mem) : -1;
			if (idx >= 0)
			{
				const dg::TileBin shape_of = B;
				dg::tile_bin_assign(B, mem, off, key_bits, n);
				B.shape = shape_of.shape;
				if (packed)
					B.packed = reinterpret_cast<double*>(static_cast<char*>(mem) + off[7]);
				B.flag_host = field->bin_flag_host;
				B.sort_launched = *reinterpret_cast<volatile uint32_t*>(field->bin_flag_host) != 0u ? 1 : 0;
				hipError_t e = dg::launch_interpolate_tiles(dev, d_xyz, n, d_phi, d_grad, B, (uint32_t)force_int("k2_tile_chunk", 64, 0, 4096), st);
				if (e == hipSuccess && B.sort_launched == 0) // predicted ordered: the queries as they came
					e = dg::launch_interpolate(dev, d_xyz, n, d_phi, d_grad, st);
				field->scratch.release(idx, st);
				DG_HIP(e);
				return DG_OK;
			}
		}
	}
	if (binning != 0 && (big || binning == 2) && n < 0xffffffffull)
	{
		dg::BinScratch S;
		const int idx = acquire_bin_scratch(field->scratch, &field->bin_flag_host, dg::field_tiles(dev, dg::kSortCells), n, st, S);
		if (idx >= 0)
		{
			const hipError_t e = dg::launch_interpolate_binned(dev, d_xyz, n, d_phi, d_grad, S, st);
			field->scratch.release(idx, st);
			DG_HIP(e);
			return DG_OK;
		}
	}
	DG_HIP(dg::launch_interpolate(dev, d_xyz, n, d_phi, d_grad, st));
	return DG_OK;
}

} // extern "C"
