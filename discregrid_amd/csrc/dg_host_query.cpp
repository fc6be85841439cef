// dg_host_query.cpp -- TriangleMeshDistance::signed_distance for ONE point, on the host
// (reference: discregrid/include/Discregrid/geometry/TriangleMeshDistance.h:269-328, declared const and
// thread safe at :188,199 and called per node or per particle from user code such as the lambda of
// cmd/generate_sdf/main.cpp:97-101).  A kernel launch per point costs 100 us; this evaluates the point
// where the caller is: the same BVH (the sibling-pair records and triangle packets dg_mesh_create
// uploaded, kept in host memory as well) walked by the SAME traversal template the kernels instantiate
// (dg_traverse.h: packet_walk + ExactWalk, here with a wave of one lane), the same conservative float bounds,
// the same double-precision triangle test and epilogue -- so the distance has the bits K1 / K1p produce.  Read-only on immutable data: no locks, any number of concurrent callers.
//
// This is the per-point evaluator of the host API (like the scalar interpolate of dg_lattice.h), not a
// substitute for the kernels: batches go to dg_signed_distance / dg_sdf_sample_nodes.  It is also all a HOST-ONLY mesh
// handle can do (dg_mesh_create without a HIP device, dg_mesh_device() == -1): the C++ host API then runs the reference's
// OpenMP node loop over it.  Compile with -ffp-contract=off.
#include "dg_capi_internal.h"
#include "dg_host_query.h"


extern "C" dg_status dg_signed_distance_point(const dg_mesh* mesh, const double xyz[3], double* dist, int32_t* tri, int32_t* entity,
											  double* nearest)
{
	if (!mesh || !xyz || !dist)
		return fail(DG_ERR_INVALID, "null argument");
	dg::LaneResult r;
	if (!dg::host::signed_distance_point(mesh->host_view, xyz[0], xyz[1], xyz[2], r))
	{
		*dist = DG_NO_VALUE;
		if (tri) *tri = -1;
		if (entity) *entity = -1;
		return DG_OK;
	}
	*dist = r.signed_dist;
	if (tri) *tri = r.tri_id;
	if (entity) *entity = r.entity;
	if (nearest)
	{
		nearest[0] = r.nearest[0];
		nearest[1] = r.nearest[1];
		nearest[2] = r.nearest[2];
	}
	return DG_OK;
}
