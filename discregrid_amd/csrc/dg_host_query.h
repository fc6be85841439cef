// dg_host_query.h -- TriangleMeshDistance::signed_distance for ONE point on the calling host thread
// (see dg_host_query.cpp for the ABI entry point and the rationale).  Header-only so that the CPU test
// emulator can run exactly this function; the product calls it from dg_signed_distance_point only.
#pragma once
#include <cmath>
#include "dg_build.h"
#include "dg_kernels.h"

namespace dg
{
namespace host
{
struct HostSqrt
{
	double operator()(double x) const { return std::sqrt(x); } // IEEE, correctly rounded
};

// Near-first depth-first walk with an explicit stack: at an inner node one record gives the lower
// bounds of both children; the nearer one is entered, the other postponed with its bound and re-tested
// against the by then tighter best when it is popped.
inline bool signed_distance_point(const MeshBuild& B, double px, double py, double pz, LaneResult& out)
{
	LaneQuery q;
	init_query(B.origin, B.mesh_l1, true, px, py, pz, q);
	struct Entry
	{
		int32_t info;
		float lb;
	};
	Entry stack[2 * kStackDepth];
	int sp = 0;
	int32_t cur = B.root_info;
	float lbcur = 0.0f;
	while (true)
	{
		bool descend = false;
		if (cur < 0)
		{
			const unsigned code = ~(unsigned)cur;
			const int first = (int)(code >> kLeafBits), cnt = (int)(code & (unsigned)(kMaxLeaf - 1)) + 1;
			for (int g = 0; g < cnt; g += 2)
			{
				const f2 lb = pair_lb2(&B.tri_pairs[(size_t)(first + g) >> 1].f[0][0], q.fp);
				for (int side = 0; side < 2; ++side)
				{
					if (!(fmax2(side == 0 ? lb.x : lb.y, lbcur) < q.bestf))
						continue;
					const int t = first + g + side;
					const Hit h = tri_closest<false>(B.tris[(size_t)t], px, py, pz);
					offer(q, h.d2, t);
				}
			}
		}
		else
		{
			const PairRec& r = B.pairs[(size_t)cur];
			f2 cd;
			const f2 lb = pair_lb2(&r.f[0][0], q.fp, &cd);
			const bool hl = lb.x < q.bestf, hr = lb.y < q.bestf;
			if (hl || hr)
			{
				bool left = hl;
				if (hl && hr)
				{
					left = cd.x <= cd.y; // nearer box centre first (as the kernels)
					if (sp < 2 * kStackDepth)
					{
						stack[sp].info = left ? r.info[1] : r.info[0];
						stack[sp].lb = left ? lb.y : lb.x;
						++sp;
					}
				}
				cur = left ? r.info[0] : r.info[1];
				lbcur = left ? lb.x : lb.y;
				descend = true;
			}
		}
		if (descend)
			continue;
		bool found = false;
		while (sp > 0)
		{
			--sp;
			if (stack[sp].lb < q.bestf)
			{
				cur = stack[sp].info;
				lbcur = stack[sp].lb;
				found = true;
				break;
			}
		}
		if (!found)
			break;
	}
	if (q.best_tri < 0)
		return false; // a NaN point: no comparison ever succeeds (the kernels write "no value" as well)
	out = finish_query(B.tris.data(), B.pn.data(), q, HostSqrt());
	return true;
}
} // namespace host
} // namespace dg
