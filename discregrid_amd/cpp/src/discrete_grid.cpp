// DiscreteGrid bookkeeping (reference: discregrid/include/Discregrid/discrete_grid.hpp:22-29,
// discregrid/src/discrete_grid.cpp:9-38).  Element-wise arithmetic is written out so that the
// results do not depend on how an expression-template library associates them.
#include <Discregrid/discrete_grid.hpp>

namespace Discregrid
{

DiscreteGrid::DiscreteGrid(Eigen::AlignedBox3d const& domain, std::array<unsigned int, 3> const& resolution)
	: m_domain(domain), m_resolution(resolution), m_n_fields(0u)
{
	for (int d = 0; d < 3; ++d)
	{
		m_cell_size[d] = (domain.max()[d] - domain.min()[d]) / static_cast<double>(resolution[d]);
		m_inv_cell_size[d] = 1.0 / m_cell_size[d];
	}
	m_n_cells = static_cast<std::size_t>(resolution[0] * (resolution[1] * resolution[2]));
}

DiscreteGrid::MultiIndex DiscreteGrid::singleToMultiIndex(unsigned int l) const
{
	const unsigned int n01 = m_resolution[0] * m_resolution[1];
	const unsigned int k = l / n01;
	const unsigned int r = l % n01;
	return {{r % m_resolution[0], r / m_resolution[0], k}};
}

unsigned int DiscreteGrid::multiToSingleIndex(MultiIndex const& ijk) const
{
	return m_resolution[1] * m_resolution[0] * ijk[2] + m_resolution[0] * ijk[1] + ijk[0];
}

Eigen::AlignedBox3d DiscreteGrid::subdomain(MultiIndex const& ijk) const
{
	Eigen::Vector3d lo, hi;
	for (int d = 0; d < 3; ++d)
	{
		lo[d] = m_domain.min()[d] + static_cast<double>(ijk[d]) * m_cell_size[d];
		hi[d] = lo[d] + m_cell_size[d];
	}
	return Eigen::AlignedBox3d(lo, hi);
}

Eigen::AlignedBox3d DiscreteGrid::subdomain(unsigned int l) const { return subdomain(singleToMultiIndex(l)); }

} // namespace Discregrid
