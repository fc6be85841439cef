// Discregrid::FieldVector -- storage of one field's coefficients (the inner vectors of the
// reference's `std::vector<std::vector<double>> m_nodes`, cubic_lagrange_discrete_grid.hpp:69).
//
// A std::vector<double> with an allocator that (a) leaves new elements UNINITIALISED when they are
// default-constructed -- resize(n) on 0.95 GB (256^3) otherwise spends 160 ms writing zeros from one
// thread, eight times what the GPU needs to compute the values that overwrite them -- and (b) hands
// out 2 MiB-aligned, transparent-huge-page-advised memory for large arrays, which the GPU's copy
// engine can be pointed at directly (pinning huge pages is 15x cheaper than pinning 4 KiB pages).
// Everything else (iteration, data(), size(), swap, copy) is std::vector.  Elements that are
// explicitly value-constructed (`FieldVector v(n, 0.0)`, push_back) behave as usual.
#pragma once

#include <cstddef>
#include <cstdlib>
#include <new>
#include <utility>
#include <vector>

#if defined(__linux__)
#include <sys/mman.h>
#endif

namespace Discregrid
{

template <class T>
struct FieldAllocator
{
	using value_type = T;
	FieldAllocator() noexcept = default;
	template <class U>
	FieldAllocator(FieldAllocator<U> const&) noexcept {}

	T* allocate(std::size_t n)
	{
		const std::size_t bytes = n * sizeof(T);
		const std::size_t huge = std::size_t(1) << 21;
		void* p = nullptr;
		if (bytes >= 2 * huge)
		{
			if (posix_memalign(&p, huge, bytes) != 0)
				throw std::bad_alloc();
#if defined(__linux__) && defined(MADV_HUGEPAGE)
			(void)madvise(p, bytes, MADV_HUGEPAGE);
#endif
		}
		else if (posix_memalign(&p, 64, bytes ? bytes : 64) != 0)
			throw std::bad_alloc();
		return static_cast<T*>(p);
	}
	void deallocate(T* p, std::size_t) noexcept { std::free(p); }

	template <class U>
	void construct(U* p) noexcept(noexcept(::new (static_cast<void*>(p)) U))
	{
		::new (static_cast<void*>(p)) U; // default-initialisation: no zero fill for arithmetic types
	}
	template <class U, class... Args>
	void construct(U* p, Args&&... args)
	{
		::new (static_cast<void*>(p)) U(std::forward<Args>(args)...);
	}
	template <class U>
	bool operator==(FieldAllocator<U> const&) const noexcept { return true; }
	template <class U>
	bool operator!=(FieldAllocator<U> const&) const noexcept { return false; }
};

using FieldVector = std::vector<double, FieldAllocator<double>>;

} // namespace Discregrid
