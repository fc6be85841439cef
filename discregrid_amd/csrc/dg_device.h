// dg_device.h -- device-side helpers shared by the kernel translation units (dg_kernels_k1.hip, _k2.hip, _k3.hip, _aux.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dg
{
namespace
{
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v16i __attribute__((ext_vector_type(16)));
#define DG_CONST_AS __attribute__((address_space(4)))

// Wave-uniform loads through the scalar data cache.  The address must be uniform across the
// wave (callers pass indices that went through readfirstlane); the data is immutable for the
// lifetime of the kernel.
__device__ __forceinline__ v8i sload8(const void* p)
{
	return *(const DG_CONST_AS v8i*)(uintptr_t)p;
}
__device__ __forceinline__ v16i sload16(const void* p)
{
	return *(const DG_CONST_AS v16i*)(uintptr_t)p;
}
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ double pack_double(int lo, int hi)
{
	return __hiloint2double(hi, lo);
}
} // namespace
} // namespace dg
