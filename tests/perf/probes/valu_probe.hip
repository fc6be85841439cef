// valu_probe.hip -- issue cost of the VALU instruction kinds K1 is built from, on gfx950.
// Every wave runs a long unrolled stream of one instruction kind on 8 independent accumulators;
// with W waves per SIMD the SIMD's cycles per instruction = (cycles a wave took) / (instructions
// per wave * W).  Test tooling (numbers quoted in DESIGN.md), not part of the product.
//   hipcc --offload-arch=gfx950 -O3 valu_probe.hip -o valu_probe && ./valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef float f2 __attribute__((ext_vector_type(2)));

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int KIND>
__global__ __launch_bounds__(256) void probe(int iters, float seed, long long* cycles, float* sink)
{
	float a[8];
	f2 p[8];
	double d[8];
	for (int i = 0; i < 8; ++i)
	{
		a[i] = seed + i + threadIdx.x;
		p[i].x = seed + i;
		p[i].y = seed - i;
		d[i] = seed + 2 * i + threadIdx.x;
	}
	const float b = seed * 0.5f, c = seed * 0.25f;
	const f2 pb = {b, c}, pc = {c, b};
	const double db = b, dc = c;
	const long long t0 = clock64();
	for (int it = 0; it < iters; ++it)
	{
#define OP_F32(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
#define OP_PK(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(pb), "v"(pc));
#define OP_PKADD(i) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(p[i]) : "v"(pb));
#define OP_F64(i) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[i]) : "v"(db), "v"(dc));
#define OP_MUL64(i) asm volatile("v_mul_f64 %0, %1, %0" : "+v"(d[i]) : "v"(db));
#define OP_ADD64(i) asm volatile("v_add_f64 %0, %1, %0" : "+v"(d[i]) : "v"(db));
#define OP_MAX32(i) asm volatile("v_max_f32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
#define OP_SUBABS(i) asm volatile("v_sub_f32 %0, |%0|, %1" : "+v"(a[i]) : "v"(b));
#define OP_CNDMASK(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b));
#define OP_CMP64(i) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(d[i]), "v"(db) : "vcc");
#define OP_MIX(i) asm volatile("v_fma_f64 %0, %2, %3, %0\n v_fma_f32 %1, %4, %5, %1" : "+v"(d[i]), "+v"(a[i]) : "v"(db), "v"(dc), "v"(b), "v"(c));
#define OP_SGPR(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "s"(pb), "v"(pc));
		if (KIND == 0) { REP8(OP_F32) REP8(OP_F32) }
		if (KIND == 1) { REP8(OP_PK) REP8(OP_PK) }
		if (KIND == 2) { REP8(OP_F64) REP8(OP_F64) }
		if (KIND == 3) { REP8(OP_MUL64) REP8(OP_MUL64) }
		if (KIND == 4) { REP8(OP_ADD64) REP8(OP_ADD64) }
		if (KIND == 5) { REP8(OP_MAX32) REP8(OP_MAX32) }
		if (KIND == 6) { REP8(OP_SUBABS) REP8(OP_SUBABS) }
		if (KIND == 7) { REP8(OP_CNDMASK) REP8(OP_CNDMASK) }
		if (KIND == 8) { REP8(OP_CMP64) REP8(OP_CMP64) }
		if (KIND == 9) { REP8(OP_PKADD) REP8(OP_PKADD) }
		if (KIND == 10) { REP8(OP_MIX) }
		if (KIND == 11) { REP8(OP_SGPR) REP8(OP_SGPR) }
	}
	const long long t1 = clock64();
	float s = 0;
	for (int i = 0; i < 8; ++i)
		s += a[i] + p[i].x + p[i].y + (float)d[i];
	sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
	if ((threadIdx.x & 63) == 0)
		cycles[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND>
static void run(const char* name, int waves_per_simd)
{
	const int iters = 4096, blocks = 256 * waves_per_simd; // 256-thread block = 1 wave on each SIMD of a CU
	long long* d_cyc;
	float* d_sink;
	hipMalloc(&d_cyc, sizeof(long long) * blocks * 4);
	hipMalloc(&d_sink, sizeof(float) * blocks * 256);
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	probe<KIND><<<blocks, 256>>>(16, 1.0f, d_cyc, d_sink);
	hipEventRecord(e0);
	probe<KIND><<<blocks, 256>>>(iters, 1.0f, d_cyc, d_sink);
	hipEventRecord(e1);
	hipDeviceSynchronize();
	float ms = 0;
	hipEventElapsedTime(&ms, e0, e1);
	std::vector<long long> cyc(blocks * 4);
	hipMemcpy(cyc.data(), d_cyc, sizeof(long long) * cyc.size(), hipMemcpyDeviceToHost);
	std::sort(cyc.begin(), cyc.end());
	const double med = (double)cyc[cyc.size() / 2];
	const double n_inst = 16.0 * iters;
	printf("%-22s W=%d  wave cycles/inst %.2f  => SIMD cycles/inst %.2f   (kernel %.3f ms, wall-derived %.2f cyc @2.4GHz)\n", name,
		   waves_per_simd, med / n_inst, med / n_inst / waves_per_simd, ms, ms * 1e-3 * 2.4e9 / (n_inst * waves_per_simd));
	hipFree(d_cyc);
	hipFree(d_sink);
}

int main()
{
	for (int w : {1, 2, 4, 8})
	{
		run<0>("v_fma_f32", w);
		run<1>("v_pk_fma_f32", w);
		run<11>("v_pk_fma_f32 (sgpr)", w);
		run<9>("v_pk_add_f32", w);
		run<2>("v_fma_f64", w);
		run<3>("v_mul_f64", w);
		run<4>("v_add_f64", w);
		run<5>("v_max_f32", w);
		run<6>("v_sub_f32 |abs|", w);
		run<7>("v_cndmask_b32", w);
		run<8>("v_cmp_lt_f64", w);
		run<10>("f64 fma + f32 fma pair", w);
		printf("\n");
	}
	return 0;
}
