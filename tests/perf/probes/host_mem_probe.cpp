// host_mem_probe.cpp -- what it costs to land 0.95 GB of results (256^3 coefficients) in host memory
// the caller owns: zero-filled std::vector, first touch, pinning (hipHostRegister / hipHostMalloc) and
// the D2H rates into pageable / registered / pinned memory.  Test tooling, numbers quoted in DESIGN.md.
//   hipcc -O2 host_mem_probe.cpp -o host_mem_probe -lpthread && ./host_mem_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <sys/mman.h>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); } } while (0)

static void touch_parallel(char* p, size_t bytes, int nt)
{
	std::vector<std::thread> th;
	const size_t per = ((bytes / nt) + 4095) & ~(size_t)4095;
	for (int t = 0; t < nt; ++t)
		th.emplace_back([=]() {
			const size_t b = std::min(bytes, per * t), e = std::min(bytes, per * (t + 1));
			for (size_t o = b; o < e; o += 4096)
				p[o] = 0;
		});
	for (auto& t : th)
		t.join();
}

int main()
{
	const size_t n = 118425857, bytes = n * sizeof(double);
	void* d = nullptr;
	CK(hipMalloc(&d, bytes));
	CK(hipMemset(d, 1, bytes));
	CK(hipDeviceSynchronize());
	hipStream_t s;
	CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
	double t0, t1;

	{
		t0 = now();
		std::vector<double> v(n);
		t1 = now();
		printf("std::vector<double>(n) zero fill           %.1f ms\n", (t1 - t0) * 1e3);
		t0 = now();
		CK(hipMemcpy(v.data(), d, bytes, hipMemcpyDeviceToHost));
		t1 = now();
		printf("hipMemcpy D2H into touched pageable        %.1f ms  (%.1f GB/s)\n", (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9);
		t0 = now();
		CK(hipHostRegister(v.data(), bytes, hipHostRegisterDefault));
		t1 = now();
		printf("hipHostRegister of touched memory          %.1f ms\n", (t1 - t0) * 1e3);
		for (int rep = 0; rep < 2; ++rep)
		{
			t0 = now();
			CK(hipMemcpyAsync(v.data(), d, bytes, hipMemcpyDeviceToHost, s));
			CK(hipStreamSynchronize(s));
			t1 = now();
			printf("D2H into registered memory                 %.1f ms  (%.1f GB/s)\n", (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9);
		}
		t0 = now();
		CK(hipHostUnregister(v.data()));
		t1 = now();
		printf("hipHostUnregister                          %.1f ms\n", (t1 - t0) * 1e3);
	}
	for (int huge = 0; huge < 2; ++huge)
	{
		void* p = nullptr;
		if (posix_memalign(&p, 1 << 21, bytes) != 0)
			return 1;
		if (huge)
			madvise(p, bytes, MADV_HUGEPAGE);
		t0 = now();
		CK(hipHostRegister(p, bytes, hipHostRegisterDefault));
		t1 = now();
		printf("hipHostRegister of UNTOUCHED memory (huge=%d) %.1f ms\n", huge, (t1 - t0) * 1e3);
		t0 = now();
		CK(hipMemcpyAsync(p, d, bytes, hipMemcpyDeviceToHost, s));
		CK(hipStreamSynchronize(s));
		t1 = now();
		printf("  D2H into it                              %.1f ms  (%.1f GB/s)\n", (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9);
		CK(hipHostUnregister(p));
		free(p);
	}
	for (int nt : {1, 8, 16, 32, 64})
		for (int huge = 0; huge < 2; ++huge)
		{
			void* p = nullptr;
			if (posix_memalign(&p, 1 << 21, bytes) != 0)
				return 1;
			if (huge)
				madvise(p, bytes, MADV_HUGEPAGE);
			t0 = now();
			touch_parallel((char*)p, bytes, nt);
			t1 = now();
			printf("first touch, %2d threads (huge=%d)            %.1f ms\n", nt, huge, (t1 - t0) * 1e3);
			if (nt == 16)
			{
				t0 = now();
				CK(hipHostRegister(p, bytes, hipHostRegisterDefault));
				t1 = now();
				printf("  hipHostRegister after parallel touch     %.1f ms\n", (t1 - t0) * 1e3);
				CK(hipHostUnregister(p));
			}
			free(p);
		}
	{
		void* p = nullptr;
		t0 = now();
		CK(hipHostMalloc(&p, bytes, hipHostMallocDefault));
		t1 = now();
		printf("hipHostMalloc 0.95 GB                      %.1f ms\n", (t1 - t0) * 1e3);
		for (int rep = 0; rep < 2; ++rep)
		{
			t0 = now();
			CK(hipMemcpyAsync(p, d, bytes, hipMemcpyDeviceToHost, s));
			CK(hipStreamSynchronize(s));
			t1 = now();
			printf("D2H into hipHostMalloc memory              %.1f ms  (%.1f GB/s)\n", (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9);
		}
		t0 = now();
		CK(hipMemcpyAsync(d, p, bytes, hipMemcpyHostToDevice, s));
		CK(hipStreamSynchronize(s));
		t1 = now();
		printf("H2D from hipHostMalloc memory              %.1f ms  (%.1f GB/s)\n", (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9);
		// host memcpy out of pinned memory into fresh pageable memory with k threads
		for (int nt : {8, 16, 32})
		{
			void* q = nullptr;
			if (posix_memalign(&q, 1 << 21, bytes) != 0)
				return 1;
			madvise(q, bytes, MADV_HUGEPAGE);
			t0 = now();
			std::vector<std::thread> th;
			const size_t per = ((bytes / nt) + 4095) & ~(size_t)4095;
			for (int t = 0; t < nt; ++t)
				th.emplace_back([=]() {
					const size_t b = std::min(bytes, per * t), e = std::min(bytes, per * (t + 1));
					memcpy((char*)q + b, (char*)p + b, e - b);
				});
			for (auto& t : th)
				t.join();
			t1 = now();
			printf("memcpy pinned -> fresh pageable, %2d threads  %.1f ms  (%.1f GB/s)\n", nt, (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9);
			free(q);
		}
		t0 = now();
		CK(hipHostFree(p));
		t1 = now();
		printf("hipHostFree                                %.1f ms\n", (t1 - t0) * 1e3);
	}
	{
		FILE* f = fopen("/sys/kernel/mm/transparent_hugepage/enabled", "r");
		char buf[128] = {0};
		if (f && fgets(buf, sizeof(buf), f))
			printf("THP: %s", buf);
		if (f)
			fclose(f);
	}
	printf("hardware_concurrency %u\n", std::thread::hardware_concurrency());
	return 0;
}
