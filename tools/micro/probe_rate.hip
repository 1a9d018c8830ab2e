// Micro-benchmark: what does a random 4-byte probe cost the memory side, by cache policy of the load?
// (the visited-bitmap probe of traverse_lean_kernel fills a 128-byte line per probe: profiles/r3/pmc_traffic_1m_balanced.json)
//   hipcc --offload-arch=gfx950 -O3 tools/micro/probe_rate.hip -o tools/micro/probe_rate.bin && tools/micro/probe_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x)                                                                    \
	do {                                                                         \
		hipError_t e = (x);                                                      \
		if (e != hipSuccess) {                                                   \
			printf("%s: %s\n", #x, hipGetErrorString(e));                        \
			exit(1);                                                             \
		}                                                                        \
	} while (0)

template <int MODE>
__device__ __forceinline__ unsigned ld(const unsigned *p)
{
	unsigned v;
	if (MODE == 0) v = *p;
	else if (MODE == 1) v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	else if (MODE == 2) v = __builtin_nontemporal_load(p);
	else if (MODE == 3) asm volatile("global_load_dword %0, %1, off sc0 sc1 nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
	else if (MODE == 4) asm volatile("global_load_dword %0, %1, off nt sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
	else asm volatile("global_load_ubyte %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
	return v;
}

// every lane walks its own pseudo-random sequence of words, 8 independent loads in flight per lane
template <int MODE>
__global__ __launch_bounds__(256) void probe(const unsigned *buf, unsigned words_mask, int iters, unsigned *out)
{
	unsigned x = ((blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u) | 1u;
	unsigned acc = 0;
	for (int i = 0; i < iters; i++) {
		unsigned v[8];
#pragma unroll
		for (int u = 0; u < 8; u++) {
			x ^= x << 13, x ^= x >> 17, x ^= x << 5;  // xorshift32
			v[u] = ld<MODE>(buf + (x & words_mask));
		}
#pragma unroll
		for (int u = 0; u < 8; u++) acc += v[u];
	}
	if (acc == 0x12345678u) out[0] = acc;
}

// random 384-byte rows the way the traversal reads them: 8 lanes x 16 bytes x 3 pieces per row, 4 rows in flight per lane group
__global__ __launch_bounds__(256) void rows384(const uint4 *buf, unsigned rows, int iters, unsigned *out)
{
	unsigned x = (((blockIdx.x * 256u + threadIdx.x) >> 3) * 2654435761u + 12345u) | 1u;
	const int s = threadIdx.x & 7;
	unsigned acc = 0;
	for (int i = 0; i < iters; i++) {
		uint4 v[4][3];
#pragma unroll
		for (int u = 0; u < 4; u++) {
			x ^= x << 13, x ^= x >> 17, x ^= x << 5;
			const uint4 *r = buf + (size_t)(x % rows) * 24 + s;
#pragma unroll
			for (int j = 0; j < 3; j++) v[u][j] = r[8 * j];
		}
#pragma unroll
		for (int u = 0; u < 4; u++)
#pragma unroll
			for (int j = 0; j < 3; j++) acc += v[u][j].x ^ v[u][j].y ^ v[u][j].z ^ v[u][j].w;
	}
	if (acc == 0x12345678u) out[0] = acc;
}

void run_rows(const unsigned *buf, size_t bytes, unsigned *out)
{
	const int blocks = 256 * 8, iters = 256;
	const unsigned rows = (unsigned)(bytes / 384);
	hipEvent_t a, b;
	CK(hipEventCreate(&a));
	CK(hipEventCreate(&b));
	rows384<<<blocks, 256>>>((const uint4 *)buf, rows, 8, out);
	CK(hipDeviceSynchronize());
	CK(hipEventRecord(a));
	rows384<<<blocks, 256>>>((const uint4 *)buf, rows, iters, out);
	CK(hipEventRecord(b));
	CK(hipDeviceSynchronize());
	float ms;
	CK(hipEventElapsedTime(&ms, a, b));
	const double n = (double)blocks * 32 * iters * 4;
	printf("%-28s %8.3f ms  %7.2f G rows/s    (x384 B = %6.2f TB/s)\n", "random 384-byte rows", ms, n / ms * 1e-6, n * 384 / ms * 1e-9);
}

template <int MODE>
void run(const char *name, const unsigned *buf, unsigned mask, unsigned *out)
{
	const int blocks = 256 * 8, iters = 256;
	hipEvent_t a, b;
	CK(hipEventCreate(&a));
	CK(hipEventCreate(&b));
	probe<MODE><<<blocks, 256>>>(buf, mask, 8, out);
	CK(hipDeviceSynchronize());
	CK(hipEventRecord(a));
	probe<MODE><<<blocks, 256>>>(buf, mask, iters, out);
	CK(hipEventRecord(b));
	CK(hipDeviceSynchronize());
	float ms;
	CK(hipEventElapsedTime(&ms, a, b));
	const double n = (double)blocks * 256 * iters * 8;
	printf("%-28s %8.3f ms  %7.2f G probes/s  (x128 B = %6.2f TB/s, x64 B = %6.2f, x32 B = %6.2f)\n", name, ms, n / ms * 1e-6,
	       n * 128 / ms * 1e-9, n * 64 / ms * 1e-9, n * 32 / ms * 1e-9);
}

int main()
{
	const size_t bytes = (size_t)4 << 30;  // 4 GiB: the L2s (32 MB) and the Infinity Cache (256 MB) hold 7 % of it
	unsigned *buf, *out;
	CK(hipMalloc(&buf, bytes));
	CK(hipMalloc(&out, 256));
	CK(hipMemset(buf, 0, bytes));
	const unsigned mask = (unsigned)(bytes / 4 - 1);
	run<0>("plain load", buf, mask, out);
	run<1>("agent-scope atomic load", buf, mask, out);
	run<2>("nontemporal load", buf, mask, out);
	run<3>("sc0 sc1 nt", buf, mask, out);
	run<4>("sc1 nt", buf, mask, out);
	run<5>("ubyte sc1", buf, mask, out);
	run_rows(buf, bytes, out);
	unsigned *ub;
	if (hipExtMallocWithFlags((void **)&ub, bytes, hipDeviceMallocUncached) == hipSuccess) {
		CK(hipMemset(ub, 0, bytes));
		run<0>("uncached memory, plain", ub, mask, out);
		run<1>("uncached memory, agent", ub, mask, out);
	}
	return 0;
}
