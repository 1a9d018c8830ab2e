// Microbenchmark: issue rate of v_mfma_i32_16x16x64_i8 and v_mfma_i32_32x32x32_i8 on gfx950 (operands in registers,
// NACC independent accumulators, 1 or 2 waves per SIMD).  hipcc --offload-arch=gfx950 -O3 -o mfma_i8_rate mfma_i8_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using v4i = __attribute__((ext_vector_type(4))) int;
using v16i = __attribute__((ext_vector_type(16))) int;

template <int NACC>
__global__ void k16(int iters, int *out, unsigned long long *cyc)
{
	v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, (int)threadIdx.x, 7};
	v4i acc[NACC];
	for (int i = 0; i < NACC; i++) acc[i] = v4i{0, 0, 0, 0};
	const unsigned long long t0 = __builtin_amdgcn_s_memtime();
	for (int it = 0; it < iters; it++) {
#pragma unroll
		for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[i], 0, 0, 0);
	}
	int s = 0;
	for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
	const unsigned long long t1 = __builtin_amdgcn_s_memtime();
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
	if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NACC>
__global__ void k32(int iters, int *out, unsigned long long *cyc)
{
	v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, (int)threadIdx.x, 7};
	v16i acc[NACC];
	for (int i = 0; i < NACC; i++)
		for (int j = 0; j < 16; j++) acc[i][j] = 0;
	const unsigned long long t0 = __builtin_amdgcn_s_memtime();
	for (int it = 0; it < iters; it++) {
#pragma unroll
		for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[i], 0, 0, 0);
	}
	int s = 0;
	for (int i = 0; i < NACC; i++)
		for (int j = 0; j < 16; j++) s += acc[i][j];
	const unsigned long long t1 = __builtin_amdgcn_s_memtime();
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
	if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <typename K>
static void run(const char *name, K kern, int threads, int nacc, double ops_per_mfma)
{
	const int blocks = 256 * 1, iters = 20000;
	int *out;
	unsigned long long *cyc;
	hipMalloc(&out, (size_t)blocks * threads * 4);
	hipMalloc(&cyc, blocks * 8);
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, 100, out, cyc);
	hipDeviceSynchronize();
	hipEventRecord(e0);
	hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, iters, out, cyc);
	hipEventRecord(e1);
	hipDeviceSynchronize();
	float ms = 0;
	hipEventElapsedTime(&ms, e0, e1);
	std::vector<unsigned long long> h(blocks);
	hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
	double mean = 0;
	for (auto c : h) mean += (double)c;
	mean /= blocks;
	const double waves_per_simd = threads / 64 / 4.0;
	const double mfma_per_wave = (double)iters * nacc;
	printf("%-28s waves/SIMD %.0f  %.3f ms  %.1f cycles per MFMA per SIMD (s_memtime)  %.0f TOP/s chip-wide\n", name, waves_per_simd,
	       ms, mean / (mfma_per_wave * waves_per_simd), ops_per_mfma * mfma_per_wave * (threads / 64) * blocks / (ms * 1e-3) / 1e12);
	hipFree(out);
	hipFree(cyc);
}

int main()
{
	run("16x16x64 i8, 4 acc", k16<4>, 256, 4, 2.0 * 16 * 16 * 64);
	run("16x16x64 i8, 8 acc", k16<8>, 256, 8, 2.0 * 16 * 16 * 64);
	run("16x16x64 i8, 4 acc", k16<4>, 512, 4, 2.0 * 16 * 16 * 64);
	run("16x16x64 i8, 8 acc", k16<8>, 512, 8, 2.0 * 16 * 16 * 64);
	run("32x32x32 i8, 2 acc", k32<2>, 256, 2, 2.0 * 32 * 32 * 32);
	run("32x32x32 i8, 4 acc", k32<4>, 256, 4, 2.0 * 32 * 32 * 32);
	run("32x32x32 i8, 2 acc", k32<2>, 512, 2, 2.0 * 32 * 32 * 32);
	return 0;
}
