// scan_seed.hip.hpp -- level 0 of the exact scan (the first kScanLevel0 nodes, no bound yet), included by
// scan_mfma.hip.hpp.
//
// Without a bound EVERY (query, node) pair of the first level is a candidate, and the sweep kernels' candidate
// path (LDS atomics, row buffers, compactions) is built for the rare case: 1024 nodes cost them ~190 us of a
// 2.3 ms sweep at 4096 queries.  This kernel does the first level the direct way: a workgroup per 16 queries computes
// the 16 x n0 distance block with MFMAs (B fragments straight from global memory: the n0 rows are L2-resident
// and shared by every workgroup; the four waves of a workgroup take every fourth node tile each, which is what
// hides the load latency), parks it in LDS, and selects each query's k smallest keys (four queries per wave) by
// k rounds of "smallest key in the wave" over 16 keys per lane.  It writes the sorted list into ONE partial-list slot per
// query (scan_plan gives level 0 a single split), so the merge that follows finds the bounds as after any level.
#pragma once

namespace mvdb {

constexpr int kSeedMaxNodes = 1024;             // node slots of a level this kernel takes (16 keys per lane)
constexpr int kSeedRowStride = kSeedMaxNodes + 4;  // ints per query row in LDS (+4: the four row groups of a store land on different banks)
constexpr size_t kSeedLdsBytes = (size_t)16 * kSeedRowStride * 4;

template <int KS, int METRIC>
__global__ __launch_bounds__(256) void scan_seed_kernel(ScanArgs a)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
	int *vals = reinterpret_cast<int *>(lds);  // [16][kSeedRowStride]: value of (query row, node slot), INT_MAX = no node
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const DevIndexView &ix = a.ix;
	const int qbase = blockIdx.x * 16;
	const int n0 = a.node_hi - a.node_lo;  // <= kSeedMaxNodes (host checks)

	// A fragments: query row lane & 15, bytes ks * 64 + (lane >> 4) * 16 .. + 16; zero past the batch or the dimension
	v4i qf[KS];
	const int qrow = qbase + (lane & 15);
	{
		const int8_t *src = a.q + (size_t)(qrow < a.nq ? qrow : 0) * (size_t)ix.dim;
#pragma unroll
		for (int ks = 0; ks < KS; ks++) {
			int w[4];
#pragma unroll
			for (int t = 0; t < 4; t++) {
				const int b = ks * 64 + (lane >> 4) * 16 + 4 * t;
				const int v = *reinterpret_cast<const int *>(src + (b < ix.dim ? b : 0));
				w[t] = (qrow < a.nq && b < ix.dim) ? v : 0;
			}
			qf[ks] = v4i{w[0], w[1], w[2], w[3]};
		}
	}
	// |q|^2 of the four rows this lane's accumulator registers belong to (row = (lane >> 4) * 4 + i)
	int qq_r[4];
	{
		int s = 0;  // lane & 15 = row: partial sum over this lane's quarter of every k-step, then over the four quarters
#pragma unroll
		for (int ks = 0; ks < KS; ks++)
#pragma unroll
			for (int t = 0; t < 4; t++) s = __builtin_amdgcn_sdot4(qf[ks][t], qf[ks][t], s, false);
		s += __shfl_xor(s, 16, 64);
		s += __shfl_xor(s, 32, 64);
#pragma unroll
		for (int i = 0; i < 4; i++) qq_r[i] = __shfl(s, (lane >> 4) * 4 + i, 64);
	}

	const int row_pieces_valid = ix.row_stride >> 4;
#pragma unroll 2
	for (int t0 = wave * 16; t0 < n0; t0 += 64) {
		const int node = a.node_lo + t0 + (lane & 15);
		const bool have = t0 + (lane & 15) < n0;
		const int8_t *row = ix.vec + (size_t)(have ? node : a.node_lo) * (size_t)ix.row_stride;
		v4i bfr[KS];
#pragma unroll
		for (int ks = 0; ks < KS; ks++) {
			const int piece = ks * 4 + (lane >> 4);
			const v4i v = *reinterpret_cast<const v4i *>(row + (piece < row_pieces_valid ? piece : 0) * 16);
			bfr[ks] = (have && piece < row_pieces_valid) ? v : v4i{0, 0, 0, 0};
		}
		const int vn = have ? a.norm2[node] : -1;
		v4i acc = {0, 0, 0, 0};
#pragma unroll
		for (int ks = 0; ks < KS; ks++) acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(qf[ks], bfr[ks], acc, 0, 0, 0);
#pragma unroll
		for (int i = 0; i < 4; i++) {
			const int val = METRIC == 0 ? qq_r[i] + vn - 2 * acc[i] : -acc[i];
			vals[((lane >> 4) * 4 + i) * kSeedRowStride + t0 + (lane & 15)] = vn < 0 ? INT_MAX : val;
		}
	}
	__syncthreads();

	// selection: per query, lane l holds node slots l, l + 64, ...; k rounds of the wave's smallest key
	for (int r = wave; r < 16; r += 4) {
		const int q = qbase + r;
		if (q >= a.nq) break;
		unsigned long long key[kSeedMaxNodes / 64];
#pragma unroll
		for (int j = 0; j < kSeedMaxNodes / 64; j++) {
			const int slot = j * 64 + lane;
			const int val = slot < n0 ? vals[r * kSeedRowStride + slot] : INT_MAX;
			key[j] = val == INT_MAX ? ~0ull : scan_key(val, a.node_lo + slot);
		}
		const size_t o = ((size_t)q * a.nslots + a.slot0) * (size_t)a.k;
		int out = 0;
		for (; out < a.k; out++) {
			unsigned long long best = key[0];
#pragma unroll
			for (int j = 1; j < kSeedMaxNodes / 64; j++) best = key[j] < best ? key[j] : best;
#pragma unroll
			for (int d = 32; d >= 1; d >>= 1) {
				const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)best, d, 64);
				const unsigned hi = (unsigned)__shfl_xor((int)(unsigned)(best >> 32), d, 64);
				const unsigned long long ob = ((unsigned long long)hi << 32) | lo;
				best = ob < best ? ob : best;
			}
			if (best == ~0ull) break;
			if (lane == 0) a.part_key[o + out] = best;
#pragma unroll
			for (int j = 0; j < kSeedMaxNodes / 64; j++) key[j] = key[j] == best ? ~0ull : key[j];  // keys are unique (ids differ)
		}
		if (lane == 0) a.part_n[(size_t)q * a.nslots + a.slot0] = out;
	}
}

}  // namespace mvdb
