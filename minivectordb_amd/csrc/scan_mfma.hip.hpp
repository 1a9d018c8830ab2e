// scan_mfma.hip.hpp -- exact brute-force int8 scan with fused top-k (north_star kernel #2).
//
// C[q][n] = sum_k Q[q][k] * V[n][k] on the int8 matrix cores (v_mfma_i32_16x16x64_i8, exact
// i32 accumulation), then L2 = |q|^2 + |v|^2 - 2C (assembly/math.ts:115 gives the same integer)
// or dot (math.ts:133).  Order of the result: (value asc, id asc) for L2, (dot desc, id asc).
//
// Shape: a workgroup = 4 waves = 256 queries; each wave keeps ITS 64 queries as MFMA A-fragments
// in registers for the whole sweep (4 row tiles x KS k-steps x 16 B per lane) and owns their
// running top-k lists in LDS, so waves never synchronise on results.  Node rows stream
// HBM -> registers -> LDS in 32-row tiles (double buffered, padded rows) shared by the 4 waves.
// Grid = query tiles x node splits; splits that share `blockIdx % 8` (one XCD, observed
// round-robin placement: speed only) sweep the SAME node range, so a node tile is fetched from
// HBM once per XCD and served to the other query tiles from that XCD's L2.
// Selection: an element is a candidate when its 64-bit key (biased value << 32 | id) is below
// the row's current k-th key; candidates are appended to a small per-row LDS buffer and merged
// by one lane per row (64 rows in parallel).  A full buffer never drops anything: the lane keeps
// the element pending, the wave flushes, and the lane retries.
#pragma once

#include <hip/hip_runtime.h>

#include "device_index.hpp"

namespace mvdb {

constexpr int kScanMaxK = 32;
constexpr int kScanQB = 256;  // queries per workgroup
constexpr int kScanTN = 32;   // node rows per tile
constexpr int kScanBuf = 8;   // pending candidates per row
constexpr int kScanListStride = kScanMaxK + 1;  // u64 entries per row (+1: bank spread)

using v4i = __attribute__((ext_vector_type(4))) int;

struct ScanArgs {
	DevIndexView ix;
	const int32_t *norm2;  // |v|^2 per node, -1 = absent
	const int8_t *q;
	int nq, k, metric;
	int qtiles, s2, nsplit;  // nsplit = 8 * s2
	int nodes_per_split;     // multiple of kScanTN
	unsigned long long *part_key;  // [nq][nsplit][k]
	int32_t *part_n;               // [nq][nsplit]
	int32_t *out_ids, *out_val, *out_count;
};

__host__ __device__ inline int scan_s2_for(int nq)
{
	const int qtiles = (nq + kScanQB - 1) / kScanQB;
	int s2 = (256 + qtiles * 8 - 1) / (qtiles * 8);  // aim at >= 256 workgroups
	return s2 < 1 ? 1 : (s2 > 64 ? 64 : s2);
}

template <int KS>
__host__ __device__ constexpr size_t scan_lds_bytes()
{
	return (size_t)2 * kScanTN * (KS * 64 + 16)                    // node tiles
	       + (size_t)2 * kScanTN * 4                                // norms of the tiles
	       + (size_t)4 * (64 * kScanListStride * 8 + 64 * kScanBuf * 8 + 64 * 8 + 64 * 4 + 64 * 4);
}

__device__ __forceinline__ unsigned long long scan_key(int val, int id)
{
	return ((unsigned long long)((unsigned)val ^ 0x80000000u) << 32) | (unsigned)id;
}

template <int KS>
__global__ __launch_bounds__(256, 1) void scan_kernel(ScanArgs a)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
	constexpr int ROWB = KS * 64 + 16;  // padded LDS row
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const DevIndexView &ix = a.ix;

	// work mapping (see header): blocks with equal blockIdx % 8 share an XCD
	const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
	const int qtile = j % a.qtiles, s2i = j / a.qtiles;
	const int split = xcd * a.s2 + s2i;
	const int n_begin = split * a.nodes_per_split;
	int n_end = n_begin + a.nodes_per_split;
	if (n_end > ix.max_elements) n_end = ix.max_elements;
	const int ntiles = n_end > n_begin ? (n_end - n_begin + kScanTN - 1) / kScanTN : 0;

	unsigned char *btile = lds;
	int *bnorm = reinterpret_cast<int *>(lds + (size_t)2 * kScanTN * ROWB);
	unsigned char *wbase = reinterpret_cast<unsigned char *>(bnorm + 2 * kScanTN) +
	                       (size_t)wave * (64 * kScanListStride * 8 + 64 * kScanBuf * 8 + 64 * 8 + 64 * 4 + 64 * 4);
	unsigned long long *list = reinterpret_cast<unsigned long long *>(wbase);
	unsigned long long *buf = list + 64 * kScanListStride;
	unsigned long long *thr_s = buf + 64 * kScanBuf;
	int *cnt = reinterpret_cast<int *>(thr_s + 64);
	int *qq_s = cnt + 64;

	// ---- this wave's 64 queries -> A fragments (row = l & 15, k bytes [64 ks + 16 (l>>4), +16))
	const int qbase = qtile * kScanQB + wave * 64;
	v4i qf[4][KS];
#pragma unroll
	for (int rt = 0; rt < 4; rt++) {
		const int qrow = qbase + rt * 16 + (lane & 15);
		const int8_t *src = a.q + (size_t)(qrow < a.nq ? qrow : 0) * (size_t)ix.dim;
#pragma unroll
		for (int ks = 0; ks < KS; ks++) {
			int w[4];
#pragma unroll
			for (int t = 0; t < 4; t++) {
				const int b = ks * 64 + (lane >> 4) * 16 + 4 * t;
				w[t] = (qrow < a.nq && b < ix.dim) ? *reinterpret_cast<const int *>(src + b) : 0;
			}
			qf[rt][ks] = v4i{w[0], w[1], w[2], w[3]};
		}
	}
	{  // |q|^2 of row `lane`
		const int qrow = qbase + lane;
		int s = 0;
		if (qrow < a.nq) {
			const int8_t *src = a.q + (size_t)qrow * (size_t)ix.dim;
			for (int b = 0; b < ix.dim; b += 4) {
				const int w = *reinterpret_cast<const int *>(src + b);
				s = __builtin_amdgcn_sdot4(w, w, s, false);
			}
		}
		qq_s[lane] = s;
		cnt[lane] = 0;
		thr_s[lane] = ~0ull;
	}
	int mylen = 0;  // length of row `lane`'s list
	int qq_r[16];
	unsigned long long thr_r[16];
#pragma unroll
	for (int i = 0; i < 16; i++) {
		const int row = (i >> 2) * 16 + (lane >> 4) * 4 + (i & 3);
		qq_r[i] = qq_s[row];
		thr_r[i] = ~0ull;
	}

	// merge pending candidates of every row: lane r walks row r
	auto flush_all = [&]() {
		int n = cnt[lane];
		if (n > kScanBuf) n = kScanBuf;
		unsigned long long *L = list + lane * kScanListStride;
		for (int t = 0; t < n; t++) {
			const unsigned long long key = buf[lane * kScanBuf + t];
			if (mylen == a.k && !(key < L[a.k - 1])) continue;
			int i = mylen == a.k ? a.k - 2 : mylen - 1;
			while (i >= 0 && L[i] > key) {
				L[i + 1] = L[i];
				i--;
			}
			L[i + 1] = key;
			if (mylen < a.k) mylen++;
		}
		cnt[lane] = 0;
		thr_s[lane] = mylen == a.k ? L[a.k - 1] : ~0ull;
#pragma unroll
		for (int i = 0; i < 16; i++) thr_r[i] = thr_s[(i >> 2) * 16 + (lane >> 4) * 4 + (i & 3)];
	};

	// ---- node tile staging: 32 rows x KS*64 bytes = KS*128 16-byte pieces, KS/2 per thread
	constexpr int PIECES = KS * 128 / 256;
	const int pieces_row = KS * 4;
	const int row_pieces_valid = ix.row_stride >> 4;
	int4 stage[PIECES];
	int stage_norm = -1;
	auto stage_load = [&](int tile) {
		const int node0 = n_begin + tile * kScanTN;
#pragma unroll
		for (int i = 0; i < PIECES; i++) {
			const int p = tid + 256 * i;
			const int r = p / pieces_row, c = p % pieces_row;
			const int node = node0 + r;
			int4 v = make_int4(0, 0, 0, 0);
			if (node < n_end && c < row_pieces_valid)
				v = *reinterpret_cast<const int4 *>(ix.vec + (size_t)node * (size_t)ix.row_stride + c * 16);
			stage[i] = v;
		}
		if (tid < kScanTN) {
			const int node = node0 + tid;
			stage_norm = node < n_end ? a.norm2[node] : -1;
		}
	};
	auto stage_store = [&](int bufi) {
		unsigned char *dst = btile + (size_t)bufi * kScanTN * ROWB;
#pragma unroll
		for (int i = 0; i < PIECES; i++) {
			const int p = tid + 256 * i;
			const int r = p / pieces_row, c = p % pieces_row;
			*reinterpret_cast<int4 *>(dst + r * ROWB + c * 16) = stage[i];
		}
		if (tid < kScanTN) bnorm[bufi * kScanTN + tid] = stage_norm;
	};

	if (ntiles > 0) {
		stage_load(0);
		stage_store(0);
	}
	__syncthreads();

	for (int tile = 0; tile < ntiles; tile++) {
		const int cur = tile & 1;
		if (tile + 1 < ntiles) stage_load(tile + 1);  // in flight under the MFMAs

		const unsigned char *bt = btile + (size_t)cur * kScanTN * ROWB;
		v4i acc[4][2];
#pragma unroll
		for (int rt = 0; rt < 4; rt++) acc[rt][0] = acc[rt][1] = v4i{0, 0, 0, 0};
#pragma unroll
		for (int ks = 0; ks < KS; ks++) {
			const int coff = (ks * 4 + (lane >> 4)) * 16;
			const v4i b0 = *reinterpret_cast<const v4i *>(bt + (lane & 15) * ROWB + coff);
			const v4i b1 = *reinterpret_cast<const v4i *>(bt + ((lane & 15) + 16) * ROWB + coff);
#pragma unroll
			for (int rt = 0; rt < 4; rt++) {
				acc[rt][0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(qf[rt][ks], b0, acc[rt][0], 0, 0, 0);
				acc[rt][1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(qf[rt][ks], b1, acc[rt][1], 0, 0, 0);
			}
		}

		// ---- epilogue: C[row][col]: col = lane & 15 (node), row = (lane >> 4) * 4 + reg (query)
		unsigned long long pend_key[32];
		unsigned pend_mask = 0;
#pragma unroll
		for (int ct = 0; ct < 2; ct++) {
			const int col = ct * 16 + (lane & 15);
			const int node = n_begin + tile * kScanTN + col;
			const int vn = bnorm[cur * kScanTN + col];
#pragma unroll
			for (int i = 0; i < 16; i++) {
				const int c = acc[i >> 2][ct][i & 3];
				const int val = a.metric == 0 ? qq_r[i] + vn - 2 * c : -c;
				const unsigned long long key = scan_key(val, node);
				pend_key[ct * 16 + i] = key;
				if (vn >= 0 && key < thr_r[i]) pend_mask |= 1u << (ct * 16 + i);
			}
		}
		while (__ballot(pend_mask != 0)) {
#pragma unroll
			for (int e = 0; e < 32; e++) {
				if (pend_mask & (1u << e)) {
					const int i = e & 15;
					const int row = (i >> 2) * 16 + (lane >> 4) * 4 + (i & 3);
					if (!(pend_key[e] < thr_r[i])) {
						pend_mask &= ~(1u << e);  // no longer a candidate after a flush
					} else {
						const int slot = atomicAdd(&cnt[row], 1);
						if (slot < kScanBuf) {
							buf[row * kScanBuf + slot] = pend_key[e];
							pend_mask &= ~(1u << e);
						}
					}
				}
			}
			// flush when something is still pending or a buffer is half full
			if (__ballot(pend_mask != 0 || cnt[lane] >= kScanBuf / 2)) flush_all();
		}

		if (tile + 1 < ntiles) stage_store(cur ^ 1);
		__syncthreads();
	}
	flush_all();

	// ---- per-(query, split) partial lists
	{
		const int qrow = qbase + lane;
		if (qrow < a.nq) {
			const size_t o = ((size_t)qrow * a.nsplit + split) * (size_t)a.k;
			const unsigned long long *L = list + lane * kScanListStride;
			for (int i = 0; i < mylen; i++) a.part_key[o + i] = L[i];
			a.part_n[(size_t)qrow * a.nsplit + split] = mylen;
		}
	}
}

// merge the nsplit sorted partial lists of each query: one thread per query
__global__ void scan_merge_kernel(ScanArgs a)
{
	const int qi = blockIdx.x * blockDim.x + threadIdx.x;
	if (qi >= a.nq) return;
	const unsigned long long *pk = a.part_key + (size_t)qi * a.nsplit * a.k;
	const int32_t *pn = a.part_n + (size_t)qi * a.nsplit;
	unsigned long long last = 0;
	bool have_last = false;
	int out = 0;
	for (; out < a.k; out++) {
		// smallest key strictly greater than `last` (keys are unique: ids differ)
		unsigned long long best = ~0ull;
		bool found = false;
		for (int s = 0; s < a.nsplit; s++) {
			const int n = pn[s];
			// lists are sorted: first entry > last by binary search would do; k is small, scan
			for (int i = 0; i < n; i++) {
				const unsigned long long key = pk[(size_t)s * a.k + i];
				if (have_last && key <= last) continue;
				if (!found || key < best) {
					best = key;
					found = true;
				}
				break;
			}
		}
		if (!found) break;
		last = best;
		have_last = true;
		const int val = (int)((unsigned)(best >> 32) ^ 0x80000000u);
		a.out_ids[(size_t)qi * a.k + out] = (int)(unsigned)best;
		a.out_val[(size_t)qi * a.k + out] = a.metric == 0 ? val : -val;
	}
	a.out_count[qi] = out;
	for (int i = out; i < a.k; i++) {  // deterministic tail
		a.out_ids[(size_t)qi * a.k + i] = -1;
		a.out_val[(size_t)qi * a.k + i] = 0;
	}
}

template <int KS>
inline bool launch_scan_t(const ScanArgs &a, hipStream_t s)
{
	const size_t lds = scan_lds_bytes<KS>();
	if (hipFuncSetAttribute(reinterpret_cast<const void *>(&scan_kernel<KS>),
	                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
		return false;
	hipLaunchKernelGGL((scan_kernel<KS>), dim3(a.qtiles * a.nsplit), dim3(256), lds, s, a);
	hipLaunchKernelGGL(scan_merge_kernel, dim3((a.nq + 63) / 64), dim3(64), 0, s, a);
	return hipGetLastError() == hipSuccess;
}

// supported: row_stride <= 512 bytes (dim <= 512)
inline bool launch_scan(const ScanArgs &a, hipStream_t s)
{
	const int ks = (a.ix.row_stride + 63) / 64;
	if (ks <= 2) return launch_scan_t<2>(a, s);
	if (ks <= 4) return launch_scan_t<4>(a, s);
	if (ks <= 6) return launch_scan_t<6>(a, s);
	if (ks <= 8) return launch_scan_t<8>(a, s);
	return false;
}

}  // namespace mvdb
