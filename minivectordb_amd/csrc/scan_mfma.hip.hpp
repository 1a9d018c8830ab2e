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
// Selection: every query row owns an unsorted LDS buffer of kScanCap 64-bit keys (biased value
// << 32 | id).  An element that passes the row's 32-bit bound (see "Fast rejection") is appended
// with one LDS atomic; when a buffer is full the wave compacts that row -- 40 lanes rank the
// keys against each other through readlane, the best k stay, the k-th becomes the row's new
// bound -- so the expensive step is paid once per ~kScanCap-k candidates, not per candidate.
// Nothing is ever dropped: an append that finds the buffer full stays pending until the
// compaction has run.
#pragma once

#include <hip/hip_runtime.h>

#include "device_index.hpp"

namespace mvdb {

constexpr int kScanMaxK = 32;
constexpr int kScanQB = 256;  // queries per workgroup
constexpr int kScanTN = 32;   // node rows per MFMA pass (two 16-column tiles)
constexpr int kScanStage = 64;  // node rows staged per barrier (two passes)
constexpr int kScanCap = 40;  // candidate keys buffered per row before the row is compacted to its best k
constexpr int kScanListStride = kScanCap + 1;  // u64 entries per row (+1: bank spread)

using v4i = __attribute__((ext_vector_type(4))) int;

struct ScanArgs {
	DevIndexView ix;
	const int32_t *norm2;  // |v|^2 per node, -1 = absent
	const int8_t *q;
	int nq, k, metric;
	int qtiles, s2, nsplit;  // nsplit = 8 * s2
	int nodes_per_split;     // multiple of kScanStage
	unsigned long long *thr_g;     // [nq] smallest k-th key published by any split, ~0 before the launch
	unsigned long long *dbg;       // MVDB_SCAN_DEBUG builds: 8 counters (see tools/scan_debug.py)
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
	return (size_t)2 * kScanStage * (KS * 64 + 16)                 // node tiles
	       + (size_t)2 * kScanStage * 4                             // norms of the tiles
	       + (size_t)4 * (64 * kScanListStride * 8 + 64 * 8 + 64 * 4 + 64 * 4);
}

// LDS image of a node tile: 16-byte piece c of row r sits at r * (row bytes + 16) + (c ^ flip(r)) * 16.
// The 16-byte pad makes the bank slot of a piece (r + c) mod 16; ds_read_b128 is served in four
// lane groups that each hold all 16 rows once but mix two neighbouring k-pieces (q, q^1) --
// {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... (MI355X_MICROARCH.md, LDS) -- so rows 4..11 store
// every piece pair swapped and each group lands on 16 distinct slots (without the swap: 2-way).
__device__ __forceinline__ int scan_row_flip(int r) { return (((r & 15) >> 2) == 1 || ((r & 15) >> 2) == 2) ? 1 : 0; }

__device__ __forceinline__ unsigned long long scan_key(int val, int id)
{
	return ((unsigned long long)((unsigned)val ^ 0x80000000u) << 32) | (unsigned)id;
}

// Fast rejection.  A candidate needs key < (row's current bound); its necessary condition on the
// 32-bit value, val <= tv (tv = value part of the bound), is turned into ONE signed compare per
// accumulator element by folding the column's norm into the MFMA's C operand:
//   L2:  val = qq + vn - 2c <= tv  <=>  2c - vn >= qq - tv =: u.  With acc = c - (vn >> 1):
//        2 acc >= 2c - vn, so every candidate has acc >= (u + 1) >> 1 =: g (conservative by the
//        parity bit of vn at most; the slow path re-checks the exact 64-bit key).
//   dot: val = -c <= tv  <=>  acc = c >= -tv =: g.
// Absent rows start from -2^30 and never pass.  While a row has no bound, g = -2^29.
// The bound of a row is the smaller of its own k-th key and the smallest k-th key any other
// node split has published for the same query (a.thr_g, 64-bit atomicMin, polled every
// kScanPoll tiles): k keys <= that bound exist, so nothing above it can be in the final top-k.
// The exchange only prunes; the merged result is the same whatever the timing.
// g of a row without a bound: below every real accumulator (|acc| < 2^25), above the -2^30 start of
// absent rows, and far enough from both for acc - g not to overflow
constexpr int kScanNoBound = -(1 << 29);
constexpr int kScanPoll = 4;  // staged tiles between polls of the shared bounds

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
	const int ntiles = n_end > n_begin ? (n_end - n_begin + kScanStage - 1) / kScanStage : 0;  // staged tiles

	unsigned char *btile = lds;
	int *bnorm = reinterpret_cast<int *>(lds + (size_t)2 * kScanStage * ROWB);
	unsigned char *wbase = reinterpret_cast<unsigned char *>(bnorm + 2 * kScanStage) +
	                       (size_t)wave * (64 * kScanListStride * 8 + 64 * 8 + 64 * 4 + 64 * 4);
	unsigned long long *list = reinterpret_cast<unsigned long long *>(wbase);
	unsigned long long *thr_s = list + 64 * kScanListStride;
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
	const int my_qrow = qbase + lane;  // the row this lane owns in the selection structures
	{  // |q|^2 of row `lane`
		int s = 0;
		if (my_qrow < a.nq) {
			const int8_t *src = a.q + (size_t)my_qrow * (size_t)ix.dim;
			for (int b = 0; b < ix.dim; b += 4) {
				const int w = *reinterpret_cast<const int *>(src + b);
				s = __builtin_amdgcn_sdot4(w, w, s, false);
			}
		}
		qq_s[lane] = s;
		cnt[lane] = 0;
		thr_s[lane] = ~0ull;
	}
	unsigned long long ext = ~0ull;  // smallest bound other splits published for row `lane`
	// per lane: the 16 rows its accumulators belong to (the same 16 for the whole quarter-wave):
	// |q|^2, the 32-bit bound of the fast test and the fill of the row's buffer (kept in step
	// across the quarter, written to cnt[] only around compactions)
	int qq_r[16], g_r[16], cnt_r[16];
#pragma unroll
	for (int i = 0; i < 16; i++) {
		qq_r[i] = qq_s[(i >> 2) * 16 + (lane >> 4) * 4 + (i & 3)];
		g_r[i] = kScanNoBound;
		cnt_r[i] = 0;
	}
	auto cnt_to_lds = [&]() {
		if ((lane & 15) == 0) {
#pragma unroll
			for (int i = 0; i < 16; i++) cnt[(i >> 2) * 16 + (lane >> 4) * 4 + (i & 3)] = cnt_r[i];
		}
	};
	auto cnt_from_lds = [&]() {
#pragma unroll
		for (int i = 0; i < 16; i++) cnt_r[i] = cnt[(i >> 2) * 16 + (lane >> 4) * 4 + (i & 3)];
	};

	// thr_s (exact 64-bit bound per row, LDS) -> g_r (32-bit fast test, registers)
	auto refresh_bounds = [&]() {
#pragma unroll
		for (int i = 0; i < 16; i++) {
			const unsigned long long t = thr_s[(i >> 2) * 16 + (lane >> 4) * 4 + (i & 3)];
			const int tv = (int)((unsigned)(t >> 32) ^ 0x80000000u);
			int g = kScanNoBound;
			if (t != ~0ull) g = a.metric == 0 ? ((qq_r[i] - tv + 1) >> 1) : -tv;
			g_r[i] = g;
		}
	};
	// Compact the rows in `rows` (bit r = row r) to their best min(n, k) keys, sorted ascending.
	// Lane j holds key j of the row; its rank is the number of smaller keys (keys are unique).
	auto compact_rows = [&](unsigned long long rows) {
		while (rows) {
			const int R = __builtin_ctzll(rows);
			rows &= rows - 1;
			int n = __builtin_amdgcn_readfirstlane(cnt[R]);
			if (n > kScanCap) n = kScanCap;
			unsigned long long *L = list + R * kScanListStride;
			const unsigned long long key = lane < n ? L[lane] : ~0ull;
			const unsigned klo = (unsigned)key, khi = (unsigned)(key >> 32);
			int rank = 0;
			for (int i = 0; i < n; i++) {
				const unsigned long long ki = (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)klo, i) |
				                              ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)khi, i) << 32);
				rank += ki < key ? 1 : 0;
			}
			if (lane < n && rank < a.k) L[rank] = key;
			if (lane == 0) cnt[R] = n < a.k ? n : a.k;
			if (n >= a.k) {
				const unsigned long long ext_R =
				    (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)ext, R) |
				    ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(ext >> 32), R) << 32);
				if (lane < n && rank == a.k - 1) {
					thr_s[R] = key < ext_R ? key : ext_R;
					if (key < ext_R && qbase + R < a.nq) atomicMin(&a.thr_g[qbase + R], key);
				}
			}
		}
	};

	// ---- node tile staging: kScanStage rows x KS*64 bytes, KS 16-byte pieces per thread.  The
	// loads of staged tile T+1 are issued before the two MFMA passes over tile T and stored to
	// the other LDS buffer after them: two passes (~2 x 1024 matrix-pipe cycles) cover an L2 hit
	// and most of an HBM miss.  (A register ring of several tiles does not work with hipcc: it
	// drains vmcnt to zero at the first MFMA of the loop body, i.e. right after the newest load.)
	constexpr int PIECES = KS * 4 * kScanStage / 256;
	const int pieces_row = KS * 4;
	const int row_pieces_valid = ix.row_stride >> 4;
	int4 stage[PIECES];
	int stage_norm = -1;
	auto stage_load = [&](int tile) {
		const int node0 = n_begin + tile * kScanStage;
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
		if (tid < kScanStage) {
			const int node = node0 + tid;
			stage_norm = node < n_end ? a.norm2[node] : -1;
		}
	};
	auto stage_store = [&](int bufi) {
		unsigned char *dst = btile + (size_t)bufi * kScanStage * ROWB;
#pragma unroll
		for (int i = 0; i < PIECES; i++) {
			const int p = tid + 256 * i;
			const int r = p / pieces_row, c = p % pieces_row;
			*reinterpret_cast<int4 *>(dst + r * ROWB + (c ^ scan_row_flip(r)) * 16) = stage[i];
		}
		if (tid < kScanStage) bnorm[bufi * kScanStage + tid] = stage_norm;
	};

#ifdef MVDB_SCAN_DEBUG
	unsigned long long dbg_c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
	const unsigned long long dbg_t0 = __builtin_amdgcn_s_memtime();
#endif
	// The (rare) elements that pass the fast test are appended to their row's buffer.  C[row][col]:
	// col = lane & 15 (node), row = (lane >> 4) * 4 + reg (query); vn0/vn1 = norms of this lane's
	// two columns; anyg[j] != 0 when some lane passed the test for an element in [8j, 8j+8).
	auto slow_path = [&](const v4i (&acc)[4][2], int node0, int vn0, int vn1, const unsigned long long (&anyg)[4]) {
#ifdef MVDB_SCAN_DEBUG
		const unsigned long long t_in = __builtin_amdgcn_s_memtime();
		dbg_c[1]++;
#endif
		unsigned pend = 0;
#pragma unroll
		for (int gi = 0; gi < 4; gi++) {
			if (anyg[gi]) {
#pragma unroll
				for (int e = gi * 8; e < gi * 8 + 8; e++)
					pend |= (acc[(e & 15) >> 2][e >> 4][e & 3] >= g_r[e & 15]) ? (1u << e) : 0u;
			}
		}
		if (vn0 < 0) pend &= 0xffff0000u;
		if (vn1 < 0) pend &= 0x0000ffffu;
		while (__ballot(pend != 0)) {
			bool ovf = false;
#pragma unroll
			for (int gi = 0; gi < 4; gi++) {
				if (__ballot((pend >> (gi * 8)) & 0xffu)) {
#pragma unroll
					for (int e = gi * 8; e < gi * 8 + 8; e++) {
						const bool pe = (pend >> e) & 1u;
						const unsigned long long m = __ballot(pe);
						if (m) {
							// the 16 lanes of a quarter-wave hold 16 columns of the SAME row: slots by prefix count
							const int i = e & 15, ct = e >> 4;
							const unsigned qm = (unsigned)(m >> (lane & 48)) & 0xffffu;
							const int slot = cnt_r[i] + __builtin_popcount(qm & ((1u << (lane & 15)) - 1u));
							const int nc = cnt_r[i] + __builtin_popcount(qm);
							if (pe && slot < kScanCap) {
								const int row = (i >> 2) * 16 + (lane >> 4) * 4 + (i & 3);
								const int vn = ct ? vn1 : vn0;
								int av = acc[i >> 2][ct][i & 3];
								// keep the key arithmetic inside this (rarely taken) branch: without the barrier
								// hipcc hoists all 32 keys out of the retry loop and pays for them on every entry
								asm volatile("" : "+v"(av));
								const int c = a.metric == 0 ? av + (vn >> 1) : av;
								const int val = a.metric == 0 ? qq_r[i] + vn - 2 * c : -c;
								list[row * kScanListStride + slot] = scan_key(val, node0 + ct * 16 + (lane & 15));
								pend &= ~(1u << e);
							}
							ovf |= nc >= kScanCap;
							cnt_r[i] = nc < kScanCap ? nc : kScanCap;
						}
					}
				}
			}
#ifdef MVDB_SCAN_DEBUG
			dbg_c[2]++;
#endif
			if (__ballot(ovf)) {  // a full buffer keeps its element pending: compact, tighten the bounds, retry
				cnt_to_lds();
				const unsigned long long full = __ballot(cnt[lane] >= kScanCap);
#ifdef MVDB_SCAN_DEBUG
				dbg_c[3] += __builtin_popcountll(full);
#endif
#ifdef MVDB_SCAN_DEBUG
				const unsigned long long t_c = __builtin_amdgcn_s_memtime();
#endif
				compact_rows(full);
				cnt_from_lds();
				refresh_bounds();
#ifdef MVDB_SCAN_DEBUG
				dbg_c[7] += __builtin_amdgcn_s_memtime() - t_c;
#endif
			}
		}
#ifdef MVDB_SCAN_DEBUG
		dbg_c[4] += __builtin_amdgcn_s_memtime() - t_in;
#endif
	};
	auto col_init = [&](int vn) -> int { return vn < 0 ? -(1 << 30) : (a.metric == 0 ? -(vn >> 1) : 0); };

	// Software pipeline: the MFMAs of tile t are issued together with the fast test of tile t-1
	// (independent instruction streams, one matrix-pipe and one VALU), the rare slow path follows.
	v4i accA[4][2], accB[4][2];
#pragma unroll
	for (int rt = 0; rt < 4; rt++) accB[rt][0] = accB[rt][1] = v4i{0, 0, 0, 0};
	int pvn0 = -1, pvn1 = -1, pnode0 = 0;
	bool have_prev = false;
	unsigned long long ext_next = ~0ull;
	if (ntiles > 0) {
		stage_load(0);
		stage_store(0);
	}
	// bounds known before the sweep starts (the warm-start launch, or splits that are ahead)
	if (my_qrow < a.nq) ext = __hip_atomic_load(&a.thr_g[my_qrow], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	thr_s[lane] = ext;
	__syncthreads();
	refresh_bounds();

	// one MFMA pass over 32 staged rows (half `half` of buffer `cur`) into accN, with the fast test
	// of the previous pass (accP) riding along
	auto pass = [&](int tile, int cur, int half, v4i (&accN)[4][2], const v4i (&accP)[4][2]) {
		const unsigned char *bt = btile + ((size_t)cur * kScanStage + half * kScanTN) * ROWB;
		const int *bn = bnorm + cur * kScanStage + half * kScanTN;
		const int vn0 = bn[lane & 15], vn1 = bn[16 + (lane & 15)];
		const int c0 = col_init(vn0), c1 = col_init(vn1);
#pragma unroll
		for (int rt = 0; rt < 4; rt++) {
			accN[rt][0] = v4i{c0, c0, c0, c0};
			accN[rt][1] = v4i{c1, c1, c1, c1};
		}
		// fast test of the previous pass, kept on the VALU: mg[j] = max over the elements of group j
		// of (acc - g); a candidate exists iff some lane ends with mg >= 0 (no SGPR traffic between MFMAs)
		int mg[4] = {-1, -1, -1, -1};
		const unsigned char *bp0 = bt + (lane & 15) * ROWB + ((lane >> 4) ^ scan_row_flip(lane & 15)) * 16;
		const unsigned char *bp1 = bp0 + 16 * ROWB;
		// fragments of k-steps ks+1 and ks+2 are in flight while the MFMAs of ks run
		v4i bf[KS + 2][2];
		bf[0][0] = *reinterpret_cast<const v4i *>(bp0);
		bf[0][1] = *reinterpret_cast<const v4i *>(bp1);
		bf[1][0] = *reinterpret_cast<const v4i *>(bp0 + 64);
		bf[1][1] = *reinterpret_cast<const v4i *>(bp1 + 64);
		constexpr int PER = 32 / (KS * 4);
#pragma unroll
		for (int ks = 0; ks < KS; ks++) {
			if (ks + 2 < KS) {
				bf[ks + 2][0] = *reinterpret_cast<const v4i *>(bp0 + (ks + 2) * 64);
				bf[ks + 2][1] = *reinterpret_cast<const v4i *>(bp1 + (ks + 2) * 64);
			}
#pragma unroll
			for (int rt = 0; rt < 4; rt++) {
				accN[rt][0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(qf[rt][ks], bf[ks][0], accN[rt][0], 0, 0, 0);
				accN[rt][1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(qf[rt][ks], bf[ks][1], accN[rt][1], 0, 0, 0);
#pragma unroll
				for (int u = 0; u < PER; u++) {
					const int e = (ks * 4 + rt) * PER + u;
					const int d = accP[(e & 15) >> 2][e >> 4][e & 3] - g_r[e & 15];
					mg[e >> 3] = d > mg[e >> 3] ? d : mg[e >> 3];
				}
			}
			__builtin_amdgcn_sched_barrier(0);
		}
#pragma unroll
		for (int e = PER * KS * 4; e < 32; e++) {
			const int d = accP[(e & 15) >> 2][e >> 4][e & 3] - g_r[e & 15];
			mg[e >> 3] = d > mg[e >> 3] ? d : mg[e >> 3];
		}
		unsigned long long anyg[4];
#pragma unroll
		for (int j = 0; j < 4; j++) anyg[j] = __ballot(mg[j] >= 0);
		if (have_prev && (anyg[0] | anyg[1] | anyg[2] | anyg[3])) slow_path(accP, pnode0, pvn0, pvn1, anyg);
		pvn0 = vn0;
		pvn1 = vn1;
		pnode0 = n_begin + tile * kScanStage + half * kScanTN;
		have_prev = true;
#ifdef MVDB_SCAN_DEBUG
		dbg_c[0]++;
#endif
	};
	for (int tile = 0; tile < ntiles; tile++) {
		const int cur = tile & 1;
		if (tile + 1 < ntiles) stage_load(tile + 1);  // in flight under both passes
		if ((tile % kScanPoll) == 0 && my_qrow < a.nq)
			ext_next = __hip_atomic_load(&a.thr_g[my_qrow], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		pass(tile, cur, 0, accA, accB);
		pass(tile, cur, 1, accB, accA);
		if ((tile % kScanPoll) == 1 && ext_next < ext) {  // a tighter bound from another split
			ext = ext_next;
			if (ext < thr_s[lane]) thr_s[lane] = ext;
			refresh_bounds();
		}
#ifdef MVDB_SCAN_DEBUG
		const unsigned long long t_b = __builtin_amdgcn_s_memtime();
#endif
		if (tile + 1 < ntiles) stage_store(cur ^ 1);
		__syncthreads();
#ifdef MVDB_SCAN_DEBUG
		dbg_c[5] += __builtin_amdgcn_s_memtime() - t_b;
#endif
	}
	if (have_prev) {  // the last pass always ran into accB
		unsigned long long anyg[4] = {0, 0, 0, 0};
#pragma unroll
		for (int e = 0; e < 32; e++) anyg[e >> 3] |= __ballot(accB[(e & 15) >> 2][e >> 4][e & 3] >= g_r[e & 15]);
		if (anyg[0] | anyg[1] | anyg[2] | anyg[3]) slow_path(accB, pnode0, pvn0, pvn1, anyg);
	}
	cnt_to_lds();
	compact_rows(__ballot(cnt[lane] > 0));  // every row: best min(n, k), sorted
#ifdef MVDB_SCAN_DEBUG
	dbg_c[6] = __builtin_amdgcn_s_memtime() - dbg_t0;
	if (lane == 0 && a.dbg)
		for (int i = 0; i < 8; i++) atomicAdd(&a.dbg[i], dbg_c[i]);
#endif

	// ---- per-(query, split) partial lists, sorted ascending
	if (my_qrow < a.nq) {
		const unsigned long long *L = list + lane * kScanListStride;
		const int mylen = cnt[lane];
		const size_t o = ((size_t)my_qrow * a.nsplit + split) * (size_t)a.k;
		for (int i = 0; i < mylen; i++) a.part_key[o + i] = L[i];
		a.part_n[(size_t)my_qrow * a.nsplit + split] = mylen;
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

// Warm start (see launch_scan): after a sweep over a prefix of the nodes, thr_g[q] = (k-th smallest
// key of the prefix) + 1 -- "k keys below this bound exist".  +1 because the full sweep visits the
// prefix again and must still accept that very key (keys are unique: value << 32 | id).
__global__ void scan_bound_kernel(ScanArgs a)
{
	const int qi = blockIdx.x * blockDim.x + threadIdx.x;
	if (qi >= a.nq) return;
	const unsigned long long *pk = a.part_key + (size_t)qi * a.nsplit * a.k;
	const int32_t *pn = a.part_n + (size_t)qi * a.nsplit;
	unsigned long long last = 0;
	bool have_last = false;
	int out = 0;
	for (; out < a.k; out++) {
		unsigned long long best = ~0ull;
		bool found = false;
		for (int s = 0; s < a.nsplit; s++) {
			const int n = pn[s];
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
	}
	a.thr_g[qi] = out == a.k ? last + 1 : ~0ull;
}

constexpr int kScanWarmNodes = 8192;  // prefix swept first to seed the bounds

template <int KS>
inline bool launch_scan_t(const ScanArgs &a, hipStream_t s)
{
	const size_t lds = scan_lds_bytes<KS>();
	if (hipFuncSetAttribute(reinterpret_cast<const void *>(&scan_kernel<KS>),
	                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
		return false;
	// Most candidates of a sweep turn up in its first few thousand nodes, when the bounds are
	// still loose and every split learns them separately.  Sweeping a short prefix first and
	// starting the full sweep from its k-th keys removes ~90 % of the slow-path entries.
	if (a.ix.max_elements >= 8 * kScanWarmNodes) {
		ScanArgs w = a;
		w.ix.max_elements = kScanWarmNodes;
		w.nodes_per_split = (((kScanWarmNodes + w.nsplit - 1) / w.nsplit) + kScanStage - 1) / kScanStage * kScanStage;
		hipLaunchKernelGGL((scan_kernel<KS>), dim3(w.qtiles * w.nsplit), dim3(256), lds, s, w);
		hipLaunchKernelGGL(scan_bound_kernel, dim3((a.nq + 63) / 64), dim3(64), 0, s, w);
	}
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
