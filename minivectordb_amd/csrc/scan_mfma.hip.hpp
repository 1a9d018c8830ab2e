// scan_mfma.hip.hpp -- exact brute-force int8 scan with fused top-k (north_star kernel #2).
//
// C[q][n] = sum_k Q[q][k] * V[n][k] on the int8 matrix cores (v_mfma_i32_16x16x64_i8, exact
// i32 accumulation), then L2 = |q|^2 + |v|^2 - 2C (assembly/math.ts:115 gives the same integer)
// or dot (math.ts:133).  Order of the result: (value asc, id asc) for L2, (dot desc, id asc).
//
// Shape: a workgroup = 8 waves = 256 queries, TWO waves per SIMD: each wave keeps ITS 32 queries as
// MFMA A-fragments in registers for the whole sweep (2 row tiles x KS k-steps x 16 B per lane) and
// owns their running top-k lists in LDS, so waves never synchronise on results, and while one
// wave of a SIMD handles candidates (serial, branchy code) its partner keeps the matrix pipe busy.
// Node rows stream HBM -> registers -> LDS in 64-row staged tiles (double buffered, padded rows)
// shared by the 8 waves; a B fragment (one ds_read_b128) feeds two MFMAs: 8 waves x 16 reads per
// 32-row pass = half of the LDS read rate at full MFMA speed.
// Grid = query tiles x node splits; splits that share `blockIdx % 8` (one XCD, observed
// round-robin placement: speed only) sweep the SAME node range, so a node tile is fetched from
// HBM once per XCD and served to the other query tiles from that XCD's L2.
// Selection: every query row owns an unsorted LDS buffer of kScanCap 64-bit keys (biased value
// << 32 | id).  An element that passes the row's 32-bit bound (see "Fast rejection") is appended
// (slot from one LDS atomic per hit element); when a buffer is full the wave compacts that row --
// 40 lanes rank the keys against each other through readlane, the best k stay, the k-th becomes
// the row's new bound -- so the expensive step is paid once per ~kScanCap-k candidates, not per
// candidate.  Nothing is ever dropped: an append that finds the buffer full stays pending until
// the compaction has run.
#pragma once

#include <hip/hip_runtime.h>

#include "device_index.hpp"

namespace mvdb {

constexpr int kScanMaxK = 32;
constexpr int kScanQB = 256;    // queries per workgroup
constexpr int kScanWaves = 8;   // waves per workgroup, 32 query rows each
constexpr int kScanWR = 32;     // query rows per wave
constexpr int kScanTN = 32;     // node rows per MFMA pass (two 16-column tiles)
constexpr int kScanStage = 64;  // node rows staged per barrier (two passes)
// candidate keys buffered per row before the row is compacted to its best k: 24 while k <= 16 (the LDS that
// saves holds a third node-tile buffer), 40 up to kScanMaxK
constexpr int kScanCapSmall = 24, kScanCapLarge = 40, kScanSmallK = 16;
constexpr int kScanThreads = kScanWaves * 64;

using v4i = __attribute__((ext_vector_type(4))) int;

struct ScanArgs {
	DevIndexView ix;
	const int32_t *norm2;  // |v|^2 per node, -1 = absent
	const int8_t *q;
	int nq, k, metric;
	int qtiles, s2, nsplit;  // nsplit = 8 * s2 (per level, set by the launcher)
	int nodes_per_split;     // multiple of kScanStage (per level, set by the launcher)
	int node_lo, node_hi;    // node range of this level
	int slot0, nslots;       // partial-list slot of split 0 of this level; slots per query over all levels
	int level0, level_ratio, wide_ratio; // level plan (scan_levels)
	int seed;                // 1: level 0 goes to scan_seed_kernel (see scan_plan)
	int wide;                // > 0: levels of at least this many nodes go to scan_wide_kernel (k <= kWideMaxK); see scan_plan
	unsigned long long *thr_g;     // [nq] smallest k-th key published by any split, ~0 before the launch
	unsigned long long *part_key;  // [nq][nslots][k]
	int32_t *part_n;               // [nq][nslots]
	int32_t *out_ids, *out_val, *out_count;
};

__host__ __device__ inline int scan_s2_for(int nq)
{
	const int qtiles = (nq + kScanQB - 1) / kScanQB;
	int s2 = (256 + qtiles * 8 - 1) / (qtiles * 8);  // aim at >= 256 workgroups
	return s2 < 1 ? 1 : (s2 > 64 ? 64 : s2);
}

template <int CAP>
__host__ __device__ constexpr size_t scan_wave_lds()  // per wave: lists (+1 entry per row: bank spread), bounds, fills, |q|^2
{
	return (size_t)kScanWR * (CAP + 1) * 8 + kScanWR * 8 + kScanWR * 4 + kScanWR * 4;
}

template <int KS, int CAP, int NB>
__host__ __device__ constexpr size_t scan_lds_bytes()
{
	return (size_t)NB * kScanStage * (KS * 64 + 16)  // node tiles
	       + (size_t)NB * kScanStage * 4              // norms of the tiles
	       + (size_t)kScanWaves * scan_wave_lds<CAP>();
}

// node-tile buffers: three when they fit beside the lists (the first fragments of the next tile
// are then readable BEFORE the barrier that ends the current one), else two
template <int KS, int CAP>
__host__ __device__ constexpr int scan_buffers()
{
	return scan_lds_bytes<KS, CAP, 3>() <= 160 * 1024 ? 3 : 2;
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
// 32-bit value, val <= tv (tv = value part of the bound), is turned into a SIGN test of the
// accumulator by folding both the column's norm and the row's bound into the MFMA's C operand:
//   L2:  val = qq + vn - 2c <= tv  <=>  2c - vn >= qq - tv =: u.  With b = c - (vn >> 1):
//        2 b >= 2c - vn, so every candidate has b >= (u + 1) >> 1 =: g (conservative by the
//        parity bit of vn at most; the exact 64-bit key is formed for the elements that pass and
//        the compaction orders by it).
//   dot: val = -c <= tv  <=>  b = c >= -tv =: g.
// The accumulator starts at -(vn >> 1) - g (dot: -g), so after the k-steps it holds b - g and
// "some element of this pass may be a candidate" is max(acc) >= 0: one v_max3 per two elements and
// one ballot per pass.  The g a pass started from travels with its accumulators (bounds may
// tighten while the pass is in flight), so the exact c is recovered as acc + g (+ vn >> 1).
// Absent rows start from -2^30 - g and never pass.  While a row has no bound, g = -2^29.
// The bound of a row is the smaller of its own k-th key and the smallest k-th key any other
// node split has published for the same query (a.thr_g, 64-bit atomicMin, polled every
// kScanPoll tiles): k keys <= that bound exist, so nothing above it can be in the final top-k.
// The exchange only prunes; the merged result is the same whatever the timing.
// g of a row without a bound: below every real b (|b| < 2^25), above the -2^30 start of absent
// rows, and far enough from both for the differences not to overflow
constexpr int kScanNoBound = -(1 << 29);

// Staging path: LDS-DMA (global_load_lds_dwordx4): node tiles go HBM/L2 -> LDS without passing through
// registers, two tiles ahead with three buffers, counted vmcnt + raw s_barrier.  (Staging through registers,
// global_load_dwordx4 + ds_write_b128, was the round-1 path: slower, see DESIGN.md.)
constexpr int kScanPoll = 16;  // staged tiles between polls of the shared bounds (an ordinary load drains the DMA queue)
// sources of the LDS-DMA lanes that have nothing to fetch: pad slots, pieces past the row, rows past the end
static __device__ __align__(16) int g_scan_zero[4] = {0, 0, 0, 0};
static __device__ int g_scan_neg1 = -1;

template <int N>
__device__ __forceinline__ void scan_wait_vm()
{
	asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// wait until at most n (0..6) of this wave's vector-memory operations are outstanding (n is wave-uniform)
__device__ __forceinline__ void scan_wait_vm_upto(int n)
{
	switch (n) {
	case 1: scan_wait_vm<1>(); break;
	case 2: scan_wait_vm<2>(); break;
	case 3: scan_wait_vm<3>(); break;
	case 4: scan_wait_vm<4>(); break;
	case 5: scan_wait_vm<5>(); break;
	case 6: scan_wait_vm<6>(); break;
	default: scan_wait_vm<0>(); break;
	}
}
// 0 (default): the two waves of a SIMD alternate tiles.  1: every wave issues its share of every tile, one
// instruction at a time between its MFMAs.  Measured the same within 2 % (1 M x 512, 4096 queries): the CU's fill
// path (~30 B/clk) is paid either as issue time or as barrier time.

template <int KS, int METRIC, int CAP>
__global__ __launch_bounds__(kScanThreads) void scan_kernel(ScanArgs a)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
	constexpr int ROWB = KS * 64 + 16;  // padded LDS row
	constexpr int NB = scan_buffers<KS, CAP>();
	constexpr int kScanCap = CAP, kScanListStride = CAP + 1;
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const DevIndexView &ix = a.ix;

	// work mapping (see header): blocks with equal blockIdx % 8 share an XCD
	const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
	const int qtile = j % a.qtiles, s2i = j / a.qtiles;
	const int split = xcd * a.s2 + s2i;
	const int n_begin = a.node_lo + split * a.nodes_per_split;
	int n_end = n_begin + a.nodes_per_split;
	if (n_end > a.node_hi) n_end = a.node_hi;
	const int ntiles = n_end > n_begin ? (n_end - n_begin + kScanStage - 1) / kScanStage : 0;  // staged tiles

	unsigned char *btile = lds;
	int *bnorm = reinterpret_cast<int *>(lds + (size_t)NB * kScanStage * ROWB);
	unsigned char *wbase = reinterpret_cast<unsigned char *>(bnorm + NB * kScanStage) + (size_t)wave * scan_wave_lds<CAP>();
	unsigned long long *list = reinterpret_cast<unsigned long long *>(wbase);
	unsigned long long *thr_s = list + kScanWR * kScanListStride;
	int *cnt = reinterpret_cast<int *>(thr_s + kScanWR);
	int *qq_s = cnt + kScanWR;

	// ---- this wave's 32 queries -> A fragments (row = l & 15, k bytes [64 ks + 16 (l>>4), +16))
	const int qbase = qtile * kScanQB + wave * kScanWR;
	v4i qf[2][KS];
#pragma unroll
	for (int rt = 0; rt < 2; rt++) {
		const int qrow = qbase + rt * 16 + (lane & 15);
		const int8_t *src = a.q + (size_t)(qrow < a.nq ? qrow : 0) * (size_t)ix.dim;
#pragma unroll
		for (int ks = 0; ks < KS; ks++) {
			// unconditional loads from a clamped address, then a select: a branch around each load would
			// make every one of them a dependent round trip
			int w[4];
#pragma unroll
			for (int t = 0; t < 4; t++) {
				const int b = ks * 64 + (lane >> 4) * 16 + 4 * t;
				const bool ok = qrow < a.nq && b < ix.dim;
				const int v = *reinterpret_cast<const int *>(src + (b < ix.dim ? b : 0));
				w[t] = ok ? v : 0;
			}
			qf[rt][ks] = v4i{w[0], w[1], w[2], w[3]};
		}
	}
	// lanes 0..31 own row `lane` of the selection structures
	const bool owner = lane < kScanWR;
	const int my_qrow = qbase + (lane & (kScanWR - 1));
	if (owner) {  // |q|^2 of row `lane`
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
	unsigned long long ext = ~0ull;  // smallest bound other splits published for row `lane` (owner lanes)
	// per lane: the 32-bit bound g of the 8 rows its accumulators belong to, j = rt * 4 + reg:
	// row = rt * 16 + (lane >> 4) * 4 + reg
	int g_r[8];
#pragma unroll
	for (int i = 0; i < 8; i++) g_r[i] = kScanNoBound;
	auto row_of = [&](int jx) -> int { return (jx >> 2) * 16 + (lane >> 4) * 4 + (jx & 3); };

	// thr_s (exact 64-bit bound per row, LDS) -> g_r (32-bit fast test, registers)
	auto refresh_bounds = [&]() {
#pragma unroll
		for (int i = 0; i < 8; i++) {
			const int row = row_of(i);
			const unsigned long long t = thr_s[row];
			const int tv = (int)((unsigned)(t >> 32) ^ 0x80000000u);
			int g = kScanNoBound;
			if (t != ~0ull) g = METRIC == 0 ? ((qq_s[row] - tv + 1) >> 1) : -tv;
			g_r[i] = g;
		}
	};
	// Compact the rows in `rows` (bit r = row r) to their best min(n, k) keys, sorted ascending.
	// Lane j holds key j of the row; its rank is the number of smaller keys (keys are unique).
	auto compact_rows = [&](unsigned rows) {
		while (rows) {
			const int R = __builtin_ctz(rows);
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

	const int row_pieces_valid = ix.row_stride >> 4;

	// ---- LDS-DMA staging.  A wave-instruction writes 64 consecutive 16-byte slots of LDS (M0 base +
	// lane * 16) from per-lane global addresses, so the padded, pair-swapped tile image is filled
	// slot by slot: slot s = row s / (KS*4+1), position s % (KS*4+1); position p holds piece
	// p ^ flip(row), the last position of a row is the pad.  64 rows x (KS*4+1) slots = KS*4+1
	// instructions per tile; one more (4 bytes per lane) brings the 64 norms.  Lanes with nothing
	// to fetch read a zero block (rows past the end: norm -1).
	// Issuing them costs the issuing wave ~150 cycles each (measured, whichever way a tile is staged
	// the LDS fill is paid in issue time), and a wave issues in order, so the two waves of a SIMD
	// must not do it at the same time: waves 0-3 (one per SIMD) fetch the even tiles, waves 4-7 the
	// odd ones, right after the barrier, and the partner has the matrix pipe to itself meanwhile
	// (measured against every wave issuing its share of every tile, staggered by half a tile: 8 %
	// faster).  Full tiles of full rows take the short form: uniform base + per-lane offset.
	constexpr int SLOTS_ROW = KS * 4 + 1;
	constexpr int DMA_PER_TILE = kScanStage * SLOTS_ROW / 64;
	constexpr int DMA_WAVES = 4;
	const int dma_grp = wave >> 2, dma_w = wave & 3;
	constexpr int DMA_MAX = (DMA_PER_TILE + DMA_WAVES - 1) / DMA_WAVES;
	int dma_off[DMA_MAX], dma_row[DMA_MAX];
	const bool full_rows = row_pieces_valid >= KS * 4;
#pragma unroll
	for (int i = 0; i < DMA_MAX; i++) {
		const int slot = (dma_w + DMA_WAVES * i) * 64 + lane;
		const int r = slot / SLOTS_ROW, cpos = slot - r * SLOTS_ROW;
		const int c = cpos ^ scan_row_flip(r);
		dma_row[i] = r;
		// pad slots are never read: with full rows they fetch piece 0 of their row, so that no lane needs the zero block
		dma_off[i] = (cpos < KS * 4 && c < row_pieces_valid) ? r * ix.row_stride + c * 16 : (full_rows ? r * ix.row_stride : -1);
	}
	using lds_ptr_t = __attribute__((address_space(3))) void *;
	using glb_ptr_t = __attribute__((address_space(1))) void *;
	auto dma_issue_one = [&](int i, int tile, int bufi) {
		const int node0 = n_begin + tile * kScanStage;
		const int rows_left = n_end - node0;
		const int8_t *base = ix.vec + (size_t)node0 * (size_t)ix.row_stride;
		unsigned char *dst = btile + (size_t)bufi * kScanStage * ROWB;
		{
			const int j = dma_w + DMA_WAVES * i;
			if (j < DMA_PER_TILE) {  // wave-uniform
				if (full_rows && rows_left >= kScanStage) {  // wave-uniform: uniform base + 32-bit lane offset
					__builtin_amdgcn_global_load_lds((glb_ptr_t)(base + (unsigned)dma_off[i]), (lds_ptr_t)(dst + j * 1024), 16, 0, 0);
				} else {
					const bool ok = dma_off[i] >= 0 && dma_row[i] < rows_left;
					const void *src = ok ? static_cast<const void *>(base + dma_off[i]) : static_cast<const void *>(g_scan_zero);
					__builtin_amdgcn_global_load_lds((glb_ptr_t)(src), (lds_ptr_t)(dst + j * 1024), 16, 0, 0);
				}
			}
		}
		if (i == 0 && dma_w == DMA_WAVES - 1) {
			const int nn = node0 + lane;
			const void *src = nn < n_end ? static_cast<const void *>(a.norm2 + nn) : static_cast<const void *>(&g_scan_neg1);
			__builtin_amdgcn_global_load_lds((glb_ptr_t)(src), (lds_ptr_t)(bnorm + bufi * kScanStage), 4, 0, 0);
		}
	};
	auto dma_issue = [&](int tile, int bufi) {
#pragma unroll
		for (int i = 0; i < DMA_MAX; i++) dma_issue_one(i, tile, bufi);
	};
	// The (rare) elements that pass the fast test are appended to their row's buffer.  C[row][col]:
	// col = lane & 15 (node), row = (lane >> 4) * 4 + reg (query) of tile (rt, ct); vn0/vn1 = norms of
	// this lane's two columns; mg[rt * 2 + ct] = the lane's largest element of that tile; gu = the
	// bounds these accumulators started from.
	auto slow_path = [&](const v4i (&acc)[2][2], const int (&mg)[4], const int (&gu)[8], int node0, int vn0, int vn1) {
		unsigned pend = 0;
#pragma unroll
		for (int t = 0; t < 4; t++) {
			if (__ballot(mg[t] >= 0)) {
#pragma unroll
				for (int i = 0; i < 4; i++) pend |= (acc[t >> 1][t & 1][i] >= 0) ? (1u << (t * 4 + i)) : 0u;
			}
		}
		while (__ballot(pend != 0)) {
			bool ovf = false;
#pragma unroll
			for (int t = 0; t < 4; t++) {
				if (__ballot((pend >> (t * 4)) & 0xfu)) {
#pragma unroll
					for (int i = 0; i < 4; i++) {
						const int e = t * 4 + i;
						const bool pe = (pend >> e) & 1u;
						if (__ballot(pe)) {
							if (pe) {
								const int rt = t >> 1, ct = t & 1;
								const int row = rt * 16 + (lane >> 4) * 4 + i;
								const int slot = atomicAdd(&cnt[row], 1);  // LDS
								if (slot < kScanCap) {
									const int vn = ct ? vn1 : vn0;
									int av = acc[rt][ct][i];
									// keep the key arithmetic inside this (rarely taken) branch
									asm volatile("" : "+v"(av));
									const int c = av + gu[rt * 4 + i] + (METRIC == 0 ? (vn >> 1) : 0);
									const int val = METRIC == 0 ? qq_s[row] + vn - 2 * c : -c;
									list[row * kScanListStride + slot] = scan_key(val, node0 + ct * 16 + (lane & 15));
									pend &= ~(1u << e);
								}
								ovf |= slot + 1 >= kScanCap;
							}
						}
					}
				}
			}
			if (__ballot(ovf)) {  // a full buffer keeps its element pending: compact, tighten the bounds, retry
				const unsigned full = (unsigned)__ballot(owner && cnt[lane & (kScanWR - 1)] >= kScanCap);
				compact_rows(full);
				refresh_bounds();
			}
		}
	};
	auto col_init = [&](int vn) -> int { return vn < 0 ? -(1 << 30) : (METRIC == 0 ? -(vn >> 1) : 0); };
	auto tile_max = [&](const v4i &v) -> int {
		const int m01 = v[0] > v[1] ? v[0] : v[1], m23 = v[2] > v[3] ? v[2] : v[3];
		return m01 > m23 ? m01 : m23;
	};

	// Software pipeline: the MFMAs of pass t are issued together with the fast test of pass t-1
	// (independent instruction streams, one matrix-pipe and one VALU), the rare slow path follows.
	// The B fragments of the first two k-steps of a pass (and its two column norms) are read during
	// the last MFMAs of the pass before it, across the two halves of a staged tile and -- with three
	// buffers, where the next tile has been complete since the previous barrier -- across tiles.
	v4i accA[2][2], accB[2][2];
	int gA[8], gB[8];
#pragma unroll
	for (int rt = 0; rt < 2; rt++) accB[rt][0] = accB[rt][1] = v4i{-1, -1, -1, -1};
#pragma unroll
	for (int i = 0; i < 8; i++) gA[i] = gB[i] = kScanNoBound;
	int pvn0 = -1, pvn1 = -1, pnode0 = 0;
	unsigned long long ext_next = ~0ull;
	for (int t = 0; t < NB && t < ntiles; t++)
		if ((t & 1) == dma_grp) dma_issue(t, t);
	scan_wait_vm<0>();
	// bounds known before the sweep starts (the levels before this one, or splits that are ahead)
	if (owner) {
		if (my_qrow < a.nq) ext = __hip_atomic_load(&a.thr_g[my_qrow], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		thr_s[lane] = ext;
	}
	__syncthreads();
	refresh_bounds();

	const int frag_off = (lane & 15) * ROWB + ((lane >> 4) ^ scan_row_flip(lane & 15)) * 16;
	v4i pf[2][2];  // fragments of k-steps 0 and 1 of the upcoming pass
	int nvn0 = -1, nvn1 = -1;  // its column norms
	auto prefetch = [&](int buf, int half) {
		const unsigned char *bp0 = btile + ((size_t)buf * kScanStage + half * kScanTN) * ROWB + frag_off;
		const unsigned char *bp1 = bp0 + 16 * ROWB;
		pf[0][0] = *reinterpret_cast<const v4i *>(bp0);
		pf[0][1] = *reinterpret_cast<const v4i *>(bp1);
		pf[1][0] = *reinterpret_cast<const v4i *>(bp0 + 64);
		pf[1][1] = *reinterpret_cast<const v4i *>(bp1 + 64);
		const int *bn = bnorm + buf * kScanStage + half * kScanTN;
		nvn0 = bn[lane & 15];
		nvn1 = bn[16 + (lane & 15)];
	};
	if (ntiles > 0) prefetch(0, 0);

	// one MFMA pass over 32 staged rows (half `half` of buffer `cur`) into accN (started from the
	// bounds gN), with the fast test of the previous pass (accP, started from gP) riding along;
	// (nbuf, nhalf): the pass whose first fragments are fetched during this one, nbuf < 0: none
	auto pass = [&](int tile, int cur, int half, int nbuf, int nhalf, v4i (&accN)[2][2],
	                int (&gN)[8], const v4i (&accP)[2][2], const int (&gP)[8]) {

		const unsigned char *bp0 = btile + ((size_t)cur * kScanStage + half * kScanTN) * ROWB + frag_off;
		const unsigned char *bp1 = bp0 + 16 * ROWB;
		const int vn0 = nvn0, vn1 = nvn1;
		const int c0 = col_init(vn0), c1 = col_init(vn1);
#pragma unroll
		for (int rt = 0; rt < 2; rt++) {
#pragma unroll
			for (int i = 0; i < 4; i++) {
				const int g = g_r[rt * 4 + i];
				gN[rt * 4 + i] = g;
				accN[rt][0][i] = c0 - g;
				accN[rt][1][i] = c1 - g;
			}
		}
		// fast test of the previous pass: the lane's largest element per 16x16 tile (accB starts at -1)
		int mg[4];
#pragma unroll
		for (int t = 0; t < 4; t++) mg[t] = tile_max(accP[t >> 1][t & 1]);
		const int m01 = mg[0] > mg[1] ? mg[0] : mg[1], m23 = mg[2] > mg[3] ? mg[2] : mg[3];
		const int mall = m01 > m23 ? m01 : m23;
		// fragments of k-steps ks+1 and ks+2 are in flight while the MFMAs of ks run
		v4i bf[KS + 2][2];
		bf[0][0] = pf[0][0];
		bf[0][1] = pf[0][1];
		bf[1][0] = pf[1][0];
		bf[1][1] = pf[1][1];
#pragma unroll
		for (int ks = 0; ks < KS; ks++) {
			if (ks + 2 < KS) {
				bf[ks + 2][0] = *reinterpret_cast<const v4i *>(bp0 + (ks + 2) * 64);
				bf[ks + 2][1] = *reinterpret_cast<const v4i *>(bp1 + (ks + 2) * 64);
			}
			if (ks == KS - 2 && nbuf >= 0) prefetch(nbuf, nhalf);
#pragma unroll
			for (int rt = 0; rt < 2; rt++) {
				accN[rt][0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(qf[rt][ks], bf[ks][0], accN[rt][0], 0, 0, 0);
				accN[rt][1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(qf[rt][ks], bf[ks][1], accN[rt][1], 0, 0, 0);
			}
		}
		if (__ballot(mall >= 0)) slow_path(accP, mg, gP, pnode0, pvn0, pvn1);
		pvn0 = vn0;
		pvn1 = vn1;
		pnode0 = n_begin + tile * kScanStage + half * kScanTN;
	};
	int cur = 0;
	for (int tile = 0; tile < ntiles; tile++) {
		const int nxt = cur + 1 == NB ? 0 : cur + 1;
		const bool more = tile + 1 < ntiles;
		if ((tile % kScanPoll) == 0 && owner && my_qrow < a.nq)
			ext_next = __hip_atomic_load(&a.thr_g[my_qrow], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		pass(tile, cur, 0, cur, 1, accA, gA, accB, gB);
		pass(tile, cur, 1, -1, 0, accB, gB, accA, gA);
		if ((tile % kScanPoll) == 1 && __ballot(owner && ext_next < ext)) {  // a tighter bound from another split
			if (owner && ext_next < ext) {
				ext = ext_next;
				if (ext < thr_s[lane]) thr_s[lane] = ext;
			}
			refresh_bounds();
		}
		// tile + 1 must have landed: the waves that fetched it wait for their own operations (they
		// have nothing newer in flight), the barrier covers the rest; tile + 2 (three buffers), fetched
		// by the other group, stays in flight across the barrier; then the buffer of the tile just
		// finished takes tile + NB
		asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
		if (((tile + 1) & 1) == dma_grp) scan_wait_vm<0>();
		__builtin_amdgcn_s_barrier();
		asm volatile("" ::: "memory");
		if (more) prefetch(nxt, 0);
		if (tile + NB < ntiles && ((tile + NB) & 1) == dma_grp) dma_issue(tile + NB, cur);
		cur = nxt;
	}
	if (ntiles > 0) {  // the last pass always ran into accB
		int mg[4];
#pragma unroll
		for (int t = 0; t < 4; t++) mg[t] = tile_max(accB[t >> 1][t & 1]);
		const int m01 = mg[0] > mg[1] ? mg[0] : mg[1], m23 = mg[2] > mg[3] ? mg[2] : mg[3];
		if (__ballot((m01 > m23 ? m01 : m23) >= 0)) slow_path(accB, mg, gB, pnode0, pvn0, pvn1);
	}
	compact_rows((unsigned)__ballot(owner && cnt[lane & (kScanWR - 1)] > 0));  // every row: best min(n, k), sorted

	// ---- per-(query, split) partial lists, sorted ascending
	if (owner && my_qrow < a.nq) {
		const unsigned long long *L = list + lane * kScanListStride;
		const int mylen = cnt[lane];
		const size_t o = ((size_t)my_qrow * a.nslots + a.slot0 + split) * (size_t)a.k;
		for (int i = 0; i < mylen; i++) a.part_key[o + i] = L[i];
		a.part_n[(size_t)my_qrow * a.nslots + a.slot0 + split] = mylen;
	}
}

// k-way merge of the sorted partial lists of one query: one wave per query, lane l owns slots
// l, l + 64, ...; k rounds of "smallest key not yet taken" (keys are unique: ids differ).  The first
// kMergeCache slots of a lane keep their next kMergeDepth keys in registers, loaded up front, so a round is a
// wave reduction over registers; only a slot that wins kMergeDepth times in a row reloads (a round that walks global
// memory costs a dependent L2 round trip: that was 12-33 us per merge, seven merges per sweep).  Slots beyond
// the cached ones (small batches: up to 512 node splits per level) are walked as before.
// FINAL: write ids / values / count.  Otherwise: thr_g[q] = the k-th smallest key seen so far
// ("k keys <= this bound exist"; the later levels visit other nodes, so no later key equals it).
constexpr int kMergeCache = 4, kMergeDepth = 2;

template <bool FINAL>
__global__ __launch_bounds__(256) void scan_merge_kernel(ScanArgs a, int slots_used)
{
	const int lane = threadIdx.x & 63;
	const int qi = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
	if (qi >= a.nq) return;
	const unsigned long long *pk = a.part_key + (size_t)qi * a.nslots * a.k;
	const int32_t *pn = a.part_n + (size_t)qi * a.nslots;
	int cn[kMergeCache], cpos[kMergeCache];  // list length; index of the key in hk[c][0]
	unsigned long long hk[kMergeCache][kMergeDepth];
#pragma unroll
	for (int c = 0; c < kMergeCache; c++) {
		const int s = lane + 64 * c;
		cn[c] = s < slots_used ? pn[s] : 0;
		cpos[c] = 0;
	}
	auto fill = [&](int c) {  // hk[c][*] = keys cpos[c] .. cpos[c] + depth - 1 of the slot
		const unsigned long long *p = pk + (size_t)(lane + 64 * c) * a.k + cpos[c];
#pragma unroll
		for (int d = 0; d < kMergeDepth; d++) hk[c][d] = cpos[c] + d < cn[c] ? p[d] : ~0ull;
	};
#pragma unroll
	for (int c = 0; c < kMergeCache; c++) fill(c);
	unsigned long long last = 0;
	bool have_last = false;
	int out = 0;
	for (; out < a.k; out++) {
		unsigned long long best = hk[0][0];
#pragma unroll
		for (int c = 1; c < kMergeCache; c++) best = hk[c][0] < best ? hk[c][0] : best;
		for (int s = lane + 64 * kMergeCache; s < slots_used; s += 64) {
			const int n = pn[s];
			for (int i = 0; i < n; i++) {
				const unsigned long long key = pk[(size_t)s * a.k + i];
				if (have_last && key <= last) continue;
				best = key < best ? key : best;
				break;
			}
		}
#pragma unroll
		for (int d = 32; d >= 1; d >>= 1) {
			const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)best, d, 64);
			const unsigned hi = (unsigned)__shfl_xor((int)(unsigned)(best >> 32), d, 64);
			const unsigned long long o = ((unsigned long long)hi << 32) | lo;
			best = o < best ? o : best;
		}
		if (best == ~0ull) break;
		last = best;
		have_last = true;
#pragma unroll
		for (int c = 0; c < kMergeCache; c++) {
			const bool won = hk[c][0] == best;  // at most one slot of one lane
#pragma unroll
			for (int d = 0; d + 1 < kMergeDepth; d++) hk[c][d] = won ? hk[c][d + 1] : hk[c][d];
			hk[c][kMergeDepth - 1] = won ? ~0ull : hk[c][kMergeDepth - 1];
			cpos[c] += won ? 1 : 0;
			if (hk[c][0] == ~0ull && cpos[c] < cn[c]) fill(c);  // the cached keys are used up and the list has more (rare)
		}
		if (FINAL && lane == 0) {
			const int val = (int)((unsigned)(best >> 32) ^ 0x80000000u);
			a.out_ids[(size_t)qi * a.k + out] = (int)(unsigned)best;
			a.out_val[(size_t)qi * a.k + out] = a.metric == 0 ? val : -val;
		}
	}
	if (FINAL) {
		if (lane == 0) a.out_count[qi] = out;
		for (int i = out + lane; i < a.k; i += 64) {  // deterministic tail
			a.out_ids[(size_t)qi * a.k + i] = -1;
			a.out_val[(size_t)qi * a.k + i] = 0;
		}
	} else if (lane == 0) {
		a.thr_g[qi] = out == a.k ? last : ~0ull;
	}
}

// Levels.  With S node splits sweeping at once, a split only learns from its own 1/S of the
// nodes (and the k-th keys the others publish, which are no tighter than its own), so a sweep
// costs ~S * k * ln(N / n0) appends per query instead of k * ln(N / n0).  The node range is
// therefore swept in levels [0, n0), [n0, r n0), [r n0, r^2 n0), ...: after each level the k-th
// key over ALL splits and levels so far becomes every row's bound, and the next level -- r - 1
// times as many nodes as everything before it -- produces ~k (r - 1) appends per query in
// total, whatever S is.  Every level writes its own partial lists (slot = level * nsplit +
// split); the final merge takes all of them.  No node is visited twice.
constexpr int kScanLevel0 = 1024;   // nodes of the first level (no bound yet)
constexpr int kScanLevelRatio = 8;  // each later level ends at ratio x its start
constexpr int kScanMaxLevels = 16;
// the levels the wide kernel takes grow by this factor (it pays more per candidate than scan_kernel, but ~100 us per
// launch: measured at 1 M x 512, 4096 queries: x2 2.08 ms, x3 2.04, x4 2.01, x6 2.02, x16 2.20 per sweep)
constexpr int kScanWideRatio = 4;

// ends[i] = end of level i (multiples of kScanStage, last = N); returns the number of levels.
// fine_from > 0: where the next level at `ratio` would hold at least that many nodes, levels grow by fine_ratio only --
// the wide kernel takes those, pays more per candidate than scan_kernel, and a level of x r brings ~k (r - 1)
// candidates per query.
__host__ inline int scan_levels(int N, int (&ends)[kScanMaxLevels], int level0 = kScanLevel0, int ratio = kScanLevelRatio, int fine_from = 0,
                               int fine_ratio = kScanWideRatio)
{
	int n = 0;
	long long e = level0;
	while (e < N && n < kScanMaxLevels - 1) {
		ends[n++] = (int)e;
		e *= (fine_from > 0 && e * (ratio - 1) >= fine_from) ? fine_ratio : ratio;
	}
	ends[n++] = N;
	return n;
}

}  // namespace mvdb
#include "scan_wide.hip.hpp"
#include "scan_seed.hip.hpp"
namespace mvdb {

// The plan of one sweep: per level its node range, which kernel takes it, its node splits and first list slot.
// Levels of at least a.wide nodes (kWideMinLevel when the engine turns it on) go to scan_wide_kernel; the small
// first levels, where almost every element is a candidate, stay with scan_kernel (its candidate path is overlapped
// by the partner wave, the wide kernel's is not).
struct ScanLevel {
	int lo, hi, wide, seed, qtiles, s2, nsplit, slot0;
};
constexpr int kWideMinLevel = 100000;

__host__ inline int scan_plan(int N, int nq, int wide, int seed, int level0, int ratio, int wide_ratio, ScanLevel (&lv)[kScanMaxLevels],
                             int *total_slots)
{
	int ends[kScanMaxLevels];
	const int levels = scan_levels(N, ends, level0, ratio, wide, wide_ratio);
	int lo = 0, slot = 0;
	for (int l = 0; l < levels; l++) {
		ScanLevel &v = lv[l];
		v.lo = lo;
		v.hi = ends[l];
		v.wide = (wide > 0 && v.hi - v.lo >= wide) ? 1 : 0;
		v.seed = (l == 0 && seed && !v.wide && v.hi - v.lo <= kSeedMaxNodes) ? 1 : 0;  // level 0: scan_seed_kernel, one list per query
		v.qtiles = v.wide ? (nq + kWideQB - 1) / kWideQB : (nq + kScanQB - 1) / kScanQB;
		v.s2 = v.wide ? scan_wide_s2_for(nq) : scan_s2_for(nq);
		v.nsplit = v.seed ? 1 : 8 * v.s2;
		v.slot0 = slot;
		slot += v.nsplit;
		lo = ends[l];
	}
	*total_slots = slot;
	return levels;
}

// one sweep = the levels in turn, each followed by the merge that turns its lists into the next level's bounds
template <int KS, int METRIC, int CAP>
inline bool launch_scan_t(const ScanArgs &a0, hipStream_t s)
{
	constexpr bool HAS_WIDE = CAP == kScanCapSmall;  // a.wide implies k <= kWideMaxK <= kScanSmallK
	constexpr size_t lds_std = scan_lds_bytes<KS, CAP, scan_buffers<KS, CAP>()>();
	constexpr int WKS = KS == 6 ? 8 : KS;  // no 6-step wide kernel (the compiler does not unroll it): rows of 257..384 bytes take the 8-step one
	constexpr size_t lds_wide = scan_wide_lds_bytes<WKS>();
	static bool attr_set[16] = {};  // per device, once
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return false;
	if (!attr_set[dev]) {
		if (hipFuncSetAttribute(reinterpret_cast<const void *>(&scan_kernel<KS, METRIC, CAP>), hipFuncAttributeMaxDynamicSharedMemorySize,
		                        (int)lds_std) != hipSuccess)
			return false;
		if (hipFuncSetAttribute(reinterpret_cast<const void *>(&scan_seed_kernel<KS, METRIC>), hipFuncAttributeMaxDynamicSharedMemorySize,
		                        (int)kSeedLdsBytes) != hipSuccess)
			return false;
		if constexpr (HAS_WIDE) {
			if (hipFuncSetAttribute(reinterpret_cast<const void *>(&scan_wide_kernel<WKS, METRIC>), hipFuncAttributeMaxDynamicSharedMemorySize,
			                        (int)lds_wide) != hipSuccess)
				return false;
		}
		attr_set[dev] = true;
	}
	if (a0.wide && (!HAS_WIDE || a0.k > kWideMaxK)) return false;
	ScanLevel lv[kScanMaxLevels];
	int total = 0;
	const int levels = scan_plan(a0.ix.max_elements, a0.nq, a0.wide, a0.seed, a0.level0, a0.level_ratio, a0.wide_ratio, lv, &total);
	if (total > a0.nslots) return false;
	ScanArgs a = a0;
	for (int l = 0; l < levels; l++) {
		a.node_lo = lv[l].lo;
		a.node_hi = lv[l].hi;
		a.qtiles = lv[l].qtiles;
		a.s2 = lv[l].s2;
		a.nsplit = lv[l].nsplit;
		a.slot0 = lv[l].slot0;
		a.nodes_per_split = (((a.node_hi - a.node_lo + a.nsplit - 1) / a.nsplit) + kScanStage - 1) / kScanStage * kScanStage;
		if (lv[l].seed) {
			hipLaunchKernelGGL((scan_seed_kernel<KS, METRIC>), dim3((a.nq + 15) / 16), dim3(256), kSeedLdsBytes, s, a);
		} else if (lv[l].wide) {
			if constexpr (HAS_WIDE)
				hipLaunchKernelGGL((scan_wide_kernel<WKS, METRIC>), dim3(a.qtiles * a.nsplit), dim3(256), lds_wide, s, a);
		} else {
			hipLaunchKernelGGL((scan_kernel<KS, METRIC, CAP>), dim3(a.qtiles * a.nsplit), dim3(kScanThreads), lds_std, s, a);
		}
		if (l + 1 < levels)
			hipLaunchKernelGGL(scan_merge_kernel<false>, dim3((a.nq + 3) / 4), dim3(256), 0, s, a, lv[l].slot0 + lv[l].nsplit);
	}
	hipLaunchKernelGGL(scan_merge_kernel<true>, dim3((a.nq + 3) / 4), dim3(256), 0, s, a, total);
	return hipGetLastError() == hipSuccess;
}

// supported: row_stride <= 512 bytes (dim <= 512)
inline bool launch_scan(const ScanArgs &a, hipStream_t s)
{
	const int ks = (a.ix.row_stride + 63) / 64;
	const int sel = (ks <= 2 ? 0 : ks <= 4 ? 1 : ks <= 6 ? 2 : ks <= 8 ? 3 : 4) * 4 + (a.metric == 0 ? 0 : 2) +
	                (a.k <= kScanSmallK ? 0 : 1);
	switch (sel) {
	case 0: return launch_scan_t<2, 0, kScanCapSmall>(a, s);
	case 1: return launch_scan_t<2, 0, kScanCapLarge>(a, s);
	case 2: return launch_scan_t<2, 1, kScanCapSmall>(a, s);
	case 3: return launch_scan_t<2, 1, kScanCapLarge>(a, s);
	case 4: return launch_scan_t<4, 0, kScanCapSmall>(a, s);
	case 5: return launch_scan_t<4, 0, kScanCapLarge>(a, s);
	case 6: return launch_scan_t<4, 1, kScanCapSmall>(a, s);
	case 7: return launch_scan_t<4, 1, kScanCapLarge>(a, s);
	case 8: return launch_scan_t<6, 0, kScanCapSmall>(a, s);
	case 9: return launch_scan_t<6, 0, kScanCapLarge>(a, s);
	case 10: return launch_scan_t<6, 1, kScanCapSmall>(a, s);
	case 11: return launch_scan_t<6, 1, kScanCapLarge>(a, s);
	case 12: return launch_scan_t<8, 0, kScanCapSmall>(a, s);
	case 13: return launch_scan_t<8, 0, kScanCapLarge>(a, s);
	case 14: return launch_scan_t<8, 1, kScanCapSmall>(a, s);
	case 15: return launch_scan_t<8, 1, kScanCapLarge>(a, s);
	default: break;
	}
	return false;
}

}  // namespace mvdb
