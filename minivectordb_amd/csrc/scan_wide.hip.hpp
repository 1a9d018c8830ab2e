// scan_wide.hip.hpp -- the exact scan's kernel for the large levels of big batches (k <= kWideMaxK, see scan_plan),
// included by scan_mfma.hip.hpp.
//
// Same algorithm, bounds, levels, LDS tile image and LDS-DMA fill as scan_kernel (see there), other shape:
// ONE wave per SIMD that keeps 128 queries as A fragments (256 registers at 512-d: the AccVGPR half of the
// 512-register file), 512 queries per workgroup.  Why: filling LDS costs the CU ~1300 cycles per 64-row tile
// whichever way it is done (the vector-memory return path moves ~30 B/clk per CU), and the MFMA work a staged
// byte is good for is set by the number of queries the CU holds.  256 queries (scan_kernel: 2 waves x 32
// queries per SIMD, the most that fits in 256 registers next to the rest) make a tile 2048 matrix-pipe cycles,
// 512 queries make it 4096, and there are half as many tiles, barriers and DMA instructions per query.  What the
// two-wave shape bought -- a partner wave that keeps the matrix pipe busy while one handles candidates or
// issues VALU instructions -- has to be made up by hand here: nothing is hidden with one wave per SIMD
// (DESIGN.md 4.3 has the measurements behind each choice below).  Row buffers hold 16 keys, two tile buffers.
#pragma once
#include <climits>
#include <type_traits>


namespace mvdb {

constexpr int kWideQB = 512;   // queries per workgroup
constexpr int kWideWR = 128;   // query rows per wave
constexpr int kWideRT = 8;     // 16-row tiles per wave
constexpr int kWideGR = 4;     // row tiles per pass (two groups)
constexpr int kWideCap = 16;   // candidate keys buffered per row
constexpr int kWideMaxK = 12;  // k up to which the buffer leaves room between compactions
constexpr int kWideStride = kWideCap + 1;
constexpr int kWideNB = 2;
constexpr int kScanWideMinQ = 3072;  // batches from this size on take the wide kernel for their large levels (measured: equal at 2048, slower at 1024)

constexpr size_t kWideWaveLds = (size_t)kWideWR * kWideStride * 8 + kWideWR * 8 + kWideWR * 4 + kWideWR * 4 + 2 * kWideWR * 4;

template <int KS>
__host__ __device__ constexpr size_t scan_wide_lds_bytes()  // tiles, their norms, per wave: selection structures + its table of DMA source offsets + 1 KB to stage an accumulator tile
{
	return (size_t)kWideNB * kScanStage * (KS * 64 + 16) + (size_t)kWideNB * kScanStage * 4 + 4 * (kWideWaveLds + (size_t)(KS + 1) * 64 * 4 + 1024);
}

__host__ __device__ inline int scan_wide_s2_for(int nq)
{
	const int qtiles = (nq + kWideQB - 1) / kWideQB;
	int s2 = (256 + qtiles * 8 - 1) / (qtiles * 8);  // aim at >= 256 workgroups
	return s2 < 1 ? 1 : (s2 > 64 ? 64 : s2);
}

// v_mfma_i32_16x16x64_i8 with the A operand in the AccVGPR half of the register file (B, C and D in ArchVGPRs).
// Written as inline assembly because the compiler only ever takes A and B from the ArchVGPR half: with 256
// registers of A fragments it parks most of them in AccVGPRs and copies four registers back before every MFMA,
// and with one wave per SIMD every such VALU instruction costs its full issue time (measured: ~3.6 cycles each,
// nothing is hidden behind the matrix pipe).  Here ALL A fragments live in the 256 AccVGPRs, for good, and the
// accumulators, the bounds and the B fragments in ArchVGPRs, where the candidate test reads them directly.
// What the compiler can no longer see is that these are matrix instructions, so the few hazards that need
// software spacing are kept apart by construction: B comes from LDS reads (waited for by s_waitcnt, which the
// compiler does place for asm operands), an accumulator is read by VALU code no sooner than 5 MFMAs after its
// last MFMA (the test is pinned between scheduling barriers), and wide_mfma_settle() stands wherever that
// distance is not given by the code's shape.
__device__ __forceinline__ void wide_mfma(v4i &d, const v4i &qa, const v4i &b)
{
	asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(d) : "a"(qa), "v"(b));
}
__device__ __forceinline__ void wide_mfma_first(v4i &d, const v4i &qa, const v4i &b, const v4i &c)
{
	asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %3" : "=&v"(d) : "a"(qa), "v"(b), "v"(c));
}
__device__ __forceinline__ int scan_max3(int x, int y, int z)
{
	const int m = x > y ? x : y;
	return m > z ? m : z;
}
// LDS stores and the slot counter of the candidate path, as inline assembly: before an LDS store or atomic the
// compiler waits for EVERY outstanding LDS-DMA load (it cannot tell that the tile buffers and the selection
// structures never overlap), i.e. for the tile fill in flight -- a full memory round trip per candidate call.
// LDS operations of a wave execute in order, so later plain loads see these stores without further ado.
__device__ __forceinline__ unsigned wide_lds_off(const void *p) { return (unsigned)(size_t)p; }  // low half of a flat LDS address = LDS offset
__device__ __forceinline__ int wide_lds_add1(int *p)
{
	int old;
	asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(old) : "v"(wide_lds_off(p)), "v"(1) : "memory");
	return old;
}
__device__ __forceinline__ void wide_lds_store64(unsigned long long *p, unsigned long long v)
{
	asm volatile("ds_write_b64 %0, %1" ::"v"(wide_lds_off(p)), "v"(v) : "memory");
}
__device__ __forceinline__ void wide_lds_store128(void *p, const v4i &v)
{
	asm volatile("ds_write_b128 %0, %1" ::"v"(wide_lds_off(p)), "v"(v) : "memory");
}
__device__ __forceinline__ void wide_mfma_settle() { asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); }

template <int KS, int METRIC>
__global__ __launch_bounds__(256) void scan_wide_kernel(ScanArgs a)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
	constexpr int ROWB = KS * 64 + 16;
	constexpr int NB = kWideNB;
	const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const DevIndexView &ix = a.ix;

	const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
	const int qtile = jb % a.qtiles, s2i = jb / a.qtiles;
	const int split = xcd * a.s2 + s2i;
	const int n_begin = a.node_lo + split * a.nodes_per_split;
	int n_end = n_begin + a.nodes_per_split;
	if (n_end > a.node_hi) n_end = a.node_hi;
	const int ntiles = n_end > n_begin ? (n_end - n_begin + kScanStage - 1) / kScanStage : 0;

	constexpr size_t WAVE_LDS = kWideWaveLds + (size_t)(KS + 1) * 64 * 4 + 1024;
	unsigned char *btile = lds;
	int *bnorm = reinterpret_cast<int *>(lds + (size_t)NB * kScanStage * ROWB);
	unsigned char *wbase = reinterpret_cast<unsigned char *>(bnorm + NB * kScanStage) + (size_t)wave * WAVE_LDS;
	unsigned long long *list = reinterpret_cast<unsigned long long *>(wbase);
	unsigned long long *thr_s = list + kWideWR * kWideStride;
	int *cnt = reinterpret_cast<int *>(thr_s + kWideWR);
	int *qq_s = cnt + kWideWR;
	int *dma_tab = reinterpret_cast<int *>(wbase + kWideWaveLds);  // [KS + 1][64]: source offset of this lane's slot in DMA instruction i
	int *stage = dma_tab + (KS + 1) * 64;  // [64][4]: one accumulator tile, for the candidate path
	int *G_s = qq_s + kWideWR;  // [2][128]: two generations of the rows' bounds in the form the MFMA C operand takes (below); 16-byte aligned

	// ---- this wave's 128 queries -> A fragments
	const int qbase = qtile * kWideQB + wave * kWideWR;
	v4i qf[kWideRT][KS];
#pragma unroll
	for (int rt = 0; rt < kWideRT; rt++) {
		const int qrow = qbase + rt * 16 + (lane & 15);
		const int8_t *src = a.q + (size_t)(qrow < a.nq ? qrow : 0) * (size_t)ix.dim;
#pragma unroll
		for (int ks = 0; ks < KS; ks++) {
			int w[4];
#pragma unroll
			for (int t = 0; t < 4; t++) {
				const int b = ks * 64 + (lane >> 4) * 16 + 4 * t;
				const bool ok = qrow < a.nq && b < ix.dim;
				const int v = *reinterpret_cast<const int *>(src + (b < ix.dim ? b : 0));
				w[t] = ok ? v : 0;
			}
			qf[rt][ks] = v4i{w[0], w[1], w[2], w[3]};
			asm volatile("" : "+a"(qf[rt][ks]));  // into its AccVGPRs, for good
		}
	}
	// lane owns rows lane and lane + 64 of the selection structures
#pragma unroll
	for (int h = 0; h < 2; h++) {
		const int r = lane + 64 * h, qrow = qbase + r;
		int s = 0;
		if (qrow < a.nq) {
			const int8_t *src = a.q + (size_t)qrow * (size_t)ix.dim;
			for (int b = 0; b < ix.dim; b += 4) {
				const int w = *reinterpret_cast<const int *>(src + b);
				s = __builtin_amdgcn_sdot4(w, w, s, false);
			}
		}
		qq_s[r] = s;
		cnt[r] = 0;
		thr_s[r] = ~0ull;
	}
	unsigned long long ext[2] = {~0ull, ~0ull};
	// Bounds.  A row's bound enters the accumulators as the C operand of the first k-step: acc = q.v + G with
	// G = -g (g as in scan_kernel: -ceil((|q|^2 - bound) / 2) for L2), so "candidate" is acc - floor(|v|^2 / 2) >= 0
	// and a pass needs no accumulator set-up.  G of the 32 rows a lane's accumulators belong to is held in registers
	// (row = rt * 16 + (lane >> 4) * 4 + reg); LDS keeps two generations, because a pass is tested while the next
	// one runs and must be decoded with the G it STARTED from: between the start of a pass and the end of its
	// candidate handling there is exactly one refresh slot, so the generation it read is intact until then.
	v4i G[kWideRT];
	int gen = 0;
	bool stale = false;  // thr_s changed since G was built
	auto refresh_bounds = [&]() {
		gen ^= 1;
#pragma unroll
		for (int h = 0; h < 2; h++) {
			const int r = lane + 64 * h;
			const unsigned long long t = thr_s[r];
			const int tv = (int)((unsigned)(t >> 32) ^ 0x80000000u);
			const int g = METRIC == 0 ? ((qq_s[r] - tv + 1) >> 1) : -tv;
			// rows past the batch: never a candidate (their A fragments are zero, so acc = G)
			G_s[gen * kWideWR + r] = qbase + r >= a.nq ? kScanNoBound : (t != ~0ull ? -g : -kScanNoBound);
		}
#pragma unroll
		for (int rt = 0; rt < kWideRT; rt++) {
			G[rt] = *reinterpret_cast<const v4i *>(G_s + gen * kWideWR + rt * 16 + (lane >> 4) * 4);
		}
		wide_mfma_settle();  // G is the C operand of the very next MFMAs
		stale = false;
	};
	// best min(n, k) keys of row R, sorted; the k-th becomes the row's bound (see scan_kernel)
	auto compact_row = [&](int R) {
		int n = __builtin_amdgcn_readfirstlane(cnt[R]);
		if (n > kWideCap) n = kWideCap;
		unsigned long long *L = list + R * kWideStride;
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
			const int h = R >> 6, rl = R & 63;
			const unsigned long long e_sel = h ? ext[1] : ext[0];
			const unsigned long long ext_R =
			    (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)e_sel, rl) |
			    ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(e_sel >> 32), rl) << 32);
			if (lane < n && rank == a.k - 1) {
				thr_s[R] = key < ext_R ? key : ext_R;
				if (key < ext_R && qbase + R < a.nq) atomicMin(&a.thr_g[qbase + R], key);
			}
		}
	};
	auto compact_rows = [&](unsigned long long lo, unsigned long long hi) {
		while (lo) {
			compact_row(__builtin_ctzll(lo));
			lo &= lo - 1;
		}
		while (hi) {
			compact_row(64 + __builtin_ctzll(hi));
			hi &= hi - 1;
		}
	};

	// ---- LDS-DMA fill of a 64-row tile: 64 x (KS*4+1) slots = KS*4+1 instructions + one for the norms
	constexpr int SLOTS_ROW = KS * 4 + 1;
	constexpr int DMA_PER_TILE = kScanStage * SLOTS_ROW / 64;
	constexpr int DMA_MAX = (DMA_PER_TILE + 3) / 4;
	const int row_pieces_valid = ix.row_stride >> 4;
	using lds_ptr_t = __attribute__((address_space(3))) void *;
	using glb_ptr_t = __attribute__((address_space(1))) void *;
	// (slot -> row, piece) is recomputed per instruction: a dozen VALU ops nine times a tile, no registers held
	// Which 16 bytes of the tile a lane moves in instruction i never changes, so the byte offsets sit in a small LDS
	// table (one ds_read per instruction instead of ~25 VALU operations, which the MFMA loop has no room for).
	// Slots that hold nothing (the pad, pieces past the row) read SOME valid bytes: a pad is never read back and the
	// queries are zero wherever a row has no piece; rows past the range (last tile) are clamped to the last valid
	// piece and switched off through their norm.  The 33rd instruction of a tile is issued by all four waves and so
	// are the norms -- same bytes to the same place -- which keeps the issue free of branches.
	static_assert(DMA_MAX == KS + 1, "table size");
#pragma unroll
	for (int i = 0; i < DMA_MAX; i++) {
		int j = wave + 4 * i;
		j = j < DMA_PER_TILE ? j : DMA_PER_TILE - 1;
		const int slot = j * 64 + lane;
		const int r = slot / SLOTS_ROW, cpos = slot - r * SLOTS_ROW;
		const int c = cpos ^ scan_row_flip(r);
		const bool piece = cpos < KS * 4 && c < row_pieces_valid;
		dma_tab[i * 64 + lane] = r * ix.row_stride + (piece ? c * 16 : 0);
	}
	auto dma_tab_read = [&](int i) -> unsigned { return (unsigned)dma_tab[i * 64 + lane]; };
	auto dma_issue_one = [&](int i, int tile, int bufi, unsigned t) {  // t = dma_tab_read(i), fetched ahead of time in the tile loop
		const int node0 = n_begin + tile * kScanStage;
		const int8_t *base = ix.vec + (size_t)node0 * (size_t)ix.row_stride;
		const unsigned last = (unsigned)((n_end - node0) * ix.row_stride - 16);
		unsigned char *dst = btile + (size_t)bufi * kScanStage * ROWB;
		int j = wave + 4 * i;
		j = j < DMA_PER_TILE ? j : DMA_PER_TILE - 1;
		__builtin_amdgcn_global_load_lds((glb_ptr_t)(base + (t < last ? t : last)), (lds_ptr_t)(dst + j * 1024), 16, 0, 0);
		if (i == 0) {
			const int nn = node0 + lane;
			__builtin_amdgcn_global_load_lds((glb_ptr_t)(a.norm2 + (nn < n_end ? nn : n_end - 1)), (lds_ptr_t)(bnorm + bufi * kScanStage), 4, 0, 0);
		}
	};

	auto tile_max = [&](const v4i &v) -> int {
		const int m01 = v[0] > v[1] ? v[0] : v[1], m23 = v[2] > v[3] ? v[2] : v[3];
		return m01 > m23 ? m01 : m23;
	};
	auto col_init = [&](int vn) -> int { return vn < 0 ? -(1 << 30) : (METRIC == 0 ? -(vn >> 1) : 0); };
	// Candidates of a finished pass (row tiles grp * 4 .. + 3): append, compact full rows, retry (see
	// scan_kernel::slow_path).  pgen = the generation of G the pass started from; the refresh is left to the caller's
	// slot.  Kept SMALL: it is inlined at every pass site, and with one wave per SIMD an instruction-cache miss is
	// paid in full -- so a 16 x 16 tile that holds a candidate goes through LDS and one runtime loop handles its
	// four elements per lane, instead of 32 unrolled copies of the append.
#define WIDE_DBG(x)
	bool compacted = false;
	auto slow_path = [&](v4i (&acc)[kWideGR][2], int grp, int node0, int vn0, int vn1, int pgen) {
		WIDE_DBG(const unsigned long long s_a = __builtin_amdgcn_s_memtime();)
		asm volatile("s_nop 15" ::: "memory");  // the MFMAs issued last may still be writing (not these accumulators, but the same file)
		const int c0 = col_init(vn0), c1 = col_init(vn1);
		unsigned lmask = 0;  // bit t: this lane's part of 16 x 16 tile t (row tile t >> 1, column tile t & 1) holds a candidate
#pragma unroll
		for (int t = 0; t < kWideGR * 2; t++) lmask |= tile_max(acc[t >> 1][t & 1]) + ((t & 1) ? c1 : c0) >= 0 ? (1u << t) : 0u;
		WIDE_DBG(asm volatile("" ::"v"(lmask)); dc[8] += __builtin_amdgcn_s_memtime() - s_a;)
		while (true) {  // one round per tile flagged by any lane
			WIDE_DBG(const unsigned long long s_b = __builtin_amdgcn_s_memtime();)
			const unsigned long long any = __ballot(lmask != 0);
			if (!any) break;
			const int t = __builtin_ctz((unsigned)__builtin_amdgcn_readlane((int)lmask, __builtin_ctzll(any)));
			lmask &= ~(1u << t);
			v4i *st = reinterpret_cast<v4i *>(stage + lane * 4);
			v4i av4;
			switch (t) {
			case 0: av4 = acc[0][0]; break;
			case 1: av4 = acc[0][1]; break;
			case 2: av4 = acc[1][0]; break;
			case 3: av4 = acc[1][1]; break;
			case 4: av4 = acc[2][0]; break;
			case 5: av4 = acc[2][1]; break;
			case 6: av4 = acc[3][0]; break;
			default: av4 = acc[3][1]; break;
			}
			wide_lds_store128(st, av4);
			const int cc = (t & 1) ? c1 : c0, vn = (t & 1) ? vn1 : vn0;
			unsigned pm = (av4[0] + cc >= 0 ? 1u : 0u) | (av4[1] + cc >= 0 ? 2u : 0u) | (av4[2] + cc >= 0 ? 4u : 0u) | (av4[3] + cc >= 0 ? 8u : 0u);
			const int row0 = (grp * kWideGR + (t >> 1)) * 16 + (lane >> 4) * 4;
			const int node = node0 + (t & 1) * 16 + (lane & 15);
			WIDE_DBG(asm volatile("s_waitcnt lgkmcnt(0)" ::"v"(pm) : "memory"); const unsigned long long s_c = __builtin_amdgcn_s_memtime(); dc[9] += s_c - s_b; dc[11]++;)
			while (__ballot(pm != 0)) {  // every lane with a candidate takes its lowest one; usually one round
				WIDE_DBG(dc[12]++;)
				bool ovf = false;
				if (pm) {
					const int i = __builtin_ctz(pm);
					const int row = row0 + i;
					// four LDS operations in flight at once: with one wave per SIMD each round trip is paid in full
					const int sv = stage[lane * 4 + i], gv = G_s[pgen * kWideWR + row], qv = qq_s[row];
					const int slot = wide_lds_add1(&cnt[row]);
					const int dot = sv - gv;
					const int val = METRIC == 0 ? qv + vn - 2 * dot : -dot;
					if (slot < kWideCap) {
						wide_lds_store64(&list[row * kWideStride + slot], scan_key(val, node));
						pm &= pm - 1;
					}
					ovf = slot + 1 >= kWideCap;
				}
				if (__ballot(ovf)) {
					const unsigned long long lo = __ballot(cnt[lane] >= kWideCap), hi = __ballot(cnt[lane + 64] >= kWideCap);
					compact_rows(lo, hi);
					compacted = true;
					WIDE_DBG(dc[13]++;)
				}
			}
			WIDE_DBG(asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); dc[10] += __builtin_amdgcn_s_memtime() - s_c;)
		}
	};

	unsigned long long ext_next[2] = {~0ull, ~0ull};
	if (ntiles > 0) {
#pragma unroll
		for (int i = 0; i < DMA_MAX; i++) dma_issue_one(i, 0, 0, dma_tab_read(i));
	}
	scan_wait_vm<0>();
#pragma unroll
	for (int h = 0; h < 2; h++) {
		const int qrow = qbase + lane + 64 * h;
		if (qrow < a.nq) ext[h] = __hip_atomic_load(&a.thr_g[qrow], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		thr_s[lane + 64 * h] = ext[h];
	}
	__syncthreads();
	refresh_bounds();
	const int frag_off = (lane & 15) * ROWB + ((lane >> 4) ^ scan_row_flip(lane & 15)) * 16;

	// A pass = 64 queries (four row tiles) x 32 staged rows; four passes per 64-row tile.  Two accumulator sets:
	// the pass in accY is tested (VALU, in the shadow of the MFMAs) while accX is computed.
	v4i accP[kWideGR][2], accQ[kWideGR][2];
#pragma unroll
	for (int rt = 0; rt < kWideGR; rt++) accP[rt][0] = accP[rt][1] = accQ[rt][0] = accQ[rt][1] = v4i{-1, -1, -1, -1};
	v4i bf[2][2];    // B fragments of the current and the next k-step
	int nvn0, nvn1;  // norms of the two staged rows this lane's columns are, of the pass whose first fragments are in bf[0]
	auto load_first = [&](int tile, int cur, int half) {
		const unsigned char *bp0 = btile + ((size_t)cur * kScanStage + half * kScanTN) * ROWB + frag_off;
		bf[0][0] = *reinterpret_cast<const v4i *>(bp0);
		bf[0][1] = *reinterpret_cast<const v4i *>(bp0 + 16 * ROWB);
		const int *bn = bnorm + cur * kScanStage + half * kScanTN;
		const int col0 = n_begin + tile * kScanStage + half * kScanTN + (lane & 15);
		const int v0 = bn[lane & 15], v1 = bn[16 + (lane & 15)];
		nvn0 = col0 < n_end ? v0 : -1;
		nvn1 = col0 + 16 < n_end ? v1 : -1;
	};
	// One pass into accX; returns the test of accY (max over its elements of acc + column term).  DMA instruction
	// i0 + ks of tile `ftile` (into buffer `fill`) goes out in k-step ks; next_half >= 0: the first fragments of that half
	// of this tile are fetched during the last k-step.
	auto compute = [&](auto grp_c, v4i (&accX)[kWideGR][2], v4i (&accY)[kWideGR][2], int yc0, int yc1, int tile, int cur, int half, int ftile,
	                   int fill, int i0, int next_half) -> int {
		constexpr int GRP = decltype(grp_c)::value;
		const unsigned char *bp0 = btile + ((size_t)cur * kScanStage + half * kScanTN) * ROWB + frag_off;
		const unsigned char *bp1 = bp0 + 16 * ROWB;
		int my[2] = {INT_MIN, INT_MIN};  // running maxima of accY, per column tile
#pragma unroll
		for (int ks = 0; ks < KS; ks++) {
			// One wave per SIMD: nothing else hides a wait or a VALU instruction, and the compiler does not know that the
			// asm statements are matrix instructions.  So the step is laid out by hand and pinned with scheduling barriers:
			// the LDS reads for the NEXT step (and the DMA table entry of this one) go out first -- their registers were
			// last used by the step before this one -- and have this step's 8 MFMAs to land; the test of accY (two
			// v_max3 per 16 x 16 tile) and the DMA instruction are dealt out behind single MFMAs.
			const bool dma = i0 + ks < DMA_MAX;
			unsigned tabv = 0;
			__builtin_amdgcn_sched_barrier(0);
			if (ks + 1 < KS) {
				bf[(ks + 1) & 1][0] = *reinterpret_cast<const v4i *>(bp0 + (ks + 1) * 64);
				bf[(ks + 1) & 1][1] = *reinterpret_cast<const v4i *>(bp1 + (ks + 1) * 64);
			} else if (next_half >= 0) {
				load_first(tile, cur, next_half);
			}
			if (dma) tabv = dma_tab_read(i0 + ks);
			__builtin_amdgcn_sched_barrier(0);
			const int T0 = ks * (kWideGR * 2) / KS, T1 = (ks + 1) * (kWideGR * 2) / KS;  // accY tiles tested in this step (all static once unrolled)
			const int NP = (T1 - T0) + 1;                                            // pieces: one per tile + the DMA instruction
#pragma unroll
			for (int m = 0; m < kWideGR * 2; m++) {
				const int rt = m >> 1, ct = m & 1;
				if (ks == 0)
					wide_mfma_first(accX[rt][ct], qf[GRP * kWideGR + rt][ks], bf[0][ct], G[GRP * kWideGR + rt]);
				else
					wide_mfma(accX[rt][ct], qf[GRP * kWideGR + rt][ks], bf[ks & 1][ct]);
#pragma unroll
				for (int p = 0; p < kWideGR * 2 + 1; p++) {
					if (p >= NP || p * (kWideGR * 2) / NP != m) continue;
					if (p == NP - 1) {
						if (dma) dma_issue_one(i0 + ks, ftile, fill, tabv);
					} else {
						const int t = T0 + p;
						const v4i &v = accY[t >> 1][t & 1];
						int &mm = my[t & 1];
						mm = scan_max3(mm, v[0], v[1]);
						mm = scan_max3(mm, v[2], v[3]);
					}
				}
				__builtin_amdgcn_sched_barrier(0);
			}
			// The compiler waits for ALL outstanding LDS reads before an asm statement that uses a loaded register.  Left
			// alone it does so at the next step's first MFMA, right behind that step's own reads; touching the registers
			// HERE puts the wait where everything outstanding was issued 8 MFMAs ago.
			asm volatile("" ::"v"(bf[(ks + 1) & 1][0]), "v"(bf[(ks + 1) & 1][1]), "v"(tabv));
			__builtin_amdgcn_sched_barrier(0);
		}
		const int t0 = my[0] + yc0, t1 = my[1] + yc1;  // (no overflow: |acc| < 2^30, column term in [-2^30, 0])
		return t0 > t1 ? t0 : t1;
	};
	static_assert((KS & 1) == 0, "bf[0] must be free at the last k-step");

	// the pass that waits for its test: where it was, its norms, the generation of G it started from
	bool p_have = false;
	int p_grp = 0, p_node0 = 0, p_vn0 = -1, p_vn1 = -1, p_gen = 0;
	auto handle = [&](v4i (&accY)[kWideGR][2], int mall) {
		if (p_have && __ballot(mall >= 0)) slow_path(accY, p_grp, p_node0, p_vn0, p_vn1, p_gen);
		if (compacted || stale) refresh_bounds();  // the one refresh slot per pass
		compacted = false;
	};
	using grp0_t = std::integral_constant<int, 0>;
	using grp1_t = std::integral_constant<int, 1>;

	WIDE_DBG(const unsigned long long d_t0 = __builtin_amdgcn_s_memtime();)
	for (int tile = 0; tile < ntiles; tile++) {
		const int cur = tile & 1;
		const int fill = cur ^ 1;  // read last in tile - 1 (barrier passed): takes tile + 1 meanwhile
		const int ftile = tile + 1 < ntiles ? tile + 1 : tile;  // (the last tile fills it with itself: no branch in the steps)
		const int tnode0 = n_begin + tile * kScanStage;
		if ((tile % kScanPoll) == 0) {
#pragma unroll
			for (int h = 0; h < 2; h++) {
				const int qrow = qbase + lane + 64 * h;
				if (qrow < a.nq) ext_next[h] = __hip_atomic_load(&a.thr_g[qrow], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			}
		}
		WIDE_DBG(const unsigned long long d_a = __builtin_amdgcn_s_memtime();)
		load_first(tile, cur, 0);
		WIDE_DBG(asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); dc[7] += __builtin_amdgcn_s_memtime() - d_a; dc[0]++;)
#pragma unroll
		for (int half = 0; half < 2; half++) {
			{
				const int vn0 = nvn0, vn1 = nvn1, g0 = gen;
				WIDE_DBG(const unsigned long long d_b = __builtin_amdgcn_s_memtime();)
				const int lm = compute(grp0_t{}, accP, accQ, col_init(p_vn0), col_init(p_vn1), tile, cur, half, ftile, fill, half * 2 * KS, half);
				WIDE_DBG(const unsigned long long d_c = __builtin_amdgcn_s_memtime(); dc[2] += d_c - d_b; dc[1] += (p_have && __ballot(lm >= 0)) ? 1 : 0;)
				handle(accQ, lm);
				WIDE_DBG(dc[3] += __builtin_amdgcn_s_memtime() - d_c;)
				p_have = true, p_grp = 0, p_node0 = tnode0 + half * kScanTN, p_vn0 = vn0, p_vn1 = vn1, p_gen = g0;
			}
			{
				const int vn0 = nvn0, vn1 = nvn1, g0 = gen;
				WIDE_DBG(const unsigned long long d_b = __builtin_amdgcn_s_memtime();)
				const int lm = compute(grp1_t{}, accQ, accP, col_init(p_vn0), col_init(p_vn1), tile, cur, half, ftile, fill, (half * 2 + 1) * KS,
				                         half == 0 ? 1 : -1);
				WIDE_DBG(const unsigned long long d_c = __builtin_amdgcn_s_memtime(); dc[2] += d_c - d_b; dc[1] += __ballot(lm >= 0) ? 1 : 0;)
				handle(accP, lm);
				WIDE_DBG(dc[3] += __builtin_amdgcn_s_memtime() - d_c;)
				p_grp = 1, p_node0 = tnode0 + half * kScanTN, p_vn0 = vn0, p_vn1 = vn1, p_gen = g0;
			}
		}
		if ((tile % kScanPoll) == 1 && __ballot(ext_next[0] < ext[0] || ext_next[1] < ext[1])) {
#pragma unroll
			for (int h = 0; h < 2; h++) {
				if (ext_next[h] < ext[h]) {
					ext[h] = ext_next[h];
					if (ext[h] < thr_s[lane + 64 * h]) thr_s[lane + 64 * h] = ext[h];
				}
			}
			stale = true;  // taken up in the next refresh slot
		}
		WIDE_DBG(const unsigned long long d_d = __builtin_amdgcn_s_memtime();)
		asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
		scan_wait_vm<0>();  // this wave's share of tile + 1 has landed; the barrier covers the others'
		WIDE_DBG(const unsigned long long d_e = __builtin_amdgcn_s_memtime(); dc[4] += d_e - d_d;)
		__builtin_amdgcn_s_barrier();
		asm volatile("" ::: "memory");
		WIDE_DBG(dc[5] += __builtin_amdgcn_s_memtime() - d_e;)
	}
	wide_mfma_settle();
	if (p_have) {  // the last pass (in accQ)
		handle(accQ, 0);  // (slow_path looks for the candidates itself)
	}
	compact_rows(__ballot(cnt[lane] > 0), __ballot(cnt[lane + 64] > 0));

#pragma unroll
	for (int h = 0; h < 2; h++) {
		const int r = lane + 64 * h, qrow = qbase + r;
		if (qrow < a.nq) {
			const unsigned long long *L = list + r * kWideStride;
			const int mylen = cnt[r];
			const size_t o = ((size_t)qrow * a.nslots + a.slot0 + split) * (size_t)a.k;
			for (int i = 0; i < mylen; i++) a.part_key[o + i] = L[i];
			a.part_n[(size_t)qrow * a.nslots + a.slot0 + split] = mylen;
		}
	}
}

}  // namespace mvdb
