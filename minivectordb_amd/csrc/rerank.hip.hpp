// rerank.hip.hpp -- float32 exact-L2 rerank fused with a wave-shuffle top-k, plus query prep.
//
// Replaces the host loops of src/mini-vector-db.ts:
//   normalizeF32InPlace :963-970, quantizeToI8 :972-984   -> prep_kernel
//   candidate filter :606-621, l2SqF32 :986-993, sort (l2,id) + cut :626-651 -> rerank_kernel
// The reference accumulates in JS doubles, strictly in element order.  To be bit-identical
// (so the final topK ids are ALWAYS the reference's, ties included) every candidate's sum is
// formed by ONE lane in element order with separate f64 multiply and add (no FMA contraction).
// Rows are still fetched coalesced: a wave gathers 64 rows x 128 B per step (8 lanes x 16 B per
// row), transposes the tile through padded LDS, and each lane then walks its own row.  The next
// step's pieces are already in flight (register ring) while the current tile is walked.
// Measured on MI355X (1000 queries x 1200 candidates x 384 f32, random rows of a 1M store):
// 128-B pieces 0.35 ms = 5.2 TB/s; 64-B pieces 0.55 ms (half-used L2 lines); no prefetch 0.71 ms.
#pragma once

#include <hip/hip_runtime.h>

#include "mvdb_hip.h"

namespace mvdb {

// waves per workgroup: 4 by default; 2 when the rerank is meant to share CUs with a running
// traversal (34 KB of LDS instead of 53 KB: it then fits beside four traversal workgroups)
constexpr int kRerankWavesDefault = 4;
#ifndef MVDB_RERANK_TILE_COLS
#define MVDB_RERANK_TILE_COLS 32
#endif
#ifndef MVDB_RERANK_DEPTH
#define MVDB_RERANK_DEPTH 1
#endif
constexpr int kTileCols = MVDB_RERANK_TILE_COLS;  // floats per row per step (128 B = one L2 line)
constexpr int kTileRowBytes = kTileCols * 4 + 16;  // padded rows (144 B): conflict-free b128 reads
constexpr int kTileLanesPerRow = kTileCols / 4;    // lanes that share one row piece in a gather pass
constexpr int kTileRowsPerPass = 64 / kTileLanesPerRow;
constexpr int kTilePasses = 64 / kTileRowsPerPass;
constexpr int kTileDepth = MVDB_RERANK_DEPTH;                    // row pieces kept in flight per lane (register ring)
constexpr int kMaxTopK = 256;

struct RerankArgs {
	const float *q;           // nq x dim (already normalised)
	const float *base;        // rows x dim (HBM or pinned host)
	int64_t rows;
	int dim, nq;
	const int32_t *cand;      // nq x stride, ANN order
	const int32_t *cand_count;  // nq
	int stride, ann_k, topk;
	const uint32_t *allow;    // bitmap over ids or null
	int32_t *out_ids;         // nq x topk
	double *out_l2;           // nq x topk
	int32_t *out_n;           // nq
	// File mode (preloadVectors = false, the store stays on the host): a kept candidate whose row lies past the
	// store is an ERROR there (readF32Vectors, mini-vector-db.ts:1120-1124 "Vector store short read"), where the
	// preloaded store skips it (:1090).  strict_store raises bit 1 of the handle's error word for it.
	int *err;
	int strict_store;
};

// LDS: q_s[dim_pad] f32 | tiles[waves][64*144] | key_l2[cap] f64 | key_id[cap] i32 | misc
__host__ __device__ inline size_t rerank_lds_bytes(int dim, int cap, int kRerankWaves = kRerankWavesDefault)
{
	size_t dim_pad = ((size_t)dim + kTileCols - 1) / kTileCols * kTileCols;
	size_t b = dim_pad * 4;
	b += (size_t)kRerankWaves * 64 * kTileRowBytes;
	b = (b + 7) & ~(size_t)7;
	b += (size_t)cap * 8 + (size_t)cap * 4;
	b += 64 * 8 + 64 * 4 + 64;  // cross-wave reduction scratch + counters
	return b;
}

struct KeyPair {
	double l2;
	int id;
};
__device__ __forceinline__ bool key_less(double al2, int aid, double bl2, int bid)
{
	return al2 < bl2 || (al2 == bl2 && aid < bid);  // :637  a.l2 - b.l2 || a.id - b.id
}

template <int kRerankWaves>
__global__ __launch_bounds__(kRerankWaves * 64) void rerank_kernel(RerankArgs a, int cap)
{
#pragma clang fp contract(off)
	constexpr int kRerankThreads = kRerankWaves * 64;
	extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int qi = blockIdx.x;
	const int dim = a.dim;
	const int dim_pad = (dim + kTileCols - 1) / kTileCols * kTileCols;

	float *q_s = reinterpret_cast<float *>(lds_raw);
	unsigned char *tile = lds_raw + (size_t)dim_pad * 4 + (size_t)wave * 64 * kTileRowBytes;
	size_t off = (size_t)dim_pad * 4 + (size_t)kRerankWaves * 64 * kTileRowBytes;
	off = (off + 7) & ~(size_t)7;
	double *key_l2 = reinterpret_cast<double *>(lds_raw + off);
	int *key_id = reinterpret_cast<int *>(lds_raw + off + (size_t)cap * 8);
	double *red_l2 = reinterpret_cast<double *>(lds_raw + off + (size_t)cap * 12);
	int *red_id = reinterpret_cast<int *>(red_l2 + 64);
	int *misc = red_id + 64;

	for (int i = tid; i < dim_pad; i += kRerankThreads) q_s[i] = i < dim ? a.q[(size_t)qi * dim + i] : 0.f;

	// ---- filter in ANN order, keep at most ann_k (:606-621); compact kept ids into key_id[]
	int cc = a.cand_count[qi];
	if (cc < 0) {  // the traversal aborted this query (traverse_report_abort): the failure stays visible, block-uniform
		if (tid == 0) a.out_n[qi] = -1;
		for (int j = tid; j < a.topk; j += kRerankThreads) {
			a.out_ids[(size_t)qi * a.topk + j] = -1;
			a.out_l2[(size_t)qi * a.topk + j] = 0.0;
		}
		return;
	}
	if (cc > a.stride) cc = a.stride;
	if (tid == 0) misc[0] = 0;
	__syncthreads();
	const int32_t *cq = a.cand + (size_t)qi * a.stride;
	for (int base = 0; base < cc; base += kRerankThreads) {
		const int j = base + tid;
		int id = -1;
		bool keep = false;
		if (j < cc) {
			id = cq[j];
			keep = true;
			if (a.allow) keep = id >= 0 && ((a.allow[(unsigned)id >> 5] >> (id & 31)) & 1u);
		}
		// ordered compaction across the block: per-wave ballot + wave offsets
		const unsigned long long m = __ballot(keep);
		if (lane == 0) misc[1 + wave] = __builtin_popcountll(m);
		__syncthreads();
		int before = misc[0];
		for (int w = 0; w < wave; w++) before += misc[1 + w];
		const int pos = before + __builtin_popcountll(m & ((1ull << lane) - 1));
		if (keep && pos < a.ann_k) key_id[pos] = id;
		__syncthreads();
		if (tid == 0) {
			int tot = misc[0];
			for (int w = 0; w < kRerankWaves; w++) tot += misc[1 + w];
			misc[0] = tot;
		}
		__syncthreads();
	}
	int nkept = misc[0];
	if (nkept > a.ann_k) nkept = a.ann_k;

	// ---- exact f64 L2^2 per candidate; each wave owns groups of 64 candidates
	// The gather is software-pipelined: kTileDepth column steps of every row are in flight in
	// registers while the wave walks the step that already landed in its LDS tile.
	const int g = lane / kTileLanesPerRow, s = lane % kTileLanesPerRow;
	for (int c0 = wave * 64; c0 < nkept; c0 += kRerankWaves * 64) {
		const int my = c0 + lane;
		const int my_id = my < nkept ? key_id[my] : -1;
		const bool my_ok = my_id >= 0 && (int64_t)my_id < a.rows;  // rows beyond the store are skipped (:1090)
		if (a.strict_store && a.err && my_id >= 0 && (int64_t)my_id >= a.rows)
			__hip_atomic_fetch_or(a.err, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
		// pass p brings rows p*kTileRowsPerPass+g, 16 B at column piece s
		const float *rowp[kTilePasses];
#pragma unroll
		for (int p = 0; p < kTilePasses; p++) {
			const int r = p * kTileRowsPerPass + g;
			const int rid = __shfl(my_id, r);
			const bool ok = __shfl((int)my_ok, r);
			rowp[p] = ok ? a.base + (size_t)rid * (size_t)dim : nullptr;
		}
		float4 ring[kTileDepth][kTilePasses];
		auto fetch = [&](int col, float4(&dst)[kTilePasses]) {
			const int e = col + s * 4;
#pragma unroll
			for (int p = 0; p < kTilePasses; p++) {
				float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
				if (rowp[p] && e < dim) {
					const float *src = rowp[p] + e;
					if (e + 4 <= dim) {
						typedef float f32x4_t __attribute__((ext_vector_type(4)));
						const f32x4_t t = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t *>(src));
						v = make_float4(t.x, t.y, t.z, t.w);
					} else {
						v.x = src[0];
						if (e + 1 < dim) v.y = src[1];
						if (e + 2 < dim) v.z = src[2];
					}
				}
				dst[p] = v;
			}
		};
#pragma unroll
		for (int d = 0; d < kTileDepth; d++)
			if (d * kTileCols < dim_pad) fetch(d * kTileCols, ring[d]);
		double acc = 0.0;
		for (int col0 = 0; col0 < dim_pad; col0 += kTileDepth * kTileCols) {
#pragma unroll
			for (int d = 0; d < kTileDepth; d++) {
				const int col = col0 + d * kTileCols;
				if (col < dim_pad) {
#pragma unroll
					for (int p = 0; p < kTilePasses; p++)
						*reinterpret_cast<float4 *>(tile + (p * kTileRowsPerPass + g) * kTileRowBytes + s * 16) =
						    ring[d][p];
					if (col + kTileDepth * kTileCols < dim_pad) fetch(col + kTileDepth * kTileCols, ring[d]);
					// each lane walks its own row in element order (sequential f64, :987-991)
#pragma unroll
					for (int c = 0; c < kTileCols / 4; c++) {
						const float4 v =
						    *reinterpret_cast<const float4 *>(tile + lane * kTileRowBytes + c * 16);
						const float4 qv = *reinterpret_cast<const float4 *>(q_s + col + c * 4);
						const int e = col + c * 4;
						if (e < dim) {
							double dd = (double)qv.x - (double)v.x;
							acc = acc + dd * dd;
						}
						if (e + 1 < dim) {
							double dd = (double)qv.y - (double)v.y;
							acc = acc + dd * dd;
						}
						if (e + 2 < dim) {
							double dd = (double)qv.z - (double)v.z;
							acc = acc + dd * dd;
						}
						if (e + 3 < dim) {
							double dd = (double)qv.w - (double)v.w;
							acc = acc + dd * dd;
						}
					}
				}
			}
		}
		if (my < nkept) {
			key_l2[my] = my_ok ? acc : __builtin_inf();
			if (!my_ok) key_id[my] = 0x7fffffff;  // dropped candidate
		}
	}
	__syncthreads();

	// ---- top-k by (l2, id): k rounds of block-wide arg-min (wave shuffles + one LDS hop)
	int emitted = 0;
	for (int round = 0; round < a.topk; round++) {
		double bl2 = __builtin_inf();
		int bid = 0x7fffffff, bpos = -1;
		for (int j = tid; j < nkept; j += kRerankThreads) {
			const double l = key_l2[j];
			const int id = key_id[j];
			if (id != 0x7fffffff && (bpos < 0 || key_less(l, id, bl2, bid))) {
				bl2 = l;
				bid = id;
				bpos = j;
			}
		}
#pragma unroll
		for (int o = 32; o >= 1; o >>= 1) {
			const double ol2 = __shfl_xor(bl2, o);
			const int oid = __shfl_xor(bid, o);
			const int opos = __shfl_xor(bpos, o);
			if (opos >= 0 && (bpos < 0 || key_less(ol2, oid, bl2, bid))) {
				bl2 = ol2;
				bid = oid;
				bpos = opos;
			}
		}
		if (lane == 0) {
			red_l2[wave] = bl2;
			red_id[wave] = bid;
			red_id[8 + wave] = bpos;
		}
		__syncthreads();
		if (tid == 0) {
			double wl2 = red_l2[0];
			int wid = red_id[0], wpos = red_id[8];
			for (int w = 1; w < kRerankWaves; w++) {
				const int p = red_id[8 + w];
				if (p >= 0 && (wpos < 0 || key_less(red_l2[w], red_id[w], wl2, wid))) {
					wl2 = red_l2[w];
					wid = red_id[w];
					wpos = p;
				}
			}
			misc[8] = wpos;
			if (wpos >= 0) {
				a.out_ids[(size_t)qi * a.topk + round] = wid;
				a.out_l2[(size_t)qi * a.topk + round] = wl2;
				key_id[wpos] = 0x7fffffff;  // consumed
			}
		}
		__syncthreads();
		if (misc[8] < 0) break;
		emitted++;
	}
	if (tid == 0) a.out_n[qi] = emitted;
	for (int j = emitted + tid; j < a.topk; j += kRerankThreads) {  // deterministic tail
		a.out_ids[(size_t)qi * a.topk + j] = -1;
		a.out_l2[(size_t)qi * a.topk + j] = 0.0;
	}
}

// ---- query prep: normalizeF32InPlace + quantizeToI8, one wave per query ---------------
// The row is staged in LDS with coalesced loads; lane 0 forms the norm strictly in element
// order (f64, :964-965); all lanes then scale and quantise their elements.
constexpr int kPrepMaxDim = 4096;
__global__ __launch_bounds__(64) void prep_kernel(const float *q_in, int nq, int dim, float *q_norm, int8_t *q_i8)
{
#pragma clang fp contract(off)
	__shared__ float row[kPrepMaxDim];
	__shared__ double norm_s;
	const int qi = blockIdx.x, lane = threadIdx.x;
	if (qi >= nq) return;
	const float *src = q_in + (size_t)qi * dim;
	for (int i = lane; i < dim; i += 64) row[i] = src[i];
	__syncthreads();
	if (lane == 0) {
		double normSq = 0.0;
		for (int i = 0; i < dim; i++) {
			const double x = (double)row[i];
			normSq = normSq + x * x;
		}
		norm_s = normSq;
	}
	__syncthreads();
	const double normSq = norm_s;
	const bool scale = !(normSq <= 0.0);   // :966 (NaN norm falls through, as in JS)
	const double inv = 1.0 / sqrt(normSq);  // :967
	for (int i = lane; i < dim; i += 64) {
		float v = row[i];
		if (scale) v = (float)((double)v * inv);  // :968 f64 product rounded to f32 on store
		if (q_norm) q_norm[(size_t)qi * dim + i] = v;
		if (q_i8) {  // :974-981
			double x = (double)v;
			if (x > 1.0) x = 1.0;
			else if (x < -1.0) x = -1.0;
			double r = floor(x * 127.0 + 0.5);  // Math.round: ties toward +inf
			if (r > 127.0) r = 127.0;
			else if (r < -127.0) r = -127.0;
			q_i8[(size_t)qi * dim + i] = (r != r) ? (int8_t)0 : (int8_t)(int)r;
		}
	}
}

// ---- merge of per-shard top-k lists: one thread per query ------------------
// Part p's arrays start p * stride bytes after the base pointers (dense [part][nq][topk] arrays, or the
// packed per-rank blocks {l2 f64 | ids i32 | n i32} that one all-gather lays out back to back).
__global__ void merge_topk_kernel(const int32_t *ids, const double *l2, const int32_t *n, size_t stride_ids, size_t stride_l2,
                                  size_t stride_n, int parts, int nq, int topk, int32_t *out_ids, double *out_l2,
                                  int32_t *out_n)
{
	const int qi = blockIdx.x * blockDim.x + threadIdx.x;
	if (qi >= nq) return;
	auto ids_of = [&](int p) { return reinterpret_cast<const int32_t *>(reinterpret_cast<const char *>(ids) + (size_t)p * stride_ids); };
	auto l2_of = [&](int p) { return reinterpret_cast<const double *>(reinterpret_cast<const char *>(l2) + (size_t)p * stride_l2); };
	auto n_of = [&](int p) { return reinterpret_cast<const int32_t *>(reinterpret_cast<const char *>(n) + (size_t)p * stride_n); };
	int head[64];
	for (int p = 0; p < parts; p++) head[p] = 0;
	int emitted = 0;
	bool failed = false;  // a shard reported an aborted query (n = -1): the merged answer is not valid either
	for (int p = 0; p < parts; p++) failed |= n_of(p)[qi] < 0;
	for (; !failed && emitted < topk; emitted++) {
		int bp = -1;
		double bl = 0;
		int bi = 0;
		for (int p = 0; p < parts; p++) {
			int np = n_of(p)[qi];
			if (np > topk) np = topk;
			if (head[p] >= np) continue;
			const size_t o = (size_t)qi * topk + head[p];
			const double cl = l2_of(p)[o];
			const int ci = ids_of(p)[o];
			if (bp < 0 || key_less(cl, ci, bl, bi)) {
				bp = p;
				bl = cl;
				bi = ci;
			}
		}
		if (bp < 0) break;
		head[bp]++;
		out_ids[(size_t)qi * topk + emitted] = bi;
		out_l2[(size_t)qi * topk + emitted] = bl;
	}
	out_n[qi] = failed ? -1 : emitted;
}

}  // namespace mvdb
