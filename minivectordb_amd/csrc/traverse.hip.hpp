// traverse.hip.hpp -- batched HNSW traversal for gfx950: one wave64 per query.
//
// Replaces search() + search_layer() of assembly/hnsw.ts:960-1027, :423-523 and the
// heap routines :141-325 for the query path.  Candidate lists must be BIT-EXACT, so
// both array heaps are emulated element for element in LDS (1-based arrays: the two
// children of p sit at 2p, 2p+1 = one aligned 8-byte LDS read).  What is parallel:
//   * the neighbour list of the popped node is read by 32 lanes in one 128-B line,
//   * visited test-and-set for all neighbours at once (one global atomicOr per lane on
//     this query's private bitmap),
//   * int8 rows are gathered 8 per wave-instruction (8 lanes x 16 B = 128 contiguous
//     bytes per row) and reduced with v_dot4c_i32_i8 + DPP,
//   * heap push = one parallel sift-up (ancestor path is index-determined).
// What stays serial per accepted neighbour: the sift-down of replace_root / pop.
#pragma once

#include <hip/hip_runtime.h>

#include "device_index.hpp"
#include "mvdb_hip.h"

namespace mvdb {

__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float rfl(float v)
{
	return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}

// sum over each aligned group of 8 lanes, result in all 8 lanes (DPP only, no LDS)
__device__ __forceinline__ int reduce8(int x)
{
	x += __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xf, 0xf, false);   // quad_perm [1,0,3,2]
	x += __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xf, 0xf, false);   // quad_perm [2,3,0,1]
	x += __builtin_amdgcn_update_dpp(0, x, 0x141, 0xf, 0xf, false);  // row_half_mirror
	return x;
}

// ---- 1-based array heaps in LDS ------------------------------------------------
// MIN = candidate min-heap (hnsw.ts:153-213), !MIN = result max-heap (:233-315).
template <bool MIN>
struct LdsHeap {
	float *d;  // d[1..size]
	int *id;

	static __device__ __forceinline__ bool stops(float parent, float child)
	{
		return MIN ? parent <= child : parent >= child;  // :163 / :243, :206 / :283
	}
	// push (nid, nd) into a heap of `size`; one parallel sift-up.  Lane j looks at
	// ancestor j+1 of the new slot; ancestors below the stopping one shift down a level.
	__device__ __forceinline__ void push(int size, int nid, float nd, int lane) const
	{
		const int pos = size + 1;
		const int anc = lane < 16 ? (pos >> (lane + 1)) : 0;
		float ad = 0.f;
		int aid = 0;
		if (anc >= 1) {
			ad = d[anc];
			aid = id[anc];
		}
		const bool stop = anc < 1 || stops(ad, nd);
		const int J = __builtin_ctzll(__ballot(stop));
		if (lane < J) {
			d[pos >> lane] = ad;
			id[pos >> lane] = aid;
		}
		if (lane == J) {
			d[pos >> J] = nd;
			id[pos >> J] = nid;
		}
	}
	// place (nid, nd) at the root of a heap of `size` and sift down (:191-210, :268-287)
	__device__ __forceinline__ void sift_root(int size, int nid, float nd, int lane) const
	{
		int p = 1;
		for (;;) {
			const int l = 2 * p;
			if (l > size) break;
			const float2 ch = *reinterpret_cast<const float2 *>(d + l);  // children l, l+1
			float dl = rfl(ch.x), dr = rfl(ch.y);
			int c = l;
			float dc = dl;
			if (l + 1 <= size && (MIN ? dr < dl : dr > dl)) {  // :201 / :278 right only if strictly better
				c = l + 1;
				dc = dr;
			}
			if (stops(nd, dc)) break;
			const int cid = id[c];
			if (lane == 0) {
				d[p] = dc;
				id[p] = cid;
			}
			p = c;
		}
		if (lane == 0) {
			d[p] = nd;
			id[p] = nid;
		}
	}
	// pop the root; returns new size.  (rid, rd) = old root.
	__device__ __forceinline__ int pop(int size, int &rid, float &rd, int lane) const
	{
		rid = rfl(id[1]);
		rd = rfl(d[1]);
		size--;
		if (size > 0) {
			const int lid = rfl(id[size + 1]);
			const float ld = rfl(d[size + 1]);
			sift_root(size, lid, ld, lane);
		}
		return size;
	}
};

// ---- int8 row distance -----------------------------------------------------------
// Query slice held by each lane: 16-byte pieces s, s+8, s+16, ... (s = lane & 7).
template <int J>
struct QueryRegs {
	int4 q[J];
	int qq;  // sum q_i^2 (whole query), same in every lane
};

template <int J>
__device__ __forceinline__ void load_query(QueryRegs<J> &Q, const int8_t *qrow, int dim, int lane)
{
	// query rows are packed dim bytes (dim % 4 == 0) and only 4-byte aligned: read dwords
	const int s = lane & 7;
	int part = 0;
#pragma unroll
	for (int j = 0; j < J; j++) {
		const int byte0 = (j * 8 + s) * 16;
		int w[4];
#pragma unroll
		for (int t = 0; t < 4; t++) {
			const int b = byte0 + 4 * t;
			w[t] = b < dim ? *reinterpret_cast<const int *>(qrow + b) : 0;
			part = __builtin_amdgcn_sdot4(w[t], w[t], part, false);
		}
		Q.q[j] = make_int4(w[0], w[1], w[2], w[3]);
	}
	Q.qq = reduce8(part);
}

// partial (vv - 2 qv) of this lane's pieces of one row
template <int J>
__device__ __forceinline__ int row_partial(const QueryRegs<J> &Q, const int4 (&v)[J])
{
	int qv = 0, vv = 0;
#pragma unroll
	for (int j = 0; j < J; j++) {
		qv = __builtin_amdgcn_sdot4(Q.q[j].x, v[j].x, qv, false);
		qv = __builtin_amdgcn_sdot4(Q.q[j].y, v[j].y, qv, false);
		qv = __builtin_amdgcn_sdot4(Q.q[j].z, v[j].z, qv, false);
		qv = __builtin_amdgcn_sdot4(Q.q[j].w, v[j].w, qv, false);
		vv = __builtin_amdgcn_sdot4(v[j].x, v[j].x, vv, false);
		vv = __builtin_amdgcn_sdot4(v[j].y, v[j].y, vv, false);
		vv = __builtin_amdgcn_sdot4(v[j].z, v[j].z, vv, false);
		vv = __builtin_amdgcn_sdot4(v[j].w, v[j].w, vv, false);
	}
	return vv - 2 * qv;
}

// FULL: row_stride == J*128, every piece exists (384-d, 512-d); otherwise pieces are predicated
template <int J, bool FULL>
__device__ __forceinline__ void load_row(int4 (&v)[J], const DevIndexView &ix, int rid, int s)
{
	const int4 *rp = reinterpret_cast<const int4 *>(ix.vec + (size_t)rid * (size_t)ix.row_stride) + s;
	const int pieces = ix.row_stride >> 4;
#pragma unroll
	for (int j = 0; j < J; j++) {
		if (FULL || (j * 8 + s) < pieces) v[j] = rp[j * 8];
		else v[j] = make_int4(0, 0, 0, 0);
	}
}

// distances of rows ids_s[0..n) (LDS) -> dist_s[0..n) as f32 (math.ts:126 <f32> cast).
// 8 rows per pass, 4 passes of loads in flight.
template <int J, bool FULL>
__device__ __forceinline__ void dist_rows(const DevIndexView &ix, const QueryRegs<J> &Q, const int *ids_s,
                                          float *dist_s, int n, int safe_row, int lane)
{
	const int g = lane >> 3, s = lane & 7;
	for (int p0 = 0; p0 < n; p0 += 32) {
		int4 v[4][J];
		int slot[4];
#pragma unroll
		for (int pp = 0; pp < 4; pp++) {
			slot[pp] = p0 + pp * 8 + g;
			const int rid = slot[pp] < n ? ids_s[slot[pp]] : safe_row;
			if (p0 + pp * 8 < n) load_row<J, FULL>(v[pp], ix, rid, s);
		}
#pragma unroll
		for (int pp = 0; pp < 4; pp++) {
			if (p0 + pp * 8 < n) {
				const int tot = Q.qq + reduce8(row_partial<J>(Q, v[pp]));
				if (s == 0 && slot[pp] < n) dist_s[slot[pp]] = (float)tot;
			}
		}
	}
}

// ---- the kernel --------------------------------------------------------------------
struct TraverseArgs {
	DevIndexView ix;
	const int8_t *q;      // nq x dim packed
	int nq, k, ef, cand_cap, results_cap;
	uint32_t *visited;    // nq x vis_words, zeroed by the launcher
	size_t vis_words;
	int32_t *out_ids;     // nq x k
	float *out_dist;      // nq x k
	int32_t *out_count;   // nq
	mvdb_query_counters *ctr;  // nq or null
};

// LDS carve (dynamic): res_d[ef+2] res_i[ef+2] cand_d[cap+2] cand_i[cap+2] ids_s[64] dist_s[64] code_s[64]
__host__ __device__ inline size_t traverse_lds_bytes(int ef, int cand_cap)
{
	return (size_t)4 * (2 * (size_t)(ef + 2) + 2 * (size_t)(cand_cap + 2) + 3 * 64);
}

template <int J, bool FULL>
__global__ __launch_bounds__(64) void traverse_kernel(TraverseArgs a)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
	const int lane = threadIdx.x;
	const int qi = blockIdx.x;
	if (qi >= a.nq) return;
	const DevIndexView &ix = a.ix;

	const int ef = a.ef, cand_cap = a.cand_cap;
	float *res_d = reinterpret_cast<float *>(lds_raw);
	int *res_i = reinterpret_cast<int *>(res_d + (ef + 2));
	float *cand_d = reinterpret_cast<float *>(res_i + (ef + 2));
	int *cand_i = reinterpret_cast<int *>(cand_d + (cand_cap + 2));
	int *ids_s = cand_i + (cand_cap + 2);
	float *dist_s = reinterpret_cast<float *>(ids_s + 64);
	int *code_s = reinterpret_cast<int *>(dist_s + 64);
	const LdsHeap<true> cand{cand_d, cand_i};
	const LdsHeap<false> res{res_d, res_i};

	QueryRegs<J> Q;
	load_query<J>(Q, a.q + (size_t)qi * (size_t)ix.dim, ix.dim, lane);

	int n_dist = 0, n_l0 = 0, n_up = 0, found = 0;

	// search(): hnsw.ts:961-967 (entry validity is established by the host before launch)
	int cur = ix.entry, cur_code = ix.entry_code;
	float cur_d;
	{
		if (lane == 0) ids_s[0] = cur;
		dist_rows<J, FULL>(ix, Q, ids_s, dist_s, 1, cur, lane);
		cur_d = rfl(dist_s[0]);  // :969
		n_dist++;
	}

	// upper-layer greedy descent, :971-1004
	for (int l = ix.max_level; l > 0; l--) {
		bool changed = true;
		while (changed) {
			changed = false;
			n_up++;
			const int lvl = cur_code & 3, cum = cur_code >> 2;
			if (lvl < l) continue;  // no list on this layer (corrupt graph): treated as empty
			const int2 *lst = ix.upper + (size_t)(cum + l - 1) * (size_t)ix.m;
			for (int base = 0; base < ix.m; base += 64) {
				int2 e = make_int2(-1, -1);
				if (base + lane < ix.m) e = lst[base + lane];
				const unsigned long long vm = __ballot(e.x >= 0);
				const int n = __builtin_popcountll(vm);  // canonical lists: valid entries are a prefix
				if (n == 0) break;
				if (e.x >= 0) {
					ids_s[lane] = e.x;
					code_s[lane] = e.y;
				}
				dist_rows<J, FULL>(ix, Q, ids_s, dist_s, n, cur, lane);
				n_dist += n;
				// sequential scan with strict `<` == first occurrence of the minimum, if below cur_d
				float d = lane < n ? dist_s[lane] : __builtin_inff();
				int idx = lane;
#pragma unroll
				for (int off = 32; off >= 1; off >>= 1) {
					const float od = __shfl_xor(d, off);
					const int oi = __shfl_xor(idx, off);
					if (od < d || (od == d && oi < idx)) {
						d = od;
						idx = oi;
					}
				}
				d = rfl(d);
				idx = rfl(idx);
				if (d < cur_d) {  // :997-1001
					cur_d = d;
					cur = rfl(ids_s[idx]);
					cur_code = rfl(code_s[idx]);
					changed = true;
				}
			}
		}
	}

	// search_layer(query, cur, 0, ef, false): hnsw.ts:423-523
	uint32_t *vis = a.visited + (size_t)qi * a.vis_words;
	int cand_size = 0, res_size = 0;
	{
		// dist0 (:450) re-evaluates the row the descent ended on: same value, counted, not re-read
		const float d0 = cur_d;
		n_dist++;
		if (lane == 0) atomicOr(&vis[(unsigned)cur >> 5], 1u << (cur & 31));  // :452
		cand.push(cand_size++, cur, d0, lane);                                 // :453
		res.push(res_size++, cur, d0, lane);                                   // :454
	}
	float worst = rfl(res_d[1]);

	while (cand_size > 0) {  // :459
		int c_id;
		float c_d;
		cand_size = cand.pop(cand_size, c_id, c_d, lane);  // :460
		if (res_size >= ef && c_d > worst) break;          // :466 (strict >)
		n_l0++;
		const int32_t *lst = ix.adj0 + (size_t)c_id * (size_t)ix.m0;
		for (int base = 0; base < ix.m0; base += 64) {
			int nb = -1;
			if (base + lane < ix.m0) nb = lst[base + lane];
			bool isnew = false;
			if (nb >= 0) {  // :494-495 visited test-and-set
				const uint32_t bit = 1u << (nb & 31);
				const uint32_t old = atomicOr(&vis[(unsigned)nb >> 5], bit);
				isnew = !(old & bit);
			}
			const unsigned long long nm = __ballot(isnew);
			const int n = __builtin_popcountll(nm);
			if (__ballot(nb >= 0) == 0) break;
			if (n == 0) continue;
			if (isnew) ids_s[__builtin_popcountll(nm & ((1ull << lane) - 1))] = nb;
			dist_rows<J, FULL>(ix, Q, ids_s, dist_s, n, c_id, lane);  // :497
			n_dist += n;
			for (int t = 0; t < n; t++) {
				const float d = rfl(dist_s[t]);
				if (res_size < ef || d < worst) {  // :499
					const int nid = rfl(ids_s[t]);
					if (cand_size < cand_cap) cand.push(cand_size++, nid, d, lane);  // :500-502
					if (res_size < ef) res.push(res_size++, nid, d, lane);           // :504-505
					else res.sift_root(res_size, nid, d, lane);                      // :506-507 (d < worst holds)
					worst = rfl(res_d[1]);
				}
			}
		}
	}

	// res_to_sorted_full (:317-325): repeated max-heap pops, written back to front.
	// The candidate heap is dead now; its arrays stage the sorted output.
	found = res_size;
	{
		int n = res_size;
		for (int i = found - 1; i >= 0; i--) {
			int rid;
			float rd;
			n = res.pop(n, rid, rd, lane);
			if (lane == 0) {
				cand_i[i] = rid;
				cand_d[i] = rd;
			}
		}
	}
	int out_count = found < a.k ? found : a.k;  // :1015-1017
	if (out_count > a.results_cap) out_count = a.results_cap;
	if (out_count > MVDB_MAX_EF) out_count = MVDB_MAX_EF;
	for (int i = lane; i < out_count; i += 64) {  // :1019-1024
		a.out_ids[(size_t)qi * (size_t)a.k + i] = cand_i[i];
		a.out_dist[(size_t)qi * (size_t)a.k + i] = cand_d[i];
	}
	for (int i = out_count + lane; i < a.k; i += 64) {  // deterministic tail of a ragged row
		a.out_ids[(size_t)qi * (size_t)a.k + i] = -1;
		a.out_dist[(size_t)qi * (size_t)a.k + i] = 0.f;
	}
	if (lane == 0) {
		a.out_count[qi] = out_count;
		if (a.ctr) {
			mvdb_query_counters c;
			c.n_dist = n_dist;
			c.n_l0lists = n_l0;
			c.n_upperlists = n_up;
			c.n_found = out_count;
			a.ctr[qi] = c;
		}
	}
}

}  // namespace mvdb
