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

// one ds_read_b128 (hipcc otherwise splits the read and sinks half of it behind the branch,
// which costs a second dependent LDS round trip per heap level)
__device__ __forceinline__ uint4 lds_read16(const void *p)
{
	using u4 = __attribute__((ext_vector_type(4))) unsigned;
	u4 v;
	const unsigned addr = (unsigned)(size_t)p;  // low 32 bits of a shared pointer = LDS byte offset
	asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
	return make_uint4(v.x, v.y, v.z, v.w);
}

// sum over each aligned group of 8 lanes, result in all 8 lanes (DPP only, no LDS)
__device__ __forceinline__ int reduce8(int x)
{
	x += __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xf, 0xf, false);   // quad_perm [1,0,3,2]
	x += __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xf, 0xf, false);   // quad_perm [2,3,0,1]
	x += __builtin_amdgcn_update_dpp(0, x, 0x141, 0xf, 0xf, false);  // row_half_mirror
	return x;
}

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) u32x4_t lds_u32x4;
typedef __attribute__((address_space(3))) u32x2_t lds_u32x2;

// ---- visited set of one query (hnsw.ts:118-132, :494-495) -------------------------------------
// A private bitmap per query in global memory, zeroed by the launcher per batch.  traverse_kernel:
// test-and-set by ONE returning atomicOr per listed neighbour, issued together with its row.  On gfx950 these
// atomics execute at the memory side (TCC_EA0_ATOMIC == TCC_ATOMIC, profiles/r2/pmc_traffic_1m_balanced.json).
// (The reference's own scheme -- a stamp per node, bumped per search -- was measured as one byte per node with an
// epoch per launch: bit-exact but slower, 3.03 vs 2.67 ms per 1k-query launch at 1M x 384; see DESIGN.md.)
struct VisitedSet {
	uint32_t *p;
	__device__ __forceinline__ unsigned probe(int id) const { return atomicOr(&p[(unsigned)id >> 5], 1u << (id & 31)); }
	__device__ __forceinline__ bool seen(unsigned probed, int id) const { return (probed >> (id & 31)) & 1u; }
	__device__ __forceinline__ void set(int) const {}
	// traverse_lean_kernel: test = a load of the word past the vector L1, set = an atomicOr that returns nothing --
	// the atomic (executed at the memory side, up to ~3000 cycles under load) is off the query's critical chain.
	// The caller makes sure its earlier atomics have completed (s_waitcnt vmcnt(0)) before it loads again.
	__device__ __forceinline__ unsigned load_word(int id) const
	{
		return __hip_atomic_load(&p[(unsigned)id >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}
	__device__ __forceinline__ void or_noret(int id) const
	{
		(void)__hip_atomic_fetch_or(&p[(unsigned)id >> 5], 1u << (id & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}
	__device__ __forceinline__ void mark(int id) const  // the entry point (:452)
	{
		atomicOr(&p[(unsigned)id >> 5], 1u << (id & 31));
	}
};
__host__ __device__ inline size_t visited_stride_bytes(int max_elements)
{
	return (((size_t)max_elements + 31) / 32 * 4 + 255) & ~(size_t)255;
}

// ---- 1-based array heaps in LDS ------------------------------------------------
// MIN = candidate min-heap (hnsw.ts:153-213), !MIN = result max-heap (:233-315).
// Entry p = {key, id} (8 bytes); children 2p, 2p+1 are one aligned 16-byte LDS read.
// key = bit pattern of the f32 distance (math.ts:126 <f32> cast).  Distances are >= 0, so
// unsigned integer order of the bit patterns == the reference's f32 order, and every heap
// decision runs on the scalar unit (readfirstlane -> s_cmp -> s_cbranch), not under EXEC masks.
template <bool MIN>
struct LdsHeap {
	uint2 *h;           // h[1..size]
	unsigned lds_base;  // LDS byte offset of h[0]

	static __device__ __forceinline__ bool stops(unsigned parent, unsigned child)
	{
		return MIN ? parent <= child : parent >= child;  // :163 / :243, :206 / :283
	}
	// push (nid, nkey) into a heap of `size`; one parallel sift-up.  Lane j looks at ancestor
	// j+1 of the new slot; ancestors below the stopping one shift down a level.  Branch-free:
	// lanes with nothing to do read and rewrite entry 0, which the 1-based layout never uses.
	__device__ __forceinline__ void push(int size, int nid, unsigned nkey, int lane) const
	{
		const unsigned pos = (unsigned)rfl(size) + 1u;
		const unsigned sh = lane < 31 ? (unsigned)lane : 31u;
		const unsigned self = pos >> sh, anc = self >> 1;
		const u32x2_t e = *reinterpret_cast<const lds_u32x2 *>((size_t)(lds_base + anc * 8u));
		const bool stop = anc == 0u || stops(e.x, nkey);
		const int J = __builtin_ctzll(__ballot(stop));
		const u32x2_t mine = {nkey, (unsigned)nid};
		const u32x2_t w = lane == J ? mine : e;
		*reinterpret_cast<lds_u32x2 *>((size_t)(lds_base + (lane <= J ? self * 8u : 0u))) = w;
	}
	// place (nid, nkey) at the root of a heap of `size` and sift down (:191-210, :268-287)
	__device__ __forceinline__ void sift_root(int size, int nid, unsigned nkey, int lane) const
	{
		sift_from(1, size, nid, nkey, lane);
	}
	// the same walk entered at slot p (a sift-down another wave started and handed over)
	__device__ __forceinline__ void sift_from(int p, int size, int nid, unsigned nkey, int lane) const
	{
		size = rfl(size);  // keep the whole walk on the scalar unit
		nkey = (unsigned)rfl((int)nkey);
		p = rfl(p);
		for (;;) {
			const int l = 2 * p;
			if (l > size) break;
			// {key_l, id_l, key_r, id_r}: entry l is 16-byte aligned (l even, base 16-byte aligned)
			const uint4 ch = lds_read16(h + l);
			const unsigned kl = (unsigned)rfl((int)ch.x), kr = (unsigned)rfl((int)ch.z);
			// :201 / :278 the right child only if strictly better
			const bool right = (l + 1 <= size) && (MIN ? kr < kl : kr > kl);
			const unsigned kc = right ? kr : kl;
			if (stops(nkey, kc)) break;
			const uint2 moved = right ? make_uint2(ch.z, ch.w) : make_uint2(ch.x, ch.y);
			if (lane == 0) h[p] = moved;
			p = l + (right ? 1 : 0);
		}
		if (lane == 0) h[p] = make_uint2(nkey, (unsigned)nid);
	}
	// The same walk, five levels per LDS round trip instead of one: lane j reads node j + 2 of the
	// subtree under p in heap numbering (children 2, 3; grandchildren 4..7; ... 32..63), every node
	// decides locally whether it is the child its parent would pick (:201 the right one only if
	// strictly better) and whether the walk would go on through it (:206), and the path is then
	// followed on the scalar unit through the ballot.  Entries on the path move up together.
	__device__ __forceinline__ void sift_from_wide(int p, int size, int nid, unsigned nkey, int lane) const
	{
		static_assert(MIN, "written for the candidate min-heap (absent nodes read as +inf)");
		size = rfl(size);
		nkey = (unsigned)rfl((int)nkey);
		p = rfl(p);
		const int li = lane + 2;
		const int d = 31 - __builtin_clz(li);  // 1..5 (6 in the two spare lanes)
		const int off = li - (1 << d);
		for (;;) {
			if (2 * p > size) break;  // a leaf
			const int pos = (p << d) + off;
			const bool valid = lane < 62 && pos <= size;
			u32x2_t en = *reinterpret_cast<const lds_u32x2 *>((size_t)(lds_base + (valid ? (unsigned)pos * 8u : 0u)));
			en.x = valid ? en.x : 0xffffffffu;  // never below a key: f32 bit patterns of finite values
			const unsigned ks = (unsigned)__builtin_amdgcn_update_dpp(0, (int)en.x, 0xB1, 0xf, 0xf, false);  // sibling
			const bool chosen = (lane & 1) ? en.x < ks : en.x <= ks;
			const unsigned long long cont = __ballot(chosen && nkey > en.x);
			unsigned long long path = 0;
			int cur = 1, steps = 0;
			for (; steps < 5; steps++) {
				const unsigned pair = (unsigned)(cont >> (2 * cur - 2)) & 3u;  // at most one bit: one child is chosen
				if (!pair) break;
				cur = 2 * cur + (int)(pair >> 1);
				path |= 1ull << (cur - 2);
			}
			if ((path >> lane) & 1ull) *reinterpret_cast<lds_u32x2 *>((size_t)(lds_base + (unsigned)(pos >> 1) * 8u)) = en;
			const int dc = 31 - __builtin_clz(cur);
			p = (p << dc) + (cur - (1 << dc));
			if (steps < 5) break;
		}
		if (lane == 0) h[p] = make_uint2(nkey, (unsigned)nid);
	}
	// pop the root; returns new size.  (rid, rkey) = old root.
	__device__ __forceinline__ int pop(int size, int &rid, unsigned &rkey, int lane) const
	{
		const uint2 root = h[1];
		rkey = (unsigned)rfl((int)root.x);
		rid = rfl((int)root.y);
		size = rfl(size) - 1;
		if (size > 0) {
			const uint2 last = h[size + 1];
			sift_root(size, rfl((int)last.y), (unsigned)rfl((int)last.x), lane);
		}
		return size;
	}
};

// ---- pipelined sift-downs ----------------------------------------------------------
// replace_root (:291-315) and pop (:173-213, :250-289) walk root->leaf one level per step and only
// ever move entries UP along their own path.  Several of them can therefore be in flight at once,
// one per lane, each one level further down per tick, provided a younger operation never reads
// a slot an older one still has to write: the younger one takes a step at level L only when
// every older one is at level >= L+2 (or done).  The resulting array is bit-identical to running
// the operations one after another, at two ticks per operation instead of ~log2(ef) dependent
// LDS round trips, and ticks can be spent while global loads are in flight.  A pop also carries
// the heap's LAST entry to the root; it may be issued only when no operation in flight sits on
// the root->last-slot chain (it could still rewrite that slot).

template <bool MIN>
struct HeapPipe {
	unsigned base = 0;  // LDS byte offset of entry 0 of the heap (16-byte aligned)
	unsigned a = 0;     // this lane's operation: 8 * position, 0 = idle
	unsigned lim = 0;   // 8 * heap size the operation works in
	unsigned key = 0, id = 0;
	unsigned root_key = 0;  // in the lane whose operation just took its root step: the new root entry
	unsigned root_id = 0;
	int next_lane = 0;

	__device__ __forceinline__ bool busy() const { return __ballot(a != 0) != 0; }
	// an operation in flight is still at level 1 or 2 (positions 1..3)
	__device__ __forceinline__ bool near_root() const { return __ballot(a != 0 && a < 32) != 0; }
	__device__ __forceinline__ bool on_chain_to(int slot) const
	{
		const int pos = (int)(a >> 3);
		const int lp = 31 - __builtin_clz(pos > 0 ? pos : 1), ls = 31 - __builtin_clz(slot);
		const bool hit = pos > 0 && lp <= ls && (slot >> (ls - lp)) == pos;
		return __ballot(hit) != 0;
	}
	// returns the lane that owns the new operation
	__device__ __forceinline__ int issue(int lane, int heap_size, unsigned nkey, unsigned nid)
	{
		const int owner = next_lane;
		if (lane == owner) {
			a = 8;
			lim = 8u * (unsigned)heap_size;
			key = nkey;
			id = nid;
		}
		next_lane = (next_lane + 1) & 15;
		return owner;
	}
	// Every operation in flight moves one level (:191-210 / :295-314).  Branch-free: an idle lane
	// (a == 0) reads and rewrites entry 0 of the heap, which the 1-based layout never uses.
	__device__ __forceinline__ void tick()
	{
		const unsigned ca = a << 1;
		const bool has_l = a != 0 && ca <= lim;
		const u32x4_t ch = *reinterpret_cast<const lds_u32x4 *>((size_t)(base + (has_l ? ca : 0u)));
		// :201 / :305 the right child only if strictly better
		const bool right = (ca + 8 <= lim) && (MIN ? ch.z < ch.x : ch.z > ch.x);
		const unsigned kc = right ? ch.z : ch.x, ic = right ? ch.w : ch.y;
		const bool stop = !has_l || (MIN ? key <= kc : key >= kc);  // :206 / :310
		const u32x2_t w = {stop ? key : kc, stop ? id : ic};
		*reinterpret_cast<lds_u32x2 *>((size_t)(base + a)) = w;
		root_key = a == 8 ? w.x : root_key;
		root_id = a == 8 ? w.y : root_id;
		a = stop ? 0u : ca + (right ? 8u : 0u);
	}
	__device__ __forceinline__ void drain()
	{
		while (busy()) tick();
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

// distances of rows ids_s[0..n) (LDS) -> dist_s[0..n): bit patterns of the f32 (math.ts:126 <f32> cast).
// 8 rows per pass, 4 passes of loads in flight.
template <int J, bool FULL>
__device__ __forceinline__ void dist_rows(const DevIndexView &ix, const QueryRegs<J> &Q, const int *ids_s,
                                          unsigned *dist_s, int n, int safe_row, int lane)
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
				if (s == 0 && slot[pp] < n) dist_s[slot[pp]] = __float_as_uint((float)tot);
			}
		}
	}
}

// ---- the kernel --------------------------------------------------------------------
// Output order.  search() returns the result heap sorted by repeated pops (hnsw.ts:317-325, ~13 % of
// the kernel).  The fused query path (mini-vector-db.ts:595-651) consumes the list as a SET whenever
// nothing is cut off: the filter keeps the first annK candidates in ANN order, which is all of
// them when found <= annK, and the final order comes from (l2, id).  With set_only the kernel then
// writes the heap array as it stands (same ids, same distances, heap order) and skips the sort;
// as soon as the list would be truncated (found > k, results cap) it sorts as the reference does.
struct TraverseArgs {
	DevIndexView ix;
	const int8_t *q;      // nq x dim packed
	int nq, k, ef, cand_cap, results_cap;
	int set_only;         // 1: the caller only needs the SET of candidates (see traverse_emit_unsorted)
	void *visited;        // nq x vis_stride bytes, see VisitedSet
	size_t vis_stride;    // bytes per query
	int32_t *out_ids;     // nq x k
	float *out_dist;      // nq x k
	int32_t *out_count;   // nq
	mvdb_query_counters *ctr;  // nq or null
	unsigned long long *prof;  // diagnostic builds only (MVDB_TRAVERSE_PROFILE): nq x 16 cycle sums
	int *err;             // device-visible error word of the handle (host-pinned): bit 0 = a query aborted
	int inject_abort_q;   // test hook (mvdb_debug_inject_abort): this query takes the abort path; -1 = none
	// traverse_lean.hip.hpp: the admitted (key, id) pairs of every query in admission order, for replay_kernel
	uint2 *adm_log;       // nq x adm_log_cap
	int adm_log_cap;
	int *adm_n;           // nq: entries logged
	int *flag;            // nq: kLeanDone, or why the query is left to traverse_kernel
	int *lean_stats;      // device-mapped host counters of the handle, indexed by flag value, or null
	int *fb_list, *fb_n;  // queries traverse_lean_kernel hands to traverse_flagged_kernel (flags 1..3), and their number
	const int32_t *perm;  // ix is the relabelled copy of the index (device_index.hpp): label -> node id for every id written out; nullptr = ids are labels
};

// An aborted query (the test hook) is reported three ways: out_count = -1,
// n_found = -1, and bit 0 of the handle's error word, which every host entry point checks after its
// synchronisation and mvdb_check_errors() exposes to `_dev` callers.
__device__ __forceinline__ void traverse_report_abort(const TraverseArgs &a, int qi, int lane)
{
	if (lane == 0) {
		a.out_count[qi] = -1;
		if (a.ctr) {
			mvdb_query_counters c;
			c.n_dist = 0;
			c.n_l0lists = 0;
			c.n_upperlists = 0;
			c.n_found = -1;
			a.ctr[qi] = c;
		}
		if (a.err) __hip_atomic_fetch_or(a.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
	}
	for (int i = lane; i < a.k; i += 64) {
		a.out_ids[(size_t)qi * (size_t)a.k + i] = -1;
		a.out_dist[(size_t)qi * (size_t)a.k + i] = 0.f;
	}
}

constexpr int kProfSlots = 32;  // u64 cycle sums per query in diagnostic builds
#ifdef MVDB_TRAVERSE_PROFILE
#define MVDB_STAMP(slot)                                        \
	do {                                                        \
		const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
		prof_acc[slot] += now_ - prof_last;                     \
		prof_last = now_;                                       \
	} while (0)
#else
#define MVDB_STAMP(slot) \
	do {                 \
	} while (0)
#endif

// LDS carve (dynamic): res[ef+2] cand[cap+2] (8-byte entries, 16-byte aligned) ids_s[64] dist_s[64] code_s[64]
__host__ __device__ inline size_t traverse_lds_bytes(int ef, int cand_cap)
{
	return (size_t)8 * (size_t)((ef + 3) & ~1) + (size_t)8 * (size_t)((cand_cap + 3) & ~1) + (size_t)4 * 3 * 64;
}

// one query, start to finish, by the calling wave (the workgroup's dynamic LDS at lds_raw)
template <int J, bool FULL>
__device__ __forceinline__ void traverse_query(const TraverseArgs &a, const int qi, unsigned char *lds_raw)
{
	const int lane = threadIdx.x;
	const DevIndexView &ix = a.ix;
	if (qi == a.inject_abort_q) {  // test hook: the one-wave kernel has no abort path of its own
		traverse_report_abort(a, qi, lane);
		return;
	}

	const int ef = a.ef, cand_cap = a.cand_cap;
	uint2 *res_h = reinterpret_cast<uint2 *>(lds_raw);
	uint2 *cand_h = res_h + ((ef + 3) & ~1);
	int *ids_s = reinterpret_cast<int *>(cand_h + ((cand_cap + 3) & ~1));
	unsigned *dist_s = reinterpret_cast<unsigned *>(ids_s + 64);
	int *code_s = reinterpret_cast<int *>(dist_s + 64);
	const LdsHeap<true> cand{cand_h, (unsigned)(size_t)cand_h};
	const LdsHeap<false> res{res_h, (unsigned)(size_t)res_h};

	QueryRegs<J> Q;
	load_query<J>(Q, a.q + (size_t)qi * (size_t)ix.dim, ix.dim, lane);

	int n_dist = 0, n_l0 = 0, n_up = 0, found = 0;
#ifdef MVDB_TRAVERSE_PROFILE
	unsigned long long prof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
	unsigned long long prof_last = __builtin_amdgcn_s_memtime();
#endif

	// search(): hnsw.ts:961-967 (entry validity is established by the host before launch)
	int cur = ix.entry, cur_code = ix.entry_code;
	unsigned cur_d;  // f32 bit pattern, see LdsHeap
	{
		if (lane == 0) ids_s[0] = cur;
		dist_rows<J, FULL>(ix, Q, ids_s, dist_s, 1, cur, lane);
		cur_d = (unsigned)rfl((int)dist_s[0]);  // :969
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
				unsigned d = lane < n ? dist_s[lane] : 0xffffffffu;
				int idx = lane;
#pragma unroll
				for (int off = 32; off >= 1; off >>= 1) {
					const unsigned od = (unsigned)__shfl_xor((int)d, off);
					const int oi = __shfl_xor(idx, off);
					if (od < d || (od == d && oi < idx)) {
						d = od;
						idx = oi;
					}
				}
				d = (unsigned)rfl((int)d);
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

	MVDB_STAMP(0);  // query load + descent
	// search_layer(query, cur, 0, ef, false): hnsw.ts:423-523
	const VisitedSet vis{reinterpret_cast<decltype(VisitedSet::p)>(static_cast<unsigned char *>(a.visited) + (size_t)qi * a.vis_stride)};
	int cand_size = 0, res_size = 0;
	{
		// dist0 (:450) re-evaluates the row the descent ended on: same value, counted, not re-read
		const unsigned d0 = cur_d;
		n_dist++;
		if (lane == 0) vis.mark(cur);  // :452
		cand.push(cand_size++, cur, d0, lane);                                 // :453
		res.push(res_size++, cur, d0, lane);                                   // :454
	}
	unsigned worst = (unsigned)rfl((int)res_h[1].x);
	HeapPipe<false> rpipe;  // result max-heap operations in flight
	HeapPipe<true> cpipe;   // the candidate pop in flight
	rpipe.base = (unsigned)(size_t)res_h;
	cpipe.base = (unsigned)(size_t)cand_h;
	// Spend up to `n` ticks on the candidate pop in flight (called while global loads are
	// outstanding).  Result-heap operations still in flight ride along with the ticks of the next
	// admission loop; only the candidate heap must be quiescent before the next push.
	auto background = [&](int n) {
		for (int i = 0; i < n; i++) {
			if (!cpipe.busy()) break;
			cpipe.tick();
		}
	};

	while (cand_size > 0) {  // :459
		// cand_pop (:173-213): the root leaves now, its sift-down runs in the background
		cpipe.drain();
		const uint2 croot = cand_h[1];
		const int c_id = rfl((int)croot.y);
		const unsigned c_d = (unsigned)rfl((int)croot.x);
		cand_size--;
		if (cand_size > 0) {
			const uint2 last = cand_h[cand_size + 1];
			cpipe.issue(lane, cand_size, last.x, last.y);
		}
		MVDB_STAMP(1);  // candidate pop
		if (res_size >= ef && c_d > worst) break;          // :466 (strict >)
		n_l0++;
		const int32_t *lst = ix.adj0 + (size_t)c_id * (size_t)ix.m0;
		for (int base = 0; base < ix.m0; base += 64) {
			int nb = -1;
			if (base + lane < ix.m0) nb = lst[base + lane];
			background(6);
			bool isnew = false;
			unsigned old = 0;
			if (nb >= 0) old = vis.probe(nb);  // :494-495 visited test-and-set
			background(6);
			isnew = nb >= 0 && !vis.seen(old, nb);
			if (isnew) vis.set(nb);
			const unsigned long long nm = __ballot(isnew);
			const int n = __builtin_popcountll(nm);
			MVDB_STAMP(2);  // neighbour list + visited test-and-set
			if (__ballot(nb >= 0) == 0) break;
			if (n == 0) continue;
			if (isnew) ids_s[__builtin_popcountll(nm & ((1ull << lane) - 1))] = nb;
			dist_rows<J, FULL>(ix, Q, ids_s, dist_s, n, c_id, lane);  // :497
			n_dist += n;
			MVDB_STAMP(3);  // row gather + distances
			cpipe.drain();  // pushes below read ancestor chains of the candidate heap
			// admission in list order (:499-508).  `worst` only falls while a list is processed, so a
			// neighbour rejected once stays rejected: jump straight to the next one below `worst`.
			const unsigned my_d = lane < n ? dist_s[lane] : 0xffffffffu;
			const int my_id = lane < n ? ids_s[lane] : -1;
			unsigned long long rem = n >= 64 ? ~0ull : ((1ull << n) - 1);
			for (;;) {
				const unsigned long long m = (res_size < ef ? ~0ull : __ballot(my_d < worst)) & rem;
				if (m == 0) break;
				const int t = __builtin_ctzll(m);
				rem &= ~((2ull << t) - 1);
				const unsigned d = (unsigned)__builtin_amdgcn_readlane((int)my_d, t);
				const int nid = __builtin_amdgcn_readlane(my_id, t);
#ifdef MVDB_TRAVERSE_PROFILE
				prof_acc[7]++;  // admitted neighbours
#endif
				if (cand_size < cand_cap) cand.push(cand_size++, nid, d, lane);  // :500-502
				if (res_size < ef) {
					// :504-505; nothing is in flight on the result heap before it is full
					res.push(res_size++, nid, d, lane);
					worst = (unsigned)rfl((int)res_h[1].x);
				} else {
					// :506-507 replace_root, pipelined.  The youngest operation in flight has taken its
					// root step (level 2 at most); one tick puts it on level 3, which frees the root.
					rpipe.tick();
					const int owner = rpipe.issue(lane, res_size, d, (unsigned)nid);
					rpipe.tick();  // its root step: the root is final again
					worst = (unsigned)__builtin_amdgcn_readlane((int)rpipe.root_key, owner);
				}
			}
			MVDB_STAMP(4);  // heap updates
		}
	}

	// res_to_sorted_full (:317-325): repeated max-heap pops, written back to front, pipelined.
	// The candidate heap is dead now; its arrays stage the sorted output.
	found = res_size;
	rpipe.drain();
	cpipe.drain();
	const bool unsorted = a.set_only && found <= a.k && found <= a.results_cap && found <= MVDB_MAX_EF;
	if (unsorted)
		for (int i = lane; i < found; i += 64) cand_h[i] = res_h[1 + i];
	for (int i = 0; i < (unsorted ? 0 : found); i++) {
		const int last = found - i;  // slot of the entry this pop carries to the root
		rpipe.tick();                // the previous pop leaves level 2
		while (rpipe.on_chain_to(last)) rpipe.tick();
		const uint2 root = res_h[1];
		const uint2 e = res_h[last];
		if (lane == 0) cand_h[found - 1 - i] = root;
		if (last > 1) {
			rpipe.issue(lane, last - 1, e.x, e.y);
			rpipe.tick();
		}
	}
	rpipe.drain();
	MVDB_STAMP(5);  // final heap sort
	int out_count = found < a.k ? found : a.k;  // :1015-1017
	if (out_count > a.results_cap) out_count = a.results_cap;
	if (out_count > MVDB_MAX_EF) out_count = MVDB_MAX_EF;
	for (int i = lane; i < out_count; i += 64) {  // :1019-1024
		const uint2 e = cand_h[i];
		a.out_ids[(size_t)qi * (size_t)a.k + i] = a.perm ? a.perm[e.y] : (int)e.y;
		a.out_dist[(size_t)qi * (size_t)a.k + i] = __uint_as_float(e.x);
	}
	for (int i = out_count + lane; i < a.k; i += 64) {  // deterministic tail of a ragged row
		a.out_ids[(size_t)qi * (size_t)a.k + i] = -1;
		a.out_dist[(size_t)qi * (size_t)a.k + i] = 0.f;
	}
#ifdef MVDB_TRAVERSE_PROFILE
	MVDB_STAMP(6);  // output
	if (lane == 0 && a.prof)
		for (int i = 0; i < 8; i++) a.prof[(size_t)qi * kProfSlots + i] = prof_acc[i];
#endif
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

template <int J, bool FULL>
__global__ __launch_bounds__(64) void traverse_kernel(TraverseArgs a)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
	const int qi = blockIdx.x;
	if (qi >= a.nq) return;
	traverse_query<J, FULL>(a, qi, lds_raw);
}

// The fall-back launch behind traverse_lean_kernel: a handful of workgroups walk the list of queries the lean kernel
// handed over (fb_list[0 .. *fb_n), usually empty: the launch then costs a few idle workgroups, not one per query
// each asking for this kernel's 30 KB of LDS).
template <int J, bool FULL>
__global__ __launch_bounds__(64) void traverse_flagged_kernel(TraverseArgs a)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
	const int n = rfl(__hip_atomic_load(a.fb_n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
	for (int i = blockIdx.x; i < n; i += gridDim.x) traverse_query<J, FULL>(a, rfl(a.fb_list[i]), lds_raw);
}

}  // namespace mvdb
