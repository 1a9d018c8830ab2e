// traverse_lean.hip.hpp -- layer-0 search with the RESULT HEAP TAKEN OUT of the traversal: one wave per query.
//
// Same algorithm as traverse.hip.hpp (search() + search_layer(), assembly/hnsw.ts:960-1027, :423-523), same
// candidate lists bit for bit.  What changed is where the reference's two heaps live:
//
//  * Every DECISION search_layer takes from the result heap -- admit iff `res_size < ef || d < worst`
//    (:499), the new `worst` after a replace_root (:506-507), the termination test (:466) -- depends only on
//    the MULTISET of result keys, never on how the heap array arranges them.  The kernel therefore keeps the
//    keys only: the largest ones sorted in registers (KeyWin: up to 64 distinct key values with their
//    multiplicities, one per lane), the rest as a flat unsorted u16 array in LDS (2 bytes per result).  Admitting
//    a neighbour is "drop the largest key, insert the new one": a lane shift, no sift-down.  When the window runs
//    empty it is refilled from the flat array (histogram pass -> threshold -> the keys above it move over).
//  * The ARRAY the reference would hold at the end (which of several equal keys was evicted, the order of the
//    returned list) is reproduced afterwards by replay_kernel: the traversal logs the admitted (key, id) pairs in
//    admission order, the replay runs res_push / res_replace_root / res_to_sorted_full (:233-325) over that
//    log on a real array heap in LDS.  Same operations in the same order as the reference => the same array.
//    It is pure LDS work, runs as its own launch after the traversal and overlaps the next batch's traversal.
//  * The candidate min-heap (:141-213) stays an exact array heap (its pop order and the pushes dropped at
//    cand_cap, :500, decide which nodes are expanded), but as 16-bit keys + 32-bit ids in separate arrays:
//    6 bytes per entry instead of 8.
//  With the 9.6 KB result heap gone and the smaller entries a balanced query (ef 1200, cand_cap 2432) needs
//  ~18 KB of LDS instead of 30 KB and ONE wave instead of three: 8 queries per CU instead of 5, and no mailboxes,
//  no polling, no abort protocol.
//
// One wave does everything, software-pipelined: the candidate-heap work of expansion e (the pushes list e-1
// admitted, then the pop of candidate e) runs in the shadow of expansion e's row gather.  That needs candidate
// e's identity before those pushes have happened, and it is known exactly: the root after the pushes is the
// first-pushed minimum of the admitted neighbours if that is strictly below the root the last pop left (a
// sift-up passes a parent only when strictly smaller, :163), else that root.  The pop re-checks it.
//
// 16-bit keys: distances are exact integers; a query whose walk meets a distance above 65535 (impossible for
// quantised unit vectors: d <= (127+127)^2 = 64516) gives up, leaves its visited bitmap clean and is flagged
// for traverse_kernel, which the launcher runs for flagged queries only.  So does a query whose log overflows.
#pragma once

#include "traverse.hip.hpp"

namespace mvdb {

constexpr int kLeanMaxM0 = 64;            // one 64-lane chunk per neighbour list
constexpr unsigned kLeanKeyMax = 65535u;  // 16-bit keys
constexpr unsigned kLeanInf = 0xffffffffu;
constexpr int kLeanTake = 48;             // a sparse refill moves at most this many entries into the window
constexpr int kLeanMaxIn = 12;            // admissions of one list that may join the window on the parallel path (48 + 12 <= 64 lanes)
constexpr int kLeanLowSlack = 192;        // dead entries the flat key array may hold before it is compacted (sized so that a replay workgroup fits beside eight traversal workgroups in a CU's LDS at ef 1200)

// flag[qi] after traverse_lean_kernel: what still has to happen to the query
enum : int {
	kLeanDone = 0,      // log written: replay_kernel produces the output
	kLeanBigKey = 1,    // a distance above 65535: traverse_kernel takes the query
	kLeanLogFull = 2,   // more admissions than the log holds: traverse_kernel takes the query
	kLeanMismatch = 3,  // the popped root was not the predicted candidate (never observed): traverse_kernel takes the query
	kLeanAborted = 4,   // test hook (mvdb_debug_inject_abort): reported as an aborted query
};

// LLVM's uniformity analysis marks EVERY phi of a join block divergent when a lane-dependent branch joins there,
// also the ones that merge wave-uniform control flow ending in the same block -- and one such phi in the
// expansion loop turns all of its scalar state into vector registers under EXEC masks.  So this kernel has (almost)
// no lane-dependent branch: a masked-off lane stores to a dummy address instead (entry 0 of a heap array, or its own
// dword of a dump area), loads a valid address and discards the value.  Where a branch remains (global atomics),
// an empty volatile asm behind it keeps its join block apart from the enclosing uniform one.
#define MVDB_LANE_JOIN() asm volatile("")
// block placement: the hot path of an expansion should fall through (a taken branch costs a lone wave a refetch)
#define MVDB_RARE(c) __builtin_expect(!!(c), 0)

typedef __attribute__((address_space(3))) unsigned short lds_u16;
typedef __attribute__((address_space(3))) unsigned lds_u32;
__device__ __forceinline__ unsigned lds_r16(unsigned a) { return *reinterpret_cast<const lds_u16 *>((size_t)a); }
__device__ __forceinline__ void lds_w16(unsigned a, unsigned v) { *reinterpret_cast<lds_u16 *>((size_t)a) = (unsigned short)v; }
__device__ __forceinline__ unsigned lds_r32(unsigned a) { return *reinterpret_cast<const lds_u32 *>((size_t)a); }
__device__ __forceinline__ void lds_w32(unsigned a, unsigned v) { *reinterpret_cast<lds_u32 *>((size_t)a) = v; }

// key (u16) and id (u32) of one heap slot in ONE LDS round trip (hipcc otherwise issues the second read only after
// it has waited for the first)
__device__ __forceinline__ void lds_r16_r32(unsigned a16, unsigned a32, unsigned &k, unsigned &i)
{
	asm volatile("ds_read_u16 %0, %2\n\tds_read_b32 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(k), "=&v"(i) : "v"(a16), "v"(a32) : "memory");
}
// two slots (root and last entry of a pop) in one round trip
__device__ __forceinline__ void lds_r16_r32_x2(unsigned a16, unsigned a32, unsigned b16, unsigned b32, unsigned &ka, unsigned &ia,
                                               unsigned &kb_, unsigned &ib_)
{
	asm volatile("ds_read_u16 %0, %4\n\tds_read_b32 %1, %5\n\tds_read_u16 %2, %6\n\tds_read_b32 %3, %7\n\ts_waitcnt lgkmcnt(0)"
	             : "=&v"(ka), "=&v"(ia), "=&v"(kb_), "=&v"(ib_)
	             : "v"(a16), "v"(a32), "v"(b16), "v"(b32)
	             : "memory");
}
__device__ __forceinline__ unsigned lean_readlane(unsigned v, int l) { return (unsigned)__builtin_amdgcn_readlane((int)v, l); }
__device__ __forceinline__ unsigned long long lean_lt_mask(int lane) { return (1ull << lane) - 1; }
// a wave-uniform 64-bit mask the compiler could not prove uniform: pin it to scalar registers (loops over its bits
// are scalar loops then, not waterfalls under EXEC)
__device__ __forceinline__ unsigned long long lean_uni64(unsigned long long x)
{
	const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)x);
	const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(x >> 32));
	return ((unsigned long long)hi << 32) | lo;
}

// minimum / maximum over the wave as a uniform value (DPP within rows of 16, then 4 readlanes)
__device__ __forceinline__ unsigned lean_wave_min(unsigned x)
{
	auto mn = [](unsigned a, unsigned b) { return a < b ? a : b; };
	x = mn(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xf, 0xf, false));
	x = mn(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xf, 0xf, false));
	x = mn(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x141, 0xf, 0xf, false));
	x = mn(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x140, 0xf, 0xf, false));
	return mn(mn(lean_readlane(x, 0), lean_readlane(x, 16)), mn(lean_readlane(x, 32), lean_readlane(x, 48)));
}
__device__ __forceinline__ unsigned lean_wave_max(unsigned x)
{
	auto mx = [](unsigned a, unsigned b) { return a > b ? a : b; };
	x = mx(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xf, 0xf, false));
	x = mx(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xf, 0xf, false));
	x = mx(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x141, 0xf, 0xf, false));
	x = mx(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x140, 0xf, 0xf, false));
	return mx(mx(lean_readlane(x, 0), lean_readlane(x, 16)), mx(lean_readlane(x, 32), lean_readlane(x, 48)));
}
__device__ __forceinline__ unsigned lean_wave_sum(unsigned x)
{
	x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xf, 0xf, false);
	x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xf, 0xf, false);
	x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x141, 0xf, 0xf, false);
	x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x140, 0xf, 0xf, false);
	return lean_readlane(x, 0) + lean_readlane(x, 16) + lean_readlane(x, 32) + lean_readlane(x, 48);
}
// c ? a : b per lane, opaque to the optimiser: LLVM turns a store whose ADDRESS is a select back into a
// lane-dependent branch around two stores -- the very thing the dummy addresses are there to avoid
__device__ __forceinline__ unsigned lean_sel(bool c, unsigned a, unsigned b)
{
	unsigned r;
	const unsigned long long m = __ballot(c);
	asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(m));
	return r;
}
// whole-wave shift by one lane (DPP wave_shr:1)
__device__ __forceinline__ unsigned lean_shr1(unsigned x, unsigned fill)  // lane l <- lane l-1, lane 0 <- fill
{
	return (unsigned)__builtin_amdgcn_update_dpp((int)fill, (int)x, 0x138, 0xf, 0xf, false);
}

// ---- LDS carve: cand keys u16[cap+2] | cand ids u32[cap+2] | flat result keys u16[ef+256] | 5 x 64 dwords of buffers
struct LeanCarve {
	unsigned ck, ci, lowk, buf;
	size_t total;
};
__host__ __device__ inline LeanCarve lean_carve(int ef, int cand_cap)
{
	LeanCarve c;
	unsigned o = 0;
	c.ck = o;
	o += ((unsigned)(cand_cap + 2) * 2u + 15u) & ~15u;
	c.ci = o;
	o += ((unsigned)(cand_cap + 2) * 4u + 15u) & ~15u;
	c.lowk = o;
	o += ((unsigned)(ef + kLeanLowSlack + 64) * 2u + 15u) & ~15u;
	c.buf = o;
	o += 5u * 64u * 4u;  // all_ids | all_dist | buf_a | buf_b | one dword per lane where masked-off stores go
	c.total = o;
	return c;
}
__host__ __device__ inline size_t traverse_lean_lds_bytes(int ef, int cand_cap) { return lean_carve(ef, cand_cap).total; }

// ---- candidate min-heap (hnsw.ts:141-213), 1-based, keys u16[] and ids u32[] apart ---------------------
struct LeanCand {
	unsigned kb, ib;  // LDS byte offsets of key[0], id[0]

	__device__ __forceinline__ unsigned key_at(int p) const { return lds_r16(kb + 2u * (unsigned)p); }
	__device__ __forceinline__ unsigned id_at(int p) const { return lds_r32(ib + 4u * (unsigned)p); }
	// cand_push (:153-171): one parallel sift-up.  Lane j looks at ancestor j+1 of the new slot; ancestors below
	// the stopping one (:163 parent <= child stops) move down a level.  Idle lanes rewrite entry 0 (unused).
	__device__ __forceinline__ void push(int size, unsigned nid, unsigned nkey, int lane) const
	{
		const unsigned pos = (unsigned)size + 1u;
		const unsigned sh = lane < 31 ? (unsigned)lane : 31u;
		const unsigned self = pos >> sh, anc = self >> 1;
		unsigned ek, ei;
		lds_r16_r32(kb + 2u * anc, ib + 4u * anc, ek, ei);
		const bool stop = anc == 0u || ek <= nkey;
		const int Jn = __builtin_ctzll(__ballot(stop));
		const unsigned dst = lean_sel(lane <= Jn, self, 0u);
		lds_w16(kb + 2u * dst, lane == Jn ? nkey : ek);
		lds_w32(ib + 4u * dst, lane == Jn ? nid : ei);
	}
	// the sift-down of cand_pop (:191-210) for (nid, nkey) entering at the root of a heap of `size`, five levels
	// per LDS round trip (LdsHeap::sift_from_wide on the split arrays): lane j holds node j + 2 of the subtree
	// under p in heap numbering, every node decides whether it is the child its parent would pick (:201 the right
	// one only if strictly smaller) and whether the walk goes on through it (:206), the path is followed on the
	// scalar unit through one ballot, and the entries on the path move up together.
	__device__ __forceinline__ void sift_root(int size, unsigned nid, unsigned nkey, int lane) const
	{
		int p = 1;
		const int li = lane + 2;
		const int d = 31 - __builtin_clz(li);  // 1..5 (6 in the two spare lanes)
		const int off = li - (1 << d);
		// the lanes of this node's ancestors inside the 62-node subtree, itself included: the walk passes through the
		// node iff every one of them is a chosen child the walk continues through
		unsigned long long anc = 0;
		for (int x = li; x >= 2; x >>= 1) anc |= 1ull << (x - 2);
		anc = lane < 62 ? anc : ~0ull;
		for (;;) {
			if (2 * p > size) break;  // a leaf
			const int pos = (p << d) + off;
			const bool valid = lane < 62 && pos <= size;
			const unsigned slot = valid ? (unsigned)pos : 0u;
			unsigned ek, ei;
			lds_r16_r32(kb + 2u * slot, ib + 4u * slot, ek, ei);
			ek = valid ? ek : kLeanInf;
			const unsigned ks = (unsigned)__builtin_amdgcn_update_dpp(0, (int)ek, 0xB1, 0xf, 0xf, false);  // sibling
			const bool chosen = (lane & 1) ? ek < ks : ek <= ks;
			const unsigned long long cont = __ballot(chosen && nkey > ek);
			const unsigned long long path = __ballot((cont & anc) == anc);  // at most one node per level
			const unsigned up = lean_sel((path >> lane) & 1ull, (unsigned)(pos >> 1), 0u);  // entry 0 is never read
			lds_w16(kb + 2u * up, ek);
			lds_w32(ib + 4u * up, ei);
			if (path == 0ull) break;  // the walk stops at p
			const int cur = 65 - __builtin_clzll(path);  // the deepest node on the path (numbering grows with depth)
			const int dc = 31 - __builtin_clz(cur);
			p = (p << dc) + (cur - (1 << dc));
			if (dc < 5) break;
		}
		lds_w16(kb + 2u * (unsigned)p, nkey);  // every lane, the same value
		lds_w32(ib + 4u * (unsigned)p, nid);
	}
};

// ---- the largest result keys, sorted in registers ---------------------------------------------------------
// Lanes 0..V-1 hold DISTINCT key values in ascending order with their multiplicities (`total` entries in all).
// Invariant: every key of the result multiset that is >= T sits here; everything below T is in the flat LDS
// array (whose entries >= T are dead: they moved here once and have been dropped since, or still sit here).
struct KeyWin {
	unsigned wk = 0, wc = 0;
	int V = 0, total = 0;
	unsigned T = 0x10000u;
	__device__ __forceinline__ unsigned worst() const { return lean_readlane(wk, V - 1); }
	// drop the m largest entries (m <= total)
	__device__ __forceinline__ void remove_largest(int m, int lane)
	{
		total -= m;
		while (m > 0) {
			const int c = (int)lean_readlane(wc, V - 1);
			if (c > m) {
				wc -= lane == V - 1 ? (unsigned)m : 0u;
				m = 0;
			} else {
				m -= c;
				V -= 1;
			}
		}
	}
};

// distances of rows ids_s[0..n) -> dist_s[0..n) as exact integers; `big` collects "above 65535" per lane
template <int J, bool FULL>
__device__ __forceinline__ void lean_dist_rows(const DevIndexView &ix, const QueryRegs<J> &Q, const int *ids_s, unsigned *dist_s,
                                               int n, int safe_row, int lane, bool &big, unsigned dump)
{
	const int g = lane >> 3, s = lane & 7;
	for (int p0 = 0; p0 < n; p0 += 32) {
		int4 v[4][J];
		int slot[4];
#pragma unroll
		for (int pp = 0; pp < 4; pp++) {
			slot[pp] = p0 + pp * 8 + g;
			int rid = ids_s[slot[pp] < n ? slot[pp] : 0];
			rid = slot[pp] < n ? rid : safe_row;
			if (p0 + pp * 8 < n) load_row<J, FULL>(v[pp], ix, rid, s);
		}
#pragma unroll
		for (int pp = 0; pp < 4; pp++) {
			if (p0 + pp * 8 < n) {
				const unsigned tot = (unsigned)(Q.qq + reduce8(row_partial<J>(Q, v[pp])));
				big = big || (slot[pp] < n && tot > kLeanKeyMax);
				lds_w32(lean_sel(s == 0 && slot[pp] < n, (unsigned)(size_t)dist_s + 4u * (unsigned)slot[pp], dump), tot);
			}
		}
	}
}

// diagnostic stamps (MVDB_TRAVERSE_PROFILE builds only), slots: 0 query + descent, 1 list + issue of the memory
// phase, 2 candidate-heap work in the shadow of the loads, 3 rows + distances, 4 visited test + compaction,
// 5 admission (6 = the refills inside it), 7 log + prediction, 16-18 + 2 = the heap work split into pushes, root check,
// sift-down, new root + list request; 13 = from the compaction to the admission loop (filling phase), 14 parallel decision, 15 its inserts, 19 one-at-a-time iterations, 20 refills; counters: 21 one-at-a-time
// admissions, 22 lists decided in parallel, 8 lists not prefetched, 9 admitted
// neighbours, 10 refills, 11 of them dense, 12 expansions whose next candidate came out of the list itself
#ifdef MVDB_TRAVERSE_PROFILE
#define MVDB_LEAN_COUNT(slot, v) prof_acc[slot] += (unsigned long long)(v)
#else
#define MVDB_LEAN_COUNT(slot, v) \
	do {                         \
	} while (0)
#endif

template <int J, bool FULL>
__global__ __launch_bounds__(64) void traverse_lean_kernel(TraverseArgs a)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
	const int lane = threadIdx.x;
	const int qi = blockIdx.x;
	if (qi >= a.nq) return;
	const DevIndexView &ix = a.ix;
	if (qi == a.inject_abort_q) {  // test hook
		if (lane == 0) a.flag[qi] = kLeanAborted;
		traverse_report_abort(a, qi, lane);
		return;
	}
#ifdef MVDB_TRAVERSE_PROFILE
	unsigned long long prof_acc[24] = {};
	unsigned long long prof_last = __builtin_amdgcn_s_memtime();
#endif

	const int ef = a.ef, cand_cap = a.cand_cap, m0 = ix.m0;
	const int lane_m0 = lane < m0 ? lane : m0 - 1;  // list loads are unconditional (lanes beyond the list re-read its last entry)
	const LeanCarve cv = lean_carve(ef, cand_cap);
	const unsigned lds0 = (unsigned)(size_t)lds_raw;
	const LeanCand cand{lds0 + cv.ck, lds0 + cv.ci};
	const unsigned lowk = lds0 + cv.lowk;
	int *all_ids = reinterpret_cast<int *>(lds_raw + cv.buf);
	unsigned *all_dist = reinterpret_cast<unsigned *>(all_ids + 64);
	int *buf_a = reinterpret_cast<int *>(all_dist + 64);      // compacted ids | descent codes | refill histogram
	unsigned *buf_b = reinterpret_cast<unsigned *>(buf_a + 64);  // compacted distances | refill staging
	const unsigned hist = lds0 + cv.buf + 512u, stage = lds0 + cv.buf + 768u;
	const unsigned dump = lds0 + cv.buf + 1024u + 4u * (unsigned)lane;  // where this lane's masked-off LDS stores go

	QueryRegs<J> Q;
	load_query<J>(Q, a.q + (size_t)qi * (size_t)ix.dim, ix.dim, lane);
	int n_dist = 0, n_l0 = 0, n_up = 0;
	bool big = false;  // some distance does not fit 16 bits
	bool touched = false;  // the visited bitmap holds bits of this query
	int bail = -1;

	// search(): hnsw.ts:961-969 (entry validity is established by the host before launch)
	int cur = ix.entry, cur_code = ix.entry_code;
	unsigned cur_d;
	{
		all_ids[0] = cur;  // every lane, the same value
		lean_dist_rows<J, FULL>(ix, Q, all_ids, all_dist, 1, cur, lane, big, dump);
		cur_d = (unsigned)rfl((int)all_dist[0]);
		n_dist++;
	}
	// upper-layer greedy descent, :971-1004 (integer distances order like the reference's f32 values below 2^24)
	for (int l = ix.max_level; l > 0; l--) {
		bool changed = true;
		while (changed) {
			changed = false;
			n_up++;
			const int lvl = cur_code & 3, cum = cur_code >> 2;
			if (lvl < l) continue;  // no list on this layer (corrupt graph): treated as empty
			const int2 *lst = ix.upper + (size_t)(cum + l - 1) * (size_t)ix.m;
			for (int base = 0; base < ix.m; base += 64) {
				int2 en = lst[base + lane < ix.m ? base + lane : ix.m - 1];
				if (base + lane >= ix.m) en = make_int2(-1, -1);
				const int n = __builtin_popcountll(__ballot(en.x >= 0));  // canonical lists: valid entries are a prefix
				if (n == 0) break;
				all_ids[lane] = en.x;
				buf_a[lane] = en.y;
				lean_dist_rows<J, FULL>(ix, Q, all_ids, all_dist, n, cur, lane, big, dump);
				n_dist += n;
				// sequential scan with strict `<` == first occurrence of the minimum, if below cur_d
				const unsigned my = lane < n ? all_dist[lane] : kLeanInf;
				const unsigned d = lean_wave_min(my);
				if (d < cur_d) {  // :997-1001
					const int idx = __builtin_ctzll(__ballot(my == d));
					cur_d = d;
					cur = rfl(all_ids[idx]);
					cur_code = rfl(buf_a[idx]);
					changed = true;
				}
			}
		}
	}
	if (__ballot(big)) bail = kLeanBigKey;
	MVDB_STAMP(0);

	// ---------------- search_layer(query, cur, 0, ef, false): :423-523 ----------------
	const VisitedSet vis{reinterpret_cast<decltype(VisitedSet::p)>(static_cast<unsigned char *>(a.visited) + (size_t)qi * a.vis_stride)};
	uint2 *log = a.adm_log + (size_t)qi * (size_t)(a.adm_log_cap + 64);
	int cand_size = 0, res_size = 0, low_n = 0, n_adm = 0, s_hint = 0;
	unsigned worst = kLeanInf;  // max of the result keys; consulted only once res_size >= ef
	const int low_cap = ef + kLeanLowSlack;
	KeyWin win;

	// keys < T stay, the dead ones go (the order of the flat array carries no meaning)
	auto low_compact = [&]() __attribute__((always_inline)) {
		int newn = 0;
		for (int i0 = 0; i0 < low_n; i0 += 64) {
			const int i = i0 + lane;
			const bool valid = i < low_n;
			const unsigned k = lds_r16(lowk + 2u * (unsigned)(valid ? i : 0));
			const bool keep = valid && k < win.T;
			const unsigned long long km = __ballot(keep);
			lds_w16(lean_sel(keep, lowk + 2u * (unsigned)(newn + __builtin_popcountll(km & lean_lt_mask(lane))), dump), k);
			newn += __builtin_popcountll(km);
		}
		low_n = newn;
	};
	// the lanes of `mask` append their key to the flat array
	auto low_append = [&](unsigned long long mask, unsigned my_d) __attribute__((always_inline)) {
		const int n = __builtin_popcountll(mask);
		if (low_n + n > low_cap) low_compact();  // <= ef alive entries remain
		lds_w16(lean_sel((mask >> lane) & 1ull, lowk + 2u * (unsigned)(low_n + __builtin_popcountll(mask & lean_lt_mask(lane))), dump), my_d);
		low_n += n;
	};
	// 8 keys of the flat array per lane and step
	auto low_vec = [&](int vi, int nvec) __attribute__((always_inline)) -> uint4 {
		return lds_read16(reinterpret_cast<const void *>((size_t)(lowk + 16u * (unsigned)(vi < nvec ? vi : 0))));  // callers mask by index
	};
	auto low_key = [](const uint4 &kv, int e) __attribute__((always_inline)) -> unsigned {
		const unsigned w = e < 2 ? kv.x : (e < 4 ? kv.y : (e < 6 ? kv.z : kv.w));
		return (e & 1) ? w >> 16 : w & 0xffffu;
	};
	// window empty: move the largest alive keys of the flat array into it.  A histogram of the keys just below the
	// floor T (64 bins of 2^s key values, bin 0 on top) finds the new floor: with bins of one key value the bins ARE
	// the window (any multiplicity); with wider bins the largest prefix of bins holding <= kLeanTake entries is
	// collected, sorted by counting and run-length encoded.  s adapts to the density of the keys.
	auto refill = [&]() __attribute__((always_inline)) {
#ifdef MVDB_TRAVERSE_PROFILE
		const unsigned long long t0_ = __builtin_amdgcn_s_memtime();
#endif
		unsigned T = win.T, Tn = 0, hc = 0;
		int s = s_hint, L = 0, acc = 0, limit = 64;
		const int nvec = (low_n + 7) >> 3;
		for (;;) {
			lds_w32(hist + 4u * (unsigned)lane, 0u);
			for (int v0 = 0; v0 < nvec; v0 += 64) {
				const int vi = v0 + lane;
				const uint4 kv = low_vec(vi, nvec);
#pragma unroll
				for (int e = 0; e < 8; e++) {
					const unsigned k = low_key(kv, e);
					const unsigned b = (T - 1u - k) >> s;  // k >= T wraps far beyond 64 bins
					const bool in = vi < nvec && 8 * vi + e < low_n && k < T && b < 64u;
					__hip_atomic_fetch_add(reinterpret_cast<lds_u32 *>((size_t)lean_sel(in, hist + 4u * b, dump)), lean_sel(in, 1u, 0u), __ATOMIC_RELAXED,
					                       __HIP_MEMORY_SCOPE_WORKGROUP);
				}
			}
			const unsigned long long nz0 = __ballot(lds_r32(hist + 4u * (unsigned)lane) != 0u);
			if (nz0 == 0ull) {  // nothing alive within 64 bins below T: the floor drops past the gap, bins get coarser
				const unsigned span = 64u << s;
				if (T <= span) {  // no alive key at all: the multiset invariant is broken (never observed)
					bail = kLeanMismatch;
					return;
				}
				T -= span;
				s = s + 3 < 10 ? s + 3 : 10;
				continue;
			}
			const int f = __builtin_ctzll(nz0);
			T -= (unsigned)f << s;  // no alive key in [T - (f << s), T): the floor drops to the first occupied bin
			limit = 64 - f;
			hc = lds_r32(hist + 4u * (unsigned)(lane < limit ? lane + f : 0));
			hc = lane < limit ? hc : 0u;
			if (s == 0) {
				Tn = T >= (unsigned)limit ? T - (unsigned)limit : 0u;
				break;
			}
			L = 0;
			acc = 0;
			for (; L < limit; L++) {  // the largest prefix of bins holding <= kLeanTake entries
				const int c = (int)lean_readlane(hc, L);
				if (acc + c > kLeanTake) break;
				acc += c;
			}
			if (L > 0) {
				const unsigned span = (unsigned)L << s;
				Tn = T > span ? T - span : 0u;
				break;
			}
			s = s > 3 ? s - 3 : 0;  // the top bin alone is too crowded: finer bins
		}
		const bool dense = s == 0;
		if (dense) {
			const unsigned long long nz = __ballot(hc != 0u);
			const unsigned long long above = lane == 63 ? 0ull : (nz >> (lane + 1));  // occupied bins of smaller key values
			lds_w32(lean_sel(hc != 0u, stage + 4u * (unsigned)__builtin_popcountll(above), dump), ((T - 1u - (unsigned)lane) & 0xffffu) | (hc << 16));
			const unsigned v = lds_r32(stage + 4u * (unsigned)lane);
			win.wk = v & 0xffffu;
			win.wc = v >> 16;
			win.V = __builtin_popcountll(nz);
			win.total = (int)lean_wave_sum(hc);
			s_hint = win.total >= 24 ? 0 : (win.total >= 12 ? 1 : (win.total >= 6 ? 2 : 3));
		} else {
			// collect the keys in [Tn, T): acc <= kLeanTake of them, in any order
			int cpos = 0;
			for (int v0 = 0; v0 < nvec; v0 += 64) {
				const int vi = v0 + lane;
				const uint4 kv = low_vec(vi, nvec);
#pragma unroll
				for (int e = 0; e < 8; e++) {
					const unsigned k = low_key(kv, e);
					const bool in = vi < nvec && 8 * vi + e < low_n && k < T && k >= Tn;
					const unsigned long long tm = __ballot(in);
					lds_w32(lean_sel(in, stage + 4u * (unsigned)(cpos + __builtin_popcountll(tm & lean_lt_mask(lane))), dump), k);
					cpos += __builtin_popcountll(tm);
				}
			}
			const int cnt = acc;
			unsigned key = lds_r32(stage + 4u * (unsigned)lane);
			key = lane < cnt ? key : kLeanInf;
			unsigned lt = 0, eqb = 0;  // rank by counting
			for (int j = 0; j < cnt; j++) {
				const unsigned kj = lean_readlane(key, j);
				lt += kj < key ? 1u : 0u;
				eqb += (kj == key && j < lane) ? 1u : 0u;
			}
			lds_w32(lean_sel(lane < cnt, hist + 4u * (lt + eqb), dump), key);
			unsigned sk = lds_r32(hist + 4u * (unsigned)lane);
			sk = lane < cnt ? sk : kLeanInf;
			const unsigned prev = lean_shr1(sk, kLeanInf);
			const bool head = lane < cnt && (lane == 0 || sk != prev);
			const unsigned long long hm = __ballot(head);
			const unsigned long long above = lane == 63 ? 0ull : (hm >> (lane + 1));
			const int next = above ? lane + 1 + __builtin_ctzll(above) : cnt;
			lds_w32(lean_sel(head, stage + 4u * (unsigned)__builtin_popcountll(hm & lean_lt_mask(lane)), dump), sk | ((unsigned)(next - lane) << 16));
			const unsigned v = lds_r32(stage + 4u * (unsigned)lane);
			win.wk = v & 0xffffu;
			win.wc = v >> 16;
			win.V = __builtin_popcountll(hm);
			win.total = cnt;
			const int span = L << s;
			s_hint = (L == limit && acc < 24) ? (s < 10 ? s + 1 : 10) : (span <= 80 ? 0 : (L < 16 ? s - 1 : s));
		}
		win.T = Tn;
		MVDB_LEAN_COUNT(10, 1);
		MVDB_LEAN_COUNT(11, dense ? 1 : 0);
#ifdef MVDB_TRAVERSE_PROFILE
		prof_acc[6] += __builtin_amdgcn_s_memtime() - t0_;
#endif
	};
	// d >= win.T joins the window
	auto win_insert = [&](unsigned d) __attribute__((always_inline)) {
		win.total += 1;
		const bool valid = lane < win.V;
		const unsigned long long eq = __ballot(valid && win.wk == d);
		if (eq) {
			win.wc += lane == __builtin_ctzll(eq) ? 1u : 0u;
			return;
		}
		// A free lane always exists: a sparse refill brings at most kLeanTake entries, the number of entries at or
		// above T never grows from list to list (every admission removes one of them) and a list adds at most
		// kLeanMaxIn before its removals; a dense window spans at most 64 key values.
		if (win.V == 64) {
			bail = kLeanMismatch;
			return;
		}
		const int r = __builtin_popcountll(__ballot(valid && win.wk < d));
		const unsigned sk = lean_shr1(win.wk, 0u), sc = lean_shr1(win.wc, 0u);
		win.wk = lane > r ? sk : (lane == r ? d : win.wk);
		win.wc = lane > r ? sc : (lane == r ? 1u : win.wc);
		win.V += 1;
	};
	// Admission of a list in list order (:499-508) on the key multiset; returns the admitted lanes and logs them.
	// Sequentially: neighbour i is admitted iff d_i < worst_i, the largest key left after the admissions before it.
	// The result multiset after i-1 neighbours is the ef smallest keys of (M0 + d_0..d_{i-1}), so
	//     admitted_i  <=>  #{j < i : d_j <= d_i}  <  #{x in M0 : x > d_i}      (M0 = the multiset the list meets).
	// Only neighbours below the current worst can be admitted (B); for one below the floor T the right-hand side is
	// at least the window's size, which the left-hand side cannot reach while |B| <= that size: admitted.  Only the few
	// that land inside the window need the two counts.  The multiset afterwards = M0 + admitted - the |admitted| largest.
	//   (the code sits inline in the expansion loop below: one call site for the refill, nothing left to a closure)

	// the entry point (:450-454): dist0 re-evaluates the row the descent ended on (counted, not re-read)
	n_dist++;
	unsigned long long pend = 0;  // pushes waiting for the candidate heap: lanes of (p_d, p_id)
	unsigned p_d = 0;
	int p_id = -1;
	int c_id = cur;  // the candidate to expand next, known before the heap has seen it
	unsigned c_d = cur_d;
	int seen_listed = 0, seen_new = 0;  // decaying counts: rows with the visited test while at least half were new
	int o_id = -1;
	unsigned o_key = 0;
	bool first = true;  // the entry point goes through the loop once as a list of one new neighbour, without a memory phase
	if (bail < 0) {
		if (lane == 0) vis.or_noret(cur);  // :452
		MVDB_LANE_JOIN();
		touched = true;
	}

	while (rfl(bail) < 0) {
		// every piece of per-query state below is wave-uniform; pinning it to scalar registers at the top of the
		// iteration keeps the loop's control flow on the scalar unit (not EXEC-masked regions over vector copies)
#define MVDB_LEAN_PIN()                             \
	do {                                                \
		bail = rfl(bail);                               \
		cand_size = rfl(cand_size);                     \
		res_size = rfl(res_size);                       \
		low_n = rfl(low_n);                             \
		n_adm = rfl(n_adm);                             \
		s_hint = rfl(s_hint);                           \
		worst = (unsigned)rfl((int)worst);              \
		win.V = rfl(win.V);                             \
		win.total = rfl(win.total);                     \
		win.T = (unsigned)rfl((int)win.T);              \
		c_id = rfl(c_id);                               \
		c_d = (unsigned)rfl((int)c_d);                  \
		o_id = rfl(o_id);                               \
		o_key = (unsigned)rfl((int)o_key);              \
		seen_listed = rfl(seen_listed);                 \
		seen_new = rfl(seen_new);                       \
		n_l0 = rfl(n_l0);                               \
		n_dist = rfl(n_dist);                           \
	} while (0)
		MVDB_LEAN_PIN();
		int n = 0;
		unsigned my_d = kLeanInf;
		int my_id = -1;
		if (MVDB_RARE(first)) {
			n = 1;  // :453-454
			my_d = lane == 0 ? cur_d : kLeanInf;
			my_id = lane == 0 ? cur : -1;
		} else {
			if (res_size >= ef && c_d > worst) break;  // :466 (strict >)
			// (every lane loads, lanes beyond the list re-read its last entry; the mask is applied HERE, where the words
			// are needed -- masking at the load would wait for it on the spot)
			int nb = ix.adj0[(size_t)c_id * (size_t)m0 + lane_m0];
			// ---- candidate-heap work, first half, under the list load: the pushes the previous list admitted (:500-502)
			for (unsigned long long m = lean_uni64(pend); m; m &= m - 1) {
				const int t = __builtin_ctzll(m);
				cand.push(cand_size++, (unsigned)__builtin_amdgcn_readlane(p_id, t), lean_readlane(p_d, t), lane);
			}
			pend = 0;
			MVDB_STAMP(16);
			nb = lane < m0 ? nb : -1;
			const int cnt = __builtin_popcountll(__ballot(nb >= 0));  // canonical lists: valid prefix
			// ---- issue the memory phase: visited test-and-set (:494-495) and, while at least half of the recently
			// listed neighbours were new, the rows of EVERY listed neighbour with it (one round trip instead of two)
			const bool spec = cnt > 0 && 2 * seen_new >= seen_listed;
			const int g = lane >> 3, sl = lane & 7;
			int4 v0[4][J];  // 32 rows per pass, all in flight at once
			int slot0[4];
			// the bits this query set so far must have landed before its visited words are read again
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			const unsigned old = vis.load_word(nb >= 0 ? nb : 0);  // (every lane loads; masked below)
			if (spec) {
				all_ids[lane] = nb;
#pragma unroll
				for (int pp = 0; pp < 4; pp++) {
					slot0[pp] = pp * 8 + g;
					int rid = all_ids[slot0[pp] < cnt ? slot0[pp] : 0];
					rid = slot0[pp] < cnt ? rid : c_id;
					if (pp * 8 < cnt) load_row<J, FULL>(v0[pp], ix, rid, sl);
				}
			}
			MVDB_STAMP(1);
			// ---- candidate-heap work, second half, under the row loads: cand_pop of this candidate (:460)
			unsigned lk, li;
			{
				unsigned rk, ri;  // root and last entry in one round trip
				const unsigned last = (unsigned)(cand_size > 0 ? cand_size : 1);
				lds_r16_r32_x2(cand.kb + 2u, cand.ib + 4u, cand.kb + 2u * last, cand.ib + 4u * last, rk, ri, lk, li);
				rk = (unsigned)rfl((int)rk);
				ri = (unsigned)rfl((int)ri);
				lk = (unsigned)rfl((int)lk);
				li = (unsigned)rfl((int)li);
				if (MVDB_RARE(cand_size <= 0 || (int)ri != c_id || rk != c_d)) {
					bail = kLeanMismatch;
					break;
				}
			}
			cand_size--;
			MVDB_STAMP(17);
			if (cand_size > 0) {
				cand.sift_root(cand_size, li, lk, lane);
				MVDB_STAMP(18);
				{
					unsigned ok_, oi_;
					lds_r16_r32(cand.kb + 2u, cand.ib + 4u, ok_, oi_);
					o_key = (unsigned)rfl((int)ok_);
					o_id = rfl((int)oi_);
				}
			}
			MVDB_STAMP(2);
			// ---- distances of the new neighbours, in list order
			if (spec) {
#pragma unroll
				for (int pp = 0; pp < 4; pp++) {
					if (pp * 8 < cnt) {
						const unsigned tot = (unsigned)(Q.qq + reduce8(row_partial<J>(Q, v0[pp])));
						lds_w32(lean_sel(sl == 0 && slot0[pp] < cnt, (unsigned)(size_t)all_dist + 4u * (unsigned)slot0[pp], dump), tot);
					}
				}
				if (MVDB_RARE(cnt > 32)) {  // lists of up to 64 neighbours: a second pass
					int4 v1[4][J];
					int slot1[4];
#pragma unroll
					for (int pp = 0; pp < 4; pp++) {
						slot1[pp] = 32 + pp * 8 + g;
						int rid = all_ids[slot1[pp] < cnt ? slot1[pp] : 0];
						rid = slot1[pp] < cnt ? rid : c_id;
						if (32 + pp * 8 < cnt) load_row<J, FULL>(v1[pp], ix, rid, sl);
					}
#pragma unroll
					for (int pp = 0; pp < 4; pp++) {
						if (32 + pp * 8 < cnt) {
							const unsigned tot = (unsigned)(Q.qq + reduce8(row_partial<J>(Q, v1[pp])));
							lds_w32(lean_sel(sl == 0 && slot1[pp] < cnt, (unsigned)(size_t)all_dist + 4u * (unsigned)slot1[pp], dump), tot);
						}
					}
				}
				MVDB_STAMP(3);
				const bool isnew = nb >= 0 && !vis.seen(old, nb);
				const unsigned long long nm = __ballot(isnew);
				n = __builtin_popcountll(nm);
				{
					const int pos = __builtin_popcountll(nm & lean_lt_mask(lane));
					const unsigned dd = all_dist[lane];  // :497
					big = big || (isnew && dd > kLeanKeyMax);
					lds_w32(lean_sel(isnew, (unsigned)(size_t)buf_a + 4u * (unsigned)pos, dump), (unsigned)nb);
					lds_w32(lean_sel(isnew, (unsigned)(size_t)buf_b + 4u * (unsigned)pos, dump), dd);
				}
				if (isnew) vis.or_noret(nb);  // :495, off the chain: nothing waits for it before the next list's words are read
				MVDB_LANE_JOIN();
				my_d = buf_b[lane];
				my_d = lane < n ? my_d : kLeanInf;
				my_id = buf_a[lane];
				my_id = lane < n ? my_id : -1;
				MVDB_STAMP(4);
			} else if (cnt > 0) {
				// rows of the NEW neighbours only, after the visited words are back
				const bool isnew = nb >= 0 && !vis.seen(old, nb);
				const unsigned long long nm = __ballot(isnew);
				n = __builtin_popcountll(nm);
				MVDB_STAMP(4);
				if (n > 0) {
					lds_w32(lean_sel(isnew, (unsigned)(size_t)buf_a + 4u * (unsigned)__builtin_popcountll(nm & lean_lt_mask(lane)), dump), (unsigned)nb);
					lean_dist_rows<J, FULL>(ix, Q, buf_a, buf_b, n, c_id, lane, big, dump);  // :497
					if (isnew) vis.or_noret(nb);  // :495
					MVDB_LANE_JOIN();
					my_d = buf_b[lane];
					my_d = lane < n ? my_d : kLeanInf;
					my_id = buf_a[lane];
					my_id = lane < n ? my_id : -1;
				}
				MVDB_STAMP(3);
			}
			if (cnt > 0) {
				seen_listed += cnt - (seen_listed >> 2);
				seen_new += n - (seen_new >> 2);
			}
			if (MVDB_RARE(__ballot(big) != 0ull)) {
				bail = kLeanBigKey;
				break;
			}
			n_l0++;
			n_dist += n;
		}
		first = false;
		n = rfl(n);

		// ---- admission (see above)
		unsigned long long adm = 0;
		if (n > 0) {
			unsigned long long rem = n >= 64 ? ~0ull : ((1ull << n) - 1);
			bool need_refill = false;
			if (MVDB_RARE(res_size < ef)) {  // :504-505 while the result set fills up everything is admitted
				const int f = n < ef - res_size ? n : ef - res_size;
				lds_w16(lean_sel(lane < f, lowk + 2u * (unsigned)(low_n + lane), dump), my_d);
				low_n += f;
				res_size += f;
				adm = f >= 64 ? ~0ull : ((1ull << f) - 1);
				rem &= ~adm;
				if (res_size >= ef) {  // full: the floor starts just above the largest key
					unsigned mx = 0;
					const int nvec = (low_n + 7) >> 3;
					for (int v0i = 0; v0i < nvec; v0i += 64) {
						const int vi = v0i + lane;
						const uint4 kv = low_vec(vi, nvec);
#pragma unroll
						for (int e = 0; e < 8; e++) {
							const unsigned k = low_key(kv, e);
							mx = ((vi < nvec) & (8 * vi + e < low_n) & (k > mx)) ? k : mx;
						}
					}
					win.T = lean_wave_max(mx) + 1u;
					need_refill = true;
				}
			}
			MVDB_STAMP(13);
			for (;;) {
				if (MVDB_RARE(need_refill)) {
					refill();
					if (MVDB_RARE(bail >= 0)) break;
					worst = win.worst();
					need_refill = false;
					MVDB_STAMP(20);
				}
				if (!rem || res_size < ef) break;
				const unsigned long long B = lean_uni64(__ballot(my_d < worst) & rem);  // `worst` only falls: the others stay rejected
				const int nB = __builtin_popcountll(B);
				if (nB == 0) break;
				const unsigned long long inwin = lean_uni64(__ballot(my_d >= win.T) & B);
				if (nB <= win.total && __builtin_popcountll(inwin) <= kLeanMaxIn) {
					unsigned long long ok = B & ~inwin;
					for (unsigned long long mm = inwin; mm; mm &= mm - 1) {
						const int i = __builtin_ctzll(mm);
						const unsigned di = lean_readlane(my_d, i);
						const int before = __builtin_popcountll(__ballot(my_d <= di) & B & lean_lt_mask(i));
						const int le = (int)lean_wave_sum((lane < win.V && win.wk <= di) ? win.wc : 0u);
						if (before < win.total - le) ok |= 1ull << i;
					}
					ok = lean_uni64(ok);
					MVDB_STAMP(14);
					for (unsigned long long mm = ok & inwin; mm; mm &= mm - 1) win_insert(lean_readlane(my_d, __builtin_ctzll(mm)));
					if (MVDB_RARE(bail >= 0)) break;
					if (ok & ~inwin) low_append(ok & ~inwin, my_d);
					MVDB_STAMP(15);
					MVDB_LEAN_COUNT(22, 1);
					win.remove_largest(__builtin_popcountll(ok), lane);  // :506-507 every replace_root drops the largest key
					adm |= ok;
					rem = 0;
					if (MVDB_RARE(win.V == 0)) {
						need_refill = true;
						continue;
					}
					worst = win.worst();
					break;
				}
				// one at a time (a nearly empty window, or many keys landing inside it): the first neighbour of B
				const int t = __builtin_ctzll(B);
				rem &= ~((2ull << t) - 1);
				adm |= 1ull << t;
				const unsigned d = lean_readlane(my_d, t);
				win.remove_largest(1, lane);
				if (d >= win.T) win_insert(d);
				else low_append(1ull << t, my_d);
				MVDB_STAMP(19);
				MVDB_LEAN_COUNT(21, 1);
				if (MVDB_RARE(bail >= 0)) break;
				if (MVDB_RARE(win.V == 0)) {
					need_refill = true;
					continue;
				}
				worst = win.worst();
			}
			if (MVDB_RARE(bail >= 0)) break;
			adm = lean_uni64(adm);
			const int na = __builtin_popcountll(adm);
			MVDB_LEAN_COUNT(9, na);
			if (MVDB_RARE(n_adm + na > a.adm_log_cap)) {
				bail = kLeanLogFull;
				break;
			}
			// (masked-off lanes write behind the log's capacity: the launcher allocates 64 entries more per query)
			log[lean_sel((adm >> lane) & 1ull, (unsigned)(n_adm + __builtin_popcountll(adm & lean_lt_mask(lane))), (unsigned)(a.adm_log_cap + lane))] = make_uint2(my_d, (unsigned)my_id);
			n_adm += na;
		}
		MVDB_STAMP(5);
		// ---- who is popped next.  Of the admitted neighbours the first `room` are pushed (:500); after those pushes
		// the root is the first-pushed minimum if it is strictly below the root the pop left (:163), else that root.
		const int room = cand_cap - cand_size;
		unsigned long long pushed = room > 0 ? adm : 0ull;
		while (__builtin_popcountll(pushed) > room) pushed &= ~(1ull << (63 - __builtin_clzll(pushed)));
		pend = lean_uni64(pushed);
		p_d = my_d;
		p_id = my_id;
		if (pend) {
			const unsigned pd = ((pend >> lane) & 1ull) ? my_d : kLeanInf;
			const unsigned dmin = lean_wave_min(pd);
			if (cand_size == 0 || dmin < o_key) {
				const int idx = __builtin_ctzll(__ballot(pd == dmin));  // first occurrence
				c_d = dmin;
				c_id = __builtin_amdgcn_readlane(my_id, idx);
				MVDB_LEAN_COUNT(12, 1);
			} else {
				c_d = o_key;
				c_id = o_id;
			}
		} else {
			if (cand_size == 0) break;  // :459 the heap is empty
			c_d = o_key;
			c_id = o_id;
		}
		MVDB_STAMP(7);
	}

	if (bail >= 0) {
		// the query goes to traverse_kernel: it must find its visited bitmap clean
		if (touched) {
			uint4 *p = reinterpret_cast<uint4 *>(vis.p);
			const size_t n16 = a.vis_stride / 16;
			for (size_t i = lane; i < n16; i += 64) p[i] = make_uint4(0u, 0u, 0u, 0u);
		}
		if (lane == 0) {
			a.flag[qi] = bail;
			if (bail >= kLeanBigKey && bail <= kLeanMismatch) a.fb_list[atomicAdd(a.fb_n, 1)] = qi;
			if (a.lean_stats) __hip_atomic_fetch_add(a.lean_stats + bail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
		}
		return;
	}
	int out_count = res_size < a.k ? res_size : a.k;  // :1015-1017
	if (out_count > a.results_cap) out_count = a.results_cap;
	if (out_count > MVDB_MAX_EF) out_count = MVDB_MAX_EF;
#ifdef MVDB_TRAVERSE_PROFILE
	if (lane == 0 && a.prof)
		for (int i = 0; i < 24; i++) a.prof[(size_t)qi * kProfSlots + i] = prof_acc[i];
#endif
	if (lane == 0) {
		a.flag[qi] = kLeanDone;
		a.adm_n[qi] = n_adm;
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

// ---- the result heap, afterwards ---------------------------------------------------------------------------
// One wave per query: res_push while the heap fills (:233-248), res_replace_root for every later admission
// (:291-315, pipelined: HeapPipe), then res_to_sorted_full (:317-325) unless the caller only needs the set
// (TraverseArgs::set_only).  Reads the log traverse_lean_kernel wrote; queries it flagged are skipped.
__host__ __device__ inline size_t replay_lds_bytes(int ef, bool sorted)
{
	const size_t heap = (size_t)8 * (size_t)((ef + 3) & ~1);  // (at least the 256 counters of the short cut's radix select)
	return (heap < 1024 ? 1024 : heap) + (sorted ? (size_t)8 * (size_t)((ef + 1) & ~1) : 0);
}

// PERM: the labels of the log are translated into node ids on the way out (TraverseArgs::perm; a template flag, not a
// test per entry: with the test inside, hipcc gives up the four-deep load batches of the loops below)
template <bool PERM>
__global__ __launch_bounds__(64) void replay_kernel(TraverseArgs a, int sorted_lds)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
	const int lane = threadIdx.x;
	const int qi = blockIdx.x;
	if (qi >= a.nq) return;
	if (rfl(a.flag[qi]) != kLeanDone) return;
	const int ef = a.ef;
	uint2 *res_h = reinterpret_cast<uint2 *>(lds_raw);
	uint2 *out_s = res_h + ((ef + 3) & ~1);  // sorted output staging (only when sorted_lds)
	const LdsHeap<false> res{res_h, (unsigned)(size_t)res_h};
	HeapPipe<false> rpipe;
	rpipe.base = (unsigned)(size_t)res_h;
	const int n_adm = rfl(a.adm_n[qi]);
	const uint2 *log = a.adm_log + (size_t)qi * (size_t)(a.adm_log_cap + 64);
	int32_t *oid = a.out_ids + (size_t)qi * (size_t)a.k;
	float *od = a.out_dist + (size_t)qi * (size_t)a.k;
	// ---- short cut when the caller takes the candidates as a SET and none can be cut off (the fused searchMany
	// path).  The heap only ever evicts its maximum, and its maximum never rises once it is full: every evicted key
	// is >= every key that stays, so the keys that stay are the `ef` smallest admitted keys, and with W = the
	// ef-th smallest, every admitted key > W has been evicted and every admitted key < W is still there.  Entries
	// with key == W are the only ones whose fate depends on where the heap had put them; when exactly `found`
	// admitted keys are <= W none of them was evicted, and the set is known without any heap.  W by a two-pass
	// radix select over the 16-bit keys of the log.
	{
		const int found0 = n_adm < ef ? n_adm : ef;
		if (a.set_only && found0 <= a.k && found0 <= a.results_cap && found0 <= MVDB_MAX_EF) {
			unsigned W = kLeanInf;
			int n_le = n_adm;
			if (n_adm > ef) {
				int *bins = reinterpret_cast<int *>(lds_raw);  // 256 counters (the heap's place: not in use yet)
				unsigned prefix = 0;                          // high byte of W once known
				int below = 0;                                // admitted keys below the bin under the lens
				for (int pass = 0; pass < 2; pass++) {
					for (int i = lane; i < 256; i += 64) bins[i] = 0;
					for (int c0 = 0; c0 < n_adm; c0 += 256) {
						unsigned kx[4];
#pragma unroll
						for (int u = 0; u < 4; u++) {
							const int i = c0 + u * 64 + lane;
							kx[u] = i < n_adm ? log[i].x : kLeanInf;
						}
#pragma unroll
						for (int u = 0; u < 4; u++) {
							const bool in = kx[u] <= kLeanKeyMax && (pass == 0 || (kx[u] >> 8) == prefix);
							if (in) atomicAdd(&bins[pass == 0 ? (kx[u] >> 8) : (kx[u] & 255u)], 1);
						}
					}
					// the bin in which the running count reaches ef - below: lane l owns bins 4l .. 4l+3
					const int b0 = bins[4 * lane], b1 = bins[4 * lane + 1], b2 = bins[4 * lane + 2], b3 = bins[4 * lane + 3];
					int incl = b0 + b1 + b2 + b3;
#pragma unroll
					for (int d = 1; d < 64; d <<= 1) {
						const int o = __shfl_up(incl, d, 64);
						incl += lane >= d ? o : 0;
					}
					const int need = ef - below;
					const unsigned long long reach = __ballot(incl >= need);
					const int wl = __builtin_ctzll(reach);  // (some lane reaches it: n_adm > ef)
					const int excl = rfl(__shfl(incl - (b0 + b1 + b2 + b3), wl, 64));
					const int c0_ = rfl(__shfl(b0, wl, 64)), c1_ = rfl(__shfl(b1, wl, 64)), c2_ = rfl(__shfl(b2, wl, 64)),
					          c3_ = rfl(__shfl(b3, wl, 64));
					int run = excl, bin = 4 * wl, cnt_bin = c0_;
					if (run + c0_ < need) {
						run += c0_;
						bin++;
						cnt_bin = c1_;
						if (run + c1_ < need) {
							run += c1_;
							bin++;
							cnt_bin = c2_;
							if (run + c2_ < need) {
								run += c2_;
								bin++;
								cnt_bin = c3_;
							}
						}
					}
					below += run;
					if (pass == 0) prefix = (unsigned)bin;
					else {
						W = (prefix << 8) | (unsigned)bin;
						n_le = below + cnt_bin;
					}
				}
			}
			if (n_le == found0) {
				int pos = 0;
				for (int c0 = 0; c0 < n_adm; c0 += 256) {
					uint2 e[4];
#pragma unroll
					for (int u = 0; u < 4; u++) {
						const int i = c0 + u * 64 + lane;
						e[u] = log[i < n_adm ? i : 0];
					}
					bool keep[4];
					int idv[4];
#pragma unroll
					for (int u = 0; u < 4; u++) keep[u] = c0 + u * 64 + lane < n_adm && e[u].x <= W;
#pragma unroll
					for (int u = 0; u < 4; u++) idv[u] = (int)e[u].y;
					if (PERM) {  // label -> node id, for the entries that stay only, the four gathers in flight together
#pragma unroll
						for (int u = 0; u < 4; u++)
							if (keep[u]) idv[u] = a.perm[e[u].y];
					}
#pragma unroll
					for (int u = 0; u < 4; u++) {
						const unsigned long long km = __ballot(keep[u]);
						if (keep[u]) {
							const int o = pos + __builtin_popcountll(km & lean_lt_mask(lane));
							oid[o] = idv[u];
							od[o] = (float)e[u].x;  // keys as the reference holds them: the f32 value of the integer distance (math.ts:126)
						}
						pos += __builtin_popcountll(km);
					}
				}
				for (int i = found0 + lane; i < a.k; i += 64) {  // deterministic tail of a ragged row
					oid[i] = -1;
					od[i] = 0.f;
				}
				if (lane == 0 && a.lean_stats) __hip_atomic_fetch_add(a.lean_stats + 0, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				return;
			}
			if (lane == 0 && a.lean_stats) __hip_atomic_fetch_add(a.lean_stats + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
	}
	int res_size = 0;
	uint2 cur = lane < n_adm ? log[lane] : make_uint2(0u, 0u);
	for (int c0 = 0; c0 < n_adm; c0 += 64) {
		const uint2 nxt = c0 + 64 + lane < n_adm ? log[c0 + 64 + lane] : make_uint2(0u, 0u);
		const int m = n_adm - c0 < 64 ? n_adm - c0 : 64;
		for (int t = 0; t < m; t++) {
			// keys as the reference holds them: the f32 value of the integer distance (math.ts:126)
			const unsigned key = __float_as_uint((float)lean_readlane(cur.x, t));
			const unsigned id = lean_readlane(cur.y, t);
			if (res_size < ef) {
				res.push(res_size++, (int)id, key, lane);  // :504-505
			} else {
				rpipe.tick();  // :506-507, see traverse_kernel
				rpipe.issue(lane, res_size, key, id);
				rpipe.tick();
			}
		}
		cur = nxt;
	}
	rpipe.drain();
	const int found = res_size;
	const bool unsorted = a.set_only && found <= a.k && found <= a.results_cap && found <= MVDB_MAX_EF;
	int out_count = found < a.k ? found : a.k;  // :1015-1017
	if (out_count > a.results_cap) out_count = a.results_cap;
	if (out_count > MVDB_MAX_EF) out_count = MVDB_MAX_EF;
	if (unsorted) {
		for (int i = lane; i < out_count; i += 64) {
			const uint2 e = res_h[1 + i];
			oid[i] = PERM ? a.perm[e.y] : (int)e.y;
			od[i] = __uint_as_float(e.x);
		}
	} else if (sorted_lds) {
		// res_to_sorted_full (:317-325): repeated max-heap pops, written back to front, pipelined
		for (int i = 0; i < found; i++) {
			const int last = found - i;  // slot of the entry this pop carries to the root
			rpipe.tick();                // the previous pop leaves level 2
			while (rpipe.on_chain_to(last)) rpipe.tick();
			const uint2 root = res_h[1];
			const uint2 e = res_h[last];
			if (lane == 0) out_s[found - 1 - i] = root;
			if (last > 1) {
				rpipe.issue(lane, last - 1, e.x, e.y);
				rpipe.tick();
			}
		}
		rpipe.drain();
		for (int i = lane; i < out_count; i += 64) {  // :1019-1024
			const uint2 e = out_s[i];
			oid[i] = PERM ? a.perm[e.y] : (int)e.y;
			od[i] = __uint_as_float(e.x);
		}
	}
	for (int i = out_count + lane; i < a.k; i += 64) {  // deterministic tail of a ragged row
		oid[i] = -1;
		od[i] = 0.f;
	}
}

}  // namespace mvdb
