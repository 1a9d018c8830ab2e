// build_gpu.hip.hpp -- batched HNSW construction on the GPU (second half).
//
// The reference builds its graph one insert at a time (assembly/hnsw.ts:719-834): greedy descent,
// search_layer in build mode per layer (:423-523, :517-522), select_neighbors_heuristic (:526-574),
// then add_connection in both directions with worst-eviction and reverse-edge removal (:593-678).
// A 32-bit WASM cannot hold more than ~3.5 M nodes and the sequential order makes a 10 M build take
// hours, so the engine needs a builder of its own for the indexes BASELINE.json asks for.  This one
// keeps the reference's per-insert ALGORITHM (same ef_construction, same pool = 2 x cap, same
// heuristic with back-fill, same eviction rule) but inserts a whole BLOCK of new nodes at a time:
//
//   build_link_kernel     one wave per new node: descent + search per layer + heuristic, all against the
//                         graph as it was before the block (nodes of one block do not see each other);
//                         writes the node's own lists and one add request per chosen neighbour
//   build_group_kernel    requests chained per target node (atomicExch on a head word per node)
//   build_apply_kernel    one wave per target: its requests in (layer, source id) order, append or
//                         evict-the-worst exactly as add_connection; victims become removal requests
//   build_group_kernel    removal requests chained per victim
//   build_remove_kernel   one wave per victim: remove_connection (swap with the last entry)
//
// Blocks grow with the graph (block <= nodes already linked / 32), so a new node misses at most ~3 %
// of its potential neighbours; the first nodes are linked sequentially on the host.  The edge set
// is therefore NOT the reference's sequential one (neither is the multi-threaded host builder's);
// the dump is a valid V3 file the oracle loads, and quality is checked by recall / n_dist against
// the sequential build (tests/test_gpu_build.py).  Result heaps are the array heaps of
// traverse.hip.hpp; the visited set of a build search is a Bloom filter in LDS (a false positive
// only skips one node of one search; the query path keeps its exact bitmap).
#pragma once

#include <hip/hip_runtime.h>

#include "traverse.hip.hpp"

namespace mvdb {

constexpr int kBuildPoolMax = 128;  // pool = 2 x cap (hnsw.ts:517-519), cap <= 64
constexpr int kBuildMaxM0 = 64;

struct BuildArgs {
	DevIndexView ix;            // vec / row_stride / dim / m / m0 / max_elements / entry / max_level
	const int8_t *level;        // [N], known for every node up front (RNG stream in id order)
	int32_t *adj0;              // [N][m0], dense prefix, -1 padded
	int32_t *cnt0;              // [N]
	const int64_t *upper_off;   // [N] offset of the node's upper block in `upper`, -1 when level < 1
	int32_t *upper;             // host layout: per node level x (count, m ids)
	int ef, cand_cap;           // ef_construction, min(2 ef + 32, 4096)
	int bloom_log2;             // Bloom filter bits = 1 << bloom_log2
	int first, count;           // this block: ids first .. first + count - 1
	int slots;                  // request slots per new node: m0 + (kMaxLayers - 1) * m
	int4 *req;                  // [count * slots] {target, source, key of dist(target, source), layer}; target -1 = empty
	int4 *rem;                  // [count * slots] {victim, source, 0, layer}; victim -1 = empty
	int32_t *next;              // [count * slots] chain links
	int32_t *head;              // [N] chain heads, -1 when idle
	int32_t *tlist;             // [count * slots] distinct targets of this pass
	int32_t *tcount;            // [1]
	unsigned long long *stats;  // [4] n_dist, n_expanded, n_evictions, n_removed
};

// minimum / maximum over the wave, in every lane's scalar view (DPP within rows of 16, then 4 readlanes)
__device__ __forceinline__ unsigned wave_min_u32(unsigned x)
{
	auto mn = [](unsigned a, unsigned b) { return a < b ? a : b; };
	x = mn(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xf, 0xf, false));
	x = mn(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xf, 0xf, false));
	x = mn(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x141, 0xf, 0xf, false));
	x = mn(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x140, 0xf, 0xf, false));
	const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)x, 0), b = (unsigned)__builtin_amdgcn_readlane((int)x, 16);
	const unsigned c = (unsigned)__builtin_amdgcn_readlane((int)x, 32), d = (unsigned)__builtin_amdgcn_readlane((int)x, 48);
	return mn(mn(a, b), mn(c, d));
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned x)
{
	auto mx = [](unsigned a, unsigned b) { return a > b ? a : b; };
	x = mx(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xf, 0xf, false));
	x = mx(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xf, 0xf, false));
	x = mx(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x141, 0xf, 0xf, false));
	x = mx(x, (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x140, 0xf, 0xf, false));
	const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)x, 0), b = (unsigned)__builtin_amdgcn_readlane((int)x, 16);
	const unsigned c = (unsigned)__builtin_amdgcn_readlane((int)x, 32), d = (unsigned)__builtin_amdgcn_readlane((int)x, 48);
	return mx(mx(a, b), mx(c, d));
}

// list of `node` on `layer`: this lane's entry, -1 beyond the cap or when the node has no such layer
__device__ __forceinline__ int build_list_entry(const BuildArgs &a, int node, int layer, int lane)
{
	if (layer == 0) return lane < a.ix.m0 ? a.adj0[(size_t)node * (size_t)a.ix.m0 + lane] : -1;
	if (a.level[node] < layer) return -1;
	const int32_t *rec = a.upper + a.upper_off[node] + (size_t)(layer - 1) * (size_t)(1 + a.ix.m);
	return lane < a.ix.m ? rec[1 + lane] : -1;
}
__device__ __forceinline__ void build_list_store(const BuildArgs &a, int node, int layer, int lane, int nb, int count)
{
	if (layer == 0) {
		if (lane < a.ix.m0) a.adj0[(size_t)node * (size_t)a.ix.m0 + lane] = nb;
		if (lane == 0) a.cnt0[node] = count;
		return;
	}
	int32_t *rec = a.upper + a.upper_off[node] + (size_t)(layer - 1) * (size_t)(1 + a.ix.m);
	if (lane < a.ix.m) rec[1 + lane] = nb;
	if (lane == 0) rec[0] = count;
}

// LDS of build_link_kernel: res[ef+2] cand[cap+2] | ids[64] dist[64] | pool_id[128] pool_d[128] | sel_id[64] sel_d[64] | bloom
__host__ __device__ inline size_t build_link_lds_bytes(int ef, int cand_cap, int bloom_log2)
{
	return (size_t)8 * (size_t)((ef + 3) & ~1) + (size_t)8 * (size_t)((cand_cap + 3) & ~1) + (size_t)4 * (64 * 2 + 128 * 2 + 64 * 2) +
	       ((size_t)1 << bloom_log2) / 8;
}

template <int J, bool FULL>
__global__ __launch_bounds__(64) void build_link_kernel(BuildArgs a)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
	const int lane = threadIdx.x;
	const int bi = blockIdx.x;
	if (bi >= a.count) return;
	const DevIndexView &ix = a.ix;
	const int id = a.first + bi;
	const int lvl = a.level[id];
	const int ef = a.ef, cand_cap = a.cand_cap;

	uint2 *res_h = reinterpret_cast<uint2 *>(lds_raw);
	uint2 *cand_h = res_h + ((ef + 3) & ~1);
	int *ids_s = reinterpret_cast<int *>(cand_h + ((cand_cap + 3) & ~1));
	unsigned *dist_s = reinterpret_cast<unsigned *>(ids_s + 64);
	int *pool_id = reinterpret_cast<int *>(dist_s + 64);
	unsigned *pool_d = reinterpret_cast<unsigned *>(pool_id + kBuildPoolMax);
	int *sel_id = reinterpret_cast<int *>(pool_d + kBuildPoolMax);
	unsigned *sel_d = reinterpret_cast<unsigned *>(sel_id + 64);
	unsigned *bloom = sel_d + 64;
	const int bloom_words = 1 << (a.bloom_log2 - 5);
	const LdsHeap<true> cand{cand_h, (unsigned)(size_t)cand_h};
	const LdsHeap<false> res{res_h, (unsigned)(size_t)res_h};

	// every request slot of this node starts empty (layers above the node's level stay so)
	for (int i = lane; i < a.slots; i += 64) a.req[(size_t)bi * a.slots + i] = make_int4(-1, 0, 0, 0);

	QueryRegs<J> Q;
	load_query<J>(Q, ix.vec + (size_t)id * (size_t)ix.row_stride, ix.dim, lane);
	unsigned long long n_dist = 0, n_exp = 0;

	// insert :772: distance to the entry point, then the greedy walk through the layers above the node's (:775-810)
	int cur = ix.entry;
	unsigned cur_d;
	{
		if (lane == 0) ids_s[0] = cur;
		dist_rows<J, FULL>(ix, Q, ids_s, dist_s, 1, cur, lane);
		cur_d = (unsigned)rfl((int)dist_s[0]);
		n_dist++;
	}
	for (int l = ix.max_level; l > lvl; l--) {
		bool changed = true;
		while (changed) {
			changed = false;
			const int nb = build_list_entry(a, cur, l, lane);
			const unsigned long long vm = __ballot(nb >= 0);
			const int n = __builtin_popcountll(vm);
			if (n == 0) break;
			if (nb >= 0) ids_s[__builtin_popcountll(vm & ((1ull << lane) - 1))] = nb;
			dist_rows<J, FULL>(ix, Q, ids_s, dist_s, n, cur, lane);
			n_dist += n;
			const unsigned my = lane < n ? dist_s[lane] : 0xffffffffu;
			const unsigned d = wave_min_u32(my);
			if (d < cur_d) {  // strict `<` in list order == first occurrence of the minimum
				const int idx = __builtin_ctzll(__ballot(my == d));
				cur_d = d;
				cur = rfl(ids_s[idx]);
				changed = true;
			}
		}
	}

	const int top = lvl < ix.max_level ? lvl : ix.max_level;
	for (int l = top; l >= 0; l--) {
		const int target = l == 0 ? ix.m0 : ix.m;  // insert :814 `cap`
		// ---------------- search_layer(q, cur, l, ef_construction, buildMode) :423-523
		for (int i = lane; i < bloom_words; i += 64) bloom[i] = 0u;
		auto seen = [&](int nid) -> bool {  // Bloom test-and-set, two hashes
			const unsigned h1 = ((unsigned)nid * 0x9E3779B1u) >> (32 - a.bloom_log2);
			const unsigned h2 = ((unsigned)nid * 0x85EBCA77u + 0xC2B2AE3Du) >> (32 - a.bloom_log2);
			const unsigned o1 = atomicOr(&bloom[h1 >> 5], 1u << (h1 & 31));
			const unsigned o2 = atomicOr(&bloom[h2 >> 5], 1u << (h2 & 31));
			return ((o1 >> (h1 & 31)) & (o2 >> (h2 & 31)) & 1u) != 0u;
		};
		int cand_size = 0, res_size = 0;
		if (lane == 0) (void)seen(cur);
		cand.push(cand_size++, cur, cur_d, lane);
		res.push(res_size++, cur, cur_d, lane);
		n_dist++;
		unsigned worst = cur_d;
		HeapPipe<false> rpipe;
		HeapPipe<true> cpipe;
		rpipe.base = (unsigned)(size_t)res_h;
		cpipe.base = (unsigned)(size_t)cand_h;
		auto background = [&](int n) {
			for (int i = 0; i < n; i++) {
				if (!cpipe.busy()) break;
				cpipe.tick();
			}
		};
		while (cand_size > 0) {
			cpipe.drain();
			const uint2 croot = cand_h[1];
			const int c_id = rfl((int)croot.y);
			const unsigned c_d = (unsigned)rfl((int)croot.x);
			cand_size--;
			if (cand_size > 0) {
				const uint2 last = cand_h[cand_size + 1];
				cpipe.issue(lane, cand_size, last.x, last.y);
			}
			if (res_size >= ef && c_d > worst) break;
			n_exp++;
			const int nb = build_list_entry(a, c_id, l, lane);
			background(6);
			const bool isnew = nb >= 0 && !seen(nb);
			const unsigned long long nm = __ballot(isnew);
			const int n = __builtin_popcountll(nm);
			if (n == 0) continue;
			if (isnew) ids_s[__builtin_popcountll(nm & ((1ull << lane) - 1))] = nb;
			dist_rows<J, FULL>(ix, Q, ids_s, dist_s, n, c_id, lane);
			n_dist += n;
			cpipe.drain();
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
				if (cand_size < cand_cap) cand.push(cand_size++, nid, d, lane);
				if (res_size < ef) {
					res.push(res_size++, nid, d, lane);
					worst = (unsigned)rfl((int)res_h[1].x);
				} else {
					rpipe.tick();
					const int owner = rpipe.issue(lane, res_size, d, (unsigned)nid);
					rpipe.tick();
					worst = (unsigned)__builtin_amdgcn_readlane((int)rpipe.root_key, owner);
				}
			}
		}
		// ---------------- buildMode tail :517-522: the `pool` smallest results, ascending (heap sort, pipelined)
		const int found = res_size;
		rpipe.drain();
		cpipe.drain();
		for (int i = 0; i < found; i++) {
			const int last = found - i;
			rpipe.tick();
			while (rpipe.on_chain_to(last)) rpipe.tick();
			const uint2 root = res_h[1];
			const uint2 en = res_h[last];
			if (lane == 0) cand_h[found - 1 - i] = root;
			if (last > 1) {
				rpipe.issue(lane, last - 1, en.x, en.y);
				rpipe.tick();
			}
		}
		rpipe.drain();
		int pool = 2 * target;
		if (pool > ef) pool = ef;
		if (pool > found) pool = found;
		if (pool > kBuildPoolMax) pool = kBuildPoolMax;
		for (int i = lane; i < pool; i += 64) {
			const uint2 en = cand_h[i];
			pool_id[i] = (int)en.y;
			pool_d[i] = en.x;
		}
		// ---------------- select_neighbors_heuristic :526-574 (accept c unless a chosen s is closer to c than q is)
		int picked = 0;
		unsigned long long used0 = 0, used1 = 0;
		for (int i = 0; i < pool && picked < target; i++) {
			const int c = rfl(pool_id[i]);
			const unsigned d_cq = (unsigned)rfl((int)pool_d[i]);
			bool accept = true;
			if (picked > 0) {
				QueryRegs<J> C;
				load_query<J>(C, ix.vec + (size_t)c * (size_t)ix.row_stride, ix.dim, lane);
				dist_rows<J, FULL>(ix, C, sel_id, dist_s, picked, c, lane);
				n_dist += picked;
				accept = __ballot(lane < picked && dist_s[lane] < d_cq) == 0;
			}
			if (accept) {
				if (lane == 0) {
					sel_id[picked] = c;
					sel_d[picked] = d_cq;
				}
				picked++;
				if (i < 64) used0 |= 1ull << i;
				else used1 |= 1ull << (i - 64);
			}
		}
		for (int i = 0; i < pool && picked < target; i++) {  // back-fill :555-571
			const bool used = i < 64 ? (used0 >> i) & 1ull : (used1 >> (i - 64)) & 1ull;
			if (used) continue;
			if (lane == 0) {
				sel_id[picked] = pool_id[i];
				sel_d[picked] = pool_d[i];
			}
			picked++;
		}
		// ---------------- the node's own list (add_connection(id, nb) into an empty list) + one request per reverse edge
		{
			const int nbv = lane < picked ? sel_id[lane] : -1;
			build_list_store(a, id, l, lane, nbv, picked);
			const int base = l == 0 ? 0 : ix.m0 + (l - 1) * ix.m;
			if (lane < picked)
				a.req[(size_t)bi * a.slots + base + lane] = make_int4(nbv, id, (int)sel_d[lane], l);
		}
		if (pool > 0) {  // insert :826: the closest found node enters the next layer
			cur = rfl(pool_id[0]);
			cur_d = (unsigned)rfl((int)pool_d[0]);
		}
	}
	if (lane == 0 && a.stats) {
		atomicAdd(&a.stats[0], n_dist);
		atomicAdd(&a.stats[1], n_exp);
	}
}

// chain the non-empty slots of `slots4` (x = key node) per node: next[r] = previous head; first arrival lists the node
__global__ void build_group_kernel(const int4 *slots4, int n_slots, int32_t *next, int32_t *head, int32_t *tlist, int32_t *tcount,
                                   int4 *clear_too)
{
	const int r = blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= n_slots) return;
	if (clear_too) clear_too[r] = make_int4(-1, 0, 0, 0);
	const int key = slots4[r].x;
	if (key < 0) return;
	const int old = atomicExch(&head[key], r);
	next[r] = old;
	if (old == -1) tlist[atomicAdd(tcount, 1)] = key;
}

// requests of one node, up to 64 per round, gathered one per lane; returns how many and the chain position to go on from
__device__ __forceinline__ int build_collect(const int32_t *next, int &r, int lane, int &mine)
{
	int n = 0;
	mine = -1;
	while (r >= 0 && n < 64) {
		if (lane == n) mine = r;
		r = next[r];
		n++;
	}
	return n;
}

// rank of every lane's (layer, source) among the n gathered requests (distinct keys)
__device__ __forceinline__ int build_rank(unsigned long long key, int n, int lane)
{
	int rank = 0;
	for (int j = 0; j < n; j++) {
		const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)key, j);
		const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(key >> 32), j);
		const unsigned long long kj = ((unsigned long long)hi << 32) | lo;
		rank += kj < key ? 1 : 0;
	}
	return rank;
}

// add_connection(target, source, layer) for every request of one target (:593-645), sources in ascending id
template <int J, bool FULL>
__global__ __launch_bounds__(64) void build_apply_kernel(BuildArgs a, int n_targets)
{
	__shared__ int ids_s[64];
	__shared__ unsigned dist_s[64];
	const int lane = threadIdx.x;
	if ((int)blockIdx.x >= n_targets) return;
	const DevIndexView &ix = a.ix;
	const int t = a.tlist[blockIdx.x];
	int r = a.head[t];
	unsigned long long n_evict = 0, n_dist = 0;
	bool have_q = false;
	QueryRegs<J> Q;
	while (r >= 0) {
		int mine;
		const int n = build_collect(a.next, r, lane, mine);
		int4 rq = make_int4(-1, 0, 0, 0);
		if (mine >= 0) rq = a.req[mine];
		const unsigned long long key = lane < n ? (((unsigned long long)(unsigned)rq.w << 32) | (unsigned)rq.y) : ~0ull;
		const int rank = build_rank(key, n, lane);
		for (int l = 0; l < kMaxLayers; l++) {
			const unsigned long long lm = __ballot(lane < n && rq.w == l);
			if (lm == 0) continue;
			const int cap = l == 0 ? ix.m0 : ix.m;
			int nb = build_list_entry(a, t, l, lane);
			int count = __builtin_popcountll(__ballot(nb >= 0));  // dense prefix
			unsigned dl = 0;
			unsigned long long have_d = 0;  // lanes whose distance to the target is known
			for (int k = 0; k < n; k++) {
				const unsigned long long who = __ballot(lane < n && rank == k && rq.w == l);
				if (who == 0) continue;
				const int src_lane = __builtin_ctzll(who);
				const int src = __builtin_amdgcn_readlane(rq.y, src_lane);
				const unsigned dnew = (unsigned)__builtin_amdgcn_readlane(rq.z, src_lane);
				const int slot = __builtin_amdgcn_readlane(mine, src_lane);
				if (__ballot(nb == src && lane < count) != 0) continue;  // :604-606 already connected
				if (count < cap) {  // :608-612
					if (lane == count) {
						nb = src;
						dl = dnew;
					}
					have_d |= 1ull << count;
					count++;
					continue;
				}
				const unsigned long long full = cap >= 64 ? ~0ull : ((1ull << cap) - 1);
				if ((have_d & full) != full) {  // distances target -> its current neighbours (:617-631), once
					if (!have_q) {
						load_query<J>(Q, ix.vec + (size_t)t * (size_t)ix.row_stride, ix.dim, lane);
						have_q = true;
					}
					if (lane < cap) ids_s[lane] = nb;
					dist_rows<J, FULL>(ix, Q, ids_s, dist_s, cap, t, lane);
					n_dist += cap;
					if (lane < cap && !((have_d >> lane) & 1ull)) dl = dist_s[lane];
					have_d = full;
				}
				const unsigned wd = wave_max_u32(lane < cap ? dl : 0u);
				if (dnew < wd) {  // :633-643 (`d > worst_d` keeps the FIRST maximum)
					const int wl = __builtin_ctzll(__ballot(lane < cap && dl == wd));
					const int victim = __builtin_amdgcn_readlane(nb, wl);
					if (lane == wl) {
						nb = src;
						dl = dnew;
					}
					n_evict++;
					if (victim != src && victim != t && lane == 0) a.rem[slot] = make_int4(victim, t, 0, l);
				}
			}
			build_list_store(a, t, l, lane, nb, count);
		}
	}
	if (lane == 0) {
		a.head[t] = -1;
		if (a.stats) {
			atomicAdd(&a.stats[0], n_dist);
			atomicAdd(&a.stats[2], n_evict);
		}
	}
}

// remove_connection(victim, source, layer) (:648-678): swap with the last active entry
__global__ __launch_bounds__(64) void build_remove_kernel(BuildArgs a, int n_targets)
{
	const int lane = threadIdx.x;
	if ((int)blockIdx.x >= n_targets) return;
	const int v = a.tlist[blockIdx.x];
	int r = a.head[v];
	unsigned long long n_removed = 0;
	while (r >= 0) {
		int mine;
		const int n = build_collect(a.next, r, lane, mine);
		int4 rq = make_int4(-1, 0, 0, 0);
		if (mine >= 0) rq = a.rem[mine];
		const unsigned long long key = lane < n ? (((unsigned long long)(unsigned)rq.w << 32) | (unsigned)rq.y) : ~0ull;
		const int rank = build_rank(key, n, lane);
		for (int l = 0; l < kMaxLayers; l++) {
			const unsigned long long lm = __ballot(lane < n && rq.w == l);
			if (lm == 0) continue;
			int nb = build_list_entry(a, v, l, lane);
			int count = __builtin_popcountll(__ballot(nb >= 0));
			for (int k = 0; k < n; k++) {
				const unsigned long long who = __ballot(lane < n && rank == k && rq.w == l);
				if (who == 0) continue;
				const int src = __builtin_amdgcn_readlane(rq.y, __builtin_ctzll(who));
				const unsigned long long at = __ballot(nb == src && lane < count);
				if (at == 0) continue;
				const int pos = __builtin_ctzll(at);
				const int lastv = __builtin_amdgcn_readlane(nb, count - 1);
				if (lane == pos) nb = lastv;
				if (lane == count - 1) nb = -1;
				count--;
				n_removed++;
			}
			build_list_store(a, v, l, lane, nb, count);
		}
	}
	if (lane == 0) {
		a.head[v] = -1;
		if (a.stats) atomicAdd(&a.stats[3], n_removed);
	}
}

}  // namespace mvdb
