// traverse_duo.hip.hpp -- the layer-0 search of traverse.hip.hpp split over TWO or THREE waves per query.
//
// Same algorithm, same bit-exact heaps (hnsw.ts:423-523, :141-325); what changes is who runs what:
//   access wave (wave 0): owns the candidate min-heap and everything that touches HBM -- pops,
//       neighbour list, visited test-and-set, int8 rows, distances -- and replays the pushes
//       the result wave hands it;
//   result wave (wave 1): owns the result max-heap -- admission in list order (:499-508),
//       replace_root sift-downs (pipelined, HeapPipe), the final heap sort.
// One wave per SIMD issues at most one instruction every four cycles, and a query is a serial
// chain; two waves give the chain two issue ports and let the memory phase of expansion t+1 run
// under the admission of expansion t.  That needs the identity of the NEXT candidate before the
// current list has been admitted, and it is known exactly, without speculation:
//   * distances of the whole list are computed before admission starts;
//   * `worst` only falls while a list is processed, so if the FIRST-occurrence minimum d_min of
//     the list is below `worst` at the start it is admitted (fewer than ef entries are <= d_min
//     then, and everything admitted before it is strictly larger);
//   * every push of this list has key >= d_min, and a sift-up passes a parent only when strictly
//     smaller (:163), so after the pushes the candidate root is the d_min neighbour when
//     d_min < key(r), else r, the root the pop's sift-down leaves (final after its first step).
// Cases where this does not hold by construction (result heap filling up inside the list,
// candidate heap about to overflow and drop pushes, :500) take the unpredicted path: the access
// wave waits for the admission, replays the pushes and reads the root.  Every pop re-checks
// that the root IS the candidate whose list was fetched; a mismatch aborts the query (count -1).
//
// The waves talk through LDS only, no barriers: single-writer 16-byte mailboxes, each written
// by lane 0 with one ds_write_b128 and read with one ds_read_b128 (LDS executes whole
// instructions in order, so a snapshot is never torn):
//   record[e & 1] (access -> result): {records published, kind | n | buffer, id, key}
//   channel       (result -> access): {admitted-lane mask (8 B), 2*record + done, worst}
//   sift_done     (access -> result): last pop whose sift-down has left the candidate heap
// Timeline of expansion e:  the access wave publishes record e as soon as the distances are
// there (the result wave may still be busy with e-1); the result wave admits the list into the
// result heap, then -- once sift_done >= e -- replays the admitted neighbours into the
// candidate heap (:500-502; the two heaps are independent, so doing the pushes after the
// admissions leaves the same arrays) and posts the channel; the access wave then pops e+1 and
// starts its memory phase while the result wave is already admitting list e+1... Both waves
// evaluate the termination test (:466) from the same values.  Every wait is bounded; a wave that
// gives up raises `abort` and the query reports count -1.
// THREE waves (WAVES == 3, the default): a push wave (wave 2) takes the candidate heap's writes off
// the other two.  The access wave still pops (it needs the root the pop leaves for the
// prediction) but takes only the root step and hands the rest of the sift-down over
// (sift_req = {pop, 8 * slot, key, id}); the push wave walks it five levels per LDS round trip
// (LdsHeap::sift_from_wide).  The result wave posts the admitted-lane mask after EVERY admission
// (adm[e & 1]; bits only appear above the ones already set), so the candidate pushes of list e
// run while list e is still being admitted; the push wave posts the channel after the last one.
// Measured at 1 M x 384, ef 1200: 3.29 vs 3.71 ms per 1 k-query batch alone, +9 % queries/s with
// three batches in flight.
// Memory phase: the visited test-and-set and the row gather of ALL listed neighbours are issued
// together (about nine in ten listed neighbours are new at ef ~ 1200, the rest is discarded;
// measured 3-4 % faster than rows-after-visited).  The list of the root the pop leaves is fetched
// at pop time (MVDB_DUO_PREF; it is the next candidate unless a new neighbour undercuts it; worth
// 1-1.5 % once the third wave had shortened the other chains).  Touching one dword of every listed
// neighbour's own list so that the next list is an L2 hit (MVDB_DUO_WARM, +25 % HBM traffic) is
// kept behind its macro and off: it never moved the batch time.
#pragma once

#include "traverse.hip.hpp"

namespace mvdb {

#ifndef MVDB_DUO_SPEC_ROWS
#define MVDB_DUO_SPEC_ROWS 1  // gather the rows of every listed neighbour together with the visited test
#endif
#ifndef MVDB_DUO_SPEC_ADAPT
#define MVDB_DUO_SPEC_ADAPT 1  // ... but per query only while at least half of the recently listed neighbours were new (see mem_phase)
#endif
#ifndef MVDB_DUO_WARM
#define MVDB_DUO_WARM 0  // touch the listed neighbours' own lists (L2 warm-up for the next pop): +25 % HBM traffic, no gain measured
#endif
#ifndef MVDB_DUO_PREF
#define MVDB_DUO_PREF 1  // fetch the list of the root the pop leaves at pop time: +1-1.5 % with three waves (none with two)
#endif
#ifndef MVDB_DUO_SPEC_LIST
#define MVDB_DUO_SPEC_LIST 0  // guess the next candidate BEFORE the wait for the previous admission and fetch its list under that wait: bit-exact, no gain (2.69 vs 2.64 ms solo, 404 k both with three batches in flight) -- the admission chain, not the list fetch, sets the period
#endif
#ifndef MVDB_DUO_POLL_SLEEP
#define MVDB_DUO_POLL_SLEEP 1  // s_sleep argument (x64 cycles) between two polls of a mailbox
#endif
#ifndef MVDB_DUO_RESULT_PRIO
#define MVDB_DUO_RESULT_PRIO 3  // s_setprio of the result wave (its admissions are the longest chain): +3.4 %
#endif
#ifndef MVDB_DUO_ACCESS_PRIO
#define MVDB_DUO_ACCESS_PRIO 2  // +1.7 % on top (the push wave at 1 or 2: nothing / worse)
#endif
#ifndef MVDB_DUO_PUSH_PRIO
#define MVDB_DUO_PUSH_PRIO 0
#endif
#ifndef MVDB_DUO_POSTPOP_TICKS
#define MVDB_DUO_POSTPOP_TICKS 0  // measured at 1M x 384, ef 1200: 0 -> 3.87 ms, 4 -> 3.98, 16 -> 4.16
#endif
constexpr int kDuoThreads = 128;
constexpr int kDuoMaxM0 = 64;       // one 64-lane chunk per neighbour list
constexpr int kDuoSpinLimit = 1 << 20;

enum : int { kRecInit = 0, kRecList = 1, kRecEnd = 2 };

struct DuoSync {
	u32x4_t record[2];  // access wave -> result wave, by record parity
	u32x4_t channel;    // result wave (two waves) / push wave (three waves) -> access wave
	u32x4_t adm[2];     // three waves only: result wave -> push wave, by record parity
	u32x4_t sift_req;   // three waves only: access wave -> push wave, {pop e, 8 * slot to go on from, key, id}
	int sift_done;      // access wave -> result wave
	int fin;            // access wave: counters below are final
	int abort;
	int n_dist, n_l0, n_up;  // handed over with the END record
	int sink;                // keeps the L2-warming loads alive, never read
};

// minimum over the wave, in every lane's scalar view (DPP within rows of 16, then 4 readlanes)
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

__device__ __forceinline__ u32x4_t duo_snapshot(const u32x4_t *p)
{
	const u32x4_t v = *reinterpret_cast<const volatile lds_u32x4 *>((size_t)(unsigned)(size_t)p);
	return u32x4_t{(unsigned)rfl((int)v.x), (unsigned)rfl((int)v.y), (unsigned)rfl((int)v.z), (unsigned)rfl((int)v.w)};
}
// publish: everything this wave wrote to LDS before is visible to whoever sees the new mailbox
__device__ __forceinline__ void duo_publish(u32x4_t *p, u32x4_t v, int lane)
{
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
	if (lane == 0) *reinterpret_cast<volatile lds_u32x4 *>((size_t)(unsigned)(size_t)p) = v;
}

// LDS carve: res[ef+2] cand[cap+2] | ids[2][64] dist[2][64] all_ids[64] all_dist[64] code[64] | DuoSync
__host__ __device__ inline size_t traverse_duo_lds_bytes(int ef, int cand_cap)
{
	return (size_t)8 * (size_t)((ef + 3) & ~1) + (size_t)8 * (size_t)((cand_cap + 3) & ~1) + (size_t)4 * 7 * 64 +
	       sizeof(DuoSync);
}

// diagnostic cycle stamps (MVDB_TRAVERSE_PROFILE builds only); slots, access wave: 0 descent,
// 1 early memory phase, 2 wait for admission + replay, 3 late (unpredicted) memory phase,
// 4 pop + publish + prediction, 8 unpredicted expansions; result wave: 5 wait for a record,
// 6 admission, 7 final sort, 9 admitted neighbours
#ifdef MVDB_TRAVERSE_PROFILE
#define MVDB_DUO_STAMP(slot)                                          \
	do {                                                              \
		const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
		prof_acc[slot] += now_ - prof_last;                           \
		prof_last = now_;                                             \
	} while (0)
#else
#define MVDB_DUO_STAMP(slot) \
	do {                     \
	} while (0)
#endif

template <int J, bool FULL, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void traverse_duo_kernel(TraverseArgs a)
{
	static_assert(WAVES == 2 || WAVES == 3, "two or three waves per query");
	constexpr bool TRIO = WAVES == 3;
#ifdef MVDB_TRAVERSE_PROFILE
	unsigned long long prof_acc[kProfSlots] = {};
	unsigned long long prof_last = __builtin_amdgcn_s_memtime();
	unsigned long long prof_unsure = 0, prof_admitted = 0;
#endif
	extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
	const int lane = threadIdx.x & 63;
	const int wave = rfl((int)(threadIdx.x >> 6));
	const int qi = blockIdx.x;
	if (qi >= a.nq) return;
	const DevIndexView &ix = a.ix;
#ifdef MVDB_TRAVERSE_PROFILE
	// where the hardware put this wave: HW_ID bits [5:4] = SIMD, [11:8] = CU, [15:13] = SE (slots 20..22)
	if (lane == 0 && a.prof) a.prof[(size_t)qi * kProfSlots + 20 + wave] = (unsigned long long)__builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4);
#endif

	const int ef = a.ef, cand_cap = a.cand_cap;
	uint2 *res_h = reinterpret_cast<uint2 *>(lds_raw);
	uint2 *cand_h = res_h + ((ef + 3) & ~1);
	int *ids_b = reinterpret_cast<int *>(cand_h + ((cand_cap + 3) & ~1));  // [2][64]
	unsigned *dist_b = reinterpret_cast<unsigned *>(ids_b + 128);          // [2][64]
	int *all_ids = reinterpret_cast<int *>(dist_b + 128);
	unsigned *all_dist = reinterpret_cast<unsigned *>(all_ids + 64);
	int *code_s = reinterpret_cast<int *>(all_dist + 64);
	DuoSync *sy = reinterpret_cast<DuoSync *>(code_s + 64);
	const LdsHeap<true> cand{cand_h, (unsigned)(size_t)cand_h};
	const LdsHeap<false> res{res_h, (unsigned)(size_t)res_h};

	// each wave clears the mailbox it WRITES; nobody reads before the barrier
	if (wave == 0) {
		if (lane == 0) {
			sy->record[0] = u32x4_t{0, 0, 0, 0};
			sy->record[1] = u32x4_t{0, 0, 0, 0};
			sy->sift_done = 0;
			sy->sift_req = u32x4_t{0, 0, 0, 0};
			sy->fin = 0;
			sy->abort = 0;
		}
	} else if (wave == 1) {
		if (lane == 0) {
			sy->adm[0] = u32x4_t{0, 0, 0, 0};
			sy->adm[1] = u32x4_t{0, 0, 0, 0};
			if (!TRIO) sy->channel = u32x4_t{0, 0, 0, 0};
		}
	} else {
		if (lane == 0) sy->channel = u32x4_t{0, 0, 0, 0};
	}
	__syncthreads();  // the only barrier: every wave reaches it unconditionally

	if (wave == 0) {
		// =========================== access wave ===========================
		if (MVDB_DUO_ACCESS_PRIO) __builtin_amdgcn_s_setprio(MVDB_DUO_ACCESS_PRIO);
		QueryRegs<J> Q;
		load_query<J>(Q, a.q + (size_t)qi * (size_t)ix.dim, ix.dim, lane);
		int n_dist = 0, n_l0 = 0, n_up = 0;

		// search(): hnsw.ts:961-969
		int cur = ix.entry, cur_code = ix.entry_code;
		unsigned cur_d;
		{
			if (lane == 0) all_ids[0] = cur;
			dist_rows<J, FULL>(ix, Q, all_ids, all_dist, 1, cur, lane);
			cur_d = (unsigned)rfl((int)all_dist[0]);
			n_dist++;
		}
		// upper-layer greedy descent, :971-1004 (as in traverse_kernel)
		for (int l = ix.max_level; l > 0; l--) {
			bool changed = true;
			while (changed) {
				changed = false;
				n_up++;
				const int lvl = cur_code & 3, cum = cur_code >> 2;
				if (lvl < l) continue;
				const int2 *lst = ix.upper + (size_t)(cum + l - 1) * (size_t)ix.m;
				for (int base = 0; base < ix.m; base += 64) {
					int2 en = make_int2(-1, -1);
					if (base + lane < ix.m) en = lst[base + lane];
					const unsigned long long vm = __ballot(en.x >= 0);
					const int n = __builtin_popcountll(vm);
					if (n == 0) break;
					if (en.x >= 0) {
						all_ids[lane] = en.x;
						code_s[lane] = en.y;
					}
					dist_rows<J, FULL>(ix, Q, all_ids, all_dist, n, cur, lane);
					n_dist += n;
					// sequential scan with strict `<` == first occurrence of the minimum, if below cur_d
					const unsigned my = lane < n ? all_dist[lane] : 0xffffffffu;
					const unsigned d = wave_min_u32(my);
					if (d < cur_d) {  // :997-1001
						const int idx = __builtin_ctzll(__ballot(my == d));
						cur_d = d;
						cur = rfl(all_ids[idx]);
						cur_code = rfl(code_s[idx]);
						changed = true;
					}
				}
			}
		}

		MVDB_DUO_STAMP(0);
		// search_layer(query, cur, 0, ef, false): :450-454
		const VisitedSet vis{reinterpret_cast<decltype(VisitedSet::p)>(static_cast<unsigned char *>(a.visited) + (size_t)qi * a.vis_stride)};
		VisitedLog vlog;
		vlog.init(a.vis_log, a.vis_log_cap, qi);
		int cand_size = 0;
		n_dist++;  // dist0 (:450) re-evaluates the row the descent ended on
		if (lane == 0) vis.mark(cur);
		vlog.append_one(cur, lane);
		cand.push(cand_size++, cur, cur_d, lane);
		duo_publish(&sy->record[0], u32x4_t{1u, (unsigned)kRecInit, (unsigned)cur, cur_d}, lane);

		HeapPipe<true> cpipe;
		cpipe.base = (unsigned)(size_t)cand_h;
		bool dead = qi == a.inject_abort_q;  // test hook: the abort protocol runs as after a failed wait
		int res_size = 0;  // tracked from the admitted masks
		unsigned worst = 0;
		int sift_e = 0;        // last pop issued
		bool sift_pub = true;  // sift_done == sift_e has been posted
		auto post_sift = [&]() {
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
			if (lane == 0) *reinterpret_cast<volatile int *>(&sy->sift_done) = sift_e;
			sift_pub = true;
		};
		// work that fits under outstanding global loads: the pop's sift-down
		auto background = [&](int iters) {
			for (int i = 0; i < iters; i++) {
				if (cpipe.busy()) {
					cpipe.tick();
					continue;
				}
				if (!sift_pub) post_sift();
				break;
			}
		};

		int pref_id = -1, pref_nb = -1;  // list fetched ahead for the root the last pop left
		int spec_id = -1, spec_nb = -1;  // list fetched ahead for the guess made before the channel wait
		// Rows of every listed neighbour together with the visited test save a memory round trip when nearly all of
		// them are new (i.i.d. data at ef ~ 1200: 95 %), and waste row traffic and dot products when few are (clustered
		// data: 4 of 32).  Both orders give the same buffers, so the choice is free: a decaying count of listed and new
		// neighbours of this query's last few lists picks rows-with-visited while at least half were new.
		int seen_listed = 0, seen_new = 0;
		auto note_list = [&](int listed, int fresh) {
			seen_listed += listed - (seen_listed >> 2);
			seen_new += fresh - (seen_new >> 2);
		};
		int sink = 0;                    // keeps the L2-warming loads alive
		// neighbour list -> {visited test-and-set, rows of every listed neighbour} -> new ones into buffer b
		auto mem_phase = [&](int c_id, int b) -> int {
			int nb = -1;
			if (pref_id == c_id) nb = pref_nb;
			else if (MVDB_DUO_SPEC_LIST && spec_id == c_id) nb = spec_nb;
			else {
				if (lane < ix.m0) nb = ix.adj0[(size_t)c_id * (size_t)ix.m0 + lane];
				background(6);
			}
			const int cnt = __builtin_popcountll(__ballot(nb >= 0));  // canonical lists: valid prefix
			MVDB_DUO_STAMP(10);
			if (cnt == 0) return 0;
			unsigned old = 0;
			if (nb >= 0) {  // :494-495
				old = vis.probe(nb);
				if (MVDB_DUO_SPEC_ROWS) all_ids[lane] = nb;
			}
			if (!MVDB_DUO_SPEC_ROWS || (MVDB_DUO_SPEC_ADAPT && 2 * seen_new < seen_listed)) {
				// rows of the NEW neighbours only, after the visited words are back
				background(6);
				const bool isnew = nb >= 0 && !vis.seen(old, nb);
				if (isnew) vis.set(nb);
				const unsigned long long nm = __ballot(isnew);
				const int n = __builtin_popcountll(nm);
				vlog.append(isnew, nb, nm, n, lane);
				note_list(cnt, n);
				if (n == 0) return 0;
				int *ids = ids_b + 64 * b;
				if (isnew) ids[__builtin_popcountll(nm & ((1ull << lane) - 1))] = nb;
				dist_rows<J, FULL>(ix, Q, ids, dist_b + 64 * b, n, c_id, lane);  // :497
				return n;
			}
			const int g = lane >> 3, sl = lane & 7;
			int warm = 0;
			// 32 rows per pass, all in flight at once; no loop around the passes (a loop makes the
			// compiler drain vmcnt before every reuse of the row registers)
			auto pass = [&](const int p0, const bool first) {
				int4 v[4][J];
				int slot[4];
#pragma unroll
				for (int pp = 0; pp < 4; pp++) {
					slot[pp] = p0 + pp * 8 + g;
					const int rid = slot[pp] < cnt ? all_ids[slot[pp]] : c_id;
					if (p0 + pp * 8 < cnt) load_row<J, FULL>(v[pp], ix, rid, sl);
				}
				if (first) {
					// first dword of every listed neighbour's own list: the next pop's list becomes an L2 hit
					if (MVDB_DUO_WARM && nb >= 0) warm = ix.adj0[(size_t)nb * (size_t)ix.m0];
					MVDB_DUO_STAMP(11);
					background(12);
					MVDB_DUO_STAMP(12);
				}
#pragma unroll
				for (int pp = 0; pp < 4; pp++) {
					if (p0 + pp * 8 < cnt) {
						const int tot = Q.qq + reduce8(row_partial<J>(Q, v[pp]));
						if (sl == 0 && slot[pp] < cnt) all_dist[slot[pp]] = __float_as_uint((float)tot);
					}
				}
			};
			pass(0, true);
			if (cnt > 32) pass(32, false);
			MVDB_DUO_STAMP(13);
			sink ^= warm;
			const bool isnew = nb >= 0 && !vis.seen(old, nb);
			if (isnew) vis.set(nb);
			const unsigned long long nm = __ballot(isnew);
			const int n = __builtin_popcountll(nm);
			vlog.append(isnew, nb, nm, n, lane);
			note_list(cnt, n);
			if (isnew) {
				const int pos = __builtin_popcountll(nm & ((1ull << lane) - 1));
				ids_b[64 * b + pos] = nb;
				dist_b[64 * b + pos] = all_dist[lane];  // :497
			}
			MVDB_DUO_STAMP(14);
			return n;
		};

		int e = 1;  // record to publish next (expansion e-1)
		int c_id = cur;
		unsigned c_d = cur_d;
		bool have_c = true, published = false;
		for (; !dead;) {
			int n = 0;
			published = false;
			// what the prediction below needs from this list alone; formed right after the record is
			// out, i.e. while the other waves still work on the previous list
			unsigned my_d = 0xffffffffu, d_min = 0xffffffffu;
			int min_idx = 0, min_id = -1;
			auto list_minimum = [&]() {
				my_d = lane < n ? dist_b[64 * (e & 1) + lane] : 0xffffffffu;
				d_min = wave_min_u32(my_d);
				min_idx = __builtin_ctzll(__ballot(my_d == d_min));  // first occurrence
				min_id = rfl(ids_b[64 * (e & 1) + min_idx]);
			};
			MVDB_DUO_STAMP(4);
			if (have_c) {
				n = mem_phase(c_id, e & 1);
				duo_publish(&sy->record[e & 1],
				            u32x4_t{(unsigned)(e + 1), (unsigned)kRecList | ((unsigned)n << 8) | ((unsigned)(e & 1) << 16),
				                    (unsigned)c_id, c_d},
				            lane);
				published = true;
				list_minimum();
			}
			MVDB_DUO_STAMP(1);
			// the result wave pushes only after this pop's sift-down has left the heap
			cpipe.drain();
			if (!sift_pub) post_sift();
			if (MVDB_DUO_SPEC_LIST) {
				// A HINT, never a decision: the pop below cannot happen before list e-1 has been admitted and pushed, and
				// that wait is the longest idle stretch of this wave.  Guess who is popped next from what is visible now
				// -- the minimum of this list against the current root's children (read while the other waves may still
				// be pushing: any entry ever seen there is a whole, valid {key, id}) -- and fetch that node's list under
				// the wait.  Lists never change during a launch, so the fetched words are used iff the exact choice made
				// after the pop names the same node; otherwise they are dropped.
				int g_id = -1;
				if (!have_c) {
					if (cand_size >= 1) g_id = rfl((int)cand_h[1].y);
				} else {
					const u32x4_t kids = *reinterpret_cast<const volatile lds_u32x4 *>((size_t)((unsigned)(size_t)cand_h + 16u));
					const unsigned kl = (unsigned)rfl((int)kids.x), kr = (unsigned)rfl((int)kids.z);
					const int il = rfl((int)kids.y), ir = rfl((int)kids.w);
					unsigned k_key = 0xffffffffu;
					int k_id = -1;
					if (cand_size >= 2) {
						const bool right = cand_size >= 3 && kr < kl;
						k_key = right ? kr : kl;
						k_id = right ? ir : il;
					}
					const bool list_wins = n > 0 && (res_size < ef || d_min < worst) && d_min < k_key;
					g_id = list_wins ? min_id : k_id;
				}
				if (g_id != c_id && g_id != pref_id && (unsigned)g_id < (unsigned)ix.max_elements) {
					spec_id = g_id;
					spec_nb = -1;
					if (lane < ix.m0) spec_nb = ix.adj0[(size_t)g_id * (size_t)ix.m0 + lane];
				}
			}
			// ---- record e-1 must be admitted and pushed before the pop
			u32x4_t ch;
			for (int spin = 0;;) {
				ch = duo_snapshot(&sy->channel);
				if ((int)ch.z == 2 * (e - 1) + 1) break;
				if (++spin > kDuoSpinLimit || rfl(*reinterpret_cast<volatile int *>(&sy->abort))) {
					dead = true;
					break;
				}
				__builtin_amdgcn_s_sleep(MVDB_DUO_POLL_SLEEP);
			}
			MVDB_DUO_STAMP(2);
			if (dead) break;
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
			worst = ch.w;
			{
				const int adm = __builtin_popcountll((unsigned long long)ch.x | ((unsigned long long)ch.y << 32));
				if (e == 1) res_size = 1;
				else {
					res_size = res_size + adm < ef ? res_size + adm : ef;
					cand_size = cand_size + adm < cand_cap ? cand_size + adm : cand_cap;  // :500 drops the rest
				}
			}
			if (cand_size == 0) {  // :459
				if (have_c) dead = true;  // a predicted candidate implies a non-empty heap
				break;
			}
			const uint2 root = cand_h[1];
			// three waves: everything the pop reads (root, last entry, the root's children) in ONE round trip
			u32x2_t pop_last = {0u, 0u};
			u32x4_t pop_kids = {0u, 0u, 0u, 0u};
			if (TRIO) {
				pop_last = *reinterpret_cast<const lds_u32x2 *>((size_t)((unsigned)(size_t)cand_h + 8u * (unsigned)cand_size));
				pop_kids = *reinterpret_cast<const lds_u32x4 *>((size_t)((unsigned)(size_t)cand_h + 16u));
			}
			const int r_id = rfl((int)root.y);
			const unsigned r_key = (unsigned)rfl((int)root.x);
			if (!have_c) {
				c_id = r_id;
				c_d = r_key;
				n = mem_phase(c_id, e & 1);
				duo_publish(&sy->record[e & 1],
				            u32x4_t{(unsigned)(e + 1), (unsigned)kRecList | ((unsigned)n << 8) | ((unsigned)(e & 1) << 16),
				                    (unsigned)c_id, c_d},
				            lane);
				published = true;
				list_minimum();
				MVDB_DUO_STAMP(3);
			} else if (r_id != c_id || r_key != c_d) {
				dead = true;  // the exactness argument in the header failed: never observed
				break;
			}
			if (res_size >= ef && c_d > worst) break;  // :466 (strict >); the result wave stops on the same test
			// cand_pop (:173-213): its sift-down runs in the background of the next memory phase
			cand_size--;
			int o_id = -1;
			unsigned o_key = 0;
			if (cand_size > 0) {
				if (TRIO) {
					// root step of the sift-down (:191-210; HeapPipe::tick at slot 1) on the scalar unit; the
					// push wave walks the rest (it pushes right after it anyway)
					const unsigned lk = (unsigned)rfl((int)pop_last.x), li = (unsigned)rfl((int)pop_last.y);
					const unsigned kl = (unsigned)rfl((int)pop_kids.x), il = (unsigned)rfl((int)pop_kids.y);
					const unsigned kr = (unsigned)rfl((int)pop_kids.z), ir = (unsigned)rfl((int)pop_kids.w);
					const bool right = cand_size >= 3 && kr < kl;  // :201 the right child only if strictly better
					const unsigned kc = right ? kr : kl, ic = right ? ir : il;
					const bool stop = cand_size < 2 || lk <= kc;  // :206
					o_key = stop ? lk : kc;
					o_id = (int)(stop ? li : ic);
					if (lane == 0) cand_h[1] = make_uint2(o_key, (unsigned)o_id);
					const unsigned rest = stop ? 0u : 8u * (2u + (right ? 1u : 0u));
					duo_publish(&sy->sift_req, u32x4_t{(unsigned)e, rest, lk, li}, lane);
				} else {
					const uint2 last = cand_h[cand_size + 1];
					const int owner = cpipe.issue(lane, cand_size, last.x, last.y);
					cpipe.tick();  // root step: the root is final
					o_key = (unsigned)__builtin_amdgcn_readlane((int)cpipe.root_key, owner);
					o_id = __builtin_amdgcn_readlane((int)cpipe.root_id, owner);
				}
				if (MVDB_DUO_PREF && !(MVDB_DUO_SPEC_LIST && spec_id == o_id) && pref_id != o_id) {
					pref_id = o_id;  // its list is needed next unless a new neighbour undercuts it
					pref_nb = -1;
					if (lane < ix.m0) pref_nb = ix.adj0[(size_t)o_id * (size_t)ix.m0 + lane];
				}
			}
			if (TRIO && cand_size == 0) duo_publish(&sy->sift_req, u32x4_t{(unsigned)e, 0u, 0u, 0u}, lane);
			sift_e = e;
			sift_pub = TRIO;  // two waves: sift_done is posted once the pipe has drained
			background(MVDB_DUO_POSTPOP_TICKS);  // part of the sift-down under the latency of the list fetch
			n_l0++;
			n_dist += n;
			// ---- who is popped next (see header)
			have_c = false;
			const bool full = res_size >= ef;
			const int free_slots = cand_cap - cand_size;  // pushes beyond these are dropped (:500)
			if (full || res_size + n <= ef) {
				// neighbours that can be admitted at all: below `worst` as it is now (it only falls)
				const unsigned long long elig = __ballot(lane < n && (!full || my_d < worst));
				const unsigned d = d_min;
				if (elig == 0 || free_slots <= 0) {
					// nothing is pushed: the root the pop leaves is next (an empty heap ends the search)
					if (cand_size > 0) {
						c_id = o_id;
						c_d = o_key;
						have_c = true;
					}
				} else if (cand_size > 0 && o_key <= d) {
					c_id = o_id;  // no push can pass the root (:163)
					c_d = o_key;
					have_c = true;
				} else {
					const int idx = min_idx;
					if (__builtin_popcountll(elig & ((1ull << idx) - 1)) < free_slots) {
						c_id = min_id;  // d_min is admitted and still finds a slot
						c_d = d;
						have_c = true;
					} else if (free_slots == 1) {
						// exactly one push: the first eligible neighbour (admitted: `worst` has not moved yet)
						const int f = __builtin_ctzll(elig);
						const unsigned fd = (unsigned)rfl((int)dist_b[64 * (e & 1) + f]);
						if (cand_size == 0 || fd < o_key) {
							c_id = rfl(ids_b[64 * (e & 1) + f]);
							c_d = fd;
						} else {
							c_id = o_id;
							c_d = o_key;
						}
						have_c = true;
					}
				}
			}
#ifdef MVDB_TRAVERSE_PROFILE
			if (!have_c) prof_unsure++;
#endif
			e++;
		}
#ifdef MVDB_TRAVERSE_PROFILE
		MVDB_DUO_STAMP(4);
		if (lane == 0 && a.prof) {
			for (int i = 0; i < 5; i++) a.prof[(size_t)qi * kProfSlots + i] = prof_acc[i];
			for (int i = 10; i < 15; i++) a.prof[(size_t)qi * kProfSlots + i] = prof_acc[i];
			a.prof[(size_t)qi * kProfSlots + 8] = prof_unsure;
		}
#endif
		cpipe.drain();
		if (!sift_pub) post_sift();
		if (lane == 0) {
			if (dead) sy->abort = 1;
			sy->n_dist = n_dist;
			sy->n_l0 = n_l0;
			sy->n_up = n_up;
			sy->sink = sink;
		}
		if (!published) duo_publish(&sy->record[e & 1], u32x4_t{(unsigned)(e + 1), (unsigned)kRecEnd, 0u, 0u}, lane);
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
		if (lane == 0) *reinterpret_cast<volatile int *>(&sy->fin) = 1;
		// this wave owns the visited set: leave it zero (behind the other waves' output, which `fin` released)
		vlog.clear(vis.p, a.vis_stride, dead, lane);
		return;
	}

	if (TRIO && wave == 2) {
		// =========================== push wave (three waves per query) ===========================
		// Finishes the sift-down of pop e (the access wave takes the root step and hands the rest over)
		// and replays the admitted neighbours of each list into the candidate heap (:500-502) WHILE the
		// result wave is still admitting that list: the result wave posts the admitted-lane mask
		// after every admission (adm[e & 1] = {mask, ((e + 1) << 8) | kind << 1 | done, worst}); bits
		// only ever appear above the ones already set, so pushing the new bits of each snapshot in
		// ascending order is the list order.  The channel to the access wave is posted from here.
		if (MVDB_DUO_PUSH_PRIO) __builtin_amdgcn_s_setprio(MVDB_DUO_PUSH_PRIO);
		int cand_size = 1;  // the access wave pushed the entry point (:453)
		bool dead = false;
		for (int e = 0; !dead; e++) {
			unsigned long long pushed = 0;
			bool popped = false, loaded = false;
			unsigned my_d = 0;
			int my_id = -1, kind = kRecList;
			u32x4_t msg;
			for (int spin = 0;;) {
				msg = duo_snapshot(&sy->adm[e & 1]);
				if ((int)(msg.z >> 8) == e + 1) {
					__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
					MVDB_DUO_STAMP(16);
					kind = (int)((msg.z >> 1) & 3u);
					const unsigned long long mask = (unsigned long long)msg.x | ((unsigned long long)msg.y << 32);
					unsigned long long fresh = mask & ~pushed;
					if (kind == kRecList && !popped) {
						// pop e: the access wave took the root step, the rest of the sift-down is walked here
						u32x4_t rq;
						for (int sp2 = 0;;) {
							rq = duo_snapshot(&sy->sift_req);
							if ((int)rq.x == e) break;
							if (++sp2 > kDuoSpinLimit || rfl(*reinterpret_cast<volatile int *>(&sy->abort))) {
								dead = true;
								break;
							}
							__builtin_amdgcn_s_sleep(MVDB_DUO_POLL_SLEEP);
						}
						if (dead) break;
						__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
						cand_size--;
						popped = true;
						MVDB_DUO_STAMP(17);
						if (rq.y != 0u) cand.sift_from_wide((int)(rq.y >> 3), cand_size, (int)rq.w, rq.z, lane);
						MVDB_DUO_STAMP(18);
					}
					if (fresh) {
						if (!loaded) {
							my_d = dist_b[64 * (e & 1) + lane];
							my_id = ids_b[64 * (e & 1) + lane];
							loaded = true;
						}
						for (; fresh && cand_size < cand_cap; fresh &= fresh - 1) {
							const int t = __builtin_ctzll(fresh);
							cand.push(cand_size++, __builtin_amdgcn_readlane(my_id, t),
							          (unsigned)__builtin_amdgcn_readlane((int)my_d, t), lane);
						}
						pushed = mask;
						spin = 0;
						MVDB_DUO_STAMP(19);
					}
					if (msg.z & 1u) break;  // the record is complete
					if (mask != pushed) continue;
				}
				if (++spin > kDuoSpinLimit || rfl(*reinterpret_cast<volatile int *>(&sy->abort))) {
					dead = true;
					break;
				}
				__builtin_amdgcn_s_sleep(MVDB_DUO_POLL_SLEEP);
			}
			if (dead) break;
			if (kind == kRecEnd) {
#ifdef MVDB_TRAVERSE_PROFILE
				if (lane == 0 && a.prof)
					for (int i = 16; i < 20; i++) a.prof[(size_t)qi * kProfSlots + i] = prof_acc[i];
#endif
				return;
			}
			duo_publish(&sy->channel, u32x4_t{msg.x, msg.y, (unsigned)(2 * e + 1), msg.w}, lane);
		}
		if (lane == 0) *reinterpret_cast<volatile int *>(&sy->abort) = 1;
		return;
	}

	// =========================== result wave ===========================
	if (MVDB_DUO_RESULT_PRIO) __builtin_amdgcn_s_setprio(MVDB_DUO_RESULT_PRIO);
	int res_size = 0, cand_size = 1;  // the access wave pushed the entry point (:453)
	unsigned worst = 0;
	bool dead = false;
	// three waves: message to the push wave.  No release fence: the push wave reads nothing this wave
	// wrote but the message itself (the distances come from the access wave, seen through the record).
	auto post_adm = [&](int e, unsigned long long mask, int kind, bool done, unsigned w) {
		if (lane == 0)
			*reinterpret_cast<volatile lds_u32x4 *>((size_t)(unsigned)(size_t)&sy->adm[e & 1]) =
			    u32x4_t{(unsigned)mask, (unsigned)(mask >> 32),
			            ((unsigned)(e + 1) << 8) | ((unsigned)kind << 1) | (done ? 1u : 0u), w};
	};
	int last_e = 0;  // record the loop was left on
	HeapPipe<false> rpipe;
	rpipe.base = (unsigned)(size_t)res_h;
	for (int e = 0;; e++) {
		last_e = e;
		u32x4_t rec;
		for (int spin = 0;;) {
			rec = duo_snapshot(&sy->record[e & 1]);
			if ((int)rec.x >= e + 1) break;
			if (++spin > kDuoSpinLimit || rfl(*reinterpret_cast<volatile int *>(&sy->abort))) {
				dead = true;
				break;
			}
			__builtin_amdgcn_s_sleep(MVDB_DUO_POLL_SLEEP);
		}
		if (dead) break;
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
		MVDB_DUO_STAMP(5);
		const int kind = (int)(rec.y & 0xffu);
		if (kind == kRecEnd) break;
		unsigned long long adm_mask = 0;
		if (kind == kRecInit) {
			res.push(res_size++, (int)rec.z, rec.w, lane);  // :454
			worst = (unsigned)rfl((int)res_h[1].x);
		} else {
			if (res_size >= ef && rec.w > worst) break;  // :466, same values as on the access wave
			const int n = (int)((rec.y >> 8) & 0xffu), b = (int)((rec.y >> 16) & 1u);
			// admission in list order (:499-508); see traverse_kernel for the ballot walk
			const unsigned my_d = lane < n ? dist_b[64 * b + lane] : 0xffffffffu;
			const int my_id = lane < n ? ids_b[64 * b + lane] : -1;
			unsigned long long rem = n >= 64 ? ~0ull : ((1ull << n) - 1);
			for (;;) {
				const unsigned long long m = (res_size < ef ? ~0ull : __ballot(my_d < worst)) & rem;
				if (m == 0) break;
				const int t = __builtin_ctzll(m);
				rem &= ~((2ull << t) - 1);
				const unsigned d = (unsigned)__builtin_amdgcn_readlane((int)my_d, t);
				const int nid = __builtin_amdgcn_readlane(my_id, t);
#ifdef MVDB_TRAVERSE_PROFILE
				prof_admitted++;
#endif
				adm_mask |= 1ull << t;
				if (TRIO) post_adm(e, adm_mask, kRecList, false, 0u);  // the push wave can start on it
				if (res_size < ef) {
					res.push(res_size++, nid, d, lane);  // :504-505
					// `worst` is only consulted once the heap is full (:466, :499): one LDS round trip
					// per neighbour saved while it fills
					if (res_size >= ef) worst = (unsigned)rfl((int)res_h[1].x);
				} else {
					rpipe.tick();  // :506-507 replace_root, pipelined (see traverse_kernel)
					const int owner = rpipe.issue(lane, res_size, d, (unsigned)nid);
					rpipe.tick();
					worst = (unsigned)__builtin_amdgcn_readlane((int)rpipe.root_key, owner);
				}
			}
			MVDB_DUO_STAMP(6);
			if (TRIO) {
				post_adm(e, adm_mask, kRecList, true, worst);
				continue;
			}
			// :500-502 for the admitted neighbours, in list order, once pop e has left the heap
			for (int spin = 0;;) {
				if (rfl(*reinterpret_cast<volatile int *>(&sy->sift_done)) >= e) break;
				if (++spin > kDuoSpinLimit || rfl(*reinterpret_cast<volatile int *>(&sy->abort))) {
					dead = true;
					break;
				}
				__builtin_amdgcn_s_sleep(MVDB_DUO_POLL_SLEEP);
			}
			if (dead) break;
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
			cand_size--;  // pop e
			for (unsigned long long pm = adm_mask; pm && cand_size < cand_cap; pm &= pm - 1) {
				const int t = __builtin_ctzll(pm);
				cand.push(cand_size++, __builtin_amdgcn_readlane(my_id, t), (unsigned)__builtin_amdgcn_readlane((int)my_d, t),
				          lane);
			}
			MVDB_DUO_STAMP(15);
		}
		if (TRIO) post_adm(e, 0ull, kRecInit, true, worst);
		else
			duo_publish(&sy->channel, u32x4_t{(unsigned)adm_mask, (unsigned)(adm_mask >> 32), (unsigned)(2 * e + 1), worst}, lane);
	}
	if (TRIO && !dead) {
		// the push wave leaves; the candidate heap's arrays are reused below, so its last pushes
		// (record last_e - 1) must have landed
		post_adm(last_e, 0ull, kRecEnd, true, 0u);
		for (int spin = 0; last_e > 0;) {
			const u32x4_t ch = duo_snapshot(&sy->channel);
			if ((int)ch.z == 2 * (last_e - 1) + 1) break;
			if (++spin > kDuoSpinLimit || rfl(*reinterpret_cast<volatile int *>(&sy->abort))) {
				dead = true;
				break;
			}
			__builtin_amdgcn_s_sleep(MVDB_DUO_POLL_SLEEP);
		}
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
	}
	// The candidate heap's arrays stage the output below, and the access wave still reads the root
	// after the last channel post (its verification at the pop it then does not take): wait until
	// it has left.  Its counters are final then, and so is `abort`.
	if (!dead) {
		for (int spin = 0; rfl(*reinterpret_cast<volatile int *>(&sy->fin)) == 0 && spin < kDuoSpinLimit; spin++)
			__builtin_amdgcn_s_sleep(MVDB_DUO_POLL_SLEEP);
		if (rfl(*reinterpret_cast<volatile int *>(&sy->fin)) == 0) dead = true;
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
	}
	if (dead || rfl(*reinterpret_cast<volatile int *>(&sy->abort))) {
		if (lane == 0) sy->abort = 1;
		traverse_report_abort(a, qi, lane);
		return;
	}

	// res_to_sorted_full (:317-325): repeated max-heap pops, written back to front, pipelined.
	// The candidate heap is dead now (the other waves have left); its arrays stage the output.
	const int found = res_size;
	rpipe.drain();
	// set_only (see TraverseArgs): nothing is cut off, the caller needs the set -> heap order, no sort
	const bool unsorted = a.set_only && found <= a.k && found <= a.results_cap && found <= MVDB_MAX_EF;
	if (unsorted)
		for (int i = lane; i < found; i += 64) cand_h[i] = res_h[1 + i];
	for (int i = 0; i < (unsorted ? 0 : found); i++) {
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
#ifdef MVDB_TRAVERSE_PROFILE
	MVDB_DUO_STAMP(7);
	if (lane == 0 && a.prof) {
		for (int i = 5; i < 8; i++) a.prof[(size_t)qi * kProfSlots + i] = prof_acc[i];
		a.prof[(size_t)qi * kProfSlots + 15] = prof_acc[15];
		a.prof[(size_t)qi * kProfSlots + 9] = prof_admitted;
	}
#endif
	int out_count = found < a.k ? found : a.k;  // :1015-1017
	if (out_count > a.results_cap) out_count = a.results_cap;
	if (out_count > MVDB_MAX_EF) out_count = MVDB_MAX_EF;
	for (int i = lane; i < out_count; i += 64) {  // :1019-1024
		const uint2 en = cand_h[i];
		a.out_ids[(size_t)qi * (size_t)a.k + i] = (int)en.y;
		a.out_dist[(size_t)qi * (size_t)a.k + i] = __uint_as_float(en.x);
	}
	for (int i = out_count + lane; i < a.k; i += 64) {
		a.out_ids[(size_t)qi * (size_t)a.k + i] = -1;
		a.out_dist[(size_t)qi * (size_t)a.k + i] = 0.f;
	}
	if (lane == 0) {
		a.out_count[qi] = out_count;
		if (a.ctr) {
			mvdb_query_counters c;
			c.n_dist = sy->n_dist;
			c.n_l0lists = sy->n_l0;
			c.n_upperlists = sy->n_up;
			c.n_found = out_count;
			a.ctr[qi] = c;
		}
	}
}

}  // namespace mvdb
