// host_graph.cpp -- see host_graph.hpp.  Host C++ only (compiled by hipcc as C++).
#include "host_graph.hpp"

#include <unistd.h>

#include <algorithm>
#include <new>
#include <thread>

namespace mvdb {

// ---------------------------------------------------------------- math.ts
int32_t dist_l2_i8(const int8_t *a, const int8_t *b, int dim)
{
	int32_t s = 0;
	for (int i = 0; i < dim; i++) {
		int32_t d = (int32_t)a[i] - (int32_t)b[i];
		s += d * d;
	}
	return s;
}
int32_t dot_i8(const int8_t *a, const int8_t *b, int dim)
{
	int32_t s = 0;
	for (int i = 0; i < dim; i++) s += (int32_t)a[i] * (int32_t)b[i];
	return s;
}

// ------------------------------------------------- array heaps, hnsw.ts:141-325
// One body for both heaps: MIN=true is the candidate min-heap (parent `<=` stops,
// right child only if strictly smaller), MIN=false the result max-heap (mirror).
template <bool MIN>
struct ArrayHeap {
	int32_t *id;
	float *d;
	static bool stops(float parent, float child) { return MIN ? parent <= child : parent >= child; }
	static bool prefer_right(float r, float l) { return MIN ? r < l : r > l; }
	int push(int size, int32_t nid, float nd)
	{
		int i = size++;
		while (i > 0) {  // shift ancestors down while they do not stop the new key
			int p = (i - 1) >> 1;
			if (stops(d[p], nd)) break;
			id[i] = id[p];
			d[i] = d[p];
			i = p;
		}
		id[i] = nid;
		d[i] = nd;
		return size;
	}
	void sift_root(int size, int32_t nid, float nd)
	{
		int i = 0;
		for (;;) {
			int l = 2 * i + 1;
			if (l >= size) break;
			int c = l;
			if (l + 1 < size && prefer_right(d[l + 1], d[l])) c = l + 1;
			if (stops(nd, d[c])) break;
			id[i] = id[c];
			d[i] = d[c];
			i = c;
		}
		id[i] = nid;
		d[i] = nd;
	}
	// pop root into (rid, rd); returns new size
	int pop(int size, int32_t &rid, float &rd)
	{
		rid = id[0];
		rd = d[0];
		size--;
		if (size > 0) sift_root(size, id[size], d[size]);
		return size;
	}
};

BuildScratch::BuildScratch(int max_elements, int max_cap)
	: visited((size_t)std::max(max_elements, 1), 0u), cand_id(kMaxEf), res_id(kMaxEf), out_id(kMaxEf),
	  sel_id(kMaxEf), old_id(kMaxEf), cand_d(kMaxEf), res_d(kMaxEf), out_d(kMaxEf), used(kMaxEf),
	  nbr_copy((size_t)max_cap + 1)
{
}

// ------------------------------------------------------------------ storage
static size_t default_alloc_limit()
{
	const long pages = sysconf(_SC_PHYS_PAGES), psz = sysconf(_SC_PAGE_SIZE);
	if (pages <= 0 || psz <= 0) return (size_t)1 << 46;
	return (size_t)((double)pages * (double)psz * 0.8);
}
size_t HostGraph::alloc_limit_bytes = default_alloc_limit();

size_t HostGraph::bytes_for_capacity(int capacity) const
{
	if (capacity <= 0) return 0;
	// level + row + layer-0 count and list + upper offset + visited stamps of the build scratch + expected upper lists
	const size_t per = 1 + (size_t)cfg.dim + 4 + 4 * (size_t)cfg.m0() + 8 + 4 + 4 * (size_t)(1 + cfg.m);
	return per * (size_t)capacity;
}

void HostGraph::init(const GraphConfig &c, int capacity)
{
	cfg = c;
	entry_point = -1;
	max_level = -1;
	element_count = 0;
	ef_clamped = false;
	version++;
	dirty_all = true;
	dirty.clear();
	dirty_flag.clear();
	level.clear();
	vec.clear();
	cnt0.clear();
	adj0.clear();
	upper_off.clear();
	upper.clear();
	locks_.reset();
	if (capacity <= 0) {  // hnsw.ts:362-369
		max_elements = 0;
		main_.reset(new BuildScratch(0, cfg.m0()));
		return;
	}
	max_elements = capacity;
	level.assign((size_t)capacity, (int8_t)-1);
	vec.assign((size_t)capacity * cfg.dim, 0);
	cnt0.assign((size_t)capacity, 0);
	adj0.assign((size_t)capacity * cfg.m0(), -1);
	upper_off.assign((size_t)capacity, -1);
	if (results_cap <= 0) results_cap = 1000;  // hnsw.ts:378-379
	if (results_cap > kMaxEf) results_cap = kMaxEf;
	main_.reset(new BuildScratch(capacity, cfg.m0()));
}

int32_t *HostGraph::list_ids(int id, int layer)
{
	if (layer == 0) return adj0.data() + (size_t)id * cfg.m0();
	if (layer > level[id]) return nullptr;  // reference would read past the block; treated as empty
	return upper.data() + upper_off[id] + (size_t)(layer - 1) * (1 + cfg.m) + 1;
}
int32_t *HostGraph::list_count(int id, int layer)
{
	if (layer == 0) return cnt0.data() + id;
	if (layer > level[id]) return nullptr;
	return upper.data() + upper_off[id] + (size_t)(layer - 1) * (1 + cfg.m);
}

void HostGraph::create_node(int id, int lvl)  // insert :730-747 / load_index :1201-1205
{
	mark_dirty(id);
	level[id] = (int8_t)lvl;
	cnt0[id] = 0;
	std::fill_n(adj0.begin() + (size_t)id * cfg.m0(), cfg.m0(), -1);
	if (lvl > 0) {
		upper_off[id] = (int64_t)upper.size();
		size_t per = 1 + (size_t)cfg.m;
		upper.resize(upper.size() + per * lvl, -1);
		for (int l = 1; l <= lvl; l++) upper[upper_off[id] + (size_t)(l - 1) * per] = 0;
	} else {
		upper_off[id] = -1;
	}
}

// ------------------------------------------------------------------ dump V3
size_t HostGraph::dump_size() const
{
	size_t sz = kHeaderBytes;
	for (int i = 0; i < max_elements; i++) {
		if (level[i] < 0) continue;
		sz += 8 + (size_t)cfg.dim + 4 + 4 * (size_t)cfg.m0() + (size_t)level[i] * (4 + 4 * (size_t)cfg.m);
	}
	return sz;
}

static inline void put32(uint8_t *&p, int32_t v)
{
	std::memcpy(p, &v, 4);
	p += 4;
}

size_t HostGraph::save(uint8_t *out) const
{
	uint8_t *p = out;
	int present_n = 0;
	for (int i = 0; i < max_elements; i++) present_n += level[i] >= 0;
	const int32_t hdr[12] = {kMagic, kDumpVersion, cfg.dim, cfg.m, cfg.m0(), cfg.ef_construction, kMaxLayers,
	                         max_elements, present_n, entry_point, max_level, results_cap};
	for (int32_t v : hdr) put32(p, v);
	for (int i = 0; i < max_elements; i++) {
		int lvl = level[i];
		if (lvl < 0) continue;
		put32(p, i);
		put32(p, lvl);
		std::memcpy(p, row(i), (size_t)cfg.dim);
		p += cfg.dim;
		put32(p, cnt0[i]);
		std::memcpy(p, adj0.data() + (size_t)i * cfg.m0(), 4 * (size_t)cfg.m0());
		p += 4 * (size_t)cfg.m0();
		if (lvl > 0) {
			size_t n = (size_t)lvl * (1 + (size_t)cfg.m) * 4;
			std::memcpy(p, upper.data() + upper_off[i], n);
			p += n;
		}
	}
	return (size_t)(p - out);
}

bool HostGraph::load(const uint8_t *buf, size_t size)
{
	auto rd = [&](size_t off) {
		int32_t v;
		std::memcpy(&v, buf + off, 4);
		return v;
	};
	if (size < kHeaderBytes) return false;                       // :1131
	if (rd(0) != kMagic) return false;                           // :1134
	if (rd(4) != kDumpVersion) return false;                     // :1137
	const int32_t d_dim = rd(8), d_m = rd(12), d_m0 = rd(16), d_ef = rd(20), d_layers = rd(24);
	const int32_t d_max = rd(28), d_present = rd(32), d_entry = rd(36), d_maxlvl = rd(40), d_rcap = rd(44);
	if (d_dim != cfg.dim || d_m != cfg.m || d_m0 != cfg.m0() || d_ef != cfg.ef_construction ||
	    d_layers != kMaxLayers)
		return false;                                            // :1151-1159
	if (d_max <= 0 || d_present < 0 || d_present > d_max) return false;  // :1161-1164
	if (d_entry != -1 && (d_entry < 0 || d_entry >= d_max)) return false;  // :1167-1169
	if (d_maxlvl < -1 || d_maxlvl >= kMaxLayers) return false;   // :1170
	if (d_present > 0 && d_entry == -1) return false;            // :1173
	if (bytes_for_capacity(d_max) > alloc_limit_bytes) return false;  // memory.grow would trap (memory.ts:46-49)
	if (d_rcap > 0) results_cap = std::min(d_rcap, kMaxEf);      // :1176-1179

	try {
		init(cfg, d_max);                                        // :1181
	} catch (const std::bad_alloc &) {
		return false;
	}
	entry_point = d_entry;
	max_level = d_maxlvl;
	size_t off = kHeaderBytes;
	const int m0 = cfg.m0(), m = cfg.m, dim = cfg.dim;
	int loaded = 0;
	for (int n = 0; n < d_present; n++) {                        // :1188-1236
		if (off + 8 > size) return false;
		const int32_t id = rd(off), lvl = rd(off + 4);
		off += 8;
		if (id < 0 || id >= max_elements) return false;
		if (lvl < 0 || lvl >= kMaxLayers) return false;
		if (off + (size_t)dim > size) return false;
		std::memcpy(vec.data() + (size_t)id * dim, buf + off, (size_t)dim);
		off += dim;
		create_node(id, lvl);  // a repeated id gets a fresh block, like a second alloc()
		for (int l = 0; l <= lvl; l++) {
			const int c = l == 0 ? m0 : m;
			if (off + 4 + 4 * (size_t)c > size) return false;
			const int32_t count = rd(off);
			off += 4;
			if (count < 0 || count > c) return false;            // :1215
			*list_count(id, l) = count;
			int32_t *ids = list_ids(id, l);
			std::memcpy(ids, buf + off, 4 * (size_t)c);
			off += 4 * (size_t)c;
			for (int i = 0; i < c; i++)                          // :1224-1229
				if (ids[i] < 0 || ids[i] >= max_elements) ids[i] = -1;
		}
		loaded++;
	}
	element_count = loaded;
	if (entry_point != -1 && level[entry_point] < 0) {           // :1241-1246
		entry_point = -1;
		max_level = -1;
		element_count = 0;
		return false;
	}
	version++;
	return true;
}

// --------------------------------------------------------------- write path
int HostGraph::random_level()
{
	int lvl = 0;
	for (;;) {  // rand01() < 0.5 && level < MAX_LAYERS-1, hnsw.ts:61-75, :416-420
		uint32_t x = rng_state;
		x ^= x << 13;
		x ^= x >> 17;
		x ^= x << 5;
		rng_state = x;
		double r = (double)(x >> 8) / (double)0x01000000;
		if (r < 0.5 && lvl < kMaxLayers - 1) lvl++;
		else break;
	}
	return lvl;
}

void HostGraph::lock(int id)
{
	auto &l = locks_[id];
	for (;;) {
		uint8_t exp = 0;
		if (l.compare_exchange_weak(exp, 1, std::memory_order_acquire)) return;
		while (l.load(std::memory_order_relaxed)) __builtin_ia32_pause();
	}
}
void HostGraph::unlock(int id) { locks_[id].store(0, std::memory_order_release); }

// upper-layer greedy walk: insert :775-810, update :865-898
void HostGraph::greedy(const int8_t *q, int &cur, float &cur_d, int from, int stop, BuildScratch &s, bool par)
{
	for (int l = from; l > stop; l--) {
		bool changed = true;
		while (changed) {
			changed = false;
			if (level[cur] < 0) break;
			const int c = cap(l);
			int32_t *cnt_p = list_count(cur, l);
			int count = 0;
			const int32_t *ids = nullptr;
			if (cnt_p) {
				if (par) {
					lock(cur);
					s.nbr_copy[0] = *cnt_p;
					std::memcpy(s.nbr_copy.data() + 1, list_ids(cur, l), 4 * (size_t)c);
					unlock(cur);
					count = s.nbr_copy[0];
					ids = s.nbr_copy.data() + 1;
				} else {
					count = *cnt_p;
					ids = list_ids(cur, l);
				}
			}
			count = std::clamp(count, 0, c);
			for (int i = 0; i < count; i++) {
				const int nb = ids[i];
				if (!present(nb)) continue;
				const float d = dist_l2_sq(q, row(nb), cfg.dim);
				if (d < cur_d) {
					cur_d = d;
					cur = nb;
					changed = true;
				}
			}
		}
	}
}

int HostGraph::search_layer_build(const int8_t *q, int entry, int layer, int ef, int target, BuildScratch &s,
                                  bool par)
{
	if (ef <= 0) return 0;
	if (ef > kMaxEf) {
		ef = kMaxEf;
		ef_clamped = true;
	}
	if (!present(entry)) return 0;
	if (++s.stamp == 0) {  // visited_next_stamp :124-132
		s.stamp = 1;
		std::fill(s.visited.begin(), s.visited.end(), 0u);
	}
	const int cand_cap = std::max(std::min(2 * ef + 32, kMaxEf), ef);
	ArrayHeap<true> cand{s.cand_id.data(), s.cand_d.data()};
	ArrayHeap<false> res{s.res_id.data(), s.res_d.data()};
	int cn = 0, rn = 0;
	const float d0 = dist_l2_sq(q, row(entry), cfg.dim);
	s.visited[entry] = s.stamp;
	cn = cand.push(cn, entry, d0);
	rn = res.push(rn, entry, d0);
	const int c = cap(layer);
	while (cn > 0) {
		int32_t cur;
		float cur_d;
		cn = cand.pop(cn, cur, cur_d);
		if (cur < 0) continue;
		if (rn >= ef && cur_d > res.d[0]) break;
		if (level[cur] < 0) continue;
		int32_t *cnt_p = list_count(cur, layer);
		if (!cnt_p) continue;
		int count;
		const int32_t *ids;
		if (par) {
			lock(cur);
			s.nbr_copy[0] = *cnt_p;
			std::memcpy(s.nbr_copy.data() + 1, list_ids(cur, layer), 4 * (size_t)c);
			unlock(cur);
			count = s.nbr_copy[0];
			ids = s.nbr_copy.data() + 1;
		} else {
			count = *cnt_p;
			ids = list_ids(cur, layer);
		}
		if (count < 0) continue;
		count = std::min(count, c);
		for (int i = 0; i < count; i++) {
			const int nb = ids[i];
			if (!present(nb)) continue;
			if (s.visited[nb] == s.stamp) continue;
			s.visited[nb] = s.stamp;
			const float d = dist_l2_sq(q, row(nb), cfg.dim);
			if (rn < ef || d < res.d[0]) {
				if (cn < cand_cap) cn = cand.push(cn, nb, d);
				if (rn < ef) rn = res.push(rn, nb, d);
				else if (d < res.d[0]) res.sift_root(rn, nb, d);
			}
		}
	}
	// buildMode tail :517-522 + res_extract_k_smallest_sorted :327-358
	int pool = std::max(target * 2, target);
	pool = std::min({pool, ef, rn});
	if (pool <= 0 || rn <= 0) return 0;
	pool = std::min(pool, kMaxEf);
	std::fill_n(s.used.begin(), rn, (uint8_t)0);
	for (int o = 0; o < pool; o++) {
		int best = -1;
		float bd = 99999999.0f;
		for (int i = 0; i < rn; i++) {
			if (s.used[i]) continue;
			if (res.d[i] < bd) {
				bd = res.d[i];
				best = i;
			}
		}
		if (best < 0) break;
		s.used[best] = 1;
		s.out_id[o] = res.id[best];
		s.out_d[o] = bd;
	}
	return pool;
}

int HostGraph::select_neighbors(int found, int target, BuildScratch &s)
{
	s.sel_count = 0;
	if (found <= 0 || target <= 0) return 0;
	target = std::min(target, kMaxEf);
	for (int i = 0; i < found; i++) {
		const int c = s.out_id[i];
		const float d_cq = s.out_d[i];
		bool accept = true;
		for (int j = 0; j < s.sel_count; j++) {
			if (dist_l2_sq(row(c), row(s.sel_id[j]), cfg.dim) < d_cq) {
				accept = false;
				break;
			}
		}
		if (accept) {
			s.sel_id[s.sel_count++] = c;
			if (s.sel_count >= target) break;
		}
	}
	for (int i = 0; i < found && s.sel_count < target; i++) {  // back-fill :555-571
		const int c = s.out_id[i];
		if (std::find(s.sel_id.begin(), s.sel_id.begin() + s.sel_count, c) != s.sel_id.begin() + s.sel_count)
			continue;
		s.sel_id[s.sel_count++] = c;
	}
	return s.sel_count;
}

void HostGraph::remove_connection(int src, int dst, int layer)
{
	if (level[src] < 0) return;
	int32_t *cnt_p = list_count(src, layer);
	if (!cnt_p || *cnt_p <= 0) return;
	int32_t *ids = list_ids(src, layer);
	const int count = *cnt_p;
	for (int i = 0; i < count; i++) {
		if (ids[i] != dst) continue;
		ids[i] = ids[count - 1];  // swap with last active, :664-670
		ids[count - 1] = -1;
		*cnt_p = count - 1;
		mark_dirty(src);
		return;
	}
}

int HostGraph::add_connection(int src, int dst, int layer, bool defer_victim)
{
	if (level[src] < 0) return -1;
	int32_t *cnt_p = list_count(src, layer);
	if (!cnt_p) return -1;
	int32_t *ids = list_ids(src, layer);
	const int count = *cnt_p, c = cap(layer);
	for (int i = 0; i < count; i++)
		if (ids[i] == dst) return -1;
	if (count < c) {
		ids[count] = dst;
		*cnt_p = count + 1;
		mark_dirty(src);
		return -1;
	}
	int worst = -1;
	float worst_d = -1.0f;
	for (int i = 0; i < count; i++) {  // :617-631
		const int nb = ids[i];
		if (nb < 0 || nb >= max_elements) {
			worst = i;
			worst_d = 99999999.0f;
			break;
		}
		const float d = dist_l2_sq(row(src), row(nb), cfg.dim);
		if (d > worst_d) {
			worst_d = d;
			worst = i;
		}
	}
	const float dst_d = dist_l2_sq(row(src), row(dst), cfg.dim);
	if (worst >= 0 && dst_d < worst_d) {
		const int victim = ids[worst];
		ids[worst] = dst;
		mark_dirty(src);
		if (victim >= 0 && victim < max_elements && victim != dst && victim != src) {  // :640-642
			if (defer_victim) return victim;
			remove_connection(victim, src, layer);
		}
	}
	return -1;
}

void HostGraph::overwrite_neighbors(int id, int layer, int picked, BuildScratch &s)
{
	if (level[id] < 0) return;
	int32_t *cnt_p = list_count(id, layer);
	if (!cnt_p) return;
	int32_t *ids = list_ids(id, layer);
	const int c = cap(layer);
	int w = 0;
	for (int i = 0; i < picked && w < c; i++) {
		const int nb = s.sel_id[i];
		if (nb < 0 || nb >= max_elements || nb == id) continue;
		if (std::find(ids, ids + w, nb) != ids + w) continue;
		ids[w++] = nb;
	}
	*cnt_p = w;
	std::fill(ids + w, ids + c, -1);
	mark_dirty(id);
}

// the per-layer connect loop of insert, :812-827
void HostGraph::link_new_node(int id, int lvl, BuildScratch &s, bool par, int entry, int top)
{
	const int8_t *q = row(id);
	int cur = entry;
	float cur_d = dist_l2_sq(q, row(cur), cfg.dim);  // :772
	greedy(q, cur, cur_d, top, lvl, s, par);         // :775-810
	for (int l = std::min(lvl, top); l >= 0; l--) {
		const int target = cap(l);
		const int found = search_layer_build(q, cur, l, cfg.ef_construction, target, s, par);
		const int picked = select_neighbors(found, target, s);
		for (int i = 0; i < picked; i++) {
			const int nb = s.sel_id[i];
			if (!par) {
				add_connection(id, nb, l, false);
				add_connection(nb, id, l, false);
			} else {  // one node lock at a time; evictions applied after release
				lock(id);
				int v = add_connection(id, nb, l, true);
				unlock(id);
				if (v >= 0) {
					lock(v);
					remove_connection(v, id, l);
					unlock(v);
				}
				lock(nb);
				v = add_connection(nb, id, l, true);
				unlock(nb);
				if (v >= 0) {
					lock(v);
					remove_connection(v, nb, l);
					unlock(v);
				}
			}
		}
		if (found > 0) cur = s.out_id[0];
	}
}

void HostGraph::insert(int id, const int8_t *v)
{
	if (id < 0 || id >= max_elements) return;  // :720
	if (level[id] >= 0) {                      // :722-725
		update_and_reconnect(id, v);
		return;
	}
	version++;
	std::memcpy(vec.data() + (size_t)id * cfg.dim, v, (size_t)cfg.dim);
	const int lvl = random_level();
	create_node(id, lvl);
	if (entry_point == -1 || !present(entry_point)) {  // :749-770
		entry_point = id;
		max_level = lvl;
		element_count++;
		return;
	}
	link_new_node(id, lvl, *main_, false, entry_point, max_level);
	if (lvl > max_level) {  // :829-832
		max_level = lvl;
		entry_point = id;
	}
	element_count++;
}

void HostGraph::update_vector(int id, const int8_t *v)
{
	if (id < 0 || id >= max_elements) return;
	std::memcpy(vec.data() + (size_t)id * cfg.dim, v, (size_t)cfg.dim);
	mark_dirty(id);
	version++;
}

void HostGraph::update_and_reconnect(int id, const int8_t *v)
{
	if (id < 0 || id >= max_elements || level[id] < 0) return;
	version++;
	BuildScratch &s = *main_;
	std::memcpy(vec.data() + (size_t)id * cfg.dim, v, (size_t)cfg.dim);
	mark_dirty(id);
	int lvl = level[id];
	if (lvl >= kMaxLayers) lvl = kMaxLayers - 1;
	if (entry_point == -1 || !present(entry_point)) return;
	const int8_t *q = row(id);
	int cur = entry_point;
	float cur_d = dist_l2_sq(q, row(cur), cfg.dim);
	greedy(q, cur, cur_d, max_level, lvl, s, false);
	for (int l = std::min(lvl, max_level); l >= 0; l--) {
		const int target = cap(l);
		int old_n = 0;  // capture old neighbours, :904-919
		{
			int c = std::clamp(*list_count(id, l), 0, cap(l));
			const int32_t *ids = list_ids(id, l);
			for (int i = 0; i < c && i < kMaxEf; i++) s.old_id[old_n++] = ids[i];
		}
		const int found = search_layer_build(q, cur, l, cfg.ef_construction, target, s, false);
		const int picked = select_neighbors(found, target, s);
		for (int i = 0; i < old_n; i++) {  // :924-940
			const int o = s.old_id[i];
			if (o < 0 || o >= max_elements || o == id) continue;
			if (std::find(s.sel_id.begin(), s.sel_id.begin() + picked, o) == s.sel_id.begin() + picked)
				remove_connection(o, id, l);
		}
		overwrite_neighbors(id, l, picked, s);
		for (int i = 0; i < picked; i++) {  // :944-949
			const int nb = s.sel_id[i];
			if (nb < 0 || nb >= max_elements || nb == id) continue;
			add_connection(nb, id, l, false);
		}
		if (found > 0) cur = s.out_id[0];
	}
}

int HostGraph::bulk_begin(const int8_t *packed, int n, int prefix)
{
	return bulk_append_begin(packed, std::min(n, max_elements), prefix);
}

// ids element_count .. element_count + count - 1 (dense ids): vectors, levels (the RNG stream goes on in id order)
// and empty lists; the first of them are linked here, sequentially, until `min_linked` nodes exist and the top
// level is taken (a later node cannot raise the entry point then).  Returns the number of linked nodes.
int HostGraph::bulk_append_begin(const int8_t *packed, int count, int min_linked)
{
	const int first = element_count;
	count = std::min(count, max_elements - first);
	if (count <= 0) return first;
	const int n = first + count;
	version++;
	dirty_all = true;
	bulk_rng0_ = rng_state;  // bulk_abort() replays the draws of the nodes that stay
	bulk_first_ = first;
	std::memcpy(vec.data() + (size_t)first * cfg.dim, packed, (size_t)count * cfg.dim);
	std::vector<int8_t> lv((size_t)count);
	size_t upper_total = 0;
	for (int i = 0; i < count; i++) {
		lv[i] = (int8_t)random_level();
		upper_total += (size_t)lv[i] * (1 + (size_t)cfg.m);
	}
	upper.reserve(upper.size() + upper_total);
	for (int i = 0; i < count; i++) create_node(first + i, lv[i]);
	int k = first;
	if (first == 0 || entry_point == -1 || !present(entry_point)) {
		entry_point = first;
		max_level = lv[0];
		k = first + 1;
	}
	for (; k < n && (k < min_linked || max_level < kMaxLayers - 1); k++) {
		link_new_node(k, lv[k - first], *main_, false, entry_point, max_level);
		if (lv[k - first] > max_level) {  // insert :829-832
			max_level = lv[k - first];
			entry_point = k;
		}
	}
	element_count = k;
	return k;
}

void HostGraph::bulk_end(int n)
{
	element_count = std::min(n, max_elements);
	version++;
}

// Undo bulk_append_begin for the nodes it created but did not link (ids element_count .. n - 1), after a failed
// GPU pass: they become absent again, their upper lists (the tail of `upper`: node blocks were created in id
// order) are dropped, and the RNG stream is put where a sequential insert of the linked nodes would have left
// it, so the ids can be inserted again by any path and draw the levels they would have drawn.
void HostGraph::bulk_abort(int n)
{
	const int linked = element_count;
	n = std::min(n, max_elements);
	int64_t cut = -1;
	for (int id = linked; id < n; id++) {
		if (upper_off[id] >= 0 && (cut < 0 || upper_off[id] < cut)) cut = upper_off[id];
		level[id] = -1;
		cnt0[id] = 0;
		std::fill_n(adj0.begin() + (size_t)id * cfg.m0(), cfg.m0(), -1);
		upper_off[id] = -1;
	}
	if (cut >= 0) upper.resize((size_t)cut);
	rng_state = bulk_rng0_;
	for (int i = bulk_first_; i < linked; i++) (void)random_level();
	dirty_all = true;
	version++;
}

// Multi-threaded builder.  Levels come from the same RNG stream in id order, so
// the level assignment equals a sequential build; edge sets differ run to run.
void HostGraph::build_parallel(const int8_t *packed, int n, int threads)
{
	n = std::min(n, max_elements);
	if (n <= 0) return;
	threads = std::max(threads, 1);
	version++;
	dirty_all = true;  // worker threads do not track single nodes
	std::memcpy(vec.data(), packed, (size_t)n * cfg.dim);
	std::vector<int8_t> lv((size_t)n);
	size_t upper_total = 0;
	for (int i = 0; i < n; i++) {
		lv[i] = (int8_t)random_level();
		upper_total += (size_t)lv[i] * (1 + (size_t)cfg.m);
	}
	upper.reserve(upper.size() + upper_total);  // no reallocation while threads run
	for (int i = 0; i < n; i++) create_node(i, lv[i]);
	locks_.reset(new std::atomic<uint8_t>[(size_t)max_elements]);
	for (int i = 0; i < max_elements; i++) locks_[i].store(0, std::memory_order_relaxed);

	entry_point = 0;
	max_level = lv[0];
	element_count = 1;
	std::atomic<int> next{1};
	std::atomic<int> done{1};
	auto worker = [&](bool is_main) {
		std::unique_ptr<BuildScratch> own;
		BuildScratch *s = main_.get();
		if (!is_main) {
			own.reset(new BuildScratch(max_elements, cfg.m0()));
			s = own.get();
		}
		for (;;) {
			const int id = next.fetch_add(1);
			if (id >= n) break;
			const int lvl = lv[id];
			// entry-point handover is serialised: a node that raises max_level
			// holds the global lock for its whole insert
			for (;;) {
				uint8_t exp = 0;
				if (global_lock_.compare_exchange_weak(exp, 1, std::memory_order_acquire)) break;
				while (global_lock_.load(std::memory_order_relaxed)) __builtin_ia32_pause();
			}
			// (entry_point, max_level) are read as a pair under the lock that guards their update
			const int entry = entry_point, top = max_level;
			const bool raises = lvl > top;
			if (!raises) global_lock_.store(0, std::memory_order_release);
			link_new_node(id, lvl, *s, true, entry, top);
			if (raises) {
				max_level = lvl;
				entry_point = id;
				global_lock_.store(0, std::memory_order_release);
			}
			done.fetch_add(1);
		}
	};
	// a sequential prefix seeds a connected graph before going wide
	{
		const int prefix = std::min(n, 1024);
		while (next.load() < prefix) {
			const int id = next.fetch_add(1);
			const int lvl = lv[id];
			link_new_node(id, lvl, *main_, true, entry_point, max_level);
			if (lvl > max_level) {
				max_level = lvl;
				entry_point = id;
			}
			done.fetch_add(1);
		}
	}
	std::vector<std::thread> pool;
	for (int t = 1; t < threads; t++) pool.emplace_back(worker, false);
	worker(true);
	for (auto &t : pool) t.join();
	element_count = done.load();
	locks_.reset();
}

}  // namespace mvdb
