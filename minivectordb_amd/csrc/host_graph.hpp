// host_graph.hpp -- host-resident HNSW graph of the engine (C++17, no GPU code).
//
// Owns what the reference keeps inside WASM linear memory (assembly/hnsw.ts:10-52,
// assembly/types.ts): node levels, per-layer neighbour lists, int8 rows.  Laid out
// as struct-of-arrays (not the reference's variable-size node blocks) because the
// device mirror is built from it by plain strided copies.
//
//  * parse/serialise dump.bin V3 byte-exactly        (hnsw.ts:1033-1249)
//  * the write path: insert / update_and_reconnect   (hnsw.ts:526-953)
//  * a multi-threaded builder for benchmark graphs    (same algorithm, unordered)
//
// The search path proper runs on the GPU (traverse.hip); the host-side
// search_layer here exists only in build mode, as the write path needs it.
#pragma once

#include <atomic>
#include <cstdint>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

namespace mvdb {

constexpr int kMaxEf = 4096;     // hnsw.ts:30
constexpr int kMaxLayers = 4;    // assembly/config.ts:27
constexpr int32_t kMagic = 0x57534e48;  // hnsw.ts:1033
constexpr int32_t kDumpVersion = 3;     // hnsw.ts:1034
constexpr size_t kHeaderBytes = 48;     // hnsw.ts:1037-1038

struct GraphConfig {
	int dim = 384, m = 16, ef_construction = 100;
	int m0() const { return 2 * m; }  // config.ts:115
};

// Scratch for one builder thread: the module-global buffers of hnsw.ts:29-52.
struct BuildScratch {
	std::vector<uint32_t> visited;
	uint32_t stamp = 1;
	std::vector<int32_t> cand_id, res_id, out_id, sel_id, old_id;
	std::vector<float> cand_d, res_d, out_d;
	std::vector<uint8_t> used;
	std::vector<int32_t> nbr_copy;
	int sel_count = 0;
	explicit BuildScratch(int max_elements, int max_cap);
};

class HostGraph {
public:
	GraphConfig cfg;
	int max_elements = 0;
	int entry_point = -1;
	int max_level = -1;
	int element_count = 0;
	int results_cap = 1000;
	uint32_t rng_state = 2463534242u;
	bool ef_clamped = false;

	// SoA storage, all sized by max_elements
	std::vector<int8_t> level;       // -1 = absent (node_ptr == 0)
	std::vector<int8_t> vec;         // [max_elements][dim]
	std::vector<int32_t> cnt0;       // layer-0 count
	std::vector<int32_t> adj0;       // [max_elements][m0]
	std::vector<int64_t> upper_off;  // offset into `upper`, -1 when level < 1
	std::vector<int32_t> upper;      // per upper node: level x (count, m ids)
	uint64_t version = 0;            // bumped by every mutation (device mirror staleness)
	// Nodes whose row or lists changed since `dirty_base` (the version at the last clear_dirty): what the device
	// mirror has to patch.  dirty_all = everything changed (init, load, bulk builders): patching is off.
	std::vector<int32_t> dirty;
	std::vector<uint8_t> dirty_flag;
	bool dirty_all = true;
	uint64_t dirty_base = 0;
	void mark_dirty(int id)
	{
		if (dirty_all || id < 0 || id >= max_elements || dirty_flag[id]) return;
		dirty_flag[id] = 1;
		dirty.push_back(id);
	}
	void clear_dirty()
	{
		for (int id : dirty) dirty_flag[id] = 0;
		dirty.clear();
		dirty_all = false;
		dirty_base = version;
		if ((int)dirty_flag.size() != max_elements) dirty_flag.assign((size_t)std::max(max_elements, 0), 0);
	}

	void init(const GraphConfig &c, int capacity);  // init_index, hnsw.ts:361-414
	bool present(int id) const { return id >= 0 && id < max_elements && level[id] >= 0; }
	int cap(int layer) const { return layer == 0 ? cfg.m0() : cfg.m; }
	// pointer to [count, ids...] of `layer` (count stored separately for layer 0)
	int32_t *list_ids(int id, int layer);
	int32_t *list_count(int id, int layer);
	const int8_t *row(int id) const { return vec.data() + (size_t)id * cfg.dim; }

	// A dump header names the capacity to allocate (init_index(dump_max_elements), hnsw.ts:1181).  The WASM traps when
	// memory.grow fails (assembly/memory.ts:46-49) and the bridge reports a failed load; here a capacity whose arrays
	// would not fit `alloc_limit_bytes` (default: 80 % of physical memory) is a failed load too, not an OOM kill.
	static size_t alloc_limit_bytes;
	size_t bytes_for_capacity(int capacity) const;
	// dump V3
	size_t dump_size() const;                          // hnsw.ts:1048-1073
	size_t save(uint8_t *out) const;                   // hnsw.ts:1075-1127
	bool load(const uint8_t *buf, size_t size);        // hnsw.ts:1129-1249 (false = returned 0)

	// write path
	void seed(uint32_t s) { rng_state = s != 0 ? s : 2463534242u; }  // hnsw.ts:57-59
	void insert(int id, const int8_t *v);              // hnsw.ts:719-834 (upsert)
	void update_vector(int id, const int8_t *v);       // hnsw.ts:837-841
	void build_parallel(const int8_t *packed, int n, int threads);
	// Shared prologue of the bulk builders: rows copied, levels drawn from the RNG stream in id order (the
	// level assignment of a sequential build), every node block created, then ids [0, k) linked one by
	// one with the reference algorithm, k >= prefix and large enough for max_level to have reached its
	// cap (so that entry point and max_level are final).  Returns k.  bulk_end() publishes n nodes.
	int bulk_begin(const int8_t *packed, int n, int prefix);
	int bulk_append_begin(const int8_t *packed, int count, int min_linked);
	void bulk_end(int n);
	void bulk_abort(int n);  // after a failure between the two: the unlinked nodes are taken back

private:
	std::unique_ptr<BuildScratch> main_;
	std::unique_ptr<std::atomic<uint8_t>[]> locks_;
	std::atomic<uint8_t> global_lock_{0};
	uint32_t bulk_rng0_ = 0;  // RNG state and first id of the bulk append in progress (bulk_abort)
	int bulk_first_ = 0;

	int random_level();                                 // hnsw.ts:416-420
	void create_node(int id, int lvl);
	void update_and_reconnect(int id, const int8_t *v); // hnsw.ts:844-953
	void greedy(const int8_t *q, int &cur, float &cur_d, int from, int stop, BuildScratch &s, bool par);
	int search_layer_build(const int8_t *q, int entry, int layer, int ef, int target, BuildScratch &s,
	                       bool par);                   // hnsw.ts:423-523, buildMode = true
	int select_neighbors(int found, int target, BuildScratch &s);  // hnsw.ts:526-574
	int add_connection(int src, int dst, int layer, bool defer_victim);  // hnsw.ts:593-645
	void remove_connection(int src, int dst, int layer);             // hnsw.ts:648-678
	void overwrite_neighbors(int id, int layer, int picked, BuildScratch &s);  // hnsw.ts:681-716
	void link_new_node(int id, int lvl, BuildScratch &s, bool par, int entry, int top);
	void lock(int id);
	void unlock(int id);
};

// exact integer distances, assembly/math.ts:115,133
int32_t dist_l2_i8(const int8_t *a, const int8_t *b, int dim);
int32_t dot_i8(const int8_t *a, const int8_t *b, int dim);
inline float dist_l2_sq(const int8_t *a, const int8_t *b, int dim) { return (float)dist_l2_i8(a, b, dim); }

}  // namespace mvdb
