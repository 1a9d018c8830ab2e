// device_index.hpp -- HBM-resident mirror of HostGraph, laid out for wave64 traversal.
//
// Layout (all 256-B aligned hipMalloc allocations):
//   vec   int8 [max_elements][row_stride]   row_stride = round_up(dim,16); pad bytes are 0, so
//                                           they add nothing to L2 / dot sums
//   adj0  int32[max_elements][m0]           layer-0 lists, CANONICALISED at upload: only ids that
//                                           the reference's scan would not skip (in range, present,
//                                           first occurrence, not the node itself), in list order,
//                                           -1 padded.  One 128-B line per node at m0 = 32.
//   upper int2 [n_upper_lists][m]           upper-layer lists of nodes with level >= 1, layer 1
//                                           first.  Entry = (neighbour id, code) with
//                                           code = (cum << 2) | level, cum = index of the
//                                           neighbour's own layer-1 list, so a walk never needs a
//                                           per-node offset lookup.  (-1,-1) padded.
// Why canonicalising is exact: search_layer skips ids <0, >=max_elements, absent or already
// visited (hnsw.ts:488-494) BEFORE computing a distance; a repeated id or the expanded node itself
// is always visited by then.  Upper lists drop only what the descent skips before its distance
// (out of range / absent, hnsw.ts:993-994); repeats and self stay, so even the work counters
// match the reference on foreign dumps.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "host_graph.hpp"

namespace mvdb {

struct DevIndexView {
	const int8_t *vec;
	const int32_t *adj0;
	const int2 *upper;
	int32_t dim, row_stride, m, m0, max_elements;
	int32_t entry, entry_code, max_level;
};

class DeviceIndex {
public:
	DevIndexView view{};
	const int32_t *norm2 = nullptr;  // |v|^2 per id, -1 = absent (exact scan only)
	uint64_t built_version = ~0ull;
	size_t bytes_vec = 0, bytes_adj0 = 0, bytes_upper = 0;

	~DeviceIndex() { release(); }
	// (re)build from the host graph; returns false and sets err on HIP failure
	bool upload(const HostGraph &g, std::string &err);
	// Incremental update (reference insert / update_and_reconnect, hnsw.ts:719-953, leave the rest
	// of the WASM memory untouched too): re-derive and overwrite only the rows of g.dirty -- int8 row, norm, canonical
	// layer-0 list, upper lists -- and append the upper lists of new nodes.  Returns 1 = patched, 0 = not possible
	// (caller falls back to upload: capacity changed, list slack exhausted, or a new id that an untouched list of a
	// foreign dump may already name), -1 = HIP failure (err set).
	int patch(const HostGraph &g, std::string &err);
	void release();
	long long full_uploads = 0, patches = 0, nodes_patched = 0;

	// A second copy of the arrays the traversal walks -- rows, layer-0 lists, upper lists -- under another LABELLING of
	// the nodes: breadth-first order over the layer-0 lists from the entry point, so that the neighbours a list names
	// share visited-bitmap lines and DRAM pages wherever the data has neighbourhoods.  Node ids are labels to
	// search_layer (its heaps compare distances only, hnsw.ts:141-315; nothing orders by id), so a traversal over the
	// relabelled arrays makes the reference's decisions one for one, and `perm` (new label -> id) translates its output.
	// Built by upload() -- and kept only if it pays: when a list's neighbours do not end up sharing bitmap lines
	// (i.i.d. rows have no neighbourhoods to find) the copy is dropped again, and with it the translation of the output
	// (relabel_mode 1; 2 = keep it regardless, 0 = never build it).  Skipped quietly when the memory is not there,
	// dropped by patch(): an edited mirror is walked under its own ids until the next full upload.  Everything that is
	// not a traversal keeps using `view`.
	DevIndexView view_r{};
	const int32_t *perm = nullptr;  // device array, nullptr = no relabelled copy
	int relabel_mode = 1;
	double relabel_lines_per_list = 0, relabel_entries_per_list = 0;  // sampled: bitmap lines / entries per layer-0 list under the new labels

private:
	void *d_vec_ = nullptr, *d_adj0_ = nullptr, *d_upper_ = nullptr, *d_norm2_ = nullptr;
	void *d_vec_r_ = nullptr, *d_adj0_r_ = nullptr, *d_upper_r_ = nullptr, *d_perm_ = nullptr;
	void relabel(const HostGraph &g, const std::vector<int32_t> &adj, const std::vector<int2> &up);
	void drop_relabelled();
	// host-side bookkeeping of the mirror
	std::vector<int32_t> code_;      // per node: (index of its layer-1 list << 2) | level, as uploaded
	std::vector<int8_t> dev_level_;  // level as uploaded, -1 = absent on the device
	int64_t n_lists_ = 0, cap_lists_ = 0;
	bool has_dangling_ = false;      // some list names an in-range id that is absent: its canonical form depends on presence
	void canon_adj0(const HostGraph &g, int i, int32_t *dst);
	void canon_upper(const HostGraph &g, int i, int l, int2 *dst);
};

}  // namespace mvdb
