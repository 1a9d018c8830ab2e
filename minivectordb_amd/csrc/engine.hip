// engine.hip -- C ABI of libmvdb_hip.so (include/mvdb_hip.h): host graph + device mirror +
// kernel launches.  gfx950 only; no CPU fallback for any compute entry point.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "device_index.hpp"
#include "host_graph.hpp"
#include "mvdb_hip.h"
#include "rerank.hip.hpp"
#include "scan.hip.hpp"
#include "traverse.hip.hpp"
#include "traverse_lean.hip.hpp"
#include "build_gpu.hip.hpp"

using namespace mvdb;

namespace {

thread_local std::string g_create_error;

#define HIP_OK(h, expr)                                                                         \
	do {                                                                                        \
		hipError_t e_ = (expr);                                                                 \
		if (e_ != hipSuccess) {                                                                 \
			(h)->set_error(std::string(#expr) + ": " + hipGetErrorString(e_));                  \
			return 0;                                                                           \
		}                                                                                       \
	} while (0)

struct DevBuf {
	void *p = nullptr;
	size_t cap = 0;
	~DevBuf() { (void)hipFree(p); }
	// grow-only scratch; returns nullptr on failure
	void *need(size_t bytes)
	{
		if (bytes <= cap && p) return p;
		if (p) (void)hipFree(p);
		p = nullptr;
		cap = 0;
		size_t want = std::max<size_t>(bytes, 256);
		if (hipMalloc(&p, want) != hipSuccess) {
			p = nullptr;
			return nullptr;
		}
		cap = want;
		return p;
	}
};

}  // namespace

// --------------------------------------------------------------------------- DeviceIndex
void DeviceIndex::drop_relabelled()
{
	(void)hipFree(d_vec_r_);  // (hipFree waits for the device: no traversal still walks these)
	(void)hipFree(d_adj0_r_);
	(void)hipFree(d_upper_r_);
	(void)hipFree(d_perm_);
	d_vec_r_ = d_adj0_r_ = d_upper_r_ = d_perm_ = nullptr;
	perm = nullptr;
	view_r = DevIndexView{};
}

void DeviceIndex::release()
{
	drop_relabelled();
	(void)hipFree(d_vec_);
	(void)hipFree(d_adj0_);
	(void)hipFree(d_upper_);
	(void)hipFree(d_norm2_);
	d_vec_ = d_adj0_ = d_upper_ = d_norm2_ = nullptr;
	norm2 = nullptr;
	view = DevIndexView{};
	built_version = ~0ull;
	code_.clear();
	dev_level_.clear();
	n_lists_ = cap_lists_ = 0;
	has_dangling_ = false;
}

// layer-0 list of node i as the device holds it (see device_index.hpp): ids the reference scan would not skip
void DeviceIndex::canon_adj0(const HostGraph &g, int i, int32_t *dst)
{
	const int m0 = g.cfg.m0();
	std::fill_n(dst, m0, -1);
	if (g.level[i] < 0) return;
	const int32_t *src = g.adj0.data() + (size_t)i * m0;
	const int cnt = std::clamp(g.cnt0[i], 0, m0);
	int w = 0;
	for (int j = 0; j < cnt; j++) {
		const int nb = src[j];
		if (nb >= 0 && nb < g.max_elements && g.level[nb] < 0) has_dangling_ = true;
		if (!g.present(nb) || nb == i) continue;
		if (std::find(dst, dst + w, nb) != dst + w) continue;
		dst[w++] = nb;
	}
}

// upper list l (>= 1) of node i as (id, code) pairs
void DeviceIndex::canon_upper(const HostGraph &g, int i, int l, int2 *dst)
{
	const int m = g.cfg.m;
	std::fill_n(dst, m, make_int2(-1, -1));
	const int32_t *rec = g.upper.data() + g.upper_off[i] + (size_t)(l - 1) * (1 + m);
	const int cnt = std::clamp(rec[0], 0, m);
	int w = 0;
	for (int j = 0; j < cnt; j++) {
		const int nb = rec[1 + j];
		if (nb >= 0 && nb < g.max_elements && g.level[nb] < 0) has_dangling_ = true;
		if (!g.present(nb)) continue;
		dst[w++] = make_int2(nb, code_[nb]);
	}
}

bool DeviceIndex::upload(const HostGraph &g, std::string &err)
{
	release();
	const int N = g.max_elements, dim = g.cfg.dim, m = g.cfg.m, m0 = g.cfg.m0();
	const int stride = (dim + 15) & ~15;
	view.dim = dim;
	view.row_stride = stride;
	view.m = m;
	view.m0 = m0;
	view.max_elements = N;
	view.entry = g.entry_point;
	view.max_level = g.max_level;
	view.entry_code = 0;
	built_version = g.version;
	full_uploads++;
	if (N <= 0) return true;

	// code[id] = (index of the node's layer-1 list among all upper lists << 2) | level
	code_.assign((size_t)N, 0);
	dev_level_.assign(g.level.begin(), g.level.begin() + N);
	n_lists_ = 0;
	for (int i = 0; i < N; i++) {
		const int lvl = g.level[i];
		if (lvl < 0) continue;
		if (n_lists_ > (int64_t)0x1fffffff) {
			err = "too many upper-layer lists for the 29-bit list index";
			return false;
		}
		code_[i] = (int32_t)((n_lists_ << 2) | lvl);
		n_lists_ += lvl;
	}
	// room for the upper lists of nodes inserted later (DeviceIndex::patch appends)
	cap_lists_ = std::min<int64_t>(n_lists_ + std::max<int64_t>(4096, n_lists_ / 8), (int64_t)0x1fffffff);
	if (g.present(g.entry_point)) view.entry_code = code_[g.entry_point];

	auto check = [&](hipError_t e, const char *what) {
		if (e == hipSuccess) return true;
		err = std::string(what) + ": " + hipGetErrorString(e);
		return false;
	};

	// int8 rows, padded to the 16-byte stride
	bytes_vec = (size_t)N * stride;
	if (!check(hipMalloc(&d_vec_, std::max<size_t>(bytes_vec, 256)), "hipMalloc(vec)")) return false;
	if (stride == dim) {
		if (!check(hipMemcpy(d_vec_, g.vec.data(), bytes_vec, hipMemcpyHostToDevice), "copy vec")) return false;
	} else {
		if (!check(hipMemset(d_vec_, 0, bytes_vec), "memset vec")) return false;
		if (!check(hipMemcpy2D(d_vec_, stride, g.vec.data(), dim, dim, N, hipMemcpyHostToDevice), "copy vec 2D"))
			return false;
	}

	// |v|^2 per node for the exact scan, -1 marks an absent id
	{
		std::vector<int32_t> n2((size_t)N, -1);
		for (int i = 0; i < N; i++)
			if (g.level[i] >= 0) n2[i] = dot_i8(g.row(i), g.row(i), dim);
		if (!check(hipMalloc(&d_norm2_, std::max<size_t>((size_t)N * 4, 256)), "hipMalloc(norm2)")) return false;
		if (!check(hipMemcpy(d_norm2_, n2.data(), (size_t)N * 4, hipMemcpyHostToDevice), "copy norm2")) return false;
		norm2 = static_cast<const int32_t *>(d_norm2_);
	}

	// layer 0, canonicalised (see device_index.hpp)
	std::vector<int32_t> adj((size_t)N * m0);
	{
		for (int i = 0; i < N; i++) canon_adj0(g, i, adj.data() + (size_t)i * m0);
		bytes_adj0 = adj.size() * 4;
		if (!check(hipMalloc(&d_adj0_, std::max<size_t>(bytes_adj0, 256)), "hipMalloc(adj0)")) return false;
		if (!check(hipMemcpy(d_adj0_, adj.data(), bytes_adj0, hipMemcpyHostToDevice), "copy adj0")) return false;
	}
	// upper layers as (id, code) pairs
	std::vector<int2> up((size_t)n_lists_ * m, make_int2(-1, -1));
	{
		for (int i = 0; i < N; i++) {
			const int lvl = g.level[i];
			if (lvl < 1) continue;
			const int64_t cum = code_[i] >> 2;
			for (int l = 1; l <= lvl; l++) canon_upper(g, i, l, up.data() + (size_t)(cum + l - 1) * m);
		}
		bytes_upper = (size_t)cap_lists_ * m * sizeof(int2);
		if (!check(hipMalloc(&d_upper_, std::max<size_t>(bytes_upper, 256)), "hipMalloc(upper)")) return false;
		if (!up.empty() &&
		    !check(hipMemcpy(d_upper_, up.data(), up.size() * sizeof(int2), hipMemcpyHostToDevice), "copy upper"))
			return false;
	}
	view.vec = static_cast<const int8_t *>(d_vec_);
	view.adj0 = static_cast<const int32_t *>(d_adj0_);
	view.upper = static_cast<const int2 *>(d_upper_);
	if (relabel_mode != 0 && N >= 2) relabel(g, adj, up);
	return true;
}

namespace {
// dst row r <- src row perm[r], rows of row_bytes (a multiple of 16), one 16-byte piece per thread
__global__ void gather_rows_kernel(uint4 *dst, const uint4 *src, const int32_t *perm, size_t pieces_per_row, size_t n_pieces)
{
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_pieces) return;
	const size_t r = i / pieces_per_row, c = i - r * pieces_per_row;
	dst[i] = src[(size_t)perm[r] * pieces_per_row + c];
}
}  // namespace

// The relabelled copy (device_index.hpp).  `adj` and `up` are the canonical lists upload() has just sent.
void DeviceIndex::relabel(const HostGraph &g, const std::vector<int32_t> &adj, const std::vector<int2> &up)
{
	const int N = g.max_elements, m0 = g.cfg.m0();
	// breadth-first order over the layer-0 lists, from the entry point; further components and, last, the absent ids
	// follow in id order.  order[] doubles as the queue.
	std::vector<int32_t> order((size_t)N), inv((size_t)N, -1);
	int w = 0, r = 0, scan = 0;
	auto found = [&](int id) {
		inv[id] = w;
		order[w++] = id;
	};
	if (g.present(g.entry_point)) found(g.entry_point);
	for (;;) {
		while (r < w) {
			const int32_t *l = adj.data() + (size_t)order[r++] * m0;
			for (int j = 0; j < m0 && l[j] >= 0; j++)
				if (inv[l[j]] < 0) found(l[j]);
		}
		while (scan < N && (g.level[scan] < 0 || inv[scan] >= 0)) scan++;
		if (scan >= N) break;
		found(scan);
	}
	for (int i = 0; i < N; i++)
		if (inv[i] < 0) found(i);
	// does it pay?  How many 1024-label lines of the visited bitmap do the entries of a list touch under the new labels
	// (a uniform sample of lists; the lists expanded first would flatter the order: their neighbours were numbered by them)
	{
		std::vector<int32_t> tv((size_t)m0);
		double lines = 0, entries = 0;
		const int step = std::max(1, N / 4096);
		for (int i = 0; i < N; i += step) {
			const int32_t *l = adj.data() + (size_t)i * m0;
			int c = 0;
			for (int j = 0; j < m0 && l[j] >= 0; j++) tv[c++] = inv[l[j]] >> 10;
			std::sort(tv.begin(), tv.begin() + c);
			lines += (double)(std::unique(tv.begin(), tv.begin() + c) - tv.begin());
			entries += c;
		}
		relabel_lines_per_list = lines;
		relabel_entries_per_list = entries;
		if (relabel_mode == 1 && 2 * lines > entries) return;  // fewer than two neighbours per line: no neighbourhoods to exploit (i.i.d. rows)
	}
	// lists under the new labels: the list of label a is the list of node order[a], its entries translated, in list order
	std::vector<int32_t> adj_r((size_t)N * m0);
	std::vector<int2> up_r(up);  // same list order (the codes index it), neighbour labels translated
	{
		const int nt = N >= (1 << 18) ? 8 : 1;
		std::vector<std::thread> th;
		for (int t = 0; t < nt; t++)
			th.emplace_back([&, t] {
				for (int a = (int)((int64_t)N * t / nt), a1 = (int)((int64_t)N * (t + 1) / nt); a < a1; a++) {
					const int32_t *src = adj.data() + (size_t)order[a] * m0;
					int32_t *dst = adj_r.data() + (size_t)a * m0;
					for (int j = 0; j < m0; j++) dst[j] = src[j] >= 0 ? inv[src[j]] : -1;
				}
				for (size_t i = up_r.size() * t / nt, i1 = up_r.size() * (t + 1) / nt; i < i1; i++)
					if (up_r[i].x >= 0) up_r[i].x = inv[up_r[i].x];
			});
		for (auto &x : th) x.join();
	}
	const bool ok = hipMalloc(&d_perm_, (size_t)N * 4) == hipSuccess &&
	                hipMalloc(&d_vec_r_, std::max<size_t>(bytes_vec, 256)) == hipSuccess &&
	                hipMalloc(&d_adj0_r_, std::max<size_t>(bytes_adj0, 256)) == hipSuccess &&
	                hipMalloc(&d_upper_r_, std::max<size_t>(bytes_upper, 256)) == hipSuccess &&
	                hipMemcpy(d_perm_, order.data(), (size_t)N * 4, hipMemcpyHostToDevice) == hipSuccess &&
	                hipMemcpy(d_adj0_r_, adj_r.data(), bytes_adj0, hipMemcpyHostToDevice) == hipSuccess &&
	                (up_r.empty() || hipMemcpy(d_upper_r_, up_r.data(), up_r.size() * sizeof(int2), hipMemcpyHostToDevice) == hipSuccess);
	if (ok) {
		const size_t ppr = (size_t)view.row_stride / 16, np = (size_t)N * ppr;
		hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, nullptr, static_cast<uint4 *>(d_vec_r_),
		                   static_cast<const uint4 *>(d_vec_), static_cast<const int32_t *>(d_perm_), ppr, np);
	}
	if (!ok || hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
		(void)hipGetLastError();  // out of memory for the copy: the traversal walks the arrays under their own ids
		drop_relabelled();
		return;
	}
	view_r = view;
	view_r.vec = static_cast<const int8_t *>(d_vec_r_);
	view_r.adj0 = static_cast<const int32_t *>(d_adj0_r_);
	view_r.upper = static_cast<const int2 *>(d_upper_r_);
	if (g.present(g.entry_point)) view_r.entry = inv[g.entry_point];
	perm = static_cast<const int32_t *>(d_perm_);
}

namespace {
// dst row idx[r] <- src row r, rows of row_bytes (a multiple of 4)
__global__ void scatter_rows_kernel(unsigned *dst, size_t row_words, const int32_t *idx, const unsigned *src, int n)
{
	const int r = blockIdx.x;
	if (r >= n) return;
	unsigned *d = dst + (size_t)idx[r] * row_words;
	const unsigned *s = src + (size_t)r * row_words;
	for (size_t i = threadIdx.x; i < row_words; i += blockDim.x) d[i] = s[i];
}
}  // namespace

int DeviceIndex::patch(const HostGraph &g, std::string &err)
{
	const int N = g.max_elements, dim = g.cfg.dim, m = g.cfg.m, m0 = g.cfg.m0();
	if (g.dirty_all || built_version != g.dirty_base || N != view.max_elements || N <= 0 || !d_vec_) return 0;
	const size_t nd = g.dirty.size();
	if (nd == 0) {
		built_version = g.version;
		return 1;
	}
	if (nd > (size_t)N / 4) return 0;  // a full upload is cheaper
	drop_relabelled();  // the edited mirror is walked under its own ids until the next full upload
	const int stride = view.row_stride;
	// new nodes first: their codes are named by the lists of the other dirty nodes
	int64_t new_lists = 0;  // upper lists the whole batch appends into the slack
	for (int id : g.dirty) {
		const int lvl = g.level[id];
		if (lvl < 0) return 0;  // nodes never disappear (no delete in the ANN index)
		if (dev_level_[id] >= 0) {
			if (dev_level_[id] != lvl) return 0;  // a re-created node: not an edit of the mirror
			continue;
		}
		if (has_dangling_) return 0;  // an untouched list may name this id: its canonical form changes with presence
		new_lists += lvl;
		if (n_lists_ + new_lists > cap_lists_) return 0;  // slack exhausted: the full upload re-sizes it
	}
	for (int id : g.dirty) {
		if (dev_level_[id] >= 0) continue;
		const int lvl = g.level[id];
		code_[id] = (int32_t)((n_lists_ << 2) | lvl);
		n_lists_ += lvl;
		dev_level_[id] = (int8_t)lvl;
	}
	// staging: one row of each array per dirty node, one row per upper list
	std::vector<int32_t> ids(g.dirty.begin(), g.dirty.end());
	std::vector<int8_t> vrows(nd * stride, 0);
	std::vector<int32_t> n2(nd), arows(nd * m0), lidx;
	std::vector<int2> urows;
	for (size_t r = 0; r < nd; r++) {
		const int id = ids[r];
		std::memcpy(vrows.data() + r * stride, g.row(id), (size_t)dim);
		n2[r] = dot_i8(g.row(id), g.row(id), dim);
		canon_adj0(g, id, arows.data() + r * m0);
		const int64_t cum = code_[id] >> 2;
		for (int l = 1; l <= g.level[id]; l++) {
			lidx.push_back((int32_t)(cum + l - 1));
			urows.resize(urows.size() + m);
			canon_upper(g, id, l, urows.data() + urows.size() - m);
		}
	}
	if (has_dangling_) return 0;  // a patched list itself names an absent id now: rebuild with the flag set
	auto fail = [&](hipError_t e, const char *what) {
		err = std::string(what) + ": " + hipGetErrorString(e);
		return -1;
	};
	const size_t b_ids = nd * 4, b_v = vrows.size(), b_n2 = nd * 4, b_a = arows.size() * 4, b_li = lidx.size() * 4,
	             b_u = urows.size() * sizeof(int2);
	auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
	const size_t o_v = al(b_ids), o_n2 = o_v + al(b_v), o_a = o_n2 + al(b_n2), o_li = o_a + al(b_a), o_u = o_li + al(b_li);
	unsigned char *stg = nullptr;
	hipError_t e = hipMalloc(reinterpret_cast<void **>(&stg), o_u + al(b_u) + 256);
	if (e != hipSuccess) return fail(e, "hipMalloc(patch staging)");
	auto put = [&](size_t off, const void *p, size_t n) { return n ? hipMemcpy(stg + off, p, n, hipMemcpyHostToDevice) : hipSuccess; };
	if ((e = put(0, ids.data(), b_ids)) != hipSuccess || (e = put(o_v, vrows.data(), b_v)) != hipSuccess ||
	    (e = put(o_n2, n2.data(), b_n2)) != hipSuccess || (e = put(o_a, arows.data(), b_a)) != hipSuccess ||
	    (e = put(o_li, lidx.data(), b_li)) != hipSuccess || (e = put(o_u, urows.data(), b_u)) != hipSuccess) {
		(void)hipFree(stg);
		return fail(e, "copy patch staging");
	}
	const int32_t *d_ids = reinterpret_cast<const int32_t *>(stg);
	auto scatter = [&](void *dst, size_t row_bytes, const int32_t *idx, size_t off, size_t n) {
		if (n == 0) return;
		hipLaunchKernelGGL(scatter_rows_kernel, dim3((unsigned)n), dim3(64), 0, nullptr, static_cast<unsigned *>(dst), row_bytes / 4,
		                   idx, reinterpret_cast<const unsigned *>(stg + off), (int)n);
	};
	scatter(d_vec_, (size_t)stride, d_ids, o_v, nd);
	scatter(d_norm2_, 4, d_ids, o_n2, nd);
	scatter(d_adj0_, (size_t)m0 * 4, d_ids, o_a, nd);
	scatter(d_upper_, (size_t)m * sizeof(int2), reinterpret_cast<const int32_t *>(stg + o_li), o_u, lidx.size());
	e = hipGetLastError();
	if (e == hipSuccess) e = hipDeviceSynchronize();
	(void)hipFree(stg);
	if (e != hipSuccess) return fail(e, "patch kernels");
	view.entry = g.entry_point;
	view.max_level = g.max_level;
	view.entry_code = g.present(g.entry_point) ? code_[g.entry_point] : 0;
	built_version = g.version;
	patches++;
	nodes_patched += (long long)nd;
	return 1;
}

// --------------------------------------------------------------------------- no-allocation scratch
// mvdb_search_many_dev_noalloc keeps its visited bitmaps here instead of in hipMalloc'ed memory.
// `scratch` is a host-side stand-in, not a device address: host code only ever uses the per-device
// pointer hipGetSymbolAddress returned (resolve_scratch, with that device current), as the kernel
// argument and for hipMemsetAsync.  Every GPU the library runs on holds its own 64 MiB copy.
constexpr size_t kNoAllocScratchBytes = (size_t)64 << 20;
constexpr int kMaxDevices = 16;
static __device__ __align__(256) unsigned char scratch[kNoAllocScratchBytes];
static void *g_scratch_ptr[kMaxDevices];

// --------------------------------------------------------------------------- handle
constexpr int kStreamSets = 4;
constexpr int kStages = 5;  // 0 prep, 1 traversal, 2 rerank, 3 scan, 4 result-heap replay (+ fall-back launch) of the lean path

struct mvdb_index {
	mvdb_config cfg{};
	HostGraph g;
	DeviceIndex dev;
	int ef_search = 50;  // EF_SEARCH, assembly/config.ts:54
	std::string err;
	int device = 0;

	// f32 store
	float *f32_dev = nullptr;    // resident copy
	float *f32_pinned = nullptr;  // host-pinned copy (device-mapped)
	const float *f32_ptr = nullptr;  // pointer the rerank kernel dereferences
	int64_t f32_rows = 0;

	// scratch
	// Per-stream scratch (kStreamSets streams at most): visited bitmaps and the intermediates of the
	// fused query path, so that batches on different streams can be in flight together; launches on
	// one stream are ordered anyway.  One stream more takes over a set once that set's stream drained.
	struct StreamScratch {
		hipStream_t stream = nullptr;
		bool used = false;
		DevBuf visited, qf, q8, ids, dist, cnt;
		DevBuf adm_log, adm_n, flag;   // traverse_lean_kernel -> replay_kernel
		DevBuf scan_part;              // mvdb_scan_many_dev: partial lists, bounds
	} ss[kStreamSets];
	int ss_next = 0;  // round-robin victim when every set is taken
	// Host slots (mvdb_query_many_submit / _wait): a stream of its own, pinned host staging and device
	// buffers per slot, so a host without device pointers (Node, plain C) keeps several batches in flight.
	struct HostSlot {
		hipStream_t stream = nullptr;
		void *pin_in = nullptr, *pin_out = nullptr;
		size_t pin_in_cap = 0, pin_out_cap = 0;
		DevBuf d_in, d_ids, d_l2, d_n, d_allow;
		int nq = 0, topk = 0;
		bool busy = false;
	} hs[kStreamSets];
	DevBuf st_q8, st_qf, st_ids, st_dist, st_cnt, st_ctr, st_allow, st_out_ids, st_out_l2, st_out_n,
		st_misc, prof_buf;

	// tuning knobs of the timed path (mvdb_set_tuning); 0 = the engine's default
	int traverse_waves = 0;  // 0: lean kernel + replay (traverse_lean.hip.hpp); 1: the one-wave kernel that keeps the result heap in the traversal
	int lean_log_cap = 0;    // test hook (mvdb_debug_set_lean_log_cap): admissions logged per query, 0 = 12 * ef + 1024
	int *lean_stats_host = nullptr;  // device-mapped: queries the lean kernel handed to traverse_kernel, by reason
	int *lean_stats_dev = nullptr;
	DevBuf replay_stats;             // device: {short cuts, full replays} of replay_kernel (mvdb_lean_replays)
	int rerank_waves = 0;    // 2 or 4 waves per rerank workgroup
	// error word the kernels raise (host-pinned, device-mapped): bit 0 = a traversal aborted a query
	int *err_host = nullptr;
	int *err_dev = nullptr;
	int inject_abort_q = -1;  // mvdb_debug_inject_abort
	int inject_build_fail_pass = -1;  // mvdb_debug_inject_build_failure

	bool timing = false;
	std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pairs[kStages];
	float stage_ms[kStages] = {-1.f, -1.f, -1.f, -1.f, -1.f};
	double stage_total_ms[kStages] = {0, 0, 0, 0, 0};
	long long stage_count[kStages] = {0, 0, 0, 0, 0};

	void set_error(const std::string &s) { err = s; }
	~mvdb_index()
	{
		(void)hipFree(f32_dev);
		if (f32_pinned) (void)hipHostFree(f32_pinned);
		if (err_host) (void)hipHostFree(err_host);
		if (lean_stats_host) (void)hipHostFree(lean_stats_host);
		for (auto &s : hs) {
			if (s.stream) {
				(void)hipStreamSynchronize(s.stream);
				(void)hipStreamDestroy(s.stream);
			}
			if (s.pin_in) (void)hipHostFree(s.pin_in);
			if (s.pin_out) (void)hipHostFree(s.pin_out);
		}
		for (auto &v : ev_pairs)
			for (auto &pr : v) {
				(void)hipEventDestroy(pr.first);
				(void)hipEventDestroy(pr.second);
			}
	}
};

namespace {

// WasmBridge.init / reinitFresh (wasm-bridge.ts:52-103, :383-418)
void bridge_reinit(mvdb_index *h)
{
	h->ef_search = 50;
	if (h->cfg.ef_search > 0) h->ef_search = h->cfg.ef_search;
	h->g.results_cap = 1000;
	{
		int cap = h->cfg.results_cap;  // `config.resultsCap ?? 1000` then clampResultsCap :105-110
		if (cap <= 0) cap = 1;
		h->g.results_cap = std::min(cap, kMaxEf);
	}
	h->g.seed(h->cfg.seed);
	GraphConfig gc;
	gc.dim = h->cfg.dim;
	gc.m = h->cfg.m;
	gc.ef_construction = h->cfg.ef_construction;
	h->g.init(gc, h->cfg.capacity);
}

// once per device, with that device current: where this GPU keeps its copy of `scratch`
int resolve_scratch(mvdb_index *h)
{
	const int d = h->device;
	if (d < 0 || d >= kMaxDevices) return 1;  // no scratch pointer there: the no-allocation entry point refuses such a device
	if (g_scratch_ptr[d]) return 1;
	void *p = nullptr;
	HIP_OK(h, hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
	g_scratch_ptr[d] = p;
	return 1;
}

int ensure_device(mvdb_index *h)
{
	HIP_OK(h, hipSetDevice(h->device));
	if (!resolve_scratch(h)) return 0;
	if (!h->err_host) {
		HIP_OK(h, hipHostMalloc(reinterpret_cast<void **>(&h->err_host), 64, hipHostMallocMapped));
		*h->err_host = 0;
		void *dp = nullptr;
		HIP_OK(h, hipHostGetDevicePointer(&dp, h->err_host, 0));
		h->err_dev = static_cast<int *>(dp);
	}
	if (!h->lean_stats_host) {
		HIP_OK(h, hipHostMalloc(reinterpret_cast<void **>(&h->lean_stats_host), 64, hipHostMallocMapped));
		std::memset(h->lean_stats_host, 0, 64);
		void *dp = nullptr;
		HIP_OK(h, hipHostGetDevicePointer(&dp, h->lean_stats_host, 0));
		h->lean_stats_dev = static_cast<int *>(dp);
	}
	if (!h->replay_stats.p) {
		if (!h->replay_stats.need(256)) {
			h->set_error("out of device memory (counters)");
			return 0;
		}
		HIP_OK(h, hipMemset(h->replay_stats.p, 0, 256));
	}
	return 1;
}

// after a synchronisation: did a kernel of this handle raise the error word?  Clears it.
int check_device_errors(mvdb_index *h)
{
	if (!h->err_host) return 1;
	const int e = __atomic_exchange_n(h->err_host, 0, __ATOMIC_ACQ_REL);
	if (e == 0) return 1;
	if (e & 1)
		h->set_error("traversal kernel aborted a query (internal wave protocol failure)");
	else  // mini-vector-db.ts:1120-1124: the file-mode reader throws when the store ends before a candidate's row
		h->set_error("Vector store short read: a candidate's row lies past the end of the float32 store");
	return 0;
}

// hipFuncSetAttribute once per (kernel, device) and LDS high-water mark, not once per launch
int set_max_lds(mvdb_index *h, const void *fn, int (&mark)[16], size_t lds)
{
	const int d = h->device & 15;
	if ((int)lds <= mark[d]) return 1;
	HIP_OK(h, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
	mark[d] = (int)lds;
	return 1;
}

int ensure_uploaded(mvdb_index *h)
{
	if (!ensure_device(h)) return 0;
	if (h->dev.built_version == h->g.version) return 1;
	std::string e;
	const int p = h->dev.patch(h->g, e);  // inserts / upserts since the last upload: only their rows move
	if (p < 0 || (p == 0 && !h->dev.upload(h->g, e))) {
		h->set_error("device upload failed: " + e);
		return 0;
	}
	h->g.clear_dirty();
	return 1;
}

// device time of a stage: one HIP event pair per launch, recorded on the launching stream
// and resolved lazily (no synchronisation inside the timed region)
struct StageTimer {
	mvdb_index *h;
	int stage;
	hipStream_t s;
	hipEvent_t e1 = nullptr;
	StageTimer(mvdb_index *h_, int stage_, hipStream_t s_) : h(h_), stage(stage_), s(s_)
	{
		if (!h->timing) return;
		hipEvent_t e0 = nullptr;
		if (hipEventCreate(&e0) != hipSuccess) return;
		if (hipEventCreate(&e1) != hipSuccess) {
			(void)hipEventDestroy(e0);
			e1 = nullptr;
			return;
		}
		(void)hipEventRecord(e0, s);
		h->ev_pairs[stage].emplace_back(e0, e1);
	}
	~StageTimer()
	{
		if (e1) (void)hipEventRecord(e1, s);
	}
};

void resolve_stage(mvdb_index *h, int stage)
{
	for (auto &pr : h->ev_pairs[stage]) {
		float ms = 0.f;
		if (hipEventSynchronize(pr.second) == hipSuccess &&
		    hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) {
			h->stage_ms[stage] = ms;
			h->stage_total_ms[stage] += ms;
			h->stage_count[stage]++;
		}
		(void)hipEventDestroy(pr.first);
		(void)hipEventDestroy(pr.second);
	}
	h->ev_pairs[stage].clear();
}

template <int J, bool FULL>
int launch_traverse_t(mvdb_index *h, const TraverseArgs &a, size_t lds, hipStream_t s, bool flagged)
{
	if (flagged) {  // the fall-back launch of the lean path: the queries listed in a.fb_list
		static int fmark[16] = {0};
		if (!set_max_lds(h, reinterpret_cast<const void *>(&traverse_flagged_kernel<J, FULL>), fmark, lds)) return 0;
		hipLaunchKernelGGL((traverse_flagged_kernel<J, FULL>), dim3(std::min(a.nq, 32)), dim3(64), lds, s, a);
		HIP_OK(h, hipGetLastError());
		return 1;
	}
	static int mark[16] = {0};
	if (!set_max_lds(h, reinterpret_cast<const void *>(&traverse_kernel<J, FULL>), mark, lds)) return 0;
	hipLaunchKernelGGL((traverse_kernel<J, FULL>), dim3(a.nq), dim3(64), lds, s, a);
	HIP_OK(h, hipGetLastError());
	return 1;
}

template <int J, bool FULL>
int launch_traverse_lean_t(mvdb_index *h, const TraverseArgs &a, size_t lds, hipStream_t s)
{
	static int mark[16] = {0};
	if (!set_max_lds(h, reinterpret_cast<const void *>(&traverse_lean_kernel<J, FULL>), mark, lds)) return 0;
	hipLaunchKernelGGL((traverse_lean_kernel<J, FULL>), dim3(a.nq), dim3(64), lds, s, a);
	HIP_OK(h, hipGetLastError());
	return 1;
}

int launch_traverse_lean(mvdb_index *h, const TraverseArgs &a, hipStream_t s)
{
	const size_t lds = traverse_lean_lds_bytes(a.ef, a.cand_cap);
	const int stride = a.ix.row_stride;
	const int J = (stride + 127) / 128;
	const bool full = stride == J * 128;
#define MVDB_L(JJ)                                                                      \
	if (J <= JJ) {                                                                      \
		if (full && J == JJ) return launch_traverse_lean_t<JJ, true>(h, a, lds, s);     \
		return launch_traverse_lean_t<JJ, false>(h, a, lds, s);                         \
	}
	MVDB_L(1)
	MVDB_L(2)
	MVDB_L(3)
	MVDB_L(4)
	MVDB_L(6)
	MVDB_L(8)
#undef MVDB_L
	h->set_error("dim > 1024 is not supported by the traversal kernel");
	return 0;
}

int launch_replay(mvdb_index *h, const TraverseArgs &a, hipStream_t s)
{
	// the sort needs staging for the popped entries; a launch in which no query can be cut off skips it
	const bool may_sort = !(a.set_only && a.ef <= a.k && a.ef <= a.results_cap);
	const size_t lds = replay_lds_bytes(a.ef, may_sort);
	static int mark[16] = {0};
	static int pmark[16] = {0};
	if (!(a.perm ? set_max_lds(h, reinterpret_cast<const void *>(&replay_kernel<true>), pmark, lds)
	             : set_max_lds(h, reinterpret_cast<const void *>(&replay_kernel<false>), mark, lds)))
		return 0;
	TraverseArgs r = a;
	r.lean_stats = static_cast<int *>(h->replay_stats.p);  // replay_kernel counts in device memory
	if (a.perm) hipLaunchKernelGGL(replay_kernel<true>, dim3(a.nq), dim3(64), lds, s, r, may_sort ? 1 : 0);
	else hipLaunchKernelGGL(replay_kernel<false>, dim3(a.nq), dim3(64), lds, s, r, may_sort ? 1 : 0);
	HIP_OK(h, hipGetLastError());
	return 1;
}

// The lean path (one wave per query, result heap replayed afterwards) whenever a neighbour list fits one 64-lane
// chunk; mvdb_set_tuning(h, 1..3, ..) selects the kernels that keep the result heap inside the traversal.
bool traverse_use_lean(const mvdb_index *h, const TraverseArgs &a)
{
	return a.ix.m0 <= kLeanMaxM0 && h->traverse_waves == 0;
}

// the one-wave kernel with the result heap inside (traverse.hip.hpp): M_MAX0 > 64, mvdb_set_tuning(h, 1, ..), the
// queries the lean kernel hands over, and the no-allocation entry point
int launch_traverse(mvdb_index *h, const TraverseArgs &a, hipStream_t s, bool flagged = false)
{
	const size_t lds = traverse_lds_bytes(a.ef, a.cand_cap);
	const int stride = a.ix.row_stride;
	const int J = (stride + 127) / 128;
	const bool full = stride == J * 128;
#define MVDB_T(JJ)                                                                          \
	if (J <= JJ) {                                                                          \
		if (full && J == JJ) return launch_traverse_t<JJ, true>(h, a, lds, s, flagged);     \
		return launch_traverse_t<JJ, false>(h, a, lds, s, flagged);                         \
	}
	MVDB_T(1)
	MVDB_T(2)
	MVDB_T(3)
	MVDB_T(4)
	MVDB_T(6)
	MVDB_T(8)
#undef MVDB_T
	h->set_error("dim > 1024 is not supported by the traversal kernel");
	return 0;
}

}  // namespace

// --------------------------------------------------------------------------- C ABI
extern "C" {

int mvdb_device_available(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) return 0;
	for (int d = 0; d < n; d++) {  // the code object is gfx950 only
		hipDeviceProp_t prop;
		if (hipGetDeviceProperties(&prop, d) != hipSuccess) continue;
		if (std::strncmp(prop.gcnArchName, "gfx950", 6) == 0) return 1;
	}
	return 0;
}

int mvdb_create(const mvdb_config *cfg, mvdb_index **out)
{
	if (!cfg || !out) {
		g_create_error = "null argument";
		return 0;
	}
	*out = nullptr;
	// update_config validations, assembly/config.ts:95-100 (the WASM traps)
	if (cfg->dim <= 0 || (cfg->dim & 3) != 0 || cfg->m <= 0 || cfg->ef_construction <= 0) {
		g_create_error = "invalid config: dim must be a positive multiple of 4, m and ef_construction positive";
		return 0;
	}
	std::unique_ptr<mvdb_index> h(new mvdb_index());
	h->cfg = *cfg;
	if (cfg->device >= 0) h->device = cfg->device;
	else if (hipGetDevice(&h->device) != hipSuccess) h->device = 0;
	bridge_reinit(h.get());
	*out = h.release();
	return 1;
}

void mvdb_destroy(mvdb_index *h)
{
	if (!h) return;
	(void)hipSetDevice(h->device);
	delete h;
}

const char *mvdb_last_error(const mvdb_index *h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int mvdb_get_max_ef(void) { return kMaxEf; }
void mvdb_set_ef_search(mvdb_index *h, int ef)
{
	if (ef <= 0) return;  // config.ts:128
	h->ef_search = ef;
}
int mvdb_get_ef_search(const mvdb_index *h) { return h->ef_search; }
void mvdb_set_results_cap(mvdb_index *h, int cap)
{
	if (cap <= 0) return;  // hnsw.ts:81
	h->g.results_cap = std::min(cap, kMaxEf);
}
int mvdb_get_results_cap(const mvdb_index *h) { return h->g.results_cap; }
int mvdb_has_node(const mvdb_index *h, int id) { return h->g.present(id) ? 1 : 0; }
int mvdb_get_max_elements(const mvdb_index *h) { return h->g.max_elements; }
int mvdb_get_dim(const mvdb_index *h) { return h->cfg.dim; }
int mvdb_element_count(const mvdb_index *h) { return h->g.element_count; }
int mvdb_was_ef_clamped(const mvdb_index *h) { return h->g.ef_clamped ? 1 : 0; }
void mvdb_clear_ef_clamped(mvdb_index *h) { h->g.ef_clamped = false; }

int mvdb_load_dump(mvdb_index *h, const uint8_t *buf, size_t size)
{
	// WasmBridge.load :421-452: reset + re-apply config, then load_index
	bridge_reinit(h);
	if (!buf || !h->g.load(buf, size)) {
		bridge_reinit(h);  // reinitFresh :463-466
		h->set_error("WASM load_index failed (format mismatch / config mismatch / corrupt dump).");
		return 0;
	}
	return 1;
}
size_t mvdb_dump_size(const mvdb_index *h) { return h->g.dump_size(); }
size_t mvdb_save_dump(const mvdb_index *h, uint8_t *buf) { return h->g.save(buf); }

int mvdb_insert_many(mvdb_index *h, const int32_t *ids, const int8_t *packed, int n)
{
	for (int i = 0; i < n; i++) h->g.insert(ids[i], packed + (size_t)i * h->cfg.dim);
	return 1;
}
int mvdb_update_vector(mvdb_index *h, int id, const int8_t *vec)
{
	h->g.update_vector(id, vec);
	return 1;
}
int mvdb_build_parallel(mvdb_index *h, const int8_t *packed, int n, int threads)
{
	if (h->g.element_count != 0) {
		h->set_error("mvdb_build_parallel needs an empty index");
		return 0;
	}
	h->g.build_parallel(packed, n, threads);
	return 1;
}

}  // extern "C"

namespace {
template <int J, bool FULL>
int launch_build_link_t(mvdb_index *h, const BuildArgs &a, size_t lds)
{
	static int mark[16] = {0};
	if (!set_max_lds(h, reinterpret_cast<const void *>(&build_link_kernel<J, FULL>), mark, lds)) return 0;
	hipLaunchKernelGGL((build_link_kernel<J, FULL>), dim3(a.count), dim3(64), lds, nullptr, a);
	HIP_OK(h, hipGetLastError());
	return 1;
}
template <int J, bool FULL>
int launch_build_apply_t(mvdb_index *h, const BuildArgs &a, int n_targets)
{
	hipLaunchKernelGGL((build_apply_kernel<J, FULL>), dim3(n_targets), dim3(64), 0, nullptr, a, n_targets);
	HIP_OK(h, hipGetLastError());
	return 1;
}
// which = 0: build_link_kernel over a.count new nodes; which = 1: build_apply_kernel over n_targets
int launch_build(mvdb_index *h, const BuildArgs &a, int which, size_t lds, int n_targets)
{
	const int stride = a.ix.row_stride;
	const int J = (stride + 127) / 128;
	const bool full = stride == J * 128;
#define MVDB_B(JJ)                                                                                       \
	if (J <= JJ) {                                                                                       \
		if (full && J == JJ)                                                                             \
			return which == 0 ? launch_build_link_t<JJ, true>(h, a, lds) : launch_build_apply_t<JJ, true>(h, a, n_targets); \
		return which == 0 ? launch_build_link_t<JJ, false>(h, a, lds) : launch_build_apply_t<JJ, false>(h, a, n_targets);   \
	}
	MVDB_B(1)
	MVDB_B(2)
	MVDB_B(3)
	MVDB_B(4)
	MVDB_B(6)
	MVDB_B(8)
#undef MVDB_B
	h->set_error("dim > 1024 is not supported by the GPU builder");
	return 0;
}
}  // namespace

extern "C" {

// Batched GPU build (build_gpu.hip.hpp).  ids 0..n-1 into an empty index; `max_block` caps the block of
// nodes linked per pass (<= 0: 32768).  stats_out (8 doubles, may be NULL): seconds total, seconds of the
// sequential host prefix, distance evaluations, nodes expanded, evictions, reverse-edge removals, passes,
// nodes linked on the host.
// Links ids first .. first + count - 1 (dense ids, everything below `first` present) block by block on the GPU.
// first == 0: a build from scratch (mvdb_build_gpu); first > 0: a bulk insert into the live index (mvdb_append_gpu).
static int gpu_link_range(mvdb_index *h, const int8_t *packed, int first, int count, int max_block, double *stats_out)
{
	using clk = std::chrono::steady_clock;
	if (!ensure_device(h)) return 0;
	HostGraph &g = h->g;
	const int n = first + count;  // nodes once the call is done
	const int dim = g.cfg.dim, m = g.cfg.m, m0 = g.cfg.m0();
	if (m0 > kBuildMaxM0) {
		h->set_error("GPU builder: M_MAX0 > 64 is not supported (use mvdb_build_parallel / mvdb_insert_many)");
		return 0;
	}
	if (max_block <= 0) max_block = 32768;
	// a pass links max(64, min(max_block, linked / 32)) nodes: the request buffers are sized for that floor too
	max_block = std::min(std::max(max_block, 64), std::max(n, 1));
	int ef = g.cfg.ef_construction;
	if (ef > kMaxEf) ef = kMaxEf;
	const int cand_cap = std::max(std::min(2 * ef + 32, kMaxEf), ef);
	const int bloom_log2 = ef <= 400 ? 16 : (ef <= 1200 ? 17 : 18);
	const size_t lds = build_link_lds_bytes(ef, cand_cap, bloom_log2);
	if (lds > 160 * 1024) {  // refused before the host graph is touched
		h->set_error("mvdb_build_gpu: ef_construction too large for the LDS heaps");
		return 0;
	}
	const auto t0 = clk::now();
	const int linked0 = g.bulk_append_begin(packed, count, std::min(n, 2048));
	const double prefix_s = std::chrono::duration<double>(clk::now() - t0).count();
	double st[8] = {0, prefix_s, 0, 0, 0, 0, 0, (double)(linked0 - first)};
	// Everything below can fail (device memory, a copy, a launch): the nodes bulk_append_begin created but did not
	// link are then taken back (HostGraph::bulk_abort), so the index holds exactly the nodes it reports and the
	// caller may insert the rest by any other path.
	auto gpu_part = [&]() -> int {
		h->dev.release();  // the search mirror is stale anyway; its memory is needed at 10 M+
		const int stride = (dim + 15) & ~15;
		const int slots = m0 + (kMaxLayers - 1) * m;
		const size_t n_slots_max = (size_t)max_block * slots;
		DevBuf b_vec, b_level, b_adj0, b_cnt0, b_off, b_upper, b_head, b_req, b_rem, b_next, b_tlist, b_misc;
		int8_t *d_vec = static_cast<int8_t *>(b_vec.need((size_t)n * stride));
		int8_t *d_level = static_cast<int8_t *>(b_level.need((size_t)n));
		int32_t *d_adj0 = static_cast<int32_t *>(b_adj0.need((size_t)n * m0 * 4));
		int32_t *d_cnt0 = static_cast<int32_t *>(b_cnt0.need((size_t)n * 4));
		int64_t *d_off = static_cast<int64_t *>(b_off.need((size_t)n * 8));
		int32_t *d_upper = static_cast<int32_t *>(b_upper.need(std::max<size_t>(g.upper.size() * 4, 256)));
		int32_t *d_head = static_cast<int32_t *>(b_head.need((size_t)n * 4));
		int4 *d_req = static_cast<int4 *>(b_req.need(n_slots_max * 16));
		int4 *d_rem = static_cast<int4 *>(b_rem.need(n_slots_max * 16));
		int32_t *d_next = static_cast<int32_t *>(b_next.need(n_slots_max * 4));
		int32_t *d_tlist = static_cast<int32_t *>(b_tlist.need(n_slots_max * 4));
		unsigned char *d_misc = static_cast<unsigned char *>(b_misc.need(256));
		if (!d_vec || !d_level || !d_adj0 || !d_cnt0 || !d_off || !d_upper || !d_head || !d_req || !d_rem || !d_next ||
		    !d_tlist || !d_misc) {
			h->set_error("out of device memory (GPU builder)");
			return 0;
		}
		if (stride == dim) HIP_OK(h, hipMemcpy(d_vec, g.vec.data(), (size_t)n * dim, hipMemcpyHostToDevice));
		else {
			HIP_OK(h, hipMemset(d_vec, 0, (size_t)n * stride));
			HIP_OK(h, hipMemcpy2D(d_vec, stride, g.vec.data(), dim, dim, n, hipMemcpyHostToDevice));
		}
		HIP_OK(h, hipMemcpy(d_level, g.level.data(), (size_t)n, hipMemcpyHostToDevice));
		HIP_OK(h, hipMemcpy(d_adj0, g.adj0.data(), (size_t)n * m0 * 4, hipMemcpyHostToDevice));
		HIP_OK(h, hipMemcpy(d_cnt0, g.cnt0.data(), (size_t)n * 4, hipMemcpyHostToDevice));
		HIP_OK(h, hipMemcpy(d_off, g.upper_off.data(), (size_t)n * 8, hipMemcpyHostToDevice));
		if (!g.upper.empty()) HIP_OK(h, hipMemcpy(d_upper, g.upper.data(), g.upper.size() * 4, hipMemcpyHostToDevice));
		HIP_OK(h, hipMemset(d_head, 0xff, (size_t)n * 4));
		HIP_OK(h, hipMemset(d_misc, 0, 256));

		BuildArgs a;
		a.ix = DevIndexView{};
		a.ix.vec = d_vec;
		a.ix.dim = dim;
		a.ix.row_stride = stride;
		a.ix.m = m;
		a.ix.m0 = m0;
		a.ix.max_elements = n;
		a.ix.entry = g.entry_point;
		a.ix.max_level = g.max_level;
		a.level = d_level;
		a.adj0 = d_adj0;
		a.cnt0 = d_cnt0;
		a.upper_off = d_off;
		a.upper = d_upper;
		a.ef = ef;
		a.cand_cap = cand_cap;
		a.bloom_log2 = bloom_log2;
		a.slots = slots;
		a.req = d_req;
		a.rem = d_rem;
		a.next = d_next;
		a.head = d_head;
		a.tlist = d_tlist;
		a.tcount = reinterpret_cast<int32_t *>(d_misc);
		a.stats = reinterpret_cast<unsigned long long *>(d_misc + 64);
		int passes = 0;
		for (int linked = linked0; linked < n;) {
			if (passes == h->inject_build_fail_pass) {  // test hook (mvdb_debug_inject_build_failure)
				h->set_error("GPU builder: injected failure");
				return 0;
			}
			const int block = std::min(n - linked, std::max(64, std::min(max_block, linked / 32)));
			a.first = linked;
			a.count = block;
			const int n_slots = block * slots;
			if (!launch_build(h, a, 0, lds, 0)) return 0;
			for (int pass = 0; pass < 2; pass++) {  // 0: reverse adds, 1: removals of the evicted back edges
				HIP_OK(h, hipMemsetAsync(a.tcount, 0, 4, nullptr));
				hipLaunchKernelGGL(build_group_kernel, dim3((n_slots + 255) / 256), dim3(256), 0, nullptr,
				                   pass == 0 ? a.req : a.rem, n_slots, a.next, a.head, a.tlist, a.tcount,
				                   pass == 0 ? a.rem : static_cast<int4 *>(nullptr));
				HIP_OK(h, hipGetLastError());
				int n_targets = 0;
				HIP_OK(h, hipMemcpy(&n_targets, a.tcount, 4, hipMemcpyDeviceToHost));
				if (n_targets <= 0) continue;
				if (pass == 0) {
					if (!launch_build(h, a, 1, 0, n_targets)) return 0;
				} else {
					hipLaunchKernelGGL(build_remove_kernel, dim3(n_targets), dim3(64), 0, nullptr, a, n_targets);
					HIP_OK(h, hipGetLastError());
				}
			}
			linked += block;
			passes++;
		}
		HIP_OK(h, hipDeviceSynchronize());
		HIP_OK(h, hipMemcpy(g.adj0.data(), d_adj0, (size_t)n * m0 * 4, hipMemcpyDeviceToHost));
		HIP_OK(h, hipMemcpy(g.cnt0.data(), d_cnt0, (size_t)n * 4, hipMemcpyDeviceToHost));
		if (!g.upper.empty()) HIP_OK(h, hipMemcpy(g.upper.data(), d_upper, g.upper.size() * 4, hipMemcpyDeviceToHost));
		unsigned long long dst[4] = {0, 0, 0, 0};
		HIP_OK(h, hipMemcpy(dst, a.stats, sizeof(dst), hipMemcpyDeviceToHost));
		st[2] = (double)dst[0];
		st[3] = (double)dst[1];
		st[4] = (double)dst[2];
		st[5] = (double)dst[3];
		st[6] = passes;
		return 1;
	};
	if (linked0 < n && !gpu_part()) {
		(void)hipDeviceSynchronize();
		g.bulk_abort(n);
		return 0;
	}
	g.bulk_end(n);
	st[0] = std::chrono::duration<double>(clk::now() - t0).count();
	if (stats_out) std::memcpy(stats_out, st, sizeof(st));
	return 1;
}

int mvdb_build_gpu(mvdb_index *h, const int8_t *packed, int n, int max_block, double *stats_out)
{
	if (h->g.element_count != 0) {
		h->set_error("mvdb_build_gpu needs an empty index");
		return 0;
	}
	if (!packed || n <= 0) return 1;
	return gpu_link_range(h, packed, 0, std::min(n, h->g.max_elements), max_block, stats_out);
}

int mvdb_append_gpu(mvdb_index *h, const int8_t *packed, int count, int max_block, double *stats_out)
{
	HostGraph &g = h->g;
	if (!packed || count <= 0) return 1;
	const int first = g.element_count;
	if (first + (long long)count > g.max_elements) {
		h->set_error("mvdb_append_gpu: ids beyond the index capacity (max_elements)");
		return 0;
	}
	for (int i = 0; i < g.max_elements; i++)
		if (g.present(i) != (i < first)) {  // the builder addresses nodes by position: ids 0 .. element_count - 1, no gaps
			h->set_error("mvdb_append_gpu needs dense ids (every id below element_count present, none above)");
			return 0;
		}
	for (const auto &o : h->hs)
		if (o.busy) {
			h->set_error("mvdb_append_gpu: batches are in flight on the host slots; wait for them first");
			return 0;
		}
	return gpu_link_range(h, packed, first, count, max_block, stats_out);
}

int mvdb_attach_f32(mvdb_index *h, const float *base, int64_t rows, int resident)
{
	if (!ensure_device(h)) return 0;
	(void)hipFree(h->f32_dev);
	h->f32_dev = nullptr;
	if (h->f32_pinned) (void)hipHostFree(h->f32_pinned);
	h->f32_pinned = nullptr;
	h->f32_ptr = nullptr;
	h->f32_rows = 0;
	if (!base || rows <= 0) return 1;
	const size_t bytes = (size_t)rows * h->cfg.dim * sizeof(float);
	if (resident) {
		HIP_OK(h, hipMalloc(reinterpret_cast<void **>(&h->f32_dev), bytes));
		HIP_OK(h, hipMemcpy(h->f32_dev, base, bytes, hipMemcpyHostToDevice));
		h->f32_ptr = h->f32_dev;
	} else {
		HIP_OK(h, hipHostMalloc(reinterpret_cast<void **>(&h->f32_pinned), bytes, hipHostMallocMapped));
		std::memcpy(h->f32_pinned, base, bytes);
		void *dp = nullptr;
		HIP_OK(h, hipHostGetDevicePointer(&dp, h->f32_pinned, 0));
		h->f32_ptr = static_cast<const float *>(dp);
	}
	h->f32_rows = rows;
	return 1;
}

// ---- stage 0
int mvdb_prepare_queries_dev(mvdb_index *h, const float *d_q_in, int nq, float *d_q_norm_out, int8_t *d_q_i8_out,
                             void *stream)
{
	if (!ensure_device(h)) return 0;
	if (nq <= 0) return 1;
	hipStream_t s = static_cast<hipStream_t>(stream);
	StageTimer t(h, 0, s);
	if (h->cfg.dim > kPrepMaxDim) {
		h->set_error("prepare_queries: dim exceeds the staging row (4096)");
		return 0;
	}
	hipLaunchKernelGGL(prep_kernel, dim3(nq), dim3(64), 0, s, d_q_in, nq, h->cfg.dim, d_q_norm_out, d_q_i8_out);
	HIP_OK(h, hipGetLastError());
	return 1;
}

int mvdb_prepare_queries(mvdb_index *h, const float *q_in, int nq, float *q_norm_out, int8_t *q_i8_out)
{
	if (!ensure_device(h)) return 0;
	if (nq <= 0) return 1;
	const size_t n = (size_t)nq * h->cfg.dim;
	float *d_in = static_cast<float *>(h->st_misc.need(n * 4));
	float *d_qf = static_cast<float *>(h->st_qf.need(n * 4));
	int8_t *d_q8 = static_cast<int8_t *>(h->st_q8.need(n));
	if (!d_in || !d_qf || !d_q8) {
		h->set_error("out of device memory (query staging)");
		return 0;
	}
	HIP_OK(h, hipMemcpy(d_in, q_in, n * 4, hipMemcpyHostToDevice));
	if (!mvdb_prepare_queries_dev(h, d_in, nq, d_qf, d_q8, nullptr)) return 0;
	if (q_norm_out) HIP_OK(h, hipMemcpy(q_norm_out, d_qf, n * 4, hipMemcpyDeviceToHost));
	if (q_i8_out) HIP_OK(h, hipMemcpy(q_i8_out, d_q8, n, hipMemcpyDeviceToHost));
	HIP_OK(h, hipDeviceSynchronize());
	return 1;
}

// ---- stage 1
}  // extern "C"

namespace {
// search(): the effective-ef rule and search_layer's candidate capacity
void traverse_params(mvdb_index *h, int k, TraverseArgs &a, bool set_only)
{
	int ef = h->ef_search;  // hnsw.ts:1006-1011
	if (ef < k) ef = k;
	if (ef > kMaxEf) {
		ef = kMaxEf;
		h->g.ef_clamped = true;
	}
	int cand_cap = ef * 2 + 32;  // hnsw.ts:443-445
	if (cand_cap > kMaxEf) cand_cap = kMaxEf;
	if (cand_cap < ef) cand_cap = ef;
	a.ix = h->dev.perm ? h->dev.view_r : h->dev.view;  // the relabelled copy when there is one (device_index.hpp)
	a.perm = h->dev.perm;
	a.k = k;
	a.ef = ef;
	a.cand_cap = cand_cap;
	a.results_cap = h->g.results_cap;
	a.set_only = set_only ? 1 : 0;
	a.vis_stride = visited_stride_bytes(h->g.max_elements);
	a.adm_log = nullptr;  // lean path only (search_many_dev_impl)
	a.adm_log_cap = 0;
	a.adm_n = a.flag = a.lean_stats = nullptr;
	a.fb_list = a.fb_n = nullptr;
	a.prof = nullptr;
}

mvdb_index::StreamScratch *scratch_for(mvdb_index *h, hipStream_t s)
{
	for (auto &c : h->ss)
		if (c.used && c.stream == s) return &c;
	for (auto &c : h->ss)
		if (!c.used) {
			c.used = true;
			c.stream = s;
			return &c;
		}
	mvdb_index::StreamScratch &c = h->ss[h->ss_next];
	h->ss_next = (h->ss_next + 1) % kStreamSets;
	// the old owner may have been destroyed by the caller already: nothing of it can still be in
	// flight then, so an error here only means the handle is stale
	if (hipStreamSynchronize(c.stream) != hipSuccess) (void)hipGetLastError();
	c.stream = s;
	return &c;
}


// The visited bitmaps of a launch, zeroed past the caches: 125 MB per 1 k queries at 1 M nodes would otherwise sweep the
// L2s (32 MB in all) once per batch and take the rows near the entry point, which every query reads, with them.
__global__ __launch_bounds__(256) void zero_nt_kernel(uint4 *p, size_t n16)
{
	typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
	const v4u_t z = {0u, 0u, 0u, 0u};
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
		__builtin_nontemporal_store(z, reinterpret_cast<v4u_t *>(p) + i);
}

// shared by mvdb_search_many_dev (sorted lists, the bridge contract) and the fused query path
// (set_only: see TraverseArgs)
int search_many_dev_impl(mvdb_index *h, const int8_t *d_q, int nq, int k, int32_t *d_out_ids, float *d_out_dist,
                         int32_t *d_out_count, mvdb_query_counters *d_ctr, void *stream, bool set_only)
{
	if (!ensure_uploaded(h)) return 0;
	if (nq <= 0) return 1;
	hipStream_t s = static_cast<hipStream_t>(stream);
	const HostGraph &g = h->g;
	// search() early-outs, hnsw.ts:961-967 -> count 0 for every query
	if (g.entry_point == -1 || g.max_elements <= 0 || k <= 0 || !g.present(g.entry_point)) {
		HIP_OK(h, hipMemsetAsync(d_out_count, 0, (size_t)nq * 4, s));
		if (d_ctr) HIP_OK(h, hipMemsetAsync(d_ctr, 0, (size_t)nq * sizeof(mvdb_query_counters), s));
		return 1;
	}
	TraverseArgs a;
	traverse_params(h, k, a, set_only);
	// visited sets: bounded scratch per stream, queries are processed in slabs beyond it
	const size_t slab_bytes_max = (size_t)16 << 30;
	int slab = (int)std::min<size_t>((size_t)nq, std::max<size_t>(1, slab_bytes_max / a.vis_stride));
	mvdb_index::StreamScratch *sc = scratch_for(h, s);
	if (!sc) return 0;
	void *vis = sc->visited.need((size_t)slab * a.vis_stride);
	if (!vis) {
		h->set_error("out of device memory (visited sets)");
		return 0;
	}
	const bool lean = traverse_use_lean(h, a);
	// lean path: the traversal logs its admissions, replay_kernel rebuilds the result heap from the log
	a.adm_log = nullptr;
	a.adm_log_cap = 0;
	a.adm_n = a.flag = nullptr;
	a.lean_stats = h->lean_stats_dev;
	a.fb_list = a.fb_n = nullptr;
	if (lean) {
		a.adm_log_cap = h->lean_log_cap > 0 ? h->lean_log_cap : 12 * a.ef + 1024;
		const size_t per_q = (size_t)(a.adm_log_cap + 64) * sizeof(uint2);  // + 64: where masked-off lanes of a log store write
		slab = (int)std::min<size_t>((size_t)slab, std::max<size_t>(1, ((size_t)2 << 30) / per_q));  // <= 2 GiB of logs per stream
		a.adm_log = static_cast<uint2 *>(sc->adm_log.need((size_t)slab * per_q));
		a.adm_n = static_cast<int *>(sc->adm_n.need((size_t)slab * 4));
		a.flag = static_cast<int *>(sc->flag.need(((size_t)2 * slab + 1) * 4));  // flags | list of handed-over queries | its length
		if (!a.adm_log || !a.adm_n || !a.flag) {
			h->set_error("out of device memory (admission logs)");
			return 0;
		}
		a.fb_list = a.flag + slab;
		a.fb_n = a.flag + 2 * slab;
	}
	for (int q0 = 0; q0 < nq; q0 += slab) {
		const int n = std::min(slab, nq - q0);
		const size_t need = (size_t)n * a.vis_stride;
		{  // the visited bitmaps of this launch (vis_stride is a multiple of 256 bytes)
			const size_t n16 = need / 16;
			const unsigned blocks = (unsigned)std::min<size_t>((n16 + 255) / 256, 256 * 16);
			hipLaunchKernelGGL(zero_nt_kernel, dim3(blocks), dim3(256), 0, s, static_cast<uint4 *>(vis), n16);
			HIP_OK(h, hipGetLastError());
		}
		a.q = d_q + (size_t)q0 * g.cfg.dim;
		a.nq = n;
		a.visited = vis;
		a.out_ids = d_out_ids + (size_t)q0 * k;
		a.out_dist = d_out_dist + (size_t)q0 * k;
		a.out_count = d_out_count + q0;
		a.ctr = d_ctr ? d_ctr + q0 : nullptr;
		a.prof = nullptr;
		a.err = h->err_dev;
		a.inject_abort_q = h->inject_abort_q >= q0 && h->inject_abort_q < q0 + n ? h->inject_abort_q - q0 : -1;
#ifdef MVDB_TRAVERSE_PROFILE
		a.prof = static_cast<unsigned long long *>(h->prof_buf.need((size_t)nq * kProfSlots * 8)) + (size_t)q0 * kProfSlots;
#endif
		if (lean) {
			HIP_OK(h, hipMemsetAsync(a.fb_n, 0, 4, s));
			{
				StageTimer t(h, 1, s);  // brackets the traversal kernel only
				if (!launch_traverse_lean(h, a, s)) return 0;
			}
			StageTimer t(h, 4, s);
			// queries the lean kernel gave up on (a distance beyond 16 bits, a full log): the one-wave kernel with the
			// result heap inside, on the bitmaps those queries left clean -- a few workgroups that walk the list
			if (!launch_traverse(h, a, s, true)) return 0;
			if (!launch_replay(h, a, s)) return 0;
		} else {
			StageTimer t(h, 1, s);
			if (!launch_traverse(h, a, s)) return 0;
		}
	}
	return 1;
}
}  // namespace

extern "C" {

int mvdb_search_many_dev(mvdb_index *h, const int8_t *d_q, int nq, int k, int32_t *d_out_ids, float *d_out_dist,
                         int32_t *d_out_count, mvdb_query_counters *d_ctr, void *stream)
{
	return search_many_dev_impl(h, d_q, nq, k, d_out_ids, d_out_dist, d_out_count, d_ctr, stream, false);
}

int mvdb_search_many(mvdb_index *h, const int8_t *q, int nq, int k, int32_t *out_ids, float *out_dist,
                     int32_t *out_count, mvdb_query_counters *ctr)
{
	if (!ensure_uploaded(h)) return 0;
	if (nq <= 0) return 1;
	const int kk = std::max(k, 1);
	const size_t nqd = (size_t)nq * h->cfg.dim, nk = (size_t)nq * kk;
	int8_t *d_q = static_cast<int8_t *>(h->st_q8.need(nqd));
	int32_t *d_ids = static_cast<int32_t *>(h->st_ids.need(nk * 4));
	float *d_dist = static_cast<float *>(h->st_dist.need(nk * 4));
	int32_t *d_cnt = static_cast<int32_t *>(h->st_cnt.need((size_t)nq * 4));
	mvdb_query_counters *d_ctr =
		static_cast<mvdb_query_counters *>(h->st_ctr.need((size_t)nq * sizeof(mvdb_query_counters)));
	if (!d_q || !d_ids || !d_dist || !d_cnt || !d_ctr) {
		h->set_error("out of device memory (search staging)");
		return 0;
	}
	HIP_OK(h, hipMemcpy(d_q, q, nqd, hipMemcpyHostToDevice));
	if (!mvdb_search_many_dev(h, d_q, nq, k, d_ids, d_dist, d_cnt, d_ctr, nullptr)) return 0;
	HIP_OK(h, hipDeviceSynchronize());
	HIP_OK(h, hipMemcpy(out_count, d_cnt, (size_t)nq * 4, hipMemcpyDeviceToHost));
	if (!check_device_errors(h)) return 0;  // a traversal aborted a query (the test hook; no kernel gives up on its own)
	if (k > 0) {
		HIP_OK(h, hipMemcpy(out_ids, d_ids, nk * 4, hipMemcpyDeviceToHost));
		HIP_OK(h, hipMemcpy(out_dist, d_dist, nk * 4, hipMemcpyDeviceToHost));
	}
	if (ctr) HIP_OK(h, hipMemcpy(ctr, d_ctr, (size_t)nq * sizeof(mvdb_query_counters), hipMemcpyDeviceToHost));
	return 1;
}

int mvdb_upload(mvdb_index *h) { return ensure_uploaded(h); }

// The same launch as mvdb_search_many_dev with the visited bitmaps in the library's static device array:
// no hipMalloc/hipFree/event creation on this path.  Everything that allocates (device mirror, error word,
// kernel attributes' first set-up is not an allocation) must have happened in mvdb_upload().
int mvdb_search_many_dev_noalloc(mvdb_index *h, const int8_t *d_q, int nq, int k, int32_t *d_out_ids,
                                 float *d_out_dist, int32_t *d_out_count, mvdb_query_counters *d_ctr, void *stream)
{
	if (nq <= 0) return 1;
	if (!h->err_host || h->dev.built_version != h->g.version) {
		h->set_error("mvdb_search_many_dev_noalloc: index not resident on the device; call mvdb_upload() after the last change");
		return 0;
	}
	const int d = h->device;
	if (d < 0 || d >= kMaxDevices || !g_scratch_ptr[d]) {
		h->set_error("mvdb_search_many_dev_noalloc: no scratch pointer for this device (device ordinal >= 16?)");
		return 0;
	}
	HIP_OK(h, hipSetDevice(d));
	hipStream_t s = static_cast<hipStream_t>(stream);
	const HostGraph &g = h->g;
	if (g.entry_point == -1 || g.max_elements <= 0 || k <= 0 || !g.present(g.entry_point)) {  // hnsw.ts:961-967
		HIP_OK(h, hipMemsetAsync(d_out_count, 0, (size_t)nq * 4, s));
		if (d_ctr) HIP_OK(h, hipMemsetAsync(d_ctr, 0, (size_t)nq * sizeof(mvdb_query_counters), s));
		return 1;
	}
	TraverseArgs a;
	traverse_params(h, k, a, false);
	const size_t need = (size_t)nq * a.vis_stride;
	if (need > kNoAllocScratchBytes) {
		h->set_error("mvdb_search_many_dev_noalloc: the visited sets of nq queries (nq * ceil(max_elements / 32) * 4 bytes, rows "
		             "padded to 256) exceed the 64 MiB static scratch (512 queries at 1 M nodes); use mvdb_search_many_dev");
		return 0;
	}
	void *vis = g_scratch_ptr[d];                // never the array's name: that is not a device address
	HIP_OK(h, hipMemsetAsync(vis, 0, need, s));  // the array keeps the last call's bits: cleared every call
	a.q = d_q;
	a.nq = nq;
	a.visited = vis;
	a.out_ids = d_out_ids;
	a.out_dist = d_out_dist;
	a.out_count = d_out_count;
	a.ctr = d_ctr;
	a.prof = nullptr;
	a.err = h->err_dev;
	a.inject_abort_q = h->inject_abort_q < nq ? h->inject_abort_q : -1;
	return launch_traverse(h, a, s);
}

int mvdb_search(mvdb_index *h, const int8_t *q, int k, int32_t *out_ids, float *out_dist)
{
	int32_t cnt = 0;
	if (!mvdb_search_many(h, q, 1, k, out_ids, out_dist, &cnt, nullptr)) return -1;
	return cnt;
}

// ---- stage 2
int mvdb_rerank_many_dev(mvdb_index *h, const float *d_q_f32, int nq, const int32_t *d_cand,
                         const int32_t *d_cand_count, int stride, int ann_k, const uint32_t *d_allow, int topk,
                         int32_t *d_out_ids, double *d_out_l2, int32_t *d_out_n, void *stream)
{
	if (!ensure_device(h)) return 0;
	if (nq <= 0) return 1;
	hipStream_t s = static_cast<hipStream_t>(stream);
	if (topk <= 0 || stride <= 0 || ann_k <= 0 || !h->f32_ptr) {
		if (!h->f32_ptr && topk > 0) {
			h->set_error("no float32 store attached (mvdb_attach_f32)");
			return 0;
		}
		HIP_OK(h, hipMemsetAsync(d_out_n, 0, (size_t)nq * 4, s));
		return 1;
	}
	RerankArgs a;
	a.q = d_q_f32;
	a.base = h->f32_ptr;
	a.rows = h->f32_rows;
	a.dim = h->cfg.dim;
	a.nq = nq;
	a.cand = d_cand;
	a.cand_count = d_cand_count;
	a.stride = stride;
	a.ann_k = ann_k;
	a.topk = topk;
	a.allow = d_allow;
	a.out_ids = d_out_ids;
	a.out_l2 = d_out_l2;
	a.out_n = d_out_n;
	a.err = h->err_dev;
	a.strict_store = h->f32_pinned != nullptr;  // file mode: a missing row is an error, not a skip
	const int cap = std::min(stride, ann_k);
	// mvdb_set_tuning(h, .., 2): small workgroups that fit beside a running traversal (batches in flight)
	const int waves = h->rerank_waves == 2 ? 2 : kRerankWavesDefault;
	const size_t lds = rerank_lds_bytes(a.dim, cap, waves);
	if (lds > 160 * 1024) {
		h->set_error("rerank: candidate list too long for LDS");
		return 0;
	}
	const void *fn = waves == 2 ? reinterpret_cast<const void *>(&rerank_kernel<2>)
	                            : reinterpret_cast<const void *>(&rerank_kernel<kRerankWavesDefault>);
	static int mark2[16] = {0}, mark4[16] = {0};
	if (!set_max_lds(h, fn, waves == 2 ? mark2 : mark4, lds)) return 0;
	StageTimer t(h, 2, s);
	if (waves == 2) hipLaunchKernelGGL(rerank_kernel<2>, dim3(nq), dim3(128), lds, s, a, cap);
	else hipLaunchKernelGGL(rerank_kernel<kRerankWavesDefault>, dim3(nq), dim3(kRerankWavesDefault * 64), lds, s, a, cap);
	HIP_OK(h, hipGetLastError());
	return 1;
}

int mvdb_rerank_many(mvdb_index *h, const float *q_f32, int nq, const int32_t *cand, const int32_t *cand_count,
                     int stride, int ann_k, const uint32_t *allow, int topk, int32_t *out_ids, double *out_l2,
                     int32_t *out_n)
{
	if (!ensure_device(h)) return 0;
	if (nq <= 0) return 1;
	const size_t nqd = (size_t)nq * h->cfg.dim;
	const int tk = std::max(topk, 1), st = std::max(stride, 1);
	float *d_q = static_cast<float *>(h->st_qf.need(nqd * 4));
	int32_t *d_cand = static_cast<int32_t *>(h->st_ids.need((size_t)nq * st * 4));
	int32_t *d_cnt = static_cast<int32_t *>(h->st_cnt.need((size_t)nq * 4));
	int32_t *d_oi = static_cast<int32_t *>(h->st_out_ids.need((size_t)nq * tk * 4));
	double *d_ol = static_cast<double *>(h->st_out_l2.need((size_t)nq * tk * 8));
	int32_t *d_on = static_cast<int32_t *>(h->st_out_n.need((size_t)nq * 4));
	uint32_t *d_allow = nullptr;
	if (allow) {
		const size_t words = ((size_t)std::max<int64_t>(h->g.max_elements, 1) + 31) / 32;
		d_allow = static_cast<uint32_t *>(h->st_allow.need(words * 4));
		if (!d_allow) {
			h->set_error("out of device memory (allow bitmap)");
			return 0;
		}
		HIP_OK(h, hipMemcpy(d_allow, allow, words * 4, hipMemcpyHostToDevice));
	}
	if (!d_q || !d_cand || !d_cnt || !d_oi || !d_ol || !d_on) {
		h->set_error("out of device memory (rerank staging)");
		return 0;
	}
	HIP_OK(h, hipMemcpy(d_q, q_f32, nqd * 4, hipMemcpyHostToDevice));
	if (stride > 0) HIP_OK(h, hipMemcpy(d_cand, cand, (size_t)nq * stride * 4, hipMemcpyHostToDevice));
	HIP_OK(h, hipMemcpy(d_cnt, cand_count, (size_t)nq * 4, hipMemcpyHostToDevice));
	if (!mvdb_rerank_many_dev(h, d_q, nq, d_cand, d_cnt, stride, ann_k, d_allow, topk, d_oi, d_ol, d_on, nullptr))
		return 0;
	HIP_OK(h, hipDeviceSynchronize());
	if (!check_device_errors(h)) return 0;
	HIP_OK(h, hipMemcpy(out_n, d_on, (size_t)nq * 4, hipMemcpyDeviceToHost));
	if (topk > 0) {
		HIP_OK(h, hipMemcpy(out_ids, d_oi, (size_t)nq * topk * 4, hipMemcpyDeviceToHost));
		HIP_OK(h, hipMemcpy(out_l2, d_ol, (size_t)nq * topk * 8, hipMemcpyDeviceToHost));
	}
	return 1;
}

// ---- fused searchMany
void mvdb_derive_params(int topk_in, int base_ef_search, int rerank_multiplier, int max_ann_k, int *topk,
                        int *ef_search, int *ann_k)
{
	const int tk = std::max(1, topk_in);                     // :568
	const int ef = std::max(base_ef_search, tk * 2);         // :571
	int annK = std::max(tk, tk * rerank_multiplier);         // :574
	annK = std::min(annK, max_ann_k);                        // :575
	annK = std::min(annK, kMaxEf);                           // :576 (wasmMaxEf = 4096)
	if (topk) *topk = tk;
	if (ef_search) *ef_search = ef;
	if (ann_k) *ann_k = annK;
}

void mvdb_ensure_results_cap_at_least(mvdb_index *h, int n)
{
	if (n <= 0) return;  // :1161
	n = std::min(n, kMaxEf);
	const int cur = mvdb_get_results_cap(h);
	if (cur >= n) return;
	int next = cur > 0 ? cur : 1000;
	while (next < n) next *= 2;
	// WasmBridge.setResultsCap -> clampResultsCap (wasm-bridge.ts:105-110, :179-185)
	mvdb_set_results_cap(h, std::min(std::max(next, 1), kMaxEf));
}

double mvdb_to_score(double l2, int mode)
{
	if (mode == 0) return l2;
	const double cosine = 1 - l2 / 2;
	if (mode == 1) return std::fmax(-1.0, std::fmin(1.0, cosine));
	const double sim = 1 - l2 / 4;
	return std::fmax(0.0, std::fmin(1.0, sim));
}

int mvdb_query_many_dev(mvdb_index *h, const float *d_q_f32, int nq, int topk, int ef_search, int ann_k,
                        const uint32_t *d_allow, int32_t *d_out_ids, double *d_out_l2, int32_t *d_out_n,
                        mvdb_query_counters *d_ctr, void *stream)
{
	if (!ensure_uploaded(h)) return 0;
	if (nq <= 0) return 1;
	mvdb_set_ef_search(h, ef_search);                 // :572
	mvdb_ensure_results_cap_at_least(h, ann_k);       // :577
	const size_t nqd = (size_t)nq * h->cfg.dim;
	const int ak = std::max(ann_k, 1);
	mvdb_index::StreamScratch *sc = scratch_for(h, static_cast<hipStream_t>(stream));
	if (!sc) return 0;
	float *d_qf = static_cast<float *>(sc->qf.need(nqd * 4));
	int8_t *d_q8 = static_cast<int8_t *>(sc->q8.need(nqd));
	int32_t *d_ids = static_cast<int32_t *>(sc->ids.need((size_t)nq * ak * 4));
	float *d_dist = static_cast<float *>(sc->dist.need((size_t)nq * ak * 4));
	int32_t *d_cnt = static_cast<int32_t *>(sc->cnt.need((size_t)nq * 4));
	if (!d_qf || !d_q8 || !d_ids || !d_dist || !d_cnt) {
		h->set_error("out of device memory (query pipeline scratch)");
		return 0;
	}
	if (!mvdb_prepare_queries_dev(h, d_q_f32, nq, d_qf, d_q8, stream)) return 0;               // :582-591
	// :593; the rerank below keeps every candidate when none is cut off, so their order is irrelevant
	if (!search_many_dev_impl(h, d_q8, nq, ann_k, d_ids, d_dist, d_cnt, d_ctr, stream, true)) return 0;
	return mvdb_rerank_many_dev(h, d_qf, nq, d_ids, d_cnt, ak, ann_k, d_allow, topk, d_out_ids, d_out_l2, d_out_n,
	                            stream);                                                       // :606-651
}

int mvdb_query_many(mvdb_index *h, const float *q_f32, int nq, int topk, int ef_search, int ann_k,
                    const uint32_t *allow, int32_t *out_ids, double *out_l2, int32_t *out_n,
                    mvdb_query_counters *ctr)
{
	if (!ensure_uploaded(h)) return 0;
	if (nq <= 0) return 1;
	const size_t nqd = (size_t)nq * h->cfg.dim;
	const int tk = std::max(topk, 1);
	float *d_in = static_cast<float *>(h->st_misc.need(nqd * 4));
	int32_t *d_oi = static_cast<int32_t *>(h->st_out_ids.need((size_t)nq * tk * 4));
	double *d_ol = static_cast<double *>(h->st_out_l2.need((size_t)nq * tk * 8));
	int32_t *d_on = static_cast<int32_t *>(h->st_out_n.need((size_t)nq * 4));
	mvdb_query_counters *d_ctr =
		static_cast<mvdb_query_counters *>(h->st_ctr.need((size_t)nq * sizeof(mvdb_query_counters)));
	uint32_t *d_allow = nullptr;
	if (allow) {
		const size_t words = ((size_t)std::max<int64_t>(h->g.max_elements, 1) + 31) / 32;
		d_allow = static_cast<uint32_t *>(h->st_allow.need(words * 4));
		if (!d_allow) {
			h->set_error("out of device memory (allow bitmap)");
			return 0;
		}
		HIP_OK(h, hipMemcpy(d_allow, allow, words * 4, hipMemcpyHostToDevice));
	}
	if (!d_in || !d_oi || !d_ol || !d_on || !d_ctr) {
		h->set_error("out of device memory (query staging)");
		return 0;
	}
	HIP_OK(h, hipMemcpy(d_in, q_f32, nqd * 4, hipMemcpyHostToDevice));
	if (!mvdb_query_many_dev(h, d_in, nq, topk, ef_search, ann_k, d_allow, d_oi, d_ol, d_on, d_ctr, nullptr)) return 0;
	HIP_OK(h, hipDeviceSynchronize());
	if (!check_device_errors(h)) return 0;
	HIP_OK(h, hipMemcpy(out_n, d_on, (size_t)nq * 4, hipMemcpyDeviceToHost));
	if (topk > 0) {
		HIP_OK(h, hipMemcpy(out_ids, d_oi, (size_t)nq * topk * 4, hipMemcpyDeviceToHost));
		HIP_OK(h, hipMemcpy(out_l2, d_ol, (size_t)nq * topk * 8, hipMemcpyDeviceToHost));
	}
	if (ctr) HIP_OK(h, hipMemcpy(ctr, d_ctr, (size_t)nq * sizeof(mvdb_query_counters), hipMemcpyDeviceToHost));
	return 1;
}

// ---- several batches in flight from a host without device pointers
int mvdb_query_many_submit(mvdb_index *h, int slot, const float *q_f32, int nq, int topk, int ef_search, int ann_k,
                           const uint32_t *allow)
{
	if (slot < 0 || slot >= kStreamSets) {
		h->set_error("mvdb_query_many_submit: slot must be 0..3");
		return 0;
	}
	if (nq < 0 || topk < 0) {
		h->set_error("mvdb_query_many_submit: negative nq / topk");
		return 0;
	}
	if (h->dev.built_version != h->g.version)
		for (const auto &o : h->hs)
			if (o.busy) {  // the upload would move device memory under the batches in flight
				h->set_error("mvdb_query_many_submit: the index changed while batches are in flight; wait for every slot first");
				return 0;
			}
	if (!ensure_uploaded(h)) return 0;
	mvdb_index::HostSlot &s = h->hs[slot];
	if (s.busy) {
		h->set_error("mvdb_query_many_submit: slot still holds a batch (call mvdb_query_many_wait first)");
		return 0;
	}
	if (!s.stream) HIP_OK(h, hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
	const size_t nqd = (size_t)nq * h->cfg.dim;
	const int tk = std::max(topk, 1);
	const size_t in_bytes = std::max<size_t>(nqd * 4, 16);
	const size_t out_bytes = (size_t)std::max(nq, 1) * tk * 12 + (size_t)std::max(nq, 1) * 4 + 16;
	if (s.pin_in_cap < in_bytes) {
		if (s.pin_in) HIP_OK(h, hipHostFree(s.pin_in));
		s.pin_in = nullptr;
		s.pin_in_cap = 0;
		HIP_OK(h, hipHostMalloc(&s.pin_in, in_bytes, hipHostMallocMapped));
		s.pin_in_cap = in_bytes;
	}
	if (s.pin_out_cap < out_bytes) {
		if (s.pin_out) HIP_OK(h, hipHostFree(s.pin_out));
		s.pin_out = nullptr;
		s.pin_out_cap = 0;
		HIP_OK(h, hipHostMalloc(&s.pin_out, out_bytes, hipHostMallocMapped));
		s.pin_out_cap = out_bytes;
	}
	s.nq = nq;
	s.topk = topk;
	if (nq == 0) {
		s.busy = true;
		return 1;
	}
	// Zero-copy: the prep kernel reads the queries from the pinned (device-mapped) staging buffer and the rerank
	// kernel writes the results into the pinned output block -- 1.5 MB in, 0.12 MB out per 1 k-query batch over the
	// host link inside the kernels, instead of three DMA operations (and their latencies) around them.
	void *d_in_v = nullptr, *d_out_v = nullptr;
	HIP_OK(h, hipHostGetDevicePointer(&d_in_v, s.pin_in, 0));
	HIP_OK(h, hipHostGetDevicePointer(&d_out_v, s.pin_out, 0));
	unsigned char *po = static_cast<unsigned char *>(d_out_v);
	float *d_in = static_cast<float *>(d_in_v);
	double *d_ol = reinterpret_cast<double *>(po);
	int32_t *d_oi = reinterpret_cast<int32_t *>(po + (size_t)nq * tk * 8);
	int32_t *d_on = reinterpret_cast<int32_t *>(po + (size_t)nq * tk * 12);
	uint32_t *d_allow = nullptr;
	if (allow) {
		const size_t words = ((size_t)std::max<int64_t>(h->g.max_elements, 1) + 31) / 32;
		d_allow = static_cast<uint32_t *>(s.d_allow.need(words * 4));
		if (!d_allow) {
			h->set_error("out of device memory (allow bitmap)");
			return 0;
		}
		// pageable source: the copy is staged by the runtime before the call returns
		HIP_OK(h, hipMemcpyAsync(d_allow, allow, words * 4, hipMemcpyHostToDevice, s.stream));
	}
	std::memcpy(s.pin_in, q_f32, nqd * 4);
	if (!mvdb_query_many_dev(h, d_in, nq, topk, ef_search, ann_k, d_allow, d_oi, d_ol, d_on, nullptr, s.stream)) return 0;
	s.busy = true;
	return 1;
}

int mvdb_query_many_slot_shape(mvdb_index *h, int slot, int *nq, int *topk)
{
	if (slot < 0 || slot >= kStreamSets || !h->hs[slot].busy) return 0;
	if (nq) *nq = h->hs[slot].nq;
	if (topk) *topk = h->hs[slot].topk;
	return 1;
}

int mvdb_query_many_wait(mvdb_index *h, int slot, int32_t *out_ids, double *out_l2, int32_t *out_n)
{
	if (slot < 0 || slot >= kStreamSets) {
		h->set_error("mvdb_query_many_wait: slot must be 0..3");
		return 0;
	}
	mvdb_index::HostSlot &s = h->hs[slot];
	if (!s.busy) {
		h->set_error("mvdb_query_many_wait: nothing was submitted on this slot");
		return 0;
	}
	s.busy = false;
	if (s.nq == 0) return 1;
	HIP_OK(h, hipStreamSynchronize(s.stream));
	if (!check_device_errors(h)) return 0;
	const int tk = std::max(s.topk, 1);
	const unsigned char *po = static_cast<const unsigned char *>(s.pin_out);
	if (s.topk > 0) {
		std::memcpy(out_l2, po, (size_t)s.nq * s.topk * 8);
		std::memcpy(out_ids, po + (size_t)s.nq * tk * 8, (size_t)s.nq * s.topk * 4);
	}
	std::memcpy(out_n, po + (size_t)s.nq * tk * 12, (size_t)s.nq * 4);
	return 1;
}

// ---- int8 distances / scan
int mvdb_dist_many(mvdb_index *h, const int8_t *q, int nq, const int32_t *ids, int per_q, int metric, int32_t *out)
{
	if (!ensure_uploaded(h)) return 0;
	if (nq <= 0 || per_q <= 0) return 1;
	const size_t nqd = (size_t)nq * h->cfg.dim, n = (size_t)nq * per_q;
	int8_t *d_q = static_cast<int8_t *>(h->st_q8.need(nqd));
	int32_t *d_ids = static_cast<int32_t *>(h->st_ids.need(n * 4));
	int32_t *d_out = static_cast<int32_t *>(h->st_dist.need(n * 4));
	if (!d_q || !d_ids || !d_out) {
		h->set_error("out of device memory (dist staging)");
		return 0;
	}
	HIP_OK(h, hipMemcpy(d_q, q, nqd, hipMemcpyHostToDevice));
	HIP_OK(h, hipMemcpy(d_ids, ids, n * 4, hipMemcpyHostToDevice));
	// presence is host state: absent ids are masked to -1 here so the kernel returns INT32_MIN
	{
		std::vector<int32_t> masked(ids, ids + n);
		for (auto &v : masked)
			if (!h->g.present(v)) v = -1;
		HIP_OK(h, hipMemcpy(d_ids, masked.data(), n * 4, hipMemcpyHostToDevice));
	}
	const long long pairs = (long long)n;
	hipLaunchKernelGGL(dist_pairs_kernel, dim3((unsigned)((pairs + 31) / 32)), dim3(256), 0, nullptr, h->dev.view, d_q,
	                   d_ids, per_q, pairs, metric, d_out);
	HIP_OK(h, hipGetLastError());
	HIP_OK(h, hipDeviceSynchronize());
	HIP_OK(h, hipMemcpy(out, d_out, n * 4, hipMemcpyDeviceToHost));
	return 1;
}

int mvdb_scan_many_dev(mvdb_index *h, const int8_t *d_q, int nq, int k, int metric, int32_t *d_out_ids,
                       int32_t *d_out_val, int32_t *d_out_count, void *stream)
{
	if (!ensure_uploaded(h)) return 0;
	if (nq <= 0) return 1;
	hipStream_t s = static_cast<hipStream_t>(stream);
	if (k <= 0 || h->g.max_elements <= 0) {
		HIP_OK(h, hipMemsetAsync(d_out_count, 0, (size_t)nq * 4, s));
		return 1;
	}
	if (k > kScanMaxK) {
		h->set_error("mvdb_scan_many: k > 32 is not supported by the fused scan kernel");
		return 0;
	}
	if (h->dev.view.row_stride > 512) {
		h->set_error("mvdb_scan_many: dim > 512 is not supported by the MFMA scan kernel");
		return 0;
	}
	const int N = h->g.max_elements;
	ScanArgs a;
	a.ix = h->dev.view;
	a.norm2 = h->dev.norm2;
	a.q = d_q;
	a.nq = nq;
	a.k = k;
	a.metric = metric;
	a.wide = (k <= kWideMaxK && nq >= kScanWideMinQ) ? kWideMinLevel : 0;
	a.seed = 1;
	a.nodes_per_split = 0;
	a.level0 = kScanLevel0;
	a.level_ratio = kScanLevelRatio;
	a.wide_ratio = kScanWideRatio;
	a.qtiles = a.s2 = a.nsplit = 0;  // per level (scan_plan)
	{
		ScanLevel plan[kScanMaxLevels];
		scan_plan(N, nq, a.wide, a.seed, a.level0, a.level_ratio, a.wide_ratio, plan, &a.nslots);
	}
	const size_t part_elems = (size_t)nq * a.nslots * k;
	const size_t part_bytes = part_elems * 8 + (((size_t)nq * a.nslots * 4 + 7) & ~(size_t)7);
	unsigned char *part = static_cast<unsigned char *>(scratch_for(h, s)->scan_part.need(part_bytes + (size_t)nq * 8));
	if (!part) {
		h->set_error("out of device memory (scan scratch)");
		return 0;
	}
	a.part_key = reinterpret_cast<unsigned long long *>(part);
	a.part_n = reinterpret_cast<int32_t *>(part + part_elems * 8);
	a.thr_g = reinterpret_cast<unsigned long long *>(part + part_bytes);
	HIP_OK(h, hipMemsetAsync(a.thr_g, 0xff, (size_t)nq * 8, s));
	a.out_ids = d_out_ids;
	a.out_val = d_out_val;
	a.out_count = d_out_count;
	StageTimer t(h, 3, s);
	if (!launch_scan(a, s)) {
		h->set_error("scan launch failed");
		return 0;
	}
	return 1;
}

int mvdb_scan_many(mvdb_index *h, const int8_t *q, int nq, int k, int metric, int32_t *out_ids, int32_t *out_val,
                   int32_t *out_count)
{
	if (!ensure_uploaded(h)) return 0;
	if (nq <= 0) return 1;
	const size_t nqd = (size_t)nq * h->cfg.dim;
	const int kk = std::max(k, 1);
	int8_t *d_q = static_cast<int8_t *>(h->st_q8.need(nqd));
	int32_t *d_ids = static_cast<int32_t *>(h->st_ids.need((size_t)nq * kk * 4));
	int32_t *d_val = static_cast<int32_t *>(h->st_dist.need((size_t)nq * kk * 4));
	int32_t *d_cnt = static_cast<int32_t *>(h->st_cnt.need((size_t)nq * 4));
	if (!d_q || !d_ids || !d_val || !d_cnt) {
		h->set_error("out of device memory (scan staging)");
		return 0;
	}
	HIP_OK(h, hipMemcpy(d_q, q, nqd, hipMemcpyHostToDevice));
	if (!mvdb_scan_many_dev(h, d_q, nq, k, metric, d_ids, d_val, d_cnt, nullptr)) return 0;
	HIP_OK(h, hipDeviceSynchronize());
	HIP_OK(h, hipMemcpy(out_count, d_cnt, (size_t)nq * 4, hipMemcpyDeviceToHost));
	if (k > 0) {
		HIP_OK(h, hipMemcpy(out_ids, d_ids, (size_t)nq * k * 4, hipMemcpyDeviceToHost));
		HIP_OK(h, hipMemcpy(out_val, d_val, (size_t)nq * k * 4, hipMemcpyDeviceToHost));
	}
	return 1;
}

int mvdb_merge_topk_dev(mvdb_index *h, const int32_t *d_ids, const double *d_l2, const int32_t *d_n, int parts, int nq,
                        int topk, int32_t *d_out_ids, double *d_out_l2, int32_t *d_out_n, void *stream)
{
	if (!ensure_device(h)) return 0;
	if (nq <= 0) return 1;
	if (parts > 64 || parts < 1 || topk < 1) {
		h->set_error("merge: parts must be 1..64, topk >= 1");
		return 0;
	}
	hipStream_t s = static_cast<hipStream_t>(stream);
	hipLaunchKernelGGL(merge_topk_kernel, dim3((nq + 63) / 64), dim3(64), 0, s, d_ids, d_l2, d_n, (size_t)nq * topk * 4,
	                   (size_t)nq * topk * 8, (size_t)nq * 4, parts, nq, topk, d_out_ids, d_out_l2, d_out_n);
	HIP_OK(h, hipGetLastError());
	return 1;
}

size_t mvdb_topk_block_bytes(int nq, int topk)
{
	if (nq < 0 || topk < 1) return 0;
	return ((size_t)nq * topk * 12 + (size_t)nq * 4 + 7) & ~(size_t)7;
}

int mvdb_merge_topk_packed_dev(mvdb_index *h, const void *d_blocks, int parts, int nq, int topk, int32_t *d_out_ids,
                               double *d_out_l2, int32_t *d_out_n, void *stream)
{
	if (!ensure_device(h)) return 0;
	if (nq <= 0) return 1;
	if (parts > 64 || parts < 1 || topk < 1) {
		h->set_error("merge: parts must be 1..64, topk >= 1");
		return 0;
	}
	hipStream_t s = static_cast<hipStream_t>(stream);
	const size_t block = mvdb_topk_block_bytes(nq, topk);
	const char *base = static_cast<const char *>(d_blocks);
	const double *l2 = reinterpret_cast<const double *>(base);
	const int32_t *ids = reinterpret_cast<const int32_t *>(base + (size_t)nq * topk * 8);
	const int32_t *n = reinterpret_cast<const int32_t *>(base + (size_t)nq * topk * 12);
	hipLaunchKernelGGL(merge_topk_kernel, dim3((nq + 63) / 64), dim3(64), 0, s, ids, l2, n, block, block, block, parts, nq,
	                   topk, d_out_ids, d_out_l2, d_out_n);
	HIP_OK(h, hipGetLastError());
	return 1;
}

#ifdef MVDB_TRAVERSE_PROFILE
// diagnostic build only: per-query cycle sums of the last traversal launch (nq x 16)
int mvdb_debug_traverse_profile(mvdb_index *h, unsigned long long *out, int nq)
{
	if (!h->prof_buf.p) return 0;
	return hipMemcpy(out, h->prof_buf.p, (size_t)nq * kProfSlots * 8, hipMemcpyDeviceToHost) == hipSuccess;
}
#endif
int mvdb_set_tuning(mvdb_index *h, int traverse_waves, int rerank_waves)
{
	if (traverse_waves < 0 || traverse_waves > 1 || (rerank_waves != 0 && rerank_waves != 2 && rerank_waves != 4)) {
		h->set_error("mvdb_set_tuning: traverse_waves must be 0 or 1, rerank_waves 0, 2 or 4");
		return 0;
	}
	h->traverse_waves = traverse_waves;
	h->rerank_waves = rerank_waves;
	return 1;
}

int mvdb_check_errors(mvdb_index *h) { return check_device_errors(h); }

void mvdb_upload_counters(const mvdb_index *h, long long *out3)
{
	out3[0] = h->dev.full_uploads;
	out3[1] = h->dev.patches;
	out3[2] = h->dev.nodes_patched;
}

void mvdb_set_relabel(mvdb_index *h, int mode)
{
	mode = mode < 0 ? 0 : (mode > 2 ? 2 : mode);
	if (h->dev.relabel_mode == mode) return;
	h->dev.relabel_mode = mode;
	h->dev.built_version = ~0ull;  // the next search uploads again, with or without the relabelled copy
}
int mvdb_relabelled(const mvdb_index *h) { return h->dev.perm != nullptr; }

void mvdb_debug_inject_abort(mvdb_index *h, int query) { h->inject_abort_q = query; }
void mvdb_debug_inject_build_failure(mvdb_index *h, int pass) { h->inject_build_fail_pass = pass; }
void mvdb_debug_set_lean_log_cap(mvdb_index *h, int cap) { h->lean_log_cap = cap > 0 ? cap : 0; }

void mvdb_lean_fallbacks(const mvdb_index *h, long long *out3)
{
	for (int i = 0; i < 3; i++)
		out3[i] = h->lean_stats_host ? (long long)__atomic_load_n(h->lean_stats_host + 1 + i, __ATOMIC_RELAXED) : 0;
}

void mvdb_lean_replays(const mvdb_index *h, long long *out2)
{
	int v[2] = {0, 0};
	if (h->replay_stats.p && hipMemcpy(v, h->replay_stats.p, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) v[0] = v[1] = 0;
	out2[0] = v[0];
	out2[1] = v[1];
}

int mvdb_release_stream(mvdb_index *h, void *stream)
{
	if (!ensure_device(h)) return 0;
	hipStream_t s = static_cast<hipStream_t>(stream);
	for (auto &c : h->ss)
		if (c.used && c.stream == s) {
			HIP_OK(h, hipStreamSynchronize(s));
			c.used = false;
			c.stream = nullptr;
		}
	return 1;
}

void mvdb_enable_timing(mvdb_index *h, int on) { h->timing = on != 0; }
float mvdb_last_stage_ms(mvdb_index *h, int stage)
{
	if (stage < 0 || stage >= kStages) return -1.f;
	resolve_stage(h, stage);
	return h->stage_ms[stage];
}
int mvdb_stage_stats(mvdb_index *h, int stage, long long *launches, double *total_ms, int reset)
{
	if (stage < 0 || stage >= kStages) return 0;
	resolve_stage(h, stage);
	if (launches) *launches = h->stage_count[stage];
	if (total_ms) *total_ms = h->stage_total_ms[stage];
	if (reset) {
		h->stage_count[stage] = 0;
		h->stage_total_ms[stage] = 0;
	}
	return 1;
}

}  // extern "C"
