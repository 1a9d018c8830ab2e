/*
 * mvdb_oracle.c -- CPU ORACLE (test infrastructure, see mvdb_oracle.h header).
 *
 * Plain-C restatement of the reference algorithm.  Reference paths are relative
 * to the upstream tree (assembly/ = AssemblyScript WASM core, src/ = TS host).
 * Written from the reference's behaviour; 64-bit host pointers replace the
 * 32-bit WASM linear-memory offsets so >3.5 M-node indexes can be built.
 *
 * Known, documented divergence: when a graph is corrupt such that a node reached
 * on layer l has level < l, the reference reads past the node block into
 * whatever the bump allocator placed next (hnsw.ts:978-985); the oracle (and the
 * HIP engine) treat such a list as empty instead.
 */
#include "mvdb_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MAX_EF MVO_MAX_EF

/* ------------------------------------------------------------------ */
/* per-thread scratch = the WASM module's global scratch buffers       */
/* (hnsw.ts:22-52, allocated in init_index :361-414)                   */
/* ------------------------------------------------------------------ */
typedef struct mvo_scratch {
	uint32_t *visited; /* u32[max_elements]  hnsw.ts:23 */
	int visited_len;
	uint32_t visited_stamp; /* hnsw.ts:24 */

	int32_t cand_ids[MAX_EF]; /* hnsw.ts:33-34 */
	float cand_dist[MAX_EF];
	int32_t res_ids[MAX_EF]; /* hnsw.ts:37-38 */
	float res_dist[MAX_EF];
	int32_t out_ids[MAX_EF]; /* hnsw.ts:41-42 */
	float out_dist[MAX_EF];
	uint8_t used_mark[MAX_EF]; /* hnsw.ts:45 */
	int32_t sel_ids[MAX_EF];   /* hnsw.ts:48 */
	int sel_count;
	int32_t old_ids[MAX_EF]; /* hnsw.ts:52 */

	int32_t tmp_pop_id; /* hnsw.ts:170-171 */
	float tmp_pop_dist;

	mvo_counters ctr;
} mvo_scratch;

#define ARENA_CHUNK_I32 (16u * 1024u * 1024u) /* 64 MiB chunks, never moved */

struct mvo_index {
	/* assembly/config.ts:16-54 */
	int DIM, MAX_LAYERS, M, M_MAX0, EF_CONSTRUCTION, EF_SEARCH;
	/* WasmBridge.currentConfig (wasm-bridge.ts:28) */
	int cfg_capacity, cfg_ef_search, cfg_results_cap;
	uint32_t cfg_seed;

	/* hnsw.ts:10-27 */
	int entry_point_id, max_level, element_count, max_elements;
	int results_cap;
	int ef_clamped_flag;
	uint32_t rng_state; /* hnsw.ts:55 */

	int32_t **node_ptr; /* node_offsets: NULL = absent (hnsw.ts:108-113) */
	int8_t *vectors;    /* vector_storage i8[cap][DIM] */

	int32_t **chunks; /* bump allocator (assembly/memory.ts:25-54) */
	int n_chunks;
	size_t chunk_used;

	mvo_scratch *main; /* scratch used by the sequential API */

	/* parallel builder only */
	volatile char *locks;
	volatile char alloc_lock, global_lock;
};

/* ------------------------------------------------------------------ */
/* assembly/math.ts                                                    */
/* ------------------------------------------------------------------ */
/* math.ts:9-18 / :48-80: both the scalar and the SIMD body produce the exact
 * integer sum (i16 differences, i32 products, no overflow for DIM <= 2^15). */
int32_t mvo_dist_l2_i8(const int8_t *a, const int8_t *b, int dim)
{
	int32_t sum = 0;
	for (int i = 0; i < dim; i++) {
		int32_t d = (int32_t)a[i] - (int32_t)b[i];
		sum += d * d;
	}
	return sum;
}
/* math.ts:21-29 / :83-109 */
int32_t mvo_dot_i8(const int8_t *a, const int8_t *b, int dim)
{
	int32_t sum = 0;
	for (int i = 0; i < dim; i++) sum += (int32_t)a[i] * (int32_t)b[i];
	return sum;
}
/* math.ts:126-128: <f32> cast of the i32 (f32.convert_i32_s, round-to-nearest-even) */
float mvo_dist_l2_sq(const int8_t *a, const int8_t *b, int dim)
{
	return (float)mvo_dist_l2_i8(a, b, dim);
}
/* math.ts:143-145 */
float mvo_dist_dot(const int8_t *a, const int8_t *b, int dim)
{
	return (float)mvo_dot_i8(a, b, dim);
}

/* ------------------------------------------------------------------ */
/* helpers                                                             */
/* ------------------------------------------------------------------ */
static inline int layer_cap(const mvo_index *ix, int layer) /* hnsw.ts:588-590 */
{
	return layer == 0 ? ix->M_MAX0 : ix->M;
}
/* assembly/types.ts:34-49, in i32 units */
static inline size_t node_size_i32(const mvo_index *ix, int level)
{
	size_t sz = 2 + 1 + (size_t)ix->M_MAX0;
	if (level > 0) sz += (size_t)level * (1 + (size_t)ix->M);
	return sz;
}
static inline const int8_t *vec_ptr(const mvo_index *ix, int id) /* hnsw.ts:114-116 */
{
	return ix->vectors + (size_t)id * (size_t)ix->DIM;
}
/* hnsw.ts:578-585: pointer to [count] of `layer`; NULL where the reference would
 * run off the end of the block (documented divergence, file header). */
static inline int32_t *layer_runner(const mvo_index *ix, int32_t *node, int layer)
{
	if (layer > node[1]) return NULL;
	int32_t *r = node + 2;
	for (int l = 0; l < layer; l++) r += 1 + layer_cap(ix, l);
	return r;
}

static void spin_lock(volatile char *l)
{
	while (__atomic_test_and_set(l, __ATOMIC_ACQUIRE)) {
		while (__atomic_load_n(l, __ATOMIC_RELAXED)) __builtin_ia32_pause();
	}
}
static void spin_unlock(volatile char *l) { __atomic_clear(l, __ATOMIC_RELEASE); }

/* memory.ts:20-54 (alloc): bump allocation; chunked so blocks never move */
static int32_t *arena_alloc(mvo_index *ix, size_t n_i32)
{
	n_i32 = (n_i32 + 3) & ~(size_t)3; /* 16-byte granularity like alloc() */
	if (ix->n_chunks == 0 || ix->chunk_used + n_i32 > ARENA_CHUNK_I32) {
		ix->chunks = (int32_t **)realloc(ix->chunks, sizeof(int32_t *) * (size_t)(ix->n_chunks + 1));
		ix->chunks[ix->n_chunks] = (int32_t *)malloc(sizeof(int32_t) * ARENA_CHUNK_I32);
		if (!ix->chunks[ix->n_chunks]) abort(); /* memory.grow failure -> unreachable() */
		ix->n_chunks++;
		ix->chunk_used = 0;
	}
	int32_t *p = ix->chunks[ix->n_chunks - 1] + ix->chunk_used;
	ix->chunk_used += n_i32;
	return p;
}
static void arena_reset(mvo_index *ix) /* memory.ts:64-66 reset_memory */
{
	for (int i = 0; i < ix->n_chunks; i++) free(ix->chunks[i]);
	free(ix->chunks);
	ix->chunks = NULL;
	ix->n_chunks = 0;
	ix->chunk_used = 0;
}

static mvo_scratch *scratch_new(int max_elements)
{
	mvo_scratch *s = (mvo_scratch *)calloc(1, sizeof(mvo_scratch));
	s->visited_len = max_elements > 0 ? max_elements : 0;
	s->visited = (uint32_t *)calloc((size_t)(s->visited_len > 0 ? s->visited_len : 1), 4);
	s->visited_stamp = 1; /* hnsw.ts:401 */
	s->tmp_pop_id = -1;
	return s;
}
static void scratch_free(mvo_scratch *s)
{
	if (!s) return;
	free(s->visited);
	free(s);
}

/* hnsw.ts:118-132 */
static inline int visited_get(const mvo_scratch *s, int id) { return s->visited[id] == s->visited_stamp; }
static inline void visited_set(mvo_scratch *s, int id) { s->visited[id] = s->visited_stamp; }
static void visited_next_stamp(mvo_scratch *s)
{
	s->visited_stamp++;
	if (s->visited_stamp == 0) {
		s->visited_stamp = 1;
		memset(s->visited, 0, (size_t)s->visited_len * 4);
	}
}

static size_t g_alloc_limit = 0;
size_t mvo_alloc_limit_bytes(void)
{
	if (g_alloc_limit == 0) {
		long pages = sysconf(_SC_PHYS_PAGES), psz = sysconf(_SC_PAGE_SIZE);
		g_alloc_limit = (pages > 0 && psz > 0) ? (size_t)((double)pages * (double)psz * 0.8) : ((size_t)1 << 46);
	}
	return g_alloc_limit;
}
void mvo_set_alloc_limit_bytes(size_t bytes) { g_alloc_limit = bytes; }

/* ------------------------------------------------------------------ */
/* hnsw.ts:361-414 init_index                                          */
/* ------------------------------------------------------------------ */
static void init_index(mvo_index *ix, int capacity)
{
	free(ix->node_ptr);
	free(ix->vectors);
	free((void *)ix->locks);
	ix->node_ptr = NULL;
	ix->vectors = NULL;
	ix->locks = NULL;
	scratch_free(ix->main);
	ix->main = NULL;

	if (capacity <= 0) { /* :362-369 */
		ix->entry_point_id = -1;
		ix->max_level = -1;
		ix->element_count = 0;
		ix->max_elements = 0;
		ix->main = scratch_new(0);
		return;
	}
	ix->max_elements = capacity;
	ix->node_ptr = (int32_t **)calloc((size_t)capacity, sizeof(int32_t *));
	ix->vectors = (int8_t *)calloc((size_t)capacity * (size_t)ix->DIM, 1);
	if (!ix->node_ptr || !ix->vectors) abort();
	if (ix->results_cap <= 0) ix->results_cap = 1000; /* :378-379 */
	if (ix->results_cap > MAX_EF) ix->results_cap = MAX_EF;
	ix->main = scratch_new(capacity);
	ix->entry_point_id = -1; /* :398-402 */
	ix->max_level = -1;
	ix->element_count = 0;
	ix->ef_clamped_flag = 0;
}

/* WasmBridge.init / reinitFresh (wasm-bridge.ts:52-103, :383-418) */
static void bridge_reinit(mvo_index *ix, int capacity)
{
	arena_reset(ix);
	ix->EF_SEARCH = 50; /* module default, config.ts:54 */
	mvo_set_ef_search(ix, ix->cfg_ef_search);
	ix->results_cap = 1000; /* module default, hnsw.ts:18 */
	{
		int cap = ix->cfg_results_cap; /* clampResultsCap :105-110 */
		if (cap <= 0) cap = 1;
		if (cap > MAX_EF) cap = MAX_EF;
		mvo_set_results_cap(ix, cap);
	}
	mvo_seed_rng(ix, ix->cfg_seed);
	init_index(ix, capacity);
}

mvo_index *mvo_create(int dim, int m, int ef_construction, int ef_search,
                      int results_cap, uint32_t seed, int capacity)
{
	/* config.ts:95-100: invalid config traps */
	if (dim <= 0 || (dim & 3) != 0 || m <= 0 || ef_construction <= 0) return NULL;
	if (m > 2048) return NULL; /* oracle limit: neighbour lists are copied to 8 Ki-entry stack buffers */
	mvo_index *ix = (mvo_index *)calloc(1, sizeof(mvo_index));
	ix->DIM = dim;
	ix->M = m;
	ix->M_MAX0 = m * 2; /* config.ts:115 */
	ix->EF_CONSTRUCTION = ef_construction;
	ix->MAX_LAYERS = 4; /* config.ts:27 */
	ix->cfg_capacity = capacity;
	ix->cfg_ef_search = ef_search;
	ix->cfg_results_cap = results_cap;
	ix->cfg_seed = seed;
	bridge_reinit(ix, capacity);
	return ix;
}

void mvo_destroy(mvo_index *ix)
{
	if (!ix) return;
	arena_reset(ix);
	free(ix->node_ptr);
	free(ix->vectors);
	free((void *)ix->locks);
	scratch_free(ix->main);
	free(ix);
}

void mvo_set_ef_search(mvo_index *ix, int ef) /* config.ts:127-130 */
{
	if (ef <= 0) return;
	ix->EF_SEARCH = ef;
}
int mvo_get_ef_search(const mvo_index *ix) { return ix->EF_SEARCH; }
void mvo_set_results_cap(mvo_index *ix, int cap) /* hnsw.ts:78-84 */
{
	if (cap <= 0) return;
	if (cap > MAX_EF) cap = MAX_EF;
	ix->results_cap = cap;
}
int mvo_get_results_cap(const mvo_index *ix) { return ix->results_cap; }
int mvo_get_max_ef(void) { return MAX_EF; }
int mvo_was_ef_clamped(const mvo_index *ix) { return ix->ef_clamped_flag; }
void mvo_clear_ef_clamped(mvo_index *ix) { ix->ef_clamped_flag = 0; }
int mvo_get_max_elements(const mvo_index *ix) { return ix->max_elements; }
int mvo_has_node(const mvo_index *ix, int id)
{
	if (id < 0 || id >= ix->max_elements) return 0;
	return ix->node_ptr[id] != NULL;
}
int mvo_element_count(const mvo_index *ix) { return ix->element_count; }
int mvo_entry_point(const mvo_index *ix) { return ix->entry_point_id; }
int mvo_max_level(const mvo_index *ix) { return ix->max_level; }
int mvo_node_level(const mvo_index *ix, int id)
{
	if (!mvo_has_node(ix, id)) return -1;
	return ix->node_ptr[id][1];
}
void mvo_seed_rng(mvo_index *ix, uint32_t seed) /* hnsw.ts:57-59 */
{
	ix->rng_state = seed != 0 ? seed : 2463534242u;
}

/* hnsw.ts:61-75 */
static inline uint32_t xorshift32(uint32_t *state)
{
	uint32_t x = *state;
	x ^= x << 13;
	x ^= x >> 17;
	x ^= x << 5;
	*state = x;
	return x;
}
static inline double rand01(uint32_t *state)
{
	uint32_t r = xorshift32(state) >> 8;
	return (double)r / (double)0x01000000;
}
/* hnsw.ts:416-420 */
static int get_random_level(mvo_index *ix)
{
	int level = 0;
	while (rand01(&ix->rng_state) < 0.5 && level < ix->MAX_LAYERS - 1) level++;
	return level;
}

/* ------------------------------------------------------------------ */
/* candidate min-heap, hnsw.ts:141-213                                 */
/* ------------------------------------------------------------------ */
static inline void cand_swap(mvo_scratch *s, int i, int j)
{
	int32_t id = s->cand_ids[i];
	s->cand_ids[i] = s->cand_ids[j];
	s->cand_ids[j] = id;
	float d = s->cand_dist[i];
	s->cand_dist[i] = s->cand_dist[j];
	s->cand_dist[j] = d;
}
static int cand_push(mvo_scratch *s, int size, int32_t id, float dist) /* :153-168 */
{
	int i = size;
	s->cand_ids[i] = id;
	s->cand_dist[i] = dist;
	size++;
	while (i > 0) {
		int p = (i - 1) >> 1;
		if (s->cand_dist[p] <= s->cand_dist[i]) break;
		cand_swap(s, i, p);
		i = p;
	}
	return size;
}
static int cand_pop(mvo_scratch *s, int size) /* :173-213 */
{
	if (size <= 0) {
		s->tmp_pop_id = -1;
		s->tmp_pop_dist = 0.0f;
		return 0;
	}
	s->tmp_pop_id = s->cand_ids[0];
	s->tmp_pop_dist = s->cand_dist[0];
	size--;
	if (size == 0) return 0;
	s->cand_ids[0] = s->cand_ids[size];
	s->cand_dist[0] = s->cand_dist[size];
	int i = 0;
	for (;;) {
		int l = (i << 1) + 1;
		if (l >= size) break;
		int r = l + 1;
		int smallest = l;
		if (r < size && s->cand_dist[r] < s->cand_dist[l]) smallest = r;
		if (s->cand_dist[i] <= s->cand_dist[smallest]) break;
		cand_swap(s, i, smallest);
		i = smallest;
	}
	return size;
}

/* ------------------------------------------------------------------ */
/* result max-heap, hnsw.ts:216-325                                    */
/* ------------------------------------------------------------------ */
static inline float res_peek_worst(const mvo_scratch *s, int size) /* :216-219 */
{
	if (size <= 0) return 99999999.0f;
	return s->res_dist[0];
}
static inline void res_swap(mvo_scratch *s, int i, int j)
{
	int32_t id = s->res_ids[i];
	s->res_ids[i] = s->res_ids[j];
	s->res_ids[j] = id;
	float d = s->res_dist[i];
	s->res_dist[i] = s->res_dist[j];
	s->res_dist[j] = d;
}
static int res_push(mvo_scratch *s, int size, int32_t id, float dist) /* :233-248 */
{
	int i = size;
	s->res_ids[i] = id;
	s->res_dist[i] = dist;
	size++;
	while (i > 0) {
		int p = (i - 1) >> 1;
		if (s->res_dist[p] >= s->res_dist[i]) break;
		res_swap(s, i, p);
		i = p;
	}
	return size;
}
static void res_sift_down(mvo_scratch *s, int size) /* shared body of :268-287 and :295-314 */
{
	int i = 0;
	for (;;) {
		int l = (i << 1) + 1;
		if (l >= size) break;
		int r = l + 1;
		int largest = l;
		if (r < size && s->res_dist[r] > s->res_dist[l]) largest = r;
		if (s->res_dist[i] >= s->res_dist[largest]) break;
		res_swap(s, i, largest);
		i = largest;
	}
}
static int res_pop(mvo_scratch *s, int size) /* :250-289 */
{
	if (size <= 0) {
		s->tmp_pop_id = -1;
		s->tmp_pop_dist = 0.0f;
		return 0;
	}
	s->tmp_pop_id = s->res_ids[0];
	s->tmp_pop_dist = s->res_dist[0];
	size--;
	if (size == 0) return 0;
	s->res_ids[0] = s->res_ids[size];
	s->res_dist[0] = s->res_dist[size];
	res_sift_down(s, size);
	return size;
}
static void res_replace_root(mvo_scratch *s, int size, int32_t id, float dist) /* :291-315 */
{
	s->res_ids[0] = id;
	s->res_dist[0] = dist;
	res_sift_down(s, size);
}
static int res_to_sorted_full(mvo_scratch *s, int size) /* :317-325 */
{
	int n = size;
	for (int i = n - 1; i >= 0; i--) {
		size = res_pop(s, size);
		s->out_ids[i] = s->tmp_pop_id;
		s->out_dist[i] = s->tmp_pop_dist;
	}
	return n;
}
static int res_extract_k_smallest_sorted(mvo_scratch *s, int res_size, int k) /* :327-358 */
{
	if (k <= 0 || res_size <= 0) return 0;
	if (k > res_size) k = res_size;
	if (k > MAX_EF) k = MAX_EF;
	memset(s->used_mark, 0, (size_t)res_size);
	for (int out = 0; out < k; out++) {
		int best_idx = -1;
		float best_dist = 99999999.0f;
		for (int i = 0; i < res_size; i++) {
			if (s->used_mark[i]) continue;
			float d = s->res_dist[i];
			if (d < best_dist) {
				best_dist = d;
				best_idx = i;
			}
		}
		if (best_idx < 0) break;
		s->used_mark[best_idx] = 1;
		s->out_ids[out] = s->res_ids[best_idx];
		s->out_dist[out] = best_dist;
	}
	return k;
}

/* ------------------------------------------------------------------ */
/* hnsw.ts:423-523 search_layer.                                       */
/* `par`: parallel-builder mode (neighbour list copied under the node  */
/* lock); 0 for every reference-faithful call.                         */
/* ------------------------------------------------------------------ */
static int search_layer(mvo_index *ix, mvo_scratch *s, const int8_t *query,
                        int entry_node, int layer, int ef, int buildMode, int target, int par)
{
	if (ef <= 0) return 0; /* :431 */
	if (ef > MAX_EF) {     /* :432-435 */
		ef = MAX_EF;
		ix->ef_clamped_flag = 1;
	}
	if (entry_node < 0 || entry_node >= ix->max_elements) return 0; /* :438 */
	if (ix->node_ptr[entry_node] == NULL) return 0;               /* :439 */

	visited_next_stamp(s); /* :441 */

	int cand_cap = ef * 2 + 32; /* :443-445 */
	if (cand_cap > MAX_EF) cand_cap = MAX_EF;
	if (cand_cap < ef) cand_cap = ef;

	int cand_size = 0, res_size = 0;
	float dist0 = mvo_dist_l2_sq(query, vec_ptr(ix, entry_node), ix->DIM); /* :450 */
	s->ctr.n_dist++;
	visited_set(s, entry_node);                                  /* :452 */
	cand_size = cand_push(s, cand_size, entry_node, dist0);      /* :453 */
	res_size = res_push(s, res_size, entry_node, dist0);         /* :454 */

	int cap_this = layer == 0 ? ix->M_MAX0 : ix->M; /* :457 */
	int32_t local_nbrs[1 + 2 * 4096];

	while (cand_size > 0) { /* :459 */
		cand_size = cand_pop(s, cand_size);
		int curr_id = s->tmp_pop_id;
		float curr_dist = s->tmp_pop_dist;
		if (curr_id < 0) continue; /* :464 */
		if (res_size >= ef && curr_dist > res_peek_worst(s, res_size)) break; /* :466 */

		int32_t *curr_node = ix->node_ptr[curr_id]; /* :468-469 */
		if (curr_node == NULL) continue;
		int32_t *runner = layer_runner(ix, curr_node, layer); /* :471-475 */
		if (runner == NULL) continue;
		if (par) {
			spin_lock(&ix->locks[curr_id]);
			memcpy(local_nbrs, runner, sizeof(int32_t) * (size_t)(1 + cap_this));
			spin_unlock(&ix->locks[curr_id]);
			runner = local_nbrs;
		}
		int neighbor_count = runner[0];   /* :477 */
		if (neighbor_count < 0) continue; /* :480 */
		if (neighbor_count > cap_this) neighbor_count = cap_this; /* :481 */
		if (layer == 0) s->ctr.n_l0lists++;
		const int32_t *neighbors = runner + 1;

		for (int i = 0; i < neighbor_count; i++) { /* :485 */
			int neighbor_id = neighbors[i];
			if (neighbor_id < 0) continue;                    /* :488 */
			if (neighbor_id >= ix->max_elements) continue;    /* :489 */
			if (ix->node_ptr[neighbor_id] == NULL) continue;  /* :492 */
			if (visited_get(s, neighbor_id)) continue;        /* :494 */
			visited_set(s, neighbor_id);                      /* :495 */

			float d = mvo_dist_l2_sq(query, vec_ptr(ix, neighbor_id), ix->DIM); /* :497 */
			s->ctr.n_dist++;

			if (res_size < ef || d < res_peek_worst(s, res_size)) { /* :499 */
				if (cand_size < cand_cap) cand_size = cand_push(s, cand_size, neighbor_id, d); /* :500-502 */
				if (res_size < ef) { /* :504-508 */
					res_size = res_push(s, res_size, neighbor_id, d);
				} else if (d < res_peek_worst(s, res_size)) {
					res_replace_root(s, res_size, neighbor_id, d);
				}
			}
		}
	}

	if (!buildMode) return res_to_sorted_full(s, res_size); /* :513-515 */

	int pool = target * 2; /* :517-522 */
	if (pool < target) pool = target;
	if (pool > ef) pool = ef;
	if (pool > res_size) pool = res_size;
	return res_extract_k_smallest_sorted(s, res_size, pool);
}

/* hnsw.ts:526-574 */
static int select_neighbors_heuristic(mvo_index *ix, mvo_scratch *s, const int8_t *query, int found, int target)
{
	(void)query;
	s->sel_count = 0;
	if (found <= 0) return 0;
	if (target <= 0) return 0;
	if (target > MAX_EF) target = MAX_EF;

	for (int i = 0; i < found; i++) {
		int cand_id = s->out_ids[i];
		float d_cq = s->out_dist[i];
		int accept = 1;
		const int8_t *cand_vec = vec_ptr(ix, cand_id);
		for (int j = 0; j < s->sel_count; j++) {
			float d_cs = mvo_dist_l2_sq(cand_vec, vec_ptr(ix, s->sel_ids[j]), ix->DIM);
			if (d_cs < d_cq) {
				accept = 0;
				break;
			}
		}
		if (accept) {
			s->sel_ids[s->sel_count++] = cand_id;
			if (s->sel_count >= target) break;
		}
	}
	if (s->sel_count < target) {
		for (int i = 0; i < found && s->sel_count < target; i++) {
			int cand_id = s->out_ids[i];
			int exists = 0;
			for (int j = 0; j < s->sel_count; j++) {
				if (s->sel_ids[j] == cand_id) {
					exists = 1;
					break;
				}
			}
			if (exists) continue;
			s->sel_ids[s->sel_count++] = cand_id;
		}
	}
	return s->sel_count;
}

/* hnsw.ts:648-678 */
static void remove_connection(mvo_index *ix, int src, int dst, int layer)
{
	int32_t *node = ix->node_ptr[src];
	if (node == NULL) return;
	int32_t *runner = layer_runner(ix, node, layer);
	if (runner == NULL) return;
	int count = runner[0];
	if (count <= 0) return;
	int32_t *nb = runner + 1;
	for (int i = 0; i < count; i++) {
		if (nb[i] != dst) continue;
		int last_idx = count - 1;
		int32_t last_id = nb[last_idx];
		nb[i] = last_id;
		nb[last_idx] = -1;
		runner[0] = count - 1;
		return;
	}
}

/* hnsw.ts:593-645.  Returns the evicted victim (or -1) instead of recursing
 * when `defer_victim` is set (parallel builder: avoids nested node locks). */
static int add_connection(mvo_index *ix, int src, int dst, int layer, int defer_victim)
{
	int32_t *node = ix->node_ptr[src];
	if (node == NULL) return -1;
	int32_t *runner = layer_runner(ix, node, layer);
	if (runner == NULL) return -1;
	int count = runner[0];
	int cap_cur = layer_cap(ix, layer);
	int32_t *nb = runner + 1;

	for (int i = 0; i < count; i++)
		if (nb[i] == dst) return -1; /* :603-605 */

	if (count < cap_cur) { /* :607-609 */
		nb[count] = dst;
		runner[0] = count + 1;
		return -1;
	}
	const int8_t *src_vec = vec_ptr(ix, src);
	int worst_idx = -1;
	float worst_dist = -1.0f;
	for (int i = 0; i < count; i++) { /* :617-631 */
		int n_id = nb[i];
		if (n_id < 0 || n_id >= ix->max_elements) {
			worst_idx = i;
			worst_dist = 99999999.0f;
			break;
		}
		float d = mvo_dist_l2_sq(src_vec, vec_ptr(ix, n_id), ix->DIM);
		if (d > worst_dist) {
			worst_dist = d;
			worst_idx = i;
		}
	}
	float dst_dist = mvo_dist_l2_sq(src_vec, vec_ptr(ix, dst), ix->DIM); /* :633 */
	if (worst_idx >= 0 && dst_dist < worst_dist) {
		int victim = nb[worst_idx];
		nb[worst_idx] = dst;
		if (victim >= 0 && victim < ix->max_elements && victim != dst && victim != src) { /* :640-642 */
			if (defer_victim) return victim;
			remove_connection(ix, victim, src, layer);
		}
	}
	return -1;
}

/* hnsw.ts:681-716 */
static void overwrite_neighbors(mvo_index *ix, mvo_scratch *s, int id, int layer, int picked_count)
{
	int32_t *node = ix->node_ptr[id];
	if (node == NULL) return;
	int32_t *runner = layer_runner(ix, node, layer);
	if (runner == NULL) return;
	int cap_cur = layer_cap(ix, layer);
	int32_t *nb = runner + 1;
	int w = 0;
	for (int i = 0; i < picked_count && w < cap_cur; i++) {
		int nid = s->sel_ids[i];
		if (nid < 0 || nid >= ix->max_elements) continue;
		if (nid == id) continue;
		int exists = 0;
		for (int j = 0; j < w; j++)
			if (nb[j] == nid) {
				exists = 1;
				break;
			}
		if (exists) continue;
		nb[w++] = nid;
	}
	runner[0] = w;
	for (int i = w; i < cap_cur; i++) nb[i] = -1;
}

/* upper-layer greedy walk shared by insert :775-810, update :865-898, search :971-1004 */
static void greedy_descent(mvo_index *ix, mvo_scratch *s, const int8_t *q, int *curr_obj_io,
                           float *curr_dist_io, int from_level, int stop_level, int par, int count_ctr)
{
	int curr_obj = *curr_obj_io;
	float curr_dist = *curr_dist_io;
	int32_t local_nbrs[1 + 4096];
	for (int l = from_level; l > stop_level; l--) {
		int changed = 1;
		while (changed) {
			changed = 0;
			int32_t *c_ptr = ix->node_ptr[curr_obj];
			if (c_ptr == NULL) break;
			int32_t *r_ptr = layer_runner(ix, c_ptr, l);
			int cap_layer = l == 0 ? ix->M_MAX0 : ix->M;
			int count = 0;
			if (r_ptr != NULL) {
				if (par) {
					spin_lock(&ix->locks[curr_obj]);
					memcpy(local_nbrs, r_ptr, sizeof(int32_t) * (size_t)(1 + cap_layer));
					spin_unlock(&ix->locks[curr_obj]);
					r_ptr = local_nbrs;
				}
				count = r_ptr[0];
			}
			if (count < 0) count = 0;
			if (count > cap_layer) count = cap_layer;
			if (count_ctr) s->ctr.n_upperlists++;
			for (int i = 0; i < count; i++) {
				int neighbor = r_ptr[1 + i];
				if (neighbor < 0 || neighbor >= ix->max_elements) continue;
				if (ix->node_ptr[neighbor] == NULL) continue;
				float d = mvo_dist_l2_sq(q, vec_ptr(ix, neighbor), ix->DIM);
				if (count_ctr) s->ctr.n_dist++;
				if (d < curr_dist) {
					curr_dist = d;
					curr_obj = neighbor;
					changed = 1;
				}
			}
		}
	}
	*curr_obj_io = curr_obj;
	*curr_dist_io = curr_dist;
}

static void update_and_reconnect(mvo_index *ix, int id, const int8_t *vec);

/* node block creation, insert :730-747 / load_index :1201-1205 */
static int32_t *make_node(mvo_index *ix, int id, int level)
{
	int32_t *node = arena_alloc(ix, node_size_i32(ix, level));
	node[0] = id;
	node[1] = level;
	int32_t *runner = node + 2;
	for (int l = 0; l <= level; l++) {
		runner[0] = 0;
		int cap = layer_cap(ix, l);
		for (int j = 0; j < cap; j++) runner[1 + j] = -1;
		runner += 1 + cap;
	}
	return node;
}

/* hnsw.ts:719-834 */
void mvo_insert(mvo_index *ix, int id, const int8_t *vec)
{
	if (id < 0 || id >= ix->max_elements) return; /* :720 */
	if (ix->node_ptr[id] != NULL) {               /* :722-725 */
		update_and_reconnect(ix, id, vec);
		return;
	}
	mvo_scratch *s = ix->main;
	int8_t *target_vec = ix->vectors + (size_t)id * (size_t)ix->DIM;
	memcpy(target_vec, vec, (size_t)ix->DIM); /* :728 */

	int level = get_random_level(ix); /* :730 */
	ix->node_ptr[id] = make_node(ix, id, level);

	if (ix->entry_point_id == -1) { /* :749-754 */
		ix->entry_point_id = id;
		ix->max_level = level;
		ix->element_count++;
		return;
	}
	int curr_obj = ix->entry_point_id;
	if (curr_obj < 0 || curr_obj >= ix->max_elements || ix->node_ptr[curr_obj] == NULL) { /* :757-770 */
		ix->entry_point_id = id;
		ix->max_level = level;
		ix->element_count++;
		return;
	}
	float curr_dist = mvo_dist_l2_sq(target_vec, vec_ptr(ix, curr_obj), ix->DIM); /* :772 */
	greedy_descent(ix, s, target_vec, &curr_obj, &curr_dist, ix->max_level, level, 0, 0); /* :775-810 */

	for (int l = (level < ix->max_level ? level : ix->max_level); l >= 0; l--) { /* :812-827 */
		int target = l == 0 ? ix->M_MAX0 : ix->M;
		int found = search_layer(ix, s, target_vec, curr_obj, l, ix->EF_CONSTRUCTION, 1, target, 0);
		int picked = select_neighbors_heuristic(ix, s, target_vec, found, target);
		for (int i = 0; i < picked; i++) {
			int nid = s->sel_ids[i];
			add_connection(ix, id, nid, l, 0);
			add_connection(ix, nid, id, l, 0);
		}
		if (found > 0) curr_obj = s->out_ids[0];
	}
	if (level > ix->max_level) { /* :829-832 */
		ix->max_level = level;
		ix->entry_point_id = id;
	}
	ix->element_count++;
}

/* hnsw.ts:837-841 */
void mvo_update_vector(mvo_index *ix, int id, const int8_t *vec)
{
	if (id < 0 || id >= ix->max_elements) return;
	memcpy(ix->vectors + (size_t)id * (size_t)ix->DIM, vec, (size_t)ix->DIM);
}

/* hnsw.ts:844-953 */
static void update_and_reconnect(mvo_index *ix, int id, const int8_t *vec)
{
	if (id < 0 || id >= ix->max_elements) return;
	int32_t *node = ix->node_ptr[id];
	if (node == NULL) return;
	mvo_scratch *s = ix->main;
	int8_t *target_vec = ix->vectors + (size_t)id * (size_t)ix->DIM;
	memcpy(target_vec, vec, (size_t)ix->DIM);

	int level = node[1];
	if (level < 0) return;
	if (level >= ix->MAX_LAYERS) level = ix->MAX_LAYERS - 1;
	if (ix->entry_point_id == -1) return;
	int curr_obj = ix->entry_point_id;
	if (curr_obj < 0 || curr_obj >= ix->max_elements) return;
	if (ix->node_ptr[curr_obj] == NULL) return;

	float curr_dist = mvo_dist_l2_sq(target_vec, vec_ptr(ix, curr_obj), ix->DIM);
	greedy_descent(ix, s, target_vec, &curr_obj, &curr_dist, ix->max_level, level, 0, 0);

	for (int l = (level < ix->max_level ? level : ix->max_level); l >= 0; l--) {
		int target = l == 0 ? ix->M_MAX0 : ix->M;
		int old_cap = layer_cap(ix, l);
		int old_ids_count = 0;
		{
			int32_t *runner = layer_runner(ix, node, l);
			int c = runner ? runner[0] : 0;
			if (c < 0) c = 0;
			if (c > old_cap) c = old_cap;
			for (int i = 0; i < c && i < MAX_EF; i++) s->old_ids[old_ids_count++] = runner[1 + i];
		}
		int found = search_layer(ix, s, target_vec, curr_obj, l, ix->EF_CONSTRUCTION, 1, target, 0);
		int picked = select_neighbors_heuristic(ix, s, target_vec, found, target);

		for (int i = 0; i < old_ids_count; i++) {
			int oldN = s->old_ids[i];
			if (oldN < 0 || oldN >= ix->max_elements) continue;
			if (oldN == id) continue;
			int keep = 0;
			for (int j = 0; j < picked; j++)
				if (s->sel_ids[j] == oldN) {
					keep = 1;
					break;
				}
			if (!keep) remove_connection(ix, oldN, id, l);
		}
		overwrite_neighbors(ix, s, id, l, picked);
		for (int i = 0; i < picked; i++) {
			int nid = s->sel_ids[i];
			if (nid < 0 || nid >= ix->max_elements) continue;
			if (nid == id) continue;
			add_connection(ix, nid, id, l, 0);
		}
		if (found > 0) curr_obj = s->out_ids[0];
	}
}

/* WasmBridge.insertManyPacked (wasm-bridge.ts:268-288) */
void mvo_insert_many(mvo_index *ix, const int32_t *ids, const int8_t *packed, int n)
{
	for (int i = 0; i < n; i++) mvo_insert(ix, ids[i], packed + (size_t)i * (size_t)ix->DIM);
}

/* ------------------------------------------------------------------ */
/* parallel builder (bench data generation; not reference order)       */
/* ------------------------------------------------------------------ */
static void insert_parallel_one(mvo_index *ix, mvo_scratch *s, int id, int level)
{
	const int8_t *target_vec = vec_ptr(ix, id);
	spin_lock(&ix->global_lock);
	int ep = ix->entry_point_id;
	int maxl = ix->max_level;
	int hold_global = level > maxl; /* this insert will move the entry point */
	if (!hold_global) spin_unlock(&ix->global_lock);

	int curr_obj = ep;
	float curr_dist = mvo_dist_l2_sq(target_vec, vec_ptr(ix, curr_obj), ix->DIM);
	greedy_descent(ix, s, target_vec, &curr_obj, &curr_dist, maxl, level, 1, 0);

	for (int l = (level < maxl ? level : maxl); l >= 0; l--) {
		int target = l == 0 ? ix->M_MAX0 : ix->M;
		int found = search_layer(ix, s, target_vec, curr_obj, l, ix->EF_CONSTRUCTION, 1, target, 1);
		int picked = select_neighbors_heuristic(ix, s, target_vec, found, target);
		for (int i = 0; i < picked; i++) {
			int nid = s->sel_ids[i];
			spin_lock(&ix->locks[id]);
			int v1 = add_connection(ix, id, nid, l, 1);
			spin_unlock(&ix->locks[id]);
			if (v1 >= 0) {
				spin_lock(&ix->locks[v1]);
				remove_connection(ix, v1, id, l);
				spin_unlock(&ix->locks[v1]);
			}
			spin_lock(&ix->locks[nid]);
			int v2 = add_connection(ix, nid, id, l, 1);
			spin_unlock(&ix->locks[nid]);
			if (v2 >= 0) {
				spin_lock(&ix->locks[v2]);
				remove_connection(ix, v2, nid, l);
				spin_unlock(&ix->locks[v2]);
			}
		}
		if (found > 0) curr_obj = s->out_ids[0];
	}
	if (hold_global) {
		ix->max_level = level;
		ix->entry_point_id = id;
		spin_unlock(&ix->global_lock);
	}
	__atomic_fetch_add(&ix->element_count, 1, __ATOMIC_RELAXED);
}

void mvo_build_parallel(mvo_index *ix, const int8_t *packed, int n, int threads)
{
	if (n <= 0) return;
	if (n > ix->max_elements) n = ix->max_elements;
	if (threads < 1) threads = 1;
	memcpy(ix->vectors, packed, (size_t)n * (size_t)ix->DIM);
	int *levels = (int *)malloc(sizeof(int) * (size_t)n);
	for (int i = 0; i < n; i++) levels[i] = get_random_level(ix); /* same RNG stream as sequential */
	for (int i = 0; i < n; i++) ix->node_ptr[i] = make_node(ix, i, levels[i]);
	free((void *)ix->locks);
	ix->locks = (volatile char *)calloc((size_t)ix->max_elements, 1);
	ix->global_lock = 0;

	/* first node sequentially (hnsw.ts:749-754) */
	ix->entry_point_id = 0;
	ix->max_level = levels[0];
	ix->element_count = 1;

	/* a short sequential prefix gives the concurrent phase a connected seed graph */
	int prefix = n < 1024 ? n : 1024;
	for (int i = 1; i < prefix; i++) insert_parallel_one(ix, ix->main, i, levels[i]);

#pragma omp parallel num_threads(threads)
	{
		mvo_scratch *s = scratch_new(ix->max_elements);
#pragma omp for schedule(dynamic, 64)
		for (int i = prefix; i < n; i++) insert_parallel_one(ix, s, i, levels[i]);
		scratch_free(s);
	}
	free(levels);
}

/* ------------------------------------------------------------------ */
/* hnsw.ts:960-1027 search                                             */
/* ------------------------------------------------------------------ */
static int search_impl(mvo_index *ix, mvo_scratch *s, const int8_t *query, int k,
                       int32_t *out_ids, float *out_dist)
{
	if (ix->entry_point_id == -1) return 0; /* :961-963 */
	if (ix->max_elements <= 0) return 0;
	if (k <= 0) return 0;
	int curr_obj = ix->entry_point_id;
	if (curr_obj < 0 || curr_obj >= ix->max_elements) return 0; /* :966 */
	if (ix->node_ptr[curr_obj] == NULL) return 0;               /* :967 */

	float curr_dist = mvo_dist_l2_sq(query, vec_ptr(ix, curr_obj), ix->DIM); /* :969 */
	s->ctr.n_dist++;
	greedy_descent(ix, s, query, &curr_obj, &curr_dist, ix->max_level, 0, 0, 1); /* :971-1004 */

	int ef = ix->EF_SEARCH; /* :1006-1011 */
	if (ef < k) ef = k;
	if (ef > MAX_EF) {
		ef = MAX_EF;
		ix->ef_clamped_flag = 1;
	}
	int found = search_layer(ix, s, query, curr_obj, 0, ef, 0, 0, 0); /* :1013 */

	int out_count = found < k ? found : k; /* :1015-1017 */
	if (out_count > ix->results_cap) out_count = ix->results_cap;
	if (out_count > MAX_EF) out_count = MAX_EF;
	for (int i = 0; i < out_count; i++) { /* :1019-1024 */
		out_ids[i] = s->out_ids[i];
		out_dist[i] = s->out_dist[i];
	}
	return out_count;
}

int mvo_search(mvo_index *ix, const int8_t *query, int k, int32_t *out_ids,
               float *out_dist, mvo_counters *ctr)
{
	memset(&ix->main->ctr, 0, sizeof(mvo_counters));
	int n = search_impl(ix, ix->main, query, k, out_ids, out_dist);
	ix->main->ctr.n_found = n;
	if (ctr) *ctr = ix->main->ctr;
	return n;
}

void mvo_search_many(mvo_index *ix, const int8_t *queries, int nq, int k,
                     int32_t *out_ids, float *out_dist, int32_t *out_count,
                     mvo_counters *ctr, int threads)
{
	int kk = k > 0 ? k : 0;
	if (threads <= 1) {
		for (int qi = 0; qi < nq; qi++) { /* wasm-bridge.ts:333 */
			mvo_counters c;
			out_count[qi] = mvo_search(ix, queries + (size_t)qi * (size_t)ix->DIM, k,
			                           out_ids + (size_t)qi * (size_t)kk, out_dist + (size_t)qi * (size_t)kk, &c);
			if (ctr) ctr[qi] = c;
		}
		return;
	}
#pragma omp parallel num_threads(threads)
	{
		mvo_scratch *s = scratch_new(ix->max_elements);
#pragma omp for schedule(dynamic, 4)
		for (int qi = 0; qi < nq; qi++) {
			memset(&s->ctr, 0, sizeof(mvo_counters));
			int n = search_impl(ix, s, queries + (size_t)qi * (size_t)ix->DIM, k,
			                    out_ids + (size_t)qi * (size_t)kk, out_dist + (size_t)qi * (size_t)kk);
			s->ctr.n_found = n;
			out_count[qi] = n;
			if (ctr) ctr[qi] = s->ctr;
		}
		scratch_free(s);
	}
}

/* ------------------------------------------------------------------ */
/* dump V3, hnsw.ts:1033-1249                                          */
/* ------------------------------------------------------------------ */
#define MAGIC_HNSW 0x57534e48 /* :1033 "HNSW" */
#define DUMP_VERSION_V3 3     /* :1034 */
#define HEADER_BYTES 48       /* :1037-1038 */

static int count_present_nodes(const mvo_index *ix) /* :1040-1046 */
{
	int n = 0;
	for (int i = 0; i < ix->max_elements; i++)
		if (ix->node_ptr[i] != NULL) n++;
	return n;
}

size_t mvo_dump_size(const mvo_index *ix) /* :1048-1073 */
{
	size_t size = HEADER_BYTES;
	for (int i = 0; i < ix->max_elements; i++) {
		const int32_t *node = ix->node_ptr[i];
		if (node == NULL) continue;
		int level = node[1];
		size += 8 + (size_t)ix->DIM;
		for (int l = 0; l <= level; l++) size += 4 + (size_t)layer_cap(ix, l) * 4;
	}
	return size;
}

static inline void put_i32(uint8_t *p, size_t *off, int32_t v)
{
	memcpy(p + *off, &v, 4); /* little-endian host assumed (x86-64) */
	*off += 4;
}
static inline int32_t get_i32(const uint8_t *p, size_t *off)
{
	int32_t v;
	memcpy(&v, p + *off, 4);
	*off += 4;
	return v;
}

size_t mvo_save(const mvo_index *ix, uint8_t *buf) /* :1075-1127 */
{
	size_t off = 0;
	int present = count_present_nodes(ix);
	put_i32(buf, &off, MAGIC_HNSW);
	put_i32(buf, &off, DUMP_VERSION_V3);
	put_i32(buf, &off, ix->DIM);
	put_i32(buf, &off, ix->M);
	put_i32(buf, &off, ix->M_MAX0);
	put_i32(buf, &off, ix->EF_CONSTRUCTION);
	put_i32(buf, &off, ix->MAX_LAYERS);
	put_i32(buf, &off, ix->max_elements);
	put_i32(buf, &off, present);
	put_i32(buf, &off, ix->entry_point_id);
	put_i32(buf, &off, ix->max_level);
	put_i32(buf, &off, ix->results_cap);
	for (int i = 0; i < ix->max_elements; i++) {
		const int32_t *node = ix->node_ptr[i];
		if (node == NULL) continue;
		int id = node[0], level = node[1];
		put_i32(buf, &off, id);
		put_i32(buf, &off, level);
		memcpy(buf + off, vec_ptr(ix, id), (size_t)ix->DIM);
		off += (size_t)ix->DIM;
		const int32_t *runner = node + 2;
		for (int l = 0; l <= level; l++) {
			int cap = layer_cap(ix, l);
			memcpy(buf + off, runner, (size_t)(1 + cap) * 4);
			off += (size_t)(1 + cap) * 4;
			runner += 1 + cap;
		}
	}
	return off;
}

static int load_index(mvo_index *ix, const uint8_t *ptr, size_t size) /* :1129-1249 */
{
	size_t off = 0;
	if (size < HEADER_BYTES) return 0;
	int32_t magic = get_i32(ptr, &off);
	if (magic != MAGIC_HNSW) return 0;
	int32_t ver = get_i32(ptr, &off);
	if (ver != DUMP_VERSION_V3) return 0;
	int32_t dump_dim = get_i32(ptr, &off);
	int32_t dump_m = get_i32(ptr, &off);
	int32_t dump_mmax0 = get_i32(ptr, &off);
	int32_t dump_ef = get_i32(ptr, &off);
	int32_t dump_layers = get_i32(ptr, &off);
	int32_t dump_max_elements = get_i32(ptr, &off);
	int32_t dump_present = get_i32(ptr, &off);
	int32_t dump_entry = get_i32(ptr, &off);
	int32_t dump_max_level = get_i32(ptr, &off);
	int32_t dump_results_cap = get_i32(ptr, &off);

	if (dump_dim != ix->DIM || dump_m != ix->M || dump_mmax0 != ix->M_MAX0 ||
	    dump_ef != ix->EF_CONSTRUCTION || dump_layers != ix->MAX_LAYERS) /* :1151-1159 */
		return 0;
	if (dump_max_elements <= 0) return 0;               /* :1161 */
	if (dump_present < 0) return 0;                     /* :1162 */
	if (dump_present > dump_max_elements) return 0;     /* :1164 */
	if (dump_entry != -1) {                             /* :1167-1169 */
		if (dump_entry < 0 || dump_entry >= dump_max_elements) return 0;
	}
	if (dump_max_level < -1 || dump_max_level >= ix->MAX_LAYERS) return 0; /* :1170 */
	if (dump_present > 0 && dump_entry == -1) return 0;                    /* :1173 */

	/* init_index(dump_max_elements) grows the WASM memory; when memory.grow fails the module traps
	 * (assembly/memory.ts:46-49) and the bridge reports a failed load.  Same outcome here for a capacity
	 * whose arrays do not fit the allocation limit (default 80 % of physical memory). */
	if ((size_t)dump_max_elements * ((size_t)ix->DIM + sizeof(int32_t *) + 4 + 64) > mvo_alloc_limit_bytes()) return 0;
	if (dump_results_cap > 0) { /* :1176-1179 */
		ix->results_cap = dump_results_cap;
		if (ix->results_cap > MAX_EF) ix->results_cap = MAX_EF;
	}
	init_index(ix, dump_max_elements); /* :1181 */
	ix->entry_point_id = dump_entry;
	ix->max_level = dump_max_level;
	int loaded_count = 0;

	for (int n = 0; n < dump_present; n++) { /* :1188-1236 */
		if (off + 8 > size) return 0;
		int32_t id = get_i32(ptr, &off);
		int32_t level = get_i32(ptr, &off);
		if (id < 0 || id >= ix->max_elements) return 0;
		if (level < 0 || level >= ix->MAX_LAYERS) return 0;
		if (off + (size_t)ix->DIM > size) return 0;
		memcpy(ix->vectors + (size_t)id * (size_t)ix->DIM, ptr + off, (size_t)ix->DIM);
		off += (size_t)ix->DIM;

		int32_t *node = arena_alloc(ix, node_size_i32(ix, level));
		node[0] = id;
		node[1] = level;
		int32_t *runner = node + 2;
		for (int l = 0; l <= level; l++) {
			int cap = layer_cap(ix, l);
			if (off + 4 + (size_t)cap * 4 > size) return 0;
			int32_t count = get_i32(ptr, &off);
			if (count < 0 || count > cap) return 0; /* :1215 */
			runner[0] = count;
			memcpy(runner + 1, ptr + off, (size_t)cap * 4);
			off += (size_t)cap * 4;
			for (int i = 0; i < cap; i++) { /* :1224-1229 */
				int32_t nid = runner[1 + i];
				if (nid < 0 || nid >= ix->max_elements) runner[1 + i] = -1;
			}
			runner += 1 + cap;
		}
		ix->node_ptr[id] = node;
		loaded_count++;
	}
	ix->element_count = loaded_count;
	if (ix->entry_point_id != -1 && ix->node_ptr[ix->entry_point_id] == NULL) { /* :1241-1246 */
		ix->entry_point_id = -1;
		ix->max_level = -1;
		ix->element_count = 0;
		return 0;
	}
	return 1;
}

/* WasmBridge.load (wasm-bridge.ts:377-468): reset, re-apply config, load_index,
 * on failure re-init a fresh empty index with the configured capacity. */
int mvo_load(mvo_index *ix, const uint8_t *buf, size_t size)
{
	arena_reset(ix); /* reset_memory :421-423 */
	ix->EF_SEARCH = 50;
	mvo_set_ef_search(ix, ix->cfg_ef_search);
	ix->results_cap = 1000;
	{
		int cap = ix->cfg_results_cap;
		if (cap <= 0) cap = 1;
		if (cap > MAX_EF) cap = MAX_EF;
		mvo_set_results_cap(ix, cap);
	}
	mvo_seed_rng(ix, ix->cfg_seed);
	int ok = load_index(ix, buf, size);
	if (!ok) bridge_reinit(ix, ix->cfg_capacity); /* :463-466 reinitFresh */
	return ok;
}

/* ------------------------------------------------------------------ */
/* src/mini-vector-db.ts host compute                                  */
/* ------------------------------------------------------------------ */
/* :963-970.  JS numbers are f64; Float32Array stores round to f32.  A NaN norm
 * is not `<= 0`, so (as in JS) the scaling loop still runs and fills NaN. */
void mvo_normalize_f32(float *v, int dim)
{
	double normSq = 0;
	for (int i = 0; i < dim; i++) normSq += (double)v[i] * (double)v[i];
	if (normSq <= 0) return;
	double inv = 1.0 / sqrt(normSq);
	for (int i = 0; i < dim; i++) v[i] = (float)((double)v[i] * inv);
}

/* :972-984.  Math.round = floor(x + 0.5) (ties toward +inf); x*127 is exact in
 * f64 for f32 x, and x*127+0.5 is exact for |x*127| <= 127, so floor() matches. */
void mvo_quantize_i8(const float *v, int dim, int8_t *out)
{
	for (int i = 0; i < dim; i++) {
		double x = (double)v[i];
		if (x > 1) x = 1;
		else if (x < -1) x = -1;
		double q = floor(x * 127.0 + 0.5);
		if (q > 127) q = 127;
		else if (q < -127) q = -127;
		/* NaN: Math.round(NaN)=NaN, comparisons false, Int8Array store of NaN = 0 */
		out[i] = (q != q) ? 0 : (int8_t)q;
	}
}
/* :986-993 */
double mvo_l2sq_f32(const float *a, const float *b, int dim)
{
	double s = 0;
	for (int i = 0; i < dim; i++) {
		double d = (double)a[i] - (double)b[i];
		s += d * d;
	}
	return s;
}
/* :995-1001 */
double mvo_to_score(double l2, int mode)
{
	if (mode == 0) return l2;
	double cosine = 1 - l2 / 2;
	if (mode == 1) return fmax(-1, fmin(1, cosine));
	double sim = 1 - l2 / 4;
	return fmax(0, fmin(1, sim));
}
/* searchMany :568-577 */
void mvo_derive_search_params(int topk_in, int base_ef_search, int rerank_mult,
                              int max_ann_k, int wasm_max_ef, int *topk, int *ef_search, int *ann_k)
{
	int tk = topk_in > 1 ? topk_in : 1; /* Math.max(1, topK|0) */
	int ef = base_ef_search > tk * 2 ? base_ef_search : tk * 2;
	int annK = tk > tk * rerank_mult ? tk : tk * rerank_mult;
	if (annK > max_ann_k) annK = max_ann_k;
	if (wasm_max_ef > 0 && annK > wasm_max_ef) annK = wasm_max_ef;
	*topk = tk;
	*ef_search = ef;
	*ann_k = annK;
}
/* ensureResultsCapAtLeast :1160-1170 */
int mvo_next_results_cap(int cur, int n, int wasm_max_ef)
{
	if (n <= 0) return cur;
	if (wasm_max_ef > 0 && n > wasm_max_ef) n = wasm_max_ef;
	if (cur >= n) return cur;
	int next = cur > 0 ? cur : 1000;
	while (next < n) next *= 2;
	return next;
}

typedef struct {
	double l2;
	int32_t id;
} rr_item;
static int rr_cmp(const void *a, const void *b) /* :637 `a.l2 - b.l2 || a.id - b.id` */
{
	const rr_item *x = (const rr_item *)a, *y = (const rr_item *)b;
	if (x->l2 < y->l2) return -1;
	if (x->l2 > y->l2) return 1;
	return (x->id > y->id) - (x->id < y->id);
}

void mvo_rerank_many(const float *q_f32, int nq, int dim, const float *base,
                     int64_t rows, const int32_t *cand, const int32_t *cand_count,
                     int stride, int ann_k, const uint32_t *allow, int topk,
                     int32_t *out_ids, double *out_l2, int32_t *out_n, int threads)
{
	if (threads < 1) threads = 1;
#pragma omp parallel for schedule(dynamic, 4) num_threads(threads)
	for (int qi = 0; qi < nq; qi++) {
		rr_item *items = (rr_item *)malloc(sizeof(rr_item) * (size_t)(stride > 0 ? stride : 1));
		int n = 0, kept = 0;
		const float *q = q_f32 + (size_t)qi * (size_t)dim;
		int cc = cand_count[qi] < stride ? cand_count[qi] : stride;
		for (int j = 0; j < cc; j++) { /* :610-618 */
			int32_t id = cand[(size_t)qi * (size_t)stride + (size_t)j];
			if (allow) {
				if (id < 0) continue;
				if (!((allow[(uint32_t)id >> 5] >> (id & 31)) & 1u)) continue;
			}
			kept++;
			if (id >= 0 && (int64_t)id < rows) { /* :632-634, :1090 */
				items[n].id = id;
				items[n].l2 = mvo_l2sq_f32(q, base + (size_t)id * (size_t)dim, dim);
				n++;
			}
			if (kept >= ann_k) break; /* :617 */
		}
		qsort(items, (size_t)n, sizeof(rr_item), rr_cmp); /* :637 */
		int outn = n < topk ? n : topk;                    /* :640-649 */
		for (int j = 0; j < outn; j++) {
			out_ids[(size_t)qi * (size_t)topk + (size_t)j] = items[j].id;
			out_l2[(size_t)qi * (size_t)topk + (size_t)j] = items[j].l2;
		}
		out_n[qi] = outn;
		free(items);
	}
}

/* ------------------------------------------------------------------ */
/* exact scans (no reference counterpart)                              */
/* ------------------------------------------------------------------ */
typedef struct {
	int64_t key; /* ordering key ascending */
	int32_t id;
	int32_t val;
} sc_item;

void mvo_scan_many(const mvo_index *ix, const int8_t *queries, int nq, int k, int metric,
                   int32_t *out_ids, int32_t *out_dist, int32_t *out_count, int threads)
{
	if (threads < 1) threads = 1;
	if (k < 0) k = 0;
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
	for (int qi = 0; qi < nq; qi++) {
		const int8_t *q = queries + (size_t)qi * (size_t)ix->DIM;
		sc_item *heap = (sc_item *)malloc(sizeof(sc_item) * (size_t)(k > 0 ? k : 1));
		int hs = 0; /* max-heap on (key,id): root = current worst */
		for (int id = 0; id < ix->max_elements && k > 0; id++) {
			if (ix->node_ptr[id] == NULL) continue;
			int32_t val = metric == 0 ? mvo_dist_l2_i8(q, vec_ptr(ix, id), ix->DIM)
			                          : mvo_dot_i8(q, vec_ptr(ix, id), ix->DIM);
			int64_t key = metric == 0 ? (int64_t)val : -(int64_t)val;
			if (hs == k) {
				/* ids ascend, so an equal key never displaces an earlier id */
				if (!(key < heap[0].key)) continue;
				/* replace root, sift down */
				int i = 0;
				sc_item it = {key, id, val};
				for (;;) {
					int l = 2 * i + 1, r = l + 1, big = l;
					if (l >= hs) break;
					if (r < hs && (heap[r].key > heap[l].key ||
					               (heap[r].key == heap[l].key && heap[r].id > heap[l].id)))
						big = r;
					if (it.key > heap[big].key || (it.key == heap[big].key && it.id > heap[big].id)) break;
					heap[i] = heap[big];
					i = big;
				}
				heap[i] = it;
			} else {
				int i = hs++;
				sc_item it = {key, id, val};
				while (i > 0) {
					int p = (i - 1) / 2;
					if (heap[p].key > it.key || (heap[p].key == it.key && heap[p].id > it.id)) break;
					heap[i] = heap[p];
					i = p;
				}
				heap[i] = it;
			}
		}
		/* sort ascending by (key,id): simple insertion sort, k is small */
		for (int i = 1; i < hs; i++) {
			sc_item it = heap[i];
			int j = i - 1;
			while (j >= 0 && (heap[j].key > it.key || (heap[j].key == it.key && heap[j].id > it.id))) {
				heap[j + 1] = heap[j];
				j--;
			}
			heap[j + 1] = it;
		}
		for (int i = 0; i < hs; i++) {
			out_ids[(size_t)qi * (size_t)k + (size_t)i] = heap[i].id;
			out_dist[(size_t)qi * (size_t)k + (size_t)i] = heap[i].val;
		}
		out_count[qi] = hs;
		free(heap);
	}
}

void mvo_exact_f32_topk(const float *q_f32, int nq, int dim, const float *base,
                        int64_t rows, int topk, int32_t *out_ids, double *out_l2, int threads)
{
	if (threads < 1) threads = 1;
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
	for (int qi = 0; qi < nq; qi++) {
		const float *q = q_f32 + (size_t)qi * (size_t)dim;
		rr_item *best = (rr_item *)malloc(sizeof(rr_item) * (size_t)(topk + 1));
		int n = 0;
		for (int64_t id = 0; id < rows; id++) {
			double l2 = mvo_l2sq_f32(q, base + (size_t)id * (size_t)dim, dim);
			if (n == topk && !(l2 < best[n - 1].l2)) continue;
			int j = n < topk ? n++ : n - 1;
			while (j > 0 && best[j - 1].l2 > l2) {
				best[j] = best[j - 1];
				j--;
			}
			best[j].l2 = l2;
			best[j].id = (int32_t)id;
		}
		for (int j = 0; j < topk; j++) {
			out_ids[(size_t)qi * (size_t)topk + (size_t)j] = j < n ? best[j].id : -1;
			out_l2[(size_t)qi * (size_t)topk + (size_t)j] = j < n ? best[j].l2 : 0.0;
		}
		free(best);
	}
}
