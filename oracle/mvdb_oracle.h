/*
 * mvdb_oracle.h -- CPU ORACLE for the MiniVectorDB two-stage retrieval path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the reference's
 * algorithm (AssemblyScript core + the compute parts of the TypeScript host).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
 * it, and only as the checker / the CPU baseline -- never as the product path.
 *
 * PARITY PINNING: the reference's WASM cannot be built or run in this image
 * (no asc/tsc/node_modules) and its tests hold no golden
 * distance vectors or dump fixtures.  The oracle is pinned by (i) the
 * reference tests' known answers re-enacted in tests/test_oracle_known_answers.py
 * (one-hot top-1, self distance 0, distinct one-hots 2*127^2, deleted ids
 * filtered, dump round trip, config-mismatch rejection) and (ii) an independent
 * numpy/pure-Python restatement (oracle/np_oracle.py) that must agree bit for
 * bit.  Everything beyond those known answers is "parity unpinned" w.r.t. the
 * WASM binary itself and rests on line-fidelity to the cited reference lines.
 *
 * Each function cites the reference file:line it follows.
 */
#ifndef MVDB_ORACLE_H
#define MVDB_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MVO_MAX_EF 4096 /* assembly/hnsw.ts:30 */

typedef struct mvo_index mvo_index;

/* per-query work counters (SURVEY.md section 8d byte accounting) */
typedef struct mvo_counters {
	int64_t n_dist;       /* dist_l2_sq evaluations in search()            */
	int64_t n_l0lists;    /* nodes expanded in search_layer(layer 0)        */
	int64_t n_upperlists; /* neighbour-list scans in the upper-layer descent */
	int64_t n_found;      /* results returned                                */
} mvo_counters;

/* ---- index lifecycle: WasmBridge.init (src/core/wasm-bridge.ts:52-103) ---- */
mvo_index *mvo_create(int dim, int m, int ef_construction, int ef_search,
                      int results_cap, uint32_t seed, int capacity);
void mvo_destroy(mvo_index *ix);

/* set_search_config (assembly/config.ts:127-130): ignores ef <= 0 */
void mvo_set_ef_search(mvo_index *ix, int ef_search);
int mvo_get_ef_search(const mvo_index *ix);
/* set_results_cap / get_results_cap (assembly/hnsw.ts:78-88) */
void mvo_set_results_cap(mvo_index *ix, int cap);
int mvo_get_results_cap(const mvo_index *ix);
int mvo_get_max_ef(void);                       /* hnsw.ts:90-92  */
int mvo_was_ef_clamped(const mvo_index *ix);    /* hnsw.ts:94-96  */
void mvo_clear_ef_clamped(mvo_index *ix);       /* hnsw.ts:98-100 */
int mvo_get_max_elements(const mvo_index *ix);  /* hnsw.ts:103-105 */
int mvo_has_node(const mvo_index *ix, int id);  /* hnsw.ts:135-138 */
int mvo_element_count(const mvo_index *ix);
int mvo_entry_point(const mvo_index *ix);
int mvo_max_level(const mvo_index *ix);
int mvo_node_level(const mvo_index *ix, int id); /* -1 when absent */
void mvo_seed_rng(mvo_index *ix, uint32_t seed); /* hnsw.ts:57-59 */

/* ---- int8 kernels: assembly/math.ts ---- */
int32_t mvo_dist_l2_i8(const int8_t *a, const int8_t *b, int dim); /* :115 */
int32_t mvo_dot_i8(const int8_t *a, const int8_t *b, int dim);     /* :133 */
float mvo_dist_l2_sq(const int8_t *a, const int8_t *b, int dim);   /* :126 */
float mvo_dist_dot(const int8_t *a, const int8_t *b, int dim);     /* :143 */

/* ---- write path (data generation only): hnsw.ts:719-953 ---- */
void mvo_insert(mvo_index *ix, int id, const int8_t *vec);        /* :719 upsert */
void mvo_update_vector(mvo_index *ix, int id, const int8_t *vec); /* :837 */
void mvo_insert_many(mvo_index *ix, const int32_t *ids, const int8_t *packed, int n);
/* NOT the reference's sequential order: a lock-based parallel run of the same
 * insert algorithm, used only to produce large benchmark graphs quickly.
 * ids 0..n-1 must be absent; levels are pre-drawn in id order from the same RNG. */
void mvo_build_parallel(mvo_index *ix, const int8_t *packed, int n, int threads);

/* ---- search: hnsw.ts:960-1027 (+ search_layer :423-523) ---- */
int mvo_search(mvo_index *ix, const int8_t *query, int k, int32_t *out_ids,
               float *out_dist, mvo_counters *ctr);
/* WasmBridge.searchMany (wasm-bridge.ts:326-355): sequential loop over packed
 * queries; out_* are nq x k row-major, out_count[nq].  threads>1 runs queries
 * on independent per-thread scratch (results are identical). ctr may be NULL
 * or an array of nq. */
void mvo_search_many(mvo_index *ix, const int8_t *queries, int nq, int k,
                     int32_t *out_ids, float *out_dist, int32_t *out_count,
                     mvo_counters *ctr, int threads);

/* ---- dump V3: hnsw.ts:1033-1249 ---- */
size_t mvo_dump_size(const mvo_index *ix);          /* :1048 */
size_t mvo_save(const mvo_index *ix, uint8_t *buf); /* :1075 */
/* :1129 load_index; returns 1 on success, 0 on any mismatch/corruption.  As in
 * WasmBridge.load (:377-468) a failed load leaves a fresh empty index with the
 * handle's own configured capacity. */
int mvo_load(mvo_index *ix, const uint8_t *buf, size_t size);
/* capacity a dump may ask for, in bytes of index arrays (a larger one fails the load, as memory.grow would trap) */
size_t mvo_alloc_limit_bytes(void);
void mvo_set_alloc_limit_bytes(size_t bytes);

/* ---- host-side compute of MiniVectorDB (src/mini-vector-db.ts) ---- */
void mvo_normalize_f32(float *v, int dim);                        /* :963-970 */
void mvo_quantize_i8(const float *v, int dim, int8_t *out);       /* :972-984 */
double mvo_l2sq_f32(const float *a, const float *b, int dim);     /* :986-993 */
double mvo_to_score(double l2, int mode /*0 l2,1 cosine,2 sim*/); /* :995-1001 */
/* ef / annK derivation, searchMany :568-577 */
void mvo_derive_search_params(int topk_in, int base_ef_search, int rerank_mult,
                              int max_ann_k, int wasm_max_ef, int *topk,
                              int *ef_search, int *ann_k);
/* ensureResultsCapAtLeast :1160-1170 -> next cap value (or cur if unchanged) */
int mvo_next_results_cap(int cur, int n, int wasm_max_ef);

/* rerank + sort + cut, searchMany :626-651 (filtering :606-621 via `allow`).
 * cand: nq x stride ids in ANN order, cand_count[nq].  allow: optional bitmap
 * (bit id set = candidate passes filter/deleted checks), NULL = all pass.
 * base: f32 store (vectors.f32.bin layout, row id at id*dim), rows = number of
 * rows available (ids >= rows are skipped like :1090).
 * out_ids/out_l2: nq x topk, out_n[nq]. */
void mvo_rerank_many(const float *q_f32, int nq, int dim, const float *base,
                     int64_t rows, const int32_t *cand, const int32_t *cand_count,
                     int stride, int ann_k, const uint32_t *allow, int topk,
                     int32_t *out_ids, double *out_l2, int32_t *out_n, int threads);

/* exact int8 brute-force scan (no reference counterpart: north_star kernel #2).
 * Order: (dist asc, id asc).  present: only nodes present in ix are scanned. */
void mvo_scan_many(const mvo_index *ix, const int8_t *queries, int nq, int k,
                   int metric /*0 l2, 1 dot(desc)*/, int32_t *out_ids,
                   int32_t *out_dist, int32_t *out_count, int threads);

/* exact float32 brute force top-k by (l2 f64 asc, id asc): recall ground truth */
void mvo_exact_f32_topk(const float *q_f32, int nq, int dim, const float *base,
                        int64_t rows, int topk, int32_t *out_ids, double *out_l2,
                        int threads);

#ifdef __cplusplus
}
#endif
#endif
