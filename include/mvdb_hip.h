/*
 * mvdb_hip.h -- C ABI of libmvdb_hip.so, the MI355X (gfx950) query engine for
 * MiniVectorDB's two-stage retrieval path (search()/searchMany()).
 *
 * This is the drop-in boundary: every entry point replaces one function the
 * reference host (`class WasmBridge`, src/core/wasm-bridge.ts:24) calls on the
 * WASM module (export list assembly/index.ts:33-67), or one compute loop of
 * `MiniVectorDB.searchMany` (src/mini-vector-db.ts:560-655).  Reference
 * file:line citations are relative to the upstream tree.
 *
 * Conventions
 *  - plain C types only; all buffers are caller-owned; no pointer outlives a call.
 *  - functions returning `int` status: 1 = ok, 0 = failure (mirrors
 *    load_index, assembly/hnsw.ts:1129); mvdb_last_error() gives the text.
 *  - "host" entry points take host pointers and are synchronous.
 *  - "_dev" entry points take DEVICE pointers (HBM) plus a hipStream_t passed as
 *    void* (NULL = default stream) and are asynchronous on that stream; they are
 *    what a resident pipeline (and bench.py) uses.
 *  - host entry points: one call in flight per handle (the reference serialises
 *    every bridge call behind a promise-chain mutex, wasm-bridge.ts:39-50).
 *  - `_dev` entry points are issued from ONE host thread at a time, but their
 *    work may overlap on the device: mvdb_query_many_dev, mvdb_search_many_dev,
 *    mvdb_prepare_queries_dev, mvdb_rerank_many_dev, mvdb_merge_topk_dev and
 *    mvdb_scan_many_dev keep their intermediates per stream (up to 4 streams; a
 *    fifth stream takes over the least recently assigned set after that set's
 *    stream has drained; mvdb_release_stream gives a set back before its stream
 *    is destroyed).
 *  - a traversal that aborts a query (only the test hook mvdb_debug_inject_abort does)
 *    reports out_count = -1 / out_n = -1 for it and raises the handle's error
 *    word: host entry points then return 0; `_dev` callers call
 *    mvdb_check_errors() after synchronising.
 *  - there is NO CPU fallback: without a HIP device every compute call fails.
 */
#ifndef MVDB_HIP_H
#define MVDB_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MVDB_MAX_EF 4096 /* assembly/hnsw.ts:30 MAX_EF */

typedef struct mvdb_index mvdb_index; /* opaque */

/* WasmBridge InitConfig (wasm-bridge.ts:14-22) + the HIP device ordinal */
typedef struct mvdb_config {
	int32_t dim;             /* DIM, multiple of 4 (assembly/config.ts:98)    */
	int32_t m;               /* M; M_MAX0 = 2*M (config.ts:115)               */
	int32_t ef_construction; /* EF_CONSTRUCTION: part of the dump compat key  */
	int32_t ef_search;       /* initial EF_SEARCH (<=0 -> 50, wasm-bridge:80) */
	int32_t capacity;        /* init_index(capacity)                          */
	int32_t results_cap;     /* <=0 -> 1000 (wasm-bridge.ts:84)               */
	uint32_t seed;           /* seed_rng; 0 -> 2463534242 (hnsw.ts:57-59)     */
	int32_t device;          /* HIP device ordinal; -1 = current device       */
} mvdb_config;

/* per-query work counters, SURVEY.md section 8(d) byte accounting */
typedef struct mvdb_query_counters {
	int32_t n_dist;       /* dist_l2_sq evaluations (hnsw.ts:450,497,969,996) */
	int32_t n_l0lists;    /* layer-0 nodes expanded (hnsw.ts:468-483)         */
	int32_t n_upperlists; /* upper-layer list scans (hnsw.ts:975-989)         */
	int32_t n_found;      /* results returned                                 */
} mvdb_query_counters;

/* ---- lifecycle ----------------------------------------------------------- */
/* WasmBridge.init (wasm-bridge.ts:52-103): set_config, set_search_config,
 * set_results_cap, seed_rng, init_index(capacity).  Invalid dim/m/ef fails
 * (update_config traps, assembly/config.ts:95-100). */
int mvdb_create(const mvdb_config *cfg, mvdb_index **out);
void mvdb_destroy(mvdb_index *h);
/* text of the last failure on this handle (or of mvdb_create when h == NULL) */
const char *mvdb_last_error(const mvdb_index *h);
/* 1 when a HIP device whose gcnArchName starts with "gfx950" is visible to this process */
int mvdb_device_available(void);

/* ---- index state (WasmBridge getters/setters) ---------------------------- */
int mvdb_get_max_ef(void);                        /* get_max_ef, hnsw.ts:90       */
void mvdb_set_ef_search(mvdb_index *h, int ef);   /* set_search_config, config.ts:127; ef<=0 ignored */
int mvdb_get_ef_search(const mvdb_index *h);
void mvdb_set_results_cap(mvdb_index *h, int cap); /* set_results_cap, hnsw.ts:78; cap<=0 ignored, clamp 4096 */
int mvdb_get_results_cap(const mvdb_index *h);     /* get_results_cap, hnsw.ts:86  */
int mvdb_has_node(const mvdb_index *h, int id);    /* has_node, hnsw.ts:135        */
int mvdb_get_max_elements(const mvdb_index *h);    /* get_max_elements, hnsw.ts:103 */
int mvdb_get_dim(const mvdb_index *h);             /* DIM, assembly/config.ts:16 (bindings size buffers by it) */
int mvdb_element_count(const mvdb_index *h);
int mvdb_was_ef_clamped(const mvdb_index *h);      /* was_ef_clamped, hnsw.ts:94   */
void mvdb_clear_ef_clamped(mvdb_index *h);         /* clear_ef_clamped, hnsw.ts:98 */

/* ---- dump.bin V3 (hnsw.ts:1033-1249) ------------------------------------- */
/* load_index + WasmBridge.load (wasm-bridge.ts:377-468): parse/validate the
 * dump byte-exactly, upload graph + int8 rows to HBM.  On failure the handle
 * holds a fresh empty index of the configured capacity and 0 is returned. */
int mvdb_load_dump(mvdb_index *h, const uint8_t *buf, size_t size);
/* get_index_dump_size / save_index (hnsw.ts:1048, :1075) */
size_t mvdb_dump_size(const mvdb_index *h);
size_t mvdb_save_dump(const mvdb_index *h, uint8_t *buf);

/* ---- write path on the host graph (assembly/hnsw.ts:719-953) -------------- */
/* insert (upsert) / WasmBridge.insertManyPacked (wasm-bridge.ts:268-288).  The graph is
 * edited on the host; before the next search only the rows of the touched nodes (int8 row,
 * layer-0 list, upper lists) are re-derived and patched into the device mirror, so an oplog
 * replay (src/mini-vector-db.ts:1544-1587) costs O(changed nodes), not a re-upload. */
int mvdb_insert_many(mvdb_index *h, const int32_t *ids, const int8_t *packed, int n);
/* update_vector (hnsw.ts:837) */
int mvdb_update_vector(mvdb_index *h, int id, const int8_t *vec);
/* Benchmark-data builder: the same insert algorithm run by `threads` workers
 * (NOT the reference's sequential order).  ids 0..n-1, index must be empty. */
int mvdb_build_parallel(mvdb_index *h, const int8_t *packed, int n, int threads);
/* Bulk builder on the GPU: the reference's per-insert algorithm (search_layer in build
 * mode hnsw.ts:423-523, select_neighbors_heuristic :526-574, add_connection / remove_connection :593-678)
 * applied to a block of new nodes at a time; nodes of one block do not see each other, blocks are at most
 * 1/32 of the nodes already linked (and <= max_block, <= 0 = 32768).  ids 0..n-1, index must be empty,
 * M_MAX0 <= 64.  Same level assignment as a sequential build with the handle's seed; the edge set is not
 * the sequential one.  The result is an ordinary host graph: mvdb_save_dump writes a V3 file the reference
 * loads.  stats_out: NULL or 8 doubles {seconds, host-prefix seconds, distance evaluations, nodes expanded,
 * evictions, removed back edges, passes, nodes linked on the host}. */
int mvdb_build_gpu(mvdb_index *h, const int8_t *packed, int n, int max_block, double *stats_out);
/* Bulk insert into a LIVE index with the same block-parallel builder: links ids element_count ..
 * element_count + count - 1 (the batch form of insertManyPacked for fresh consecutive ids,
 * src/core/wasm-bridge.ts:268-288 -> insert, assembly/hnsw.ts:719-834).  Needs dense ids (every id
 * below element_count present, none above) and count ids of spare capacity; levels continue the
 * RNG stream in id order as sequential inserts would, the edge order is block-parallel
 * (mvdb_build_gpu's).  The device mirror is re-uploaded in full at the next search.  stats_out as
 * mvdb_build_gpu.  Not to be called while host slots hold batches in flight. */
int mvdb_append_gpu(mvdb_index *h, const int8_t *packed, int count, int max_block, double *stats_out);

/* ---- float32 store (vectors.f32.bin, src/mini-vector-db.ts:1013-1036) ---- */
/* Upload `rows` x dim f32 rows (row id at byte id*dim*4) for the rerank stage.
 * resident=1: copy into HBM (preloadVectors=true analogue).
 * resident=0: keep in pinned host memory, gathered over the host link (preloadVectors=false, the
 *   reference's file mode).  As in the reference, a kept candidate whose row lies past the store is
 *   skipped by the resident store (:1090) and is an error in file mode ("Vector store short read",
 *   :1120-1124): the call that synchronises -- or mvdb_check_errors -- fails with that message. */
int mvdb_attach_f32(mvdb_index *h, const float *base, int64_t rows, int resident);

/* ---- stage 0: query prep (src/mini-vector-db.ts:963-984) ----------------- */
/* normalizeF32InPlace + quantizeToI8 for nq x dim f32 queries, bit-exact.
 * q_norm_out (nq x dim f32) and q_i8_out (nq x dim i8) may each be NULL. */
int mvdb_prepare_queries(mvdb_index *h, const float *q_in, int nq, float *q_norm_out, int8_t *q_i8_out);
int mvdb_prepare_queries_dev(mvdb_index *h, const float *d_q_in, int nq, float *d_q_norm_out,
                             int8_t *d_q_i8_out, void *stream);

/* ---- stage 1: batched HNSW traversal ------------------------------------- */
/* WasmBridge.searchMany (wasm-bridge.ts:326-355) == nq x search()
 * (hnsw.ts:960-1027): the effective-ef rule ef = max(EF_SEARCH, k) clamped to
 * 4096 lives inside.  q: nq x dim packed int8.  out_ids/out_dist: nq x k
 * row-major (row qi valid for out_count[qi] entries, ascending dist, exactly the
 * (i32 id, f32 dist) pairs the WASM writes).  ctr: optional nq counters. */
int mvdb_search_many(mvdb_index *h, const int8_t *q, int nq, int k, int32_t *out_ids, float *out_dist,
                     int32_t *out_count, mvdb_query_counters *ctr);
int mvdb_search_many_dev(mvdb_index *h, const int8_t *d_q, int nq, int k, int32_t *d_out_ids,
                         float *d_out_dist, int32_t *d_out_count, mvdb_query_counters *d_ctr, void *stream);
/* Upload the index to the device now (every compute entry point otherwise does it lazily after a
 * change).  Also the set-up call of mvdb_search_many_dev_noalloc.  No reference counterpart. */
int mvdb_upload(mvdb_index *h);
/* NO COUNTERPART IN THE REFERENCE.  mvdb_search_many_dev for callers that cannot allocate: code under HIP
 * graph capture, tight serving loops, harnesses that forbid hipMalloc.  Same kernels, same outputs; the
 * per-query visited bitmaps live in a static 64 MiB device array of the library (one copy per GPU, held
 * until the process ends) instead of hipMalloc'ed scratch, so the call makes no hipMalloc, hipFree or
 * event creation.  Requirements: mvdb_upload(h) since the last change of the index (else status 0);
 * nq * round_up(ceil(max_elements / 32) * 4, 256) bytes <= 64 MiB (512 queries at 1 M nodes; else status 0 with a clear
 * text); device ordinal < 16.  Every call on one GPU shares that GPU's array: calls that overlap on one
 * GPU (any handle, any stream) race -- serialise them. */
int mvdb_search_many_dev_noalloc(mvdb_index *h, const int8_t *d_q, int nq, int k, int32_t *d_out_ids,
                                 float *d_out_dist, int32_t *d_out_count, mvdb_query_counters *d_ctr, void *stream);
/* WasmBridge.search (wasm-bridge.ts:301-321): one query, returns the count */
int mvdb_search(mvdb_index *h, const int8_t *q, int k, int32_t *out_ids, float *out_dist);

/* ---- stage 2: float32 rerank + top-k (src/mini-vector-db.ts:606-651) ------ */
/* For each query: walk cand[qi][0..cand_count) in ANN order keeping ids whose
 * bit is set in `allow` (NULL = all; replaces the per-candidate metadata /
 * deleted / filter checks :610-616), at most ann_k; L2^2 against the f32 row in
 * f64 exactly as l2SqF32 (:986-993); order by (l2 asc, id asc) (:637); emit the
 * first topk.  out_ids/out_l2: nq x topk, out_n[nq]. */
int mvdb_rerank_many(mvdb_index *h, const float *q_f32, int nq, const int32_t *cand,
                     const int32_t *cand_count, int stride, int ann_k, const uint32_t *allow, int topk,
                     int32_t *out_ids, double *out_l2, int32_t *out_n);
int mvdb_rerank_many_dev(mvdb_index *h, const float *d_q_f32, int nq, const int32_t *d_cand,
                         const int32_t *d_cand_count, int stride, int ann_k, const uint32_t *d_allow,
                         int topk, int32_t *d_out_ids, double *d_out_l2, int32_t *d_out_n, void *stream);

/* ---- fused searchMany (src/mini-vector-db.ts:560-655) --------------------- */
/* prep -> traversal (k = ann_k) -> filter -> rerank -> topk, all on device.
 * ef_search / ann_k are the values derived at :571-576 (mvdb_derive_params).
 * d_allow: device bitmap over internal ids or NULL.  Outputs nq x topk. */
int mvdb_query_many_dev(mvdb_index *h, const float *d_q_f32, int nq, int topk, int ef_search, int ann_k,
                        const uint32_t *d_allow, int32_t *d_out_ids, double *d_out_l2, int32_t *d_out_n,
                        mvdb_query_counters *d_ctr, void *stream);
int mvdb_query_many(mvdb_index *h, const float *q_f32, int nq, int topk, int ef_search, int ann_k,
                    const uint32_t *allow, int32_t *out_ids, double *out_l2, int32_t *out_n,
                    mvdb_query_counters *ctr);
/* The same fused path with several batches in flight, for hosts that hold no device pointers
 * (Node N-API, plain C).  The reference's searchMany (mini-vector-db.ts:560-655) is synchronous, so
 * this pair has no single counterpart: submit = the body of searchMany up to and including the
 * rerank, enqueued on the slot's own stream (host -> device copy of the queries, prep, traversal,
 * rerank, device -> host copy of the results into pinned staging) and returns at once; wait blocks
 * on that slot only and hands the results over.  slot 0..3; a slot holds one batch at a time;
 * allow is copied before submit returns.  Do not change the index (insert / load / setResultsCap)
 * while any slot is busy: submit refuses to upload then.  A traversal abort raised by any slot is
 * reported by the next wait (one error word per handle). */
int mvdb_query_many_submit(mvdb_index *h, int slot, const float *q_f32, int nq, int topk, int ef_search,
                           int ann_k, const uint32_t *allow);
int mvdb_query_many_wait(mvdb_index *h, int slot, int32_t *out_ids, double *out_l2, int32_t *out_n);
/* nq / topk of the batch a busy slot holds (what the wait will copy: size the output arrays from THIS);
 * 0 when the slot is idle or out of range.  No reference counterpart. */
int mvdb_query_many_slot_shape(mvdb_index *h, int slot, int *nq, int *topk);
/* ef / annK derivation (mini-vector-db.ts:568-576) and the results-cap growth
 * rule ensureResultsCapAtLeast (:1160-1170), host-only helpers */
void mvdb_derive_params(int topk_in, int base_ef_search, int rerank_multiplier, int max_ann_k, int *topk,
                        int *ef_search, int *ann_k);
void mvdb_ensure_results_cap_at_least(mvdb_index *h, int n);
/* toScore (mini-vector-db.ts:995-1001): mode 0 "l2", 1 "cosine", else "sim" */
double mvdb_to_score(double l2, int mode);

/* ---- int8 distance kernels (assembly/math.ts) ----------------------------- */
/* dist_l2_i8 (:115) / dot_i8 (:133) of query qi against node ids[qi][j]:
 * out[qi][j] = exact i32 (ids < 0 or absent -> INT32_MIN).  metric 0 = L2, 1 = dot. */
int mvdb_dist_many(mvdb_index *h, const int8_t *q, int nq, const int32_t *ids, int per_q, int metric,
                   int32_t *out);
/* exact brute-force int8 scan over all present nodes; order (dist asc, id asc)
 * for L2, (dot desc, id asc) for dot; k <= 32, dim <= 512.  No reference counterpart
 * (north_star kernel #2); parity target is the oracle's mvo_scan_many. */
int mvdb_scan_many(mvdb_index *h, const int8_t *q, int nq, int k, int metric, int32_t *out_ids,
                   int32_t *out_val, int32_t *out_count);
int mvdb_scan_many_dev(mvdb_index *h, const int8_t *d_q, int nq, int k, int metric, int32_t *d_out_ids,
                       int32_t *d_out_val, int32_t *d_out_count, void *stream);

/* ---- top-k merge for id-range shards --------------- */
/* Merge `parts` per-shard results (each nq x topk, ids already global) into
 * nq x topk ordered by (l2 asc, id asc).  Inputs laid out [part][nq][topk]. */
int mvdb_merge_topk_dev(mvdb_index *h, const int32_t *d_ids, const double *d_l2, const int32_t *d_n,
                        int parts, int nq, int topk, int32_t *d_out_ids, double *d_out_l2,
                        int32_t *d_out_n, void *stream);

/* The same merge over PACKED per-shard blocks, the unit of the one all-gather per batch: block =
 * {l2 f64[nq][topk] | ids i32[nq][topk] | n i32[nq]} padded to a multiple of 8 bytes, mvdb_topk_block_bytes(nq, topk)
 * bytes, `parts` blocks back to back (what all_gather_into_tensor produces).  A block whose n[q] is -1 (an
 * aborted query on that shard) makes the merged n[q] -1. */
size_t mvdb_topk_block_bytes(int nq, int topk);
int mvdb_merge_topk_packed_dev(mvdb_index *h, const void *d_blocks, int parts, int nq, int topk, int32_t *d_out_ids,
                               double *d_out_l2, int32_t *d_out_n, void *stream);

/* ---- engine state beyond the WASM surface ---------------------------------- */
/* Kernel variants of the timed path (handle state, read at launch): traverse_waves
 * 0 = default = the lean path when M_MAX0 <= 64 (one wave per query, result keys only, the
 * result heap of hnsw.ts:233-325 replayed from the admission log by a second launch;
 * csrc/traverse_lean.hip.hpp); 1 forces the kernel that keeps the result heap inside the
 * traversal (csrc/traverse.hip.hpp; also what M_MAX0 > 64 and the no-allocation entry point use;
 * bit-identical, tests/test_gpu_traverse_modes.py);
 * rerank_waves 0 = default (4), 2 = small workgroups that fit beside a running traversal.
 * No reference counterpart. */
int mvdb_set_tuning(mvdb_index *h, int traverse_waves, int rerank_waves);
/* After the caller synchronised its stream(s): 1 = no kernel of this handle raised the
 * error word since the last check, 0 = one did (mvdb_last_error has the text).  Clears it.
 * The WASM cannot fail this way; the reference's nearest analogue is the abort import
 * that only logs (wasm-bridge.ts:57-59). */
int mvdb_check_errors(mvdb_index *h);
/* Device-mirror traffic since mvdb_create: out3 = {full uploads, incremental patches, nodes patched}. */
void mvdb_upload_counters(const mvdb_index *h, long long *out3);
/* The traversal's copy of the index under a locality order of the node labels (breadth-first over the layer-0 lists
 * from the entry point: neighbours share visited-bitmap lines and DRAM pages where the data has neighbourhoods; node
 * ids are labels to search_layer, hnsw.ts:423-523, so results do not change -- csrc/device_index.hpp).
 * mode 1 (default): every full upload of the mirror builds the order and keeps the copy if a list's neighbours then
 * share bitmap lines (at least two per 1024-label line on average over sampled lists; i.i.d. rows do not, and run on
 * their own ids); 2: keep it regardless (tests); 0: never.  The copy costs a second set of rows and lists in HBM; an
 * incremental patch drops it until the next full upload; a change of mode makes the next search upload again.
 * mvdb_relabelled = 1 while traversals walk the relabelled copy.  No reference counterpart. */
void mvdb_set_relabel(mvdb_index *h, int mode);
int mvdb_relabelled(const mvdb_index *h);
/* Test hook: query index `query` of every following traversal launch takes the abort
 * path (-1 = off). */
void mvdb_debug_inject_abort(mvdb_index *h, int query);
/* Test hook: mvdb_build_gpu / mvdb_append_gpu fail before GPU pass `pass` (-1 = off), as an allocation or launch
 * failure would; the nodes not linked yet are taken back (status 0, the index keeps what it reports). */
void mvdb_debug_inject_build_failure(mvdb_index *h, int pass);
/* Queries the lean traversal handed to the one-wave kernel since mvdb_create, by reason: out3 = {a distance
 * beyond 16 bits (cannot happen for quantised unit vectors), admission log full, predicted candidate not at
 * the root (an internal error: never observed)}.  They are answered correctly either way; the counters exist
 * so that tests and benchmarks can tell which kernel ran.  No reference counterpart. */
void mvdb_lean_fallbacks(const mvdb_index *h, long long *out3);
/* Queries whose result heap replay_kernel did not have to rebuild / had to rebuild since mvdb_create: out2 =
 * {short cuts, full replays}.  The short cut applies on the fused path (candidates taken as a set) when no admitted
 * key equal to the final threshold was evicted; the answers are the same either way.  No reference counterpart. */
void mvdb_lean_replays(const mvdb_index *h, long long *out2);
/* Test hook: admissions the lean traversal may log per query (0 = the default, 12 * ef + 1024). */
void mvdb_debug_set_lean_log_cap(mvdb_index *h, int cap);
/* Synchronise `stream` and give its per-stream scratch set back (call before
 * destroying a stream that was passed to a `_dev` entry point). */
int mvdb_release_stream(mvdb_index *h, void *stream);

/* ---- measurement ----------------------------------------------------------- */
/* average device time (ms) of the last launch of each stage, measured with HIP
 * events on the launching stream when timing is enabled. stage: 0 prep,
 * 1 traversal, 2 rerank, 3 scan, 4 result-heap replay of the lean traversal (with the
 * fall-back launch in front of it). Returns <0 when no sample exists. */
void mvdb_enable_timing(mvdb_index *h, int on);
float mvdb_last_stage_ms(mvdb_index *h, int stage);
/* launches and summed device ms of a stage since the last reset (one HIP event pair
 * per launch, resolved here, never inside the caller's timed region) */
int mvdb_stage_stats(mvdb_index *h, int stage, long long *launches, double *total_ms, int reset);

#ifdef __cplusplus
}
#endif
#endif
