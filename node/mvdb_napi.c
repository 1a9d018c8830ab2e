/*
 * mvdb_napi.c -- thin N-API addon over the C ABI of libmvdb_hip.so (include/mvdb_hip.h).
 *
 * This is the binding the reference host (Node.js) loads in place of the AssemblyScript
 * loader: node/hip-bridge.js wraps it in a class with the method surface of WasmBridge
 * (reference src/core/wasm-bridge.ts:24).  Every function is synchronous and copies nothing
 * beyond what the C ABI itself copies: typed arrays are passed by pointer.
 */
#include <node_api.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "mvdb_hip.h"

#define NAPI_OK(call)                                                   \
	do {                                                                \
		if ((call) != napi_ok) {                                        \
			napi_throw_error(env, NULL, "N-API call failed: " #call);   \
			return NULL;                                                \
		}                                                               \
	} while (0)

static mvdb_index *get_handle(napi_env env, napi_value v)
{
	void *p = NULL;
	if (napi_get_value_external(env, v, &p) != napi_ok || !p) {
		napi_throw_error(env, NULL, "WasmBridge not initialized");
		return NULL;
	}
	return *(mvdb_index **)p;
}
static void finalize_handle(napi_env env, void *data, void *hint)
{
	(void)env;
	(void)hint;
	mvdb_index **box = (mvdb_index **)data;
	if (*box) mvdb_destroy(*box);
	free(box);
}
static int32_t get_i32(napi_env env, napi_value v)
{
	int32_t x = 0;
	napi_get_value_int32(env, v, &x);
	return x;
}
static napi_value throw_last(napi_env env, mvdb_index *h)
{
	napi_throw_error(env, NULL, mvdb_last_error(h));
	return NULL;
}
static napi_value throw_type(napi_env env, const char *msg)
{
	napi_throw_type_error(env, NULL, msg);
	return NULL;
}
/* nq rows of `elem`-byte elements of the handle's dim must fit in `nbytes` (nq < 0 never fits) */
static int rows_fit(mvdb_index *h, int nq, size_t elem, size_t nbytes)
{
	const int dim = mvdb_get_dim(h);
	if (nq < 0 || dim <= 0) return 0;
	return (size_t)nq <= nbytes / elem / (size_t)dim;
}
/* an allow bitmap must cover every internal id: ceil(max_elements / 32) words */
static int allow_fits(mvdb_index *h, size_t nbytes)
{
	int n = mvdb_get_max_elements(h);
	if (n < 1) n = 1;
	return nbytes / 4 >= ((size_t)n + 31) / 32;
}
/* typed array / Buffer -> pointer + byte length (NULL for null/undefined) */
static void *bytes_of(napi_env env, napi_value v, size_t *nbytes)
{
	bool is = false;
	void *data = NULL;
	size_t len = 0;
	*nbytes = 0;
	if (napi_is_typedarray(env, v, &is) == napi_ok && is) {
		napi_typedarray_type t;
		napi_value ab;
		size_t off;
		if (napi_get_typedarray_info(env, v, &t, &len, &data, &ab, &off) != napi_ok) return NULL;
		size_t es = (t == napi_int8_array || t == napi_uint8_array || t == napi_uint8_clamped_array) ? 1
		            : (t == napi_int16_array || t == napi_uint16_array)                             ? 2
		            : (t == napi_float64_array)                                                     ? 8
		                                                                                            : 4;
		*nbytes = len * es;
		return data;
	}
	if (napi_is_buffer(env, v, &is) == napi_ok && is) {
		if (napi_get_buffer_info(env, v, &data, &len) != napi_ok) return NULL;
		*nbytes = len;
		return data;
	}
	return NULL;
}
static napi_value make_typed(napi_env env, napi_typedarray_type t, size_t count, size_t elem, void **data)
{
	napi_value ab, ta;
	if (napi_create_arraybuffer(env, count * elem, data, &ab) != napi_ok) return NULL;
	if (napi_create_typedarray(env, t, count, ab, 0, &ta) != napi_ok) return NULL;
	return ta;
}

/* create(dim, m, ef, efSearch, capacity, resultsCap, seed, device) */
static napi_value Create(napi_env env, napi_callback_info info)
{
	size_t argc = 8;
	napi_value a[8];
	NAPI_OK(napi_get_cb_info(env, info, &argc, a, NULL, NULL));
	mvdb_config c;
	c.dim = get_i32(env, a[0]);
	c.m = get_i32(env, a[1]);
	c.ef_construction = get_i32(env, a[2]);
	c.ef_search = get_i32(env, a[3]);
	c.capacity = get_i32(env, a[4]);
	c.results_cap = get_i32(env, a[5]);
	uint32_t seed = 0;
	napi_get_value_uint32(env, a[6], &seed);
	c.seed = seed;
	c.device = get_i32(env, a[7]);
	mvdb_index *h = NULL;
	if (!mvdb_create(&c, &h)) {
		napi_throw_error(env, NULL, mvdb_last_error(NULL));
		return NULL;
	}
	mvdb_index **box = (mvdb_index **)malloc(sizeof(*box));
	*box = h;
	napi_value ext;
	NAPI_OK(napi_create_external(env, box, finalize_handle, NULL, &ext));
	return ext;
}
static napi_value Destroy(napi_env env, napi_callback_info info)
{
	size_t argc = 1;
	napi_value a[1];
	NAPI_OK(napi_get_cb_info(env, info, &argc, a, NULL, NULL));
	void *p = NULL;
	if (napi_get_value_external(env, a[0], &p) == napi_ok && p) {
		mvdb_index **box = (mvdb_index **)p;
		if (*box) mvdb_destroy(*box);
		*box = NULL;
	}
	return NULL;
}

#define INT_GETTER(NAME, EXPR)                                              \
	static napi_value NAME(napi_env env, napi_callback_info info)           \
	{                                                                       \
		size_t argc = 2;                                                    \
		napi_value a[2], out;                                               \
		NAPI_OK(napi_get_cb_info(env, info, &argc, a, NULL, NULL));         \
		mvdb_index *h = get_handle(env, a[0]);                              \
		if (!h) return NULL;                                                \
		int32_t arg = argc > 1 ? get_i32(env, a[1]) : 0;                    \
		(void)arg;                                                          \
		NAPI_OK(napi_create_int32(env, (int32_t)(EXPR), &out));             \
		return out;                                                         \
	}
INT_GETTER(GetMaxEf, mvdb_get_max_ef())
INT_GETTER(GetResultsCap, mvdb_get_results_cap(h))
INT_GETTER(GetEfSearch, mvdb_get_ef_search(h))
INT_GETTER(HasNode, mvdb_has_node(h, arg))
INT_GETTER(GetMaxElements, mvdb_get_max_elements(h))
INT_GETTER(ElementCount, mvdb_element_count(h))
INT_GETTER(SetEfSearch, (mvdb_set_ef_search(h, arg), 0))
INT_GETTER(SetResultsCap, (mvdb_set_results_cap(h, arg), 0))

/* insertManyPacked(h, Int32Array ids, Int8Array packed) */
static napi_value InsertManyPacked(napi_env env, napi_callback_info info)
{
	size_t argc = 3, nb_ids, nb_vec;
	napi_value a[3];
	NAPI_OK(napi_get_cb_info(env, info, &argc, a, NULL, NULL));
	mvdb_index *h = get_handle(env, a[0]);
	if (!h) return NULL;
	const int32_t *ids = (const int32_t *)bytes_of(env, a[1], &nb_ids);
	const int8_t *vec = (const int8_t *)bytes_of(env, a[2], &nb_vec);
	if (!ids || !vec) {
		napi_throw_type_error(env, NULL, "insertManyPacked(ids: Int32Array, packed: Int8Array)");
		return NULL;
	}
	if (nb_ids / 4 > 0x7fffffff || nb_vec != (nb_ids / 4) * (size_t)mvdb_get_dim(h))
		return throw_type(env, "insertManyPacked: packed.length must equal ids.length * dim");
	if (!mvdb_insert_many(h, ids, vec, (int)(nb_ids / 4))) return throw_last(env, h);
	return NULL;
}
/* bulkLink(h, Int8Array packed, append) -> seconds: mvdb_build_gpu (append = 0: ids 0..n-1 into an empty index) or
 * mvdb_append_gpu (append = 1: ids elementCount.. into the live index); n = packed.length / dim */
static napi_value BulkLink(napi_env env, napi_callback_info info)
{
	size_t argc = 3, nb;
	napi_value a[3], out;
	NAPI_OK(napi_get_cb_info(env, info, &argc, a, NULL, NULL));
	mvdb_index *h = get_handle(env, a[0]);
	if (!h) return NULL;
	const int8_t *vec = (const int8_t *)bytes_of(env, a[1], &nb);
	const int append = get_i32(env, a[2]);
	const size_t dim = (size_t)mvdb_get_dim(h);
	if (!vec || dim == 0 || nb % dim != 0 || nb / dim > 0x7fffffff)
		return throw_type(env, "bulkLink: packed must be an Int8Array of n * dim values");
	double st[8] = {0};
	const int n = (int)(nb / dim);
	if (!(append ? mvdb_append_gpu(h, vec, n, 0, st) : mvdb_build_gpu(h, vec, n, 0, st))) return throw_last(env, h);
	NAPI_OK(napi_create_double(env, st[0], &out));
	return out;
}
/* loadDump(h, Buffer) -> boolean (false = load_index returned 0; the index is fresh and empty) */
static napi_value LoadDump(napi_env env, napi_callback_info info)
{
	size_t argc = 2, nb;
	napi_value a[2], out;
	NAPI_OK(napi_get_cb_info(env, info, &argc, a, NULL, NULL));
	mvdb_index *h = get_handle(env, a[0]);
	if (!h) return NULL;
	const uint8_t *buf = (const uint8_t *)bytes_of(env, a[1], &nb);
	if (!buf) return throw_type(env, "loadDump(dump: Buffer | Uint8Array)");
	NAPI_OK(napi_get_boolean(env, mvdb_load_dump(h, buf, nb) != 0, &out));
	return out;
}
/* saveDump(h) -> Buffer */
static napi_value SaveDump(napi_env env, napi_callback_info info)
{
	size_t argc = 1;
	napi_value a[1], out;
	NAPI_OK(napi_get_cb_info(env, info, &argc, a, NULL, NULL));
	mvdb_index *h = get_handle(env, a[0]);
	if (!h) return NULL;
	size_t n = mvdb_dump_size(h);
	void *data = NULL;
	NAPI_OK(napi_create_buffer(env, n, &data, &out));
	mvdb_save_dump(h, (uint8_t *)data);
	return out;
}
/* searchMany(h, Int8Array packed, nq, k) -> {ids: Int32Array(nq*k), dist: Float32Array(nq*k), count: Int32Array(nq)} */
static napi_value SearchMany(napi_env env, napi_callback_info info)
{
	size_t argc = 4, nb;
	napi_value a[4];
	NAPI_OK(napi_get_cb_info(env, info, &argc, a, NULL, NULL));
	mvdb_index *h = get_handle(env, a[0]);
	if (!h) return NULL;
	const int8_t *q = (const int8_t *)bytes_of(env, a[1], &nb);
	int nq = get_i32(env, a[2]), k = get_i32(env, a[3]);
	if (!q || !rows_fit(h, nq, 1, nb)) return throw_type(env, "searchMany: packed must hold nq rows of dim int8 values");
	size_t kk = k > 0 ? (size_t)k : 1;
	void *ids, *dist, *cnt;
	napi_value t_ids = make_typed(env, napi_int32_array, (size_t)nq * kk, 4, &ids);
	napi_value t_dist = make_typed(env, napi_float32_array, (size_t)nq * kk, 4, &dist);
	napi_value t_cnt = make_typed(env, napi_int32_array, (size_t)nq, 4, &cnt);
	if (!q || !t_ids || !t_dist || !t_cnt) {
		napi_throw_error(env, NULL, "searchMany: bad arguments");
		return NULL;
	}
	if (!mvdb_search_many(h, q, nq, k, (int32_t *)ids, (float *)dist, (int32_t *)cnt, NULL)) return throw_last(env, h);
	napi_value out;
	NAPI_OK(napi_create_object(env, &out));
	NAPI_OK(napi_set_named_property(env, out, "ids", t_ids));
	NAPI_OK(napi_set_named_property(env, out, "dist", t_dist));
	NAPI_OK(napi_set_named_property(env, out, "count", t_cnt));
	return out;
}
/* attachF32(h, Float32Array base, resident) */
static napi_value AttachF32(napi_env env, napi_callback_info info)
{
	size_t argc = 4, nb;
	napi_value a[4];
	NAPI_OK(napi_get_cb_info(env, info, &argc, a, NULL, NULL));
	mvdb_index *h = get_handle(env, a[0]);
	if (!h) return NULL;
	const float *base = (const float *)bytes_of(env, a[1], &nb);
	int dim = get_i32(env, a[2]), resident = get_i32(env, a[3]);
	if (dim != mvdb_get_dim(h)) return throw_type(env, "attachF32: dim differs from the index's dim");
	if (!mvdb_attach_f32(h, base, dim > 0 ? (int64_t)(nb / 4 / (size_t)dim) : 0, resident)) return throw_last(env, h);
	return NULL;
}
/* queryMany(h, Float32Array q, nq, topk, efSearch, annK, Uint32Array allow | null)
 *   -> {ids: Int32Array(nq*topk), l2: Float64Array(nq*topk), n: Int32Array(nq)} */
static napi_value QueryMany(napi_env env, napi_callback_info info)
{
	size_t argc = 7, nb, nba;
	napi_value a[7];
	NAPI_OK(napi_get_cb_info(env, info, &argc, a, NULL, NULL));
	mvdb_index *h = get_handle(env, a[0]);
	if (!h) return NULL;
	const float *q = (const float *)bytes_of(env, a[1], &nb);
	int nq = get_i32(env, a[2]), topk = get_i32(env, a[3]), ef = get_i32(env, a[4]), annk = get_i32(env, a[5]);
	const uint32_t *allow = argc > 6 ? (const uint32_t *)bytes_of(env, a[6], &nba) : NULL;
	if (!q || !rows_fit(h, nq, 4, nb)) return throw_type(env, "queryMany: q must hold nq rows of dim float32 values");
	if (allow && !allow_fits(h, nba)) return throw_type(env, "queryMany: allow bitmap shorter than ceil(max_elements / 32) words");
	size_t tk = topk > 0 ? (size_t)topk : 1;
	void *ids, *l2, *n;
	napi_value t_ids = make_typed(env, napi_int32_array, (size_t)nq * tk, 4, &ids);
	napi_value t_l2 = make_typed(env, napi_float64_array, (size_t)nq * tk, 8, &l2);
	napi_value t_n = make_typed(env, napi_int32_array, (size_t)nq, 4, &n);
	if (!q || !t_ids || !t_l2 || !t_n) {
		napi_throw_error(env, NULL, "queryMany: bad arguments");
		return NULL;
	}
	if (!mvdb_query_many(h, q, nq, topk, ef, annk, allow, (int32_t *)ids, (double *)l2, (int32_t *)n, NULL))
		return throw_last(env, h);
	napi_value out;
	NAPI_OK(napi_create_object(env, &out));
	NAPI_OK(napi_set_named_property(env, out, "ids", t_ids));
	NAPI_OK(napi_set_named_property(env, out, "l2", t_l2));
	NAPI_OK(napi_set_named_property(env, out, "n", t_n));
	return out;
}
/* submitMany(h, slot, Float32Array q, nq, topk, efSearch, annK, Uint32Array allow | null): enqueue a batch on the
 * slot's stream and return (mvdb_query_many_submit) */
static napi_value SubmitMany(napi_env env, napi_callback_info info)
{
	size_t argc = 8, nb, nba;
	napi_value a[8];
	NAPI_OK(napi_get_cb_info(env, info, &argc, a, NULL, NULL));
	mvdb_index *h = get_handle(env, a[0]);
	if (!h) return NULL;
	int slot = get_i32(env, a[1]);
	const float *q = (const float *)bytes_of(env, a[2], &nb);
	int nq = get_i32(env, a[3]), topk = get_i32(env, a[4]), ef = get_i32(env, a[5]), annk = get_i32(env, a[6]);
	const uint32_t *allow = argc > 7 ? (const uint32_t *)bytes_of(env, a[7], &nba) : NULL;
	if (!q || !rows_fit(h, nq, 4, nb)) return throw_type(env, "submitMany: q must hold nq rows of dim float32 values");
	if (topk < 0) return throw_type(env, "submitMany: topk must not be negative");
	if (allow && !allow_fits(h, nba)) return throw_type(env, "submitMany: allow bitmap shorter than ceil(max_elements / 32) words");
	if (!mvdb_query_many_submit(h, slot, q, nq, topk, ef, annk, allow)) return throw_last(env, h);
	return NULL;
}
/* waitMany(h, slot, nq, topk) -> {ids, l2, n} of the batch submitted on the slot (the same nq / topk) */
static napi_value WaitMany(napi_env env, napi_callback_info info)
{
	size_t argc = 4;
	napi_value a[4];
	NAPI_OK(napi_get_cb_info(env, info, &argc, a, NULL, NULL));
	mvdb_index *h = get_handle(env, a[0]);
	if (!h) return NULL;
	int slot = get_i32(env, a[1]), nq = get_i32(env, a[2]), topk = get_i32(env, a[3]);
	if (nq < 0 || topk < 0) return throw_type(env, "waitMany: nq / topk must not be negative");
	/* the wait copies what was SUBMITTED on the slot: the output arrays are sized from that, never from the caller */
	int s_nq = 0, s_topk = 0;
	if (mvdb_query_many_slot_shape(h, slot, &s_nq, &s_topk) && (s_nq != nq || s_topk != topk))
		return throw_type(env, "waitMany: nq / topk differ from the batch submitted on this slot");
	size_t tk = topk > 0 ? (size_t)topk : 1;
	void *ids, *l2, *n;
	napi_value t_ids = make_typed(env, napi_int32_array, (size_t)nq * tk, 4, &ids);
	napi_value t_l2 = make_typed(env, napi_float64_array, (size_t)nq * tk, 8, &l2);
	napi_value t_n = make_typed(env, napi_int32_array, (size_t)nq, 4, &n);
	if (!t_ids || !t_l2 || !t_n) {
		napi_throw_error(env, NULL, "waitMany: allocation failed");
		return NULL;
	}
	if (!mvdb_query_many_wait(h, slot, (int32_t *)ids, (double *)l2, (int32_t *)n)) return throw_last(env, h);
	napi_value out;
	NAPI_OK(napi_create_object(env, &out));
	NAPI_OK(napi_set_named_property(env, out, "ids", t_ids));
	NAPI_OK(napi_set_named_property(env, out, "l2", t_l2));
	NAPI_OK(napi_set_named_property(env, out, "n", t_n));
	return out;
}
/* prepareQueries(h, Float32Array q, nq) -> {norm: Float32Array, i8: Int8Array} (normalise + quantise) */
static napi_value PrepareQueries(napi_env env, napi_callback_info info)
{
	size_t argc = 3, nb;
	napi_value a[3];
	NAPI_OK(napi_get_cb_info(env, info, &argc, a, NULL, NULL));
	mvdb_index *h = get_handle(env, a[0]);
	if (!h) return NULL;
	const float *q = (const float *)bytes_of(env, a[1], &nb);
	int nq = get_i32(env, a[2]);
	if (!q || !rows_fit(h, nq, 4, nb)) return throw_type(env, "prepareQueries: q must hold nq rows of dim float32 values");
	void *qn, *q8;
	napi_value t_n = make_typed(env, napi_float32_array, nb / 4, 4, &qn);
	napi_value t_8 = make_typed(env, napi_int8_array, nb / 4, 1, &q8);
	if (!q || !t_n || !t_8) {
		napi_throw_error(env, NULL, "prepareQueries: bad arguments");
		return NULL;
	}
	if (!mvdb_prepare_queries(h, q, nq, (float *)qn, (int8_t *)q8)) return throw_last(env, h);
	napi_value out;
	NAPI_OK(napi_create_object(env, &out));
	NAPI_OK(napi_set_named_property(env, out, "norm", t_n));
	NAPI_OK(napi_set_named_property(env, out, "i8", t_8));
	return out;
}
static napi_value DeviceAvailable(napi_env env, napi_callback_info info)
{
	(void)info;
	napi_value out;
	NAPI_OK(napi_get_boolean(env, mvdb_device_available() != 0, &out));
	return out;
}

static napi_value Init(napi_env env, napi_value exports)
{
	const struct {
		const char *name;
		napi_callback fn;
	} fns[] = {{"create", Create}, {"destroy", Destroy}, {"getMaxEf", GetMaxEf}, {"getResultsCap", GetResultsCap},
	           {"getEfSearch", GetEfSearch}, {"hasNode", HasNode}, {"getMaxElements", GetMaxElements},
	           {"elementCount", ElementCount}, {"setEfSearch", SetEfSearch}, {"setResultsCap", SetResultsCap},
	           {"insertManyPacked", InsertManyPacked}, {"loadDump", LoadDump}, {"saveDump", SaveDump},
	           {"searchMany", SearchMany}, {"attachF32", AttachF32}, {"queryMany", QueryMany},
	           {"prepareQueries", PrepareQueries}, {"deviceAvailable", DeviceAvailable}, {"submitMany", SubmitMany},
	           {"waitMany", WaitMany}, {"bulkLink", BulkLink}};
	for (size_t i = 0; i < sizeof(fns) / sizeof(fns[0]); i++) {
		napi_value f;
		if (napi_create_function(env, fns[i].name, NAPI_AUTO_LENGTH, fns[i].fn, NULL, &f) != napi_ok) return NULL;
		if (napi_set_named_property(env, exports, fns[i].name, f) != napi_ok) return NULL;
	}
	return exports;
}
NAPI_MODULE(mvdb_napi, Init)
