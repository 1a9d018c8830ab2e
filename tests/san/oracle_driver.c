/* Sanitizer driver for the CPU oracle (oracle/mvdb_oracle.c, test infrastructure): build, search, rerank,
 * exact scan, dump save -> load, and load_index on truncated / corrupted dumps.  Built with
 * -fsanitize=address,undefined by tests/san/Makefile; run by tests/test_sanitizers.py. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mvdb_oracle.h"

static uint32_t rs = 12345;
static uint32_t rnd(void)
{
	rs ^= rs << 13;
	rs ^= rs >> 17;
	rs ^= rs << 5;
	return rs;
}

int main(void)
{
	const int dim = 32, n = 1200, cap = 1500;
	mvo_set_alloc_limit_bytes((size_t)64 << 20); /* corrupted capacities beyond this are failed loads */
	mvo_index *ix = mvo_create(dim, 8, 60, 50, 1000, 7, cap);
	if (!ix) return 2;
	int8_t *v = (int8_t *)malloc((size_t)n * dim);
	for (int i = 0; i < n * dim; i++) v[i] = (int8_t)((int)(rnd() % 255) - 127);
	for (int i = 0; i < n; i++) mvo_insert(ix, (i * 5) % cap, v + (size_t)i * dim);
	mvo_insert(ix, -3, v);
	mvo_insert(ix, cap + 1, v);
	int32_t ids[400];
	float dist[400];
	mvo_set_ef_search(ix, 120);
	for (int q = 0; q < 40; q++) {
		/* id 5q was last written by insert number 900 + q (ids repeat every 300 inserts: upserts) */
		int c = mvo_search(ix, v + (size_t)(900 + q) * dim, 400, ids, dist, NULL);
		if (c <= 0 || dist[0] != 0.0f) {
			fprintf(stderr, "self search failed (%d)\n", c);
			return 1;
		}
	}
	size_t sz = mvo_dump_size(ix);
	uint8_t *dump = (uint8_t *)malloc(sz);
	if (mvo_save(ix, dump) != sz) return 1;
	for (size_t cut = 0; cut < sz; cut += (cut < 200 ? 1 : 1013)) {
		mvo_index *t = mvo_create(dim, 8, 60, 50, 1000, 1, 3);
		(void)mvo_load(t, dump, cut);
		mvo_destroy(t);
	}
	for (int k = 0; k < 2000; k++) {
		size_t len = sz < 50000 ? sz : 50000;
		uint8_t *bad = (uint8_t *)malloc(len);
		memcpy(bad, dump, len);
		size_t off = k < 1000 ? rnd() % 48 : 48 + rnd() % (len - 48);
		bad[off] = (uint8_t)rnd();
		mvo_index *t = mvo_create(dim, 8, 60, 50, 1000, 1, 3);
		if (mvo_load(t, bad, len)) {
			int8_t q[32] = {0};
			(void)mvo_search(t, q, 50, ids, dist, NULL);
			mvo_insert(t, 0, q);
		}
		mvo_destroy(t);
		free(bad);
	}
	mvo_index *t = mvo_create(dim, 8, 60, 50, 1000, 1, 3);
	if (!mvo_load(t, dump, sz)) return 1;
	uint8_t *again = (uint8_t *)malloc(mvo_dump_size(t));
	if (mvo_save(t, again) != sz || memcmp(again, dump, sz) != 0) return 1;
	mvo_destroy(t);
	mvo_destroy(ix);
	free(again);
	free(dump);
	free(v);
	puts("oracle driver ok");
	return 0;
}
