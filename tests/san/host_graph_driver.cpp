// Sanitizer driver for the product's host graph (csrc/host_graph.cpp; host-only C++): the write path
// (insert / update_and_reconnect / update_vector), dump V3 save -> load round trips, the loader's reject paths on
// corrupted dumps, dirty tracking, the bulk prologue and the multi-threaded builder.  Built by tests/san/Makefile
// with -fsanitize=address,undefined (asan) and -fsanitize=thread (tsan, `threads` mode only); run by
// tests/test_sanitizers.py.  Exit code 0 + no sanitizer report = pass.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "host_graph.hpp"

using namespace mvdb;

static std::vector<int8_t> rows(int n, int dim, unsigned seed)
{
	std::mt19937 rng(seed);
	std::vector<int8_t> v((size_t)n * dim);
	for (auto &x : v) x = (int8_t)((int)(rng() % 255) - 127);
	return v;
}
#define CHECK(c)                                                     \
	do {                                                             \
		if (!(c)) {                                                  \
			std::fprintf(stderr, "CHECK failed: %s (line %d)\n", #c, __LINE__); \
			return 1;                                                \
		}                                                            \
	} while (0)

static int run_single()
{
	HostGraph::alloc_limit_bytes = (size_t)64 << 20;  // corrupted capacities beyond this are failed loads, not allocations
	const int dim = 32, n = 1500, cap = 2000;
	GraphConfig c;
	c.dim = dim;
	c.m = 8;
	c.ef_construction = 60;
	HostGraph g;
	g.seed(7);
	g.init(c, cap);
	auto v = rows(n, dim, 1);
	for (int i = 0; i < n; i++) g.insert(i * 4 % cap, v.data() + (size_t)i * dim);  // sparse ids, some repeated (upserts)
	g.insert(-1, v.data());
	g.insert(cap, v.data());
	g.update_vector(3, v.data());
	g.update_vector(cap + 5, v.data());
	g.clear_dirty();
	g.insert(1, v.data() + dim);
	CHECK(!g.dirty.empty() && !g.dirty_all);
	std::vector<uint8_t> dump(g.dump_size());
	CHECK(g.save(dump.data()) == dump.size());
	HostGraph h;
	h.init(c, 3);
	CHECK(h.load(dump.data(), dump.size()));
	std::vector<uint8_t> again(h.dump_size());
	CHECK(h.save(again.data()) == again.size() && again == dump);
	// every truncation and a sweep of single-byte corruptions of the header and the first records: reject or accept,
	// never read out of bounds
	for (size_t cut = 0; cut < dump.size(); cut += (cut < 200 ? 1 : 977)) {
		HostGraph t;
		t.init(c, 3);
		(void)t.load(dump.data(), cut);
	}
	std::mt19937 rng(5);
	for (int k = 0; k < 3000; k++) {
		std::vector<uint8_t> bad(dump.begin(), dump.begin() + std::min<size_t>(dump.size(), 60000));
		const size_t off = k < 1500 ? rng() % 48 : 48 + rng() % (bad.size() - 48);
		bad[off] = (uint8_t)rng();
		if (k % 3 == 0) bad[(off + 1) % bad.size()] = 0xff;
		HostGraph t;
		t.init(c, 3);
		if (t.load(bad.data(), bad.size())) {
			std::vector<uint8_t> out(t.dump_size());
			(void)t.save(out.data());
			const int8_t q[32] = {0};
			t.insert(0, q);  // an accepted (possibly odd) graph must stay usable by the write path
		}
	}
	// bulk prologue below and above the sequential prefix
	HostGraph b;
	b.seed(9);
	b.init(c, 600);
	CHECK(b.bulk_begin(v.data(), 600, 100) >= 100);
	b.bulk_end(600);
	return 0;
}

static int run_threads(int threads)
{
	const int dim = 32, n = 6000;
	GraphConfig c;
	c.dim = dim;
	c.m = 8;
	c.ef_construction = 40;
	HostGraph g;
	g.seed(11);
	g.init(c, n);
	auto v = rows(n, dim, 2);
	g.build_parallel(v.data(), n, threads);
	CHECK(g.element_count == n);
	for (int i = 0; i < n; i++) {
		CHECK(g.cnt0[i] >= 0 && g.cnt0[i] <= c.m0());
		for (int j = 0; j < g.cnt0[i]; j++) {
			const int nb = g.adj0[(size_t)i * c.m0() + j];
			CHECK(nb >= 0 && nb < n && nb != i);
		}
	}
	std::vector<uint8_t> dump(g.dump_size());
	CHECK(g.save(dump.data()) == dump.size());
	return 0;
}

int main(int argc, char **argv)
{
	const bool threads = argc > 1 && std::strcmp(argv[1], "threads") == 0;
	const int rc = threads ? run_threads(argc > 2 ? std::atoi(argv[2]) : 8) : run_single();
	if (rc == 0) std::puts(threads ? "host_graph threads driver ok" : "host_graph driver ok");
	return rc;
}
