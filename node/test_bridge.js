// Re-enactment of reference tests/search_many_test.ts + basic_test.ts at the ANN-interface level,
// through the N-API addon.  Run on a GPU box: `node node/test_bridge.js`.
"use strict";
const os = require("os"), fs = require("fs"), path = require("path");
const { HipBridge, deviceAvailable } = require("./hip-bridge");
function assert(c, m) { if (!c) { console.error("FAILED:", m); process.exit(1); } }
function oneHotI8(dim, i) { const v = new Int8Array(dim); v[i % dim] = 127; return v; }

(async () => {
	assert(deviceAvailable(), "no HIP device visible (there is no CPU fallback)");
	const dim = 384, N = 128;
	const b = new HipBridge();
	await b.init({ dim, m: 16, ef: 300, efSearch: 140, capacity: 50000, seed: 42, resultsCap: 4096 });
	assert((await b.getMaxEf()) === 4096, "getMaxEf");
	for (let i = 0; i < N; i++) await b.insert(i, oneHotI8(dim, i));
	assert(await b.hasNode(5) && !(await b.hasNode(N)), "hasNode");
	await b.setEfSearch(140);
	const qs = []; for (let i = 0; i < 64; i++) qs.push(oneHotI8(dim, i));
	const res = await b.searchMany(qs, 360);
	assert(res.length === 64, "searchMany length");
	for (let i = 0; i < 10; i++) {
		assert(res[i].length > 0 && res[i][0].id === i && res[i][0].dist === 0, `top1 of query ${i}`);
		assert(res[i][1].dist === 2 * 127 * 127, "distinct one-hots are 32258 apart");
	}
	const one = await b.search(oneHotI8(dim, 3), 5);
	assert(one[0].id === 3 && one.length === 5, "search");
	// save -> load round trip (basic_test.ts:107-124), config mismatch (config_mismatch_test.ts)
	const fp = path.join(fs.mkdtempSync(path.join(os.tmpdir(), "mvdb-node-")), "dump.bin");
	await b.save(fp);
	const b2 = new HipBridge();
	await b2.init({ dim, m: 16, ef: 300, capacity: 10, seed: 1, resultsCap: 4096 });
	await b2.load(fp);
	const again = await b2.searchMany(qs.slice(0, 10), 360);
	for (let i = 0; i < 10; i++) assert(JSON.stringify(again[i]) === JSON.stringify(res[i]), "reloaded index answers identically");
	const b3 = new HipBridge();
	await b3.init({ dim: 512, m: 16, ef: 300, capacity: 10, seed: 1 });
	let threw = "";
	try { await b3.load(fp); } catch (e) { threw = String(e.message); }
	assert(/mismatch/.test(threw) && /config/.test(threw), "load of a dim-384 dump at dim 512 must throw the reference message");
	let mism = "";
	try { await b.insertManyPacked(Int32Array.of(1, 2), new Int8Array(5), dim); } catch (e) { mism = String(e.message); }
	assert(/length mismatch/.test(mism), "insertManyPacked length check");
	// wrong-length inputs are errors, never out-of-bounds reads (search, searchMany, queryMany, allow bitmap)
	for (const bad of [() => b.search(new Int8Array(dim - 1), 3), () => b.searchMany([oneHotI8(dim, 1), new Int8Array(7)], 3),
		() => b.queryMany(new Float32Array(dim), 2, 3, 140, 360, null), () => b.queryMany(new Float32Array(dim), -1, 3, 140, 360, null),
		() => b.insert(7, new Int8Array(dim + 1))]) {
		let msg = "";
		try { await bad(); } catch (e) { msg = String(e.message); }
		assert(msg.length > 0, "a wrong-length argument must throw");
	}
	// fused pipeline: prep -> traversal -> rerank -> topK
	const base = new Float32Array(N * dim); for (let i = 0; i < N; i++) base[i * dim + (i % dim)] = 1;
	await b.attachF32(base, dim, true);
	const q = new Float32Array(4 * dim); for (let i = 0; i < 4; i++) q[i * dim + i] = 3.5; // un-normalised
	const r = await b.queryMany(q, 4, 3, 140, 360, null);
	for (let i = 0; i < 4; i++) assert(r.ids[i * 3] === i && r.l2[i * 3] === 0 && r.l2[i * 3 + 1] === 2 && r.n[i] === 3, "queryMany");
	let shortAllow = "";
	try { await b.queryMany(q, 4, 3, 140, 360, new Uint32Array(2)); } catch (e) { shortAllow = String(e.message); }
	assert(/allow bitmap/.test(shortAllow), "a short allow bitmap must throw");
	// several batches in flight: 7 batches of different sizes over 3 slots answer exactly like one-at-a-time queryMany
	const batches = [];
	for (let bi = 0; bi < 7; bi++) {
		const nqb = 1 + (bi * 5) % 9;
		const qb = new Float32Array(nqb * dim);
		for (let i = 0; i < nqb; i++) { qb[i * dim + ((bi * 11 + i) % N)] = 1 + bi; qb[i * dim + ((bi * 3 + i + 1) % N)] = 0.25; }
		batches.push(qb);
	}
	const piped = await b.queryManyBatches(batches, 5, 140, 360, null, 3);
	for (let bi = 0; bi < 7; bi++) {
		const single = await b.queryMany(batches[bi], batches[bi].length / dim, 5, 140, 360, null);
		for (const key of ["ids", "l2", "n"]) assert(JSON.stringify(Array.from(piped[bi][key])) === JSON.stringify(Array.from(single[key])), `queryManyBatches batch ${bi} ${key}`);
	}
	let badBatch = "";
	try { await b.queryManyBatches([batches[0], new Float32Array(dim + 1)], 5, 140, 360); } catch (e) { badBatch = String(e.message); }
	assert(badBatch.length > 0, "a ragged batch must throw");
	const after = await b.queryManyBatches([batches[1]], 5, 140, 360); // no slot was left busy by the failed sweep
	assert(after.length === 1 && after[0].n.length === batches[1].length / dim, "slots are free again after an error");
	// GPU bulk build + bulk append through the addon: every stored vector finds itself
	const bb = new HipBridge();
	await bb.init({ dim: 64, m: 16, ef: 200, efSearch: 64, capacity: 6000, seed: 7, resultsCap: 4096 });
	const pk = new Int8Array(5000 * 64);
	let sd = 12345;
	for (let i = 0; i < pk.length; i++) { sd = (sd * 1103515245 + 12345) & 0x7fffffff; pk[i] = (sd >> 16) % 255 - 127; }
	await bb.buildGpu(pk.subarray(0, 4000 * 64));
	await bb.appendGpu(pk.subarray(4000 * 64));
	assert((await bb.searchMany([pk.subarray(10 * 64, 11 * 64), pk.subarray(4500 * 64, 4501 * 64)], 1)).map((r) => r[0].id).join() === "10,4500", "bulk-linked vectors find themselves");
	let noRoom = "";
	try { await bb.appendGpu(pk.subarray(0, 2000 * 64)); } catch (e) { noRoom = String(e.message); }
	assert(/capacity/.test(noRoom), "appendGpu beyond the capacity must throw");
	bb.close();
	b.close(); b2.close(); b3.close();
	console.log("node bridge test passed");
})().catch((e) => { console.error("FAILED:", e); process.exit(1); });
