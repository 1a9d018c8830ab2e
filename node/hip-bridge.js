// node/hip-bridge.js -- drop-in for `class WasmBridge` (reference src/core/wasm-bridge.ts:24).
//
// Same async method names, arity and return shapes; the AssemblyScript/WASM module is replaced by
// the N-API addon over libmvdb_hip.so (include/mvdb_hip.h).  In the reference host the only change
// is the import in src/mini-vector-db.ts:
//     - import { WasmBridge } from "./core/wasm-bridge";
//     + const { HipBridge: WasmBridge } = require("mvdb-hip/node/hip-bridge");
// Extras beyond the WasmBridge surface (attachF32 / queryMany / prepareQueries) expose the fused
// device pipeline for searchMany (src/mini-vector-db.ts:560-655).
"use strict";
const fs = require("fs");
const path = require("path");
const native = require(path.join(__dirname, "mvdb_napi.node"));

const LOAD_ERROR = "WASM load_index failed (format mismatch / config mismatch / corrupt dump)."; // :457-459

class HipBridge {
	constructor(device = -1) {
		this.h = null;
		this.currentConfig = null; // :28
		this.lock = Promise.resolve(); // :36
		this.wasmMaxEf = 0;
		this.device = device;
	}

	// FIFO promise-chain mutex, wasm-bridge.ts:39-50
	async withLock(fn) {
		let release;
		const next = new Promise((r) => (release = r));
		const prev = this.lock;
		this.lock = prev.then(() => next);
		await prev;
		try {
			return await fn();
		} finally {
			release();
		}
	}

	_create(config) {
		const seed = (config.seed != null ? config.seed : Date.now() ^ (process.pid << 16)) >>> 0; // :89-93
		const resultsCap = config.resultsCap != null ? config.resultsCap : 1000; // :84
		if (this.h) native.destroy(this.h);
		this.h = native.create(config.dim | 0, config.m | 0, config.ef | 0, (config.efSearch != null ? config.efSearch : 50) | 0,
			config.capacity | 0, resultsCap | 0, seed, this.device | 0);
		this.wasmMaxEf = native.getMaxEf(this.h);
	}

	async init(config) { // :52-103
		await this.withLock(() => {
			this.currentConfig = Object.assign({}, config);
			this._create(config);
		});
	}

	clampResultsCap(cap) { // :105-110
		const x = cap | 0;
		if (x <= 0) return 1;
		if (this.wasmMaxEf > 0) return Math.min(x, this.wasmMaxEf);
		return x;
	}
	async getMaxEf() { return this.withLock(() => this.wasmMaxEf || native.getMaxEf(this.h)); } // :112
	async setEfSearch(efSearch) { // :123
		await this.withLock(() => { const ef = efSearch | 0; if (ef > 0) native.setEfSearch(this.h, ef); });
	}
	async getResultsCap() { return this.withLock(() => native.getResultsCap(this.h)); } // :170
	async setResultsCap(cap) { await this.withLock(() => { native.setResultsCap(this.h, this.clampResultsCap(cap)); }); } // :179
	async hasNode(id) { return this.withLock(() => !!native.hasNode(this.h, id | 0)); } // :187

	async reinitIndex(capacity) { // :199-240
		await this.withLock(() => {
			const cfg = this.currentConfig;
			if (!cfg) throw new Error("WasmBridge not initialized");
			this.currentConfig = Object.assign({}, cfg, { capacity });
			this._create(this.currentConfig);
		});
	}

	// every vector handed to the addon has the configured dim (the WASM copies DIM bytes whatever the
	// array holds, wasm-bridge.ts:151-160; here a wrong length is an error, not an out-of-bounds read)
	_vec(vectorI8, what) {
		const dim = this.currentConfig ? this.currentConfig.dim | 0 : 0;
		if (!vectorI8 || vectorI8.length !== dim) {
			throw new Error(`${what}: vector length ${vectorI8 ? vectorI8.length : 0} != dim ${dim}`);
		}
		return vectorI8 instanceof Int8Array ? vectorI8 : Int8Array.from(vectorI8);
	}

	async insert(id, vectorI8) { // :242
		await this.withLock(() => { native.insertManyPacked(this.h, Int32Array.of(id), this._vec(vectorI8, "insert")); });
	}
	async insertMany(pairs) { // :253
		await this.withLock(() => {
			for (const p of pairs) native.insertManyPacked(this.h, Int32Array.of(p.id), this._vec(p.vectorI8, "insertMany"));
		});
	}
	async insertManyPacked(ids, packed, dim) { // :268-288
		if (ids.length === 0) return;
		if (packed.length !== ids.length * dim) {
			throw new Error(`insertManyPacked length mismatch: packed=${packed.length}, ids=${ids.length}, dim=${dim}`);
		}
		await this.withLock(() => { native.insertManyPacked(this.h, ids, packed); });
	}
	// Engine extensions (no WasmBridge counterpart): block-parallel linking on the GPU, same per-insert algorithm.
	// buildGpu: ids 0..n-1 into an empty index; appendGpu: ids elementCount.. into the live index (dense ids).
	async buildGpu(packed) { return this.withLock(() => native.bulkLink(this.h, packed, 0)); }
	async appendGpu(packed) { return this.withLock(() => native.bulkLink(this.h, packed, 1)); }
	async updateVector(id, vectorI8) { // :290 update_and_reconnect: an absent id is a silent no-op
		await this.withLock(() => {
			if (native.hasNode(this.h, id | 0)) native.insertManyPacked(this.h, Int32Array.of(id), this._vec(vectorI8, "updateVector"));
		});
	}

	_decode(r, nq, k) { // result decode, :310-319 / :341-349
		const out = new Array(nq);
		const kk = Math.max(k, 1);
		for (let qi = 0; qi < nq; qi++) {
			const n = r.count[qi];
			const row = new Array(n);
			for (let i = 0; i < n; i++) row[i] = { id: r.ids[qi * kk + i], dist: r.dist[qi * kk + i] };
			out[qi] = row;
		}
		return out;
	}
	async search(vectorI8, k) { // :301
		return this.withLock(() => this._decode(native.searchMany(this.h, this._vec(vectorI8, "search"), 1, k | 0), 1, k | 0)[0]);
	}
	async searchMany(vectorsI8, k) { // :326
		return this.withLock(() => {
			const nq = vectorsI8.length;
			if (nq === 0) return [];
			const dim = this.currentConfig.dim | 0;
			const packed = new Int8Array(nq * dim);
			for (let i = 0; i < nq; i++) packed.set(this._vec(vectorsI8[i], "searchMany"), i * dim); // one H2D copy for the batch
			return this._decode(native.searchMany(this.h, packed, nq, k | 0), nq, k | 0);
		});
	}

	async save(filePath) { // :357
		await this.withLock(async () => { await fs.promises.writeFile(filePath, native.saveDump(this.h)); });
	}
	async load(filePath) { // :377-468
		await this.withLock(async () => {
			if (!fs.existsSync(filePath)) return; // :379
			const buf = await fs.promises.readFile(filePath);
			if (!native.loadDump(this.h, buf)) throw new Error(LOAD_ERROR); // engine already holds a fresh empty index
		});
	}

	// ---- fused device pipeline for MiniVectorDB.search()/searchMany() ----
	async attachF32(baseF32, dim, resident = true) {
		await this.withLock(() => { native.attachF32(this.h, baseF32, dim | 0, resident ? 1 : 0); });
	}
	async prepareQueries(qF32, nq) { return this.withLock(() => native.prepareQueries(this.h, qF32, nq | 0)); }
	async queryMany(qF32, nq, topK, efSearch, annK, allowBitmap = null) {
		return this.withLock(() => native.queryMany(this.h, qF32, nq | 0, topK | 0, efSearch | 0, annK | 0, allowBitmap));
	}
	// Several batches in flight (the engine runs each slot on its own stream; mvdb_query_many_submit / _wait).
	// `batches`: array of Float32Array, each nq_i x dim; returns the per-batch {ids, l2, n} in order.  The whole
	// sweep holds the bridge lock, like every other call that touches the engine (wasm-bridge.ts:112-129).
	async queryManyBatches(batches, topK, efSearch, annK, allowBitmap = null, inFlight = 3) {
		return this.withLock(() => {
			const dim = this.currentConfig.dim | 0;
			const depth = Math.max(1, Math.min(4, inFlight | 0));
			const out = new Array(batches.length);
			const nqOf = (b) => {
				if (!(b instanceof Float32Array) || b.length % dim !== 0) throw new TypeError(`queryManyBatches: every batch must be a Float32Array of nq x ${dim} values`);
				return b.length / dim;
			};
			const pending = []; // {slot, i, nq}
			let error = null;
			for (let i = 0; i < batches.length && !error; i++) {
				if (pending.length === depth) {
					const p = pending.shift();
					out[p.i] = native.waitMany(this.h, p.slot, p.nq, topK | 0);
				}
				const slot = i % depth;
				try {
					const nq = nqOf(batches[i]);
					native.submitMany(this.h, slot, batches[i], nq, topK | 0, efSearch | 0, annK | 0, allowBitmap);
					pending.push({ slot, i, nq });
				} catch (e) { error = e; }
			}
			for (const p of pending) { // drain every slot, also after an error, so that none stays busy
				try { out[p.i] = native.waitMany(this.h, p.slot, p.nq, topK | 0); } catch (e) { error = error || e; }
			}
			if (error) throw error;
			return out;
		});
	}
	close() { if (this.h) { native.destroy(this.h); this.h = null; } }
}

module.exports = { HipBridge, deviceAvailable: native.deviceAvailable };
