"""The CPU oracle (and the product's host-side dump parser) against the committed golden vectors."""
import numpy as np
import pytest

from minivectordb_amd import HipBridge
from oracle import np_oracle as NP
from oracle import oracle as O
from tests._golden import NAMES, load


@pytest.mark.parametrize("name", NAMES)
def test_oracle_reproduces_golden(name):
    g = load(name)
    ix = O.OracleIndex(g["dim"], g["m"], g["ef_construction"], 3, seed=9)
    assert ix.load_bytes(g["dump"])
    assert ix.save_bytes() == g["dump"]
    qn = O.normalize_rows_f32(g["queries"])
    assert np.array_equal(qn.view(np.uint32), g["q_norm"].view(np.uint32))
    assert np.array_equal(O.quantize_rows_i8(qn), g["q_i8"])
    ids, l2, n, ann_ids, ann_cnt, ctr = O.search_many_db(ix, g["queries"], g["base_f32"], g["topk"], g["preset"])
    assert np.array_equal(ann_ids, g["ann_ids"]) and np.array_equal(ann_cnt, g["ann_cnt"])
    assert np.array_equal(ids, g["top_ids"]) and np.array_equal(n, g["top_n"])
    assert np.array_equal(l2.view(np.uint64), g["top_l2"].view(np.uint64))
    assert np.array_equal(ctr["n_dist"], g["n_dist"])
    ids, l2, n, *_ = O.search_many_db(ix, g["queries"], g["base_f32"], g["topk"], g["preset"], allow_bitmap=g["allow"])
    assert np.array_equal(ids, g["f_ids"]) and np.array_equal(n, g["f_n"])


@pytest.mark.parametrize("name", NAMES)
def test_numpy_restatement_reproduces_golden(name):
    g = load(name)
    gr = NP.parse_dump(g["dump"], g["dim"], g["m"], g["ef_construction"])
    assert gr is not None
    for qi in range(0, 24, 5):
        ids, dist, ctr = NP.search(gr, g["q_i8"][qi], g["ann_k"], g["ef_search"], 4096)
        n = g["ann_cnt"][qi]
        assert ids == g["ann_ids"][qi, :n].tolist()
        assert dist == g["ann_dist"][qi, :n].tolist()
        assert ctr["n_dist"] == g["n_dist"][qi]


@pytest.mark.parametrize("name", NAMES)
def test_product_host_graph_round_trips_golden_dump(name):
    g = load(name)
    b = HipBridge()
    b.init(dict(dim=g["dim"], m=g["m"], ef=g["ef_construction"], capacity=3, seed=1))
    b.loadBytes(g["dump"])
    assert b.saveBytes() == g["dump"]
