"""Search/build presets and the ef / annK derivation of the reference host.

Values: reference src/config.ts:23-54 (resolvePreset); derivation:
src/mini-vector-db.ts:568-576.
"""
from __future__ import annotations

MAX_EF = 4096  # assembly/hnsw.ts:30

PRESETS = {
    "fast": dict(m=12, ef_construction=150, baseEfSearch=80, rerankMultiplier=40, maxAnnK=2048, resultsCap=2048),
    "balanced": dict(m=16, ef_construction=300, baseEfSearch=140, rerankMultiplier=120, maxAnnK=4096,
                     resultsCap=4096),
    "accurate": dict(m=24, ef_construction=600, baseEfSearch=200, rerankMultiplier=250, maxAnnK=4096,
                     resultsCap=4096),
}


def resolve_preset(mode: str) -> dict:
    """resolvePreset: anything but "fast"/"accurate" is "balanced" (config.ts:43-44)."""
    return dict(PRESETS.get(mode, PRESETS["balanced"]))


def derive_search_params(topK, preset: dict, wasm_max_ef: int = MAX_EF):
    """(topK, efSearch, annK) as searchMany computes them (mini-vector-db.ts:568-576)."""
    tk = max(1, int(topK if topK is not None else 10))
    ef = max(int(preset["baseEfSearch"]), tk * 2)
    ann_k = max(tk, tk * int(preset["rerankMultiplier"]))
    ann_k = min(ann_k, int(preset["maxAnnK"]))
    if wasm_max_ef > 0:
        ann_k = min(ann_k, wasm_max_ef)
    return tk, ef, ann_k
