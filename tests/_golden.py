import os

import numpy as np

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NAMES = ["g384_balanced", "g64_fast_ties"]


def load(name):
    z = np.load(os.path.join(HERE, name + ".npz"), allow_pickle=False)
    g = {k: z[k] for k in z.files}
    for k in ("dim", "m", "ef_construction", "topk", "ef_search", "ann_k"):
        g[k] = int(g[k])
    g["preset"] = str(g["preset"])
    g["dump"] = g["dump"].tobytes()
    return g
