"""Build tuning variants of libmvdb_hip.so: python tools/build_variants.py name:DEF=V,DEF=V ..."""
import os
import sys
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from minivectordb_amd.build import build  # noqa: E402


def one(spec):
    name, _, defs = spec.partition(":")
    return build(defines=[d for d in defs.split(",") if d], out=f"libmvdb_hip_{name}.so")


if __name__ == "__main__":
    with ThreadPoolExecutor(4) as ex:
        for p in ex.map(one, sys.argv[1:]):
            print(p)
