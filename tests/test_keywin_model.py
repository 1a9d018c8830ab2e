"""The CPU model of the lean traversal's result-key structure (tools/keywin_model.py: sorted key window + flat array +
refills + the parallel admission rule of csrc/traverse_lean.hip.hpp) against a plain sorted multiset: same admission
decisions and the same `worst` after every list, over 320 random streams (dense, sparse, tie-heavy, falling keys, ef from 1
to 2500).  The kernel itself is tested on the GPU; this keeps the rule it implements pinned on the CPU."""
import importlib.util
import os


def test_key_window_model_agrees_with_a_sorted_multiset(capsys):
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "keywin_model.py")
    spec = importlib.util.spec_from_file_location("keywin_model", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert (mod.TAKE, mod.MAXIN, mod.SLACK) == (48, 12, 192)      # kLeanTake, kLeanMaxIn, kLeanLowSlack
    mod.main()
    assert "320 streams ok" in capsys.readouterr().out
