"""Longer run of tests/test_fuzz_parity.py's layered-silhouette cases: random merged scenes, the one-pass layered call against
one pass per object, bit for bit.  python scripts/fuzz_layered.py [first] [count]"""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
import test_fuzz_parity as T

first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
bad = 0
for seed in range(first, first + count):
    try:
        T.test_random_merged_scenes_layered_silhouettes_equal_per_object_passes(torch.device("cuda:0"), seed)
    except AssertionError as e:
        bad += 1
        print("seed", seed, "MISMATCH", str(e)[:200])
print(f"{count} random merged scenes (layered silhouettes) from seed {first}: {bad} with a mismatch")
