#!/usr/bin/env python3
"""Run the randomised GPU parity test (tests/test_gpu_parity.py::test_random_shapes_against_the_oracle)
over a range of seeds beyond the 16 the suite runs: python tools/fuzz_gpu.py [first] [count]."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_parity as t  # noqa: E402


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    gpu = torch.device("cuda:0")
    bad, t0 = [], time.time()
    for seed in range(first, first + count):
        try:
            t.test_random_shapes_against_the_oracle(gpu, seed)
        except AssertionError as exc:
            bad.append(seed)
            print(f"seed {seed}: {exc}", flush=True)
        if (seed - first) % 20 == 19:
            print(f"... {seed - first + 1} seeds, {len(bad)} failures, {time.time() - t0:.0f} s", flush=True)
    print(f"{count} seeds from {first}: {len(bad)} failures {bad}")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
