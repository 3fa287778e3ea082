#!/usr/bin/env python3
"""Diagnostic: the fwd+loss+bwd step of tools/bench_small.py at one size, eager or replayed from a
hipGraph (argv[1] = eager|graph, argv[2] = lattice side n, argv[3] = M) -- run under rocprofv3
--kernel-trace --stats to compare kernel durations between the two modes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools.bench_small import make_case, timed  # noqa: E402
from tools.bench_small import compare  # noqa: E402

mode, n, M = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
print(mode, compare(n, M, "fwd+loss+bwd", modes=(mode,)), flush=True)
