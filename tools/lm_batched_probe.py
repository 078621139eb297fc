#!/usr/bin/env python3
"""Orpheus-3B, one sequence and N sentences side by side, bf16 and packed q4 (bench.py's lm leg without the rest of the bench)."""
import json
import sys
import time

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import torch

import bench
import mlx_swift_audio_amd as m

ctx = m.Context(0)
print(json.dumps(bench.lm_bench(ctx, torch, "orpheus-3b", int(sys.argv[1]) if len(sys.argv) > 1 else 32)))
