#!/bin/bash
# Qwen3-8B bs 48 / 64 x 2048 decode (the 33..64-row routes) + the split-k op tests
cd "$(dirname "$0")/.."
python -m pytest tests/test_kernels_gpu.py -q -x -k "splitk" 2>&1 | tail -3
python scratch/bs_1024.py qwen3-8b:64:2048 qwen3-8b:48:2048 qwen3-8b:32:2048 2>&1 | cut -c1-420
