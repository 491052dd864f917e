#!/bin/bash
# A/B of configs[4] decode: product library against the fork probe (shared pass on a second stream beside the own-key launch; timing only)
cd "$(dirname "$0")/.."
for i in 1 2 3; do
  python scratch/bs512.py 2>&1 | grep "ms/step" | sed 's/^/product: /'
  NVR_LIBNVR=$PWD/nano-vllm-rs_amd/libnvr_fork.so python scratch/bs512.py 2>&1 | grep "ms/step" | sed 's/^/fork:    /'
done
