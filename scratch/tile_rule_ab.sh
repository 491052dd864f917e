#!/bin/bash
# A/B of the tile-height rule (NVR_TILE_RULE=old: a probe switch of the experiment build) over batch sizes, interleaved twice
cd "$(dirname "$0")/.."
A="257 320 321 352 384 385 512 513 576 640 768"
B="129 192 257 321 384 512"
for i in 1 2; do
echo "== new rule"; python scratch/route_scan.py qwen3-0.6b 256 $A 2>&1 | cut -c20-60
echo "== old rule"; NVR_TILE_RULE=old python scratch/route_scan.py qwen3-0.6b 256 $A 2>&1 | cut -c20-60
echo "== 8B new rule"; python scratch/route_scan.py qwen3-8b 256 $B 2>&1 | cut -c20-60
echo "== 8B old rule"; NVR_TILE_RULE=old python scratch/route_scan.py qwen3-8b 256 $B 2>&1 | cut -c20-60
done
