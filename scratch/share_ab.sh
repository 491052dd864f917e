#!/bin/bash
# the work-balanced decode attention (default) against the one-workgroup-per-pair launches (NVR_ATTN_SHARE=0), ms per decode step
cd "$(dirname "$0")/.."
A="${@:-33 36 40 44 48 56 65 70 74 97 100 129}"
for i in 1 2; do
echo "== new rule"; python scratch/route_scan.py ${MODEL:-qwen3-0.6b} ${CTX:-1024} $A 2>&1 | cut -c${CUT:-20}-62
echo "== old rule"; NVR_ATTN_SHARE=0 python scratch/route_scan.py ${MODEL:-qwen3-0.6b} ${CTX:-1024} $A 2>&1 | cut -c${CUT:-20}-62
done
