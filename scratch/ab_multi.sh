# interleaved comparison of several library builds on one box: bash scratch/ab_multi.sh <rounds> <lib>...
R=$1; shift
for i in $(seq $R); do
  for L in "$@"; do echo "== $L round $i"; NVR_LIBNVR=$PWD/nano-vllm-rs_amd/$L python tools/prefill_layer_bench.py 2>&1 | grep -E "rope|silu|plain N=4096|resid"; done
done
