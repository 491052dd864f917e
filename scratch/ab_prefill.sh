# A/B of two library builds on ONE box, interleaved: bash scratch/ab_prefill.sh <libA> <libB> [rounds]
A=$1; B=$2; R=${3:-3}
for i in $(seq $R); do
  for L in $A $B; do echo "== $(basename $L) round $i"; NVR_LIBNVR=$PWD/nano-vllm-rs_amd/$L python tools/prefill_layer_bench.py 2>&1 | grep -E "rope|silu|plain N=4096|resid|flash" | awk '{printf "%s ", $0; print ""}'; done
done
