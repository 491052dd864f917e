# interleaved flash prefill timing of several library builds on one box: bash scratch/ab_flash.sh <rounds> <lib>...
R=$1; shift
for i in $(seq $R); do
  for L in "$@"; do printf "%-16s " $L; NVR_LIBNVR=$PWD/nano-vllm-rs_amd/$L python tools/prefill_layer_bench.py ${SEQ:-1024} 2>&1 | grep -E "flash"; done
done
