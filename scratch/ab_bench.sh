# A/B of two library builds on ONE box, interleaved: bash scratch/ab_bench.sh <suffixA> <suffixB> [rounds]   ("" = libnvr.so)
a=$1; b=$2; n=${3:-3}
for i in $(seq $n); do for t in "$a" "$b"; do
  export NVR_LIBNVR=$GRAFT_REPO_ROOT/nano-vllm-rs_amd/libnvr$t.so
  python3 bench.py --no-cpu-baseline --no-configs3 --no-prefill-sweep --no-batch-sweep --no-default-engine --no-shared-prefix --no-chain 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib$t', d['ms_per_step'], d['roofline']['frac'], d['prefill']['seconds'])"
done; done
