#!/bin/bash
# r06: dry runs of `bench.py --gpus 2` as two PROCESSES sharing one GPU (NVR_BENCH_SHARED_GPU=1), one per fallback edge of the tensor-parallel phase
# (NRANKS=4 / 8 in the environment: the same with 4 / 8 processes).  Each run's JSON line goes to gpurun_out/r06_dryrun_<name>.json (copied to profiles/ by hand), stderr next to it.
#   default                   fence-free self-test passes: the tensor-parallel line, collective_backend = p2p_fence_free
#   selftest_fence_free_fails NVR_SELFTEST_INJECT=1: every rank's first self-test fails -> p2p_reset -> fenced -> passes: p2p_fenced
#   selftest_p2p_fails        NVR_SELFTEST_INJECT=2: both protocols fail -> RCCL alone (two ranks on ONE device: RCCL refuses; the line falls back to replicas)
#   crc_mismatch              the first child's ranks "disagree" on the tokens -> error line -> the second child (fenced from the start)
#   child_hang                the first child never gets anywhere -> its watchdog ends it -> the second child
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export NVR_BENCH_SHARED_GPU=1 GPU_MAX_HW_QUEUES=16
port=29610
run() {
    name=$1; shift
    port=$((port + 10))
    t0=$(date +%s)
    env "$@" timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node ${NRANKS:-2} --master-addr 127.0.0.1 --master-port $port \
        bench.py --gpus ${NRANKS:-2} --steps 20 --warmup 5 $BENCH_ARGS > gpurun_out/r06_dryrun_$name.json 2> gpurun_out/r06_dryrun_$name.err
    echo "$name: rc $? in $(( $(date +%s) - t0 )) s: $(python - <<PY
import json
try:
    l = [x for x in open("gpurun_out/r06_dryrun_$name.json") if x.startswith("{")][-1]
    d = json.loads(l); c = d.get("config", {})
    print(c.get("parallelism"), "|", c.get("collective_backend"), "| attempt", c.get("tp_attempt"), "|", d.get("value"), d.get("unit"), "|", [t for t in c.get("p2p_protocols_tried", [])], "|", (c.get("earlier_attempts") or d.get("tensor_parallel", {}).get("error", ""))[:1] if not isinstance(c.get("earlier_attempts"), list) else [e[:160] for e in c["earlier_attempts"]])
except Exception as ex:
    print("no line:", ex)
PY
)"
}
for which in "${@:-default selftest_fence_free_fails selftest_p2p_fails crc_mismatch child_hang}"; do
  for w in $which; do
    case $w in
      default) run default X=1 ;;
      selftest_fence_free_fails) run selftest_fence_free_fails NVR_SELFTEST_INJECT=1 ;;
      selftest_p2p_fails) run selftest_p2p_fails NVR_SELFTEST_INJECT=2 ;;
      crc_mismatch) run crc_mismatch NVR_BENCH_INJECT=crc_mismatch ;;
      child_hang) run child_hang NVR_BENCH_INJECT=child_hang ;;
    esac
  done
done
