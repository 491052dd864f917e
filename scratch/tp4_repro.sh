#!/bin/bash
# the tensor-parallel child of bench.py alone, N rank processes on one GPU (NVR_BENCH_SHARED_GPU=1): where does a step hang?
cd "$(dirname "$0")/.."
export NVR_BENCH_SHARED_GPU=1 GPU_MAX_HW_QUEUES=16 NVR_BENCH_CHILD=1 NVR_BENCH_ATTEMPT=0
N=${1:-4}; shift
env "$@" timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29877 \
    bench.py --gpus $N --steps 20 --warmup 5 --parallel tp --no-cpu-baseline --attn-reps 1 --no-configs3 > gpurun_out/tp_repro.json 2> gpurun_out/tp_repro.err
echo "rc $?"; tail -c 600 gpurun_out/tp_repro.json; grep -v "^\*\*\|OMP_NUM\|amdgpu.ids\|socket.cpp" gpurun_out/tp_repro.err | tail -${TAIL:-25} | cut -c1-400
