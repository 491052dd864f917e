#!/bin/bash
# An experiment build of ONE kernel file next to the product library: bash tools/build_variant.sh <name> <kernel.hip> "<-D flags>"
# -> nano-vllm-rs_amd/libnvr_<name>.so (all other objects are the product build's; run with NVR_LIBNVR=<path>).  Experiment builds are
# git-ignored and removed by `make clean-variants`.
set -e
name=$1; src=$2; flags=$3
cd "$(dirname "$0")/../nano-vllm-rs_amd/csrc"
base=$(basename $src)
tmp=build/variant_$name; mkdir -p $tmp
CXX="/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-function -Wno-unused-result -Wno-unused-value --offload-arch=gfx950"
nofma=""
case $base in elementwise.hip|linear.hip|linear_decode.hip|linear_stream.hip|gemm_tiled.hip|gemm256.hip|comm_p2p.hip) nofma="-ffp-contract=off";; esac
$CXX $nofma $flags -x hip -c kernels/$base -o $tmp/$base.o &
$CXX $nofma $flags -DNVR_BF16 -x hip -c kernels/$base -o $tmp/$base.bf16.o &
wait
objs=$(ls build/*.o build/kernels/*.o | grep -v "/$base\.o$" | grep -v "/$base\.bf16\.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../libnvr_$name.so $objs $tmp/$base.o $tmp/$base.bf16.o -ldl -Wl,-rpath,/opt/rocm/lib
echo built ../libnvr_$name.so
