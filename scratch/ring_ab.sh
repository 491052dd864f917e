# r06: configs[4] decode (scratch/bs512.py) with the mid-batch GEMM rings 4 deep (product) and 6 / 6 / 5 deep (libnvr_ring6.so), interleaved
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  echo -n "ring 4:     "; timeout 300 python3 scratch/bs512.py 2>&1 | tail -1
  echo -n "ring 6/6/5: "; NVR_LIBNVR=$GRAFT_REPO_ROOT/nano-vllm-rs_amd/libnvr_ring6.so timeout 300 python3 scratch/bs512.py 2>&1 | tail -1
done
