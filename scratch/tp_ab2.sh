cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
: > gpurun_out/tp_ab2.log
for tp in ${TPS:-2}; do
for cfg in "0 0" "0 7" "1 7" "0 7" "1 7" "0 0"; do
  set -- $cfg
  NVR_TP_FUSED=$1 NVR_DBG=$2 timeout 100 python3 -X faulthandler scratch/tp_inproc_ab.py $tp ${MODEL:-qwen3-0.6b} >> gpurun_out/tp_ab2.log 2>&1 || echo "rc=$? cfg=$cfg tp=$tp" >> gpurun_out/tp_ab2.log
done
done
tail -30 gpurun_out/tp_ab2.log
