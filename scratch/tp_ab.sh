# A/B of one tensor-parallel rank's compute with / without the fused launches; usage: bash scratch/tp_ab.sh
cd $GRAFT_REPO_ROOT
m=qwen3-0.6b; tp=8
for cfg in "0 0 0" "1 0 0" "0 1 0" "0 1 1" "0 1 2" "0 1 4" "0 1 7" "1 1 7" "0 0 0"; do
  set -- $cfg
  echo -n "ATTN_FUSED_MERGE=$1 TP_FUSED=$2 DBG=$3  "; NVR_ATTN_FUSED_MERGE=$1 NVR_TP_FUSED=$2 NVR_DBG=$3 python3 scratch/tp_rank_compute.py $tp $m 2>&1 | tail -1
done
