# fused split-KV merge (last arriver) against the merge launch: single GPU small batches and a tensor-parallel rank's compute
cd $GRAFT_REPO_ROOT
for bs in 1 4 8 16; do for f in 0 1 0 1; do echo -n "FUSED_MERGE=$f "; NVR_ATTN_FUSED_MERGE=$f timeout 120 python3 scratch/bs_step.py $bs 2>&1 | tail -1; done; done
for f in 0 1 0 1; do echo -n "FUSED_MERGE=$f "; NVR_ATTN_FUSED_MERGE=$f timeout 120 python3 scratch/tp_rank_compute.py 8 qwen3-0.6b 2>&1 | tail -1; done
for f in 0 1 0 1; do echo -n "FUSED_MERGE=$f "; NVR_ATTN_FUSED_MERGE=$f timeout 120 python3 scratch/tp_rank_compute.py 8 qwen3-8b 2>&1 | tail -1; done
