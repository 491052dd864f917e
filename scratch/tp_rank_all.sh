cd $GRAFT_REPO_ROOT
for m in qwen3-0.6b qwen3-8b; do for tp in 1 2 4 8; do timeout 200 python3 scratch/tp_rank_compute.py $tp $m 2>&1 | tail -1; done; done
