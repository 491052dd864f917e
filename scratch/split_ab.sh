cd $GRAFT_REPO_ROOT
for i in 1 2 3; do for f in 0 1; do echo -n "NVR_STREAM_SPLIT=$f "; NVR_STREAM_SPLIT=$f timeout 200 python3 scratch/tp_rank_compute.py 1 qwen3-8b 2>&1 | tail -1; done; done
