# what the one-shot exchange protocol costs per collective with a real peer on the device: two in-process ranks with and without their exchanges
cd $GRAFT_REPO_ROOT
for m in qwen3-0.6b qwen3-8b; do for i in 1 2; do
  NOCOMM=1 timeout 120 python3 -X faulthandler scratch/tp_inproc_ab.py 2 $m 2>&1 | tail -1
  timeout 120 python3 -X faulthandler scratch/tp_inproc_ab.py 2 $m 2>&1 | tail -1
done; done
