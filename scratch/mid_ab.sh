# configs[4] decode (512 sequences behind one shared prompt): ms per step with / without the 96-row gate_up tiles
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  echo -n "128-row tiles: "; NVR_NO_SILU96=1 timeout 300 python3 scratch/bs512.py 2>&1 | tail -1
  echo -n " 96-row tiles: "; timeout 300 python3 scratch/bs512.py 2>&1 | tail -1
done
