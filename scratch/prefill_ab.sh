# prefill 32 x 1024 wall time with RoPE on q in the flash kernel's query load (default) against the r04 form (the qkv GEMM rotates q): NVR_FLASH_Q_ROPE
cd $GRAFT_REPO_ROOT
for f in 0 1 0 1 0 1; do echo "NVR_FLASH_Q_ROPE=$f"; NVR_FLASH_Q_ROPE=$f timeout 200 python3 scratch/prefill_wall.py 2>&1 | grep "prefill [123]" | tr '\n' ' '; echo; done
