# headline attention launch (B=32, H=16, KVH=8, ctx 1044) on this tree and on the scratch/_pre worktree (an earlier commit), interleaved
for i in 1 2 3; do
  H=16 KVH=8 timeout 100 python scratch/attn_tp_shape.py | sed 's/^/HEAD /'
  (cd scratch/_pre && H=16 KVH=8 timeout 100 python scratch/attn_tp_shape.py | sed 's/^/pre  /')
done
