# memory-side counter passes of the flash prefill kernel alone (tools/flash_bench.py): bash tools/pmc_flash_mem.sh <tag> [L]
tag=$1; L=${2:-1024}
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_flashmem_$tag
rm -rf $out && mkdir -p $out
cd $GRAFT_REPO_ROOT
for set in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum"; do
  n=$(echo $set | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/$n -o pmc -- python3 tools/flash_bench.py $L 16 8 6 > $out/$n.log 2>&1
  f=$(find $out/$n -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:60]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k, d in acc.items():
    if "flash" not in k: continue
    print(k)
    for c, v in d.items(): print(f"   {c:28s} {v / cnt[(k, c)]:16.1f} per launch")
PY
done 2>&1 | tee $out/summary.txt
