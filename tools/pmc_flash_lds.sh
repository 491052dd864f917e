# LDS counters of the flash prefill kernel alone; usage (GPU box): bash tools/pmc_flash_lds.sh <tag>
tag=$1
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_flash_lds_$tag
rm -rf $out && mkdir -p $out
cd $GRAFT_REPO_ROOT
for set in "SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_LDS_BANK_CONFLICT SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS"; do
  n=$(echo $set | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/$n -o pmc -- python3 tools/flash_bench.py 1024 16 8 6 > $out/$n.log 2>&1
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
find $out -name "*.csv" -size +1M -delete
