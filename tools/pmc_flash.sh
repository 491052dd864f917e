# counter passes of the flash prefill kernel alone (tools/flash_bench.py); usage (on the GPU box): bash tools/pmc_flash.sh <tag> [L]
tag=$1; L=${2:-1024}
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_flash_$tag
rm -rf $out && mkdir -p $out
cd $GRAFT_REPO_ROOT
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM" "GRBM_GUI_ACTIVE FETCH_SIZE"; do
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
