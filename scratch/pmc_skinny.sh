# FETCH_SIZE / WRITE_SIZE per dispatch of the decode step's split-k GEMM (linear_skinny<1,2,4,4>: o_proj and down_proj alternate), eager steps
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_skinny
rm -rf $out && mkdir -p $out
cd $GRAFT_REPO_ROOT
for ctr in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
tag=$(echo $ctr | tr ' ' '_')
timeout 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $out/$tag -o pmc -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-chain --no-shared-prefix --no-configs3 --no-prefill-sweep --sync-decode --eager > $out/$tag.log 2>&1
f=$(find $out/$tag -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "linear_skinny" in r["Kernel_Name"] and "ILi1ELi2ELi4ELi4" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Dispatch_Id"]))
by = collections.defaultdict(list)
for r in rows: by[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c, v in by.items():
    tail = v[-56:]
    ev, od = tail[0::2], tail[1::2]
    print(c, "dispatches", len(v), "last step: even-position median %.1f  odd-position median %.1f  (grid %s)" % (sorted(ev)[len(ev)//2], sorted(od)[len(od)//2], rows[-1]["Grid_Size"]))
PY
find $out/$tag -name "*.csv" -delete
done
