# r06: HBM bytes per decode kernel of Qwen3-8B on one GPU (scratch/bench_8b.py, bs 32 x 2048): one rocprofv3 --pmc FETCH_SIZE pass (--kernel-trace only beside it).
# gfx950: bytes = FETCH_SIZE KiB x 1024 x 2 for 16-B-per-lane streaming reads (MI355X_MICROARCH.md, HBM).  usage (GPU box): bash tools/pmc_8b.sh
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_8b
rm -rf $out && mkdir -p $out
cd $GRAFT_REPO_ROOT
NVR_NO_EXIT=1 timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out -o pmc -- python3 scratch/bench_8b.py > $out/run.log 2>&1
echo "rc=$?"; tail -1 $out/run.log
python3 - $out <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
f = glob.glob(f"{out}/**/*counter_collection.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
last_prefill = max((i for i, r in enumerate(rows) if "gemm256" in r["Kernel_Name"] or "flash_prefill" in r["Kernel_Name"]), default=-1)
agg = collections.defaultdict(list)
for r in rows[last_prefill + 1:]:
    if r["Counter_Name"] == "FETCH_SIZE": agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items(), key=lambda kv: -sorted(kv[1])[len(kv[1]) // 2])[:10]:
    v.sort()
    print(f"{v[len(v) // 2] * 2048 / 1e6:9.2f} MB read (median, x2 corrected)  n={len(v):5d}  {k[:110]}")
PY
find $out -name "*.csv" -delete
