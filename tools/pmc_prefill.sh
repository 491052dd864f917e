# Counter pass of ONE prefill step of the bench workload (run on the GPU box through gpurun): MFMA-pipe busy fraction per kernel and
# for the whole step.  rocprofv3 --kernel-trace --pmc only (never combined with --stats or other trace domains).
#   bash tools/pmc_prefill.sh          -> gpurun_out/pmc_prefill/pmc_mfma_prefill.json  (copy to profiles/pmc_mfma_prefill_latest.json)
# SQ_VALU_MFMA_BUSY_CYCLES = 16 cycles per v_mfma_f32_16x16x32_f16 summed over the SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs
# (MI355X_MICROARCH.md, DVFS note): cycles = GRBM / 8, busy fraction = BUSY / (1024 SIMDs x cycles), clock = cycles / duration.
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_prefill
rm -rf $out && mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/raw -o pmc -- python3 scratch/prefill_step.py > $out/run.log 2>&1
echo "rc=$?"
python3 - $out <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
cc = glob.glob(f"{out}/raw/**/*counter_collection.csv", recursive=True)
kt = glob.glob(f"{out}/raw/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(cc[0])))
dur = {}
if kt:
    for r in csv.DictReader(open(kt[0])):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
disp = collections.defaultdict(dict)
for r in rows:
    d = disp[r["Dispatch_Id"]]; d["name"] = r["Kernel_Name"]; d[r["Counter_Name"]] = float(r["Counter_Value"])
    if "Start_Timestamp" in r and r.get("End_Timestamp"): d["us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
ids = sorted(disp, key=int)
# the LAST prefill step: dispatches after the last embedding kernel
last_emb = max(i for i, k in enumerate(ids) if "embedding" in disp[k]["name"])
step = [disp[k] | {"id": k} for k in ids[last_emb:]]
agg = collections.defaultdict(lambda: dict(n=0, us=0.0, busy=0.0, cyc=0.0))
for d in step:
    nm = d["name"].split("(")[0][:70]
    a = agg[nm]; a["n"] += 1; a["us"] += d.get("us", dur.get(d["id"], 0.0)); a["busy"] += d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0); a["cyc"] += d.get("GRBM_GUI_ACTIVE", 0.0) / 8
tot_us = sum(a["us"] for a in agg.values()); tot_busy = sum(a["busy"] for a in agg.values()); tot_cyc = sum(a["cyc"] for a in agg.values())
res = {"what": "one prefill step, Qwen3-0.6B fp16 32 x 1024 tokens, rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE (profiled clocks run ~3 % below un-profiled ones)",
       "prefill_step_weighted": round(tot_busy / (1024 * tot_cyc), 4), "kernel_time_ms": round(tot_us / 1e3, 3),
       "mean_clock_ghz": round(tot_cyc / tot_us / 1e3, 3), "kernels": {}}
for nm, a in sorted(agg.items(), key=lambda kv: -kv[1]["us"]):
    if a["cyc"] > 0:
        res["kernels"][nm] = dict(n=a["n"], us_per_launch=round(a["us"] / a["n"], 1), mfma_busy_frac=round(a["busy"] / (1024 * a["cyc"]), 4), clock_ghz=round(a["cyc"] / a["us"] / 1e3, 3))
json.dump(res, open(f"{out}/pmc_mfma_prefill.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
rm -rf $out/raw
