"""Per-kernel stats of the decode steps only, from a rocprofv3 kernel_trace.csv: the trace is cut at the
argmax kernels (one per engine step); the last N steps are aggregated and one step's timeline is printed."""
import csv, sys, collections
path, nsteps = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 16
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: n.replace("nvr::k::", "").replace("_ZN3nvr1k", "")[:70]
ends = [i for i, r in enumerate(rows) if "argmax" in r["Kernel_Name"]]
first = ends[-nsteps - 1] + 1
sel = rows[first:ends[-1] + 1]
agg = collections.defaultdict(list)
for r in sel:
    agg[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot_k = sum(sum(v) for v in agg.values())
span = (int(sel[-1]["End_Timestamp"]) - int(sel[0]["Start_Timestamp"])) / 1e3
print(f"{nsteps} decode steps: kernel time {tot_k / nsteps:.1f} us/step, wall span {span / nsteps:.1f} us/step")
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(f"{sum(v) / nsteps:9.1f} us/step  n={len(v) // nsteps:4d}  avg {sum(v) / len(v):7.2f}  min {min(v):7.2f}  max {max(v):7.2f}  {k}")
# wall cost per kernel type = start of the NEXT kernel - start of this one (duration + boundary), over the selected steps
wall = collections.defaultdict(list)
for a, b in zip(sel, sel[1:]):
    if "argmax" in a["Kernel_Name"]:
        continue
    wall[short(a["Kernel_Name"])].append((int(b["Start_Timestamp"]) - int(a["Start_Timestamp"])) / 1e3)
print("start-to-start (duration + boundary) per kernel type:")
for k, v in sorted(wall.items(), key=lambda kv: -sum(kv[1])):
    print(f"{sum(v) / nsteps:9.1f} us/step  n={len(v) // nsteps:4d}  avg {sum(v) / len(v):7.2f}  {k}")
# gaps inside the last step
last = rows[ends[-2] + 1:ends[-1] + 1]
gaps = [(int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3 for a, b in zip(last, last[1:])]
print(f"last step: {len(last)} kernels, sum of gaps {sum(gaps):.1f} us, mean gap {sum(gaps) / len(gaps):.2f} us")
# prefill step = kernels between the last weight-fill and the first argmax
emb = [i for i, r in enumerate(rows) if "embedding_kernel" in r["Kernel_Name"]]
pre = rows[emb[0]:ends[0] + 1]
agg2 = collections.defaultdict(list)
for r in pre:
    agg2[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print(f"prefill step: {len(pre)} kernels, kernel time {sum(sum(v) for v in agg2.values()) / 1e3:.2f} ms")
for k, v in sorted(agg2.items(), key=lambda kv: -sum(kv[1]))[:8]:
    print(f"{sum(v) / 1e3:9.3f} ms  n={len(v):4d}  avg {sum(v) / len(v):9.1f} us  {k}")
