# SQ counter pass (wave cycles, VALU busy, waits, instruction counts) over the bench; usage: bash tools/pmc_sq.sh <tag>
set -e
tag=${1:-rXX}
export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES -d gpurun_out/${tag}_pmc_sq -o run --output-format csv -- python3 bench.py --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline > gpurun_out/${tag}_pmc_sq.log 2>&1
python3 - <<PY
import csv, collections, json
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(set)
for r in csv.DictReader(open("gpurun_out/${tag}_pmc_sq/run_counter_collection.csv")):
    k=r["Kernel_Name"].split("(")[0]
    if not any(t in k for t in ("k_rollout","k_backward","k_linearize","k_rollout_lin")): continue
    acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
out={"note":"rocprofv3 --kernel-trace --pmc (8 SQ counters, one pass) over python3 bench.py --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline (4096x200 SE3); per launch; SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* are quad-cycles summed over waves","kernels":{}}
for k in acc:
    n=len(cnt[k]); d={c:v/n for c,v in acc[k].items()}
    d["launches"]=n
    wc=d.get("SQ_WAVE_CYCLES",0)
    if wc:
        d["valu_busy_frac"]=d["SQ_ACTIVE_INST_VALU"]/wc; d["wait_any_frac"]=d["SQ_WAIT_ANY"]/wc; d["wait_inst_frac"]=d["SQ_WAIT_INST_ANY"]/wc
        d["valu_insts_per_wave"]=d["SQ_INSTS_VALU"]/d["SQ_WAVES"]
    out["kernels"][k]=d
    print(k, {x:round(y,3) for x,y in d.items() if "frac" in x or "per_wave" in x})
json.dump(out,open("gpurun_out/${tag}_sq_counters.json","w"),indent=1)
PY
