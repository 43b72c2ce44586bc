set -e
export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS -d gpurun_out/ic_pmc -o run --output-format csv -- python3 bench.py --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline > gpurun_out/ic_pmc.log 2>&1
python3 - <<PY
import csv, collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(set)
for r in csv.DictReader(open("gpurun_out/ic_pmc/run_counter_collection.csv")):
    k=r["Kernel_Name"].split("(")[0]
    if not any(t in k for t in ("k_backward","k_rollout_lin")): continue
    acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
for k in acc:
    n=len(cnt[k]); print(k, n, {c:round(v/n) for c,v in acc[k].items()})
PY
