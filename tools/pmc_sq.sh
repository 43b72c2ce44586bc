# SQ counter passes over the bench; usage: bash tools/pmc_sq.sh <tag> [busy|mix|pipes ...]   (default: busy)
#   busy  : wave cycles, VALU busy, waits, VALU instruction count  -> gpurun_out/<tag>_sq_counters.json
#   mix   : instruction mix (fp64 fma / mul / add / trans, SALU, LDS, VMEM writes) -> <tag>_sq_mix.json
#   pipes : cycles the wave spends issuing VMEM / SALU / LDS / misc instructions, FIFO-full stalls -> <tag>_sq_pipes.json
# One --pmc pass per set (counters only fit eight at a time), each in its own run with --kernel-trace only.
set -e
tag=${1:-rXX}; shift || true
passes=${@:-busy}
export TMPDIR=/tmp
for p in $passes; do
case $p in
  busy)  ctr="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES"; out=sq_counters;;
  mix)   ctr="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SALU SQ_INSTS_LDS"; out=sq_mix;;
  pipes) ctr="SQ_WAVES SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; out=sq_pipes;;
  *) echo "unknown pass $p"; exit 2;;
esac
rocprofv3 --kernel-trace --pmc $ctr -d gpurun_out/${tag}_pmc_$p -o run --output-format csv -- python3 bench.py --steps 3 --warmup 2 --repeats 1 --fresh-regions 0 --no-cpu-baseline > gpurun_out/${tag}_pmc_$p.log 2>&1
python3 - <<PY
import csv, collections, json
rows=[r for r in csv.DictReader(open("gpurun_out/${tag}_pmc_$p/run_counter_collection.csv"))]
# the backward sweep is two kernels since round 4 and both have launches that leave at once (the fast sweep hands a group back on
# entry, the full kernel behind it finds nothing flagged): a launch counts only if its heaviest counter reaches a tenth of the
# kernel's largest -- the averages are per REAL sweep; the dropped launches are counted in launches_dropped
big=collections.defaultdict(float); per=collections.defaultdict(float)
for r in rows:
    k=r["Kernel_Name"].split("(")[0]; v=float(r["Counter_Value"])
    per[(k,r["Dispatch_Id"])]=max(per[(k,r["Dispatch_Id"])],v); big[k]=max(big[k],v)
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(set); dropped=collections.defaultdict(set)
for r in rows:
    k=r["Kernel_Name"].split("(")[0]
    if not any(t in k for t in ("k_rollout","k_backward","k_linearize","k_rollout_lin")): continue
    if per[(k,r["Dispatch_Id"])] < 0.1*big[k]: dropped[k].add(r["Dispatch_Id"]); continue
    acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
import time
out={"captured":time.strftime("%Y-%m-%dT%H:%M:%SZ",time.gmtime()),"note":"rocprofv3 --kernel-trace --pmc $ctr (one pass) over python3 bench.py --steps 3 --warmup 2 --repeats 1 --fresh-regions 0 --no-cpu-baseline (4096x200 SE3); per launch; SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* are quad-cycles summed over waves","kernels":{}}
for k in acc:
    n=len(cnt[k]); d={c:v/n for c,v in acc[k].items()}
    d["launches"]=n; d["launches_dropped"]=len(dropped[k])
    wc=d.get("SQ_WAVE_CYCLES",0); w=d.get("SQ_WAVES",0)
    if wc and "SQ_ACTIVE_INST_VALU" in d:
        d["valu_busy_frac"]=d["SQ_ACTIVE_INST_VALU"]/wc; d["wait_any_frac"]=d["SQ_WAIT_ANY"]/wc; d["wait_inst_frac"]=d["SQ_WAIT_INST_ANY"]/wc
    if w:
        for c in list(d):
            if c.startswith("SQ_INSTS_"): d[c[3:].lower()+"_per_wave"]=d[c]/w
    if wc and "$p"=="pipes":
        for c in list(d):
            if c.startswith("SQ_INST_CYCLES_") or c.startswith("SQ_ACTIVE_INST_"): d[c[3:].lower()+"_frac"]=d[c]/wc
    out["kernels"][k]=d
    print(k, {x:round(y,3) for x,y in d.items() if "frac" in x or "per_wave" in x})
json.dump(out,open("gpurun_out/${tag}_$out.json","w"),indent=1)
PY
done
