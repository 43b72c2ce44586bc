export TMPDIR=/tmp
for spec in "ss:--mode ss" "merit:--line-search"; do
  name=${spec%%:*}; flags=${spec#*:}
  rocprofv3 --kernel-trace --stats -d gpurun_out/r03b_stats_${name} -o run --output-format csv -- python3 bench.py --steps 10 --warmup 2 --repeats 2 --no-cpu-baseline $flags > gpurun_out/r03b_stats_${name}.json 2> gpurun_out/r03b_stats_${name}.err
done
python3 - <<'PY'
import csv
for m in ("ss","merit"):
    rows=list(csv.DictReader(open(f"gpurun_out/r03b_stats_{m}/run_kernel_trace.csv")))
    ev=[r for r in rows if "k_rollout" in r["Kernel_Name"] or "k_ls_" in r["Kernel_Name"] or "k_expected" in r["Kernel_Name"]]
    ev.sort(key=lambda r:int(r["Start_Timestamp"]))
    out=[(r["Kernel_Name"].split("::")[1][:14], r["Grid_Size_Y"], round((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)) for r in ev]
    print(m, out[20:44])
PY
