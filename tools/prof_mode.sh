# kernel stats of one bench mode: bash tools/prof_mode.sh <tag> <name> <bench flags...>  -> gpurun_out/<tag>_kstats_<name>.csv
tag=$1; name=$2; shift 2
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/${tag}_stats_${name} -o run --output-format csv -- python3 bench.py --steps 10 --warmup 3 --repeats 2 --fresh-regions 0 --no-cpu-baseline "$@" > gpurun_out/${tag}_stats_${name}.json 2> gpurun_out/${tag}_stats_${name}.err
f=$(ls gpurun_out/${tag}_stats_${name}/*kernel_stats.csv gpurun_out/${tag}_stats_${name}/*/*kernel_stats.csv 2>/dev/null | head -1)
cp "$f" gpurun_out/${tag}_kstats_${name}.csv
python3 - <<PY
import csv
rows = list(csv.DictReader(open("gpurun_out/${tag}_kstats_${name}.csv")))
print("== ${name}")
for r in rows[:12]:
    print("  %-70s calls %5s  avg %9.1f us  total %5.1f %%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
