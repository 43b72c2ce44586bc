# Round profile on the GPU box: kernel trace + stats, HBM traffic counters (separate passes), default bench.
# usage (from the repo root): bash tools/profile_round.sh <tag>      -> writes gpurun_out/<tag>_*
set -e
tag=${1:-rXX}
export TMPDIR=/tmp
out=gpurun_out
python3 bench.py > $out/${tag}_bench_default.json 2> $out/${tag}_bench_default.err
rocprofv3 --kernel-trace --stats -d $out/${tag}_stats -o run --output-format csv -- python3 bench.py --steps 20 --warmup 3 --repeats 10 --fresh-regions 2 --no-cpu-baseline > $out/${tag}_bench_under_rocprof.json 2> $out/${tag}_stats.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/${tag}_pmc_fetch -o run --output-format csv -- python3 bench.py --steps 3 --warmup 2 --repeats 1 --fresh-regions 0 --no-cpu-baseline > $out/${tag}_pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/${tag}_pmc_write -o run --output-format csv -- python3 bench.py --steps 3 --warmup 2 --repeats 1 --fresh-regions 0 --no-cpu-baseline > $out/${tag}_pmc_write.log 2>&1
python3 tools/summarize_pmc.py $out/${tag}_pmc_fetch/run_counter_collection.csv $out/${tag}_pmc_write/run_counter_collection.csv $out/${tag}_hbm_traffic_pmc.json
echo profile_round done
