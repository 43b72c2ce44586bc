# Secondary bench lines + kernel stats of the modes the headline does not cover (round 3): SS backtracking, MS merit
# line search, longer horizons, the drone workload.  usage (repo root, GPU box): bash tools/profile_modes.sh <tag>
set -e
tag=${1:-rXX}
export TMPDIR=/tmp
out=gpurun_out
for spec in "ss:--mode ss" "merit:--line-search" "n400:--horizon 400 --batch 2048" "n955:--horizon 955 --batch 848" "drone400:--workload drone400"; do
  name=${spec%%:*}; flags=${spec#*:}
  python3 bench.py --no-cpu-baseline --repeats 5 $flags > $out/${tag}_bench_${name}.json 2> $out/${tag}_bench_${name}.err
done
for spec in "ss:--mode ss" "merit:--line-search"; do
  name=${spec%%:*}; flags=${spec#*:}
  rocprofv3 --kernel-trace --stats -d $out/${tag}_stats_${name} -o run --output-format csv -- python3 bench.py --steps 10 --warmup 2 --repeats 2 --no-cpu-baseline $flags > $out/${tag}_stats_${name}.json 2> $out/${tag}_stats_${name}.err
done
echo profile_modes done
