set -e
for m in "--mode ss" "--line-search" "--line-search --rollout linear" "--mode ss --rollout linear"; do
  python3 bench.py $m --steps 20 --warmup 5 --repeats 20 --fresh-regions 0 --no-cpu-baseline > gpurun_out/ab.json 2>gpurun_out/ab.err
  python3 tools/bench_line.py gpurun_out/ab.json "$m"
done
