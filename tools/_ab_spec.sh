set -e
for v in spec nospec onewave; do
  unset TOLG_LS_NOSPEC TOLG_LS_ONEWAVE
  if [ $v = nospec ]; then export TOLG_LS_NOSPEC=1; fi
  if [ $v = onewave ]; then export TOLG_LS_NOSPEC=1 TOLG_LS_ONEWAVE=1; fi
  for m in "--mode ss" "--line-search"; do
    python3 bench.py $m --steps 20 --warmup 5 --repeats 20 --fresh-regions 0 --no-cpu-baseline > gpurun_out/ab_spec.json 2>gpurun_out/ab_spec.err
    python3 tools/bench_line.py gpurun_out/ab_spec.json "$v $m"
  done
done
