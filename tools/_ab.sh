set -e
for v in base prev; do
  if [ $v = base ]; then unset TOLG_HIP_LIB; else export TOLG_HIP_LIB=build_ab/libtolg_$v.so; fi
  for m in "--mode ss" "--line-search" ; do
    python3 bench.py $m --steps 20 --warmup 5 --repeats 20 --fresh-regions 0 --no-cpu-baseline --allow-lib-override > gpurun_out/ab.json 2>gpurun_out/ab.err
    python3 tools/bench_line.py gpurun_out/ab.json "$v $m"
  done
done
