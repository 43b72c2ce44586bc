# same-box A/B of library builds: bash tools/ab_libs.sh TAG...   (build_ab/libtolg_TAG.so through TOLG_HIP_LIB, the in-tree
# library before, between and after; three rounds, so that box warm-up and drift show)
for round in 1 2 3; do
for v in intree "$@"; do
  if [ $v = intree ]; then L=""; else L="TOLG_HIP_LIB=$PWD/build_ab/libtolg_$v.so"; fi
  echo -n "round $round $v: "
  env $L timeout -k 10 200 python bench.py --no-cpu-baseline --repeats 7 --allow-lib-override 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), d['config']['kernel_ms_per_step'])"
done
done
