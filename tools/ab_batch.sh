for B in 256 1024 2048 4096; do
  for v in v5 intree; do
    if [ $v = intree ]; then L=""; else L="TOLG_HIP_LIB=$PWD/build_ab/libtolg_$v.so"; fi
    echo -n "B=$B $v: "
    env $L timeout -k 10 200 python bench.py --batch $B --no-cpu-baseline --repeats 5 --allow-lib-override 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), d['config']['kernel_ms_per_step'])"
  done
done
