# fixed-cost A/B: bash tools/ab_horizons.sh TAG   (build_ab/libtolg_TAG.so against the in-tree library at N = 8, 32, 200)
for N in 8 32 200; do for v in intree "$@" intree "$@"; do
  if [ $v = intree ]; then L=""; else L="TOLG_HIP_LIB=$PWD/build_ab/libtolg_$v.so"; fi
  echo -n "N=$N $v: "
  env $L timeout -k 10 200 python bench.py --horizon $N --steps 50 --repeats 7 --no-cpu-baseline --allow-lib-override 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['config']['kernel_ms_per_step']; print('step %.1f us backward %.1f fused %.1f' % (d['ms_per_step']*1e3, k['backward']*1e3, k['rollout']*1e3))"
done; done
