# same-box A/B of backward-sweep builds (build_ab/libtolg_<tag>.so through TOLG_HIP_LIB): bash tools/ab_k2.sh TAG...
for v in "$@"; do
  echo -n "$v: "; TOLG_HIP_LIB=$PWD/build_ab/libtolg_$v.so timeout -k 10 200 python bench.py --no-cpu-baseline --repeats 3 --allow-lib-override 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), d['config']['kernel_ms_per_step'])"
done
echo -n "in-tree: "; timeout -k 10 200 python bench.py --no-cpu-baseline --repeats 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), d['config']['kernel_ms_per_step'])"
