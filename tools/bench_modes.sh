# Secondary bench lines of one tree on one box: bash tools/bench_modes.sh <tag>   -> gpurun_out/<tag>_modes.txt (+ one JSON per mode)
# (SS, merit, rollout='linear' for both solvers, dense inertia blocks, the pendulum, the three line-search forms of those)
tag=${1:-rXX}; out=gpurun_out
: > $out/${tag}_modes.txt
for spec in "headline:" "ss:--mode ss" "merit:--line-search" "linear_ms:--rollout linear" "linear_merit:--rollout linear --line-search" \
            "linear_ss:--rollout linear --mode ss" "dense:--inertia dense" "dense_ss:--inertia dense --mode ss" "dense_merit:--inertia dense --line-search" \
            "pendulum:--workload pendulum" "pendulum_ss:--workload pendulum --mode ss" "pendulum_merit:--workload pendulum --line-search" \
            "dense_linear_ms:--inertia dense --rollout linear" "dense_linear_ss:--inertia dense --rollout linear --mode ss" \
            "dense_linear_merit:--inertia dense --rollout linear --line-search" \
            "pendulum_linear_ss:--workload pendulum --rollout linear --mode ss" "pendulum_linear_merit:--workload pendulum --rollout linear --line-search" \
            "se3_8192:--batch 8192" "drone400_8192:--workload drone400" "se3_256:--batch 256" "al1024:--workload al1024" "so3:--workload so3"; do
  name=${spec%%:*}; flags=${spec#*:}
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --repeats 6 --fresh-regions 0 $flags > $out/${tag}_bench_${name}.json 2> $out/${tag}_bench_${name}.err
  python3 - <<PY >> $out/${tag}_modes.txt
import json
try:
    d = json.load(open("$out/${tag}_bench_${name}.json"))
    c = d["config"]
    print("%-15s %8.1f it/s  %7.4f ms/step  kernels %s  active %.3f%s" % ("$name", d["value"], d["ms_per_step"], {k: round(v, 4) for k, v in c["kernel_ms_per_step"].items()},
          c["active_fraction_at_region_end"], "  INVALID" if "invalid" in c else ""))
except Exception as e:
    print("%-15s FAILED %s" % ("$name", e))
PY
done
cat $out/${tag}_modes.txt
