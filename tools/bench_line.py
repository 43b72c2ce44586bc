"""One-line summary of a bench.py JSON line:  python tools/bench_line.py FILE [label]"""
import json
import sys

d = json.load(open(sys.argv[1]))
c = d["config"]
f = c.get("fresh_solve")
print("%-16s %8.1f it/s  %.4f ms/step  kernels %s  active %.3f%s%s" % (
    sys.argv[2] if len(sys.argv) > 2 else "", d["value"], d["ms_per_step"], {k: round(v, 4) for k, v in c["kernel_ms_per_step"].items()},
    c["active_fraction_at_region_end"], ("  fresh %.4f ms/step" % f["median_ms_per_step"]) if f else "", "  INVALID" if "invalid" in c else ""))
