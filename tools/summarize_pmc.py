"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into profiles/<tag>_hbm_traffic_pmc.json.

usage: python tools/summarize_pmc.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of a wide coalesced streaming read
(MI355X_MICROARCH.md, HBM section), so the corrected figure doubles it (both are kept)."""
import collections
import csv
import json
import sys
import time


def per_kernel(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    # (round 4: both kernels of the backward sweep have launches that leave at once; a launch counts only if it moved a tenth of
    # what the kernel's largest launch moved, so the averages are per real sweep)
    per = collections.defaultdict(float)
    for r in rows:
        per[(r["Kernel_Name"].split("(")[0], r["Dispatch_Id"])] += float(r["Counter_Value"])
    big = collections.defaultdict(float)
    for (k, _d), v in per.items():
        big[k] = max(big[k], v)
    acc = collections.defaultdict(float)
    n = collections.defaultdict(set)
    for (k, d), v in per.items():
        if v < 0.1 * big[k]:
            continue
        acc[k] += v
        n[k].add(d)
    return {k: (acc[k] / len(n[k]), len(n[k])) for k in acc}


def main():
    f = per_kernel(sys.argv[1], "FETCH_SIZE")
    w = per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {"captured": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime()), "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `python3 bench.py --steps 3 "
                   "--warmup 1 --no-cpu-baseline` (4096x200 SE3); counters are KiB; on gfx950 FETCH_SIZE reports half "
                   "of a wide coalesced streaming read (MI355X_MICROARCH.md HBM section), so the corrected figure "
                   "doubles it", "kernels": {}}
    for k in sorted(set(f) | set(w)):
        fk, nl = f.get(k, (0.0, 0))
        wk, _ = w.get(k, (0.0, 0))
        out["kernels"][k] = {"launches": nl, "FETCH_SIZE_KiB_per_launch": fk, "WRITE_SIZE_KiB_per_launch": wk,
                             "hbm_bytes_per_launch_fetch_doubled": (2 * fk + wk) * 1024.0,
                             "hbm_bytes_per_launch_raw": (fk + wk) * 1024.0}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    for k, v in out["kernels"].items():
        if any(t in k for t in ("k_backward", "k_linearize", "k_rollout")):
            print(k, v)


if __name__ == "__main__":
    main()
