#!/bin/bash
# Collect the round's profiling evidence on the GPU box (run through gpurun from the repo root):
#   1. rocprofv3 --kernel-trace --stats over a short bench.py run      -> gpurun_out/profiles/<tag>_bench_kernel_stats.csv
#   2. rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, kernel-trace only) over tools/kernel_bench.py
#      -> gpurun_out/profiles/<tag>_pmc_traffic.json  (per-launch HBM-side bytes; FETCH_SIZE doubled on gfx950 as
#      MI355X_MICROARCH.md prescribes; calibrated on add_rmsnorm whose byte count is known exactly)
# Copy the files you want judged into profiles/ afterwards.
set -u
TAG=${1:-r02}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/profiles
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_bench /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -o bench -- \
    python $REPO/bench.py --steps 64 --warmup 4 --decode-steps 32 --no-cpu-baseline > $OUT/${TAG}_bench_under_rocprof.json 2>/dev/null
cp /tmp/prof_bench/bench_kernel_stats.csv $OUT/${TAG}_bench_kernel_stats.csv
# the same trace keyed by (kernel, grid): one row per launch SHAPE, so the 256-token step's launches are not mixed with the
# decode / 4096-token / batched shapes of the same kernel name (avg_ns of gdn_chunk_prepare@32768 + gdn_chunk_scan<2>@49152
# is the prefill step's gdn_chunk launch; bench.py's HIP-event `avg_launch_ms` must agree with it)
python - <<PY
import csv, collections, glob
rows = []
for f in glob.glob("/tmp/prof_bench/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
agg = collections.defaultdict(list)
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    def dim(prefix):
        if prefix in r:
            return int(r[prefix])
        return int(r[prefix + "_X"]) * int(r[prefix + "_Y"]) * int(r[prefix + "_Z"])
    agg[(name, dim("Grid_Size"), dim("Workgroup_Size"))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
with open("$OUT/${TAG}_bench_kernel_by_grid.csv", "w") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "grid_threads", "workgroup_threads", "launches", "avg_ns", "min_ns", "max_ns", "total_ns"])
    for (name, grid, wg), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        w.writerow([name, grid, wg, len(v), round(sum(v) / len(v), 1), min(v), max(v), sum(v)])
PY
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o p -- python $REPO/tools/kernel_bench.py --only '!large' > /dev/null 2>&1
done
python - <<PY
import csv, collections, json
out = {"units": "bytes per launch", "note": "FETCH_SIZE x2 (gfx950 rocprofv3 reports half of a wide coalesced read), "
       "counter unit KiB; WRITE_SIZE as reported x KiB; separate --pmc passes with --kernel-trace only", "kernels": {}}
data = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f"/tmp/pmc_{c}/p_counter_collection.csv")):
        name = r["Kernel_Name"]
        if "ivl::" not in name:
            continue
        short = name.split("(")[0].replace("void ", "")
        agg[(short, int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    data[c] = agg
for key in sorted(data["FETCH_SIZE"]):
    f = data["FETCH_SIZE"][key]
    w = data["WRITE_SIZE"].get(key, [0.0])
    fetch = 2.0 * 1024.0 * (sum(f) / len(f))
    write = 1024.0 * (sum(w) / len(w))
    out["kernels"][f"{key[0]}@grid{key[1]}"] = {"launches_sampled": len(f), "fetch_bytes": fetch, "write_bytes": write,
                                               "hbm_bytes": fetch + write}
json.dump(out, open("$OUT/${TAG}_pmc_traffic.json", "w"), indent=1)
for k, v in out["kernels"].items():
    print(f"{k:60s} fetch {v['fetch_bytes']/1e6:9.2f} MB  write {v['write_bytes']/1e6:8.2f} MB")
PY
ls -la $OUT
