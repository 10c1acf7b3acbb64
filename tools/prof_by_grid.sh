#!/bin/bash
# rocprofv3 --kernel-trace of a command, summarised per (kernel, grid): tools/prof_by_grid.sh <out.csv> <filter-substr> -- <command...>
set -u
OUTCSV=$1; FILTER=$2; shift 3
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pbg
rocprofv3 --kernel-trace --output-format csv -d /tmp/pbg -o t -- "$@" > /tmp/pbg_stdout.log 2>&1
python - "$REPO/$OUTCSV" "$FILTER" <<'PY'
import csv, collections, glob, sys
rows = []
for f in glob.glob("/tmp/pbg/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
agg = collections.defaultdict(list)
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    def dim(prefix):
        if prefix in r:
            return int(r[prefix])
        return int(r[prefix + "_X"]) * int(r[prefix + "_Y"]) * int(r[prefix + "_Z"])
    agg[(name, dim("Grid_Size"), dim("Workgroup_Size"))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
with open(sys.argv[1], "w") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "grid_threads", "workgroup_threads", "launches", "avg_ns", "min_ns", "max_ns", "total_ns"])
    for (name, grid, wg), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        w.writerow([name, grid, wg, len(v), round(sum(v) / len(v), 1), min(v), max(v), sum(v)])
        if sys.argv[2] in name:
            print(f"{name[:70]:70s} grid {grid:8d} wg {wg:5d} n {len(v):5d} avg {sum(v)/len(v)/1e3:9.2f} us  min {min(v)/1e3:8.2f}")
PY
