#!/bin/bash
# Generic counter pass (developer tool; run through gpurun from the repo root):
#   tools/pmc.sh <tag> "<CTR,CTR,...>[;<CTR,...>]" <kernel-name substring> -- <command...>
# One rocprofv3 --pmc run per ';'-separated group (kernel-trace only, as MI355X_MICROARCH.md prescribes); prints the mean
# counter value per launch for every (kernel, grid) whose name contains the substring and writes
# gpurun_out/profiles/<tag>_pmc.json
set -u
TAG=$1; GROUPS_=$2; SUB=$3; shift 4
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/profiles
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
IFS=';' read -ra GR <<< "$GROUPS_"
for grp in "${GR[@]}"; do
  rm -rf /tmp/pmcg_$i
  rocprofv3 --pmc ${grp//,/ } --kernel-trace --output-format csv -d /tmp/pmcg_$i -o p -- "$@" > /tmp/pmcg_$i.log 2>&1 || tail -3 /tmp/pmcg_$i.log
  i=$((i+1))
done
python - "$SUB" "$OUT/${TAG}_pmc.json" $i <<'PY'
import csv, collections, json, sys, glob
sub, outp, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for i in range(n):
    for f in glob.glob(f"/tmp/pmcg_{i}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            if sub not in name:
                continue
            short = name.split("(")[0].replace("void ", "")
            agg[f"{short}@grid{r['Grid_Size']}"][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}
json.dump(out, open(outp, "w"), indent=1)
for k, d in out.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"    {c:32s} {v:16.1f}")
PY
