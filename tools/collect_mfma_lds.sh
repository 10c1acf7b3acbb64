#!/bin/bash
# MFMA-pipe and LDS counters of the hot-path kernels (run through gpurun from the repo root; separate --pmc passes with
# --kernel-trace only, as MI355X_MICROARCH.md prescribes):
#   SQ_VALU_MFMA_BUSY_CYCLES  cycles the MFMA pipes were busy (16 per v_mfma_f32_16x16x32_bf16, 32 per 32x32x16)
#   GRBM_GUI_ACTIVE           GPU-active cycles of the launch
#   SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE   extra LDS cycles lost to bank conflicts / all LDS-array cycles
# -> gpurun_out/profiles/<tag>_pmc_mfma_lds.json (per kernel and grid: mean counter values per launch + derived ratios)
set -u
TAG=${1:-r02}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/profiles
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o p -- python $REPO/tools/kernel_bench.py --only '!large' > /dev/null 2>&1
done
python - <<PY
import csv, collections, json
ctrs = ("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE")
data = {}
for c in ctrs:
    agg = collections.defaultdict(list)
    try:
        rows = list(csv.DictReader(open(f"/tmp/pmc_{c}/p_counter_collection.csv")))
    except FileNotFoundError:
        rows = []
    for r in rows:
        name = r["Kernel_Name"]
        if "ivl::" not in name:
            continue
        short = name.split("(")[0].replace("void ", "")
        agg[(short, int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    data[c] = agg
out = {"note": "mean counter value per launch, summed over the device as rocprofv3 reports it.  "
               "SQ_VALU_MFMA_BUSY_CYCLES = 32 per v_mfma_f32_32x32x16_bf16 / 16x16x4_f32, 16 per 16x16x32 (bf16 or fp8).  "
               "GRBM_GUI_ACTIVE = 8 XCDs x active cycles + a ~2e5 profiling offset: not used for ratios; MFMA utilisation = "
               "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel duration x 2.4 GHz) with the duration from "
               "<tag>_bench_kernel_by_grid.csv / bench.py.  lds_conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE", "kernels": {}}
keys = set()
for c in ctrs:
    keys |= set(data[c])
for key in sorted(keys):
    e = {}
    for c in ctrs:
        v = data[c].get(key)
        e[c] = (sum(v) / len(v)) if v else None
    if e["SQ_LDS_BANK_CONFLICT"] is not None and e["SQ_LDS_IDX_ACTIVE"]:
        e["lds_conflict_frac"] = e["SQ_LDS_BANK_CONFLICT"] / e["SQ_LDS_IDX_ACTIVE"]
    out["kernels"][f"{key[0]}@grid{key[1]}"] = e
json.dump(out, open("$OUT/${TAG}_pmc_mfma_lds.json", "w"), indent=1)
for k, e in out["kernels"].items():
    print(f"{k:58s} mfma_busy {e['SQ_VALU_MFMA_BUSY_CYCLES']}  gui {e['GRBM_GUI_ACTIVE']}  lds_conf {e.get('lds_conflict_frac')}")
PY
