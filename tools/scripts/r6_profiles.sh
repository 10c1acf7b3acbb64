#!/bin/bash
# Round-6 evidence run (through gpurun from the repo root): rocprofv3 kernel trace + stats of bench.py, PMC traffic / MFMA / LDS passes
# over tools/kernel_bench.py, a default bench.py line, and the FULL configs[3] sequence on this GPU (524,288 tokens in 128 calls).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/profiles
bash tools/collect_profiles.sh r06 > gpurun_out/profiles_r06.log 2>&1
bash tools/collect_mfma_lds.sh r06 >> gpurun_out/profiles_r06.log 2>&1
python bench.py > gpurun_out/profiles/r06_bench_default.json 2> gpurun_out/profiles/r06_bench_default.err
python bench.py --cfg3-tokens 524288 --no-cfg1 --no-fp8 --no-cpu-baseline --no-kernel-timing --steps 8 --warmup 2 --decode-steps 8 \
    > gpurun_out/profiles/r06_bench_cfg3_full_512k.json 2> gpurun_out/profiles/r06_bench_cfg3_full_512k.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/profiles/r06_bench_default.json') if l.startswith('{')][-1])
print({k:d[k] for k in ('value','ms_per_step','decode_tok_s','hot_path_ms_per_step','gemm_ms_per_step')}, d['roofline']['frac'], d['cfg1_4k_prefill_decode']['prefill_ms'], d['cfg3_512k_prefill']['ms_per_call'], d['cpu_baseline']['value'])
c=json.loads([l for l in open('gpurun_out/profiles/r06_bench_cfg3_full_512k.json') if l.startswith('{')][-1])['cfg3_512k_prefill']
print({k:v for k,v in c.items() if k not in ('workload','kernel_ms_in_one_call')})
PY
tail -5 gpurun_out/profiles_r06.log
