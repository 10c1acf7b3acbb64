cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4a
timeout 1500 python bench.py > gpurun_out/r4a/bench_default.json 2> gpurun_out/r4a/bench_default.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r4a/bench_default.json') if l.startswith('{')][-1])
print({k:d[k] for k in ('value','ms_per_step','decode_tok_s','decode_ms_per_token','hot_path_ms_per_step','gemm_ms_per_step','in_scope_share_of_step','hot_path_ms_per_decode_token')})
print(d['roofline'])
print(d['cfg1_4k_prefill_decode']); print(d['cfg3_512k_prefill']); print(d['fp8_e4m3']); print(d['cpu_baseline'])
PY
