cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4a
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -k "bench_two_ranks or plain_rmsnorm or norm_linear or without_output_gate or sequence_parallel" 2>&1 | tail -15 > gpurun_out/r4a/pytest4.log
cat gpurun_out/r4a/pytest4.log
timeout 1200 python bench.py > gpurun_out/r4a/bench_default.json 2> gpurun_out/r4a/bench_default.err
tail -c 3000 gpurun_out/r4a/bench_default.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r4a/bench_default.json') if l.startswith('{')][-1])
print({k:d[k] for k in ('value','ms_per_step','decode_tok_s','hot_path_ms_per_step','gemm_ms_per_step','in_scope_share_of_step')})
print(d['roofline'])
for k,v in d['kernels'].items():
    print(k, v.get('ms'), v.get('in_step_us'), v.get('bound'), round(v.get('frac',0),4))
PY
