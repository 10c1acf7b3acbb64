cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4a
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1500 python bench.py > gpurun_out/r4a/bench_final.json 2> gpurun_out/r4a/bench_final.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r4a/bench_final.json') if l.startswith('{')][-1])
print({k:d[k] for k in ('value','ms_per_step','decode_tok_s','hot_path_ms_per_step','gemm_ms_per_step')}, d['roofline']['frac'], d['cfg1_4k_prefill_decode']['prefill_ms'], d['cfg3_512k_prefill']['ms_per_call'])
PY
