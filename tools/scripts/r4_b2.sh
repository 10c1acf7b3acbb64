cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4a
IVL_DIST_BACKEND=gloo timeout 1500 python bench.py --gpus 2 > gpurun_out/r4a/bench_g2.json 2> gpurun_out/r4a/bench_g2.err
echo rc=$?
tail -c 1500 gpurun_out/r4a/bench_g2.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r4a/bench_g2.json') if l.startswith('{')][-1])
print({k:d[k] for k in ('value','n_gpus','ms_per_step','decode_tok_s')}); print(d['dist'])
PY
