cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4a
for i in 1 2; do
python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-cfg1 --no-cfg3 --no-fp8 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
k=d['kernels']
print('decode tok/s', round(d['decode_tok_s'],1), 'ms/token', round(d['decode_ms_per_token'],4), 'gdn_decode_step in-step', [v.get('in_step_us') for n,v in k.items() if 'gdn_decode_step' in n], 'swa_decode', [v.get('in_step_us') for n,v in k.items() if n=='swa_decode'])
"
done
