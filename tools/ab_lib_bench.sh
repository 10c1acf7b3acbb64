#!/bin/bash
# Same-box A/B of the default bench line with another build of the library (developer use):
#   tools/ab_lib_bench.sh tools/ab/libivl_other.so [rounds]      -> gpurun_out/ab_lib/{base,other}_N.json, one summary line each
lib=$1; rounds=${2:-2}
mkdir -p gpurun_out/ab_lib
for r in $(seq 1 $rounds); do
  python bench.py > gpurun_out/ab_lib/base_$r.json 2>/dev/null
  python -c "
import runpy, sys
from infinitevl_amd import _lib
_lib.load('$lib')
sys.argv = ['bench.py']
runpy.run_path('bench.py', run_name='__main__')" > gpurun_out/ab_lib/other_$r.json 2>/dev/null
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/ab_lib/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, 'unreadable', e); continue
    ks = d['kernels']
    pick = {n.split('(')[0]: round(v.get('ms', 0) * 1e3, 2) for n, v in ks.items() if n.split('(')[0] in ('add_rmsnorm', 'silu_mul', 'rmsnorm_swish_gate', 'counter_add', 'gdn_chunk_fused', 'swa_prefill')}
    print(f.split('/')[-1], 'tok/s', round(d['value'], 1), 'ms/step', round(d['ms_per_step'], 4), 'decode ms/token', round(d['decode_ms_per_token'], 4),
          'gdn in-step us', round(d['roofline']['avg_launch_ms'] * 1e3, 2), 'hot-path ms/step', round(d['hot_path_ms_per_step'], 4), pick)
PY
