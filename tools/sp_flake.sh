#!/bin/bash
# developer loop: the 2-rank (gloo, one GPU) bench.py run N times; prints the sequence-parallel self-check of each run
export IVL_DIST_BACKEND=gloo IVL_NO_TUNABLEOP=1 HSA_ENABLE_IPC_MODE_LEGACY=0
N=${1:-8}
for i in $(seq 1 $N); do
  python bench.py --gpus 2 --steps 2 --warmup 1 --layers 4 --context 8192 --decode-steps 2 --no-cpu-baseline --no-cfg1 --no-cfg3 --no-fp8 \
    --no-kernel-timing --sp-tokens 1024 2>/dev/null | tail -1 | python -c "import json,sys; s=json.loads(sys.stdin.read())['dist']['sp']; print(s.get('last_token_logits_equal_single_rank_run'), s.get('max_abs_diff'), s.get('error'))"
done
