#!/bin/bash
# Times the deferred kernel variants (VKR_DEFERRED_VARIANT = pixels-per-pass*10 + light-unroll).
for v in "$@"; do
  VKR_DEFERRED_VARIANT=$v timeout -k 10 120 python -u bench.py --steps 30 --warmup 10 --no-cull --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('variant $v', round(d['value'],3), 'Gpix/s', round(d['roofline']['kernel_ms'],3), 'ms', round(d['roofline']['frac'],3))" || exit 1
done
