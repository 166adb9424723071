#!/bin/bash
# Times the deferred resolve of the built library at several light counts (4K, rgba8).
for L in "${@:-0 1 13 64 256}"; do
  timeout -k 10 120 python -u bench.py --steps 30 --warmup 10 --lights $L --no-cull --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('L=$L', round(d['value'],3), 'Gpix/s', round(d['roofline']['kernel_ms'],4), 'ms')" || exit 1
done
