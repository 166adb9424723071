#!/bin/bash
for f in vk-renderer_amd/lib/variants/libvkr_*.so; do
  VKR_LIB_PATH=$PWD/$f timeout -k 10 120 python -u bench.py --steps 60 --warmup 10 --no-cpu-baseline --width 256 --height 256 --lights 4 --no-callers 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['cull']; print('$f', 'atomic', round(c['atomic']['ms_per_call']*1e3,1),'us', round(c['atomic']['roofline']['frac'],3), ' det', round(c['deterministic']['ms_per_call']*1e3,1),'us', round(c['deterministic']['roofline']['frac'],3))" || exit 1
done
