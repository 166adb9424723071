p() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['roofline']['kernel_ms'],3), 'ms')"; }
for i in 1 2; do
python bench.py --steps 30 --warmup 10 --no-cull --no-cpu-baseline 2>/dev/null | p "cur short"
VKR_LIB_PATH=$PWD/vk-renderer_amd/lib/variants/libvkr_zold.so python bench.py --steps 30 --warmup 10 --no-cull --no-cpu-baseline 2>/dev/null | p "old short"
python bench.py 2>/dev/null | p "cur default"
VKR_LIB_PATH=$PWD/vk-renderer_amd/lib/variants/libvkr_zold.so python bench.py 2>/dev/null | p "old default"
done
