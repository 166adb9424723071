#!/bin/bash
set -e
cd "$(dirname "$0")/.."
mkdir -p vk-renderer_amd/lib/variants
# the production flags of cull.hip (vk-renderer_amd/build.py), so that a variant differs from the library by its -D only
COMMON="--offload-arch=gfx950 -fPIC -std=c++17 -fvisibility=hidden -fno-slp-vectorize -mllvm -amdgpu-atomic-optimizer-strategy=None -I include -I vk-renderer_amd/csrc"
while [ $# -gt 0 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc -c ${CULL_SRC:-vk-renderer_amd/csrc/cull.hip} -o /tmp/cull_$name.o $COMMON -O3 -ffp-contract=off $flags
  /opt/rocm/bin/hipcc -shared -o vk-renderer_amd/lib/variants/libvkr_$name.so --offload-arch=gfx950 vk-renderer_amd/lib/deferred.o /tmp/cull_$name.o vk-renderer_amd/lib/vkr_api.o vk-renderer_amd/lib/scene.o vk-renderer_amd/lib/gbuffer.o vk-renderer_amd/lib/vkr_host.o
  echo built $name
done
