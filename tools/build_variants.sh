#!/bin/bash
# Builds tuning variants of the library: tools/build_variants.sh name "flags" [name "flags"]...
set -e
cd "$(dirname "$0")/.."
mkdir -p vk-renderer_amd/lib/variants
COMMON="--offload-arch=gfx950 -fPIC -std=c++17 -fvisibility=hidden -I include -I vk-renderer_amd/csrc"
while [ $# -gt 0 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc -c vk-renderer_amd/csrc/deferred.hip -o /tmp/deferred_$name.o $COMMON -O3 -ffp-contract=off -fno-slp-vectorize $flags
  /opt/rocm/bin/hipcc -shared -o vk-renderer_amd/lib/variants/libvkr_$name.so --offload-arch=gfx950 /tmp/deferred_$name.o vk-renderer_amd/lib/cull.o vk-renderer_amd/lib/vkr_api.o vk-renderer_amd/lib/scene.o vk-renderer_amd/lib/gbuffer.o vk-renderer_amd/lib/vkr_host.o
  echo built $name
done
