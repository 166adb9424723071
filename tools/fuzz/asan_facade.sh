#!/bin/bash
# Builds the library with AddressSanitizer + UndefinedBehaviorSanitizer on the HOST pass only (device code is
# compiled as usual) together with the two C++ facade programs, into tools/_asan/.  Run them on the GPU box with
#   VKR_TEST_BIN_DIR=$PWD/tools/_asan ASAN_OPTIONS=verify_asan_link_order=0:detect_leaks=0 \
#     python -m pytest tests/test_facade_gpu.py -m gpu -k cpp
# (the Python process keeps the regular library; only the two host programs load the instrumented one).
set -e
cd "$(dirname "$0")/../.."
out=tools/_asan
mkdir -p $out
objs=""
for u in deferred cull scene gbuffer vkr_api vkr_host; do
  /opt/rocm/bin/hipcc -c vk-renderer_amd/csrc/$u.hip -o $out/$u.o --offload-arch=gfx950 -fPIC -std=c++17 -fvisibility=hidden \
    -I include -I vk-renderer_amd/csrc -O1 -g -ffp-contract=off -fno-slp-vectorize -Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined
  objs="$objs $out/$u.o"
done
/opt/rocm/bin/hipcc -shared -o $out/libvkr_hip.so --offload-arch=gfx950 $objs -fsanitize=address -fsanitize=undefined
for name in facade_frame camera_path; do
  /opt/rocm/lib/llvm/bin/clang++ -std=c++17 -O1 -g -fsanitize=address -fsanitize=undefined -I include tests/cpp/$name.cpp -o $out/$name \
    -L $out -lvkr_hip -Wl,-rpath,'$ORIGIN'
done
rm -f $out/*.o
echo "built $out"
