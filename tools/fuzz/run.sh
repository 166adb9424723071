#!/bin/bash
# Builds the host-side entry points that take untrusted input (the .fscene reader, the scene-table builder) with
# AddressSanitizer + UndefinedBehaviorSanitizer on the host pass only, and runs the two mutation harnesses on the CPU.
# usage: tools/fuzz/run.sh [iterations]        (no GPU needed: nothing here launches a kernel)
set -e
cd "$(dirname "$0")/../.."
n=${1:-200000}
out=$(mktemp -d)
python3 -c "
import sys, importlib
sys.path.insert(0, '.')
open('$out/base.fscene', 'w').write(importlib.import_module('vk-renderer_amd.synth').fscene_text(6))"
for h in fuzz_fscene fuzz_scene; do
  /opt/rocm/bin/hipcc -x hip --offload-arch=gfx950 -std=c++17 -O1 -g \
    -Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined -Xarch_host -fno-sanitize-recover=undefined \
    -I include -I vk-renderer_amd/csrc vk-renderer_amd/csrc/vkr_host.hip tools/fuzz/$h.cpp -o $out/$h \
    -fsanitize=address -fsanitize=undefined 2> $out/$h.build.log || { cat $out/$h.build.log; exit 1; }
done
$out/fuzz_fscene $out/base.fscene $n
$out/fuzz_scene $n
rm -rf $out
