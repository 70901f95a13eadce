#!/bin/sh
# Builds the ThreadSanitizer hammer: csrc/api.hip compiled for the HOST only, against stand-ins for the HIP runtime and the
# device pipelines (no GPU code, never run on a GPU box).  usage: tools/tsan/build.sh <output binary>
set -e
cd "$(dirname "$0")/../.."
OUT=${1:-/tmp/eip_tsan_hammer}
CLANG=/opt/rocm/lib/llvm/bin/clang++
FLAGS="-O1 -g -std=c++17 -DEIP_HOST_ONLY -fsanitize=thread -fno-omit-frame-pointer -mbmi2 -madx -fPIC -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include"
T=$(mktemp -d)
/opt/rocm/bin/hipcc -x hip --cuda-host-only $FLAGS -c blst_eip2537_amd/csrc/api.hip -o $T/api.o
$CLANG $FLAGS -c tools/tsan/hip_stub.cpp -o $T/hip_stub.o
$CLANG $FLAGS -c tools/tsan/engine_stub.cpp -o $T/engine_stub.o
$CLANG $FLAGS -c tools/tsan/hammer.cpp -o $T/hammer.o
$CLANG -fsanitize=thread $T/api.o $T/hip_stub.o $T/engine_stub.o $T/hammer.o -lpthread -o $OUT
rm -rf $T
echo built $OUT
