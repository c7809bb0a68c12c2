#!/bin/bash
# CPU AddressSanitizer run of the host-only JPEG decoder over a mutated corpus (GPU sanitizers are not available on the
# pool; the decoder is host code).  Usage: scratch/asan_jpeg/run.sh <dir with .jpg files>
set -e
here=$(cd "$(dirname "$0")" && pwd); root=$(cd "$here/../.." && pwd); out=${TMPDIR:-/tmp}/icl_asan; mkdir -p "$out"
/opt/rocm/bin/hipcc -x hip --cuda-host-only -O1 -g -fsanitize=address -std=c++17 -I"$root/include" -c "$root/imageclust_amd/csrc/jpeg_decode.hip" -o "$out/jd.o"
/opt/rocm/lib/llvm/bin/clang++ -fsanitize=address -g "$here/harness.cpp" "$here/fail.cpp" "$out/jd.o" -o "$out/harness"
ASAN_OPTIONS=detect_leaks=0 "$out/harness" "$1"/*.jpg
