#!/bin/bash
# Builds the MSDA part of the library (msda_api + every MSDA kernel file) for the CPU against the
# lane-level workgroup model: same C ABI (include/rlipv2_msda.h), host pointers instead of device pointers, streams ignored.
# TEST INFRASTRUCTURE ONLY: nothing in rlipv2_amd/ loads this library, and the product never falls back to it.
#   usage: tools/emu/build_lib.sh <output.so>
set -e
HERE=$(cd "$(dirname "$0")" && pwd)
ROOT=$(cd "$HERE/../.." && pwd)
OUT=${1:-/tmp/libmsda_emu.so}
CXX=${EMU_CXX:-/opt/rocm/lib/llvm/bin/clang++}
TMP=$(mktemp -d)
# EMU_DEFINES="-DMSDA_ABLATION": the ablation build of the same sources (the experiment arms behind their environment switches)
FLAGS="-x c++ -std=c++20 -O1 -fPIC -pthread -DMSDA_EMU ${EMU_DEFINES:-} -I$HERE/stub -I$ROOT/include -Wno-unknown-pragmas -Wno-unused-value"
# EMU_SANITIZE=1: AddressSanitizer + UndefinedBehaviorSanitizer build (global-memory accesses of the kernels against the
# host allocator's red zones; run python with LD_PRELOAD=$($CXX -print-file-name=libclang_rt.asan-x86_64.so))
if [ "${EMU_SANITIZE:-0}" = "1" ]; then FLAGS="$FLAGS -g -fno-omit-frame-pointer -fsanitize=address,undefined -shared-libasan"; SAN="-fsanitize=address,undefined -shared-libasan"; fi
# EMU_TSAN=1: ThreadSanitizer build -- every LDS / global access of the lane threads is checked for a happens-before edge (the
# rendezvous and barriers of the model are what provides one): a missing barrier or wave-level ordering in a kernel shows as a
# data race even when the test's result happens to come out right (run python with LD_PRELOAD of libclang_rt.tsan-x86_64.so)
if [ "${EMU_TSAN:-0}" = "1" ]; then FLAGS="$FLAGS -g -fno-omit-frame-pointer -fsanitize=thread -shared-libsan"; SAN="-fsanitize=thread -shared-libsan"; fi
pids=()
for f in msda_api msda_generic msda_quad msda_dest msda_patch msda_sparse msda_rows msda_prep msda_window; do
    $CXX $FLAGS -c $ROOT/rlipv2_amd/csrc/$f.hip -o $TMP/$f.o &
    pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
$CXX -shared -pthread $SAN $TMP/*.o -o $OUT -Wl,--no-undefined
rm -rf $TMP
