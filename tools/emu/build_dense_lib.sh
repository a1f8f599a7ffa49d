#!/bin/bash
# The dense helper kernels without inline assembly (fused AdamW, add + LayerNorm, the backbone's elementwise tails, token-major
# GroupNorm, decoder glue, the ALIF attention core, the MFMA weight-gradient and expand-GEMM kernels) built for the CPU against the lane-level workgroup model -- same C ABIs, host pointers.
# TEST INFRASTRUCTURE ONLY (tests/test_dense_emulated.py).   usage: tools/emu/build_dense_lib.sh <output.so>
set -e
HERE=$(cd "$(dirname "$0")" && pwd)
ROOT=$(cd "$HERE/../.." && pwd)
OUT=${1:-/tmp/libdense_emu.so}
CXX=${EMU_CXX:-/opt/rocm/lib/llvm/bin/clang++}
TMP=$(mktemp -d)
FLAGS="-x c++ -std=c++20 -O1 -fPIC -pthread -DMSDA_EMU -I$HERE/stub -I$ROOT/include -Wno-unknown-pragmas -Wno-unused-value"
# EMU_SANITIZE=1: AddressSanitizer + UndefinedBehaviorSanitizer build (global-memory accesses of the kernels against the
# host allocator's red zones; run python with LD_PRELOAD=$($CXX -print-file-name=libclang_rt.asan-x86_64.so))
if [ "${EMU_SANITIZE:-0}" = "1" ]; then FLAGS="$FLAGS -g -fno-omit-frame-pointer -fsanitize=address,undefined -shared-libasan"; SAN="-fsanitize=address,undefined -shared-libasan"; fi
# EMU_TSAN=1: ThreadSanitizer build (see build_lib.sh)
if [ "${EMU_TSAN:-0}" = "1" ]; then FLAGS="$FLAGS -g -fno-omit-frame-pointer -fsanitize=thread -shared-libsan"; SAN="-fsanitize=thread -shared-libsan"; fi
pids=()
for f in fused_adamw add_layernorm layernorm_wide elementwise groupnorm_tokens decoder_glue alif_attention window_attention token_gemm expand_gemm; do
    $CXX $FLAGS -c $ROOT/rlipv2_amd/csrc/$f.hip -o $TMP/$f.o &
    pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
$CXX -shared -pthread $SAN $TMP/*.o -o $OUT -Wl,--no-undefined
rm -rf $TMP
