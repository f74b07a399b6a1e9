#!/usr/bin/env bash
# AddressSanitizer + UBSan build of libhqtick_test.so (host code only: -fno-gpu-sanitize) and the CPU tests that drive the host side of the tick through it:
# the coupled solve (price.cpp, milp.cpp, host_model.cpp with the emulated sweeps), the host stages, the membership / retracting deltas.
#   bash tools/host_asan.sh [pytest arguments; default: tests/test_price.py tests/test_host_stages.py -x -q]
# A finding ends the process at once and pytest's capture swallows what was printed: the reports are also written to $HQTICK_ASAN_DIR/report.<pid>.
set -eu
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
CS="$ROOT/hyperqueue_amd/csrc"
OUT="${HQTICK_ASAN_DIR:-/tmp/hqtick_asan}"
mkdir -p "$OUT"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
pids=()
for f in hqtick.cpp host_model.cpp milp.cpp price.cpp price_shard.cpp debug_capi.cpp price_emul.cpp kernels.hip graph.hip wire.hip block_solve.hip price.hip; do
    extra=""
    case "$f" in price.hip|price_emul.cpp) extra="-ffp-contract=off";; esac
    obj="$OUT/$(echo "$f" | tr . _).o"
    if [ ! -e "$obj" ] || [ "$CS/$f" -nt "$obj" ] || [ -n "$(find "$CS" "$ROOT/include" -name '*.h' -newer "$obj" | head -1)" ]; then
        "$HIPCC" --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -Wno-unused-value -fsanitize=address,undefined -fno-gpu-sanitize -fno-sanitize-recover=undefined -DHQTICK_TEST_HOOKS=1 $extra -c "$CS/$f" -o "$obj" &
        pids+=($!)
    fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -fno-gpu-sanitize -shared-libsan -o "$OUT/libhqtick_test.so" "$OUT"/*.o -ldl
RT="$("$(dirname "$(readlink -f "$HIPCC")")/../lib/llvm/bin/clang" -print-file-name=libclang_rt.asan-x86_64.so)"
cd "$ROOT"
if [ $# -eq 0 ]; then set -- tests/test_price.py tests/test_host_stages.py -x -q; fi
ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:log_path="$OUT/report" UBSAN_OPTIONS=print_stacktrace=1:log_path="$OUT/report" LD_PRELOAD="$RT" HQTICK_TEST_LIB="$OUT/libhqtick_test.so" python -m pytest "$@"
