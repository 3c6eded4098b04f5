#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer over the host-side C++ (class Particlebot in its
# HostOnly engine, .cfg loader, C wrappers, threaded ensemble construction, frame writer) on the CPU
# -- GPU sanitizers are not available on the pool.  Needs no GPU; libparticlebot_hip.so must be built.
#   bash tools/sanitize/run.sh          (prints sanitizer reports, if any, and "asan driver done")
#   bash tools/sanitize/run.sh thread   the same driver under ThreadSanitizer (producer pool, shared placements)
set -eu
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
SRC=$ROOT/particlerobotsimulations_amd/csrc
OUT=${TMPDIR:-/tmp}/pb_sanitize
mkdir -p "$OUT"
SAN=address,undefined
[ "${1:-}" = "thread" ] && SAN=thread
g++ -std=c++17 -O1 -g -fsanitize=$SAN -fno-omit-frame-pointer -ffp-contract=off -pthread \
    -I"$ROOT/include" -I"$SRC" -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include \
    "$ROOT/tools/sanitize/host_driver.cpp" "$SRC/particlebot.cpp" "$SRC/pb_config.cpp" "$SRC/pb_capi.cpp" \
    -o "$OUT/drv" -L"$ROOT/particlerobotsimulations_amd/lib" -lparticlebot_hip \
    -Wl,-rpath,"$ROOT/particlerobotsimulations_amd/lib"
cd "$OUT"
ASAN_OPTIONS=detect_leaks=1:halt_on_error=0 UBSAN_OPTIONS=print_stacktrace=1 TSAN_OPTIONS=halt_on_error=0 ./drv "$ROOT" 2>&1 | grep -v '^[0-9. -]*$'
