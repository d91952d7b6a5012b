#!/bin/bash
# Builds a copy of the library with -fsanitize=address,undefined (host code; the device code is not instrumented) and
# drives every host-only C-ABI entry point (pf_host_*) with random inputs: tests/native/abi_host_fuzz.c.
# Also builds and runs the stand-alone harnesses (shard runner and copy helpers under TSan / ASan, host parsers under ASan + UBSan).
#   tools/sanitize_host.sh [build dir] [iterations]
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="${1:-/tmp/pf_sanitize}"
IT="${2:-5000}"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
mkdir -p "$OUT/obj"
SAN="-fsanitize=address,undefined -fno-omit-frame-pointer -g"
make -C "$ROOT/aliparaformerasr_amd/csrc" -j"$(nproc)" BUILD="$OUT/obj" OUT="$OUT/libpf_asan.so" EXTRA="$SAN" \
  LDFLAGS="-shared -fPIC --offload-arch=gfx950 -fsanitize=address,undefined -Wl,-rpath,/opt/rocm/lib -ldl" 2>&1 | grep -v "Woption-ignored" | grep -i "error" || true
/opt/rocm/lib/llvm/bin/clang -O1 $SAN -I"$ROOT/include" "$ROOT/tests/native/abi_host_fuzz.c" -o "$OUT/abi_host_fuzz" \
  -L"$OUT" -lpf_asan -Wl,-rpath,"$OUT" -Wl,-rpath,/opt/rocm/lib -lm
echo "== C-ABI host entry points (ASan + UBSan)"
ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=halt_on_error=1 "$OUT/abi_host_fuzz" "$IT"
CS="$ROOT/aliparaformerasr_amd/csrc"
for san in thread address; do
  "$HIPCC" -x hip --offload-arch=gfx950 -g -O1 -fsanitize=$san -fno-omit-frame-pointer -std=c++17 -I"$CS" \
    "$ROOT/tests/native/shards_sanitize.cpp" "$CS/shards.cpp" -o "$OUT/shards_$san" -lpthread 2>&1 | grep -v "Woption-ignored\|^$" || true
  echo "== shard runner (-fsanitize=$san)"
  TSAN_OPTIONS=halt_on_error=1 "$OUT/shards_$san" 1000
done
for san in thread address; do
  "$HIPCC" -x hip --offload-arch=gfx950 -g -O1 -fsanitize=$san -fno-omit-frame-pointer -std=c++17 -I"$CS" \
    "$ROOT/tests/native/copycrew_sanitize.cpp" "$CS/copycrew.cpp" -o "$OUT/copycrew_$san" -lpthread 2>&1 | grep -v "Woption-ignored\|^$" || true
  echo "== copy helpers of the staged uploads (-fsanitize=$san)"
  TSAN_OPTIONS=halt_on_error=1 "$OUT/copycrew_$san" 60
done
"$HIPCC" -x hip --offload-arch=gfx950 $SAN -O1 -std=c++17 -I"$CS" "$ROOT/tests/native/host_fuzz.cpp" "$CS/hostutil.cpp" -o "$OUT/host_fuzz" 2>&1 | grep -v "Woption-ignored\|^$" || true
mkdir -p "$OUT/scratch"
echo "== host parsers (ASan + UBSan)"
UBSAN_OPTIONS=halt_on_error=1 "$OUT/host_fuzz" "$IT" "$OUT/scratch"
