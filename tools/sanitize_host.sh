#!/bin/bash
# Sanitizer builds of the HOST side of liblitcoder_hip.so (CPU only -- sanitizers never run on the GPU): csrc/lc_upload.hip
# (staging threads, slot ring, coordinator, condition variables, NUMA probing) and csrc/lc_core.hip (host casts, 2-D copies,
# event timers) compiled as C++ by g++ against the HIP stand-in of tools/sanitize/hip_stub/, once with
# -fsanitize=address,undefined and once with -fsanitize=thread, and run through tools/sanitize/host_upload_test.cpp.
#   tools/sanitize_host.sh [outdir] [reps]      logs: <outdir>/sanitize_asan_ubsan.log, <outdir>/sanitize_tsan.log
set -u
HERE=$(cd "$(dirname "$0")" && pwd)
ROOT=$(dirname "$HERE")
OUT=${1:-/tmp/lc_sanitize}
REPS=${2:-2}
mkdir -p "$OUT"
SRC="$ROOT/litcoder_core_amd/csrc/lc_upload.hip $ROOT/litcoder_core_amd/csrc/lc_core.hip"
COMMON="-std=c++17 -g -O1 -fno-omit-frame-pointer -ffp-contract=off -Wall -Wno-unknown-pragmas -Wno-unused-function -pthread -I$HERE/sanitize/hip_stub"
rc=0
for mode in asan_ubsan tsan; do
  if [ $mode = asan_ubsan ]; then SAN="-fsanitize=address,undefined -fno-sanitize-recover=undefined"; else SAN="-fsanitize=thread"; fi
  BIN="$OUT/host_upload_test_$mode"
  LOG="$OUT/sanitize_$mode.log"
  {
    echo "# $(g++ --version | head -1); flags: $SAN $COMMON"
    objs=""
    for f in $SRC; do
      o="$OUT/$(basename $f .hip)_$mode.o"
      g++ $SAN $COMMON -x c++ -c "$f" -o "$o" || exit 3
      objs="$objs $o"
    done
    g++ $SAN $COMMON "$HERE/sanitize/hip_stub.cpp" "$HERE/sanitize/host_upload_test.cpp" $objs -o "$BIN" || exit 3
    echo "# run: $BIN $REPS"
    ASAN_OPTIONS=detect_leaks=1:abort_on_error=0 UBSAN_OPTIONS=print_stacktrace=1 TSAN_OPTIONS=halt_on_error=0:second_deadlock_stack=1 \
      LITCODER_AMD_UPLOAD_NO_AFFINITY= "$BIN" "$REPS"
    echo "# exit code $?"
  } > "$LOG" 2>&1
  if ! grep -q "^# exit code 0$" "$LOG" || grep -q "WARNING: ThreadSanitizer\|ERROR: AddressSanitizer\|runtime error:\|ERROR: LeakSanitizer" "$LOG"; then
    echo "sanitize_host: $mode FAILED (see $LOG)"; rc=1
  else
    echo "sanitize_host: $mode clean ($(grep -c . "$LOG") log lines)"
  fi
done
exit $rc
