#!/bin/bash
# Build the host sources under a sanitizer and run tests against that library (the reference's memcheck job,
# ci/test_cpp_memcheck.sh:67-76).
#   scripts/run_sanitized.sh address [pytest args...]   AddressSanitizer + UBSan   (default tests: the CPU suite's host tests)
#   scripts/run_sanitized.sh thread  [pytest args...]   ThreadSanitizer            (default: the same; on a GPU box pass -m gpu tests
#                                                                                   that race threads: tests/test_method_and_multigpu_gpu.py)
# Exit code: pytest's, or 66 when a sanitizer report was printed.
set -u
SAN=${1:-address}; shift || true
ROOT=$(cd "$(dirname "$0")/.." && pwd)
make -C "$ROOT/cuopt_amd/csrc" sanitize SAN=$SAN >/dev/null || exit 2
LIB="$ROOT/cuopt_amd/lib/libcuopt_san_$SAN.so"
if [ "$SAN" = thread ]; then RT=$(gcc -print-file-name=libtsan.so); PRE="setarch $(uname -m) -R"; else PRE=""; RT="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)"; fi
LOG=$(mktemp -d)/san
TESTS=("$@"); [ ${#TESTS[@]} -eq 0 ] && TESTS=(tests/test_dual_simplex.py tests/test_capi_host.py tests/test_python_api.py -m "not gpu")
cd "$ROOT"
# (thread flavour: ThreadSanitizer needs the address-space layout it was built for -- "unexpected memory mapping" on kernels with more
#  mmap entropy -- hence setarch -R; the interpreter itself is not instrumented: leaks of CPython are not ours to report; the HIP runtime's threads are outside the build)
LD_PRELOAD="$RT" CUOPT_AMD_LIB="$LIB" \
  ASAN_OPTIONS="detect_leaks=0:log_path=$LOG:abort_on_error=0${ASAN_OPTIONS_EXTRA:+:$ASAN_OPTIONS_EXTRA}" UBSAN_OPTIONS="print_stacktrace=1:log_path=$LOG" \
  TSAN_OPTIONS="log_path=$LOG:report_signal_unsafe=0:ignore_noninstrumented_modules=1" \
  $PRE python -m pytest "${TESTS[@]}" -q -x -p no:cacheprovider
rc=$?
if ls $LOG.* >/dev/null 2>&1; then
  echo "---- sanitizer reports ($SAN) ----"; head -c 20000 $LOG.*; exit 66
fi
echo "sanitizer ($SAN): no report; pytest exit code $rc"
exit $rc
