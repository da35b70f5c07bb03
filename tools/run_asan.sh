#!/bin/bash
# CPU only. Runs the oracle's own tests and the oracle-vs-reference tests with the AddressSanitizer + UBSan
# builds of oracle/oracle.cpp and of the reference harness (oracle/_asan/, `make -C oracle asan`), and the host
# classes' CPU-only pieces (3x3 SVD / Kabsch, PLY reader) built with the same flags.
# Output: the sanitizer reports, deduplicated, on stdout; the full log in $1 (default /tmp/icpx_asan.log).
set -u
cd "$(dirname "$0")/.."
LOG=${1:-/tmp/icpx_asan.log}
make -C oracle asan >/dev/null || exit 1
make -C icp_amd/csrc asan >/dev/null || exit 1
export ICPX_LIB_DIR=$PWD/icp_amd/_asan
ASAN_LIB=$(g++ -print-file-name=libasan.so)
UBSAN_LIB=$(g++ -print-file-name=libubsan.so)
export ICPX_ORACLE_SAN=1
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=0:abort_on_error=0
export UBSAN_OPTIONS=print_stacktrace=0:halt_on_error=0
export OMP_NUM_THREADS=4
LD_PRELOAD="$ASAN_LIB $UBSAN_LIB" python -m pytest tests/test_oracle_ref.py tests/test_oracle_cpu.py tests/test_abi_and_host_cpu.py -x -q -s -p no:cacheprovider >"$LOG" 2>&1
echo "pytest exit: $?"
tail -3 "$LOG"
echo "--- distinct sanitizer reports ---"
grep -E "runtime error|ERROR: AddressSanitizer" "$LOG" | sed -E 's/0x[0-9a-f]+/0x…/g' | sort | uniq -c | sort -rn
