# CPU-only: the oracle (test infrastructure) under AddressSanitizer + UBSan, driven by the CPU test-suite.
# (GPU sanitizers are not available on this pool; the product library has no CPU build.)
#   bash tools/oracle_sanitize.sh
set -e
cd "$(dirname "$0")/.."
cp oracle/libjets_oracle.so /tmp/libjets_oracle.so.bak
gcc -O1 -g -ffp-contract=off -fno-fast-math -fcx-limited-range -fPIC -std=gnu11 -fopenmp -fsanitize=address,undefined \
    -fno-omit-frame-pointer -shared -o oracle/libjets_oracle.so oracle/jets_oracle.c -lm
LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) ASAN_OPTIONS=detect_leaks=0 \
    python -m pytest tests/test_oracle_pinning.py tests/test_golden.py tests/test_oracle_lsqr.py tests/test_scalar_types.py tests/test_known_answers.py -q -m "not gpu" -p no:cacheprovider || true
cp /tmp/libjets_oracle.so.bak oracle/libjets_oracle.so
touch oracle/libjets_oracle.so
