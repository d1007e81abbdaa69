#!/bin/bash
# tools/oracle_sanitize.sh -- run the oracle's golden-vector tests with oracle.c built under ASan + UBSan (CPU only;
# GPU sanitizers are not available on the pool).  Restores the normal build afterwards.
set -e
cd "$(dirname "$0")/.."
cp oracle/liboracle.so /tmp/liboracle_backup.so
trap 'cp /tmp/liboracle_backup.so oracle/liboracle.so' EXIT
gcc -O1 -g -std=c11 -fopenmp -fPIC -shared -fsanitize=address,undefined -fno-omit-frame-pointer -o oracle/liboracle.so oracle/oracle.c
LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) ASAN_OPTIONS=detect_leaks=0 \
    python -m pytest tests/test_oracle_golden.py -x -q
