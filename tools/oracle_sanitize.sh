#!/bin/bash
# CPU only: the C oracle built with AddressSanitizer + UndefinedBehaviorSanitizer (GPU sanitizers are not available on the
# pool), run against every golden vector (tests/test_oracle_golden.py).  The regular oracle build is restored afterwards.
#   usage: bash tools/oracle_sanitize.sh
set -e
cd "$(dirname "$0")/.."
make -s -C oracle
cp -r oracle/build /tmp/oracle_build_backup.$$
trap 'rm -rf oracle/build; mv /tmp/oracle_build_backup.$$ oracle/build; touch oracle/build/*.so' EXIT
gcc -O1 -g -std=c11 -fPIC -shared -fopenmp -ffp-contract=off -fno-math-errno -fsanitize=address,undefined \
    -fno-sanitize-recover=undefined -mfma -mavx2 oracle/eamrl_oracle.c -o oracle/build/liboracle.so -lm
cp oracle/build/liboracle.so oracle/build/liboracle_generic.so
LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 python -m pytest tests/test_oracle_golden.py -x -q
