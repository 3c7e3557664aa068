#!/bin/bash
# Host-side sanitizers for the native code: build dtf-stored with ASan + UBSan and run
# the store test-suite against it.  GPU code is never built with sanitizers (not available on the pool).
set -e
cd "$(dirname "$0")/.."
OUT=/tmp/dtf-stored-asan
g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer -o $OUT distributed_tensorflow_amd/native/store_server.cpp
DTF_STORED_EXE=$OUT ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=halt_on_error=1 \
  python -m pytest tests/test_store.py tests/test_three_daemons.py -q -x
echo "sanitizer run clean"
