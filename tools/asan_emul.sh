#!/bin/bash
# The host emulation of the device kernels (tests/emul/emul_capi.cpp = misp_dd_core.hpp + misp_dd_inplace.hpp built for the host)
# under AddressSanitizer + UndefinedBehaviorSanitizer: builds the emulation library with the sanitizers in place of the regular one,
# runs the emulation suites, restores the regular library.  CPU only; about eight minutes.
cd "$(dirname "$0")/.." || exit 1
LIB=tests/emul/libddo_emul.so
python -c "from tests.emul_binding import build_emul; build_emul()" || exit 1
cp $LIB /tmp/libddo_emul.so.regular
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -Wno-unknown-pragmas -fPIC -shared -o $LIB tests/emul/emul_capi.cpp || exit 1
LD_PRELOAD=$(g++ -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1 \
    python -m pytest tests/test_emulation.py tests/test_emulation_models.py tests/test_emulation_tsptw.py tests/test_emulation_cache.py tests/test_emulation_pooled.py -x -q -p no:cacheprovider "$@"
rc=$?
cp /tmp/libddo_emul.so.regular $LIB; touch $LIB
exit $rc
