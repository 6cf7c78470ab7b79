#!/bin/bash
# Runs the emulation suites (tests/test_emulation*.py: the device source of both engines compiled for the host) under the switches that
# make two classes of device-only bugs visible on the CPU:
#   DDO_EMU_ORDER = 0 / 1 / 2   the "threads" of a PAR block run ascending / descending / in a fixed pseudo-random permutation: a block in
#                               which a thread reads what another thread of the SAME block wrote (a data race on the GPU) changes its result;
#   DDO_EMU_POISON = byte       what workspace memory holds before the kernel writes it (hipMalloc does not clear): a read of memory that
#                               was never written shows with some byte or other (0xCD, the default, reads as a large negative int).
# Usage: tools/diag/emu_order.sh [pytest arguments]
cd "$(dirname "$0")/../.." || exit 1
rc=0
for o in 0 1 2; do
  DDO_EMU_ORDER=$o python -m pytest tests/test_emulation.py tests/test_emulation_models.py tests/test_emulation_tsptw.py tests/test_emulation_cache.py tests/test_emulation_pooled.py -q -n 8 "$@" | tail -n 2 || rc=1
done
for p in 0x00 0x01 0x7F 0xFF; do
  DDO_EMU_POISON=$p python -m pytest tests/test_emulation.py tests/test_emulation_models.py tests/test_emulation_tsptw.py tests/test_emulation_cache.py tests/test_emulation_pooled.py -q -n 8 "$@" | tail -n 2 || rc=1
done
exit $rc
