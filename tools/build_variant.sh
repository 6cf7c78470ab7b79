#!/bin/bash
# tools/build_variant.sh <name> <extra compiler flags ...>: builds ddo_amd/_build_<name>/libddo_hip.so with EXTRA flags (e.g.
# -DDDO_G8_DENSE=4 -DDDO_WS_ONLY=7).  Objects of translation units the flags do not touch are taken from ddo_amd/_build.
name=$1; shift
cd "$(dirname "$0")/../ddo_amd/csrc" || exit 1
B=../_build_$name
mkdir -p $B/obj
for o in ddo_hip_engine kernels_core_lds kernels_core_glb host_solver misp_io; do cp -p ../_build/obj/$o.o $B/obj/ 2>/dev/null && touch $B/obj/$o.o; done
make -s BUILD=$B EXTRA="$*" 2>&1 | grep -v "warning" | tail -3
ls -la $B/libddo_hip.so
