#!/bin/bash
# tools/build_variant.sh <name> <extra hipcc flags...>: a build of libmollyhip.so whose packed pair loop (forces_uniform.hip) is compiled
# with extra -D switches, for A/B timing (tools/force_ab.py); the in-tree build is restored afterwards
name=$1; shift
cd "$(dirname "$0")/../molly.jl_amd/csrc" || exit 1
mkdir -p ../../ab
touch forces_uniform.hip; make -j8 EXTRA="$*" 2>&1 | grep -E " error|Error " ; cp ../libmollyhip.so ../../ab/lib_$name.so
touch forces_uniform.hip; make -j8 2>&1 | grep -E " error|Error "
