#!/bin/bash
# tools/build_variant.sh <name> <extra hipcc flags...>: a second build of libmollyhip.so under ab/lib_<name>.so (every translation unit recompiled with
# the extra flags, e.g. -DMHIP_STAMPS=1 for the time-stamp library of tools/gpu_stamps.sh); the in-tree build is left as it was.
name=$1; shift
cd "$(dirname "$0")/../molly.jl_amd/csrc" || exit 1
mkdir -p ../../ab /tmp/mhip_variant_$name
make -j16 EXTRA="$*" OUT=../../ab/lib_$name.so BUILD=/tmp/mhip_variant_$name 2>&1 | grep -E " error|Error "
ls -la ../../ab/lib_$name.so
