#!/bin/bash
# A/B builds of one source file: tools/build_variant.sh <name> <file.hip> <extra hipcc flags...>
#   -> mvsnet_amd/variants/lib_<name>.so (the product library with that one object rebuilt); load it with MVS_LIB_PATH.
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; shift 2
mkdir -p mvsnet_amd/variants
obj=/tmp/variant_${name}.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c mvsnet_amd/csrc/$src -o $obj
objs=$(ls mvsnet_amd/csrc/*.o | grep -v "/${src%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o mvsnet_amd/variants/lib_${name}.so $objs $obj
echo mvsnet_amd/variants/lib_${name}.so
