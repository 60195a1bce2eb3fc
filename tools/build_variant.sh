#!/bin/bash
# build a variant of libhconv.so with extra compile flags: tools/build_variant.sh NAME "-DHC_JOB_FAST=0 ..."  -> tools/_variants/libhconv_NAME.so
# (select it with HCONV_LIB=<path>; experiments only — the product is optimal_conv_amd/libhconv.so built by __graft_entry__.build())
set -eu
cd "$(dirname "$0")/.."
mkdir -p tools/_variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-value -Wno-unused-result $2 -shared -o tools/_variants/libhconv_$1.so optimal_conv_amd/csrc/hconv.hip
echo built tools/_variants/libhconv_$1.so
