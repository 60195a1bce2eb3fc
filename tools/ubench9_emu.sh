#!/bin/bash
# CPU check of ubench9's indexing under the fiber emulator (tests/kernel_emu): the one-launch transform must equal the emulated library's hc_lv_ntt word for word.
# usage: bash tools/ubench9_emu.sh [level] [images]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
make -s -C $R/tests/kernel_emu $R/tests/kernel_emu/_build/libhconv_emu.so
g++ -O2 -std=c++17 -DHC_EMU -Wno-unknown-pragmas -I$R/tests/kernel_emu -x c++ $R/tools/ubench9.hip $R/tests/kernel_emu/hip_emu.cpp -L$R/tests/kernel_emu/_build -lhconv_emu -Wl,-rpath,$R/tests/kernel_emu/_build -o $R/tools/_variants/ubench9_emu
$R/tools/_variants/ubench9_emu ${1:-3} ${2:-2} 0
