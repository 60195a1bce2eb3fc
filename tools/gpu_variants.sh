#!/bin/bash
# A/B of compile-time variants: VARIANTS="name:flags;name2:flags2" ; each is built into libhconv.so and benched
set -u
mkdir -p gpurun_out/variants
export TMPDIR=/tmp
IFS=';' read -ra VS <<< "${VARIANTS:-base:}"
for v in "${VS[@]}"; do
  name="${v%%:*}"; flags="${v#*:}"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-unused-result $flags -shared -o optimal_conv_amd/libhconv.so optimal_conv_amd/csrc/hconv.hip 2> gpurun_out/variants/build_$name.log || { echo "BUILD FAILED $name"; tail -3 gpurun_out/variants/build_$name.log; continue; }
  touch optimal_conv_amd/libhconv.so
  ok=$(timeout 300 python -m pytest tests/test_gpu_a_parity.py -x -q -k "digests" 2>&1 | tail -1)
  for st in ${STREAMS:-1 3}; do
    timeout 300 python bench.py --steps 24 --warmup 6 --chunk ${CHUNK:-64} --streams $st --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); k=d['roofline']['kernels']; print('$name','streams',d['config']['ciphertexts_in_flight_per_gpu'],'ms/conv %.3f'%d['ms_per_step'], 'conv/s %.1f'%d['value'], ' '.join('%s=%.3f'%(n.split('_')[0],v['ms_per_conv']) for n,v in sorted(k.items())))"
  done
  echo "   parity: $ok"
done | tee gpurun_out/variants/results.txt
