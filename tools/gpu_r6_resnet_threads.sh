#!/bin/bash
# round 6: ResNet-20 throughput, image batches x image threads (contexts on their own streams): do VALU-bound and memory-bound kernels of different contexts overlap?
# usage: OUT=name CONFIGS="8x1 8x2 4x2" IMAGES=32 bash tools/gpu_r6_resnet_threads.sh   (BxT = HCONV_IMAGE_BATCH x HCONV_IMAGE_THREADS)
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${OUT:-r6rt}; mkdir -p $O; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || echo "BUILD FAILED"
W=/tmp/r6rt; mkdir -p $W; cd $W
N=${IMAGES:-32}
PYTHONPATH=$R/tests python -c "import golden.gen_resnet_csv as g; g.write_case('.',3,20,$N)" > /dev/null
for cfg in ${CONFIGS:-8x1 8x2}; do
  b=${cfg%x*}; t=${cfg#*x}
  env HCONV_IMAGE_BATCH=$b HCONV_IMAGE_THREADS=$t timeout 1200 $R/optimal_conv_amd/host/conv resnet 3 20 1 $N false > $O/resnet_$cfg.log 2> $O/resnet_$cfg.err
  echo "== $cfg: rc $?"; grep -E "^Total done|images done" $O/resnet_$cfg.log | tail -5
  rocm-smi --showmeminfo vram 2>/dev/null | grep -i "used" | head -1
done
