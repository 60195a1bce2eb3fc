R=$GRAFT_REPO_ROOT; W=/tmp/chk; mkdir -p $W; cd $W
python -c "import sys; sys.path.insert(0,'$R'); import __graft_entry__ as g; g.build()" > /dev/null 2>&1
PYTHONPATH=$R/tests python -c "import golden.gen_conv_csv as g; [g.write_case('test_conv_data',5,1,i) for i in range(1)]; [g.write_case('test_conv_data',3,0,i) for i in range(1)]"
for env in "HCONV_GPUS=2" "HCONV_GPUS=2 HCONV_IMAGE_BATCH=2" "HCONV_ASYNC_ALLOC=0 HCONV_GPUS=2" "HCONV_OPWISE=1"; do
  env $env HCONV_SKIP_BL=1 timeout 600 $R/optimal_conv_amd/host/conv convReLU 5 1 1 > out.log 2>&1; echo "[$env] convReLU 5 1 1 exit $? : $(grep -E 'MED Prec' out.log | tail -1) $(grep -ci 'error\|panic' out.log)"
done
timeout 600 $R/optimal_conv_amd/host/conv convReLU 3 0 1 > out.log 2>&1; echo "convReLU 3 0 1 (both columns) exit $? : $(grep -E 'MED Prec' out.log | tr '\n' ' ')"
