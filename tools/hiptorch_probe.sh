cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" >/dev/null 2>&1
echo "--- torch alone"; python -c "
import torch; print('avail', torch.cuda.is_available(), 'count', torch.cuda.device_count()); print(torch.version.hip)
x=torch.ones(4,device='cuda'); print(x.sum().item())
import os; print([l.split()[-1] for l in open('/proc/self/maps') if 'amdhip64' in l][:2])"
echo "--- torch first, then hconv"; python -c "
import torch; print('avail', torch.cuda.is_available()); x=torch.ones(4,device='cuda')
import sys; sys.path.insert(0,'.')
from optimal_conv_amd import Context
c=Context([0x80000000080001,0x1ffffffea0001],[0x1fffffffffe00001]); print('ctx ok')
print(sorted(set(l.split()[-1] for l in open('/proc/self/maps') if 'amdhip64' in l)))"
echo "--- hconv first, then torch"; python -c "
import sys; sys.path.insert(0,'.')
from optimal_conv_amd import Context
c=Context([0x80000000080001,0x1ffffffea0001],[0x1fffffffffe00001]); print('ctx ok')
import torch; print('avail', torch.cuda.is_available(), torch.cuda.device_count())
print(sorted(set(l.split()[-1] for l in open('/proc/self/maps') if 'amdhip64' in l)))"
ldd optimal_conv_amd/libhconv.so | grep -i hip; rocminfo | grep -c gfx950; python -c "import torch,os; print(os.path.dirname(torch.__file__)); print([f for f in os.listdir(os.path.join(os.path.dirname(torch.__file__),'lib')) if 'hip' in f][:10])"
