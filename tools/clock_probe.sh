cd $GRAFT_REPO_ROOT
rocm-smi --showperflevel --showmaxpower --showclocks 2>&1 | grep -v "^=\|^$" | head -30
(python - <<'PY'
import sys, os, time
sys.path.insert(0, "tools"); sys.path.insert(0, "cpcstoryvisualization-pytorch_amd")
import torch, patch_probe as P
from cpcsv import kernels as K
d, fl, keep = P.case("sub", 120, 4, 2048, 1024)
d.patch = -1
t0 = time.time()
while time.time() - t0 < 12:
    for _ in range(200): K.gemm_nt(d)
    torch.cuda.synchronize()
print("done")
PY
) &
sleep 6
for i in 1 2 3; do rocm-smi --showclocks --showpower --showtemp 2>&1 | grep -i "sclk\|power\|mclk\|fclk\|junction" | head -8; sleep 1.5; done
wait
