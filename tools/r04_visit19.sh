#!/bin/bash
export TMPDIR=/tmp
PA_EMD_TAIL_FROM=8 python - <<'PY'
import torch
from patchaugnet_amd import emd_module
g = torch.Generator().manual_seed(11)
p1, p2 = (torch.rand(16, 4096, 3, generator=g).cuda() for _ in range(2))
d, a = emd_module.emdModule()(p1, p2, 0.02, 1024)
torch.cuda.synchronize()
PY
