"""The north-star block on its own: the `inc` double-conv forward (3 -> 64 -> 64 @320x427, train mode, batch 32) of either
engine, as a one-level network (layer_dimensions=[64]: inc + the 1x1 output conv).  Under `rocprofv3 --pmc FETCH_SIZE` /
`--pmc WRITE_SIZE` (profiles/r03_profile.sh) the per-kernel counters of THIS process are the block's real HBM traffic
(profiles/make_inc_traffic.py sums them, leaving out the output conv); alone it prints the block's time from HIP events.
usage (GPU box): PYTHONPATH=. python profiles/inc_block.py [fp32|bf16] [batch] [iterations]"""
import os
import sys

# a one-level network's `inc` is also its LAST unit, whose BatchNorm + ReLU the bf16 engine would fold into the output convolution
# (GSD_BF16_FUSED_OUT) -- outside the measured region.  The block under test must produce its activation, as it does in the real net:
os.environ["GSD_BF16_FUSED_OUT"] = "0"

import torch

from gelslim_depth_amd import synth
from gelslim_depth_amd.models.unet import UNet

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 1
st = synth.make_state(3, 1, [64], 0, "conditioned")
m = UNet(n_channels=3, n_classes=1, layer_dimensions=[64], precision=prec)
m.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()}, strict=True)
m = m.to("cuda").train()
g = torch.Generator(device="cuda")
g.manual_seed(1234)
x = torch.rand((B, 3, 320, 427), device="cuda", generator=g)
eng = m._engine
with torch.no_grad():
    m(x=x)                       # allocates the buffers
    torch.cuda.synchronize()
    eng.region_log = []
    for _ in range(iters):
        m(x=x)
    torch.cuda.synchronize()
ms = [a.elapsed_time(b) for name, a, b in eng.region_log if name == "inc_forward"]
print("inc double-conv forward, %s, batch %d: %.4f ms (HIP events, %d runs)" % (prec, B, sum(ms) / max(len(ms), 1), len(ms)))
