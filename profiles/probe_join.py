"""How long does the bf16 backward's main stream wait for the side stream (weight gradients) at its joins?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gelslim_depth_amd import synth
from gelslim_depth_amd.models.unet import UNet
from gelslim_depth_amd.train import TrainStep
from gelslim_depth_amd.engine_bf16 import UNetEngineBF16
DIMS = [64, 128, 256, 512, 1024]
m = UNet(3, 1, layer_dimensions=DIMS, precision="bf16").to("cuda").train()
step = TrainStep(m, lr=1e-3, weight_decay=1e-6)
x, t = synth.make_batch(32, 320, 427, 1)
xd, td = torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda()
log = []
orig = UNetEngineBF16._join_side
def timed(self):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); orig(self); e1.record(); log.append((e0, e1))
for _ in range(3): step(xd, td)
UNetEngineBF16._join_side = timed
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(5): step(xd, td)
b.record(); torch.cuda.synchronize()
waits = [e0.elapsed_time(e1) for e0, e1 in log]
per = len(waits) // 5
print("step %.2f ms; joins per step %d; waited per step %.3f ms; per join (last step): %s" % (a.elapsed_time(b) / 5, per, sum(waits) / 5, ["%.3f" % w for w in waits[-per:]]))
