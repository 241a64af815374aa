import os, sys, torch, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from gdkvm_amd.model import GDKVM, GDKVMConfig
torch.manual_seed(0)
dev = torch.device("cuda")
model = GDKVM(GDKVMConfig()).eval().fuse_for_inference().to(dev).to(torch.bfloat16).to(memory_format=torch.channels_last)
frames = torch.rand(16, 32, 3, 112, 112, device=dev).bfloat16()
g = model.graphed_segment(frames)
torch.cuda.synchronize(); time.sleep(0.5)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(121)]
ev[0].record()
for i in range(120):
    g(frames); ev[i + 1].record()
torch.cuda.synchronize()
ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(120)]
print("streams", g.streams, " ".join(f"{m:.3f}" for m in ms[:30]))
print("mean 0-4 %.3f  5-24 %.3f  25-59 %.3f  60-119 %.3f" % (sum(ms[:5]) / 5, sum(ms[5:25]) / 20, sum(ms[25:60]) / 35, sum(ms[60:]) / 60))
