import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from gdkvm_amd.model import GDKVM, GDKVMConfig
torch.manual_seed(0)
dev = torch.device("cuda")
model = GDKVM(GDKVMConfig()).eval().fuse_for_inference().to(dev).to(torch.bfloat16).to(memory_format=torch.channels_last)
f5 = torch.rand(2, 512, 3, 256, 256, device=dev).to(torch.bfloat16)
def rate(fn, it=3):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it
m = model.segment_clip(f5, 32, graph=True)
print("streams env", os.environ.get("GDKVM_SEGMENT_STREAMS"), f"{rate(lambda: model.segment_clip(f5, 32, graph=True)):.3f} ms per 1024 frames", int(m[0].sum()))
