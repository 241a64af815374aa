import os, sys, torch
sys.path.insert(0, os.getcwd())
from gdkvm_amd import ops
from tools.conv_probe import ev
torch.backends.cudnn.benchmark = True
cl = dict(memory_format=torch.channels_last)
for (n, hs) in ((512, 56), (160, 128)):
    xs = torch.randn(n, 16, hs, hs, device="cuda").bfloat16().contiguous(**cl)
    w = (torch.randn(64, 16, 4, 4, device="cuda") / 16).bfloat16().contiguous(**cl)
    b = torch.randn(64, device="cuda")
    def two():
        y = torch.nn.functional.conv2d(xs, w, None, 1, 2)
        return ops.bias_relu_maxpool(y.contiguous(**cl)[:, :, :hs, :hs], b)
    a, c = two(), ops.stem_conv_pool(xs, w, b)
    print(n, hs, "max diff", (a.float() - c.float()).abs().max().item(), "two kernels %.1f us   fused %.1f us" % (ev(two), ev(lambda: ops.stem_conv_pool(xs, w, b))), flush=True)
