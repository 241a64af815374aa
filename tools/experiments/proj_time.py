import os, sys, torch
sys.path.insert(0, os.getcwd())
from gdkvm_amd import ops
from tools.conv_probe import ev
x = torch.randn(25088, 256, device="cuda").bfloat16()
ws = [torch.randn(n, 256, device="cuda").bfloat16() / 16 for n in (64, 64, 256)]
bs = [torch.randn(n, device="cuda").bfloat16() for n in (64, 64, 256)]
wp = ops.pack_rows_weight(torch.cat(ws).float()); b = torch.cat(bs).float()
t3 = ev(lambda: [torch.nn.functional.linear(x, w, bb) for w, bb in zip(ws, bs)])
t1 = ev(lambda: ops.proj_rows(x, wp, b, (64, 64, 256)))
print("three library GEMMs %.1f us   proj_rows %.1f us" % (t3, t1))
