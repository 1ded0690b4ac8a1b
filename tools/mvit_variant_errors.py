import sys, os, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import test_gpu_models as T
from oracle import functional as OF
from pytorchvideo_amd.models import create_multiscale_vision_transformers as F
for name in ["mvit_b_small", "mvit_v2ish_small", "mvit_bn_small"]:
    g, m, x = T._golden_case(name, F)
    want = T._oracle(m.state_dict(), x, torch.bfloat16, lambda sd, xx: OF.mvit_forward(sd, xx, g["cfg"]))
    want32 = OF.mvit_forward(m.state_dict(), x, g["cfg"])
    dm, xd = T._deploy(m, x, torch.bfloat16)
    got = dm(xd).float().cpu()
    e = lambda a, b: ((a - b).abs().max() / b.abs().max()).item()
    print(name, "bf16 vs quantised oracle %.2e | vs fp32 oracle %.2e | weights alone %.2e | absmax %.1f" % (e(got, want), e(got, want32), e(want, want32), want.abs().max()))
