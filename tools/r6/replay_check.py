"""Are two replays of the deploy form bit-identical?  X3D-M at the bench batch, per block_stages setting (run on the GPU box)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
from bench import make_model, synth_input  # noqa: E402
from pytorchvideo_amd import _lib as L  # noqa: E402
from pytorchvideo_amd.accelerator import convert_to_deployable_form, transmute_model  # noqa: E402
from pytorchvideo_amd.utils import synthetic_trained_like_weights  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "x3d_m"
from pytorchvideo_amd.accelerator.mi355x import tuning  # noqa: E402
tuning.OPTIONS["arena_margin"] = int(sys.argv[2]) if len(sys.argv) > 2 else 0
if len(sys.argv) > 3:      # "joint=0": one graph per sub-batch on its own stream instead of one graph with two branches
    tuning.OPTIONS["split_joint_graph"] = bool(int(sys.argv[3]))
CASES = ((4, 0, 0), (28, 16, 16)) if len(sys.argv) > 2 else ((0, 0, 0), (16, 0, 0), (8, 0, 0), (4, 0, 0), (0, 16, 0), (0, 0, 16), (28, 16, 16))
torch.manual_seed(0)
m, shape = make_model(wl)
synthetic_trained_like_weights(m, synth_input(shape, 2, 7))
m.eval()
transmute_model(m, "mi355x")
x = synth_input(shape, 32, 99).cuda().bfloat16()
for streams in (2, 1):
    for full, ab, gc in CASES:
        L.tune(block_stages=full, block_stages_ab=ab)
        dm = convert_to_deployable_form(m, x, dtype=torch.bfloat16, streams=streams)
        outs = [dm(x).float().cpu() for _ in range(6)]
        neq = [int((o != outs[0]).sum().item()) for o in outs[1:]]
        print("%s streams=%d block_stages full=%d ab=%d gc=%d: differing logits per replay vs the first: %s" % (wl, streams, full, ab, gc, neq), flush=True)
        del dm
        torch.cuda.empty_cache()
