"""Which op's output differs first between two replays?  X3D-M bench batch as two sub-batch branches, every arena buffer kept (debug
plan: tuning arena_guards > 0, no buffer re-use), the arenas compared after each of several replays.  Run on the GPU box."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from bench import make_model, synth_input  # noqa: E402
from pytorchvideo_amd import _lib as L  # noqa: E402
from pytorchvideo_amd.accelerator import convert_to_deployable_form, transmute_model  # noqa: E402
from pytorchvideo_amd.accelerator.mi355x import tuning  # noqa: E402
from pytorchvideo_amd.utils import synthetic_trained_like_weights  # noqa: E402

torch.manual_seed(0)
m, shape = make_model("x3d_m")
synthetic_trained_like_weights(m, synth_input(shape, 2, 7))
m.eval()
transmute_model(m, "mi355x")
x = synth_input(shape, 32, 99).cuda().bfloat16()
tuning.OPTIONS["arena_guards"] = 256
full = int(sys.argv[1]) if len(sys.argv) > 1 else 4
L.tune(block_stages=full, block_stages_ab=0)
dm = convert_to_deployable_form(m, x, dtype=torch.bfloat16, streams=2)
sessions = dm._pv_sessions
ref = None
for rep in range(6):
    out = dm(x)
    torch.cuda.synchronize()
    snap = [s.arena_t.clone() for s in sessions]
    if ref is None:
        ref = snap
        continue
    for bi, (a, b, s) in enumerate(zip(ref, snap, sessions)):
        diff = (a != b).nonzero().flatten()
        if diff.numel() == 0:
            print("replay %d branch %d: arena identical" % (rep, bi), flush=True)
            continue
        first = int(diff[0].item())
        # map byte offsets to the ops that WRITE them (field 'y', in plan order)
        hits = {}
        for idx, (kind, dcls, fields, label, ab, fl) in enumerate(s.ops):
            y = fields.get("y")
            if y is None or getattr(y, "space", None) != "arena":
                continue
            hits[idx] = (y.off, label)
        offs = sorted((off, idx, label) for idx, (off, label) in hits.items())
        import bisect
        starts = [o for o, _, _ in offs]
        def owner(byte):
            k = bisect.bisect_right(starts, byte) - 1
            return offs[k] if k >= 0 else None
        owners = {}
        for byte in diff[:: max(1, diff.numel() // 2000)].tolist():
            o = owner(byte)
            if o:
                owners.setdefault((o[1], o[2]), 0)
                owners[(o[1], o[2])] += 1
        firsts = sorted(owners.items())[:6]
        print("replay %d branch %d: %d bytes differ; first at %d; earliest ops with differing outputs: %s" % (rep, bi, diff.numel(), first, firsts), flush=True)
