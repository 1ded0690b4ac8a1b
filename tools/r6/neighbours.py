"""Is every op of a workload's plan bit-reproducible whatever kernel of the other sub-batch runs beside it?

The two-branch deploy form at the bench batch; sub-plan 0 runs ops [0, t] for every t while sub-plan 1 loops ONE op -- one per
distinct (kernel symbol, label head, stride signature) of the plan -- on a second stream; op t's own output regions must be the same
bytes in every repetition.  (Round 6 found one instantiation of bottleneck_block_kernel that was not: DESIGN section 7.)
    python tools/r6/neighbours.py x3d_m [reps] [loops]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
from bench import WORKLOADS, make_model, synth_input  # noqa: E402
from pytorchvideo_amd.accelerator import convert_to_deployable_form, transmute_model  # noqa: E402
from pytorchvideo_amd.accelerator.mi355x.conversion import _ingest_inputs  # noqa: E402
from pytorchvideo_amd.utils import synthetic_trained_like_weights  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "x3d_m"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
loops = int(sys.argv[3]) if len(sys.argv) > 3 else 20
batch = WORKLOADS[wl]["batch"]
torch.manual_seed(0)
m, shape = make_model(wl)
multi = isinstance(shape, (list, tuple)) and isinstance(shape[0], (list, tuple))
x = synth_input(shape, batch, 99)
x = [t.cuda().bfloat16() for t in x] if isinstance(x, (list, tuple)) else x.cuda().bfloat16()
synthetic_trained_like_weights(m, synth_input(shape, 2, 7))
m.eval()
transmute_model(m, "mi355x")
dm = convert_to_deployable_form(m, x, dtype=torch.bfloat16, streams=2, use_graph=False)
dm(x)
torch.cuda.synchronize()
parts = list(dm.parts)
s0, s1 = [p._pv_session for p in parts]
n = len(s0.ops)
s0.profile(iters=1)
half = batch // 2 + (batch % 2)
x0 = [t[:half] for t in x] if isinstance(x, list) else x[:half]
is_multi = isinstance(x, list)
seen, neigh = set(), []
for j in range(n):
    lab = s0.ops[j][3]
    key = (s0.op_kernels[j], lab.split("|")[0], lab.rsplit(" ", 1)[-1] if " s" in lab else "")
    if key not in seen:
        seen.add(key)
        neigh.append(j)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def regions(idx):
    f = s0.ops[idx][2]
    out = []
    for key, v in f.items():
        if key in ("y", "y2", "o", "psum", "gate", "yn", "out") and getattr(v, "space", None) == "arena":
            nb = 1 << 14
            if key == "y" and f.get("y_bs") and f.get("B"):
                nb = int(f["B"]) * int(f["y_bs"]) * (4 if f.get("y_f32") else 2)
            out.append((v.off, min(nb, s0.arena_t.numel() - v.off)))
    return out


print("%s batch %d as two branches: %d ops, %d distinct neighbour kernels %s" % (wl, batch, n, len(neigh), [s0.op_kernels[j] for j in neigh]), flush=True)
bad = 0
for t in range(n):
    regs = regions(t)
    if not regs:
        continue
    msgs = []
    for j in neigh:
        snaps = []
        for _ in range(reps):
            with torch.cuda.stream(sb):
                for _ in range(loops):
                    s1.launch(j, j + 1)
            with torch.cuda.stream(sa):
                _ingest_inputs(s0, x0, parts[0]._pv_inputs, is_multi)
                s0.launch(0, t + 1)
            torch.cuda.synchronize()
            snaps.append([s0.arena_t[o:o + nb].clone() for (o, nb) in regs])
        if not all(all(torch.equal(a, b) for a, b in zip(snaps[0], sn)) for sn in snaps[1:]):
            msgs.append("op %d %s" % (j, s0.op_kernels[j]))
    if msgs:
        bad += 1
        print("  op %d %s [%s] varied beside: %s" % (t, s0.ops[t][3], s0.op_kernels[t], msgs), flush=True)
print("%s: %d of %d ops varied beside a neighbour (%d neighbours x %d repetitions each)" % (wl, bad, n, len(neigh), reps), flush=True)
