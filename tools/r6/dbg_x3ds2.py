import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle.weights import reference_style_fill, seeded_input
from pytorchvideo_amd import _lib as L
from pytorchvideo_amd.models import hub
from pytorchvideo_amd.accelerator import convert_to_deployable_form, transmute_model
from pytorchvideo_amd.accelerator.mi355x import tuning
src = hub.x3d_s(); reference_style_fill(src, 3); src.eval()
x = seeded_input((2, 3, 13, 160, 160), 3)
transmute_model(src, "mi355x")
xd = x.cuda().bfloat16()
tuning.OPTIONS["arena_guards"] = 256
outs = {}
for ab in (0, 0x10):
    L.tune(block_stages=0, block_stages_ab=ab)
    dm = convert_to_deployable_form(src, xd, dtype=torch.bfloat16)
    dm(xd); torch.cuda.synchronize()
    s = dm._pv_session
    rec = []
    for idx, (kind, dcls, f, label, _, _) in enumerate(s.ops):
        head = label.split("|")[0]
        if head in ("conv_c", "se_gate", "conv_ab.fused+se", "conv_b.dw+se", "conv_ab.dw+se", "conv_b"):
            y = f.get("y") or f.get("gate")
            if head == "se_gate":
                g = f["gate"]; n = int(f["B"]) * int(f["c_p"]) * 4
                rec.append((head, label, s.arena_t[g.off:g.off + n].clone().view(torch.float32), dict(nblk=f.get("nblk"), inv=f.get("inv_count"), C=f.get("C"), c_p=f.get("c_p"))))
            else:
                n = int(f["B"]) * int(f["y_bs"]) * 2
                rec.append((head, label, s.arena_t[y.off:y.off + n].clone().view(torch.bfloat16).float(), {k: f.get(k) for k in ("ldy", "y_bs", "C", "cout", "H", "W", "T")}))
    outs[ab] = rec
    print("ab=0x%02x: %d recorded ops" % (ab, len(rec)), flush=True)
gates = {ab: [r for r in outs[ab] if r[0] == "se_gate"] for ab in outs}
convc = {ab: [r for r in outs[ab] if r[0] == "conv_c"] for ab in outs}
for i, (a, b) in enumerate(zip(gates[0], gates[0x10])):
    d = (a[2] - b[2]).abs().max().item()
    print("se_gate %d: max |gate diff| %.3e  %s | %s" % (i, d, a[3], b[3]), flush=True)
for i, (a, b) in enumerate(zip(convc[0], convc[0x10])):
    d = (a[2] - b[2]).abs().max().item() / (a[2].abs().max().item() + 1e-9)
    print("conv_c %d %s: rel diff %.3e" % (i, a[1][:60], d), flush=True)
