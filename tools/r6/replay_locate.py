"""Which op of the X3D-M plan gives different bytes from run to run when the other sub-batch's graph runs beside it?

Sub-plan 0 of the two-branch deploy form is replayed op by op on one stream, the whole arena hashed after every op (on the
same stream, no host sync), while sub-plan 1's graph replays back to back on a second stream.  Same inputs every repetition,
so the first op whose hash varies over the repetitions has produced different output from identical input.  A second part
poisons the arena of a single-plan form with different byte patterns before the first replay: logits that depend on the
pattern mean a kernel reads bytes no kernel wrote.  Run on the GPU box:
    python tools/r6/replay_locate.py [block_stages] [block_stages_ab]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
from bench import make_model, synth_input  # noqa: E402
from pytorchvideo_amd import _lib as L  # noqa: E402
from pytorchvideo_amd.accelerator import convert_to_deployable_form, transmute_model  # noqa: E402
from pytorchvideo_amd.accelerator.mi355x.conversion import _ingest_inputs  # noqa: E402
from pytorchvideo_amd.utils import synthetic_trained_like_weights  # noqa: E402

full = int(sys.argv[1]) if len(sys.argv) > 1 else 4
ab = int(sys.argv[2]) if len(sys.argv) > 2 else 0
wl = "x3d_m"
torch.manual_seed(0)
m, shape = make_model(wl)
synthetic_trained_like_weights(m, synth_input(shape, 2, 7))
m.eval()
transmute_model(m, "mi355x")
x = synth_input(shape, 32, 99).cuda().bfloat16()
L.tune(block_stages=full, block_stages_ab=ab)


def arena_hash(s):
    a = s.arena_t
    n = a.numel() // 8 * 8
    v = a[:n].view(torch.int64)
    # two independent order-insensitive sums (plain, and weighted by position parity classes) -- enough to see a changed byte
    return torch.stack([v.sum(), (v[::2]).sum() * 3 + (v[1::2]).sum() * 5])


def locate(noise):
    dm = convert_to_deployable_form(m, x, dtype=torch.bfloat16, streams=2)
    ref = dm(x).float().cpu()
    torch.cuda.synchronize()
    p0, p1 = dm.parts[0], dm.parts[1]
    s0, s1 = p0._pv_session, p1._pv_session
    n = len(s0.ops)
    reps = 8
    H = torch.zeros(reps, n, 2, dtype=torch.int64, device="cuda")
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    x0 = x[:16]
    if noise:
        with torch.cuda.stream(sb):
            for _ in range(600):
                s1.launch(use_graph=True)
    with torch.cuda.stream(sa):
        for r in range(reps):
            _ingest_inputs(s0, x0, p0._pv_inputs, False)
            for k in range(n):
                s0.launch(k, k + 1)
                H[r, k] = arena_hash(s0)
    torch.cuda.synchronize()
    Hc = H.cpu()
    bad = [k for k in range(n) if not bool((Hc[:, k] == Hc[0, k]).all())]
    print("block_stages full=%d ab=%d, %s: %d ops; ops whose arena hash varies over %d repetitions: %s" % (
        full, ab, "other sub-plan replaying beside it" if noise else "alone on the chip", n, reps, bad[:12]), flush=True)
    if bad:
        k = bad[0]
        kern = s0.profile(iters=1) and s0.op_kernels[k]
        print("  first: op %d  %s  kernel %s" % (k, s0.ops[k][3], kern), flush=True)
        print("  repetitions grouped by hash at that op:", [int((Hc[:, k] == Hc[r, k]).all(dim=-1).sum()) for r in range(reps)], flush=True)
        f = s0.ops[k][2]
        print("  fields:", {kk: (vv if not hasattr(vv, "off") else ("arena+%d" % vv.off)) for kk, vv in f.items() if kk in (
            "x", "y", "residual", "psum", "ldx", "ldy", "ldr", "B", "T", "H", "W", "cin", "C", "cout", "x_bs", "y_bs", "mode")}, flush=True)
        if k > 0:
            print("  the op before it: op %d  %s  kernel %s" % (k - 1, s0.ops[k - 1][3], s0.op_kernels[k - 1]), flush=True)
    del dm
    torch.cuda.empty_cache()
    return bad


def locate_prefix(reps=6):
    """Ops [0, k) launched back to back (as the graph does) for every k, the other sub-plan's graph replaying beside them: the
    smallest k whose hash varies names op k - 1."""
    dm = convert_to_deployable_form(m, x, dtype=torch.bfloat16, streams=2)
    dm(x)
    torch.cuda.synchronize()
    p0, p1 = dm.parts[0], dm.parts[1]
    s0, s1 = p0._pv_session, p1._pv_session
    n = len(s0.ops)
    H = torch.zeros(reps, n + 1, 2, dtype=torch.int64, device="cuda")
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    x0 = x[:16]
    s0.profile(iters=1)
    for k in range(1, n + 1):
        with torch.cuda.stream(sb):
            for _ in range(2 * reps + 4):
                s1.launch(use_graph=True)
        with torch.cuda.stream(sa):
            for r in range(reps):
                _ingest_inputs(s0, x0, p0._pv_inputs, False)
                s0.launch(0, k)
                H[r, k] = arena_hash(s0)
        torch.cuda.synchronize()
    Hc = H.cpu()
    bad = [k for k in range(1, n + 1) if not bool((Hc[:, k] == Hc[0, k]).all())]
    print("prefix launches [0, k), other sub-plan's graph beside them, %d repetitions per k: k whose arena hash varies: %s" % (reps, bad[:20]), flush=True)
    for k in bad[:4]:
        print("  k=%d: last op %d  %s  kernel %s; groups %s" % (k, k - 1, s0.ops[k - 1][3], s0.op_kernels[k - 1],
              [int((Hc[:, k] == Hc[r, k]).all(dim=-1).sum()) for r in range(reps)]), flush=True)
        f = s0.ops[k - 1][2]
        print("     fields:", {kk: (vv if not hasattr(vv, "off") else ("arena+%d" % vv.off)) for kk, vv in f.items() if kk in (
            "x", "y", "residual", "psum", "ldx", "ldy", "ldr", "B", "T", "H", "W", "cin", "C", "cout", "x_bs", "y_bs", "mode")}, flush=True)
    del dm
    torch.cuda.empty_cache()
    return bad


def locate_lockstep(reps=6):
    """Both sub-plans launch ops [0, k) at the same time on their own streams (the joint graph's situation: op j of one branch runs
    beside op j of the other), for every k; arenas of both hashed."""
    dm = convert_to_deployable_form(m, x, dtype=torch.bfloat16, streams=2)
    dm(x)
    torch.cuda.synchronize()
    parts = list(dm.parts)
    ss = [p._pv_session for p in parts]
    n = len(ss[0].ops)
    H = torch.zeros(2, reps, n + 1, 2, dtype=torch.int64, device="cuda")
    st = [torch.cuda.Stream(), torch.cuda.Stream()]
    xs = [x[:16], x[16:]]
    ss[0].profile(iters=1)
    for k in range(1, n + 1):
        for r in range(reps):
            for b in (0, 1):
                with torch.cuda.stream(st[b]):
                    _ingest_inputs(ss[b], xs[b], parts[b]._pv_inputs, False)
            torch.cuda.synchronize()
            for b in (0, 1):
                with torch.cuda.stream(st[b]):
                    ss[b].launch(0, k)
            for b in (0, 1):
                with torch.cuda.stream(st[b]):
                    H[b, r, k] = arena_hash(ss[b])
        torch.cuda.synchronize()
    Hc = H.cpu()
    for b in (0, 1):
        bad = [k for k in range(1, n + 1) if not bool((Hc[b, :, k] == Hc[b, 0, k]).all())]
        print("lockstep prefixes [0, k) of both sub-plans, %d repetitions per k, sub-plan %d: k whose arena hash varies: %s" % (reps, b, bad[:30]), flush=True)
        for k in bad[:3]:
            print("  k=%d: last op %d  %s  kernel %s; groups %s" % (k, k - 1, ss[0].ops[k - 1][3], ss[0].op_kernels[k - 1],
                  [int((Hc[b, :, k] == Hc[b, r, k]).all(dim=-1).sum()) for r in range(reps)]), flush=True)
    del dm
    torch.cuda.empty_cache()


def final_arena_diff(reps=6, use_graph=True):
    """The failing form itself (joint graph of both sub-plans): which bytes of the final arenas differ between replays, which op
    wrote them last (exact, 64-byte granules), and how the differing bytes of the earliest such op are laid out."""
    dm = convert_to_deployable_form(m, x, dtype=torch.bfloat16, streams=2, use_graph=use_graph)
    ss = list(dm._pv_sessions)
    G = 64
    owners = []
    for s in ss:
        own = torch.full(((s.arena_t.numel() + G - 1) // G,), -1, dtype=torch.int32)
        for idx, (kind, dcls, fields, label, ab_, fl) in enumerate(s.ops):
            y = fields.get("y")
            if y is not None and getattr(y, "space", None) == "arena" and fields.get("y_bs"):
                nbytes = int(fields.get("B", 1)) * int(fields["y_bs"]) * (4 if fields.get("y_f32") else 2)
                own[y.off // G:(y.off + nbytes + G - 1) // G] = idx
            for key in ("psum", "gate"):
                q = fields.get(key)
                if q is not None and getattr(q, "space", None) == "arena" and key == "psum" and "se" in label and "gate" not in label:
                    own[q.off // G:q.off // G + 1] = idx
        owners.append(own.cuda())
    ref = None
    for rep in range(reps):
        out = dm(x)
        torch.cuda.synchronize()
        snap = [s.arena_t.clone() for s in ss]
        if ref is None:
            ref, ref_out = snap, out.clone()
            continue
        print("%s replay %d: %d logits differ from replay 0 (max |diff| %.3e)" % ("graph" if use_graph else "eager", rep, int((out != ref_out).sum()), float((out.float() - ref_out.float()).abs().max())), flush=True)
        for bi, (a, b2, s, own) in enumerate(zip(ref, snap, ss, owners)):
            neq = a != b2
            if not bool(neq.any()):
                print("  sub-plan %d: final arena identical" % bi, flush=True)
                continue
            n = neq.numel() // G * G
            gran = neq[:n].view(-1, G).any(dim=1)
            ids, cnt = torch.unique(own[:gran.numel()][gran], return_counts=True)
            print("  sub-plan %d: %d bytes differ; granules by last writer: %s" % (bi, int(neq.sum()), [(int(i), int(c)) for i, c in zip(ids, cnt)][:14]), flush=True)
            first = int(ids[ids >= 0].min()) if bool((ids >= 0).any()) else None
            if first is not None and rep <= 2:
                kind, dcls, fields, label, ab_, fl = s.ops[first]
                y = fields["y"]
                ld, ybs = int(fields["ldy"]), int(fields["y_bs"])
                idx = (neq[:n].view(-1, G).any(dim=1) & (own[:gran.numel()] == first)).nonzero().flatten()
                byt = neq.nonzero().flatten()
                byt = byt[(byt >= y.off) & (byt < y.off + int(fields.get("B", 1)) * ybs * 2)] - y.off
                el = (byt // 2).unique()
                print("    earliest: op %d %s; %d elements differ; y_bs %d ldy %d" % (first, label, el.numel(), ybs, ld), flush=True)
                W_, H_ = int(fields.get("W", 0) or fields.get("Wo", 0) or 1), int(fields.get("H", 0) or fields.get("Ho", 0) or 1)
                shown = 0
                for e in el.tolist()[:4000:100]:
                    bb, r = divmod(e, ybs)
                    vox, c = divmod(r, ld)
                    t, r2 = divmod(vox, H_ * W_)
                    h, w = divmod(r2, W_)
                    va = a[y.off + 2 * e: y.off + 2 * e + 2].view(torch.bfloat16).item()
                    vb = b2[y.off + 2 * e: y.off + 2 * e + 2].view(torch.bfloat16).item()
                    print("      clip %d t %d h %d w %d c %d: %g vs %g" % (bb, t, h, w, c, va, vb), flush=True)
                    shown += 1
                    if shown >= 24:
                        break
                vox_ids = (el // ld)
                print("      distinct voxels %d; distinct clips %s; elements per voxel (max) %d" % (vox_ids.unique().numel(), sorted(set((el // ybs).tolist()))[:16], int(torch.unique(vox_ids, return_counts=True)[1].max())), flush=True)
    del dm
    torch.cuda.empty_cache()


def truncated(reps=24, ks=(4, 5, 6)):
    """A full two-stream forward (leaves the arenas as a replay leaves them), then ops [0, k) of both sub-plans exactly as the
    eager forward starts them (ingest + launch on the sub-plan's stream, no sync between); the output regions of ops 0 .. k-1
    compared byte for byte over the repetitions."""
    dm = convert_to_deployable_form(m, x, dtype=torch.bfloat16, streams=2, use_graph=False)
    parts = list(dm.parts)
    ss = [p._pv_session for p in parts]
    st = [torch.cuda.Stream(), torch.cuda.current_stream()]
    xs = [x[:16], x[16:]]

    def regions(s, k):
        out = []
        for idx in range(k):
            kind, dcls, fields, label, ab_, fl = s.ops[idx]
            y = fields.get("y")
            if y is not None and getattr(y, "space", None) == "arena" and fields.get("y_bs"):
                out.append((idx, "y", y.off, int(fields.get("B", 1)) * int(fields["y_bs"]) * 2))
            for key in ("psum", "gate"):
                q = fields.get(key)
                if q is not None and getattr(q, "space", None) == "arena":
                    out.append((idx, key, q.off, 4096))
        return out

    for k in ks:
        snaps = []
        for rep in range(reps):
            dm(x)
            torch.cuda.synchronize()
            for b in (0, 1):
                with torch.cuda.stream(st[b]):
                    _ingest_inputs(ss[b], xs[b], parts[b]._pv_inputs, False)
                    ss[b].launch(0, k)
            torch.cuda.synchronize()
            snaps.append([[s.arena_t[off:off + n].clone() for (_, _, off, n) in regions(s, k)] for s in ss])
        for b in (0, 1):
            rg = regions(ss[b], k)
            res = []
            for i, (idx, key, off, n) in enumerate(rg):
                neq = [int((snaps[r][b][i] != snaps[0][b][i]).sum()) for r in range(1, reps)]
                res.append("op %d %s: %s" % (idx, key, neq))
            print("k=%d sub-plan %d: bytes differing from repetition 0 per region: %s" % (k, b, "; ".join(res)), flush=True)
            for i, (idx, key, off, n) in enumerate(rg):
                if key != "y":
                    continue
                fields = ss[b].ops[idx][2]
                ld, ybs = int(fields["ldy"]), int(fields["y_bs"])
                W_ = int(fields.get("W", 0) or fields.get("Wo", 0) or 1)
                H_ = int(fields.get("H", 0) or fields.get("Ho", 0) or 1)
                for r in range(1, reps):
                    neq = (snaps[r][b][i] != snaps[0][b][i])
                    if not bool(neq.any()):
                        continue
                    el = (neq.nonzero().flatten() // 2).unique()
                    va = snaps[0][b][i].view(torch.bfloat16)[el].float()
                    vb = snaps[r][b][i].view(torch.bfloat16)[el].float()
                    bb = el // ybs
                    vox = (el % ybs) // ld
                    c = (el % ybs) % ld
                    t = vox // (H_ * W_)
                    h = (vox % (H_ * W_)) // W_
                    w = vox % W_
                    print("    op %d y, repetition %d: %d elements; clips %s; t %s; h %d..%d; w %d..%d; channels %s; max |diff| %.3e, max relative %.3e" % (
                        idx, r, el.numel(), sorted(set(bb.tolist()))[:8], sorted(set(t.tolist()))[:16], int(h.min()), int(h.max()), int(w.min()), int(w.max()),
                        sorted(set(c.tolist()))[:24], float((va - vb).abs().max()), float(((va - vb).abs() / (va.abs() + 1e-6)).max())), flush=True)
                    for j in range(0, min(el.numel(), 400), 40):
                        print("        clip %d t %d h %d w %d c %d: %g vs %g" % (int(bb[j]), int(t[j]), int(h[j]), int(w[j]), int(c[j]), float(va[j]), float(vb[j])), flush=True)
                    break
    ss[0].profile(iters=1)
    for idx in range(8):
        print("   op %d: %s  kernel %s" % (idx, ss[0].ops[idx][3], ss[0].op_kernels[idx]), flush=True)
    del dm
    torch.cuda.empty_cache()


def sweep(reps=4):
    """truncated() for every k, comparing only the regions op k - 1 itself wrote."""
    dm = convert_to_deployable_form(m, x, dtype=torch.bfloat16, streams=2, use_graph=False)
    parts = list(dm.parts)
    ss = [p._pv_session for p in parts]
    st = [torch.cuda.Stream(), torch.cuda.current_stream()]
    xs = [x[:16], x[16:]]
    n = len(ss[0].ops)
    ss[0].profile(iters=1)

    def regions(s, idx):
        kind, dcls, fields, label, ab_, fl = s.ops[idx]
        out = []
        y = fields.get("y")
        if y is not None and getattr(y, "space", None) == "arena":
            nb = int(fields.get("B", 1)) * int(fields["y_bs"]) * 2 if fields.get("y_bs") else 4096
            out.append(("y", y.off, nb))
        for key in ("psum", "gate"):
            q = fields.get(key)
            if q is not None and getattr(q, "space", None) == "arena":
                out.append((key, q.off, 4096))
        return out

    for k in range(1, n + 1):
        snaps = []
        for rep in range(reps):
            dm(x)
            torch.cuda.synchronize()
            for b in (0, 1):
                with torch.cuda.stream(st[b]):
                    _ingest_inputs(ss[b], xs[b], parts[b]._pv_inputs, False)
                    ss[b].launch(0, k)
            torch.cuda.synchronize()
            snaps.append([[s.arena_t[off:off + nb].clone() for (_, off, nb) in regions(s, k - 1)] for s in ss])
        msg = []
        for b in (0, 1):
            for i, (key, off, nb) in enumerate(regions(ss[b], k - 1)):
                neq = [int((snaps[r][b][i] != snaps[0][b][i]).sum()) for r in range(1, reps)]
                if any(neq):
                    msg.append("sub-plan %d %s %s" % (b, key, neq))
        print("k=%d op %d %s [%s]: %s" % (k, k - 1, ss[0].ops[k - 1][3], ss[0].op_kernels[k - 1], "; ".join(msg) if msg else "identical"), flush=True)
    del dm
    torch.cuda.empty_cache()


def poison_sweep(ks=(9, 11, 12, 91), chunks=96):
    """ONE sub-plan alone on the chip: the arena zeroed, one chunk of it filled with a pattern, ingest, ops [0, k): does the output of
    op k - 1 depend on bytes no op of this pass wrote?  Sensitive chunks are mapped to the plan's buffers."""
    dm = convert_to_deployable_form(m, x, dtype=torch.bfloat16, streams=2, use_graph=False)
    p0 = dm.parts[0]
    s = p0._pv_session
    x0 = x[:16]
    n = len(s.ops)
    s.profile(iters=1)
    A = s.arena_t
    size = A.numel()
    step = (size + chunks - 1) // chunks // 256 * 256 + 256

    def out_region(idx):
        f = s.ops[idx][2]
        y = f["y"]
        nb = int(f.get("B", 1)) * int(f["y_bs"]) * (4 if f.get("y_f32") else 2) if f.get("y_bs") else 4096
        return y.off, nb

    def run(k, lo=None, hi=None, byte=0x3c):
        A.zero_()
        if lo is not None:
            A[lo:hi] = byte
        _ingest_inputs(s, x0, p0._pv_inputs, False)
        s.launch(0, k)
        torch.cuda.synchronize()
        off, nb = out_region(k - 1)
        return A[off:off + nb].clone()

    bufs = []
    for idx, (kind, dcls, fields, label, ab_, fl) in enumerate(s.ops):
        for key, v in fields.items():
            if getattr(v, "space", None) == "arena":
                bufs.append((v.off, idx, key, label.split("|")[0]))
    for k in ks:
        k = min(k, n)
        base = run(k)
        again = run(k)
        print("k=%d (op %d %s, %s): two clean runs identical %s" % (k, k - 1, s.ops[k - 1][3], s.op_kernels[k - 1], bool((base == again).all())), flush=True)
        for byte in (0x3c, 0xff):
            hits = []
            for c in range(chunks):
                lo, hi = c * step, min(size, (c + 1) * step)
                if lo >= size:
                    break
                o = run(k, lo, hi, byte)
                d = int((o != base).sum())
                if d:
                    hits.append((lo, hi, d))
            print("  fill 0x%02x: %d sensitive chunks of %d bytes: %s" % (byte, len(hits), step, [(lo, d) for lo, hi, d in hits][:12]), flush=True)
            for lo, hi, d in hits[:6]:
                inside = sorted(set((off, idx, key, lab) for off, idx, key, lab in bufs if lo - (64 << 20) <= off < hi))
                # the buffers that START at or before this chunk and are the nearest ones (the chunk lies inside one of them)
                near = [b_ for b_ in inside if b_[0] < hi][-10:]
                print("    chunk [%d, %d): buffers starting at or below it (offset, op, field, label): %s" % (lo, hi, near), flush=True)
    del dm
    torch.cuda.empty_cache()


def beside(k=5, reps=6, loops=60, only=None):
    """Sub-plan 0 runs ops [0, k) while sub-plan 1 loops ONE of its ops (op j, for every j): beside which kernel do the outputs of
    ops k-2 / k-1 vary?"""
    dm = convert_to_deployable_form(m, x, dtype=torch.bfloat16, streams=2, use_graph=False)
    dm(x)
    torch.cuda.synchronize()
    parts = list(dm.parts)
    ss = [p._pv_session for p in parts]
    s0, s1 = ss
    n = len(s0.ops)
    s0.profile(iters=1)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()

    def region(idx):
        f = s0.ops[idx][2]
        y = f["y"]
        return y.off, (int(f.get("B", 1)) * int(f["y_bs"]) * 2 if f.get("y_bs") else 4096)

    watch = [i for i in (k - 2, k - 1) if s0.ops[i][2].get("y") is not None]
    for j in (only or range(n)):
        snaps = []
        for r in range(reps):
            with torch.cuda.stream(sb):
                for _ in range(loops):
                    s1.launch(j, j + 1)
            with torch.cuda.stream(sa):
                _ingest_inputs(s0, x[:16], parts[0]._pv_inputs, False)
                s0.launch(0, k)
            torch.cuda.synchronize()
            snaps.append([s0.arena_t[o:o + nb].clone() for (o, nb) in map(region, watch)])
        neq = {w: [int((snaps[r][i] != snaps[0][i]).sum()) for r in range(1, reps)] for i, w in enumerate(watch)}
        if any(any(v) for v in neq.values()):
            print("beside op %d %s [%s]: bytes differing from repetition 0: %s" % (j, s1.ops[j][3], s0.op_kernels[j], neq), flush=True)
            i = len(watch) - 1
            f = s0.ops[watch[i]][2]
            ld, ybs = int(f["ldy"]), int(f["y_bs"])
            W_ = int(f.get("W", 0) or f.get("Wo", 0) or 1)
            H_ = int(f.get("H", 0) or f.get("Ho", 0) or 1)
            for r in (1, 2):
                el = ((snaps[r][i] != snaps[0][i]).nonzero().flatten() // 2).unique()
                if not el.numel():
                    continue
                va = snaps[0][i].view(torch.bfloat16)[el].float()
                vb = snaps[r][i].view(torch.bfloat16)[el].float()
                bb = el // ybs
                vox = (el % ybs) // ld
                c = (el % ybs) % ld
                t = vox // (H_ * W_)
                h = (vox % (H_ * W_)) // W_
                w = vox % W_
                import collections
                print("    op %d y, repetition %d vs 0: %d elements; clips %s; t histogram %s; h%%2 %s; w%%14 %s; channels %s; max |diff| %.3e" % (
                    watch[i], r, el.numel(), sorted(set(bb.tolist())), sorted(collections.Counter(t.tolist()).items()),
                    sorted(collections.Counter((h % 2).tolist()).items()), sorted(collections.Counter((w % 14).tolist()).items()),
                    sorted(collections.Counter(c.tolist()).items()), float((va - vb).abs().max())), flush=True)
                vs = (vox + bb * (ybs // ld)).unique()
                print("      distinct voxels %d; first: %s" % (vs.numel(), [(int(bb[q_]), int(t[q_]), int(h[q_]), int(w[q_]), int(c[q_]), round(float(va[q_]), 5), round(float(vb[q_]), 5)) for q_ in range(0, min(el.numel(), 60), 3)]), flush=True)
    print("beside(): k=%d done, %d co-running ops tried, watched ops %s" % (k, n, watch), flush=True)
    del dm
    torch.cuda.empty_cache()


def beside_all(reps=4, loops=30):
    """Every DISTINCT geometry of bottleneck_block_kernel in the plan (first op of each label), beside every op of the other sub-plan."""
    dm = convert_to_deployable_form(m, x, dtype=torch.bfloat16, streams=2, use_graph=False)
    dm(x)
    torch.cuda.synchronize()
    parts = list(dm.parts)
    s0, s1 = [p._pv_session for p in parts]
    n = len(s0.ops)
    s0.profile(iters=1)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    seen, targets = set(), []
    for i in range(n):
        if "bottleneck_block_kernel" in s0.op_kernels[i] and s0.ops[i][3] not in seen:
            seen.add(s0.ops[i][3])
            targets.append(i)
    print("block_stages full=%d ab=%d: %d ops, %d of them on bottleneck_block_kernel, distinct geometries at ops %s" % (
        full, ab, n, sum("bottleneck_block_kernel" in k_ for k_ in s0.op_kernels), targets), flush=True)
    for tgt in targets:
        f = s0.ops[tgt][2]
        y = f["y"]
        nb = int(f.get("B", 1)) * int(f["y_bs"]) * 2
        regs = [(y.off, nb)]
        if f.get("psum") is not None and getattr(f["psum"], "space", None) == "arena":
            regs.append((f["psum"].off, 1 << 16))
        bad = []
        for j in range(n):
            snaps = []
            for r in range(reps):
                with torch.cuda.stream(sb):
                    for _ in range(loops):
                        s1.launch(j, j + 1)
                with torch.cuda.stream(sa):
                    _ingest_inputs(s0, x[:16], parts[0]._pv_inputs, False)
                    s0.launch(0, tgt + 1)
                torch.cuda.synchronize()
                snaps.append([s0.arena_t[o:o + nb_].clone() for (o, nb_) in regs])
            neq = [sum(int((snaps[r][i] != snaps[0][i]).sum()) for i in range(len(regs))) for r in range(1, reps)]
            if any(neq):
                bad.append((j, s0.op_kernels[j], neq))
        print("  op %d %s: co-running ops beside which its output varied: %s" % (tgt, s0.ops[tgt][3], bad if bad else "none of %d" % n), flush=True)
    del dm
    torch.cuda.empty_cache()


def victims(corunners=(0, 1), reps=4, loops=40):
    """Every op of sub-plan 0 as the watched op (its own output regions), beside the other sub-plan's stem / stride-2 pwdw kernel."""
    dm = convert_to_deployable_form(m, x, dtype=torch.bfloat16, streams=2, use_graph=False)
    dm(x)
    torch.cuda.synchronize()
    parts = list(dm.parts)
    s0, s1 = [p._pv_session for p in parts]
    n = len(s0.ops)
    s0.profile(iters=1)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    print("block_stages full=%d ab=%d: %d ops; co-running ops %s" % (full, ab, n, [(j, s0.op_kernels[j]) for j in corunners]), flush=True)
    nbad = 0
    for tgt in range(n):
        f = s0.ops[tgt][2]
        regs = []
        y = f.get("y")
        if y is not None and getattr(y, "space", None) == "arena":
            regs.append((y.off, int(f.get("B", 1)) * int(f["y_bs"]) * 2 if f.get("y_bs") else 4096))
        for key in ("psum", "gate"):
            q = f.get(key)
            if q is not None and getattr(q, "space", None) == "arena":
                regs.append((q.off, 4096))
        msgs = []
        for j in corunners:
            snaps = []
            for r in range(reps):
                with torch.cuda.stream(sb):
                    for _ in range(loops):
                        s1.launch(j, j + 1)
                with torch.cuda.stream(sa):
                    _ingest_inputs(s0, x[:16], parts[0]._pv_inputs, False)
                    s0.launch(0, tgt + 1)
                torch.cuda.synchronize()
                snaps.append([s0.arena_t[o:o + nb_].clone() for (o, nb_) in regs])
            neq = [sum(int((snaps[r][i] != snaps[0][i]).sum()) for i in range(len(regs))) for r in range(1, reps)]
            if any(neq):
                msgs.append("beside op %d: %s" % (j, neq))
        if msgs:
            nbad += 1
            print("  op %d %s [%s]: %s" % (tgt, s0.ops[tgt][3], s0.op_kernels[tgt], "; ".join(msgs)), flush=True)
    print("victims(): %d of %d ops varied" % (nbad, n), flush=True)
    del dm
    torch.cuda.empty_cache()


def poison():
    dm = convert_to_deployable_form(m, x[:4], dtype=torch.bfloat16, streams=1)
    s = dm._pv_session
    outs = {}
    for name, byte in (("0x00", 0), ("0x3c", 0x3c), ("0xff", 0xff), ("0x7f", 0x7f)):
        s.arena_t.fill_(byte)
        torch.cuda.synchronize()
        outs[name] = dm(x[:4]).float().cpu()
        again = dm(x[:4]).float().cpu()
        print("single plan, batch 4, arena filled with %s before the replay: logits finite %s; second replay (arena as the first left it) identical %s; |diff| to the 0x00 fill %.3e" % (
            name, bool(torch.isfinite(outs[name]).all()), bool((again == outs[name]).all()), float((outs[name] - outs["0x00"]).abs().max())), flush=True)


if __name__ == "__main__":
    if "victims" in sys.argv:
        victims()
    elif "besideall" in sys.argv:
        beside_all()
    elif "beside" in sys.argv:
        for abl in [int(v) for v in os.environ.get("PV_BESIDE_ABL", "0,11,12,13,9").split(",")]:
            L.tune(block_abl=abl)
            print("---- block_abl = %d (0 product kernel, 8 two barriers per iteration, 9 stencil LDS loads in reverse order + pinned, 10 channel pairs, 11 forward order + pinned, 12 lgkmcnt(0) behind a row's loads, 13 MID stores not reordered)" % abl, flush=True)
            beside(5, only=(1, 8))
    elif "poisonsweep" in sys.argv:
        poison_sweep()
    elif "sweep" in sys.argv:
        sweep()
    elif "truncated" in sys.argv:
        truncated()
    elif "lockstep" in sys.argv:
        final_arena_diff()
        final_arena_diff(reps=4, use_graph=False)
    elif "prefix" in sys.argv:
        locate_prefix()
    else:
        locate(noise=False)
        locate(noise=True)
        poison()
