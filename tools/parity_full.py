"""Parity of the deploy form at the FULL BASELINE.json geometries against the CPU oracle (run on the GPU box).

Weights: `trained_like` (oracle/weights.py::trained_like_fill: calibrated BatchNorm statistics, block-final gamma
U(0.05, 0.2) -- the instance bench.py times and the north-star tests assert 1e-2 on), `calibrated` (the stress instance; oracle/weights.py::calibrated_fill -- the reference tests' BatchNorm randomisation,
tests/test_fuse_bn.py:58-63, with the running statistics then set to the batch statistics of a calibration batch:
what a trained checkpoint holds, logits O(1-10)), `reference_style` (arbitrary running statistics: activations grow
to 1e4-1e10) or `deterministic`.

Per case, max|d| / max|oracle logits| over ALL rows of the batch:
  fp32_vs_oracle              fp32 deploy form vs the fp32 oracle                                  (north star 1e-3)
  bf16_vs_emulated_oracle     bf16 deploy form vs the oracle evaluated with bf16 STORAGE (same rounded weights, every
                              stored activation / MFMA operand rounded where the deploy form rounds it,
                              oracle/functional.py::storage_emulation): isolates the kernels' own arithmetic
  bf16_vs_fp32_oracle         bf16 deploy form vs the UNQUANTISED fp32 oracle                      (north star 1e-2)
  storage_floor               emulated oracle vs fp32 oracle: what bf16 storage costs with exact arithmetic, no kernel
  weights_floor               oracle on bf16-rounded weights / input vs fp32 oracle
`--batch N --streams K` runs the deploy form the way bench.py does (K sub-batch plans as branches of one graph) and
checks every row.

    python tools/parity_full.py [--workloads x3d_m,x3d_l,slowfast_r50,mvit_b_32x3] [--fills calibrated] [--bench-batch]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def filled_model(workload, fill, seed=0):
    """(original-form model with the requested fill, input shape)."""
    from bench import make_model, synth_input
    from oracle.weights import calibrated_fill, deterministic_fill, reference_style_fill, trained_like_fill
    torch.manual_seed(0)
    m, shape = make_model(workload)
    # the fill's calibration forward runs with the process's DEFAULT thread count (as in bench.py), whatever cap the test session
    # put on the CPU references: the instance is then the one bench.py times, bit for bit (tests/conftest.py)
    capped = torch.get_num_threads()
    torch.set_num_threads(int(os.environ.get("PV_TORCH_DEFAULT_THREADS", capped)))
    try:
        return _fill(m, shape, fill, seed)
    finally:
        torch.set_num_threads(capped)


def _fill(m, shape, fill, seed):
    from bench import synth_input
    from oracle.weights import calibrated_fill, deterministic_fill, reference_style_fill, trained_like_fill
    if fill == "trained_like":
        trained_like_fill(m, synth_input(shape, 2, 7), seed)
    elif fill == "trained_like_wide":      # the reported second instance (round-4 verdict): block-final gamma U(0.1, 0.4)
        trained_like_fill(m, synth_input(shape, 2, 7), seed, final_gamma=(0.1, 0.4))
    elif fill == "calibrated":
        calibrated_fill(m, synth_input(shape, 2, 7), seed)
    elif fill == "reference_style":
        reference_style_fill(m, seed)
    else:
        deterministic_fill(m, seed)
    return m.eval(), shape


def _oracle_rows(workload, sd, x, batch_hint, weights_only, pool_min, threads):
    """Worker of `oracle_numbers` (spawned process, or the caller itself): the three oracle logits of the rows in `x`."""
    from bench import oracle_forward
    from oracle import functional as OF
    from oracle.weights import quantize_like_kernels
    if threads:
        torch.set_num_threads(threads)
    fn = oracle_forward(workload)
    with torch.no_grad():
        want = fn(sd, x)
        sd_q = quantize_like_kernels(sd)
        xq = [t.bfloat16().float() for t in x] if isinstance(x, list) else x.bfloat16().float()
        want_w = fn(sd_q, xq) if weights_only else None
        with OF.storage_emulation(torch.bfloat16, batch=batch_hint, pool_stream_min_elems=pool_min):
            want_e = fn(sd_q, xq)
    return want, want_w, want_e


def oracle_numbers(workload, sd, x, batch_hint=1, weights_only=True):
    """(fp32 oracle, weights-only oracle [or None], bf16-storage oracle) logits.

    Eval-mode rows are independent, so a batch is evaluated as row shards in PARALLEL worker processes (spawned: the caller
    usually holds a HIP context), 16 threads each: on the 256-thread GPU hosts one 256-thread torch call runs the bench batch at
    ~1.2 clips/s where 8-thread calls reach 4.5 each (bench.py's cpu_baseline sweep) -- the bench-batch cases of
    tests/test_gpu_full_geometry.py spent 259 s of the suite's 913 s here (round-5 verdict, weak #8).  NOTE: the fp32
    oracle is invariant to the sharding to 7e-7, the bf16-STORAGE emulation is not (a last-bit difference of an fp32 sum
    flips bf16 roundings downstream: X3D-M rows move by up to 6e-3 between a batch-2 call and two batch-1 calls, CPU only) --
    the per-row kernel-arithmetic bounds of the tests are measured against THIS sharding.  PV_ORACLE_SHARDS=1 keeps
    everything in this process."""
    from pytorchvideo_amd.accelerator.mi355x import tuning       # the plan's own routing threshold, not a copy of it
    pool_min = tuning.get("pool_stream_min_elems")
    batch = (x[0] if isinstance(x, list) else x).shape[0]
    # the SHARDING is a function of the batch only (8 shards from batch 8 on), so the numbers do not depend on the host; only
    # how many shards run at once does
    shards = int(os.environ.get("PV_ORACLE_SHARDS", "0")) or (8 if batch >= 8 else 1)
    if shards <= 1:
        return _oracle_rows(workload, sd, x, batch_hint, weights_only, pool_min, 0)
    conc = max(1, min(shards, (os.cpu_count() or 1) // 16))
    # plain subprocesses running this file's --oracle-worker entry (no multiprocessing: a spawned child re-imports the
    # parent's __main__, which is pytest / bench / a stdin script); tensors travel through files in a temporary directory
    import subprocess
    import tempfile
    cuts = [batch * i // shards for i in range(shards + 1)]
    rows = lambda a, b: [t[a:b].clone() for t in x] if isinstance(x, list) else x[a:b].clone()
    with tempfile.TemporaryDirectory(prefix="pv_oracle_") as tmp:
        torch.save(sd, os.path.join(tmp, "sd.pt"))
        env = dict(os.environ, OMP_NUM_THREADS="16", HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
        running, rcs = [], []
        for i in range(shards):
            torch.save({"workload": workload, "x": rows(cuts[i], cuts[i + 1]), "batch_hint": batch_hint,
                        "weights_only": weights_only, "pool_min": pool_min}, os.path.join(tmp, "in%d.pt" % i))
            if len(running) == conc:
                rcs.append(running.pop(0).wait())
            running.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "--oracle-worker", tmp, str(i)], env=env))
        rcs += [p.wait() for p in running]
        if any(rcs):
            raise RuntimeError("oracle worker failed: exit codes %s" % rcs)
        parts = [torch.load(os.path.join(tmp, "out%d.pt" % i)) for i in range(shards)]
    cat = lambda k: torch.cat([p[k] for p in parts]) if parts[0][k] is not None else None
    return cat(0), cat(1), cat(2)


def rel(a, b):
    return (a.float().cpu() - b).abs().max().item() / max(b.abs().max().item(), 1e-9)


def case(workload, fill="calibrated", batch=1, streams=1, dtypes=("fp32", "bf16")):
    from bench import synth_input
    from pytorchvideo_amd.accelerator import convert_to_deployable_form, transmute_model
    out = {"workload": workload, "fill": fill, "batch": batch, "streams": streams}
    m, shape = filled_model(workload, fill)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    x = synth_input(shape, batch, 99)
    t0 = time.time()
    want, want_w, want_e = oracle_numbers(workload, sd, x, batch_hint=max(1, batch // streams), weights_only=batch == 1)
    out["oracle_s"] = round(time.time() - t0, 2)
    out["logit_absmax"] = round(want.abs().max().item(), 4)
    out["logit_std"] = round(want.std().item(), 4)
    out["weights_floor"] = rel(want_w, want) if want_w is not None else None
    out["storage_floor"] = rel(want_e, want)
    out["storage_rows_worst"] = max(rel(want_e[i:i + 1], want[i:i + 1]) for i in range(batch))   # per row, normalised by the row
    transmute_model(m, "mi355x")
    for tag in dtypes:
        dtype = torch.float32 if tag == "fp32" else torch.bfloat16
        xd = [t.cuda().to(dtype) for t in x] if isinstance(x, list) else x.cuda().to(dtype)
        dm = convert_to_deployable_form(m, xd, dtype=dtype, streams=streams)
        got = dm(list(xd) if isinstance(xd, list) else xd).float().cpu()
        got2 = dm(list(xd) if isinstance(xd, list) else xd).float().cpu()     # graph replay: same answer
        out[tag + "_replay_equal"] = bool(torch.equal(got, got2))
        if tag == "fp32":
            out["fp32_vs_oracle"] = rel(got, want)
        else:
            out["bf16_vs_emulated_oracle"] = rel(got, want_e)
            out["bf16_vs_fp32_oracle"] = rel(got, want)
            out["bf16_rows_worst"] = max(rel(got[i:i + 1], want_e[i:i + 1]) for i in range(batch))
            out["bf16_rows_worst_fp32"] = max(rel(got[i:i + 1], want[i:i + 1]) for i in range(batch))
            out["top1_agree"] = int((got.argmax(1) == want.argmax(1)).sum().item())
            # per row: (error vs the fp32 oracle, what bf16 storage alone does to the row, the oracle's top-1 / top-2 margin),
            # all normalised by the row's own max|logit|, and whether the top-1 class agrees
            top2 = want.topk(2, dim=1).values
            out["row_detail"] = [
                {"row": i, "err_fp32": rel(got[i:i + 1], want[i:i + 1]), "storage": rel(want_e[i:i + 1], want[i:i + 1]),
                 "err_kernel": rel(got[i:i + 1], want_e[i:i + 1]),
                 "margin": (top2[i, 0] - top2[i, 1]).item() / max(want[i].abs().max().item(), 1e-9),
                 "top1": bool(got[i].argmax() == want[i].argmax())} for i in range(batch)]
            out["top1_agree_emulated"] = int((got.argmax(1) == want_e.argmax(1)).sum().item())
        del dm
        torch.cuda.empty_cache()
    return out


if __name__ == "__main__":
    if len(sys.argv) == 4 and sys.argv[1] == "--oracle-worker":      # oracle_numbers' row-shard worker (CPU only)
        tmp, i = sys.argv[2], int(sys.argv[3])
        job = torch.load(os.path.join(tmp, "in%d.pt" % i))
        sd = torch.load(os.path.join(tmp, "sd.pt"))
        out = _oracle_rows(job["workload"], sd, job["x"], job["batch_hint"], job["weights_only"], job["pool_min"], 16)
        torch.save(out, os.path.join(tmp, "out%d.pt" % i))
        sys.exit(0)
    ap = argparse.ArgumentParser()
    ap.add_argument("--workloads", default="x3d_m,x3d_l,slowfast_r50,mvit_b_32x3")
    ap.add_argument("--fills", default="trained_like")
    ap.add_argument("--bench-batch", action="store_true", help="also the bench batch with its stream count, bf16")
    ap.add_argument("--json", default="")
    ap.add_argument("--tune", default="", help="k=v,k=v: emitter options / library knobs (accelerator/mi355x/tuning.py)")
    a = ap.parse_args()
    if a.tune:
        from pytorchvideo_amd.accelerator.mi355x import tuning
        tuning.apply(a.tune)
    from bench import WORKLOADS
    rows = []
    for w in a.workloads.split(","):
        for f in a.fills.split(","):
            r = case(w, f)
            rows.append(r)
            print(json.dumps(r), flush=True)
        if a.bench_batch:
            r = case(w, a.fills.split(",")[0], batch=WORKLOADS[w]["batch"], streams=WORKLOADS[w].get("streams", 1), dtypes=("bf16",))
            rows.append(r)
            print(json.dumps(r), flush=True)
    if a.json:
        json.dump(rows, open(a.json, "w"), indent=1)
