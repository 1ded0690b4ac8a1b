"""Does running the batch as k independent sub-batches on k streams (kernels of different sub-batches overlap,
tails and launch gaps of one fill with work of the other) beat one big batch?   (GPU box)

    python tools/try_concurrent.py --workload x3d_m --splits 1,2,4 [--steps 30]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="x3d_m")
    ap.add_argument("--splits", default="1,2,4")
    ap.add_argument("--steps", type=int, default=30)
    a = ap.parse_args()
    from bench import WORKLOADS, build_model
    B = WORKLOADS[a.workload]["batch"]
    dev = torch.device("cuda", 0)
    for k in [int(s) for s in a.splits.split(",")]:
        models = [build_model(a.workload, B // k, dev, torch.bfloat16) for _ in range(k)]
        streams = [torch.cuda.Stream() for _ in range(k)]
        def step():
            for (m, x), s in zip(models, streams):
                with torch.cuda.stream(s):
                    m(list(x) if isinstance(x, list) else x)
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.steps
        print("%s: %d x batch %d on %d streams: %.3f ms/step  %.1f clips/s" % (a.workload, k, B // k, k, dt * 1e3, B / dt), flush=True)
        del models, streams
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
