"""Golden vectors for the video-level ensembling that follows the forward path, produced by the
REFERENCE'S OWN CODE (run where /root/reference exists; the fixture travels, the reference does not).

`pytorchvideo_trainer` cannot be imported here (pytorch_lightning, hydra, torchrecipes are absent), so the three
methods of `VideoClassificationModule` that implement the protocol are lifted out of the reference source
with `ast` at fixture-generation time -- their bodies are compiled and executed UNCHANGED, bound to a stand-in
`self` that only carries the attributes they touch:

  _test_step_with_data_ensembling   pytorchvideo_trainer/module/video_classification.py:244-263  (softmax of the logits)
  _ensemble_at_video_level          :290-311  (per-video sum / max + clip count, dict insertion order)
  on_test_epoch_end                 :275-288  (division by the clip count, stacking)

    python tests/golden/make_ensemble_golden.py      # writes tests/golden/ensemble.pt
"""
import ast
import os
import sys

import torch

REF = os.environ.get("PV_REFERENCE_ROOT", "/root/reference")
SRC = os.path.join(REF, "pytorchvideo_trainer", "pytorchvideo_trainer", "module", "video_classification.py")
WANTED = ("_test_step_with_data_ensembling", "_ensemble_at_video_level", "on_test_epoch_end")


def lift_methods():
    tree = ast.parse(open(SRC).read())
    fns = {}
    for node in ast.walk(tree):
        if isinstance(node, ast.ClassDef) and node.name == "VideoClassificationModule":
            for item in node.body:
                if isinstance(item, ast.FunctionDef) and item.name in WANTED:
                    item.returns = None
                    for a in item.args.args:            # drop annotations (Batch, torch.Tensor ...): names not needed
                        a.annotation = None
                    mod = ast.Module(body=[item], type_ignores=[])
                    ns = {"torch": torch}
                    exec(compile(ast.fix_missing_locations(mod), SRC, "exec"), ns)
                    fns[item.name] = (ns[item.name], item.lineno, item.end_lineno)
    assert set(fns) == set(WANTED), sorted(fns)
    return fns


class StandIn:
    """The attributes the three methods read / write (set by the reference's __init__ and setup, :93-137)."""

    def __init__(self, fns, method, num_classes):
        self.ensemble_method, self.num_classes, self.device = method, num_classes, torch.device("cpu")
        self.modality_key = "video"
        self.video_preds, self.video_labels, self.video_clips_cnts = {}, {}, {}
        self.logged = None
        for name, (fn, _, _) in fns.items():
            setattr(self, name, fn.__get__(self))

    def __call__(self, x):           # `self(batch[self.modality_key])`: the model forward; here the logits themselves
        return x

    def _compute_metrics(self, video_preds, video_labels, phase):
        self.final = (video_preds.clone(), video_labels.clone())
        return {}

    def log_dict(self, d):
        self.logged = d


def main():
    fns = lift_methods()
    g = torch.Generator().manual_seed(21)
    C = 400
    batches = [(torch.randn(8, C, generator=g) * 3, [0, 0, 1, 5, 1, 0, 3, 3], [7, 7, 1, 9, 1, 7, 4, 4]),
               (torch.randn(5, C, generator=g) * 3, [3, 2, 2, 0, 5], [4, 2, 2, 7, 9]),
               (torch.randn(1, C, generator=g) * 3, [4], [11]),
               (torch.randn(30, C, generator=g) * 5, [6] * 30, [3] * 30)]        # one video's 10 clips x 3 crops
    out = {"source": {k: "%s:%d-%d" % (os.path.relpath(SRC, REF), a, b) for k, (_, a, b) in fns.items()},
           "batches": [(l, i, y) for l, i, y in batches], "methods": {}}
    for method in ("sum", "max"):
        s = StandIn(fns, method, C)
        for logits, ids, labels in batches:
            s._test_step_with_data_ensembling({"video": logits, "label": torch.tensor(labels), "video_index": ids}, 0)
        order = list(s.video_preds.keys())
        counts = dict(s.video_clips_cnts)
        s.on_test_epoch_end()
        preds, labels = s.final
        out["methods"][method] = {"video_order": order, "counts": [counts[v] for v in order], "video_preds": preds,
                                  "video_labels": labels}
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ensemble.pt")
    torch.save(out, dst)
    print("wrote", dst, out["source"], {m: (v["video_order"], v["counts"]) for m, v in out["methods"].items()})


if __name__ == "__main__":
    sys.exit(main())
