import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_kernels as K
_re = K.rel_err
vals = []
def _pr(a, b):
    v = _re(a, b); vals.append(float(v)); return v
K.rel_err = _pr
for shape in ((2, 4, 10, 10), (2, 3, 10, 10), (1, 3, 10, 10), (2, 2, 10, 10), (2, 5, 10, 10), (3, 3, 10, 10), (4, 3, 10, 10)):
    try:
        K.test_fused_bottleneck_conv_ab_with_squeeze_sums((96, 216), *shape)
        print(shape, "ok", ["%.2e" % v for v in vals], flush=True); vals.clear()
    except AssertionError as e:
        import traceback
        tb = traceback.extract_tb(e.__traceback__)[-1]
        print(["%.2e" % v for v in vals]); vals.clear()
        print(shape, "FAILED at line %d: %s | %s" % (tb.lineno, tb.line, str(e)[:200].replace("\n", " ")), flush=True)
