"""Build libpv_mi355x.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

    python -m pytorchvideo_amd.csrc.build [--force]

Objects are rebuilt only when their source (or a header) is newer.  hipcc cross-compiles
without a GPU, so this also runs in the CPU-only build container.
"""
import concurrent.futures as cf
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
INCLUDE = os.path.join(ROOT, "include")
OUT_DIR = os.path.join(os.path.dirname(HERE), "_lib")
LIB = os.path.join(OUT_DIR, "libpv_mi355x.so")
SOURCES = ["pv_conv.hip", "pv_pwconv.hip", "pv_gemm.hip", "pv_gemm8.hip", "pv_gemm9.hip", "pv_gemm9h.hip", "pv_stem.hip", "pv_dwconv.hip", "pv_pwdw.hip", "pv_misc.hip", "pv_tokpool.hip", "pv_attn.hip", "pv_attn64.hip", "pv_roi.hip", "pv_lateral.hip", "pv_mlp.hip", "pv_block.hip", "pv_headgemm.hip", "pv_comm.hip", "pv_plan.hip"]
HEADERS = [os.path.join(HERE, "pv_common.h"), os.path.join(INCLUDE, "pv_mi355x.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", INCLUDE, "-I", HERE,
         "-Wno-unused-result"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


# Debug variants (sanitizer / bounds-check builds of the native code), built on demand next to the product library:
#   "asan": the HOST side of every translation unit (descriptor validation, kernel routing, the launch plan, the
#           communicator) instrumented by AddressSanitizer (device code is left alone: -fno-gpu-sanitize); loaded by a
#           python started with LD_PRELOAD=<ASAN_RUNTIME> and PV_MI355X_LIB=<variant library> (tests/test_sanitizers.py).
#   "dev":  -DPV_DEV_ABLATION: the ablation builds of the GEMM / fused-MLP kernels (timing only, they skip loads, MFMAs or the
#           epilogue and give WRONG results) and their A/B variants, selectable through pv_tune_set("gemm_abl" | "mlp_abl").  The
#           PRODUCT library does not contain them: no knob of the public ABI can make it compute a wrong answer.
VARIANTS = {
    "asan": ["-fsanitize=address", "-fno-gpu-sanitize", "-fno-omit-frame-pointer", "-g", "-O1"],
    "dev": ["-DPV_DEV_ABLATION", "-O3"],
}


def asan_runtime():
    """libclang_rt.asan of the ROCm clang that built the variant (LD_PRELOAD for the python that loads it)."""
    import glob
    hits = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
    return hits[-1] if hits else None


def _compile(src, force, out_dir=None, extra=()):
    out_dir = out_dir or OUT_DIR
    obj = os.path.join(out_dir, src.replace(".hip", ".o"))
    path = os.path.join(HERE, src)
    if force or _stale(obj, [path] + HEADERS):
        flags = [f for f in FLAGS if not (extra and f == "-O3")] + list(extra)
        cmd = [HIPCC] + flags + ["-c", path, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
        return obj, True
    return obj, False


def build(force=False, verbose=True):
    os.makedirs(OUT_DIR, exist_ok=True)
    with cf.ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        results = list(ex.map(lambda s: _compile(s, force), SOURCES))
    objs = [o for o, _ in results]
    if force or any(changed for _, changed in results) or not os.path.exists(LIB):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
        if verbose:
            print("built", LIB)
    elif verbose:
        print("up to date:", LIB)
    return LIB


def build_variant(name, force=False, verbose=True):
    """Build libpv_mi355x.so with the flags of VARIANTS[name] under _lib/<name>/ (objects cached like the product build)."""
    extra = VARIANTS[name]
    out_dir = os.path.join(OUT_DIR, name)
    os.makedirs(out_dir, exist_ok=True)
    lib = os.path.join(out_dir, "libpv_mi355x.so")
    with cf.ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        results = list(ex.map(lambda s: _compile(s, force, out_dir, extra), SOURCES))
    if force or any(changed for _, changed in results) or not os.path.exists(lib):
        link = [f for f in extra if f.startswith("-fsanitize") or f == "-fno-gpu-sanitize"]
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + [o for o, _ in results] + link + ["-ldl"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
        if verbose:
            print("built", lib)
    return lib


if __name__ == "__main__":
    if "--variant" in sys.argv:
        build_variant(sys.argv[sys.argv.index("--variant") + 1], force="--force" in sys.argv)
    else:
        build(force="--force" in sys.argv)
