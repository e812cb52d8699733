"""Build libsurfel_hip.so (gfx950) in-tree with hipcc.  Usage: python build.py [--force]

The .so is git-ignored but travels to the GPU box with the gpurun snapshot.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib", "libsurfel_hip.so")
SOURCES = ["surfel_preprocess.hip", "surfel_forward.hip", "surfel_backward.hip", "surfel_backward_scan.hip", "surfel_sort.hip", "surfel_api.hip", "knn.hip", "box_probe.hip",
           "train_loss.hip", "train_post.hip", "train_fused.hip", "train_optim.hip", "train_api.hip"]
# blend kernels: packed-f32 VALU (SLP) costs ~1.6x a scalar op on gfx950 plus the v_movs that pair the operands
# surfel_backward.hip spells every fused multiply-add out (FMA macro) and is compiled with contraction off, so its kernel variants
# round identically per (pixel, surfel) pair
EXTRA = {"surfel_forward.hip": ["-fno-slp-vectorize"], "surfel_backward.hip": ["-fno-slp-vectorize", "-ffp-contract=off"],
         "surfel_backward_scan.hip": ["-fno-slp-vectorize", "-ffp-contract=off"]}
HEADERS = ["surfel_common.h", "surfel_kernels.h", "surfel_blend_bwd.h", "train_kernels.h", "train_loss_body.h", "train_post_body.h", os.path.join("..", "..", "include", "surfel_hip.h"), os.path.join("..", "..", "include", "surfel_debug.h"),
           os.path.join("..", "..", "include", "surfel_train.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-Wall", "-Wno-unused-result"]


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build_variant(tag, defines, verbose=False):
    """Diagnostic twin lib/libsurfel_hip_<tag>.so with extra -D flags (e.g. -DSCAN_TIMING); loaded only through SURFEL_LIB."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    bdir = os.path.join(HERE, "build_" + tag)
    os.makedirs(bdir, exist_ok=True)
    os.makedirs(os.path.join(HERE, "lib"), exist_ok=True)
    lib = os.path.join(HERE, "lib", "libsurfel_hip_%s.so" % tag)
    objs = []
    for src in SOURCES:
        obj = os.path.join(bdir, src + ".o")
        objs.append(obj)
        cmd = [hipcc] + FLAGS + list(defines) + EXTRA.get(src, []) + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    return lib


def build(force=False, verbose=False, ieee=False):
    """ieee=True: the diagnostic twin lib/libsurfel_hip_ieee.so (-DSURFEL_IEEE_MATH: IEEE division and libm expf in the blend kernels);
    never loaded by the product (surfel_native loads it only when SURFEL_LIB points at it)."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(os.path.join(HERE, "lib"), exist_ok=True)
    bdir = os.path.join(HERE, "build_ieee" if ieee else "build")
    os.makedirs(bdir, exist_ok=True)
    LIB = os.path.join(HERE, "lib", "libsurfel_hip_ieee.so" if ieee else "libsurfel_hip.so")
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        obj = os.path.join(bdir, src + ".o")
        objs.append(obj)
        if force or _stale(obj, [sp] + hdrs):
            cmd = [hipcc] + FLAGS + (["-DSURFEL_IEEE_MATH"] if ieee else []) + EXTRA.get(src, []) + ["-c", sp, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
    if force or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    if "--variant" in sys.argv:      # python build.py --variant TAG -DFOO -DBAR=1
        k = sys.argv.index("--variant")
        print(build_variant(sys.argv[k + 1], sys.argv[k + 2:], verbose=False))
        sys.exit(0)
    print(build(force="--force" in sys.argv, verbose=True, ieee="--ieee" in sys.argv))
