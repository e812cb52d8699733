#!/usr/bin/env python
"""Build lib/libsurfel_hip_<tag>.so from the csrc/ + include/ of another git revision (A/B baselines for scripts/ab_libs.py):
    python scripts/build_rev_lib.py <rev> <tag> [-DFLAG ...]
The library travels to the GPU box like the product's; it is loaded only through SURFEL_LIB."""
import os
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "2d-gaussian-splatting_amd"))
import build as B


def main():
    rev, tag, defs = sys.argv[1], sys.argv[2], sys.argv[3:]
    tmp = tempfile.mkdtemp(prefix="surfel_rev_")
    subprocess.check_call("git archive %s 2d-gaussian-splatting_amd/csrc include | tar -x -C %s" % (rev, tmp), shell=True, cwd=REPO)
    csrc = os.path.join(tmp, "2d-gaussian-splatting_amd", "csrc")
    srcs = [f for f in sorted(os.listdir(csrc)) if f.endswith(".hip")]
    objs = []
    for src in srcs:
        obj = os.path.join(tmp, src + ".o")
        objs.append(obj)
        subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + B.FLAGS + defs + B.EXTRA.get(src, []) + ["-c", os.path.join(csrc, src), "-o", obj])
    lib = os.path.join(REPO, "2d-gaussian-splatting_amd", "lib", "libsurfel_hip_%s.so" % tag)
    subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    print(lib)


if __name__ == "__main__":
    main()
