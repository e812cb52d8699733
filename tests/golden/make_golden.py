"""Generate tests/golden/ref_intree.npz by IMPORTING the reference's own Python code.

Runs only in the build container (needs /root/reference); the .npz it writes is committed so the
tests can run on the GPU box where /root/reference does not exist.

The reference's native rasterizer is an absent submodule, so what can be pinned from the reference
itself is everything *around* it that states the same maths in-tree:
  * Camera matrices            scene/cameras.py:17-59, utils/graphics_utils.py:38-71
  * cov3D_precomp (= transMat) gaussian_renderer/__init__.py:64-75 via GaussianModel.get_covariance
                               (scene/gaussian_model.py:27-33,118-119), utils/general_utils.py:78-110
  * activations                scene/gaussian_model.py:95-115
  * SH -> RGB                  utils/sh_utils.py:57-112 (+0.5, clamp_min 0: gaussian_renderer/__init__.py:88-91)
  * the exact argument list render() hands to GaussianRasterizer   gaussian_renderer/__init__.py:37-53,97-106
We call the reference's real render() with a stub `diff_surfel_rasterization` that records what it is
given (and answers with the CPU oracle's images so render()'s post-processing can run).

Also stores the fp64 oracle's forward outputs and input gradients for the same scene ("oracle_*" keys,
oracle-minted — NOT reference outputs) as regression vectors.

Usage:  python tests/golden/make_golden.py
"""
import os
import sys
import types
from typing import NamedTuple

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference"
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "2d-gaussian-splatting_amd"))

# ---------------------------------------------------------------- make the reference importable on CPU
for name in ["plyfile", "cv2", "matplotlib", "matplotlib.pyplot", "simple_knn", "simple_knn._C",
             "diff_surfel_rasterization"]:
    sys.modules[name] = types.ModuleType(name)
sys.modules["plyfile"].PlyData = object
sys.modules["plyfile"].PlyElement = object
sys.modules["simple_knn._C"].distCUDA2 = lambda x: None
sys.modules["matplotlib"].pyplot = sys.modules["matplotlib.pyplot"]
torch.Tensor.cuda = lambda self, *a, **k: self


def _cpu_factory(fn):
    def wrapped(*a, **k):
        if "device" in k and str(k["device"]).startswith("cuda"):
            k["device"] = "cpu"
        return fn(*a, **k)
    return wrapped


for _n in ["zeros", "ones", "tensor", "arange", "empty", "zeros_like", "ones_like", "rand", "randn", "full"]:
    setattr(torch, _n, _cpu_factory(getattr(torch, _n)))

CAPTURED = []


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


class GaussianRasterizer:
    def __init__(self, raster_settings):
        self.rs = raster_settings

    def __call__(self, means3D, means2D, shs=None, colors_precomp=None, opacities=None, scales=None, rotations=None,
                 cov3D_precomp=None):
        from oracle.surfel_oracle import Oracle
        rs = self.rs
        n = lambda t: None if t is None else t.detach().numpy()
        CAPTURED.append(dict(settings={k: (n(v) if torch.is_tensor(v) else v) for k, v in rs._asdict().items()},
                             means3D=n(means3D), shs=n(shs), colors_precomp=n(colors_precomp), opacities=n(opacities),
                             scales=n(scales), rotations=n(rotations), cov3D_precomp=n(cov3D_precomp)))
        o = Oracle("f64")
        R, col, oth, radii, st = o.rasterize_forward(n(rs.bg), n(means3D), n(colors_precomp), n(opacities), n(scales),
                                                     n(rotations), rs.scale_modifier, n(cov3D_precomp), n(rs.viewmatrix),
                                                     n(rs.projmatrix), rs.tanfovx, rs.tanfovy, rs.image_height,
                                                     rs.image_width, n(shs), rs.sh_degree, n(rs.campos))
        CAPTURED[-1]["oracle_state"] = st
        return (torch.tensor(col, dtype=torch.float32), torch.tensor(radii), torch.tensor(oth, dtype=torch.float32))


sys.modules["diff_surfel_rasterization"].GaussianRasterizationSettings = GaussianRasterizationSettings
sys.modules["diff_surfel_rasterization"].GaussianRasterizer = GaussianRasterizer

sys.path.insert(0, REF)
from gaussian_renderer import render                      # noqa: E402  (the reference's own render())
from scene.cameras import Camera                          # noqa: E402
from scene.gaussian_model import GaussianModel            # noqa: E402
from utils.sh_utils import eval_sh                        # noqa: E402

import synthetic                                          # noqa: E402


def main():
    torch.manual_seed(0)
    P, W, H = 512, 72, 56
    sc = synthetic.make_scene(P, W, H, seed=7, px_radius=4.0, z_near=1.0, z_far=6.0)
    # --- a reference Camera built from (R, T, FoV) exactly as dataset readers do
    Rt = sc["viewmatrix"].T.astype(np.float64)            # W2C
    R_c2w = Rt[:3, :3].T                                   # Camera takes R = C2W rotation (graphics_utils.py:38-41)
    T_w2c = Rt[:3, 3]
    import math
    fovx, fovy = 2 * math.atan(sc["tanfovx"]), 2 * math.atan(sc["tanfovy"])
    cam = Camera(colmap_id=0, R=R_c2w, T=T_w2c, FoVx=fovx, FoVy=fovy, image=torch.zeros(3, H, W), gt_alpha_mask=None,
                 image_name="synthetic", uid=0)
    # --- a reference GaussianModel holding raw (pre-activation) parameters
    pc = GaussianModel(3)
    pc.active_sh_degree = 3
    raw_rot = torch.tensor(sc["rotations"]) * torch.linspace(0.5, 2.0, P)[:, None]     # un-normalised on purpose
    pc._xyz = torch.tensor(sc["means3D"]).requires_grad_(True)
    pc._scaling = torch.log(torch.tensor(sc["scales"])).requires_grad_(True)
    pc._rotation = raw_rot.requires_grad_(True)
    pc._opacity = torch.logit(torch.tensor(sc["opacities"])).requires_grad_(True)
    pc._features_dc = torch.tensor(sc["shs"][:, :1]).contiguous().requires_grad_(True)
    pc._features_rest = torch.tensor(sc["shs"][:, 1:]).contiguous().requires_grad_(True)
    bg = torch.tensor([0.2, 0.5, 0.9])

    out = {}
    for flag in (False, True):
        pipe = types.SimpleNamespace(compute_cov3D_python=flag, convert_SHs_python=False, depth_ratio=0.0, debug=False)
        with torch.no_grad():
            rets = render(cam, pc, pipe, bg, scaling_modifier=1.0)
        cap = CAPTURED[-1]
        tag = "precomp" if flag else "native"
        if not flag:
            s = cap["settings"]
            out.update(image_height=s["image_height"], image_width=s["image_width"], tanfovx=s["tanfovx"],
                       tanfovy=s["tanfovy"], bg=s["bg"], scale_modifier=s["scale_modifier"], viewmatrix=s["viewmatrix"],
                       projmatrix=s["projmatrix"], sh_degree=s["sh_degree"], campos=s["campos"],
                       means3D=cap["means3D"], shs=cap["shs"], opacities=cap["opacities"], scales=cap["scales"],
                       rotations=cap["rotations"])
            st = cap["oracle_state"]
            out["ref_render_%s_surf_normal" % tag] = rets["surf_normal"].numpy()
            out["ref_render_%s_rend_normal" % tag] = rets["rend_normal"].numpy()
            out["ref_render_%s_surf_depth" % tag] = rets["surf_depth"].numpy()
        else:
            out["ref_cov3D_precomp"] = cap["cov3D_precomp"]          # the reference's own transMat
    # reference SH colours (gaussian_renderer/__init__.py:85-91, branch disabled at :82 but maths identical)
    with torch.no_grad():
        shs_view = pc.get_features.transpose(1, 2).view(-1, 3, 16)
        dir_pp = pc.get_xyz - cam.camera_center.repeat(P, 1)
        dirn = dir_pp / dir_pp.norm(dim=1, keepdim=True)
        for deg in range(4):
            out["ref_sh_rgb_deg%d" % deg] = torch.clamp_min(eval_sh(deg, shs_view, dirn) + 0.5, 0.0).numpy()
    out["raw_rotation"] = raw_rot.detach().numpy()

    # ------------------------------------------------ oracle-minted regression vectors (fp64)
    from oracle.surfel_oracle import Oracle
    o = Oracle("f64")
    R, col, oth, radii, st = o.rasterize_forward(out["bg"], out["means3D"], None, out["opacities"], out["scales"],
                                                 out["rotations"], 1.0, None, out["viewmatrix"], out["projmatrix"],
                                                 out["tanfovx"], out["tanfovy"], H, W, out["shs"], 3, out["campos"])
    rng = np.random.default_rng(11)
    gC = rng.normal(size=col.shape).astype(np.float32)
    gO = rng.normal(size=oth.shape).astype(np.float32)
    g = o.rasterize_backward(st, gC, gO)
    out.update(oracle_R=R, oracle_color=col, oracle_others=oth, oracle_radii=radii, oracle_depths=st.depths,
               oracle_transMat=st.transMat, oracle_xy=st.xy, oracle_normal_opacity=st.normal_opacity, oracle_rgb=st.rgb,
               grad_color=gC, grad_others=gO, oracle_dL_dmeans3D=g.dL_dmeans3D, oracle_dL_dmeans2D=g.dL_dmean2D,
               oracle_dL_dsh=g.dL_dsh, oracle_dL_dopacity=g.dL_dopacity, oracle_dL_dscales=g.dL_dscales,
               oracle_dL_drots=g.dL_drots)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_intree.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes; R =", R, "visible =", int((radii > 0).sum()))


if __name__ == "__main__":
    main()
