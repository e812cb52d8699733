"""Build container only (needs /root/reference; there is no GPU here): the reference's OWN render()
(/root/reference/gaussian_renderer/__init__.py:19-158) and its own GaussianModel / Camera run against THIS package's
`diff_surfel_rasterization` and `simple_knn._C` modules — the literal import and call sites of the drop-in boundary
(gaussian_renderer/__init__.py:14, :37-53, :97-106) — with only the one native call stubbed: `rasterize_gaussians` is replaced by
a recorder that checks what reaches the C-ABI wrapper and returns tensors of the documented shapes.  Everything above it (settings
NamedTuple, GaussianRasterizer.__init__/forward and its "exactly one of" checks) is the product's code; everything around it is
the reference's.  Prints one JSON line.  Run by tests/test_dropin_reference_cpu.py in a subprocess (it monkeypatches torch
factories to the CPU, as tests/golden/make_golden_train.py does)."""
import json
import math
import os
import sys
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference"
sys.path.insert(0, os.path.join(REPO, "2d-gaussian-splatting_amd"))
sys.path.insert(0, REF)

for name in ["plyfile", "cv2", "matplotlib", "matplotlib.pyplot"]:
    sys.modules[name] = types.ModuleType(name)
sys.modules["plyfile"].PlyData = object
sys.modules["plyfile"].PlyElement = object
sys.modules["matplotlib"].pyplot = sys.modules["matplotlib.pyplot"]
torch.Tensor.cuda = lambda self, *a, **k: self


def _cpu_factory(fn):
    def wrapped(*a, **k):
        if "device" in k and str(k["device"]).startswith("cuda"):
            k["device"] = "cpu"
        return fn(*a, **k)
    return wrapped


for _name in ["zeros", "ones", "tensor", "arange", "empty", "zeros_like", "ones_like", "rand", "randn", "full"]:
    setattr(torch, _name, _cpu_factory(getattr(torch, _name)))

import diff_surfel_rasterization as dsr          # noqa: E402  the product's module (loads libsurfel_hip.so: no fallback)
import simple_knn._C as knn                       # noqa: E402  the product's module
assert dsr.__file__.startswith(REPO) and knn.__file__.startswith(REPO)

CALLS = []


def recorder(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings):
    rs = raster_settings
    P = means3D.shape[0]
    assert isinstance(rs, dsr.GaussianRasterizationSettings)
    assert means2D.shape == (P, 3) and means2D.requires_grad and opacities.shape == (P, 1)
    assert (sh is None) != (colors_precomp is None)
    assert (scales is None) == (rotations is None) and (scales is None) != (cov3Ds_precomp is None)
    if sh is not None:
        assert sh.shape == (P, 16, 3)
    if colors_precomp is not None:
        assert colors_precomp.shape == (P, 3)
    if scales is not None:
        assert scales.shape == (P, 2) and rotations.shape == (P, 4)          # 2DGS: two scales per surfel
    else:
        assert cov3Ds_precomp.shape == (P, 9)
    assert rs.viewmatrix.shape == (4, 4) and rs.projmatrix.shape == (4, 4) and rs.campos.shape == (3,) and rs.bg.shape == (3,)
    assert rs.prefiltered is False and rs.debug is False and isinstance(rs.sh_degree, int)
    CALLS.append(dict(sh=sh is not None, colors=colors_precomp is not None, scales=scales is not None, cov=cov3Ds_precomp is not None,
                      sh_degree=int(rs.sh_degree), H=int(rs.image_height), W=int(rs.image_width), scale_modifier=float(rs.scale_modifier)))
    H, W = rs.image_height, rs.image_width
    g = torch.Generator().manual_seed(P + H)
    color = torch.rand((3, H, W), generator=g) + 0 * means3D.sum()
    allmap = torch.rand((7, H, W), generator=g) + 0.5 + 0 * opacities.sum()
    radii = torch.arange(P, dtype=torch.int32) % 3
    return color, radii, allmap


dsr.rasterize_gaussians = recorder

from gaussian_renderer import render            # noqa: E402  the reference's render(), importing the product's modules
from scene.cameras import Camera                # noqa: E402
from scene.gaussian_model import GaussianModel  # noqa: E402

P, H, W = 50, 24, 32
rng = np.random.default_rng(0)
pc = GaussianModel(3)
pc.active_sh_degree = 2
par = lambda a: torch.nn.Parameter(torch.tensor(a, dtype=torch.float32).requires_grad_(True))
pc._xyz = par(rng.normal(size=(P, 3)))
pc._features_dc = par(rng.normal(size=(P, 1, 3)))
pc._features_rest = par(rng.normal(size=(P, 15, 3)))
pc._scaling = par(rng.normal(size=(P, 2)) - 3.0)
pc._rotation = par(rng.normal(size=(P, 4)))
pc._opacity = par(rng.normal(size=(P, 1)))
cam = Camera(colmap_id=0, R=np.eye(3), T=np.array([0.0, 0.0, 4.0]), FoVx=math.radians(50), FoVy=math.radians(40),
             image=torch.zeros((3, H, W)), gt_alpha_mask=None, image_name="v", uid=0, data_device="cpu")
bg = torch.zeros(3)
out = {"calls": None, "keys": None}
for cov_py, override in ((False, None), (True, None), (False, torch.rand(P, 3))):
    pipe = types.SimpleNamespace(compute_cov3D_python=cov_py, convert_SHs_python=False, depth_ratio=0.0, debug=False)
    pkg = render(cam, pc, pipe, bg, scaling_modifier=0.9 if cov_py else 1.0, override_color=override)
    assert pkg["render"].shape == (3, H, W) and pkg["radii"].shape == (P,) and pkg["visibility_filter"].dtype == torch.bool
    for k in ("rend_alpha", "rend_normal", "rend_dist", "surf_depth", "surf_normal"):
        assert k in pkg, k
    assert pkg["viewspace_points"].shape == (P, 3)
    (pkg["render"].sum() + pkg["rend_normal"].sum()).backward()      # the reference's graph closes through the boundary tensors
    out["keys"] = sorted(pkg.keys())
# a wrong combination must be rejected by the product's GaussianRasterizer.forward with the reference's exceptions
settings = dsr.GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=1.0, tanfovy=1.0, bg=bg, scale_modifier=1.0,
                                             viewmatrix=torch.eye(4), projmatrix=torch.eye(4), sh_degree=0, campos=torch.zeros(3),
                                             prefiltered=False, debug=False)
rast = dsr.GaussianRasterizer(raster_settings=settings)
rejected = 0
for kw in (dict(), dict(shs=pc.get_features, colors_precomp=torch.rand(P, 3)), dict(shs=pc.get_features, scales=pc.get_scaling),
           dict(shs=pc.get_features, scales=pc.get_scaling, rotations=pc.get_rotation, cov3D_precomp=torch.rand(P, 9))):
    try:
        rast(means3D=pc.get_xyz, means2D=torch.zeros(P, 3, requires_grad=True), opacities=pc.get_opacity, **kw)
    except Exception as e:      # noqa: BLE001
        rejected += "Please provide" in str(e)
out["calls"] = CALLS
out["rejected"] = rejected
out["distCUDA2_is_product"] = callable(knn.distCUDA2)
print(json.dumps(out))
