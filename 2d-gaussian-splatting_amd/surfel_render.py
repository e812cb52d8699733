"""The render boundary on MI355X — drop-in for `gaussian_renderer.render` (/root/reference/gaussian_renderer/__init__.py:19-158)
and the `Camera` it is handed (/root/reference/scene/cameras.py:17-59).

render() has the reference's signature and returns the reference's dict (render, viewspace_points, visibility_filter,
radii, rend_alpha, rend_normal, rend_dist, surf_depth, surf_normal).  What differs is the execution: the rasterizer
is libsurfel_hip.so, and everything the reference does to `allmap` afterwards (:118-147 + utils/point_utils.py:9-37,
~40 small PyTorch kernels forward and backward) is ONE HIP kernel each way (`render_post`).
"""
import math

import numpy as np
import torch

import surfel_native as _n
from diff_surfel_rasterization import GaussianRasterizationSettings, GaussianRasterizer

_n.load()


def _check(rc, what):
    if rc < 0:
        raise RuntimeError("%s failed (%d): %s" % (what, rc, _n.last_error()))
    return rc


# ------------------------------------------------------------------------------------------------ camera
def world2view(R, t, translate=(0.0, 0.0, 0.0), scale=1.0):
    """W2C 4x4 from the C2W rotation R and the W2C translation t, with the reference's optional scene
    re-centring applied to the camera centre (utils/graphics_utils.py:38-49)."""
    Rt = np.eye(4)
    Rt[:3, :3] = np.asarray(R, np.float64).T
    Rt[:3, 3] = np.asarray(t, np.float64)
    c2w = np.linalg.inv(Rt)
    c2w[:3, 3] = (c2w[:3, 3] + np.asarray(translate, np.float64)) * scale
    return np.linalg.inv(c2w).astype(np.float32)


def projection(znear, zfar, fovx, fovy):
    """OpenGL-style perspective with w_clip = z_view (utils/graphics_utils.py:51-71)."""
    tx, ty = math.tan(fovx / 2), math.tan(fovy / 2)
    P = np.zeros((4, 4), np.float32)
    P[0, 0] = 1.0 / tx
    P[1, 1] = 1.0 / ty
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def post_consts(world_view_transform, full_proj_transform, W, H):
    """The 24-float camera block of include/surfel_train.h, from the camera's two matrices with the formulas of
    utils/point_utils.py:10-22 and gaussian_renderer/__init__.py:123 (fp64 on the host, once per camera)."""
    wvt = np.asarray(world_view_transform, np.float64); fpt = np.asarray(full_proj_transform, np.float64)
    c2w = np.linalg.inv(wvt.T)
    ndc2pix = np.array([[W / 2, 0, 0, W / 2], [0, H / 2, 0, H / 2], [0, 0, 0, 1]], np.float64).T
    intrins = ((c2w.T @ fpt) @ ndc2pix)[:3, :3].T
    K = np.linalg.inv(intrins).T @ c2w[:3, :3].T
    out = np.zeros(24, np.float32)
    out[0:9] = wvt[:3, :3].reshape(-1)
    out[9:18] = K.reshape(-1)
    out[18:21] = c2w[:3, 3]
    return out


def post_consts_rows(consts, row0):
    """The camera block for a sub-image that starts at image row `row0` (tile-band sharding): the pixel ray of local row y is
    (x, y + row0, 1) @ K, i.e. K's third row gains row0 x its second row; everything else is unchanged."""
    if row0 == 0:
        return consts
    c = consts.clone()
    c[15:18] = consts[15:18] + float(row0) * consts[12:15]
    return c


class Camera:
    """Same attributes as the reference's Camera (scene/cameras.py:17-59): R = C2W rotation, T = W2C translation,
    matrices stored transposed (row-vector convention) on the device, znear 0.01 / zfar 100."""

    def __init__(self, colmap_id, R, T, FoVx, FoVy, image, gt_alpha_mask=None, image_name="", uid=0,
                 trans=np.array([0.0, 0.0, 0.0]), scale=1.0, data_device="cuda"):
        self.uid, self.colmap_id, self.R, self.T, self.FoVx, self.FoVy, self.image_name = uid, colmap_id, R, T, FoVx, FoVy, image_name
        self.data_device = torch.device(data_device)
        self.original_image = image.clamp(0.0, 1.0).to(self.data_device)
        self.image_width, self.image_height = int(self.original_image.shape[2]), int(self.original_image.shape[1])
        self.gt_alpha_mask = None if gt_alpha_mask is None else gt_alpha_mask.to(self.data_device)
        self.zfar, self.znear = 100.0, 0.01
        self.trans, self.scale = trans, scale
        w2c = world2view(R, T, trans, scale)
        proj = projection(self.znear, self.zfar, FoVx, FoVy)
        wvt = torch.tensor(w2c).transpose(0, 1).contiguous()
        pm = torch.tensor(proj).transpose(0, 1).contiguous()
        full = wvt @ pm
        self.world_view_transform = wvt.to(self.data_device)
        self.projection_matrix = pm.to(self.data_device)
        self.full_proj_transform = full.to(self.data_device)
        self.camera_center = torch.linalg.inv(wvt)[3, :3].contiguous().to(self.data_device)
        self._post = None

    def post_consts(self):
        if self._post is None:
            self._post = torch.tensor(post_consts(self.world_view_transform.cpu().numpy(), self.full_proj_transform.cpu().numpy(),
                                                  self.image_width, self.image_height)).to(self.data_device)
        return self._post


def _cam_consts(view):
    if hasattr(view, "post_consts"):
        return view.post_consts()
    c = getattr(view, "_surfel_post_consts", None)     # any camera-like object (e.g. the reference's Camera / MiniCam)
    if c is None:
        c = torch.tensor(post_consts(view.world_view_transform.detach().cpu().numpy(), view.full_proj_transform.detach().cpu().numpy(),
                                     int(view.image_width), int(view.image_height))).to(view.world_view_transform.device)
        try:
            view._surfel_post_consts = c
        except Exception:
            pass
    return c


# ------------------------------------------------------------------------------------------------ allmap post-processing
class _RenderPost(torch.autograd.Function):
    """allmap [7,H,W] -> maps [9,H,W] (0 alpha | 1-3 rend_normal | 4 dist | 5 surf_depth | 6-8 surf_normal)."""

    @staticmethod
    def forward(ctx, allmap, cam, depth_ratio):
        if allmap.device.type != "cuda":
            raise RuntimeError("render_post: tensors must live on a HIP device (got %s)" % allmap.device)
        am = allmap.detach().contiguous().float()
        H, W = int(am.shape[1]), int(am.shape[2])
        maps = torch.empty((9, H, W), dtype=torch.float32, device=am.device)
        with torch.cuda.device(am.device):
            _check(_n.load().surfel_render_post_forward(H, W, _n.ptr(am), _n.ptr(cam), float(depth_ratio), _n.ptr(maps), None,
                                                        _n.current_stream_ptr(am.device)), "surfel_render_post_forward")
        ctx.save_for_backward(am, cam)
        ctx.ratio = float(depth_ratio)
        return maps

    @staticmethod
    def backward(ctx, g_maps):
        am, cam = ctx.saved_tensors
        H, W = int(am.shape[1]), int(am.shape[2])
        g = g_maps.contiguous().float()
        out = torch.empty_like(am)
        with torch.cuda.device(am.device):
            _check(_n.load().surfel_render_post_backward(H, W, _n.ptr(am), _n.ptr(cam), ctx.ratio, _n.ptr(g), 0.0, 0.0, None, _n.ptr(out),
                                                         _n.current_stream_ptr(am.device)), "surfel_render_post_backward")
        return out, None, None


def render_post(allmap, viewpoint_camera, depth_ratio):
    return _RenderPost.apply(allmap, _cam_consts(viewpoint_camera), depth_ratio)


class _Regularizers(torch.autograd.Function):
    """lambda_normal * mean(1 - rend_normal . surf_normal) + lambda_dist * mean(rend_dist) (train.py:80-85) straight from
    allmap: one forward kernel (+ a fixed-order reduction) and one backward kernel, no intermediate maps kept."""

    @staticmethod
    def forward(ctx, allmap, cam, depth_ratio, lambda_normal, lambda_dist):
        am = allmap.detach().contiguous().float()
        dev = am.device
        H, W = int(am.shape[1]), int(am.shape[2])
        lib = _n.load()
        nblk = ((W + 15) // 16) * ((H + 15) // 16)
        maps = torch.empty((9, H, W), dtype=torch.float32, device=dev)
        partials = torch.empty((nblk, 2), dtype=torch.float32, device=dev)
        means = torch.empty((2,), dtype=torch.float32, device=dev)
        s = _n.current_stream_ptr(dev)
        with torch.cuda.device(dev):
            _check(lib.surfel_render_post_forward(H, W, _n.ptr(am), _n.ptr(cam), float(depth_ratio), _n.ptr(maps), _n.ptr(partials), s),
                   "surfel_render_post_forward")
            _check(lib.surfel_reduce_partials(_n.ptr(partials), 1, nblk, 2, 1.0 / (H * W), _n.ptr(means), s), "surfel_reduce_partials")
        ctx.save_for_backward(am, cam)
        ctx.k = (float(depth_ratio), float(lambda_normal), float(lambda_dist))
        ctx.mark_non_differentiable(means)
        return lambda_normal * means[0] + lambda_dist * means[1], means

    @staticmethod
    def backward(ctx, g_loss, g_means):
        am, cam = ctx.saved_tensors
        ratio, ln, ld = ctx.k
        H, W = int(am.shape[1]), int(am.shape[2])
        g = g_loss.contiguous().float().reshape(1)
        out = torch.empty_like(am)
        with torch.cuda.device(am.device):
            _check(_n.load().surfel_render_post_backward(H, W, _n.ptr(am), _n.ptr(cam), ratio, None, ln / (H * W), ld / (H * W), _n.ptr(g),
                                                         _n.ptr(out), _n.current_stream_ptr(am.device)), "surfel_render_post_backward")
        return out, None, None, None, None


def regularizers(allmap, viewpoint_camera, depth_ratio, lambda_normal, lambda_dist):
    """Returns (normal_loss + dist_loss, means) with means = [mean normal error, mean distortion] (detached)."""
    return _Regularizers.apply(allmap, _cam_consts(viewpoint_camera), depth_ratio, lambda_normal, lambda_dist)


# ------------------------------------------------------------------------------------------------ render()
def rasterize(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, override_color=None, zero_means2D=True, band=None, debug_bits=0):
    """The rasterizer call of render() (gaussian_renderer/__init__.py:27-106): returns (image, radii, allmap, means2D).
    means2D is only the sink of the densification statistic (its values are never read); zero_means2D=False skips the fill.
    band = (y0, y1), multiples of 16: render only those image rows (tile-band sharding, surfel_dist.band_settings).
    debug_bits: per-call library options (surfel_native.OPT_*) OR-ed into the settings' debug word."""
    means3D = pc.get_xyz
    screenspace_points = torch.zeros_like(means3D, requires_grad=True) if zero_means2D else torch.empty_like(means3D).requires_grad_(True)
    raster_settings = GaussianRasterizationSettings(
        image_height=int(viewpoint_camera.image_height), image_width=int(viewpoint_camera.image_width),
        tanfovx=math.tan(viewpoint_camera.FoVx * 0.5), tanfovy=math.tan(viewpoint_camera.FoVy * 0.5), bg=bg_color,
        scale_modifier=scaling_modifier, viewmatrix=viewpoint_camera.world_view_transform,
        projmatrix=viewpoint_camera.full_proj_transform, sh_degree=pc.active_sh_degree, campos=viewpoint_camera.camera_center,
        prefiltered=False, debug=int(getattr(pipe, "debug", 0)) | int(debug_bits))
    if band is not None:
        import surfel_dist
        raster_settings = surfel_dist.band_settings(raster_settings, int(band[0]), int(band[1]))
    rasterizer = GaussianRasterizer(raster_settings=raster_settings)
    scales = rotations = cov3D_precomp = None
    if getattr(pipe, "compute_cov3D_python", False):
        if band is not None:
            raise ValueError("tile-band sharding needs the native homography (compute_cov3D_python builds full-image pixel matrices)")
        # the reference's python homography (gaussian_renderer/__init__.py:64-75)
        splat2world = pc.get_covariance(scaling_modifier)
        W, H = viewpoint_camera.image_width, viewpoint_camera.image_height
        near, far = viewpoint_camera.znear, viewpoint_camera.zfar
        ndc2pix = torch.tensor([[W / 2, 0, 0, (W - 1) / 2], [0, H / 2, 0, (H - 1) / 2], [0, 0, far - near, near], [0, 0, 0, 1]],
                               dtype=torch.float32, device=means3D.device).T
        world2pix = viewpoint_camera.full_proj_transform @ ndc2pix
        cov3D_precomp = (splat2world[:, [0, 1, 3]] @ world2pix[:, [0, 1, 3]]).permute(0, 2, 1).reshape(-1, 9)
    else:
        scales, rotations = pc.get_scaling, pc.get_rotation
    shs = colors_precomp = None
    if override_color is None:
        shs = pc.get_features
    else:
        colors_precomp = override_color
    image, radii, allmap = rasterizer(means3D=means3D, means2D=screenspace_points, shs=shs, colors_precomp=colors_precomp,
                                      opacities=pc.get_opacity, scales=scales, rotations=rotations, cov3D_precomp=cov3D_precomp)
    return image, radii, allmap, screenspace_points


def rasterize_manual(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, debug_bits=0):
    """The rasterizer call of a training iteration without autograd (surfel_native.ManualCtx): the store's raw views go straight
    into the forward, rasterize_manual_backward(ctx, dL/dimage, dL/dallmap) runs the backward — it writes the model's bound gradient
    store (GaussianModel.bind) like the autograd path — and returns dL/dmeans2D, the densification statistic.  Native homography
    and SH colours only (what train.py uses).  Returns (ctx, image, radii, allmap).  Call under torch.no_grad()."""
    from diff_surfel_rasterization import _RasterizeGaussians
    import surfel_native as _n
    P = pc.P
    rs = GaussianRasterizationSettings(
        image_height=int(viewpoint_camera.image_height), image_width=int(viewpoint_camera.image_width),
        tanfovx=math.tan(viewpoint_camera.FoVx * 0.5), tanfovy=math.tan(viewpoint_camera.FoVy * 0.5), bg=bg_color,
        scale_modifier=scaling_modifier, viewmatrix=viewpoint_camera.world_view_transform,
        projmatrix=viewpoint_camera.full_proj_transform, sh_degree=pc.active_sh_degree, campos=viewpoint_camera.camera_center,
        prefiltered=False, debug=int(getattr(pipe, "debug", 0)) | int(debug_bits))
    ctx = _n.ManualCtx()
    none = pc._pv["xyz"].new_empty(0)
    image, radii, allmap = _RasterizeGaussians.forward(ctx, pc._pv["xyz"], none, pc._pv["sh"].view(P, 16, 3), none, pc._av["opacity"],
                                                       pc._av["scaling"], pc._av["rotation"], none, rs)
    return ctx, image, radii, allmap


def rasterize_manual_backward(ctx, grad_image, grad_allmap):
    from diff_surfel_rasterization import _RasterizeGaussians
    return _RasterizeGaussians.backward(ctx, grad_image, None, grad_allmap)[1]


def render(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, override_color=None):
    """Render the scene; background tensor must be on the GPU.  Same dict as the reference's render()."""
    image, radii, allmap, means2D = rasterize(viewpoint_camera, pc, pipe, bg_color, scaling_modifier, override_color)
    maps = render_post(allmap, viewpoint_camera, float(getattr(pipe, "depth_ratio", 0.0)))
    return {"render": image, "viewspace_points": means2D, "visibility_filter": radii > 0, "radii": radii,
            "rend_alpha": maps[0:1], "rend_normal": maps[1:4], "rend_dist": maps[4:5], "surf_depth": maps[5:6],
            "surf_normal": maps[6:9], "allmap": allmap}
