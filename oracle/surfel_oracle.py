"""ctypes front-end of oracle/surfel_oracle.c — the CPU checker for the surfel rasterizer.

TEST INFRASTRUCTURE ONLY (parity unpinned, see the header of surfel_oracle.c).  Imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg; never by the product package.

`rasterize_forward` / `rasterize_backward` mirror the argument meaning of the reference's native
entry points `_C.rasterize_gaussians` / `_C.rasterize_gaussians_backward` (call site
/root/reference/gaussian_renderer/__init__.py:97-106) but on numpy arrays.
"""
import ctypes as C
import os
import subprocess
from types import SimpleNamespace

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


class _Params(C.Structure):
    _fields_ = [("P", C.c_int), ("D", C.c_int), ("M", C.c_int), ("W", C.c_int), ("H", C.c_int),
                ("tan_fovx", C.c_double), ("tan_fovy", C.c_double), ("scale_modifier", C.c_double)]


def build(force=False):
    """Compile liboracle_f64.so / liboracle_f32.so with gcc (oracle/Makefile)."""
    libs = [os.path.join(_HERE, n) for n in ("liboracle_f64.so", "liboracle_f32.so", "liboracle_mca.so")]
    src = os.path.join(_HERE, "surfel_oracle.c")
    if force or any((not os.path.exists(l)) or os.path.getmtime(l) < os.path.getmtime(src) for l in libs):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s", "all"])
    return libs


_LIBS = {}


def _lib(precision):
    if precision not in _LIBS:
        build()
        _LIBS[precision] = C.CDLL(os.path.join(_HERE, "liboracle_%s.so" % precision))
        _LIBS[precision].oracle_preprocess.restype = C.c_int64
    return _LIBS[precision]


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return None if a is None else np.ascontiguousarray(np.asarray(a, dtype=np.float32))


class Oracle:
    """precision: 'f64' (checker), 'f32' (timed CPU port) or 'mca' (Monte Carlo arithmetic: every operation of the algorithm carries a
    random fp32-sized rounding error — surfel_oracle.c, ORACLE_MCA; set_seed() picks the draw)."""

    def __init__(self, precision="f64"):
        self.lib = _lib(precision)
        self.precision = precision
        self.real = np.float32 if precision == "f32" else np.float64
        assert self.lib.oracle_real_size() == np.dtype(self.real).itemsize

    def set_seed(self, seed):
        assert self.precision == "mca"
        self.lib.oracle_set_mca_seed(C.c_uint64(int(seed)))

    def _params(self, P, D, M, W, H, tanfovx, tanfovy, scale_modifier):
        return _Params(P, D, M, W, H, tanfovx, tanfovy, scale_modifier)

    # ------------------------------------------------------------------ forward
    def rasterize_forward(self, bg, means3D, colors_precomp, opacities, scales, rotations, scale_modifier,
                          transMat_precomp, viewmatrix, projmatrix, tanfovx, tanfovy, H, W, sh, degree, campos,
                          depth_key_f32=None):
        real = self.real
        means3D = _f32(means3D); P = means3D.shape[0]
        bg = _f32(bg); opacities = _f32(opacities).reshape(-1)
        scales = _f32(scales); rotations = _f32(rotations); transMat_precomp = _f32(transMat_precomp)
        colors_precomp = _f32(colors_precomp); sh = _f32(sh)
        viewmatrix = _f32(viewmatrix); projmatrix = _f32(projmatrix); campos = _f32(campos)
        M = sh.shape[1] if sh is not None else 0
        prm = self._params(P, int(degree), M, int(W), int(H), tanfovx, tanfovy, scale_modifier)
        st = SimpleNamespace()
        st.prm = prm
        st.inputs = dict(bg=bg, means3D=means3D, colors_precomp=colors_precomp, opacities=opacities, scales=scales,
                         rotations=rotations, transMat_precomp=transMat_precomp, viewmatrix=viewmatrix,
                         projmatrix=projmatrix, sh=sh, campos=campos)
        st.depths = np.zeros(P, real); st.radii = np.zeros(P, np.int32); st.xy = np.zeros((P, 2), real)
        st.transMat = np.zeros((P, 9), real); st.normal_opacity = np.zeros((P, 4), real)
        st.rgb = np.zeros((P, 3), real); st.clamped = np.zeros((P, 3), np.uint8)
        st.tiles_touched = np.zeros(P, np.uint32)
        # decision signatures: one word per pixel / surfel identifying the decisions taken there; the AABB extent in front of ceil()
        st.sig_pix = np.zeros((H, W), np.uint64); st.sig_surf = np.zeros(P, np.uint64); st.extent = np.zeros(P, np.float64)
        self.lib.oracle_set_signature_buffers(_p(st.sig_pix), _p(st.sig_surf), _p(st.extent))
        R = self.lib.oracle_preprocess(C.byref(prm), _p(means3D), _p(opacities), _p(scales), _p(rotations),
                                       _p(transMat_precomp), _p(colors_precomp), _p(sh), _p(viewmatrix),
                                       _p(projmatrix), _p(campos), _p(st.depths), _p(st.radii), _p(st.xy),
                                       _p(st.transMat), _p(st.normal_opacity), _p(st.rgb), _p(st.clamped),
                                       _p(st.tiles_touched))
        st.R = int(R)
        gx, gy = (W + 15) // 16, (H + 15) // 16
        st.point_list = np.zeros(max(st.R, 1), np.uint32)
        st.keys = np.zeros(max(st.R, 1), np.uint64)
        st.ranges = np.zeros((gx * gy, 2), np.uint32)
        dk = _f32(depth_key_f32)
        rc = self.lib.oracle_bin(C.byref(prm), _p(st.depths), _p(dk), _p(st.radii), _p(st.xy), C.c_int64(st.R),
                                 _p(st.point_list), _p(st.keys), _p(st.ranges))
        assert rc == 0, "oracle_bin failed: %d" % rc
        st.out_color = np.zeros((3, H, W), real); st.out_others = np.zeros((7, H, W), real)
        st.final_T = np.zeros((3, H, W), real); st.n_contrib = np.zeros((2, H, W), np.uint32)
        self.lib.oracle_blend_forward(C.byref(prm), _p(st.ranges), _p(st.point_list), _p(st.xy), _p(st.transMat),
                                      _p(transMat_precomp), _p(st.normal_opacity), _p(st.rgb), _p(colors_precomp),
                                      _p(bg), _p(st.out_color), _p(st.out_others), _p(st.final_T), _p(st.n_contrib))
        self.lib.oracle_set_signature_buffers(None, None, None)
        return st.R, st.out_color, st.out_others, st.radii, st

    def preprocess_extents(self, means3D, colors_precomp, opacities, scales, rotations, scale_modifier, transMat_precomp, viewmatrix,
                           projmatrix, tanfovx, tanfovy, H, W, sh, degree, campos):
        """Stage 1 alone (oracle_preprocess): (radii, AABB half-extent in front of ceil()) of every surfel.  Tests use extra
        Monte-Carlo-arithmetic draws of this cheap stage to pin which radii fp32 determines (tests/determinacy.py)."""
        real = self.real
        means3D = _f32(means3D); P = means3D.shape[0]
        opacities = _f32(opacities).reshape(-1)
        scales = _f32(scales); rotations = _f32(rotations); transMat_precomp = _f32(transMat_precomp)
        colors_precomp = _f32(colors_precomp); sh = _f32(sh)
        viewmatrix = _f32(viewmatrix); projmatrix = _f32(projmatrix); campos = _f32(campos)
        M = sh.shape[1] if sh is not None else 0
        prm = self._params(P, int(degree), M, int(W), int(H), tanfovx, tanfovy, scale_modifier)
        depths = np.zeros(P, real); radii = np.zeros(P, np.int32); xy = np.zeros((P, 2), real)
        transMat = np.zeros((P, 9), real); normal_opacity = np.zeros((P, 4), real)
        rgb = np.zeros((P, 3), real); clamped = np.zeros((P, 3), np.uint8); tiles_touched = np.zeros(P, np.uint32)
        extent = np.zeros(P, np.float64)
        self.lib.oracle_set_signature_buffers(None, None, _p(extent))
        self.lib.oracle_preprocess(C.byref(prm), _p(means3D), _p(opacities), _p(scales), _p(rotations), _p(transMat_precomp),
                                   _p(colors_precomp), _p(sh), _p(viewmatrix), _p(projmatrix), _p(campos), _p(depths), _p(radii), _p(xy),
                                   _p(transMat), _p(normal_opacity), _p(rgb), _p(clamped), _p(tiles_touched))
        self.lib.oracle_set_signature_buffers(None, None, None)
        return radii, extent

    # ------------------------------------------------------------------ backward
    def rasterize_backward(self, st, dL_dout_color, dL_dout_others):
        real = self.real
        i = st.inputs; prm = st.prm; P = prm.P
        gC = np.ascontiguousarray(np.asarray(dL_dout_color, real)); gO = np.ascontiguousarray(np.asarray(dL_dout_others, real))
        g = SimpleNamespace()
        g.dL_dtransMat = np.zeros((P, 9), real); g.dL_dmean2D = np.zeros((P, 3), real)
        g.dL_dnormal = np.zeros((P, 3), real); g.dL_dopacity = np.zeros((P, 1), real)
        g.dL_dcolors = np.zeros((P, 3), real)
        self.lib.oracle_blend_backward(C.byref(prm), _p(st.ranges), _p(st.point_list), _p(st.xy), _p(st.transMat),
                                       _p(i["transMat_precomp"]), _p(st.normal_opacity), _p(st.rgb),
                                       _p(i["colors_precomp"]), _p(i["bg"]), _p(st.final_T), _p(st.n_contrib),
                                       _p(gC), _p(gO), _p(g.dL_dtransMat), _p(g.dL_dmean2D), _p(g.dL_dnormal),
                                       _p(g.dL_dopacity), _p(g.dL_dcolors))
        g.blend_dL_dtransMat = g.dL_dtransMat.copy(); g.blend_dL_dmean2D = g.dL_dmean2D.copy()
        M = prm.M
        g.dL_dsh = np.zeros((P, M, 3), real); g.dL_dmeans3D = np.zeros((P, 3), real)
        g.dL_dscales = np.zeros((P, 2), real); g.dL_drots = np.zeros((P, 4), real)
        self.lib.oracle_preprocess_backward(C.byref(prm), _p(i["means3D"]), _p(st.radii), _p(i["sh"]), _p(st.clamped),
                                            _p(i["scales"]), _p(i["rotations"]), _p(i["transMat_precomp"]),
                                            _p(st.transMat), _p(i["viewmatrix"]), _p(i["projmatrix"]), _p(i["campos"]),
                                            _p(g.dL_dtransMat), _p(g.dL_dnormal), _p(g.dL_dcolors), _p(g.dL_dsh),
                                            _p(g.dL_dmean2D), _p(g.dL_dmeans3D), _p(g.dL_dscales), _p(g.dL_drots))
        return g

    def mark_visible(self, means3D, viewmatrix):
        means3D = _f32(means3D); out = np.zeros(means3D.shape[0], np.uint8)
        self.lib.oracle_mark_visible(means3D.shape[0], _p(means3D), _p(_f32(viewmatrix)), _p(out))
        return out.astype(bool)

    def knn_dist2(self, points):
        points = _f32(points); out = np.zeros(points.shape[0], self.real)
        self.lib.oracle_knn_dist2(points.shape[0], _p(points), _p(out))
        return out
