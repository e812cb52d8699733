"""Shared helpers for the parity tests: call the HIP path through the C ABI, call the oracle."""
import ctypes as C

import numpy as np


def scene_args(sc):
    """dict of numpy inputs in the reference's argument vocabulary from a synthetic / golden scene."""
    return dict(bg=sc["bg"], means3D=sc["means3D"], opacities=sc["opacities"], scales=sc["scales"],
                rotations=sc["rotations"], shs=sc["shs"], viewmatrix=sc["viewmatrix"], projmatrix=sc["projmatrix"],
                campos=sc["campos"], tanfovx=float(sc["tanfovx"]), tanfovy=float(sc["tanfovy"]), W=int(sc["W"]), H=int(sc["H"]),
                sh_degree=int(sc["sh_degree"]), scale_modifier=float(sc.get("scale_modifier", 1.0)))


def oracle_forward(o, a, colors_precomp=None, transMat_precomp=None, depth_key=None, use_sh=True):
    return o.rasterize_forward(a["bg"], a["means3D"], colors_precomp, a["opacities"],
                               None if transMat_precomp is not None else a["scales"],
                               None if transMat_precomp is not None else a["rotations"], a["scale_modifier"],
                               transMat_precomp, a["viewmatrix"], a["projmatrix"], a["tanfovx"], a["tanfovy"], a["H"], a["W"],
                               a["shs"] if (use_sh and colors_precomp is None) else None, a["sh_degree"], a["campos"],
                               depth_key_f32=depth_key)


class HipRun:
    """One forward (+ optional backward) through libsurfel_hip.so's C ABI with torch-owned memory."""

    def __init__(self, a, colors_precomp=None, transMat_precomp=None, debug=False, device="cuda:0"):
        import torch
        import surfel_native as n
        self.n, self.torch, self.dev = n, torch, torch.device(device)
        self.lib = n.load()
        t = lambda x: None if x is None else torch.as_tensor(np.ascontiguousarray(x, dtype=np.float32)).to(self.dev)
        self.a = a
        self.P = a["means3D"].shape[0]
        self.W, self.H, self.D = a["W"], a["H"], a["sh_degree"]
        self.bg, self.means3D, self.opac = t(a["bg"]), t(a["means3D"]), t(a["opacities"])
        self.scales = None if transMat_precomp is not None else t(a["scales"])
        self.rots = None if transMat_precomp is not None else t(a["rotations"])
        self.trans = t(transMat_precomp)
        self.colors = t(colors_precomp)
        self.shs = None if colors_precomp is not None else t(a["shs"])
        self.M = 0 if self.shs is None else self.shs.shape[1]
        self.view, self.proj, self.campos = t(a["viewmatrix"]), t(a["projmatrix"]), t(a["campos"])
        self.debug = int(debug)

    def forward(self):
        torch, n = self.torch, self.n
        self.color = torch.empty((3, self.H, self.W), device=self.dev)
        self.others = torch.empty((7, self.H, self.W), device=self.dev)
        self.radii = torch.full((self.P,), -7, dtype=torch.int32, device=self.dev)
        self.ga, self.ba, self.ia = n.TorchAllocator(self.dev), n.TorchAllocator(self.dev), n.TorchAllocator(self.dev)
        a = self.a
        R = self.lib.surfel_rasterize_forward(self.ga.cb, None, self.ba.cb, None, self.ia.cb, None, self.P, self.D, self.M,
                                              n.ptr(self.bg), self.W, self.H, n.ptr(self.means3D), n.ptr(self.shs),
                                              n.ptr(self.colors), n.ptr(self.opac), n.ptr(self.scales), a["scale_modifier"],
                                              n.ptr(self.rots), n.ptr(self.trans), n.ptr(self.view), n.ptr(self.proj),
                                              n.ptr(self.campos), a["tanfovx"], a["tanfovy"], 0, n.ptr(self.color),
                                              n.ptr(self.others), n.ptr(self.radii), self.debug, n.current_stream_ptr(self.dev))
        assert R >= 0, "forward failed: %s" % n.last_error()
        self.R = int(R)
        torch.cuda.synchronize()
        return self

    def depths(self):
        """float32 view depths the device sorted on (white-box: geom buffer = [rec P*112 B | depths ...])."""
        off = (self.P * 112 + 255) // 256 * 256
        return self.ga.last()[off:off + 4 * self.P].view(self.torch.float32).cpu().numpy()

    def backward(self, gC, gO, skip=()):
        """skip: names of optional outputs ("normal", "transMat") passed as NULL; they are left out of the returned dict."""
        torch, n = self.torch, self.n
        z = lambda *s: torch.full(s, float('nan'), device=self.dev)   # poison: the kernels must write every element
        P, M = self.P, self.M
        self.g = dict(means2D=z(P, 3), normal=z(P, 3), opacity=z(P, 1), colors=z(P, 3), means3D=z(P, 3), transMat=z(P, 9),
                      sh=z(P, max(M, 1), 3), scales=z(P, 2), rots=z(P, 4))
        gC = torch.as_tensor(np.ascontiguousarray(gC, np.float32)).to(self.dev)
        gO = torch.as_tensor(np.ascontiguousarray(gO, np.float32)).to(self.dev)
        self.sa = n.TorchAllocator(self.dev)
        g, a = self.g, self.a
        rc = self.lib.surfel_rasterize_backward(self.sa.cb, None, P, self.D, M, self.R, n.ptr(self.bg), self.W, self.H,
                                                n.ptr(self.means3D), n.ptr(self.shs), n.ptr(self.colors), n.ptr(self.scales),
                                                a["scale_modifier"], n.ptr(self.rots), n.ptr(self.trans), n.ptr(self.view),
                                                n.ptr(self.proj), n.ptr(self.campos), a["tanfovx"], a["tanfovy"],
                                                n.ptr(self.radii), n.ptr(self.ga.last()), n.ptr(self.ba.last()),
                                                n.ptr(self.ia.last()), n.ptr(gC), n.ptr(gO), n.ptr(g["means2D"]),
                                                None if "normal" in skip else n.ptr(g["normal"]), n.ptr(g["opacity"]), n.ptr(g["colors"]),
                                                n.ptr(g["means3D"]), None if "transMat" in skip else n.ptr(g["transMat"]),
                                                n.ptr(g["sh"]) if M else None, n.ptr(g["scales"]),
                                                n.ptr(g["rots"]), self.debug, n.current_stream_ptr(self.dev))
        assert rc >= 0, "backward failed: %s" % n.last_error()
        torch.cuda.synchronize()
        return {k: v.cpu().numpy() for k, v in g.items() if k not in skip}


def frac_close(x, ref, atol, rtol):
    x = np.asarray(x, np.float64); ref = np.asarray(ref, np.float64)
    ok = np.abs(x - ref) <= atol + rtol * np.abs(ref)
    return float(ok.mean())


def cosine(x, ref):
    x = np.asarray(x, np.float64).ravel(); ref = np.asarray(ref, np.float64).ravel()
    d = np.linalg.norm(x) * np.linalg.norm(ref)
    return float(x @ ref / d) if d > 0 else 1.0
