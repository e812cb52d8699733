"""`from simple_knn._C import distCUDA2` (/root/reference/scene/gaussian_model.py:20,134)."""
import torch

import surfel_native as _n

_n.load()


def distCUDA2(points):
    """points [P,3] fp32 on a HIP device -> [P] mean squared distance to the 3 nearest neighbours."""
    if points.device.type != "cuda":
        raise RuntimeError("distCUDA2: points must live on a HIP device")
    pts = points.detach().float().contiguous()
    P = pts.shape[0]
    out = torch.empty((P,), dtype=torch.float32, device=pts.device)
    sa = _n.TorchAllocator(pts.device)
    with torch.cuda.device(pts.device):
        rc = _n.load().surfel_knn_dist2(sa.cb, None, P, _n.ptr(pts), _n.ptr(out), _n.current_stream_ptr(pts.device))
    if rc < 0:
        raise RuntimeError("surfel_knn_dist2 failed: %s" % _n.last_error())
    return out
