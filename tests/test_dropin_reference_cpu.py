"""CPU, build container only: the reference's literal render() (gaussian_renderer/__init__.py:19-158) driven through THIS
package's diff_surfel_rasterization / simple_knn modules with the native call stubbed (tests/dropin/run_reference_render.py).
On the GPU box /root/reference does not exist, so the same boundary is exercised there by the package's own mirror
(surfel_render.render, tests/test_gpu_train.py::test_render_dict_matches_reference_contract)."""
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.isdir("/root/reference/gaussian_renderer"), reason="reference tree not present (GPU box)")
def test_reference_render_runs_against_the_drop_in_modules():
    r = subprocess.run([sys.executable, os.path.join(REPO, "tests", "dropin", "run_reference_render.py")], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    calls = out["calls"]
    assert len(calls) == 3
    # default path: SHs + (scales, rotations); compute_cov3D_python: the 9-float homography; override_color: precomputed colours
    assert calls[0] == dict(sh=True, colors=False, scales=True, cov=False, sh_degree=2, H=24, W=32, scale_modifier=1.0)
    assert calls[1]["cov"] and not calls[1]["scales"] and abs(calls[1]["scale_modifier"] - 0.9) < 1e-6
    assert calls[2]["colors"] and not calls[2]["sh"]
    assert out["rejected"] == 4 and out["distCUDA2_is_product"]
    assert out["keys"] == sorted(["render", "viewspace_points", "visibility_filter", "radii", "rend_alpha", "rend_normal", "rend_dist",
                                  "surf_depth", "surf_normal"])
