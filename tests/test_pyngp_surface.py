"""The `pyngp` boundary: every name the reference's module binds (tests/golden/pyngp_names.json, extracted from
src/python_api.cu:306-888 by tests/golden/make_pyngp_names.py) exists here, except the ones listed below with the reason they are out of the
NeRF hot path; plus the value types that need no GPU."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "blender-ngp_amd")]

# names of the reference module that this build does not carry, by reason
OUT_OF_SCOPE = {
    "SDF primitive (mesh BVH, sphere tracing, BRDF shading: testbed_sdf.cu)": {
        "BRDFParams", "ambientcolor", "basecolor", "clearcoat", "clearcoat_gloss", "metallic", "roughness", "sheen", "specular", "subsurface", "brdf",
        "MeshSdfMode", "Watertight", "Raystab", "PathEscape", "SDFGroundTruthMode", "RaytracedMesh", "SpheretracedMesh", "SDFBricks", "mesh_sdf_mode",
        "analytic_normals", "shadow_sharpness", "fd_normals_epsilon", "use_triangle_octree", "zero_offset", "distance_scale", "calculate_iou_online",
        "groundtruth_mode", "brick_level", "brick_res", "generate_sdf_data_online", "surface_offset_scale"},
    "GUI / window (NGP_GUI builds only)": {"keyboard_event_callback", "is_key_pressed", "is_key_down", "is_alt_down", "is_ctrl_down", "is_shift_down", "is_super_down", "screenshot"},
    "dead binding in the reference (commented out, python_api.cu:822)": {"focal_lengths"},
}


def _all_names(mod):
    seen, names = set(), set()

    def walk(o):
        if id(o) in seen:
            return
        seen.add(id(o))
        for n in dir(o):
            if n.startswith("__"):
                continue
            names.add(n)
            try:
                v = getattr(o, n)
            except Exception:
                continue
            if isinstance(v, type):
                walk(v)
    walk(mod)
    return names


def test_every_reference_name_on_the_nerf_path_is_bound():
    import torch  # noqa: F401
    import pyngp
    ref = set(json.load(open(os.path.join(ROOT, "tests", "golden", "pyngp_names.json"))))
    ours = _all_names(pyngp)
    skipped = set().union(*OUT_OF_SCOPE.values())
    assert skipped <= ref, sorted(skipped - ref)           # the exclusion list names real reference bindings only
    missing = sorted(ref - skipped - ours)
    assert not missing, missing
    assert len(ref - skipped) >= 240


def test_enums_and_value_types():
    import torch  # noqa: F401
    import pyngp
    assert pyngp.LossType.SmoothL1 == pyngp.LossType.Huber                    # legacy alias (python_api.cu:347-349)
    assert [int(pyngp.LensMode.Perspective), int(pyngp.LensMode.OpenCV), int(pyngp.LensMode.FTheta), int(pyngp.LensMode.LatLong)] == [0, 1, 2, 3]
    assert pyngp.OpenCV == pyngp.LensMode.OpenCV                              # export_values()
    b = pyngp.BoundingBox([0, 0, 0], [1, 2, 3])
    assert b.distance([2, 0, 0]) == 1.0 and b.distance_sq([2, 3, 0]) == 2.0 and b.distance([0.5, 0.5, 0.5]) == 0.0
    t = b.ray_intersect([-1, 1, 1], [1, 0, 0])
    assert t.tolist() == [1.0, 2.0]
    assert b.ray_intersect([-1, 5, 1], [1, 0, 0])[0] > 1e38                  # miss
    assert b.intersects(pyngp.BoundingBox([0.5, 0.5, 0.5], [4, 4, 4])) and not b.intersects(pyngp.BoundingBox([2, 2, 4], [4, 4, 5]))
    i = b.intersection(pyngp.BoundingBox([0.5, -1, 1], [4, 1, 2]))
    assert i.min.tolist() == [0.5, 0, 1] and i.max.tolist() == [1, 1, 2]
    assert len(b.get_vertices()) == 8 and b.get_vertices()[1].tolist() == [0, 0, 3] and b.get_vertices()[7].tolist() == [1, 2, 3]
    # signed_distance as the reference writes it (bounding_box.cuh:250-253): |p - min| - diag per axis
    assert abs(b.signed_distance([2, 0, 0]) - 1.0) < 1e-6 and b.signed_distance([0.5, 1, 1.5]) < 0
    lens = pyngp.Lens()
    lens.mode = pyngp.LensMode.FTheta
    assert lens.mode == pyngp.LensMode.FTheta and lens.params.shape == (7,)
