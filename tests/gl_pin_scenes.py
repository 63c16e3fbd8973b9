"""Scenes shared by tests/golden/make_golden_gl.py (which renders them with real OpenGL rasterizers in the build container)
and tests/test_gl_pin.py (which compares the oracle and the HIP kernels with those renders)."""
import numpy as np

from geograypher_amd.cameras.cameras import vtk_like_clipping_range
from geograypher_amd.utils import synthetic


def bounds_of(points):
    p = np.asarray(points, dtype=np.float64)
    return np.array([p[:, 0].min(), p[:, 0].max(), p[:, 1].min(), p[:, 1].max(), p[:, 2].min(), p[:, 2].max()])


def clip_scenes():
    """(name, points, faces, camera set, h, w, near mode) of the clipping fixtures -- shared with tests/test_gl_pin.py."""
    pts = np.array([[-500, -500, 0], [500, -500, 0], [500, 500, 0], [-500, 500, 0]], dtype=np.float64)
    quad = np.array([[0, 1, 2], [0, 2, 3]])
    pose = synthetic.look_at((0.0, 0.0, 2.0), (0.0, 100.0, 2.0), up_hint=(0, 0, 1))
    yield "horizon", pts, quad, synthetic.camera_set_from_poses([pose], f=300.0, width=320, height=240), 240, 320
    (points, faces), _ = synthetic.config1_scene()
    poses = [synthetic.look_at((1.0, 2.0, 0.9), (30.0, 20.0, 0.0), up_hint=(0, 0, 1)),
             synthetic.look_at((-3.0, 0.5, 0.7), (0.0, 40.0, 5.0), up_hint=(0, 0, 1)),
             synthetic.nadir_pose(0.0, 0.0, 0.8, tilt_x_deg=70.0)]
    yield "inside_c1", points, faces, synthetic.camera_set_from_poses(poses, f=250.0, width=333, height=251), 251, 333


def vtk_ranges(cams, points):
    b = bounds_of(points)
    return [vtk_like_clipping_range(np.asarray(c.cam_to_world_transform, dtype=np.float64), b) for c in cams.cameras]
