"""tests/golden/make_golden.py -- generates tests/golden/reference_numpy_stages.npz (+ reference_cameras.npz).

Runs the REAL reference (geograypher at /root/reference, v0.4.0) in THIS container to produce golden input/output
vectors for the numpy stages of the projection path.  The reference cannot be imported as shipped here (pyvista, vtk,
geopandas, pyproj, shapely, skimage, ... are absent), so -- exactly as SURVEY.md Appendix B describes -- a meta-path
finder serves MagicMock packages for the absent third-party roots; the reference's own numpy code then runs
unmodified.  Nothing of the reference travels: only the .npz data files written here are committed.

    PYTHONPATH=/root/reference PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Captured (function -> reference lines executed):
    TexturedPhotogrammetryMeshIndexPredictions.aggregate_projected_images   derived_meshes.py:414-550 (sparse)
    TexturedPhotogrammetryMesh.project_images              meshes.py:1970-2002
    TexturedPhotogrammetryMesh.aggregate_projected_images  meshes.py:2033-2084
    TexturedPhotogrammetryMesh.render_flat                 meshes.py:1891-1942
    Segmentor.inds_to_one_hot                              predictors/segmentor.py:37-69
    find_argmax_nonzero_value                              utils/indexing.py:9-32
    PhotogrammetryCamera.get_image_size / get_camera_hash  cameras/cameras.py:179-200, 104-134
    PhotogrammetryCameraSet.distortion_key                 cameras/cameras.py:968-993
    MetashapeCameraSet.__init__ / ideal_to_warped          cameras/derived_cameras.py:52-208 (inputs of the "next" row f1)
"""
import importlib.abc
import importlib.machinery
import sys
import tempfile
from pathlib import Path
from unittest.mock import MagicMock

import numpy as np

REFERENCE = "/root/reference"
MISSING = ("fiona", "geopandas", "pyproj", "pyvista", "rasterio", "shapely", "skimage", "ubelt", "imageio", "piexif",
           "trimesh", "rtree", "rasterstats", "setcoverpy", "chardet", "cchardet", "IPython")


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, name, path, target=None):
        if name.split(".")[0] in MISSING:
            return importlib.machinery.ModuleSpec(name, self, is_package=True)

    def create_module(self, spec):
        m = MagicMock()
        m.__path__ = []
        m.__spec__ = spec
        m.__name__ = spec.name
        return m

    def exec_module(self, module):
        pass


def main():
    sys.dont_write_bytecode = True
    sys.path.insert(0, REFERENCE)
    sys.meta_path.insert(0, _Finder())
    from geograypher.cameras.cameras import PhotogrammetryCamera, PhotogrammetryCameraSet
    from geograypher.meshes.meshes import TexturedPhotogrammetryMesh as TPM
    from geograypher.predictors.segmentor import Segmentor
    from geograypher.utils.indexing import find_argmax_nonzero_value

    rng = np.random.default_rng(20260101)
    out = {}

    # ---- scene: F faces, N views of h x w; ids contain background (-1), repeated faces, never-seen faces ---------------
    F, N, h, w, C = 50, 4, 12, 16, 4
    ids = rng.integers(-1, F - 8, size=(N, h, w)).astype(np.int64)  # faces F-8 .. F-1 are never rendered
    ids[0, :3] = -1
    ids[1, -1, -1] = 7  # last pixel is a real face in view 1 (so -1 aliasing comes from an earlier pixel)
    label_inds = rng.integers(0, C, size=(N, h, w)).astype(np.uint8)
    label_inds[rng.random((N, h, w)) < 0.05] = 255  # ignore label -> all-False one-hot row
    onehot = np.stack([Segmentor.inds_to_one_hot(label_inds[v], num_classes=C) for v in range(N)], axis=0)
    rgb = rng.random((N, h, w, 3))
    rgb[2, 4:6, :, 1] = np.nan  # NaN in one channel
    scalar = rng.random((N, h, w)) * 10.0
    scalar[rng.random((N, h, w)) < 0.3] = np.nan
    scalar[3] = np.nan  # a null image
    face_texture = rng.random((F, 2)) * 100.0

    cams = [
        PhotogrammetryCamera(Path(f"/tmp/golden/{i}.png"), np.eye(4), f=100.0, cx=0.0, cy=0.0, image_width=w,
                             image_height=h, local_to_epsg_4978_transform=np.eye(4))
        for i in range(N)
    ]
    index_of = {id(c): i for i, c in enumerate(cams)}

    class FakeSet(PhotogrammetryCameraSet):
        def __init__(self, cameras, images):
            super().__init__(cameras, local_to_epsg_4978_transform=np.eye(4))
            self.images = images

        def get_subset_cameras(self, inds):
            return FakeSet([self.cameras[i] for i in inds], self.images)

        def get_image_by_index(self, index, image_scale=1.0):
            return self.images[index_of[id(self.cameras[index])]]

    class FakeSelf:
        faces = np.zeros((F, 3), dtype=int)
        logger = MagicMock()

        def get_mesh_in_cameras_coords(self, cameras):
            return None

        def get_texture(self, request_vertex_texture=False, try_verts_faces_conversion=True):
            return face_texture

        def pix2face(self, cameras, mesh=None, render_img_scale=1, **kw):
            if isinstance(cameras, PhotogrammetryCamera):
                return ids[index_of[id(cameras)]]
            return np.stack([ids[index_of[id(c)]] for c in cameras.cameras], axis=0)

        def project_images(self, **kw):
            return TPM.project_images(self, **kw)

    fs = FakeSelf()
    out.update(F=F, ids=ids, label_inds=label_inds, onehot=onehot, rgb=rgb, scalar=scalar, face_texture=face_texture)

    for name, images in (("onehot", onehot), ("rgb", rgb), ("scalar", scalar)):
        cs = FakeSet(cams, images)
        proj = list(TPM.project_images(fs, cameras=cs))
        out[f"project_{name}"] = np.stack(proj, axis=0)
        proj_null = list(TPM.project_images(fs, cameras=cs, check_null_image=True))
        out[f"project_{name}_checknull"] = np.stack(proj_null, axis=0)
        avg, info = TPM.aggregate_projected_images(fs, cs)
        out[f"agg_{name}_average"] = avg
        out[f"agg_{name}_counts"] = info["projection_counts"]
        out[f"agg_{name}_summed"] = info["summed_projections"]
        # single-view aggregation keeps the first projection un-nansummed (meshes.py:2057-2058)
        for v in (0, 2):
            cs1 = FakeSet([cams[v]], images)
            avg1, info1 = TPM.aggregate_projected_images(fs, cs1)
            out[f"agg1_{name}_v{v}_average"] = avg1
            out[f"agg1_{name}_v{v}_counts"] = info1["projection_counts"]
            out[f"agg1_{name}_v{v}_summed"] = info1["summed_projections"]
        # batch_size 3 over 4 cameras: the reference silently drops the trailing camera (meshes.py:1976-1977)
        avg3, info3 = TPM.aggregate_projected_images(fs, cs, batch_size=3)
        out[f"agg_{name}_bs3_average"] = avg3
        out[f"agg_{name}_bs3_counts"] = info3["projection_counts"]
    avg_all, info_all = TPM.aggregate_projected_images(fs, FakeSet(cams, rgb), return_all=True)
    out["agg_rgb_all_projections"] = np.stack(info_all["all_projections"], axis=0)

    # render_flat (gather)
    cs = FakeSet(cams, rgb)
    out["render_flat"] = np.stack(list(TPM.render_flat(fs, cs)), axis=0)
    out["render_flat_bs3"] = np.stack(list(TPM.render_flat(fs, cs, batch_size=3)), axis=0)

    # argmax
    crafted = np.array([[0, 0, 0, 0], [1, 3, 3, 0], [np.nan, 1, 0, 0], [0.5, 0.25, 0.25, 0], [0, 0, 0, 2.0],
                        [np.inf, 1, 0, 0], [-1, 1, 0, 0]])
    out["argmax_in"] = np.concatenate([out["agg_onehot_average"], crafted], axis=0)
    out["argmax_out"] = find_argmax_nonzero_value(out["argmax_in"], keepdims=True)
    out["argmax_out_flat"] = find_argmax_nonzero_value(out["argmax_in"])

    # camera helpers
    sizes = []
    for (H, W) in ((3956, 5280), (3000, 4000), (257, 257), (480, 640)):
        cam = PhotogrammetryCamera(None, np.eye(4), 100.0, 0, 0, W, H)
        for s in (1.0, 0.9, 0.7, 0.5, 0.25, 0.1234):
            sizes.append((H, W, s) + tuple(cam.get_image_size(s)))
    out["image_sizes"] = np.array(sizes, dtype=np.float64)
    T = np.array([[0.6, -0.8, 0.0, 1.5], [0.8, 0.6, 0.0, -2.25], [0.0, 0.0, 1.0, 40.0], [0, 0, 0, 1.0]])
    cam = PhotogrammetryCamera(Path("/tmp/golden/a.png"), T, f=3705.4728792737214, cx=11.67, cy=-27.75,
                               image_width=5280, image_height=3956, distortion_params={"k1": -0.09, "p1": 1e-4},
                               lon_lat=(-120.4, 39.4))
    out["hash_transform"] = T
    out["hash_plain"] = np.array(cam.get_camera_hash())
    out["hash_with_image"] = np.array(cam.get_camera_hash(include_image_hash=True))
    out["distortion_key"] = np.array(PhotogrammetryCameraSet([cam]).distortion_key({"k1": -0.0919367147, "b1": 0.5262}, 0.5))

    # ---- sparse index aggregation (derived_meshes.py:414-550), real reference, scipy from this interpreter ---------------
    from geograypher.meshes.derived_meshes import TexturedPhotogrammetryMeshIndexPredictions as TPMI

    n_classes = 37
    index_imgs = rng.integers(0, n_classes, size=(N, h, w)).astype(float)
    index_imgs[rng.random((N, h, w)) < 0.4] = np.nan
    index_imgs[2] = np.nan  # a null image (skipped by check_null_image)
    cs = FakeSet(cams, index_imgs)
    avg_sp, info_sp = TPMI.aggregate_projected_images(fs, cs, n_classes=n_classes)
    out["index_imgs"] = index_imgs
    out["index_n_classes"] = n_classes
    out["index_average"] = np.asarray(avg_sp.todense())
    out["index_counts"] = np.asarray(info_sp["projection_counts"].todense())
    out["index_summed"] = np.asarray(info_sp["summed_projections"].todense())

    np.savez_compressed(Path(__file__).with_name("reference_numpy_stages.npz"), **out)
    print("wrote reference_numpy_stages.npz with", len(out), "arrays")

    # ---- Metashape camera parsing + distortion model (inputs for row f1) ---------------------------------------------
    import pyproj  # the MagicMock

    pyproj.Transformer.from_crs.return_value.transform.side_effect = lambda xx, yy, zz: (0 * xx, 0 * xx, 0 * xx)
    from geograypher.cameras.derived_cameras import MetashapeCameraSet

    xml = (Path(__file__).with_name("metashape_camera.xml")).read_text()
    with tempfile.TemporaryDirectory() as tmp:
        p = Path(tmp, "camera.xml")
        p.write_text(xml)
        ms = MetashapeCameraSet(camera_file=p, image_folder=tmp)
    c0 = ms.cameras[0]
    cam_out = {
        "f": c0.f, "cx": c0.cx, "cy": c0.cy, "image_width": c0.image_width, "image_height": c0.image_height,
        "cam_to_world": np.asarray(c0.cam_to_world_transform, dtype=np.float64),
        "local_to_epsg_4978": np.asarray(ms.get_local_to_epsg_4978_transform(), dtype=np.float64),
        "distortion_keys": np.array(sorted(c0.distortion_params.keys())),
        "distortion_values": np.array([c0.distortion_params[k] for k in sorted(c0.distortion_params.keys())]),
    }
    xs, ys = np.meshgrid(np.linspace(0, c0.image_width - 1, 23), np.linspace(0, c0.image_height - 1, 17))
    wx, wy = ms.ideal_to_warped(c0, xs.copy(), ys.copy())
    cam_out.update(warp_in_x=xs, warp_in_y=ys, warp_out_x=np.asarray(wx), warp_out_y=np.asarray(wy))
    np.savez_compressed(Path(__file__).with_name("reference_cameras.npz"), **cam_out)
    print("wrote reference_cameras.npz")


if __name__ == "__main__":
    main()
