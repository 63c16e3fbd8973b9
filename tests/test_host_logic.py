"""Host logic of geograypher_amd.meshes / cameras (generators, batching, API shapes, error behaviour) driven through
the CPU oracle backend, against golden outputs of the real reference."""
import numpy as np
import pytest
import torch

from geograypher_amd.cameras import PhotogrammetryCamera, PhotogrammetryCameraSet, SegmentorPhotogrammetryCameraSet
from geograypher_amd.meshes import TexturedPhotogrammetryMesh
from geograypher_amd.predictors import ArrayLabelSegmentor, Segmentor
from geograypher_amd.utils import synthetic


def _same(a, b):
    np.testing.assert_array_equal(np.isnan(a), np.isnan(b))
    np.testing.assert_array_equal(np.nan_to_num(a, nan=-7.0), np.nan_to_num(b, nan=-7.0))


class _FixedIdsBackend:
    """Oracle backend whose rasterizer returns the golden pix2face images (the raster stage is tested elsewhere)."""

    def __new__(cls, base_cls, ids, F):
        class B(base_cls):
            def upload_mesh(self, verts, faces):
                self.n_faces = F
                self.uploads += 1

            def raster_face_ids(self, cams, h, w, out=None, want_depth=False, check=True):
                cams = np.asarray(cams).reshape(-1, 16)
                picks = [int(round(float(c[9]))) for c in cams]  # view index is encoded in the camera's x position
                return torch.from_numpy(np.stack([ids[i] for i in picks]).astype(np.int32))

        return B()


def _golden_scene(golden, oracle_backend_cls, images, segmentor=None):
    ids, F = golden["ids"], int(golden["F"])
    N, h, w = ids.shape
    cams = []
    for i in range(N):
        T = np.eye(4)
        T[0, 3] = i
        cams.append(PhotogrammetryCamera(f"/tmp/golden/{i}.png", T, f=100.0, cx=0, cy=0, image_width=w, image_height=h,
                                         local_to_epsg_4978_transform=np.eye(4)))
    cs = PhotogrammetryCameraSet(cams, local_to_epsg_4978_transform=np.eye(4))

    class ImgSet(PhotogrammetryCameraSet):
        def get_image_by_index(self, index, image_scale=1.0):
            return images[int(self.cameras[index].cam_to_world_transform[0, 3])]

    cs.__class__ = ImgSet
    mesh = TexturedPhotogrammetryMesh(
        (np.zeros((F + 3, 3)), np.zeros((F, 3), dtype=int)), texture=golden["face_texture"], log_level="ERROR",
        backend=_FixedIdsBackend(oracle_backend_cls, ids, F),
    )
    return mesh, cs


@pytest.mark.parametrize("kind", ["onehot", "rgb", "scalar"])
def test_project_and_aggregate_match_reference(golden, oracle_backend_cls, kind):
    mesh, cs = _golden_scene(golden, oracle_backend_cls, golden[kind])
    proj = list(mesh.project_images(cs))
    assert len(proj) == 4 and proj[0].dtype == np.float64
    _same(np.stack(proj), golden[f"project_{kind}"])
    _same(np.stack(list(mesh.project_images(cs, check_null_image=True))), golden[f"project_{kind}_checknull"])
    avg, info = mesh.aggregate_projected_images(cs)
    _same(avg, golden[f"agg_{kind}_average"])
    _same(info["projection_counts"], golden[f"agg_{kind}_counts"])
    _same(info["summed_projections"], golden[f"agg_{kind}_summed"])
    # batch_size 3 of 4 cameras: the trailing camera is dropped exactly like the reference does
    avg3, info3 = mesh.aggregate_viewpoints(cs, batch_size=3)
    _same(avg3, golden[f"agg_{kind}_bs3_average"])
    _same(info3["projection_counts"], golden[f"agg_{kind}_bs3_counts"])
    for v in (0, 2):
        avg1, info1 = mesh.aggregate_projected_images(cs.get_subset_cameras([v]))
        _same(avg1, golden[f"agg1_{kind}_v{v}_average"])
        _same(info1["summed_projections"], golden[f"agg1_{kind}_v{v}_summed"])
        _same(info1["projection_counts"], golden[f"agg1_{kind}_v{v}_counts"])


def test_return_all_and_render_flat_match_reference(golden, oracle_backend_cls):
    mesh, cs = _golden_scene(golden, oracle_backend_cls, golden["rgb"])
    avg, info = mesh.aggregate_projected_images(cs, return_all=True)
    _same(np.stack(info["all_projections"]), golden["agg_rgb_all_projections"])
    _same(avg, golden["agg_rgb_average"])
    renders = list(mesh.render_flat(cs, apply_distortion=False))
    _same(np.stack(renders), golden["render_flat"])
    _same(np.stack(list(mesh.render_flat(cs, batch_size=3, apply_distortion=False))), golden["render_flat_bs3"])
    img, cam = next(iter(mesh.render_flat(cs, return_camera=True, apply_distortion=False)))
    assert isinstance(cam, PhotogrammetryCamera) and img.shape == golden["render_flat"][0].shape


def test_index_label_fast_path_matches_reference(golden, oracle_backend_cls):
    """SegmentorPhotogrammetryCameraSet + class-index images -> the uint32 vote path; same numbers as the reference's
    one-hot float path."""
    labels = golden["label_inds"]
    mesh, cs = _golden_scene(golden, oracle_backend_cls, golden["onehot"])
    names = [c.image_filename for c in cs.cameras]
    seg = ArrayLabelSegmentor(labels, num_classes=golden["onehot"].shape[-1], filenames=names)
    seg_set = SegmentorPhotogrammetryCameraSet(cs, seg)
    assert seg_set.n_image_channels() == 4 and len(seg_set) == 4
    np.testing.assert_array_equal(seg_set.get_image_by_index(1), golden["onehot"][1])
    avg, info = mesh.aggregate_projected_images(seg_set)
    _same(avg, golden["agg_onehot_average"])
    _same(info["projection_counts"], golden["agg_onehot_counts"])
    _same(info["summed_projections"], golden["agg_onehot_summed"])
    # and the generic generator path through the same wrapper
    _same(np.stack(list(mesh.project_images(seg_set))), golden["project_onehot"])


def test_lookup_segmentor_png_files_through_the_threaded_loader(golden, oracle_backend_cls, tmp_path):
    """LookUpSegmentor (derived_segmentors.py:32-51): `<lookup>/<image path relative to base>.png` class-index files, read by
    the aggregation input pipeline (several files at once) -- same numbers as the in-memory labels, and the nearest-neighbour
    resize of a scaled look-up keeps the class indices."""
    from PIL import Image

    from geograypher_amd.predictors import LookUpSegmentor

    labels = golden["label_inds"]
    mesh, cs = _golden_scene(golden, oracle_backend_cls, golden["onehot"])
    base, lookup = tmp_path / "images", tmp_path / "labels"
    for i, cam in enumerate(cs.cameras):
        cam.image_filename = base / "flight_a" / f"{i}.png"
        (lookup / "flight_a").mkdir(parents=True, exist_ok=True)
        Image.fromarray(labels[i].astype(np.uint8)).save(lookup / "flight_a" / f"{i}.png")
    seg = LookUpSegmentor(base, lookup, num_classes=golden["onehot"].shape[-1])
    assert seg.thread_safe_lookup
    seg_set = SegmentorPhotogrammetryCameraSet(cs, seg)
    np.testing.assert_array_equal(seg_set.get_label_index_image(2), labels[2])
    avg, info = mesh.aggregate_projected_images(seg_set, loader_threads=3)
    _same(avg, golden["agg_onehot_average"])
    _same(info["projection_counts"], golden["agg_onehot_counts"])
    half = seg.segment_image_indices(None, filename=cs.cameras[1].image_filename, image_scale=0.5)
    h, w = labels[1].shape
    assert half.shape == (int(h * 0.5), int(w * 0.5)) and set(np.unique(half)) <= set(np.unique(labels[1]))
    # reference_float_rescale: what the reference's resize-without-preserve_range makes of a scaled look-up (index / 255 as
    # float: only 0 and 255 select a class); the one-hot image and its index form agree, scale 1 is untouched
    ref = LookUpSegmentor(base, lookup, num_classes=golden["onehot"].shape[-1], reference_float_rescale=True)
    onehot = ref.segment_image(None, filename=cs.cameras[1].image_filename, image_scale=0.5)
    np.testing.assert_array_equal(onehot[..., 0], half == 0)
    np.testing.assert_array_equal(onehot[..., 1], half == 255)
    assert not onehot[..., 2:].any()
    np.testing.assert_array_equal(ref.inds_to_one_hot(ref.segment_image_indices(None, filename=cs.cameras[1].image_filename, image_scale=0.5),
                                                      golden["onehot"].shape[-1]), onehot)
    np.testing.assert_array_equal(ref.segment_image(None, filename=cs.cameras[1].image_filename, image_scale=1),
                                  seg.segment_image(None, filename=cs.cameras[1].image_filename, image_scale=1))


def test_camera_helpers_match_reference(golden):
    for H, W, s, h, w in golden["image_sizes"]:
        cam = PhotogrammetryCamera(None, np.eye(4), 100.0, 0, 0, int(W), int(H))
        assert cam.get_image_size(float(s)) == (int(h), int(w))
    cam = PhotogrammetryCamera(
        __import__("pathlib").Path("/tmp/golden/a.png"), golden["hash_transform"], f=3705.4728792737214, cx=11.67,
        cy=-27.75, image_width=5280, image_height=3956, distortion_params={"k1": -0.09, "p1": 1e-4},
        lon_lat=(-120.4, 39.4),
    )
    assert cam.get_camera_hash() == str(golden["hash_plain"])
    assert cam.get_camera_hash(include_image_hash=True) == str(golden["hash_with_image"])
    key = PhotogrammetryCameraSet([cam]).distortion_key({"k1": -0.0919367147, "b1": 0.5262}, 0.5)
    assert key == str(golden["distortion_key"])
    np.testing.assert_array_equal(Segmentor.inds_to_one_hot(golden["label_inds"][0], 4), golden["onehot"][0])


def test_view_record_matches_pyvista_camera_model():
    """f_eff = f*h/H, principal point at the window centre, R|t straight from cam_to_world (cameras.py:446-477)."""
    T = synthetic.nadir_pose(3.0, -2.0, 50.0, yaw_deg=30.0, tilt_x_deg=4.0)
    cam = PhotogrammetryCamera(None, T, f=3705.47, cx=11.6, cy=-27.7, image_width=5280, image_height=3956)
    rec = cam.get_raster_record(0.7, near=0.5)
    h, w = int(3956 * 0.7), int(5280 * 0.7)
    assert rec.dtype == np.float32 and rec.shape == (16,)
    np.testing.assert_allclose(rec[:9].reshape(3, 3), T[:3, :3], rtol=1e-6)
    np.testing.assert_allclose(rec[9:12], T[:3, 3], rtol=1e-6)
    vp = cam.get_view_parameters()
    f_from_fov = (h / 2) / np.tan(np.deg2rad(vp["view_angle"]) / 2)
    np.testing.assert_allclose(rec[12], f_from_fov, rtol=1e-6)
    assert rec[13] == np.float32(w / 2) and rec[14] == np.float32(h / 2) and rec[15] == np.float32(0.5)
    rec_i = cam.get_raster_record(1.0, principal_point="intrinsics")
    np.testing.assert_allclose(rec_i[13:15], [5280 / 2 + 11.6, 3956 / 2 - 27.7], rtol=1e-6)
    np.testing.assert_allclose(vp["up"], T[:3, :3] @ np.array([0, -1, 0]))


def test_container_semantics():
    cs = synthetic.config1_scene()[1]
    assert len(cs) == 8 and cs.n_cameras() == 8 and cs.n_image_channels() == 3
    assert isinstance(cs[2], PhotogrammetryCamera)
    sub = cs[2:5]
    assert isinstance(sub, PhotogrammetryCameraSet) and len(sub) == 3
    pick = cs.get_subset_cameras([7, 0])
    assert [c.image_filename for c in pick.cameras] == [cs[7].image_filename, cs[0].image_filename]
    with pytest.raises(IndexError):
        cs.get_subset_cameras([8])
    with pytest.raises(ValueError):
        PhotogrammetryCameraSet(cam_to_world_transforms=[np.eye(4)] * 3, sensor_IDs=[0, 0])
    recs = cs.get_raster_records(0.5)
    assert recs.shape == (8, 16)


def test_errors_match_reference_style(oracle_backend_cls):
    (mesh, colors) = synthetic.make_simple_mesh([], None)
    tm = TexturedPhotogrammetryMesh(mesh, texture=colors, log_level="ERROR", backend=oracle_backend_cls())
    cams = synthetic.make_simple_camera_set()
    # tests/test_derived_cameras.py:318-337 -- warp requested from a base camera set
    with pytest.raises(NotImplementedError):
        tm.pix2face(cameras=cams, cache_folder=None, distortion_set=cams, apply_distortion=True)
    with pytest.raises(ValueError):
        tm.set_texture(np.zeros(7))
    with pytest.raises(IndexError):
        tm.aggregate_projected_images(cams[0:0])
    with pytest.raises(NotImplementedError):
        TexturedPhotogrammetryMesh(mesh, input_CRS="EPSG:26910", log_level="ERROR")


def test_mesh_upload_happens_once_per_frame(oracle_backend_cls):
    (mesh, colors) = synthetic.make_simple_mesh([], None)
    be = oracle_backend_cls()
    tm = TexturedPhotogrammetryMesh(mesh, texture=colors, log_level="ERROR", backend=be)
    cams = synthetic.make_simple_camera_set()
    a = tm.pix2face(cams, apply_distortion=False)
    b = tm.pix2face(cams[0], apply_distortion=False)
    assert be.uploads == 1
    assert a.shape == (1, 200, 200) and b.shape == (200, 200) and a.dtype == np.int64
    np.testing.assert_array_equal(a[0], b)
