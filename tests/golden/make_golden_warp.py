"""tests/golden/make_golden_warp.py -- golden vectors for the distortion-warp stage.

Two stages, because no single interpreter of this container has every dependency of the reference:

  stage "maps"  (python3.10 + MagicMock stubs, see make_golden.py): the REAL MetashapeCameraSet.make_distortion_map
      (cameras/cameras.py:995-1062 -> derived_cameras.py:163-208, utils/indexing.py:87-150 with scipy griddata) for
      the simplified camera of tests/test_derived_cameras.py:98-113, 339-415 (f=100, k1=-0.05) on a 97 x 97 sensor.
          PYTHONPATH=/root/reference python tests/golden/make_golden_warp.py maps
  stage "warps" (/opt/conda/bin/python3.9, which has scikit-image 0.18.3; piexif stubbed): the REAL
      utils.image.flexible_inputs_warp (utils/image.py:72-126) on those maps: a face-id image (order 0, fill -1: what
      pix2face does, meshes.py:1842-1854), a float image (order 1) and a uint8 mask (order 0).
          PYTHONPATH=/root/reference /opt/conda/bin/python3.9 tests/golden/make_golden_warp.py warps

Output: tests/golden/reference_warp.npz.  The pinned scikit-image is 0.21.0 (mode "grid-constant"); 0.18.3 uses scipy's
legacy "constant" mode, which differs only for samples within half a pixel outside the input -- the tests compare
where the sample position lies inside [0, n-1].
"""
import sys
from pathlib import Path
from unittest.mock import MagicMock

import numpy as np

HERE = Path(__file__).resolve().parent
OUT = HERE / "reference_warp.npz"
SENSOR = 97


def stage_maps():
    sys.path.insert(0, str(HERE))
    import make_golden  # the stub finder

    sys.meta_path.insert(0, make_golden._Finder())
    sys.path.insert(0, make_golden.REFERENCE)
    import tempfile

    import pyproj

    pyproj.Transformer.from_crs.return_value.transform.side_effect = lambda xx, yy, zz: (0 * xx, 0 * xx, 0 * xx)
    from geograypher.cameras.derived_cameras import MetashapeCameraSet

    with tempfile.TemporaryDirectory() as tmp:
        p = Path(tmp, "camera.xml")
        p.write_text((HERE / "metashape_camera.xml").read_text())
        cams = MetashapeCameraSet(camera_file=p, image_folder=tmp)
    cam = cams.cameras[0]
    cam.cx = 0; cam.cy = 0; cam.f = 100
    cam.image_height = SENSOR; cam.image_width = SENSOR; cam.image_size = (SENSOR, SENSOR)
    for k in ["b1", "b2", "k1", "k2", "k3", "k4", "p1", "p2"]:
        cam.distortion_params[k] = 0
    cam.distortion_params["k1"] = -0.05
    out = {"sensor": SENSOR, "k1": -0.05, "f": 100.0}
    for scale, ds in ((1.0, 8), (0.5, 2)):
        cams._maps_ideal_to_warped.clear(); cams._maps_warped_to_ideal.clear()
        cams.make_distortion_map(cam, ds, scale)
        key = cams.distortion_key(cam.distortion_params, scale)
        tag = f"s{int(scale * 100)}_d{ds}"
        out[f"i2w_{tag}"] = cams._maps_ideal_to_warped[key]
        out[f"w2i_{tag}"] = cams._maps_warped_to_ideal[key]
    # a camera with the full 8-parameter model of the XML, original sensor scaled down
    with tempfile.TemporaryDirectory() as tmp:
        p = Path(tmp, "camera.xml")
        p.write_text((HERE / "metashape_camera.xml").read_text())
        cams2 = MetashapeCameraSet(camera_file=p, image_folder=tmp)
    cam2 = cams2.cameras[0]
    cams2.make_distortion_map(cam2, 64, 0.02)
    key = cams2.distortion_key(cam2.distortion_params, 0.02)
    out["full_i2w_s2"] = cams2._maps_ideal_to_warped[key]
    out["full_w2i_s2"] = cams2._maps_warped_to_ideal[key]
    np.savez_compressed(OUT, **out)
    print("maps written:", sorted(out))


def stage_warps():
    sys.modules["piexif"] = MagicMock()
    sys.path.insert(0, "/root/reference")
    from geograypher.utils.image import flexible_inputs_warp

    with np.load(OUT) as d:
        out = {k: d[k] for k in d.files}
    rng = np.random.default_rng(77)
    ids = rng.integers(-1, 80000, size=(SENSOR, SENSOR)).astype(np.int64)
    ids[:, :7] = -1
    half = int(SENSOR * 0.5)
    ids_half = rng.integers(-1, 1201250, size=(half, half)).astype(np.int64)
    fimg = rng.random((SENSOR, SENSOR, 3))
    mask = np.ones((SENSOR, SENSOR), dtype=np.uint8)
    mask[:40] = 0
    mask[:, 30:] = 2
    out.update(ids=ids, ids_half=ids_half, fimg=fimg, mask=mask)
    out["ids_warped"] = flexible_inputs_warp(ids, out["w2i_s100_d8"], interpolation_order=0, fill_value=-1)
    out["ids_dewarped"] = flexible_inputs_warp(ids, out["i2w_s100_d8"], interpolation_order=0, fill_value=-1)
    out["ids_half_warped"] = flexible_inputs_warp(ids_half, out["w2i_s50_d2"], interpolation_order=0, fill_value=-1)
    out["fimg_dewarped_o1"] = flexible_inputs_warp(fimg, out["i2w_s100_d8"], interpolation_order=1, fill_value=0.0)
    out["fimg_warped_o1"] = flexible_inputs_warp(fimg, out["w2i_s100_d8"], interpolation_order=1, fill_value=0.0)
    out["mask_warped"] = flexible_inputs_warp(mask, out["w2i_s100_d8"], interpolation_order=0, fill_value=0.0)
    const = np.full((SENSOR, SENSOR), 7, dtype=np.int64)
    out["const_warped"] = flexible_inputs_warp(const, out["w2i_s100_d8"], interpolation_order=0, fill_value=7)
    np.savez_compressed(OUT, **out)
    print("warps written:", sorted(out))


if __name__ == "__main__":
    {"maps": stage_maps, "warps": stage_warps}[sys.argv[1]]()
