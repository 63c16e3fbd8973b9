#!/usr/bin/env python3
"""tools/fuzz_api.py <seconds> [first_seed] -- randomised differential campaign at the level of the reference-shaped classes: every
call is made twice, once on a mesh whose backend is the HIP library and once on a mesh whose backend is the oracle-backed
stand-in of the tests (tests/oracle_backend.py: oracle_raster.c + oracle_np.py behind the same methods), and the results are
compared -- exactly for ids, votes, counts and label paths, to 1e-12 for float sums.  GPU box only; a checker like the tests.

Per seed: a scene of tools/fuzz_parity.py (at most a few thousand faces), 2-5 cameras of a random size, a render / aggregation
scale from {1, 0.5, 0.37, 0.25, 0.13}, then
  pix2face(cameras, render_img_scale)                          (n, h, w) int64
  render_flat(cameras, render_img_scale)                       per view (h, w, C) float64, NaN background
  aggregate_projected_images(segmentor camera set, scale)      class-index label arrays at NATIVE size (nearest-resized by the
                                                               segmentor), values >= C and 255 included; 1-30 classes
  project_images(...)                                          the per-view generator of the same
  aggregate_projected_images(image set, scale)                 float64 / uint8 / bool images of the scaled size (general path)
  TexturedPhotogrammetryMeshIndexPredictions.aggregate_...     sparse (face, class) aggregation
with batch sizes drawn per seed."""
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
sys.path.insert(0, str(ROOT / "tools"))
import fuzz_parity
from geograypher_amd._hip import HipRaster
from geograypher_amd.cameras import PhotogrammetryCameraSet, SegmentorPhotogrammetryCameraSet
from geograypher_amd.meshes import TexturedPhotogrammetryMesh, TexturedPhotogrammetryMeshIndexPredictions
from geograypher_amd.predictors import ArrayLabelSegmentor
from oracle_backend import OracleBackend


class ImageSet(PhotogrammetryCameraSet):
    """In-memory images on top of a camera set (as tests/test_api_pipelines.py and bench.host_image_set build them)."""

    def __init__(self, base, images):
        self.base_camera_set, self.images, self.cameras = base, images, base.cameras
        self._local_to_epsg_4978_transform = base._local_to_epsg_4978_transform
        self._maps_ideal_to_warped, self._maps_warped_to_ideal = {}, {}
        self.image_folder = None

    def __len__(self):
        return len(self.images)

    def n_image_channels(self):
        im = np.asarray(self.images[0])
        return 1 if im.ndim == 2 else int(im.shape[-1])

    def get_subset_cameras(self, inds):
        return ImageSet(self.base_camera_set.get_subset_cameras(inds), [self.images[i] for i in inds])

    def get_image_by_index(self, i, image_scale=1.0):
        return self.images[i]


def same(a, b, rtol=0.0):
    a, b = np.asarray(a), np.asarray(b)
    if a.shape != b.shape:
        return False
    if rtol:
        return bool(np.allclose(a, b, rtol=rtol, atol=0, equal_nan=True))
    return bool(np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(np.nan_to_num(a, nan=-7.0), np.nan_to_num(b, nan=-7.0)))


def one(hip, seed):
    rng = np.random.default_rng(seed)
    for _ in range(20):
        points, faces = fuzz_parity.scene(rng)
        if 2 <= faces.shape[0] <= 6000 and faces.shape[0] != points.shape[0]:  # (V == F: set_texture cannot tell, as in the reference)
            break
    else:
        points, faces = points[: 3 * 2000], faces[:2000]
    h0, w0 = int(rng.integers(16, 260)), int(rng.integers(16, 340))
    cams = fuzz_parity.cameras(rng, points, w0, h0)
    if len(cams) < 2:
        cams = fuzz_parity.cameras(np.random.default_rng(seed + 7), points, w0, h0)
    scale = float(rng.choice([1.0, 0.5, 0.37, 0.25, 0.13]))
    h, w = cams[0].get_image_size(scale)
    if h < 1 or w < 1:
        scale, (h, w) = 1.0, (h0, w0)
    F = faces.shape[0]
    C = int(rng.choice([1, 2, 3, 4, 5, 6, 8, 9, 12, 16, 17, 30]))
    tex = rng.random((F, int(rng.integers(1, 4))))
    bs = min(int(rng.choice([1, 1, 2, 3])), len(cams))  # (the reference's batching drops trailing views: mirrored, both sides)
    info = {"seed": seed, "faces": int(F), "views": len(cams), "native": f"{w0}x{h0}", "scale": scale, "C": C, "batch": bs}
    bad = []
    orc = OracleBackend()
    m_hip = TexturedPhotogrammetryMesh((points, faces), texture=tex, log_level="ERROR", backend=hip)
    m_orc = TexturedPhotogrammetryMesh((points, faces), texture=tex, log_level="ERROR", backend=orc)
    a = m_hip.pix2face(cams, render_img_scale=scale, apply_distortion=False)
    b = m_orc.pix2face(cams, render_img_scale=scale, apply_distortion=False)
    if not (a.dtype == b.dtype and np.array_equal(a, b)):
        bad.append("pix2face")
    for v, (ra, rb) in enumerate(zip(m_hip.render_flat(cams, render_img_scale=scale, apply_distortion=False),
                                     m_orc.render_flat(cams, render_img_scale=scale, apply_distortion=False))):
        if not same(ra, rb):
            bad.append(f"render_flat view {v}")
    # label arrays at native size: the segmentor resizes them (nearest) by the aggregation scale
    labels = [rng.integers(0, C + 2, (h0, w0)).astype(np.uint8) for _ in range(len(cams))]
    for lab in labels:
        lab[rng.random(lab.shape) < 0.03] = 255
    seg = ArrayLabelSegmentor(labels, C, filenames=[c.image_filename for c in cams.cameras])

    def seg_set():
        return SegmentorPhotogrammetryCameraSet(cams, seg)

    oh = m_hip.aggregate_projected_images(seg_set(), aggregate_img_scale=scale, batch_size=bs)
    oo = m_orc.aggregate_projected_images(seg_set(), aggregate_img_scale=scale, batch_size=bs)
    if not (same(oh[0], oo[0]) and same(oh[1]["projection_counts"], oo[1]["projection_counts"]) and
            same(oh[1]["summed_projections"], oo[1]["summed_projections"])):
        bad.append("aggregate_projected_images (labels)")
    for v, (pa, pb) in enumerate(zip(m_hip.project_images(seg_set(), aggregate_img_scale=scale),
                                     m_orc.project_images(seg_set(), aggregate_img_scale=scale))):
        if not same(pa, pb):
            bad.append(f"project_images view {v}")
    # general path: images of the scaled size, one of three dtypes
    kind = rng.choice(["float64", "uint8", "bool"])
    Ci = int(rng.integers(1, 4))
    if kind == "float64":
        imgs = [rng.normal(0, 3, (h, w, Ci)) for _ in range(len(cams))]
        for im in imgs:
            im[rng.random(im.shape) < 0.1] = np.nan
    elif kind == "uint8":
        imgs = [rng.integers(0, 256, (h, w, Ci)).astype(np.uint8) for _ in range(len(cams))]
    else:
        imgs = [rng.random((h, w, Ci)) < 0.3 for _ in range(len(cams))]
    gh = m_hip.aggregate_projected_images(ImageSet(cams, imgs), aggregate_img_scale=scale, batch_size=bs)
    go = m_orc.aggregate_projected_images(ImageSet(cams, imgs), aggregate_img_scale=scale, batch_size=bs)
    if not (same(gh[0], go[0], rtol=1e-12) and same(gh[1]["projection_counts"], go[1]["projection_counts"]) and
            same(gh[1]["summed_projections"], go[1]["summed_projections"], rtol=1e-12)):
        bad.append(f"aggregate_projected_images ({kind} images)")
    # sparse index aggregation
    nc = int(rng.integers(1, 8))
    idx = [rng.integers(0, nc, (h, w)).astype(np.float64) for _ in range(len(cams))]
    for im in idx:
        im[rng.random(im.shape) < rng.choice([0.0, 0.5, 0.95])] = np.nan
    s_hip = TexturedPhotogrammetryMeshIndexPredictions((points, faces), log_level="ERROR", backend=hip)
    s_orc = TexturedPhotogrammetryMeshIndexPredictions((points, faces), log_level="ERROR", backend=orc)
    sh = s_hip.aggregate_projected_images(ImageSet(cams, idx), aggregate_img_scale=scale, n_classes=nc)
    so = s_orc.aggregate_projected_images(ImageSet(cams, idx), aggregate_img_scale=scale, n_classes=nc)
    dense = lambda x: np.asarray(x.todense()) if hasattr(x, "todense") else np.asarray(x)
    if not (same(dense(sh[0]), dense(so[0]), rtol=1e-15) and same(dense(sh[1]["projection_counts"]), dense(so[1]["projection_counts"])) and
            same(dense(sh[1]["summed_projections"]), dense(so[1]["summed_projections"]))):
        bad.append("sparse index aggregation")
    return info, bad


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 600000
    # "selftest": the stand-in on both sides -- checks this script, not the library (no GPU needed)
    hip = OracleBackend() if (len(sys.argv) > 3 and sys.argv[3] == "selftest") else HipRaster(0)
    t0 = time.time()
    n = 0
    failures = []
    while time.time() - t0 < budget:
        try:
            info, bad = one(hip, seed)
        except Exception as e:
            import traceback
            info, bad = {"seed": seed}, [f"exception: {type(e).__name__}: {e} | {traceback.format_exc().splitlines()[-3].strip()}"]
        if bad:
            failures.append({**info, "problems": bad})
            print("FAIL", json.dumps(failures[-1]), flush=True)
        n += 1
        seed += 1
    print(json.dumps({"cases": n, "failures": len(failures), "first_seed": seed - n, "seconds": round(time.time() - t0, 1)}))
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
