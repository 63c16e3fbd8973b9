"""tests/golden/make_golden_photo_resize.py -- golden vectors for the PHOTO down-scale of the aggregation path, made with the
REAL scikit-image (0.18.3, the version this container has under /opt/conda/bin/python3.9; the reference pins 0.21.0):

    /opt/conda/bin/python3.9 tests/golden/make_golden_photo_resize.py

  cameras/cameras.py:154-174 (`PhotogrammetryCamera.get_image`): `image = imread(f); if uint8: image = image / 255.0;
  if image_scale != 1: image = resize(image, (int(h * s), int(w * s)))` -- skimage.transform.resize with its defaults:
  order 1, mode "reflect", anti_aliasing on (Gaussian sigma = (n_in / n_out - 1) / 2 per axis, truncated at 4 sigma, boundary
  "mirror"), clip to the input range.  Reached from meshes.py:1988 through cameras.py:866-867 at the `aggregate_image_scale`
  the entrypoint passes (entrypoints/aggregate_images.py:184; the example scale is 0.25).

  photo_u8        (97, 131, 3) uint8 RGB "photo" (smooth gradients + noise + hard edges)
  photo_sXX       resize(photo_u8 / 255.0, ...) at s = 0.25, 0.37, 0.5                (float64, (h, w, 3))
  gray_f64        (90, 64) float64 image with values outside [0, 1] and negative ones
  gray_sXX        resize(gray_f64, ...) at s = 0.25, 0.37, 0.5, 0.9
  rgb_f32         (48, 80, 3) float32 image;  rgb32_s50: its resize at 0.5 (scikit-image keeps float32)
  up_s150         resize(gray_f64[:20, :24], x1.5): no anti-aliasing when up-scaling, samples beyond the border mirrored
  zoom_*          the same calls the way scikit-image >= 0.19 (the pinned 0.21.0) makes them: the same
                  scipy.ndimage.gaussian_filter, then scipy.ndimage.zoom(order=1, mode="mirror", grid_mode=True) instead
                  of the warp -- recorded with the scipy of this interpreter (no anti-aliasing clip differences: order 1
                  cannot leave the input range).

Output: tests/golden/reference_photo_resize.npz (inputs and outputs)."""
from pathlib import Path

import numpy as np
import scipy
import scipy.ndimage as ndi
import skimage
from skimage.transform import resize

OUT = Path(__file__).resolve().parent / "reference_photo_resize.npz"
SCALES = (("s25", 0.25), ("s37", 0.37), ("s50", 0.5), ("s90", 0.9))


def zoom_like_019(image, out_hw):
    """skimage >= 0.19 `resize` for a float image and a smaller (h, w): _warps.py (filter, then ndi.zoom with grid_mode)."""
    out_shape = tuple(out_hw) + image.shape[2:]
    factors = np.asarray(image.shape, dtype=float) / np.asarray(out_shape, dtype=float)
    sigma = np.maximum(0, (factors - 1) / 2)
    filtered = ndi.gaussian_filter(image, sigma, cval=0, mode="mirror")
    return ndi.zoom(filtered, [1 / f for f in factors], order=1, mode="mirror", cval=0, grid_mode=True)


def main():
    rng = np.random.default_rng(7)
    out = {"skimage_version": np.array(skimage.__version__), "scipy_version": np.array(scipy.__version__)}
    h, w = 97, 131
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([xx * (255.0 / w), yy * (255.0 / h), 128 + 100 * np.sin(xx / 9.0) * np.cos(yy / 5.0)], axis=-1)
    base += rng.normal(0, 20, size=base.shape)
    base[30:55, 40:75] = (250, 10, 30)      # hard edges
    base[::17, :, 1] = 0
    photo = np.clip(base, 0, 255).astype(np.uint8)
    out["photo_u8"] = photo
    img = photo / 255.0
    for tag, s in SCALES[:3]:
        shape = (int(h * s), int(w * s))
        out[f"photo_{tag}"] = resize(img, shape)
        out[f"zoom_photo_{tag}"] = zoom_like_019(img, shape)
        assert out[f"photo_{tag}"].shape == shape + (3,) and out[f"zoom_photo_{tag}"].shape == shape + (3,)
    gray = np.round(rng.normal(0.4, 1.3, size=(90, 64)), 3)
    out["gray_f64"] = gray
    for tag, s in SCALES:
        shape = (int(gray.shape[0] * s), int(gray.shape[1] * s))
        out[f"gray_{tag}"] = resize(gray, shape)
        out[f"zoom_gray_{tag}"] = zoom_like_019(gray, shape)
    rgb32 = rng.random((48, 80, 3)).astype(np.float32)
    out["rgb_f32"] = rgb32
    out["rgb32_s50"] = resize(rgb32, (24, 40))
    small = gray[:20, :24].copy()
    out["up_s150"] = resize(small, (30, 36))
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: (v.shape, str(v.dtype)) for k, v in out.items()})


if __name__ == "__main__":
    main()
