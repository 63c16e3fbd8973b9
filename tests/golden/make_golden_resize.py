"""tests/golden/make_golden_resize.py -- golden vectors for the two image resizes on the projection path, made with the
REAL scikit-image (0.18.3, the version this container has under /opt/conda/bin/python3.9; the reference pins 0.21.0):

    /opt/conda/bin/python3.9 tests/golden/make_golden_resize.py

  label_*   predictors/derived_segmentors.py:44-49: `resize(index_png, (int(h*s), int(w*s)), order=0)` of a class-index
            image.  scikit-image >= 0.19 (so the pinned 0.21.0) switches anti-aliasing OFF by default for integer input
            with order 0; 0.18.3 needs it said explicitly, which is the only deviation of this recipe.  The reference
            lets resize rescale the uint8 indices to floats in [0, 1]; recorded here with preserve_range=True, because
            what is pinned is WHICH source pixel every output pixel takes.  Scales whose sample positions fall exactly
            between two source pixels (0.25, 0.5: every output pixel) are left out on purpose: scikit-image 0.18 resolves
            such ties by the rounding noise of a least-squares-estimated affine map (not reproducible even between
            its own runs on different shapes), 0.19+ by scipy.ndimage.zoom's round-half-up -- which is what the
            product's `floor((j + 0.5) * n_in / n_out)` does.
  labelf_*  the same call WITHOUT preserve_range -- what the reference really hands to `inds_to_one_hot`
            (derived_segmentors.py:44-50): float64 in [0, 1] (index / 255), so that only index 0 (-> class 0) and index 255
            (1.0 -> class 1) match any class.  `reference_float_rescale=True` of the product reproduces this bit for bit.
  zoom_*    the TIE scales 0.25 and 0.5 (the reference's own example scales): scikit-image >= 0.19 (the pinned 0.21.0) resizes
            order 0 without anti-aliasing through `scipy.ndimage.zoom(image, out/in, order=0, mode="mirror", grid_mode=True)`
            (skimage/transform/_warps.py, resize); recorded with the scipy of this interpreter.
  up0_* / up1_*   meshes.py:2312-2323: `resize(rendered, native_size, order=0 | 1)` of a float render (NaN = no face)
            from a down-scaled render to the native image size: nearest for discrete textures, bilinear otherwise
            (default mode "reflect").

Output: tests/golden/reference_resize.npz (inputs and outputs)."""
from pathlib import Path

import numpy as np
import scipy
import scipy.ndimage as ndi
import skimage
from skimage.transform import resize

OUT = Path(__file__).resolve().parent / "reference_resize.npz"


def main():
    rng = np.random.default_rng(42)
    out = {"skimage_version": np.array(skimage.__version__)}
    label = rng.integers(0, 7, size=(60, 83)).astype(np.uint8)
    label[rng.random(label.shape) < 0.05] = 255
    out["label_in"] = label
    for tag, s in (("s30", 0.3), ("s37", 0.37), ("s45", 0.45), ("s90", 0.9)):
        shape = (int(label.shape[0] * s), int(label.shape[1] * s))
        out[f"label_{tag}"] = resize(label, shape, order=0, anti_aliasing=False, preserve_range=True).astype(np.uint8)
        out[f"labelf_{tag}"] = resize(label, shape, order=0, anti_aliasing=False)
    out["scipy_version"] = np.array(scipy.__version__)
    for tag, s in (("s25", 0.25), ("s50", 0.5)):
        shape = (int(label.shape[0] * s), int(label.shape[1] * s))
        out[f"zoom_{tag}"] = ndi.zoom(label, (shape[0] / label.shape[0], shape[1] / label.shape[1]), order=0, mode="mirror",
                                      grid_mode=True)
        assert out[f"zoom_{tag}"].shape == shape
    small = rng.random((23, 31))
    small[rng.random(small.shape) < 0.1] = np.nan
    ids_like = rng.integers(0, 5, size=(23, 31)).astype(np.float64)
    out["up_in"], out["up_ids_in"] = small, ids_like
    for tag, native in (("a", (92, 124)), ("b", (57, 80))):
        out[f"up0_{tag}"] = resize(ids_like, native, order=0)
        out[f"up1_{tag}"] = resize(small, native, order=1)
        out[f"up1ids_{tag}"] = resize(ids_like, native, order=1)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
