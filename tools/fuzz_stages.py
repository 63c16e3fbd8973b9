#!/usr/bin/env python3
"""tools/fuzz_stages.py <seconds> [first_seed] -- randomised differential campaign of the numpy-stage kernels (everything the
reference does with numpy around pix2face) against the oracle (oracle/oracle_np.py, oracle/oracle_resize.py).  GPU box only; a
checker like the tests (the product never calls the oracle).

Per seed: F faces, N views of h x w ids (piecewise-constant patches + noise, -1 background, faces missing), then
  project_view        (meshes.py:1987-2001: last pixel per face wins; -1 -> last face)            exact, NaN for NaN
  project_values + finalize_sums (meshes.py:2057-2082: nansum, counts, average)                    rtol 1e-12 / exact counts
  gather_texture, gather_texture_u8 (render_flat, save_renders epilogue)                             exact
  argmax_nonzero      (utils/indexing.py:9-32)                                                       exact
  project_index_pairs (derived_meshes.py:470-520: sparse (face, class) pairs)                       exact
  resize_image        (cameras.py:154-174: /255 + scikit-image resize; uint8 / float32 / float64)    1e-12 absolute
  warp_image          (utils/image.py:72-126: nearest / bilinear through a coordinate map)           exact / 1e-12
on images with NaN / inf / negative zeros / out-of-range values where the reference's code admits them."""
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from geograypher_amd._hip import HipRaster
from oracle import oracle_np, oracle_resize, oracle_warp


DEV = "cuda"


def same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(np.nan_to_num(a, nan=-7.0), np.nan_to_num(b, nan=-7.0))


def random_ids(rng, n, h, w, F):
    ids = np.empty((n, h, w), dtype=np.int32)
    for v in range(n):
        ph, pw = int(rng.integers(1, 9)), int(rng.integers(1, 9))
        coarse = rng.integers(-1, F, size=((h + ph - 1) // ph, (w + pw - 1) // pw))
        img = np.kron(coarse, np.ones((ph, pw), dtype=np.int64))[:h, :w]
        noise = rng.random((h, w)) < rng.choice([0.0, 0.05, 0.5])
        img = np.where(noise, rng.integers(-1, F, size=(h, w)), img)
        if rng.random() < 0.3:
            img[rng.random((h, w)) < 0.5] = -1
        ids[v] = img
    return ids


def random_values(rng, shape):
    x = rng.normal(0, 1, shape) * np.exp(rng.uniform(-3, 6))
    kind = rng.integers(0, 4)
    if kind >= 1:
        x[rng.random(shape) < rng.choice([0.01, 0.3])] = np.nan
    if kind == 2:
        x[rng.random(shape) < 0.02] = np.inf
        x[rng.random(shape) < 0.02] = -np.inf
    if kind == 3:
        x[rng.random(shape) < 0.1] = -0.0
        nan_rows = rng.random(shape[:-1]) < 0.2  # all-NaN pixels: they still take the face from an earlier pixel (NaN row)
        x[nan_rows] = np.nan
    return x


def one(hip, seed):
    rng = np.random.default_rng(seed)
    bad = []
    F = int(np.exp(rng.uniform(np.log(1), np.log(5000))))
    n = int(rng.integers(1, 5))
    h, w = int(rng.integers(1, 120)), int(rng.integers(1, 160))
    C = int(rng.integers(1, 7))
    # the stage kernels take F from the context's mesh: a mesh of F degenerate faces stands in
    hip.upload_mesh(np.zeros((3, 3), dtype=np.float32), np.zeros((F, 3), dtype=np.int32))
    ids = random_ids(rng, n, h, w, F)
    img = random_values(rng, (n, h, w, C))
    compat = bool(rng.random() < 0.5)
    # project_view + project_values / finalize_sums
    projs = []
    for v in range(n):
        want = oracle_np.project_image(ids[v].astype(np.int64), img[v], F, neg1_is_last_face=compat)
        projs.append(want)
        got = hip.project_view(ids[v], img[v], neg1_is_last_face=compat).cpu().numpy()
        if not same(got, want):
            bad.append(f"project_view view {v}")
    sums = torch.zeros((F, C), dtype=torch.float64, device=DEV)
    cnt = torch.zeros((F,), dtype=torch.int32, device=DEV)
    hip.project_values(ids, img, sums, cnt, neg1_is_last_face=compat)
    avg, summed, counts = (t.cpu().numpy() for t in hip.finalize_sums(sums, cnt))
    if n > 1:
        w_avg, w_info = oracle_np.aggregate(projs, F)
        w_sum, w_cnt = w_info["summed_projections"], w_info["projection_counts"].reshape(F)
    else:
        # ONE view: the reference keeps the projection as it is (NaN channels of a seen face stay NaN, meshes.py:2060-2061) --
        # the host mirror does that itself (meshes.py here: `single_view`); the kernel's contract is the nansum of >= 2 views
        w_sum = np.where(np.isnan(projs[0]), 0.0, projs[0])
        w_cnt = np.any(np.isfinite(projs[0]), axis=1).astype(np.float64)
        w_sum[w_cnt == 0] = np.nan
        with np.errstate(divide="ignore", invalid="ignore"):
            w_avg = w_sum / w_cnt[:, None]
    with np.errstate(invalid="ignore"):
        ok = (np.allclose(avg, w_avg, rtol=1e-12, atol=0, equal_nan=True) and
              np.allclose(summed, w_sum, rtol=1e-12, atol=0, equal_nan=True) and same(counts, w_cnt))
    if not ok:
        bad.append("project_values / finalize_sums")
    # gather
    tex = random_values(rng, (F, C)) * rng.choice([1.0, 100.0])
    if not same(hip.gather_texture(ids[0], tex).cpu().numpy(), oracle_np.render_flat_gather(ids[0].astype(np.int64), tex)):
        bad.append("gather_texture")
    f64 = oracle_np.render_flat_gather(ids[0].astype(np.int64), tex)
    null = int(rng.integers(0, 256))
    with np.errstate(invalid="ignore"):
        m = (f64 < 0) | (f64 > 255) | ~np.isfinite(f64)
    f64 = f64.copy(); f64[m] = null
    if not np.array_equal(hip.gather_texture_u8(ids[0], tex, null_value=null).cpu().numpy(), f64.astype(np.uint8)):
        bad.append("gather_texture_u8")
    # argmax
    arr = rng.integers(0, 4, (F, C)).astype(np.float64) * rng.choice([1.0, 0.5])
    arr[rng.random((F, C)) < 0.05] = np.nan
    arr[rng.random(F) < 0.2] = 0.0
    if not same(hip.argmax_nonzero(arr).cpu().numpy(), np.asarray(oracle_np.find_argmax_nonzero_value(arr)).reshape(F)):
        bad.append("argmax_nonzero")
    # sparse index pairs
    nc = int(rng.integers(1, 9))
    cls = rng.integers(0, nc, (n, h, w)).astype(np.float64)
    cls[rng.random((n, h, w)) < rng.choice([0.0, 0.3, 0.9])] = np.nan
    pc = torch.zeros((F,), dtype=torch.int32, device=DEV)
    keys, mult = hip.project_index_pairs(ids, cls, nc, pc, neg1_is_last_face=compat)
    sp = [oracle_np.project_image(ids[v].astype(np.int64), cls[v][..., None], F, neg1_is_last_face=compat) for v in range(n)]
    if F > 1:  # (the reference squeezes the (F, 1) projection: a one-face mesh is an error there)
        _, w_counts, w_summed = oracle_np.aggregate_index_sparse(sp, F, nc)
        dense = np.zeros(F * nc, dtype=np.int64)
        dense[keys] = mult
        if not (np.array_equal(dense.reshape(F, nc), w_summed) and np.array_equal(pc.cpu().numpy(), w_counts[:, 0])):
            bad.append("project_index_pairs")
    # resize
    hi, wi = int(rng.integers(1, 200)), int(rng.integers(1, 260))
    ho, wo = max(1, int(hi * np.exp(rng.uniform(np.log(0.05), np.log(2.0))))), max(1, int(wi * np.exp(rng.uniform(np.log(0.05), np.log(2.0)))))
    Cr = int(rng.choice([0, 1, 3, 4]))
    shape = (hi, wi) if Cr == 0 else (hi, wi, Cr)
    dt = rng.choice(["uint8", "float32", "float64"])
    raw = rng.integers(0, 256, shape).astype(np.uint8) if dt == "uint8" else (rng.normal(0, 1, shape) * 50).astype(dt)
    src = raw.astype(np.float64) / 255.0 if dt == "uint8" else raw.astype(np.float64)
    want = oracle_resize.resize_antialias(src, (ho, wo)) if (ho, wo) != (hi, wi) else src
    got = hip.resize_image(raw, (ho, wo)).cpu().numpy()
    err = float(np.abs(got - want).max()) if got.shape == want.shape else np.inf
    tol = 1e-12 * max(1.0, float(np.abs(want).max()))
    if not err <= tol:
        bad.append(f"resize_image {dt} {shape} -> {(ho, wo)}: max |diff| {err:.3e}")
    # warp (utils/image.py:72-126 without the float round trip): coordinate maps with ties at k + 0.5, positions on and beyond
    # the border (scipy's "grid-constant": the fill value is interpolated in within one pixel outside), both orders
    hs, ws = int(rng.integers(1, 60)), int(rng.integers(1, 80))
    Hm, Wm = int(rng.integers(1, 70)), int(rng.integers(1, 90))
    rr, cc = np.meshgrid(np.linspace(-2, hs + 1, Hm), np.linspace(-2, ws + 1, Wm), indexing="ij")
    m = np.stack([rr + rng.normal(0, 1, rr.shape) * rng.choice([0.0, 0.3, 3.0]), cc + rng.normal(0, 1, cc.shape) * rng.choice([0.0, 0.3, 3.0])])
    snap = rng.random(m.shape) < 0.3
    m = np.where(snap, np.round(m * 2) / 2, m)  # integers and exact halves
    wdt = rng.choice(["int32", "int64", "uint8", "float64", "float32"])
    wshape = (hs, ws) if rng.random() < 0.6 else (hs, ws, int(rng.integers(1, 4)))
    wimg = rng.integers(-5 if "int" in wdt and wdt != "uint8" else 0, 200, wshape).astype(wdt) if "float" not in wdt else rng.normal(0, 10, wshape).astype(wdt)
    order = int(rng.integers(0, 2))
    fill = float(rng.choice([0, -1, 7])) if wdt != "uint8" else float(rng.choice([0, 7]))
    mt = hip.upload_map(m)
    gotw = hip.warp_image(wimg, mt, order=order, fill_value=fill)
    wantw = oracle_warp.warp_exact(wimg, m, order, fill)
    gotw = np.asarray(gotw)
    if gotw.shape != wantw.shape or gotw.dtype != wantw.dtype:
        bad.append(f"warp_image {wdt} order {order}: shape/dtype {gotw.shape} {gotw.dtype} vs {wantw.shape} {wantw.dtype}")
    elif order == 0 or "int" in wdt:
        # (an order-1 result is truncated to the integer type: a value within rounding of an integer may fall either way)
        diff = gotw.astype(np.float64) != wantw.astype(np.float64)
        if order == 1:
            diff = np.abs(gotw.astype(np.float64) - wantw.astype(np.float64)) > 1
        if diff.any():
            bad.append(f"warp_image {wdt} {wshape} order {order} fill {fill}: {int(diff.sum())} values differ")
    elif not np.allclose(gotw, wantw, rtol=0, atol=1e-12 * max(1.0, float(np.abs(wantw).max())) if wdt == "float64" else 1e-5):
        bad.append(f"warp_image {wdt} {wshape} order 1: max |diff| {float(np.abs(gotw.astype(np.float64) - wantw).max()):.3e}")
    return {"seed": seed, "F": F, "views": n, "image": f"{w}x{h}", "C": C, "resize": f"{dt} {wi}x{hi}->{wo}x{ho}"}, bad


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
    global DEV
    if len(sys.argv) > 3 and sys.argv[3] == "selftest":  # the checker against itself (no GPU): the oracle-backed stand-in of tests/
        sys.path.insert(0, str(ROOT / "tests"))
        from oracle_backend import OracleBackend
        hip, DEV = OracleBackend(), "cpu"
    else:
        hip = HipRaster(0)
    t0 = time.time()
    n = 0
    failures = []
    while time.time() - t0 < budget:
        try:
            info, bad = one(hip, seed)
        except Exception as e:
            info, bad = {"seed": seed}, [f"exception: {type(e).__name__}: {e}"]
        if bad:
            failures.append({**info, "problems": bad})
            print("FAIL", json.dumps(failures[-1]), flush=True)
        n += 1
        seed += 1
    print(json.dumps({"cases": n, "failures": len(failures), "first_seed": seed - n, "seconds": round(time.time() - t0, 1)}))
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
