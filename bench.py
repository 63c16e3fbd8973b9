#!/usr/bin/env python3
"""bench.py -- face-ID rasterization throughput on BASELINE.json config 2 (1.2 M-face mesh, 4000x3000 views).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path (pix2face: setup/cull -> bin -> tile raster) over this rank's batch of 50
synthetic views, inputs (mesh, camera records) resident in HBM, ids written to HBM.  Weak scaling: every rank
rasterizes its own 50 views of the same replicated mesh, no data-path collective (pix2face has no exchange step).
The per-kernel HIP-event durations of the roofline object are collected over exactly the K timed steps (the library
records one event pair per kernel group on the stream the kernels run on; their cost is below the run-to-run noise).
The same run also times, outside the headline region, (a) the aggregation pipeline (raster + last-writer-wins
projection + per-face votes, one RCCL all-reduce of the votes at N > 1), reported under "aggregate", (b) the CPU
oracle on a bounded sample (rank 0, N == 1).
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

H, W = 3000, 4000
VIEWS_PER_RANK = 50
N_CLASSES = 4
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--views", type=int, default=VIEWS_PER_RANK, help="views per rank per step")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget for the CPU baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-aggregate", action="store_true")
    return ap.parse_args()


def _cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def torch_hash32(x):
    import torch

    m = 0xFFFFFFFF
    x = x & m
    x = x ^ (x >> 16)
    x = (x * 0x7FEB352D) & m
    x = x ^ (x >> 15)
    x = (x * 0x846CA68B) & m
    x = x ^ (x >> 16)
    return x


def device_labels(ids, view, n_classes=N_CLASSES, seed_face=4, seed_pix=5):
    """Device twin of geograypher_amd.utils.synthetic.synthetic_labels (same hash, same output)."""
    import torch

    flat = ids.reshape(-1).to(torch.int64)
    cls = torch_hash32((flat & 0xFFFFFFFF) ^ seed_face) % n_classes
    pix = torch.arange(flat.numel(), dtype=torch.int64, device=ids.device)
    r = torch_hash32(pix * 2654435761 + view * 40503 + seed_pix)
    u = r % 1000
    cls = torch.where(u < 100, (r >> 10) % n_classes, cls)
    cls = torch.where(u >= 990, torch.full_like(cls, 255), cls)
    return cls.to(torch.uint8).reshape(ids.shape)


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist

    from geograypher_amd._hip import HipRaster
    from geograypher_amd.utils import synthetic

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = "WORLD_SIZE" in os.environ and "RANK" in os.environ  # launched by torch.distributed.run
    torch.cuda.set_device(local_rank)
    if distributed:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # ---- workload: C2 mesh replicated, this rank's own 50 views (lawn-mower grid, per-rank tilt seed) -----------------
    points, faces = synthetic.terrain_mesh()
    V, F = points.shape[0], faces.shape[0]
    cams = synthetic.survey_cameras(10, 5, 40.0, 60.0, seed=3 + rank)
    nv = min(args.views, len(cams))
    recs_np = cams.get_raster_records(1.0, near=1.0)[:nv]
    hip = HipRaster(local_rank)
    hip.upload_mesh(points.astype(np.float32), faces.astype(np.int32))
    recs = torch.from_numpy(recs_np).to(dev)
    ids = torch.empty((nv, H, W), dtype=torch.int32, device=dev)
    P = H * W

    def step():
        hip.raster_face_ids(recs, H, W, out=ids, check=False)

    hip.raster_face_ids(recs, H, W, out=ids, check=True)  # sizes the bin lists once (any overflow is retried here)
    # untimed pre-conditioning (clocks, TLBs of the 12 GB scratch): the device needs ~15 ms of this workload to reach its
    # steady state after start-up; without it a short run (K <= 10) reads 5-8 % lower than a long one
    for _ in range(12):
        step()
    for _ in range(args.warmup):
        step()
    barrier()
    # the library's HIP events (one pair per kernel group, recorded on the stream the kernels run on) stay ON through the
    # timed region: the per-kernel durations of the roofline object are those of exactly the K timed steps
    hip.set_profiling(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    st = hip.stage_times()
    hip.set_profiling(False)
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    stats = hip.raster_status()
    assert stats["overflow"] == 0
    total_views = world * nv * args.steps
    views_per_s = total_views / elapsed
    mpix_per_s = views_per_s * P / 1e6

    # ---- per-kernel HIP-event times: st, collected over the timed region above ------------------------------------------
    raster_ms_per_launch = st["raster_ms"] / max(st["raster_launches"], 1)
    views_per_launch = st["views"] / max(st["raster_launches"], 1)
    # algorithmic bytes of the dominant kernel (k_raster_rows): the int32 id image it writes, 4*P per view.
    # (k_setup_cull owns the other part of B_r = 12V + 12F + 4P: the mesh read.)  DESIGN.md section "Kernels".
    raster_bytes_per_launch = 4.0 * P * views_per_launch
    achieved = raster_bytes_per_launch / (raster_ms_per_launch * 1e-3) / 1e9
    stage_ms_per_view = {k: st[k] / max(st["views"], 1) for k in ("setup_ms", "scan_ms", "fill_ms", "raster_ms")}
    pipeline_ms_per_view = sum(stage_ms_per_view.values())
    br_bytes = 12.0 * V + 12.0 * F + 4.0 * P
    traffic = None
    tfile = ROOT / "profiles" / "traffic.json"
    if tfile.is_file():
        try:
            traffic = json.loads(tfile.read_text()).get("k_raster_rows", {}).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    roofline = {
        "bound": "hbm",
        "kernel": "k_raster_rows",
        "achieved": round(achieved, 2),
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 5),
        "traffic": traffic,
        "kernel_ms_per_launch": round(raster_ms_per_launch, 4),
        "views_per_launch": round(views_per_launch, 2),
        "algorithmic_bytes_per_launch": raster_bytes_per_launch,
        "stage_ms_per_view": {k: round(v, 5) for k, v in stage_ms_per_view.items()},
        "pipeline_GBs": round(br_bytes / (pipeline_ms_per_view * 1e-3) / 1e9, 2),
        "pipeline_frac": round(br_bytes / (pipeline_ms_per_view * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
    }

    # ---- aggregation pipeline (config 3/4 shape): raster + winner + votes, one all-reduce of the votes at N > 1 --------
    aggregate = None
    if not args.no_aggregate:
        labels = torch.empty((nv, H, W), dtype=torch.uint8, device=dev)
        for v in range(nv):
            labels[v] = device_labels(ids[v], rank * nv + v)
        votes, counts = hip.new_vote_buffers(N_CLASSES)

        def agg_step():
            votes.zero_()
            counts.zero_()
            hip.raster_project_labels(recs, labels, N_CLASSES, votes, counts, ids_out=None, check=False)
            if distributed:
                from geograypher_amd.distributed import all_reduce_votes

                all_reduce_votes(votes, counts)
            return hip.finalize_votes(votes, counts)

        agg_step()
        barrier()
        t0 = time.perf_counter()
        agg_steps = max(1, args.steps // 2)
        for _ in range(agg_steps):
            avg, summed, cnt = agg_step()
        barrier()
        agg_elapsed = time.perf_counter() - t0
        if distributed:
            t = torch.tensor([agg_elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            agg_elapsed = float(t.item())
        hip.set_profiling(True)
        agg_step()
        ast = hip.stage_times()
        hip.set_profiling(False)
        agg_views = world * nv * agg_steps
        aggregate = {
            "workload": f"C3/C4-shaped: fused raster + last-writer-wins projection (ids stay in LDS) + uint32 votes, "
                        f"{N_CLASSES} classes, {nv} views/GPU, one RCCL all-reduce of [F x {N_CLASSES + 1}] int32 per step at N>1",
            "views_per_s": round(agg_views / agg_elapsed, 2),
            "mpix_per_s": round(agg_views / agg_elapsed * P / 1e6, 1),
            "faces_observed": int((cnt > 0).sum().item()),
            "raster_fused_ms_per_view": round(ast["raster_ms"] / max(ast["views"], 1), 5),
            "vote_ms_per_view": round(ast["vote_ms"] / max(ast["views"], 1), 5),
        }

    # ---- CPU baseline: the C oracle (a port of the rule-set; the reference's VTK path cannot run here) ------------------
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle_c

        cores = os.cpu_count() or 1
        t0 = time.perf_counter()
        one, _ = oracle_c.raster_views(points, faces, recs_np[:1], H, W, n_threads=1)
        t1 = time.perf_counter() - t0
        assert np.array_equal(one[0], ids[0].cpu().numpy()), "GPU ids differ from the CPU oracle on view 0"
        # bounded sample: passes over the rank's views on all host cores until ~cpu_seconds of CPU work are done
        n_done, tc, used = 0, 0.0, 1
        per_pass = int(min(max(cores, nv), 64))  # one view per thread (more threads only contend for host memory bandwidth)
        recs_pass = np.concatenate([recs_np] * (per_pass // nv + 1), axis=0)[:per_pass]
        while tc < args.cpu_seconds and n_done < 4000:
            t0 = time.perf_counter()
            _, used = oracle_c.raster_views(points, faces, recs_pass, H, W, n_threads=min(cores, per_pass))
            tc += time.perf_counter() - t0
            n_done += per_pass
        n_sample = n_done
        cpu_baseline = {
            "value": round(n_sample * P / tc / 1e6, 2),
            "unit": "Mpix/s",
            "cores": int(used),
            "kind": "port",
            "sample": f"{n_sample} C2 views at 4000x3000 ({n_sample // per_pass} passes of {per_pass}, one view per thread) on "
                      f"{used} threads of {cores} cores ({_cpu_model()}) in {tc:.1f} s; single thread: {P / t1 / 1e6:.1f} Mpix/s",
            "views_per_s": round(n_sample / tc, 3),
        }

    if rank == 0:
        line = {
            "metric": "Mpix/s rasterized (face-ID pix2face), 1.2M-face mesh @ 4000x3000",
            "value": round(mpix_per_s, 1),
            "unit": "Mpix/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int32 (fixed-point edge functions) / fp32 (vertex transform, depth)",
            "data": "synthetic",
            "config": {
                "workload": f"BASELINE config 2: {F}-face heightfield (V={V}), {nv} pinhole views/GPU @ {W}x{H}, "
                            "f=3000 px, 120 m AGL lawn-mower grid, face-ID raster to int32 in HBM",
                "views_per_gpu_per_step": nv,
                "faces": F,
                "vertices": V,
                "parallelism": f"views sharded, mesh replicated, dp{world}",
            },
            "views_per_s": round(views_per_s, 2),
            "records_per_view": round(stats["records"] / max(nv, 1), 1),
            "bin_entries_per_view": round(stats["entries"] / max(nv, 1), 1),
            "roofline": roofline,
            "cpu_baseline": cpu_baseline,
            "aggregate": aggregate,
        }
        print(json.dumps(line))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
