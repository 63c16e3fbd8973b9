#!/usr/bin/env python3
"""bench.py -- face-ID rasterization throughput on BASELINE.json config 2 (1.2 M-face mesh, 4000x3000 views).

    python bench.py [--gpus N] [--steps K] [--warmup W]

`--gpus N` with N > 1 and no WORLD_SIZE in the environment makes this process a LAUNCHER: it starts N fresh ranks
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`, one rank per GPU over
RCCL) and relays rank 0's JSON line; the launcher itself never touches the GPU.  Started by torchrun (WORLD_SIZE set) it
is a rank.

One "step" = one pass of the hot path (pix2face: cull -> set-up/bin -> tile raster) over this rank's batch of 50
synthetic views, inputs (mesh, camera records) resident in HBM, ids written to HBM.  Weak scaling: every rank rasterizes
its own 50 views of the same replicated mesh, no data-path collective (pix2face has no exchange step).  After W
warm-up steps, R windows of EXACTLY K steps each are timed, every window bracketed by barrier + synchronize on both
sides and reduced with MAX over ranks; `value` / `ms_per_step` are the median window, the spread is reported beside it.
The per-kernel HIP-event durations of the roofline object are collected over the timed windows themselves.

Outside the headline region the same run times: (a) "aggregate": BASELINE config 3 -- the 500-view grid with 4-class labels through the fused raster +
last-writer-wins projection + per-face votes (one RCCL all-reduce of the votes at N > 1), checked against the CPU oracle on one view;
(b) "c4": BASELINE config 4's per-GPU shard (250 views of the 2000-view set, view i -> GPU i mod N) with the single
all-reduce of the packed [F x (C+1)] int32 votes timed separately; (c) at N == 1, "workload_2": a hostile scene (terrain +
20 000 trees, cameras tilted 30-45 degrees) at full and at quarter resolution, with sampled oracle parity; (d) at
N == 1, the CPU oracle on a bounded sample ("cpu_baseline", one thread and all cores).
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import statistics
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

H, W = 3000, 4000
VIEWS_PER_RANK = 50
N_CLASSES = 4
C4_VIEWS_PER_RANK = 250
C3_VIEWS = 500
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--windows", type=int, default=5, help="timed windows of --steps steps each")
    ap.add_argument("--views", type=int, default=VIEWS_PER_RANK, help="views per rank per step")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget for the all-core CPU baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-aggregate", action="store_true")
    ap.add_argument("--no-c4", action="store_true")
    ap.add_argument("--no-workload2", action="store_true")
    ap.add_argument("--master-port", type=int, default=0)
    return ap.parse_args(argv)


def launch_ranks(args, argv) -> int:
    """Start `args.gpus` ranks of this script under torch.distributed.run and return its exit code.  Nothing here imports
    torch or touches HIP: the children initialise their GPUs, the launcher only waits (never exec from a GPU process)."""
    port = args.master_port
    if port == 0:
        import socket

        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve()), *argv]
    return subprocess.run(cmd, env=env).returncode


def _cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def torch_hash32(x):
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


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args, argv)
    return run(args)


def run(args) -> int:
    import torch
    import torch.distributed as dist

    from geograypher_amd._hip import HipRaster
    from geograypher_amd.distributed import all_reduce_votes
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

    def max_over_ranks(seconds: float) -> float:
        if not distributed:
            return seconds
        t = torch.tensor([seconds], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    ranks_seen = 1
    if distributed:
        one = torch.ones(1, dtype=torch.int32, device=dev)
        dist.all_reduce(one)
        ranks_seen = int(one.item())

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
    # untimed pre-conditioning (clocks, TLBs of the scratch): the device needs ~15 ms of this workload to reach its
    # steady state after start-up; without it a short run (K <= 10) reads 5-8 % lower than a long one
    for _ in range(12):
        step()
    for _ in range(args.warmup):
        step()
    # the library's HIP events (one pair per kernel group, recorded on the stream the kernels run on) stay ON through the
    # timed windows: the per-kernel durations of the roofline object are those of exactly the timed steps
    hip.set_profiling(True)
    window_s = []
    for _ in range(max(args.windows, 1)):
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        window_s.append(max_over_ranks(time.perf_counter() - t0))
    st = hip.stage_times()
    hip.set_profiling(False)
    stats = hip.raster_status()
    assert stats["overflow"] == 0
    elapsed = statistics.median(window_s)
    views_per_window = world * nv * args.steps
    views_per_s = views_per_window / elapsed
    mpix_per_s = views_per_s * P / 1e6
    ms_windows = [round(s / args.steps * 1e3, 4) for s in window_s]

    # ---- per-kernel HIP-event times: st, collected over the timed windows above -----------------------------------------
    raster_ms_per_launch = st["raster_ms"] / max(st["raster_launches"], 1)
    views_per_launch = st["views"] / max(st["raster_launches"], 1)
    # algorithmic bytes of the dominant kernel (k_raster_tile): the int32 id image it writes, 4*P per view.
    # (k_setup_cull owns the other part of B_r = 12V + 12F + 4P: the mesh read.)  DESIGN.md section "Kernels".
    raster_bytes_per_launch = 4.0 * P * views_per_launch
    achieved = raster_bytes_per_launch / (raster_ms_per_launch * 1e-3) / 1e9
    stage_ms_per_view = {k: st[k] / max(st["views"], 1) for k in ("setup_ms", "scan_ms", "fill_ms", "raster_ms")}
    pipeline_ms_per_view = sum(stage_ms_per_view.values())
    br_bytes = 12.0 * V + 12.0 * F + 4.0 * P
    traffic, traffic_source = None, None
    tfile = ROOT / "profiles" / "traffic.json"
    if tfile.is_file():
        try:
            tj = json.loads(tfile.read_text())
            entry = tj.get("k_raster_tile") or tj.get("k_raster_rows") or {}
            traffic = entry.get("hbm_bytes_per_launch")
            traffic_source = ("profiles/traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes recorded by "
                              f"tools/profile.sh ({tj.get('_source', 'committed profile')}); not measured in this run")
        except Exception:
            traffic = None
    roofline = {
        "bound": "hbm",
        "kernel": "k_raster_tile",
        "achieved": round(achieved, 2),
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 5),
        "traffic": traffic,
        "traffic_source": traffic_source,
        "kernel_ms_per_launch": round(raster_ms_per_launch, 4),
        "views_per_launch": round(views_per_launch, 2),
        "algorithmic_bytes_per_launch": raster_bytes_per_launch,
        "stage_ms_per_view": {k: round(v, 5) for k, v in stage_ms_per_view.items()},
        "pipeline_GBs": round(br_bytes / (pipeline_ms_per_view * 1e-3) / 1e9, 2),
        "pipeline_frac": round(br_bytes / (pipeline_ms_per_view * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
    }

    # ---- aggregation pipeline = BASELINE config 3: the 500-view camera grid, 4-class labels, aggregate_viewpoints on one GPU:
    #      fused raster + winners + votes in ONE call per step (8 launch groups), one all-reduce of the votes at N > 1 ----------
    aggregate = None
    labels = None
    if not args.no_aggregate:
        cams3 = synthetic.config3_cameras(C3_VIEWS)
        n3 = len(cams3)
        recs3_np = cams3.get_raster_records(1.0, near=1.0)
        recs3 = torch.from_numpy(recs3_np).to(dev)
        labels = torch.empty((n3, H, W), dtype=torch.uint8, device=dev)
        for c0 in range(0, n3, nv):  # labels are generated on the device from the ids, chunk by chunk
            c1 = min(c0 + nv, n3)
            hip.raster_face_ids(recs3[c0:c1], H, W, out=ids[: c1 - c0], check=(c0 == 0))
            for k in range(c1 - c0):
                labels[c0 + k] = device_labels(ids[k], c0 + k)
        votes, counts = hip.new_vote_buffers(N_CLASSES)

        def agg_step():
            votes.zero_()
            counts.zero_()
            hip.raster_project_labels(recs3, labels, N_CLASSES, votes, counts, ids_out=None, check=False)
            if distributed:
                all_reduce_votes(votes, counts)
            return hip.finalize_votes(votes, counts)

        hip.raster_project_labels(recs3, labels, N_CLASSES, votes, counts, check=True)  # sizing / warm-up pass
        agg_step()
        barrier()
        t0 = time.perf_counter()
        agg_steps = 5
        for _ in range(agg_steps):
            avg, summed, cnt = agg_step()
        barrier()
        agg_elapsed = max_over_ranks(time.perf_counter() - t0)
        hip.set_profiling(True)
        agg_step()
        ast = hip.stage_times()
        hip.set_profiling(False)
        agg_views = world * n3 * agg_steps
        aggregate = {
            "workload": f"BASELINE config 3: {n3} views (25 x 20 grid) of the C2 mesh per GPU, {N_CLASSES}-class labels, fused raster + "
                        f"last-writer-wins projection (ids stay in LDS) + uint32 votes in one call per step, one RCCL all-reduce of "
                        f"[F x {N_CLASSES + 1}] int32 per step at N>1",
            "views_per_s": round(agg_views / agg_elapsed, 2),
            "mpix_per_s": round(agg_views / agg_elapsed * P / 1e6, 1),
            "ms_per_step": round(agg_elapsed / agg_steps * 1e3, 3),
            "faces_observed": int((cnt > 0).sum().item()),
            "setup_ms_per_view": round(ast["setup_ms"] / max(ast["views"], 1), 5),
            "raster_fused_ms_per_view": round(ast["raster_ms"] / max(ast["views"], 1), 5),
            "vote_ms_per_view": round(ast["vote_ms"] / max(ast["views"], 1), 5),
            "oracle_check": None,
        }

    # ---- BASELINE config 4: this GPU's shard of the 2000-view set + the single all-reduce, timed separately --------------
    c4 = None
    if not args.no_c4:
        cams4 = synthetic.config4_cameras()
        mine = list(range(rank, len(cams4), world))[:C4_VIEWS_PER_RANK]
        recs4_np = cams4.get_subset_cameras(mine).get_raster_records(1.0, near=1.0)
        recs4 = torch.from_numpy(recs4_np).to(dev)
        n4 = len(mine)
        labels4 = torch.empty((n4, H, W), dtype=torch.uint8, device=dev)
        for c0 in range(0, n4, nv):  # labels are generated on the device from the ids, chunk by chunk
            c1 = min(c0 + nv, n4)
            hip.raster_face_ids(recs4[c0:c1], H, W, out=ids[: c1 - c0], check=(c0 == 0))
            for k in range(c1 - c0):
                labels4[c0 + k] = device_labels(ids[k], mine[c0 + k])
        votes4, counts4 = hip.new_vote_buffers(N_CLASSES)
        hip.raster_project_labels(recs4, labels4, N_CLASSES, votes4, counts4, check=True)  # sizing / warm-up pass
        reps = 3
        t_local, t_reduce = [], []
        for _ in range(reps):
            votes4.zero_()
            counts4.zero_()
            barrier()
            t0 = time.perf_counter()
            hip.raster_project_labels(recs4, labels4, N_CLASSES, votes4, counts4, check=False)
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            if distributed:
                all_reduce_votes(votes4, counts4)
            barrier()
            t2 = time.perf_counter()
            t_local.append(max_over_ranks(t1 - t0))
            t_reduce.append(max_over_ranks(t2 - t1))
        tl, tr = statistics.median(t_local), statistics.median(t_reduce)
        total_counts = int(counts4.to(torch.int64).sum().item())
        c4 = {
            "workload": f"BASELINE config 4: {len(cams4)}-view set (C3 grid x 4 altitudes), view i -> GPU i mod {world}, "
                        f"{n4} views on this GPU, {N_CLASSES} classes, fused aggregation + ONE all-reduce of "
                        f"[{F} x {N_CLASSES + 1}] int32 ({F * (N_CLASSES + 1) * 4 / 1e6:.1f} MB)",
            "views_per_gpu": n4,
            "aggregate_ms": round(tl * 1e3, 3),
            "all_reduce_ms": round(tr * 1e3, 3),
            "views_per_s": round(world * n4 / (tl + tr), 2),
            "face_observations_after_reduce": total_counts,
        }
        del labels4, votes4, counts4

    # ---- hostile workload (N == 1): terrain + 20 000 trees, cameras tilted 30-45 degrees; full and quarter resolution -------
    workload_2 = None
    if rank == 0 and world == 1 and not args.no_workload2:
        from oracle import oracle_c

        fpts, ffaces = synthetic.forest_scene()
        fcams = synthetic.oblique_cameras(20)
        hip2 = HipRaster(local_rank)
        hip2.upload_mesh(fpts.astype(np.float32), ffaces.astype(np.int32))
        workload_2 = {"workload": f"C2 terrain + 20 000 trees ({ffaces.shape[0]} faces, cone canopies on cylinder trunks: "
                                  "geograypher/utils/example_data.py:30-112 restated), 20 cameras tilted 30-45 degrees"}
        for scale in (1.0, 0.25):
            h2, w2 = fcams[0].get_image_size(scale)
            r2_np = fcams.get_raster_records(scale, near=1.0)
            r2 = torch.from_numpy(r2_np).to(dev)
            out2 = torch.empty((len(fcams), h2, w2), dtype=torch.int32, device=dev)
            hip2.raster_face_ids(r2, h2, w2, out=out2, check=True)
            retries = hip2.last_retries
            st2 = dict(hip2.last_stats)
            for _ in range(3):
                hip2.raster_face_ids(r2, h2, w2, out=out2, check=False)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            n_rep = 10
            for _ in range(n_rep):
                hip2.raster_face_ids(r2, h2, w2, out=out2, check=False)
            torch.cuda.synchronize(dev)
            dt = time.perf_counter() - t0
            want = oracle_c.raster(fpts, ffaces, r2_np[3], h2, w2)
            same = bool(np.array_equal(out2[3].cpu().numpy(), want))
            workload_2[f"scale_{scale:g}"] = {
                "image": f"{w2}x{h2}",
                "mpix_per_s": round(n_rep * len(fcams) * h2 * w2 / dt / 1e6, 1),
                "views_per_s": round(n_rep * len(fcams) / dt, 1),
                "entries_per_view": round(st2["entries"] / len(fcams), 1),
                "max_entries_per_tile": int(st2["max_entries"]),
                "overflow_retries_first_call": int(retries),
                "oracle_parity_view_3": same,
                "covered_fraction": round(float((want >= 0).mean()), 4),
            }
            assert same, f"workload_2 scale {scale}: GPU ids differ from the CPU oracle"
        del hip2

    # ---- CPU baseline: the C oracle (a port of the rule-set; the reference's VTK path cannot run here) ------------------
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle_c

        cores = os.cpu_count() or 1
        t0 = time.perf_counter()
        one, _ = oracle_c.raster_views(points, faces, recs_np[:1], H, W, n_threads=1)
        t1 = time.perf_counter() - t0
        assert np.array_equal(one[0], ids_first_view(hip, recs, ids)), "GPU ids differ from the CPU oracle on view 0"
        if aggregate is not None:
            # the aggregate leg against the oracle: votes of view 0 alone, fused on the GPU vs rasterized + projected on the CPU
            v1, c1 = hip.new_vote_buffers(N_CLASSES)
            hip.raster_project_labels(recs3[:1], labels[:1], N_CLASSES, v1, c1, check=True)
            want_v = np.zeros((F, N_CLASSES), dtype=np.uint32)
            want_c = np.zeros(F, dtype=np.uint32)
            one3 = oracle_c.raster(points, faces, recs3_np[0], H, W)
            oracle_c.project_labels(one3, labels[0].cpu().numpy(), F, N_CLASSES, want_v, want_c)
            ok = bool(np.array_equal(v1.cpu().numpy().view(np.uint32), want_v) and
                      np.array_equal(c1.cpu().numpy().view(np.uint32), want_c))
            aggregate["oracle_check"] = f"votes and counts of view 0 (fused call) equal the CPU oracle's: {ok}"
            assert ok, "fused aggregation differs from the CPU oracle on view 0"
        # bounded samples, one view per thread: (a) 64 threads -- where the un-culled oracle stops scaling on a 2 x 64-core host
        # (it is bound by host memory bandwidth: every thread streams the whole mesh and its own 150 MB of image buffers) --
        # and (b) ALL logical cores, as SURVEY section 8d asks; `value` is the better of the two, both are in the line
        try:
            with open("/proc/meminfo") as fh:
                avail = next(int(l.split()[1]) * 1024 for l in fh if l.startswith("MemAvailable"))
        except (OSError, StopIteration):
            avail = 16 << 30
        fit = int(max(1, (avail // 2) // (200 << 20)))

        def sample(n_thr, seconds):
            n_done, tc, used = 0, 0.0, 1
            recs_pass = np.concatenate([recs_np] * (n_thr // nv + 1), axis=0)[:n_thr]
            while tc < seconds and n_done < 4000:
                t0 = time.perf_counter()
                _, used = oracle_c.raster_views(points, faces, recs_pass, H, W, n_threads=n_thr)
                tc += time.perf_counter() - t0
                n_done += n_thr
            return {"value": round(n_done * P / tc / 1e6, 2), "unit": "Mpix/s", "cores": int(used),
                    "sample": f"{n_done} C2 views at 4000x3000 ({n_done // n_thr} passes of {n_thr}, one view per thread) on "
                              f"{used} threads of {cores} cores ({_cpu_model()}) in {tc:.1f} s",
                    "views_per_s": round(n_done / tc, 3)}

        s64 = sample(int(min(cores, 64, fit)), args.cpu_seconds / 2)
        sall = sample(int(min(cores, fit, 512)), args.cpu_seconds / 2) if cores > 64 and fit > 64 else s64
        best = s64 if s64["value"] >= sall["value"] else sall
        cpu_baseline = dict(best)
        cpu_baseline["kind"] = "port"
        cpu_baseline["all_cores"] = sall
        cpu_baseline["single_thread"] = {"value": round(P / t1 / 1e6, 2), "unit": "Mpix/s", "cores": 1,
                                         "sample": f"1 C2 view at 4000x3000 in {t1:.2f} s"}
        cpu_baseline["host_cores"] = cores

    if rank == 0:
        line = {
            "metric": "Mpix/s rasterized (face-ID pix2face), 1.2M-face mesh @ 4000x3000",
            "value": round(mpix_per_s, 1),
            "unit": "Mpix/s",
            "n_gpus": world,
            "ranks_seen": ranks_seen,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "ms_per_step_windows": {"n": len(ms_windows), "median": statistics.median(ms_windows), "min": min(ms_windows),
                                    "max": max(ms_windows), "all": ms_windows},
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
                "workload_2": None if workload_2 is None else workload_2["workload"],
            },
            "views_per_s": round(views_per_s, 2),
            "records_per_view": round(stats["records"] / max(nv, 1), 1),
            "bin_entries_per_view": round(stats["entries"] / max(nv, 1), 1),
            "roofline": roofline,
            "cpu_baseline": cpu_baseline,
            "aggregate": aggregate,
            "c4": c4,
            "workload_2": workload_2,
        }
        print(json.dumps(line), flush=True)
    if distributed:
        dist.destroy_process_group()
    return 0


def ids_first_view(hip, recs, ids):
    """View 0 of the headline workload, rasterized again (the c4 / workload legs reuse the id buffer)."""
    hip.raster_face_ids(recs[:1], H, W, out=ids[:1], check=True)
    return ids[0].cpu().numpy()


if __name__ == "__main__":
    sys.exit(main())
