#!/usr/bin/env python3
"""bench.py -- face-ID rasterization throughput on BASELINE.json config 2 (1.2 M-face mesh, 4000x3000 views).

    python bench.py [--gpus N] [--steps K] [--warmup W]

`--gpus N` with N > 1 and no WORLD_SIZE in the environment makes this process a LAUNCHER: it starts N fresh ranks
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`, one rank per GPU over
RCCL) and relays rank 0's JSON line; the launcher itself never touches the GPU.  Started by torchrun (WORLD_SIZE set) it
is a rank.  A line that claims N GPUs is only printed when N ranks answered (exit code 3 otherwise).

One "step" = one pass of the hot path (pix2face: cull -> set-up/bin -> tile raster) over this rank's batch of 50
synthetic views, inputs (mesh, camera records) resident in HBM, ids written to HBM.  Weak scaling: every rank rasterizes
its own 50 views of the same replicated mesh, no data-path collective (pix2face has no exchange step).  After W
warm-up steps, windows of EXACTLY K steps each are timed -- at least --windows of them, and as many more as it takes to
put --min-timed-s (1 s) of GPU time inside timed regions --, every window bracketed by barrier + synchronize on both
sides and reduced with MAX over ranks; `value` / `ms_per_step` are the median window, the spread is reported beside it.
The per-kernel HIP-event durations of the roofline objects are collected over the timed windows themselves.

Outside the headline region the same run times
  (a) "aggregate": BASELINE config 3 -- the 500-view grid with 4-class labels through the fused raster + last-writer-wins
      projection + per-face votes (one RCCL all-reduce of the votes at N > 1), checked against the CPU oracle on one view;
  (b) "c4": BASELINE config 4's per-GPU shard (250 views of the 2000-view set, view i -> GPU i mod N) with the single
      all-reduce of the packed [F x (C+1)] int32 votes timed separately, its bytes and its algorithmic xGMI time;
  (c) "c5": BASELINE config 5's per-GPU shard (5 M faces, 6000x4000, 10 classes, view i -> GPU i mod N), ids-only and
      fused aggregation, one view checked against the oracle (N == 1);
  (d) at N == 1, "workload_2": a hostile scene (terrain + 20 000 trees, cameras tilted 30-45 degrees) at full and at quarter
      resolution, with sampled oracle parity, and "api": the PCIe-inclusive rates of the reference-shaped numpy API;
  (e) at N == 1, the CPU oracle on a bounded sample ("cpu_baseline", one thread and all cores).
Rank 0 prints ONE JSON line.

`run(args, rig)` takes the platform glue as an object: the default `GpuRig` is the MI355X box (RCCL, HipRaster, the
BASELINE sizes); tests/test_bench_distributed.py drives the same control flow -- every collective, every MAX reduction,
the rank-0-only line -- at world size 2 on gloo with a CPU stand-in and a toy workload.
"""
import argparse
import json
import os
import statistics
import subprocess
import sys
import time
from dataclasses import dataclass
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
N_SIMD = 1024           # 256 CUs x 4 SIMDs
SHADER_GHZ = 2.4        # nominal shader clock the VALU-issue fraction is priced at
XGMI_LINK_GBS = 153.0   # one xGMI link, per direction; 7 links per GPU


@dataclass
class Workload:
    """Sizes of every leg.  The defaults are the BASELINE.json configs; the distributed control-flow test shrinks them."""
    n_side: int = 776            # C2 / C3 / C4 heightfield: 776 x 776 vertices -> 1 201 250 faces
    extent: float = 400.0
    H: int = 3000
    W: int = 4000
    f: float = 3000.0
    views_per_rank: int = 50     # C2
    c3_views: int = 500
    c4_views_total: int = 2000   # C4: the 2000-view set, view i -> GPU i mod N
    c4_views_per_rank: int = 250
    n_classes: int = 4
    c5_n_side: int = 1582        # C5: 4 999 122 faces over 800 m, 6000 x 4000, f = 4500 px, 150 m AGL, 10 classes
    c5_extent: float = 800.0
    c5_H: int = 4000
    c5_W: int = 6000
    c5_f: float = 4500.0
    c5_views_total: int = 2000
    c5_views_per_rank: int = 250
    c5_classes: int = 10
    c5_raster_views: int = 20    # ids-only sample of the shard (96 MB of ids per view)
    min_leg_s: float = 0.3       # every side leg is timed until this much GPU time lies inside its timed regions

    def cam_kw(self):
        return dict(f=self.f, width=self.W, height=self.H)


class GpuRig:
    """Everything that ties bench.run to the MI355X box."""
    dist_backend = "nccl"

    def __init__(self):
        self.workload = Workload()

    def device(self, local_rank):
        import torch

        torch.cuda.set_device(local_rank)
        return torch.device("cuda", local_rank)

    def init_process_group(self, dev):
        import torch.distributed as dist

        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(self.dist_backend, device_id=dev)

    def synchronize(self, dev):
        import torch

        torch.cuda.synchronize(dev)

    def describe(self, rank, local_rank, dev):
        import torch

        p = torch.cuda.get_device_properties(dev)
        return {"rank": rank, "local_rank": local_rank, "device": str(dev), "name": p.name, "cus": p.multi_processor_count,
                "hbm_GiB": round(p.total_memory / 2**30, 1), "visible": os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES")}

    def backend_version(self):
        import torch

        try:
            v = torch.cuda.nccl.version()
            return "rccl " + ".".join(str(x) for x in v) + " (torch.distributed backend 'nccl')"
        except Exception as exc:  # the line still goes out
            return f"nccl (version unavailable: {exc!r})"

    def make_raster(self, local_rank):
        from geograypher_amd._hip import HipRaster

        return HipRaster(local_rank)

    def checker(self):
        """The CPU oracle: the checker of the oracle-checked legs and the thing timed as `cpu_baseline` -- never on the
        measured GPU path."""
        from oracle import oracle_c

        return oracle_c

    def side_legs(self):  # workload_2 / api need the real library and the real sizes
        return True


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--windows", type=int, default=5, help="least number of timed windows of --steps steps each")
    ap.add_argument("--min-timed-s", type=float, default=1.0,
                    help="keep adding timed windows until this much GPU time lies inside timed regions")
    ap.add_argument("--views", type=int, default=0, help="views per rank per step (default: the workload's 50)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget for the all-core CPU baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-aggregate", action="store_true")
    ap.add_argument("--no-c4", action="store_true")
    ap.add_argument("--no-c5", action="store_true")
    ap.add_argument("--no-workload2", action="store_true")
    ap.add_argument("--no-api", action="store_true")
    ap.add_argument("--no-io", action="store_true")
    ap.add_argument("--master-port", type=int, default=0)
    ap.add_argument("--variant", type=int, default=0,
                    help="GR_OPT_VARIANT bits for every context (results identical); tools/profile.sh passes 4 -- fused votes on "
                         "the caller's stream -- in its counter passes, where rocprofv3 does not survive the side stream")
    return ap.parse_args(argv)


def launch_ranks(args, argv) -> int:
    """Start `args.gpus` ranks of this script under torch.distributed.run and return its exit code.  Nothing here imports
    torch or touches HIP: the children initialise their GPUs, the launcher only waits (never exec from a GPU process)."""
    # are there N devices at all?  Asked in a child (counting devices does not initialise the GPU, but the launcher stays
    # free of torch and HIP anyway): a line for N GPUs must not be attempted on a smaller node
    probe = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True)
    try:
        n_dev = int(probe.stdout.strip().splitlines()[-1])
    except (ValueError, IndexError):
        n_dev = -1
    if n_dev < args.gpus:
        print(f"bench.py: --gpus {args.gpus} but this node shows {n_dev} device(s) (torch.cuda.device_count() in a child process)"
              f"{': ' + probe.stderr.strip()[-300:] if n_dev < 0 else ''}", file=sys.stderr, flush=True)
        return 3
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


def _cpu_quota_cores():
    """CPU time the container may use, in cores (cgroup v2 cpu.max / v1 cfs quota), or None when unlimited / unknown: the
    hosts of the GPU pool show 256 logical cores but zlib and the oracle stop scaling at about 16 threads."""
    try:
        txt = open("/sys/fs/cgroup/cpu.max").read().split()
        if txt[0] != "max":
            return round(int(txt[0]) / int(txt[1]), 2)
        return None
    except (OSError, ValueError, IndexError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return round(q / p, 2) if q > 0 else None
    except (OSError, ValueError):
        return None


def torch_hash32(x):
    m = 0xFFFFFFFF
    x = x & m
    x = x ^ (x >> 16)
    x = (x * 0x7FEB352D) & m
    x = x ^ (x >> 15)
    x = (x * 0x846CA68B) & m
    x = x ^ (x >> 16)
    return x


def device_labels(ids, view, n_classes=4, seed_face=4, seed_pix=5):
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


def kernel_source_sha256() -> str:
    """sha256 over the library's device + host sources (geograypher_amd/csrc/*.hip, *.hpp, include/geograster.h), in name
    order: tools/summarize_profile.py stamps profiles/traffic.json and valu.json with it, and a bench line quotes their
    counter figures only while the tree still holds the kernels they were measured on."""
    import hashlib

    h = hashlib.sha256()
    files = sorted((ROOT / "geograypher_amd" / "csrc").glob("*.hip")) + sorted((ROOT / "geograypher_amd" / "csrc").glob("*.hpp"))
    for path in files + [ROOT / "include" / "geograster.h"]:
        h.update(path.name.encode())
        h.update(path.read_bytes())
    return h.hexdigest()


_PROFILE_STALE = {}


def _profile_json(name):
    """profiles/<name> -- or None when it is missing, unreadable or STALE (its `_kernel_sha256` is not the tree's)."""
    path = ROOT / "profiles" / name
    if path.is_file():
        try:
            data = json.loads(path.read_text())
        except Exception:
            return None
        stale = data.get("_kernel_sha256") != kernel_source_sha256()
        _PROFILE_STALE[name] = stale
        return None if stale else data
    return None


def valu_bound(kernel_key, kernel_ms_per_launch, views_per_launch):
    """Second roofline of a VALU-bound kernel: wave-level VALU instructions (SQ_INSTS_VALU of the committed PMC pass,
    profiles/valu.json, per view) x 4 issue cycles / (SIMDs x kernel duration x shader clock)."""
    vj = _profile_json("valu.json") or {}
    entry = vj.get(kernel_key)
    if not entry or not kernel_ms_per_launch:
        return None
    insts = entry["valu_insts_per_view"] * views_per_launch
    frac = insts * 4.0 / (N_SIMD * kernel_ms_per_launch * 1e-3 * SHADER_GHZ * 1e9)
    return {"bound": "valu", "achieved": round(insts * 4.0 / (kernel_ms_per_launch * 1e-3) / 1e9, 1), "peak": N_SIMD * SHADER_GHZ,
            "unit": "G SIMD-cycles/s", "frac": round(frac, 4), "valu_insts_per_view": entry["valu_insts_per_view"],
            "source": f"profiles/valu.json ({vj.get('_source', 'committed PMC pass')}: SQ_INSTS_VALU per launch / views per "
                      f"launch, not measured in this run), 4 issue cycles per wave64 instruction, {N_SIMD} SIMDs at {SHADER_GHZ} GHz"}


def hbm_roofline(kernel, bytes_per_launch, kernel_ms_per_launch, views_per_launch, traffic_key=None, note=None):
    achieved = bytes_per_launch / (kernel_ms_per_launch * 1e-3) / 1e9 if kernel_ms_per_launch else 0.0
    traffic, source = None, None
    tj = _profile_json("traffic.json")
    if tj and traffic_key:
        entry = tj.get(traffic_key) or {}
        if entry.get("hbm_bytes_per_view") is not None:
            traffic = entry["hbm_bytes_per_view"] * views_per_launch
        elif entry.get("hbm_bytes_per_launch") is not None and traffic_key == "k_raster_tile":
            traffic = entry["hbm_bytes_per_launch"]
        if traffic is not None:
            source = ("profiles/traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes recorded by tools/profile.sh "
                      f"({tj.get('_source', 'committed profile')}); not measured in this run")
    out = {"bound": "hbm", "kernel": kernel, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": source,
           "traffic_stale": bool(_PROFILE_STALE.get("traffic.json", False)),
           "kernel_ms_per_launch": round(kernel_ms_per_launch, 4), "views_per_launch": round(views_per_launch, 2),
           "algorithmic_bytes_per_launch": bytes_per_launch}
    if note:
        out["algorithmic_bytes"] = note
    return out


def run(args, rig=None) -> int:
    import torch
    import torch.distributed as dist

    from geograypher_amd.distributed import all_reduce_votes
    from geograypher_amd.utils import synthetic

    rig = rig or GpuRig()
    if args.variant:
        make_plain = rig.make_raster

        def make_raster(local_rank):
            r = make_plain(local_rank)
            r.set_option(7, args.variant)
            return r

        rig.make_raster = make_raster
    wl = rig.workload
    H, W, N_CLASSES = wl.H, wl.W, wl.n_classes
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = "WORLD_SIZE" in os.environ and "RANK" in os.environ  # launched by torch.distributed.run
    dev = rig.device(local_rank)
    if distributed:
        rig.init_process_group(dev)

    def barrier():
        if distributed:
            dist.barrier()
        rig.synchronize(dev)

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
    # who runs where: rank -> device, gathered on every rank (a collective: all ranks take part), printed by rank 0
    me = rig.describe(rank, local_rank, dev) if hasattr(rig, "describe") else {"rank": rank, "local_rank": local_rank, "device": str(dev)}
    rank_map = [me]
    if distributed and ranks_seen == world:
        rank_map = [None] * world
        dist.all_gather_object(rank_map, me)
    if ranks_seen != args.gpus or world != args.gpus:
        # a line that claims N GPUs must come from N ranks: refuse to print one otherwise
        print(f"bench.py: --gpus {args.gpus} but {ranks_seen} rank(s) answered (WORLD_SIZE={world})", file=sys.stderr, flush=True)
        if distributed:
            dist.destroy_process_group()
        return 3

    # ---- workload: C2 mesh replicated, this rank's own 50 views (lawn-mower grid, per-rank tilt seed) -----------------
    points, faces = synthetic.terrain_mesh(wl.n_side, wl.extent)
    V, F = points.shape[0], faces.shape[0]
    cams = synthetic.survey_cameras(10, 5, 40.0, 60.0, seed=3 + rank, **wl.cam_kw())
    nv = min(args.views or wl.views_per_rank, len(cams))
    recs_np = cams.get_raster_records(1.0, near=1.0)[:nv]
    hip = rig.make_raster(local_rank)
    hip.upload_mesh(points.astype(np.float32), faces.astype(np.int32))
    recs = torch.from_numpy(recs_np).to(dev)
    ids = torch.empty((nv, H, W), dtype=torch.int32, device=dev)
    P = H * W

    def step():
        hip.raster_face_ids(recs, H, W, out=ids, check=False)

    hip.raster_face_ids(recs, H, W, out=ids, check=True)  # sizes the bin lists once (any overflow is retried here)
    blocks_per_view = hip.last_stats.get("blocks", 0) / max(nv, 1)  # 64-face blocks that pass the per-view frustum cull
    # untimed pre-conditioning (clocks, TLBs of the scratch): the device needs ~15 ms of this workload to reach its
    # steady state after start-up; without it a short run (K <= 10) reads 5-8 % lower than a long one
    for _ in range(12):
        step()
    for _ in range(args.warmup):
        step()
    # the library's HIP events (on the stream the kernels run on; for the two stages of a raster call they ride on the kernel
    # launches themselves -- hipExtLaunchKernelGGL stop events, a stage = [end of the kernel before it, end of its last kernel]
    # -- so that timing the kernels does not slow them: recorded events cost the step 1.65 %) stay ON through the
    # timed windows: the per-kernel durations of the roofline object are those of exactly the timed steps
    hip.set_profiling(True)
    # windows of EXACTLY --steps steps; at least --windows of them, and as many more as it takes to put --min-timed-s of GPU
    # time inside timed regions (every rank takes the same decision: it is made on the MAX-reduced times)
    window_s = []
    while len(window_s) < max(args.windows, 1) or (sum(window_s) < args.min_timed_s and len(window_s) < 400):
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
    launches = max(st["raster_launches"], 1)
    raster_ms_per_launch = st["raster_ms"] / launches
    setup_ms_per_launch = st["setup_ms"] / launches
    views_per_launch = st["views"] / launches
    # algorithmic bytes of the dominant kernel (k_raster_tile): the int32 id image it writes, 4*P per view.
    # (k_setup_cull owns the other part of B_r = 12V + 12F + 4P: the mesh read.)  DESIGN.md section "Kernels".
    stage_ms_per_view = {k: st[k] / max(st["views"], 1) for k in ("setup_ms", "scan_ms", "fill_ms", "raster_ms")}
    pipeline_ms_per_view = sum(stage_ms_per_view.values())
    br_bytes = 12.0 * V + 12.0 * F + 4.0 * P
    roofline = hbm_roofline("k_raster_tile", 4.0 * P * views_per_launch, raster_ms_per_launch, views_per_launch, "k_raster_tile")
    roofline["stage_ms_per_view"] = {k: round(v, 5) for k, v in stage_ms_per_view.items()}
    roofline["pipeline_GBs"] = round(br_bytes / max(pipeline_ms_per_view * 1e-3, 1e-12) / 1e9, 2)
    roofline["pipeline_frac"] = round(br_bytes / max(pipeline_ms_per_view * 1e-3, 1e-12) / 1e9 / HBM_PEAK_GBS, 5)
    roofline["valu"] = valu_bound("k_raster_tile", raster_ms_per_launch, views_per_launch)
    # the binning tax: HBM bytes the whole pipeline moves per view (committed PMC passes) over its algorithmic bytes B_r
    tj = _profile_json("traffic.json") or {}
    moved = [(tj.get(k) or {}).get("hbm_bytes_per_view") for k in ("k_cull_blocks", "k_setup_cull", "k_clip_faces", "k_bin_stats", "k_raster_tile")]
    # ... and over the bytes a CULLED pass must move: the id image, one 16-byte sphere per 64-face block tested, 36 bytes per
    # face of the blocks that pass the frustum test (B_r charges the whole mesh, which the block cull never reads)
    culled_bytes = 4.0 * P + 16.0 * ((F + 63) // 64) + 36.0 * 64.0 * blocks_per_view
    roofline["culled_algorithmic_bytes_per_view"] = round(culled_bytes, 1)
    roofline["blocks_surviving_cull_per_view"] = round(blocks_per_view, 1)
    if all(m is not None for m in moved):
        roofline["pipeline_traffic_per_view"] = round(sum(moved), 1)
        roofline["pipeline_traffic_over_algorithmic"] = round(sum(moved) / br_bytes, 4)
        roofline["pipeline_traffic_over_culled_algorithmic"] = round(sum(moved) / culled_bytes, 4)
    rooflines = {
        "k_setup_cull": hbm_roofline(
            "set-up stage of pix2face: k_cull_blocks + k_setup_cull + k_clip_faces (HIP events: end of k_bin_init -> end of k_clip_faces)",
            (12.0 * V + 12.0 * F) * views_per_launch, setup_ms_per_launch, views_per_launch, "k_setup_cull",
            note="12 V + 12 F per view: the mesh read of B_r (SURVEY 8d)"),
    }
    rooflines["k_setup_cull"]["valu"] = valu_bound("k_setup_cull", setup_ms_per_launch, views_per_launch)

    # ---- aggregation pipeline = BASELINE config 3: the 500-view camera grid, 4-class labels, aggregate_viewpoints on one GPU:
    #      fused raster + winners + votes in ONE call per step (8 launch groups), one all-reduce of the votes at N > 1 ----------
    aggregate = None
    labels = recs3 = recs3_np = None
    if not args.no_aggregate:
        cams3 = synthetic.config3_cameras(wl.c3_views, **wl.cam_kw())
        n3 = len(cams3)
        recs3_np = cams3.get_raster_records(1.0, near=1.0)
        recs3 = torch.from_numpy(recs3_np).to(dev)
        labels = torch.empty((n3, H, W), dtype=torch.uint8, device=dev)
        for c0 in range(0, n3, nv):  # labels are generated on the device from the ids, chunk by chunk
            c1 = min(c0 + nv, n3)
            hip.raster_face_ids(recs3[c0:c1], H, W, out=ids[: c1 - c0], check=(c0 == 0))
            for k in range(c1 - c0):
                labels[c0 + k] = device_labels(ids[k], c0 + k, N_CLASSES)
        votes, counts = hip.new_vote_buffers(N_CLASSES)
        local_obs = [0.0]

        def agg_step():
            votes.zero_()
            counts.zero_()
            hip.raster_project_labels(recs3, labels, N_CLASSES, votes, counts, ids_out=None, check=False)
            if distributed:
                local_obs[0] = float(counts.to(torch.int64).sum().item())  # this rank's own face observations
                all_reduce_votes(votes, counts)
            return hip.finalize_votes(votes, counts)

        hip.raster_project_labels(recs3, labels, N_CLASSES, votes, counts, check=True)  # sizing / warm-up pass
        chunk_visits_per_view = hip.last_stats.get("chunk_visits", 0) / max(n3, 1)  # 64-face groups the vote pass visits per view
        agg_step()
        # windows of 5 steps until at least 0.3 s of GPU time lie inside timed regions (decided on the MAX-reduced times)
        agg_steps, agg_elapsed = 0, 0.0
        while agg_steps == 0 or (agg_elapsed < wl.min_leg_s and agg_steps < 200):
            barrier()
            t0 = time.perf_counter()
            for _ in range(5):
                avg, summed, cnt = agg_step()
            barrier()
            agg_elapsed += max_over_ranks(time.perf_counter() - t0)
            agg_steps += 5
        hip.set_profiling(True)
        agg_step()
        ast = hip.stage_times()
        hip.set_profiling(False)
        agg_views = world * n3 * agg_steps
        f_vis = (local_obs[0] if distributed else float(cnt.sum().item())) / max(n3, 1)  # faces a view shows, on average
        a_launches = max(ast["raster_launches"], 1)
        a_vpl = ast["views"] / a_launches
        aggregate = {
            "workload": f"BASELINE config 3: {n3} views (25 x 20 grid) of the C2 mesh per GPU, {N_CLASSES}-class labels, fused raster + "
                        f"last-writer-wins projection (ids stay in LDS) + uint32 votes in one call per step, one RCCL all-reduce of "
                        f"[F x {N_CLASSES + 1}] int32 per step at N>1",
            "views_per_s": round(agg_views / agg_elapsed, 2),
            "mpix_per_s": round(agg_views / agg_elapsed * P / 1e6, 1),
            "ms_per_step": round(agg_elapsed / agg_steps * 1e3, 3),
            "timed_s": round(agg_elapsed, 4),
            "faces_observed": int((cnt > 0).sum().item()),
            "faces_seen_per_view": round(f_vis, 1),
            "setup_ms_per_view": round(ast["setup_ms"] / max(ast["views"], 1), 5),
            "raster_fused_ms_per_view": round(ast["raster_ms"] / max(ast["views"], 1), 5),
            "vote_ms_per_view": round(ast["vote_ms"] / max(ast["views"], 1), 5),
            "oracle_check": None,
        }
        # the fused tile kernel moves almost nothing through HBM by design (the ids stay in LDS; what leaves is one 4-byte
        # winner per candidate pixel): its HBM line is reported for completeness, its VALU line is the one that binds
        rooflines["k_raster_tile_fused"] = hbm_roofline(
            "k_raster_tile<FUSE> (fused aggregation)", 4.0 * f_vis * a_vpl, ast["raster_ms"] / a_launches, a_vpl,
            "k_raster_tile_fused", note="4 F_vis per view: the winners it writes (ids are never written); the entries it reads "
                                        "are a binning tax, not algorithmic bytes")
        rooflines["k_raster_tile_fused"]["valu"] = valu_bound("k_raster_tile_fused", ast["raster_ms"] / a_launches, a_vpl)
        # the vote kernel reads and resets the winners only of the 64-face groups in which the view's tile pass produced a winner
        # (gr_raster_stats.chunk_visits: the byte map the fused epilogue fills): 64 x 8 B per visited group, not 8 F
        vote_bytes = 64.0 * 8.0 * chunk_visits_per_view + 9.0 * f_vis
        rooflines["k_vote_labels"] = hbm_roofline(
            "k_vote_labels", vote_bytes * a_vpl, ast["vote_ms"] / a_launches, a_vpl, "k_vote_labels",
            note=f"64 x 8 B per visited group of 64 faces (winner read + reset; {chunk_visits_per_view:.0f} of {(F + 63) // 64} groups per view by "
                 "the winner map) + 9 F_vis (label byte, vote and count read-modify-write) per view; SURVEY 8d's 8 F + 9 F_vis "
                 f"is the upper bound {(8.0 * F + 9.0 * f_vis) / 1e6:.2f} MB per view")
        rooflines["k_vote_labels"]["chunk_visits_per_view"] = round(chunk_visits_per_view, 1)
        rooflines["k_vote_labels"]["valu"] = valu_bound("k_vote_labels", ast["vote_ms"] / a_launches, a_vpl)
        # whole fused pipeline against B_f = 12V + 12F + 1P + 8F + 8F_vis
        bf_bytes = 12.0 * V + 12.0 * F + 1.0 * P + 8.0 * F + 8.0 * f_vis
        aggregate["pipeline_GBs"] = round(bf_bytes * n3 * agg_steps / agg_elapsed / 1e9, 2)
        aggregate["pipeline_frac"] = round(bf_bytes * n3 * agg_steps / agg_elapsed / 1e9 / HBM_PEAK_GBS, 5)

    # ---- BASELINE config 4: this GPU's shard of the 2000-view set + the single all-reduce, timed separately --------------
    c4 = None
    if not args.no_c4:
        cams4 = synthetic.config4_cameras(**wl.cam_kw())
        total4 = min(wl.c4_views_total, len(cams4))
        mine = list(range(rank, total4, world))[:wl.c4_views_per_rank]
        n4 = len(mine)   # 0 when there are fewer views than ranks: the rank still joins every barrier and the vote reduce
        votes4, counts4 = hip.new_vote_buffers(N_CLASSES)
        recs4 = labels4 = None
        if n4:
            recs4_np = cams4.get_subset_cameras(mine).get_raster_records(1.0, near=1.0)
            recs4 = torch.from_numpy(recs4_np).to(dev)
            labels4 = torch.empty((n4, H, W), dtype=torch.uint8, device=dev)
            for c0 in range(0, n4, nv):  # labels are generated on the device from the ids, chunk by chunk
                c1 = min(c0 + nv, n4)
                hip.raster_face_ids(recs4[c0:c1], H, W, out=ids[: c1 - c0], check=(c0 == 0))
                for k in range(c1 - c0):
                    labels4[c0 + k] = device_labels(ids[k], mine[c0 + k], N_CLASSES)
            hip.raster_project_labels(recs4, labels4, N_CLASSES, votes4, counts4, check=True)  # sizing / warm-up pass
        t_local, t_reduce = [], []
        while len(t_local) < 3 or (sum(t_local) + sum(t_reduce) < wl.min_leg_s and len(t_local) < 400):
            votes4.zero_()
            counts4.zero_()
            barrier()
            t0 = time.perf_counter()
            if n4:
                hip.raster_project_labels(recs4, labels4, N_CLASSES, votes4, counts4, check=False)
            rig.synchronize(dev)
            t1 = time.perf_counter()
            if distributed:
                all_reduce_votes(votes4, counts4)
            barrier()
            t2 = time.perf_counter()
            t_local.append(max_over_ranks(t1 - t0))
            t_reduce.append(max_over_ranks(t2 - t1))
        tl, tr = statistics.median(t_local), statistics.median(t_reduce)
        total_counts = int(counts4.to(torch.int64).sum().item())
        reduce_bytes = F * (N_CLASSES + 1) * 4
        ring = 2.0 * (world - 1) / world * reduce_bytes
        c4 = {
            "workload": f"BASELINE config 4: {total4}-view set (C3 grid x 4 altitudes), view i -> GPU i mod {world}, "
                        f"{n4} views on this GPU, {N_CLASSES} classes, fused aggregation + ONE all-reduce of "
                        f"[{F} x {N_CLASSES + 1}] int32 ({reduce_bytes / 1e6:.1f} MB)",
            "views_per_gpu": n4,
            "aggregate_ms": round(tl * 1e3, 3),
            "all_reduce_ms": round(tr * 1e3, 3),
            "reps": len(t_local),
            "timed_s": round(sum(t_local) + sum(t_reduce), 4),
            "all_reduce_bytes": reduce_bytes,
            # ring all-reduce: every GPU sends and receives 2 (N-1)/N x S; over ONE xGMI link per direction, and over all
            # seven (the fully connected node lets RCCL run several rings side by side)
            "all_reduce_algorithmic_ms": {"bytes_on_the_wire_per_gpu": ring,
                                          "one_link": round(ring / (XGMI_LINK_GBS * 1e9) * 1e3, 4),
                                          "seven_links": round(ring / (7 * XGMI_LINK_GBS * 1e9) * 1e3, 4),
                                          "link_GBs": XGMI_LINK_GBS},
            "views_per_s": round(min(total4, world * wl.c4_views_per_rank) / (tl + tr), 2) if world > 1 else round(n4 / (tl + tr), 2),
            "face_observations_after_reduce": total_counts,
        }
        del labels4, votes4, counts4

    # ---- BASELINE config 5: this GPU's shard of the 5 M-face / 6000x4000 / 10-class stress config ------------------------
    c5 = None
    if not args.no_c5:
        c5 = leg_c5(rig, wl, rank, world, local_rank, dev, distributed, barrier, max_over_ranks)

    # ---- hostile workload (N == 1): terrain + 20 000 trees, cameras tilted 30-45 degrees; full and quarter resolution -------
    workload_2 = None
    workload_3 = None
    quarter = None
    if rank == 0 and world == 1 and not args.no_workload2 and rig.side_legs():
        workload_2 = leg_workload2(rig, local_rank, dev)
        workload_3 = leg_workload3(rig, local_rank, dev, wl)
        quarter = leg_quarter_scale(rig, local_rank, dev, points, faces, wl)
        if not args.no_aggregate:
            quarter["fused"] = leg_quarter_scale_fused(rig, local_rank, dev, points, faces, wl)

    # ---- PCIe-inclusive rates of the reference-shaped numpy API (N == 1) -------------------------------------------------------
    api = None
    if rank == 0 and world == 1 and not args.no_api and rig.side_legs():
        api = leg_api(points, faces, wl)

    # ---- the file-fed paths under the driver's clock (N == 1): label PNGs in, renders out ------------------------------------
    io = None
    if rank == 0 and world == 1 and not args.no_io and rig.side_legs():
        try:
            io = leg_io(points, faces, wl)
        except Exception as exc:  # the leg reports, it does not take the line down
            io = {"failed": repr(exc)}

    # ---- CPU baseline: the C oracle (a port of the rule-set; the reference's VTK path cannot run here) ------------------
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        oracle_c = rig.checker()
        cores = os.cpu_count() or 1
        t0 = time.perf_counter()
        one, _ = oracle_c.raster_views(points, faces, recs_np[:1], H, W, n_threads=1)
        t1 = time.perf_counter() - t0
        hip.raster_face_ids(recs[:1], H, W, out=ids[:1], check=True)  # view 0 again (the other legs reuse the id buffer)
        assert np.array_equal(one[0], ids[0].cpu().numpy()), "GPU ids differ from the CPU oracle on view 0"
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
                    "sample": f"{n_done} C2 views at {W}x{H} ({n_done // n_thr} passes of {n_thr}, one view per thread) on "
                              f"{used} threads of {cores} cores ({_cpu_model()}) in {tc:.1f} s",
                    "views_per_s": round(n_done / tc, 3)}

        quota = _cpu_quota_cores()  # the container's CPU allowance in cores (the pool's hosts: 16 of 256), or None
        n_samples = 3 if quota else 2
        s64 = sample(int(min(cores, 64, fit)), args.cpu_seconds / n_samples)
        sall = sample(int(min(cores, fit, 512)), args.cpu_seconds / n_samples) if cores > 64 and fit > 64 else s64
        squota = sample(int(max(1, min(cores, fit, round(quota)))), args.cpu_seconds / n_samples) if quota else None
        best = max([x for x in (s64, sall, squota) if x is not None], key=lambda x: x["value"])
        cpu_baseline = dict(best)
        cpu_baseline["kind"] = "port"
        cpu_baseline["all_cores"] = sall
        cpu_baseline["threads_64"] = s64
        cpu_baseline["threads_equal_to_cpu_quota"] = squota
        # `cores` is the number of threads the best sample ran; what the container lets them use at once is the smaller of
        # that and its cgroup quota
        cpu_baseline["effective_cores"] = min(best["cores"], quota) if quota else best["cores"]
        cpu_baseline["single_thread"] = {"value": round(P / t1 / 1e6, 2), "unit": "Mpix/s", "cores": 1,
                                         "sample": f"1 C2 view at {W}x{H} in {t1:.2f} s"}
        cpu_baseline["host_cores"] = cores
        # the real reference's rasterizer, for scale: a quoted constant with its source, NOT measured in this run (and not on this host)
        cpu_baseline["reference_gl"] = {
            "value": 22.0, "unit": "Mpix/s", "cores": 8, "kind": "mesa-llvmpipe", "where": "build container",
            "sample": "Mesa 23.2.1 llvmpipe (the software OpenGL of the reference's Dockerfile:6-13) draws the id image of one C2 view "
                      "(1 201 250 faces, 4000x3000) in 0.5-0.6 s on 8 cores: profiles/r05_gl_pin.log, tests/golden/gl_raster.py; the "
                      "draw alone, before VTK's per-view mesh upload, read-back and decode (meshes.py:1776-1836)"}
        cpu_baseline["cgroup_cpu_quota_cores"] = quota
        try:
            cpu_baseline["affinity_cores"] = len(os.sched_getaffinity(0))
        except (AttributeError, OSError):
            cpu_baseline["affinity_cores"] = None

    if rank == 0:
        # The driver keeps the `roofline` object whole and only the NAMES of the side legs: the scalars a reader needs from them
        # are repeated here.  binding_resource: what binds the dominant kernel (VALU issue) -- "bound": "hbm"
        # above names the roofline BASELINE.json asks the fraction of.
        roofline["binding_resource"] = "valu"
        roofline["valu_frac"] = (roofline.get("valu") or {}).get("frac")
        roofline["setup_us_per_view"] = round(stage_ms_per_view["setup_ms"] * 1e3, 3)
        roofline["raster_us_per_view"] = round(stage_ms_per_view["raster_ms"] * 1e3, 3)
        if c5:
            roofline["c5_kernel_frac"] = c5.get("raster_kernel_frac_of_hbm_peak")
            roofline["c5_raster_views_per_s"] = c5.get("raster_views_per_s")
        if aggregate:
            roofline["c3_views_per_s"] = aggregate.get("views_per_s")
        if workload_2:
            roofline["hostile_gpix_scale_1"] = round(workload_2["scale_1"]["mpix_per_s"] / 1e3, 2)
            roofline["hostile_gpix_scale_0.25"] = round(workload_2["scale_0.25"]["mpix_per_s"] / 1e3, 2)
            roofline["hostile_overflow_retries_cold"] = workload_2["scale_1"].get("overflow_retries_cold")
            roofline["hostile_first_group_rebinned_cold"] = workload_2["scale_1"].get("first_group_rebinned_cold")
        if workload_3:
            roofline["tin_gpix_scale_1"] = round(workload_3["scale_1"]["mpix_per_s"] / 1e3, 2)
            roofline["tin_gpix_scale_0.25"] = round(workload_3["scale_0.25"]["mpix_per_s"] / 1e3, 2)
            roofline["tin_fused_views_per_s_scale_1"] = workload_3["scale_1"]["fused_views_per_s"]
            roofline["tin_fused_views_per_s_scale_0.25"] = workload_3["scale_0.25"]["fused_views_per_s"]
        if quarter:
            roofline["c2_quarter_scale_gpix"] = round(quarter["mpix_per_s"] / 1e3, 2)
            roofline["c2_quarter_scale_views_per_s"] = quarter["views_per_s"]
            roofline["c2_quarter_scale_us_per_view"] = quarter["us_per_view"]
            if quarter.get("fused"):
                roofline["c3_quarter_scale_fused_views_per_s"] = quarter["fused"]["views_per_s"]
                roofline["c3_quarter_scale_fused_us_per_view"] = quarter["fused"]["us_per_view"]
        line = {
            "metric": "Mpix/s rasterized (face-ID pix2face), 1.2M-face mesh @ 4000x3000",
            "value": round(mpix_per_s, 1),
            "unit": "Mpix/s",
            "n_gpus": world,
            "ranks_seen": ranks_seen,
            "ranks": rank_map,
            "collective_backend": rig.backend_version() if hasattr(rig, "backend_version") else rig.dist_backend,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "ms_per_step_windows": {"n": len(ms_windows), "median": statistics.median(ms_windows), "min": min(ms_windows),
                                    "max": max(ms_windows), "first": ms_windows[:5], "last": ms_windows[-5:]},
            "timed_gpu_s": round(sum(window_s), 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int32 (fixed-point edge functions) / fp32 (vertex transform, depth)",
            "data": "synthetic",
            "config": {
                "workload": f"BASELINE config 2: {F}-face heightfield (V={V}), {nv} pinhole views/GPU @ {W}x{H}, "
                            f"f={wl.f:g} px, 120 m AGL lawn-mower grid, face-ID raster to int32 in HBM",
                "views_per_gpu_per_step": nv,
                "faces": F,
                "vertices": V,
                "parallelism": f"views sharded, mesh replicated, dp{world}",
                "workload_2": None if workload_2 is None else workload_2["workload"],
                "workload_3": None if workload_3 is None else workload_3["workload"],
            },
            "views_per_s": round(views_per_s, 2),
            "records_per_view": round(stats["records"] / max(nv, 1), 1),
            "bin_entries_per_view": round(stats["entries"] / max(nv, 1), 1),
            "roofline": roofline,
            "rooflines": rooflines,
            "cpu_baseline": cpu_baseline,
            "aggregate": aggregate,
            "c4": c4,
            "c5": c5,
            "workload_2": workload_2,
            "workload_3": workload_3,
            "quarter_scale": quarter,
            "api": api,
            "io": io,
        }
        print(json.dumps(line), flush=True)
    if distributed:
        dist.destroy_process_group()
    return 0


def leg_c5(rig, wl, rank, world, local_rank, dev, distributed, barrier, max_over_ranks):
    """BASELINE config 5 on this GPU's share of the views (view i -> GPU i mod N): ids-only on a sample of the shard, fused
    10-class aggregation on all of it, the vote all-reduce timed apart; one view against the oracle at N == 1."""
    import torch

    from geograypher_amd.distributed import all_reduce_votes
    from geograypher_amd.utils import synthetic

    H5, W5, C5 = wl.c5_H, wl.c5_W, wl.c5_classes
    P5 = H5 * W5
    pts5, faces5 = synthetic.terrain_mesh(wl.c5_n_side, wl.c5_extent)
    V5, F5 = pts5.shape[0], faces5.shape[0]
    cams5 = synthetic.survey_cameras(50, 40, 15.0, 18.0, agl=150.0, f=wl.c5_f, width=W5, height=H5, seed=6)
    total = min(wl.c5_views_total, len(cams5))
    mine = list(range(rank, total, world))[:wl.c5_views_per_rank]
    n5 = len(mine)   # 0 when there are fewer views than ranks: the rank still joins every barrier and the vote reduce
    hip5 = rig.make_raster(local_rank)
    hip5.upload_mesh(pts5.astype(np.float32), faces5.astype(np.int32))
    nr = min(wl.c5_raster_views, n5)
    recs_np = cams5.get_subset_cameras(mine).get_raster_records(1.0, near=1.0) if n5 else np.zeros((0, 16), dtype=np.float32)
    recs = torch.from_numpy(recs_np).to(dev)
    retries, st0 = 0, {"entries": 0, "max_entries": 0}
    ids5 = torch.empty((max(nr, 1), H5, W5), dtype=torch.int32, device=dev)
    if nr:
        hip5.raster_face_ids(recs[:nr], H5, W5, out=ids5[:nr], check=True)
        retries = hip5.last_retries
        st0 = dict(hip5.last_stats)
        for _ in range(2):
            hip5.raster_face_ids(recs[:nr], H5, W5, out=ids5[:nr], check=False)
    hip5.set_profiling(True)
    reps, t_raster = 0, 0.0
    while reps == 0 or (t_raster < wl.min_leg_s and reps < 2000):
        barrier()
        t0 = time.perf_counter()
        for _ in range(5):
            if nr:
                hip5.raster_face_ids(recs[:nr], H5, W5, out=ids5[:nr], check=False)
        barrier()
        t_raster += max_over_ranks(time.perf_counter() - t0)
        reps += 5
    st = hip5.stage_times()
    hip5.set_profiling(False)
    labels5 = torch.empty((n5, H5, W5), dtype=torch.uint8, device=dev)
    check_view, check_ids = nr // 2, None
    for c0 in range(0, n5, max(nr, 1)):  # labels are generated on the device from the ids, chunk by chunk
        c1 = min(c0 + nr, n5)
        hip5.raster_face_ids(recs[c0:c1], H5, W5, out=ids5[: c1 - c0], check=True)
        for k in range(c1 - c0):
            labels5[c0 + k] = device_labels(ids5[k], mine[c0 + k], C5)
        if c0 == 0:
            check_ids = ids5[check_view].cpu().numpy()
    del ids5
    votes5, counts5 = hip5.new_vote_buffers(C5)
    if n5:
        hip5.raster_project_labels(recs, labels5, C5, votes5, counts5, check=True)  # sizing / warm-up pass
    t_local, t_reduce = [], []
    while len(t_local) < 2 or (sum(t_local) + sum(t_reduce) < wl.min_leg_s and len(t_local) < 50):
        votes5.zero_()
        counts5.zero_()
        barrier()
        t0 = time.perf_counter()
        if n5:
            hip5.raster_project_labels(recs, labels5, C5, votes5, counts5, check=False)
        rig.synchronize(dev)
        t1 = time.perf_counter()
        if distributed:
            all_reduce_votes(votes5, counts5)
        barrier()
        t2 = time.perf_counter()
        t_local.append(max_over_ranks(t1 - t0))
        t_reduce.append(max_over_ranks(t2 - t1))
    tl, tr = min(t_local), min(t_reduce)
    reduce_bytes = F5 * (C5 + 1) * 4
    ring = 2.0 * (world - 1) / world * reduce_bytes
    br5 = 12.0 * V5 + 12.0 * F5 + 4.0 * P5
    launches = max(st["raster_launches"], 1)
    out = {
        "workload": f"BASELINE config 5: {F5}-face heightfield over {wl.c5_extent:g} m (V={V5}), {total}-view set {W5}x{H5} f={wl.c5_f:g} px "
                    f"150 m AGL, view i -> GPU i mod {world}: {n5} views on this GPU, {C5} classes; ids-only on {nr} of them",
        "views_per_gpu": n5,
        "raster_views_per_s": round(world * nr * reps / t_raster, 2),   # rank 0's share x N (every rank but a short last one has it)
        "raster_mpix_per_s": round(world * nr * reps / t_raster * P5 / 1e6, 1),
        "raster_stage_ms_per_view": {k: round(st[k] / max(st["views"], 1), 5) for k in ("setup_ms", "scan_ms", "fill_ms", "raster_ms")},
        "raster_kernel_frac_of_hbm_peak": round(4.0 * P5 * st["views"] / launches / max(st["raster_ms"] / launches * 1e-3, 1e-12) / 1e9
                                                / HBM_PEAK_GBS, 5),
        "pipeline_frac_of_hbm_peak": round(br5 * nr * reps / t_raster / 1e9 / HBM_PEAK_GBS, 5),
        "entries_per_view": round(st0["entries"] / max(nr, 1), 1),
        "max_entries_per_tile": int(st0["max_entries"]),
        "overflow_retries_first_call": int(retries),
        "raster_timed_s": round(t_raster, 4),
        "aggregate_ms": round(tl * 1e3, 3),
        "aggregate_reps": len(t_local),
        "aggregate_timed_s": round(sum(t_local) + sum(t_reduce), 4),
        "aggregate_views_per_s": round(min(total, world * wl.c5_views_per_rank) / (tl + tr), 2) if world > 1 else round(n5 / (tl + tr), 2),
        "all_reduce_ms": round(tr * 1e3, 3),
        "all_reduce_bytes": reduce_bytes,
        "all_reduce_algorithmic_ms": {"one_link": round(ring / (XGMI_LINK_GBS * 1e9) * 1e3, 4),
                                      "seven_links": round(ring / (7 * XGMI_LINK_GBS * 1e9) * 1e3, 4)},
        "face_observations_after_reduce": int(counts5.to(torch.int64).sum().item()),
        "oracle_check": None,
    }
    if rank == 0 and world == 1 and rig.side_legs():
        oracle_c = rig.checker()
        want = oracle_c.raster(pts5, faces5, recs_np[check_view], H5, W5)
        ok_ids = bool(np.array_equal(check_ids, want))
        v1, c1 = hip5.new_vote_buffers(C5)
        hip5.raster_project_labels(recs[check_view:check_view + 1], labels5[check_view:check_view + 1], C5, v1, c1, check=True)
        want_v = np.zeros((F5, C5), dtype=np.uint32)
        want_c = np.zeros(F5, dtype=np.uint32)
        oracle_c.project_labels(want, labels5[check_view].cpu().numpy(), F5, C5, want_v, want_c)
        ok_votes = bool(np.array_equal(v1.cpu().numpy().view(np.uint32), want_v) and
                        np.array_equal(c1.cpu().numpy().view(np.uint32), want_c))
        out["oracle_check"] = f"view {mine[check_view]}: ids equal the CPU oracle's: {ok_ids}; fused votes and counts equal: {ok_votes}"
        assert ok_ids and ok_votes, "config 5 differs from the CPU oracle"
    del labels5, votes5, counts5, hip5
    return out


def leg_quarter_scale(rig, local_rank, dev, points, faces, wl):
    """BASELINE config 2's mesh and cameras at render_img_scale = 0.25 (1000 x 750): the reference's documented operating point
    for aggregation (AGGREGATE_IMAGE_SCALE = 0.25, examples/aggregate_predictions.ipynb:60-61).  A face is about 3 pixels wide
    there: the first call teaches the library to keep micro lists for this mesh and image size (one face per lane in the tile
    kernel), the timed calls use them.  One view against the CPU oracle."""
    import torch

    from geograypher_amd.utils import synthetic

    oracle_c = rig.checker()
    cams = synthetic.survey_cameras(10, 5, 40.0, 60.0, seed=3, **wl.cam_kw())
    h, w = cams[0].get_image_size(0.25)
    recs_np = cams.get_raster_records(0.25, near=1.0)
    recs = torch.from_numpy(recs_np).to(dev)
    hip = rig.make_raster(local_rank)
    hip.upload_mesh(points.astype(np.float32), faces.astype(np.int32))
    out = torch.empty((recs.shape[0], h, w), dtype=torch.int32, device=dev)
    hip.raster_face_ids(recs, h, w, out=out, check=True)       # sizes the segments; counts the micro faces
    hip.raster_face_ids(recs, h, w, out=out, check=True)       # micro lists from here on
    for _ in range(3):
        hip.raster_face_ids(recs, h, w, out=out, check=False)
    hip.set_profiling(True)
    n_rep, windows = 40, []
    for _ in range(3):   # the median of three windows (11 ms each)
        rig.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(n_rep):
            hip.raster_face_ids(recs, h, w, out=out, check=False)
        rig.synchronize(dev)
        windows.append(time.perf_counter() - t0)
    dt = sorted(windows)[1]
    stg = hip.stage_times()
    hip.set_profiling(False)
    st = hip.raster_status()
    want = oracle_c.raster(points, faces, recs_np[7], h, w)
    same = bool(np.array_equal(out[7].cpu().numpy(), want))
    assert same, "quarter-scale leg: GPU ids differ from the CPU oracle"
    res = {
        "workload": f"BASELINE config 2 at render_img_scale 0.25: {faces.shape[0]} faces, {recs.shape[0]} views {w}x{h}",
        "views_per_s": round(n_rep * recs.shape[0] / dt, 1),
        "mpix_per_s": round(n_rep * recs.shape[0] * h * w / dt / 1e6, 1),
        "us_per_view": {"setup": round(stg["setup_ms"] / max(stg["views"], 1) * 1e3, 2),
                        "raster": round(stg["raster_ms"] / max(stg["views"], 1) * 1e3, 2)},
        "entries_per_view": round(st["entries"] / recs.shape[0], 1),
        "max_entries_per_tile": int(st["max_entries"]),
        "oracle_parity_view_7": same,
    }
    del out, hip
    return res


def leg_quarter_scale_fused(rig, local_rank, dev, points, faces, wl):
    """BASELINE config 3 at the reference's operating point: the 500-view camera grid of the C2 mesh through the FUSED
    aggregation (`raster_project_labels`: raster + last-writer-wins projection + votes, ids never leave the chip) at
    aggregate_img_scale = 0.25 (1000 x 750; AGGREGATE_IMAGE_SCALE of examples/aggregate_predictions.ipynb:61, consumed at
    meshes.py:1959, 1987-2002), 4 classes.  Set-up / tile kernel / vote microseconds per view from the library's HIP events; the
    votes of one view against the CPU oracle."""
    import torch

    from geograypher_amd.utils import synthetic

    oracle_c = rig.checker()
    C = wl.n_classes
    cams = synthetic.config3_cameras(wl.c3_views, **wl.cam_kw())
    n = len(cams)
    h, w = cams[0].get_image_size(0.25)
    recs_np = cams.get_raster_records(0.25, near=1.0)
    recs = torch.from_numpy(recs_np).to(dev)
    hip = rig.make_raster(local_rank)
    hip.upload_mesh(points.astype(np.float32), faces.astype(np.int32))
    F = faces.shape[0]
    labels = torch.empty((n, h, w), dtype=torch.uint8, device=dev)
    step_views = 50
    ids = torch.empty((step_views, h, w), dtype=torch.int32, device=dev)
    for c0 in range(0, n, step_views):   # labels are generated on the device from the ids, chunk by chunk
        c1 = min(c0 + step_views, n)
        hip.raster_face_ids(recs[c0:c1], h, w, out=ids[: c1 - c0], check=True)
        for k in range(c1 - c0):
            labels[c0 + k] = device_labels(ids[k], c0 + k, C)
    del ids
    votes, counts = hip.new_vote_buffers(C)
    hip.raster_project_labels(recs, labels, C, votes, counts, check=True)   # sizes the segments (fused call), learns micro lists
    hip.raster_project_labels(recs, labels, C, votes, counts, check=True)
    st_sized = dict(hip.last_stats)
    reps, elapsed = 0, 0.0
    while reps == 0 or (elapsed < wl.min_leg_s and reps < 400):
        rig.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(5):
            votes.zero_()
            counts.zero_()
            hip.raster_project_labels(recs, labels, C, votes, counts, check=False)
        rig.synchronize(dev)
        elapsed += time.perf_counter() - t0
        reps += 5
    f_vis = float(counts.to(torch.int64).sum().item()) / max(n, 1)
    hip.set_profiling(True)
    hip.raster_project_labels(recs, labels, C, votes, counts, check=False)
    stg = hip.stage_times()
    hip.set_profiling(False)
    check_view = min(7, n - 1)
    v1, c1 = hip.new_vote_buffers(C)
    hip.raster_project_labels(recs[check_view:check_view + 1], labels[check_view:check_view + 1], C, v1, c1, check=True)
    want = oracle_c.raster(points, faces, recs_np[check_view], h, w)
    want_v = np.zeros((F, C), dtype=np.uint32)
    want_c = np.zeros(F, dtype=np.uint32)
    oracle_c.project_labels(want, labels[check_view].cpu().numpy(), F, C, want_v, want_c)
    same = bool(np.array_equal(v1.cpu().numpy().view(np.uint32), want_v) and np.array_equal(c1.cpu().numpy().view(np.uint32), want_c))
    assert same, "quarter-scale fused leg: votes differ from the CPU oracle"
    views = max(stg["views"], 1)
    res = {
        "workload": f"BASELINE config 3 at aggregate_img_scale 0.25: {n} views {w}x{h} of the {F}-face C2 mesh, {C} classes, fused "
                    "raster + projection + votes in one call per step",
        "views_per_s": round(n * reps / elapsed, 1),
        "ms_per_step": round(elapsed / reps * 1e3, 3),
        "timed_s": round(elapsed, 4),
        "us_per_view": {"setup": round(stg["setup_ms"] / views * 1e3, 2), "raster_fused": round(stg["raster_ms"] / views * 1e3, 2),
                        "vote": round(stg["vote_ms"] / views * 1e3, 2)},
        "faces_seen_per_view": round(f_vis, 1),
        "entries_per_view": round(st_sized.get("entries", 0) / max(n, 1), 1),
        "chunk_visits_per_view": round(st_sized.get("chunk_visits", 0) / max(n, 1), 1),
        "oracle_check": f"votes and counts of view {check_view} (fused call) equal the CPU oracle's: {same}",
    }
    del labels, votes, counts, hip
    return res


def leg_workload2(rig, local_rank, dev):
    """Terrain + 20 000 trees seen obliquely, full and quarter resolution.  The slots per tile such images need are learned
    by the very first call in this process: the library reads the counts of that call's first launch group before the group's
    tile kernel runs and bins it again with segments that fit (`first_group_rebinned_cold`; rounds 1-4 finished the overflowed
    call and the caller repeated it: `overflow_retries_cold`); a second, fresh context -- the one that is timed -- starts with
    what the process has learned (`overflow_retries_first_call`)."""
    import torch

    from geograypher_amd.utils import synthetic

    oracle_c = rig.checker()
    fpts, ffaces = synthetic.forest_scene()
    fcams = synthetic.oblique_cameras(20)
    out = {"workload": f"C2 terrain + 20 000 trees ({ffaces.shape[0]} faces, cone canopies on cylinder trunks: "
                       "geograypher/utils/example_data.py:30-112 restated), 20 cameras tilted 30-45 degrees"}
    cold = {}
    # an EMPTY learned-binning cache for this leg: `overflow_retries_cold` is what a process pays that has never seen the
    # scene; the cache file it leaves behind is what a NEW process starts from (`overflow_retries_new_process`)
    import tempfile

    from geograypher_amd import _hip

    cache_dir = tempfile.mkdtemp(prefix="geograster_cache_")
    lib = _hip.load_library()
    lib.gr_learned_cache_clear()   # whatever earlier legs (or an earlier run's file in the user's cache folder) taught the process
    lib.gr_learned_cache_file(str(Path(cache_dir, "geograster_learned.txt")).encode())
    hip_cold = rig.make_raster(local_rank)
    hip_cold.upload_mesh(fpts.astype(np.float32), ffaces.astype(np.int32))
    for scale in (1.0, 0.25):
        h2, w2 = fcams[0].get_image_size(scale)
        r2 = torch.from_numpy(fcams.get_raster_records(scale, near=1.0)).to(dev)
        o2 = torch.empty((len(fcams), h2, w2), dtype=torch.int32, device=dev)
        rig.synchronize(dev)
        t0 = time.perf_counter()
        hip_cold.raster_face_ids(r2, h2, w2, out=o2, check=True)
        # (GR_EOVERFLOW retries of the call, times the library binned the call's first launch group again after its look at the
        # group's counts -- gr_raster_stats.rebinned_groups: what replaced the retry in round 5 --, wall time of the call incl.
        # the allocation of the entry memory)
        cold[scale] = (int(hip_cold.last_retries), int(hip_cold.last_stats.get("rebinned_groups", 0)), round((time.perf_counter() - t0) * 1e3, 2))
        del o2
    del hip_cold
    new_process = {}
    try:  # a fresh process (a child: never exec from a GPU process) that finds the cache file: no overflowed first pass
        child = (
            "import sys, json, numpy as np, torch\n"
            f"sys.path.insert(0, {str(ROOT)!r})\n"
            "from geograypher_amd._hip import HipRaster\n"
            "from geograypher_amd.utils import synthetic\n"
            "pts, faces = synthetic.forest_scene(); cams = synthetic.oblique_cameras(20)\n"
            "h = HipRaster(0); h.upload_mesh(pts.astype(np.float32), faces.astype(np.int32)); out = {}\n"
            "for scale in (1.0, 0.25):\n"
            "    hh, ww = cams[0].get_image_size(scale)\n"
            "    h.raster_face_ids(cams.get_raster_records(scale, near=1.0)[:4], hh, ww)\n"
            "    out[str(scale)] = int(h.last_retries)\n"
            "print('RETRIES', json.dumps(out))\n"
        )
        res = subprocess.run([sys.executable, "-c", child], env=dict(os.environ, GEOGRAYPHER_AMD_CACHE=cache_dir), capture_output=True,
                             text=True, timeout=300)
        line = [l for l in res.stdout.splitlines() if l.startswith("RETRIES")]
        new_process = json.loads(line[-1].split(" ", 1)[1]) if line else {"failed": res.stderr[-300:]}
    except Exception as exc:
        new_process = {"failed": repr(exc)}
    hip2 = rig.make_raster(local_rank)
    hip2.upload_mesh(fpts.astype(np.float32), ffaces.astype(np.int32))
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
        hip2.set_profiling(True)
        n_rep, windows = 10, []
        for _ in range(3):   # the median of three windows: one host hiccup in 36 ms of calls moved the figure by 15 % (round 6)
            rig.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(n_rep):
                hip2.raster_face_ids(r2, h2, w2, out=out2, check=False)
            rig.synchronize(dev)
            windows.append(time.perf_counter() - t0)
        dt = sorted(windows)[1]
        stg = hip2.stage_times()
        hip2.set_profiling(False)
        want = oracle_c.raster(fpts, ffaces, r2_np[3], h2, w2)
        same = bool(np.array_equal(out2[3].cpu().numpy(), want))
        out[f"scale_{scale:g}"] = {
            "image": f"{w2}x{h2}",
            "mpix_per_s": round(n_rep * len(fcams) * h2 * w2 / dt / 1e6, 1),
            "views_per_s": round(n_rep * len(fcams) / dt, 1),
            "us_per_view": {"setup": round(stg["setup_ms"] / max(stg["views"], 1) * 1e3, 1),
                            "raster": round(stg["raster_ms"] / max(stg["views"], 1) * 1e3, 1)},
            "entries_per_view": round(st2["entries"] / len(fcams), 1),
            "max_entries_per_tile": int(st2["max_entries"]),
            "overflow_retries_first_call": int(retries),
            "overflow_retries_cold": cold[scale][0],
            "first_group_rebinned_cold": cold[scale][1],
            "cold_call_ms": cold[scale][2],
            "overflow_retries_new_process": new_process.get(f"{scale:g}" if f"{scale:g}" in new_process else str(scale), new_process.get("failed")),
            "oracle_parity_view_3": same,
            "covered_fraction": round(float((want >= 0).mean()), 4),
        }
        assert same, f"workload_2 scale {scale}: GPU ids differ from the CPU oracle"
        del out2
    del hip2
    # the leg's cache file goes away with the leg: the process-wide table points back at where the binding attached it (or
    # nowhere), and the temporary directory is removed
    import shutil

    lib.gr_learned_cache_file(None)
    _hip._attach_learned_cache(lib)
    shutil.rmtree(cache_dir, ignore_errors=True)
    return out


def leg_workload3(rig, local_rank, dev, wl):
    """A realistic third workload between the friendliest (C2 height field: every triangle 13 px, depth complexity 1.0) and the
    most hostile (forest) one: an irregular TIN of 1.2 M faces (synthetic.tin_mesh: log-normal vertex density, Delaunay slivers,
    folded bumps on 5 % of the area -- what BASELINE config 2 calls the "Example-data Metashape mesh") under the 50 C2 cameras,
    ids-only and fused (4 classes), at full size and at the reference's aggregate_img_scale 0.25; one view per scale against the
    CPU oracle."""
    import torch

    from geograypher_amd.utils import synthetic

    oracle_c = rig.checker()
    C = wl.n_classes
    pts, faces = synthetic.tin_mesh()
    F = faces.shape[0]
    cams = synthetic.config2_cameras(50, **wl.cam_kw())
    n = len(cams)
    hip = rig.make_raster(local_rank)
    hip.upload_mesh(pts.astype(np.float32), faces.astype(np.int32))
    tri = pts[faces]
    area = 0.5 * np.linalg.norm(np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]), axis=1)
    out = {"workload": f"irregular TIN, {F} faces (V={pts.shape[0]}): Delaunay of blue-noise points with log-normal density over the C2 "
                       f"terrain spectrum (log-area sigma {float(np.log(area).std()):.2f}, area p1 / p50 / p99 = "
                       f"{np.percentile(area, 1):.3f} / {np.percentile(area, 50):.3f} / {np.percentile(area, 99):.3f} m2) + folded "
                       f"bumps on 5 % of the area; {n} C2 cameras"}
    for scale in (1.0, 0.25):
        h, w = cams[0].get_image_size(scale)
        recs_np = cams.get_raster_records(scale, near=1.0)
        recs = torch.from_numpy(recs_np).to(dev)
        ids = torch.empty((n, h, w), dtype=torch.int32, device=dev)
        hip.raster_face_ids(recs, h, w, out=ids, check=True)
        retries = int(hip.last_retries)
        hip.raster_face_ids(recs, h, w, out=ids, check=True)      # (a second checked call: micro lists, once learned, are on)
        st = dict(hip.last_stats)
        for _ in range(3):
            hip.raster_face_ids(recs, h, w, out=ids, check=False)
        hip.set_profiling(True)
        reps, dt = 0, 0.0
        while reps == 0 or (dt < wl.min_leg_s and reps < 400):
            rig.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(5):
                hip.raster_face_ids(recs, h, w, out=ids, check=False)
            rig.synchronize(dev)
            dt += time.perf_counter() - t0
            reps += 5
        stg = hip.stage_times()
        hip.set_profiling(False)
        check_view = 23
        want = oracle_c.raster(pts, faces, recs_np[check_view], h, w)
        same = bool(np.array_equal(ids[check_view].cpu().numpy(), want))
        assert same, f"workload_3 scale {scale}: GPU ids differ from the CPU oracle"
        labels = torch.empty((n, h, w), dtype=torch.uint8, device=dev)
        for k in range(n):
            labels[k] = device_labels(ids[k], k, C)
        del ids
        votes, counts = hip.new_vote_buffers(C)
        hip.raster_project_labels(recs, labels, C, votes, counts, check=True)
        hip.raster_project_labels(recs, labels, C, votes, counts, check=True)
        freps, fdt = 0, 0.0
        while freps == 0 or (fdt < wl.min_leg_s and freps < 400):
            rig.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(5):
                votes.zero_()
                counts.zero_()
                hip.raster_project_labels(recs, labels, C, votes, counts, check=False)
            rig.synchronize(dev)
            fdt += time.perf_counter() - t0
            freps += 5
        hip.set_profiling(True)
        hip.raster_project_labels(recs, labels, C, votes, counts, check=False)
        fst = hip.stage_times()
        hip.set_profiling(False)
        v1, c1 = hip.new_vote_buffers(C)
        hip.raster_project_labels(recs[check_view:check_view + 1], labels[check_view:check_view + 1], C, v1, c1, check=True)
        want_v = np.zeros((F, C), dtype=np.uint32)
        want_c = np.zeros(F, dtype=np.uint32)
        oracle_c.project_labels(want, labels[check_view].cpu().numpy(), F, C, want_v, want_c)
        same_votes = bool(np.array_equal(v1.cpu().numpy().view(np.uint32), want_v) and np.array_equal(c1.cpu().numpy().view(np.uint32), want_c))
        assert same_votes, f"workload_3 scale {scale}: fused votes differ from the CPU oracle"
        T = ((w + 63) // 64) * ((h + 31) // 32)
        views, fviews = max(stg["views"], 1), max(fst["views"], 1)
        out[f"scale_{scale:g}"] = {
            "image": f"{w}x{h}",
            "mpix_per_s": round(reps * n * h * w / dt / 1e6, 1),
            "views_per_s": round(reps * n / dt, 1),
            "us_per_view": {"setup": round(stg["setup_ms"] / views * 1e3, 2), "raster": round(stg["raster_ms"] / views * 1e3, 2)},
            "fused_views_per_s": round(freps * n / fdt, 1),
            "fused_us_per_view": {"setup": round(fst["setup_ms"] / fviews * 1e3, 2), "raster_fused": round(fst["raster_ms"] / fviews * 1e3, 2),
                                  "vote": round(fst["vote_ms"] / fviews * 1e3, 2)},
            "records_per_view": round(st["records"] / n, 1),
            "entries_per_view": round(st["entries"] / n, 1),
            "entries_per_tile": round(st["entries"] / n / T, 1),
            "max_entries_per_tile": int(st["max_entries"]),
            "overflow_retries_first_call": retries,
            "depth_complexity_view_23": round(synthetic.depth_complexity(pts, faces, recs_np[check_view], h, w), 3),
            "covered_fraction_view_23": round(float((want >= 0).mean()), 4),
            "faces_visible_view_23": int(np.unique(want[want >= 0]).size),
            "oracle_parity_view_23": same,
            "oracle_votes_view_23": same_votes,
        }
        del labels, votes, counts
    del hip
    return out


def host_image_set(base, images):
    """A camera set that serves in-memory images (float images, class-index images) to the general aggregation paths."""
    from geograypher_amd.cameras import PhotogrammetryCameraSet

    class HostImageSet(PhotogrammetryCameraSet):
        thread_safe_lookup = True  # in-memory arrays: the view loop may stage view i + 1 on its loader thread

        def __init__(self, base, images):
            self.base_camera_set, self.images, self.cameras = base, images, base.cameras
            self._local_to_epsg_4978_transform = base._local_to_epsg_4978_transform
            self._maps_ideal_to_warped, self._maps_warped_to_ideal = {}, {}
            self.image_folder = getattr(base, "image_folder", None)

        def __len__(self):
            return len(self.images)

        def n_image_channels(self):
            im = np.asarray(self.images[0])
            return 1 if im.ndim == 2 else int(im.shape[-1])

        def get_subset_cameras(self, inds):
            return HostImageSet(self.base_camera_set.get_subset_cameras(inds), [self.images[i] for i in inds])

        def get_image_by_index(self, i, image_scale=1.0):
            return self.images[i]

    return HostImageSet(base, images)


def leg_api(points, faces, wl, n_views=16):
    """What a drop-in caller of the reference-shaped API gets, host copies included (C2 views): numpy in, numpy out."""
    import torch

    from geograypher_amd.cameras import SegmentorPhotogrammetryCameraSet
    from geograypher_amd.meshes import TexturedPhotogrammetryMesh
    from geograypher_amd.meshes.derived_meshes import TexturedPhotogrammetryMeshIndexPredictions
    from geograypher_amd.predictors import ArrayLabelSegmentor
    from geograypher_amd.utils import synthetic

    cams = synthetic.config2_cameras(n_views, **wl.cam_kw())
    n = len(cams)
    C = wl.n_classes
    tex = (np.arange(faces.shape[0]) % C).astype(float)
    mesh = TexturedPhotogrammetryMesh((points, faces), texture=tex, IDs_to_labels={i: str(i) for i in range(C)}, log_level="ERROR")
    mesh.pix2face(cams[0:2], apply_distortion=False)  # warm up (upload, scratch)
    out = {"workload": f"{n} C2 views {wl.W}x{wl.H} through geograypher_amd's numpy-in / numpy-out methods, host copies included"}

    def timed(fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        return r, time.perf_counter() - t0

    ids, dt = timed(lambda: mesh.pix2face(cams, apply_distortion=False))
    del ids
    ids, dt = timed(lambda: mesh.pix2face(cams, apply_distortion=False))  # the pinned block of the first call is reused
    out["pix2face_int64_numpy_views_per_s"] = round(n / dt, 1)
    _, dt = timed(lambda: mesh.pix2face(cams, apply_distortion=False, return_tensor=True))
    out["pix2face_device_tensor_views_per_s"] = round(n / dt, 1)
    cnt, dt = timed(lambda: sum(1 for _ in mesh.render_flat(cams, apply_distortion=False)))
    out["render_flat_f64_numpy_views_per_s"] = round(cnt / dt, 1)
    labels = [synthetic.synthetic_labels(ids[v], v, C) for v in range(n)]
    names = [c.image_filename for c in cams.cameras]
    seg = SegmentorPhotogrammetryCameraSet(cams, ArrayLabelSegmentor(labels, C, filenames=names))
    mesh.aggregate_projected_images(seg)
    _, dt = timed(lambda: mesh.aggregate_projected_images(seg))
    out["aggregate_uint8_labels_from_host_views_per_s"] = round(n / dt, 1)
    # the same call through devices=[0, 0]: two libgeograster contexts, two host threads, two streams on this ONE GPU, partial votes
    # added on the first -- the single-process multi-device route of the unchanged caller (no second GPU on this box: what the
    # number shows is that the threaded route costs nothing; the votes must equal the single-device call's bit for bit)
    try:
        mesh2 = TexturedPhotogrammetryMesh((points, faces), texture=tex, IDs_to_labels={i: str(i) for i in range(C)}, log_level="ERROR",
                                           devices=[0, 0])
        want_avg, want_info = mesh.aggregate_projected_images(seg)
        mesh2.aggregate_projected_images(seg)
        (got_avg, got_info), dt2 = timed(lambda: mesh2.aggregate_projected_images(seg))
        same = bool(np.array_equal(np.nan_to_num(got_avg, nan=-7.0), np.nan_to_num(want_avg, nan=-7.0))
                    and np.array_equal(got_info["projection_counts"], want_info["projection_counts"]))
        out["aggregate_uint8_labels_from_host_devices_0_0_views_per_s"] = round(n / dt2, 1)
        out["aggregate_devices_0_0_equals_single_device"] = same
        assert same, "devices=[0, 0] aggregation differs from the single-device result"
        del mesh2
    except AssertionError:
        raise
    except Exception as exc:  # the leg reports, it does not take the line down
        out["aggregate_uint8_labels_from_host_devices_0_0_views_per_s"] = f"failed: {exc!r}"
    # the general path of meshes.py:2057-2067, images from host memory: (h, w, 3) uint8 photos (what `aggregate_images` feeds
    # it), (h, w, C) bool one-hot masks (what a segmentor returns) and (h, w, 3) float64 images
    nf = min(n, 8)
    rng = np.random.default_rng(0)
    sets = {
        "uint8_rgb": [rng.integers(0, 255, size=(wl.H, wl.W, 3), dtype=np.uint8) for _ in range(nf)],
        "bool_one_hot": [labels[v][..., None] == np.arange(C) for v in range(nf)],
        "float64_rgb": [rng.random((wl.H, wl.W, 3)) for _ in range(nf)],
    }
    for tag, imgs in sets.items():
        key = f"aggregate_{tag}_images_from_host_views_per_s"
        try:
            fset = host_image_set(cams[0:nf], imgs)
            mesh.aggregate_projected_images(fset)
            _, dt = timed(lambda: mesh.aggregate_projected_images(fset))
            out[key] = round(nf / dt, 1)
        except Exception as exc:  # the leg reports, it does not take the line down
            out[key] = f"failed: {exc!r}"
        del imgs
    sets.clear()
    # FILE-BACKED photos through a plain PhotogrammetryCameraSet (cameras.py:154-174 -> meshes.py:1988): uint8 PNGs read on the
    # loader thread, uploaded as uint8, `/ 255.0` and scikit-image's anti-aliased resize on the device (gr_resize_image_f64);
    # scale 0.25 is the reference's example `aggregate_image_scale` (entrypoints/aggregate_images.py:184)
    try:
        import tempfile

        from PIL import Image

        from geograypher_amd.cameras import PhotogrammetryCamera, PhotogrammetryCameraSet

        with tempfile.TemporaryDirectory() as d:
            files = []
            for v in range(nf):
                path = Path(d) / f"photo_{v}.png"
                Image.fromarray(rng.integers(0, 255, size=(wl.H, wl.W, 3), dtype=np.uint8)).save(path, compress_level=0)
                files.append(path)
            fcams = PhotogrammetryCameraSet([
                PhotogrammetryCamera(files[v], c.cam_to_world_transform, c.f, c.cx, c.cy, c.image_width, c.image_height)
                for v, c in enumerate(cams.cameras[:nf])])
            for tag, scale in (("scale_1", 1.0), ("scale_0.25", 0.25)):
                mesh.aggregate_projected_images(fcams, aggregate_img_scale=scale)
                _, dt = timed(lambda: mesh.aggregate_projected_images(fcams, aggregate_img_scale=scale))
                out[f"aggregate_uint8_photo_files_{tag}_views_per_s"] = round(nf / dt, 1)
            # with the opt-in decoded-input cache: the first pass stores the decoded photos, later passes memory-map them
            cache_dir = Path(d) / "decoded_cache"
            mesh.aggregate_projected_images(fcams, aggregate_img_scale=0.25, decoded_cache=cache_dir)
            _, dt = timed(lambda: mesh.aggregate_projected_images(fcams, aggregate_img_scale=0.25, decoded_cache=cache_dir))
            out["aggregate_uint8_photo_files_scale_0.25_decoded_cache_later_pass_views_per_s"] = round(nf / dt, 1)
            # the resize alone, photo resident on the device
            dev_img = torch.from_numpy(np.asarray(Image.open(files[0]))).cuda()
            mesh.backend.resize_image(dev_img, (wl.H // 4, wl.W // 4))
            _, dt = timed(lambda: [mesh.backend.resize_image(dev_img, (wl.H // 4, wl.W // 4)) for _ in range(20)])
            out["device_resize_uint8_photo_to_quarter_ms"] = round(dt / 20 * 1e3, 3)
    except Exception as exc:
        out["aggregate_uint8_photo_files_views_per_s"] = f"failed: {exc!r}"
    # sparse index aggregation (derived_meshes.py:414-550): one (h, w) class-index image per view
    try:
        sparse_mesh = TexturedPhotogrammetryMeshIndexPredictions((points, faces), log_level="ERROR", backend=mesh.backend)
        idx_imgs = []
        for v in range(nf):
            im = labels[v].astype(np.float64)
            im[im >= C] = np.nan
            idx_imgs.append(im)
        iset = host_image_set(cams[0:nf], idx_imgs)
        sparse_mesh.aggregate_projected_images(iset, n_classes=C)
        _, dt = timed(lambda: sparse_mesh.aggregate_projected_images(iset, n_classes=C))
        out["aggregate_sparse_index_images_from_host_views_per_s"] = round(nf / dt, 1)
    except Exception as exc:
        out["aggregate_sparse_index_images_from_host_views_per_s"] = f"failed: {exc!r}"
    return out


def leg_io(points, faces, wl, n_views=64):
    """The two places a real `aggregate_images` / `render_labels` run spends its time, under the driver's clock:
    (i) aggregate_projected_images fed by LookUpSegmentor from class-index PNG files (predictors/derived_segmentors.py:38-51),
    scale 1 and 0.25; (ii) save_renders (meshes.py:2248-2397) as deflate TIFF and as .npy.  Views per second, the host threads
    used, the container's CPU allowance, and the stage that binds."""
    import tempfile
    from concurrent.futures import ThreadPoolExecutor

    import torch
    from PIL import Image

    from geograypher_amd.cameras import SegmentorPhotogrammetryCameraSet
    from geograypher_amd.meshes import TexturedPhotogrammetryMesh
    from geograypher_amd.predictors import LookUpSegmentor
    from geograypher_amd.utils import synthetic

    t_leg = time.perf_counter()
    quota = _cpu_quota_cores()
    threads = int(min(16, os.cpu_count() or 1))
    C = 6
    cams = synthetic.config3_cameras(n_views, **wl.cam_kw())
    n = len(cams)
    tex = (synthetic.hash32(np.arange(faces.shape[0])) % 5).astype(np.float64)
    mesh = TexturedPhotogrammetryMesh((points, faces), texture=tex, IDs_to_labels={i: str(i) for i in range(5)}, log_level="ERROR")
    ids = mesh.pix2face(cams[0:4], apply_distortion=False)
    out = {"workload": f"{n} C3 views {wl.W}x{wl.H}: class-index PNG files -> aggregate_projected_images (LookUpSegmentor); "
                       "save_renders of a discrete 1-channel texture, cast to uint8", "host_threads": threads,
           "cgroup_cpu_quota_cores": quota, "host_cores": os.cpu_count()}

    def timed(fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        return r, time.perf_counter() - t0

    with tempfile.TemporaryDirectory() as d:
        base, lookup = Path(d) / "images", Path(d) / "labels"
        lookup.mkdir(parents=True)
        labs = [synthetic.synthetic_labels(ids[v % 4], v, C) for v in range(4)]
        for i, cam in enumerate(cams.cameras):
            cam.image_filename = Path(base / f"{i}.png")
        with ThreadPoolExecutor(max_workers=threads) as pool:   # writing the inputs is set-up, not measured
            list(pool.map(lambda i: Image.fromarray(labs[i % 4]).save(lookup / f"{i}.png", compress_level=1), range(n)))
        png_mb = sum(f.stat().st_size for f in lookup.iterdir()) / 1e6
        # decode alone: what the loader threads can deliver
        _, dt_dec = timed(lambda: list(ThreadPoolExecutor(max_workers=threads).map(
            lambda i: np.asarray(Image.open(lookup / f"{i}.png")), range(n))))
        seg = SegmentorPhotogrammetryCameraSet(cams, LookUpSegmentor(base, lookup, num_classes=C))
        labels_in = {"png_MB_total": round(png_mb, 1), "png_decode_alone_views_per_s": round(n / dt_dec, 1)}
        for tag, scale in (("scale_1", 1.0), ("scale_0.25", 0.25)):
            mesh.aggregate_projected_images(seg, aggregate_img_scale=scale, loader_threads=threads)   # warm-up (scratch, pinned buffers)
            _, dt = timed(lambda: mesh.aggregate_projected_images(seg, aggregate_img_scale=scale, loader_threads=threads))
            labels_in[f"{tag}_views_per_s"] = round(n / dt, 1)
        # the opt-in decoded-input cache (aggregate_projected_images(decoded_cache=...)): the first pass decodes and stores
        # uncompressed .npy entries, later passes memory-map them
        cache_dir = Path(d) / "decoded_cache"
        for tag, scale in (("scale_1", 1.0), ("scale_0.25", 0.25)):
            _, dt_first = timed(lambda: mesh.aggregate_projected_images(seg, aggregate_img_scale=scale, loader_threads=threads,
                                                                         decoded_cache=cache_dir))
            _, dt_cached = timed(lambda: mesh.aggregate_projected_images(seg, aggregate_img_scale=scale, loader_threads=threads,
                                                                          decoded_cache=cache_dir))
            labels_in[f"{tag}_decoded_cache_first_pass_views_per_s"] = round(n / dt_first, 1)
            labels_in[f"{tag}_decoded_cache_later_pass_views_per_s"] = round(n / dt_cached, 1)
        labels_in["decoded_cache_MB"] = round(sum(f.stat().st_size for f in cache_dir.glob("*.npy")) / 1e6, 1)
        labels_in["binds"] = ("PNG decode on the host threads (zlib inflate, one file per thread): the aggregate rate follows the "
                              "decode-alone rate; the link and the kernels are 5-30x faster") \
            if labels_in["scale_1_views_per_s"] < 2.0 * labels_in["png_decode_alone_views_per_s"] else "link / kernels"
        out["aggregate_from_label_png_files"] = labels_in
    cams.image_folder = Path("/synthetic")
    for i, cam in enumerate(cams.cameras):
        cam.image_filename = Path(f"/synthetic/view_{i:04d}.png")
    renders = {}
    for fmt in ("tif", "npy"):
        with tempfile.TemporaryDirectory() as d:
            kw = dict(output_folder=d, apply_distortion=False, writer_threads=threads, save_as_npy=fmt == "npy")
            mesh.save_renders(cams[0:2 * threads], **kw)   # warm-up: the pinned ring (2 x writer_threads slots)
            _, dt = timed(lambda: mesh.save_renders(cams, **kw))
            size = sum(f.stat().st_size for f in Path(d).rglob("*") if f.is_file()) / 1e6
        renders[fmt] = {"views_per_s": round(n / dt, 1), "MB_written": round(size, 1)}
    renders["binds"] = {"tif": "zlib deflate on the writer threads (one image per thread)", "npy": "device -> host copies and the file system"}
    out["save_renders"] = renders
    out["leg_wall_s"] = round(time.perf_counter() - t_leg, 2)
    return out


if __name__ == "__main__":
    sys.exit(main())
