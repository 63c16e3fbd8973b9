"""bench.run's DISTRIBUTED control flow, end to end, before the first real multi-GPU run: two gloo ranks on the CPU drive
the very function the driver launches -- every `all_reduce` call site (ranks_seen, the MAX-over-ranks reductions of the
timed windows, the packed vote reduce of the aggregate / c4 / c5 legs), the barriers, the rank-0-only JSON line, the exit
code of a line that would claim more GPUs than answered -- with a stand-in for the HIP backend (the CPU oracle dressed as
`HipRaster`, tests/oracle_backend.py) and a toy workload.  What it cannot cover is RCCL itself."""
import io
import json
import os
import socket
import sys
import time
from contextlib import redirect_stdout

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.oracle_backend import OracleBackend


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _BenchStub(OracleBackend):
    """OracleBackend + the bookkeeping surface bench.py reads: profiling spans, status, retry counters."""

    def __init__(self):
        super().__init__()
        self.last_retries = 0
        self.last_stats = {}
        self._prof = None

    def set_profiling(self, enabled):
        self._prof = dict(setup_ms=0.0, scan_ms=0.0, fill_ms=0.0, raster_ms=0.0, project_ms=0.0, vote_ms=0.0, gather_ms=0.0,
                          raster_launches=0, views=0) if enabled else None

    def stage_times(self):
        out = dict(self._prof)
        self.set_profiling(True)
        return out

    def _note(self, n_views, seconds, fused=False):
        self.last_stats = {"records": 10 * n_views, "entries": 20 * n_views, "max_entries": 5, "entry_cap": 512, "overflow": 0,
                           "views_done": n_views}
        if self._prof is not None:
            self._prof["setup_ms"] += 250.0 * seconds
            self._prof["raster_ms"] += 700.0 * seconds
            self._prof["vote_ms"] += 50.0 * seconds if fused else 0.0
            self._prof["raster_launches"] += 1
            self._prof["views"] += n_views

    def raster_face_ids(self, cams, h, w, out=None, want_depth=False, check=True):
        t0 = time.perf_counter()
        ids = super().raster_face_ids(np.asarray(cams), h, w)
        if out is not None:
            out.copy_(ids)
        self._note(ids.shape[0], time.perf_counter() - t0)
        if not check:
            self.last_stats = {"unchecked": True}
        return out if out is not None else ids

    def raster_project_labels(self, cams, labels, C, votes, counts, ids_out=None, neg1_is_last_face=True, check=True):
        t0 = time.perf_counter()
        super().raster_project_labels(np.asarray(cams), np.asarray(labels), C, votes, counts, neg1_is_last_face=neg1_is_last_face)
        self._note(int(np.asarray(cams).shape[0]), time.perf_counter() - t0, fused=True)

    def raster_status(self):
        return {"records": 10, "entries": 20, "max_entries": 5, "entry_cap": 512, "overflow": 0, "views_done": 1}


class _StubRig:
    dist_backend = "gloo"
    overrides = {}

    def __init__(self):
        import bench

        kw = dict(n_side=24, extent=400.0, H=48, W=64, f=48.0, views_per_rank=3, c3_views=6, c4_views_per_rank=4, n_classes=4,
                  c5_n_side=30, c5_extent=800.0, c5_H=40, c5_W=60, c5_f=45.0, c5_views_total=8, c5_views_per_rank=4,
                  c5_classes=10, c5_raster_views=2, min_leg_s=0.0)
        kw.update(self.overrides)
        self.workload = bench.Workload(**kw)

    def device(self, local_rank):
        return torch.device("cpu")

    def init_process_group(self, dev):
        dist.init_process_group(self.dist_backend)

    def synchronize(self, dev):
        pass

    def make_raster(self, local_rank):
        return _BenchStub()

    def checker(self):
        from oracle import oracle_c

        return oracle_c

    def side_legs(self):
        return False


def _rank(rank, world, port, out_dir, gpus_flag, overrides=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank))
    sys.modules.pop("bench", None)
    import bench

    _StubRig.overrides = dict(overrides or {})

    args = bench.parse_args(["--gpus", str(gpus_flag), "--steps", "2", "--warmup", "1", "--windows", "2", "--min-timed-s", "0"])
    buf = io.StringIO()
    with redirect_stdout(buf):
        rc = bench.run(args, _StubRig())
    with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as fh:
        json.dump({"rc": rc, "stdout": buf.getvalue()}, fh)


def _run(tmp_path, world, gpus_flag, overrides=None):
    mp.spawn(_rank, args=(world, _free_port(), str(tmp_path), gpus_flag, overrides), nprocs=world, join=True)
    return [json.load(open(tmp_path / f"rank{r}.json")) for r in range(world)]


@pytest.mark.timeout(600)
def test_two_ranks_run_the_whole_bench_and_rank0_prints_one_line(tmp_path):
    res = _run(tmp_path, 2, 2)
    assert [r["rc"] for r in res] == [0, 0]
    assert res[1]["stdout"].strip() == ""                      # only rank 0 speaks
    lines = [l for l in res[0]["stdout"].splitlines() if l.strip()]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["scaling"] == "weak" and line["steps"] == 2
    assert line["ms_per_step_windows"]["n"] == 2 and line["timed_gpu_s"] > 0 and line["value"] > 0
    assert line["config"]["parallelism"].endswith("dp2") and line["config"]["views_per_gpu_per_step"] == 3
    # whole-job aggregate: both ranks' views in the numerator
    assert abs(line["views_per_s"] - 2 * 3 * 2 / (line["ms_per_step"] * 2e-3)) / line["views_per_s"] < 1e-3
    for key in ("roofline", "rooflines", "aggregate", "c4", "c5"):
        assert line[key] is not None, key
    assert line["cpu_baseline"] is None and line["workload_2"] is None and line["api"] is None  # N == 1 legs
    F = line["config"]["faces"]
    assert line["c4"]["all_reduce_bytes"] == F * 5 * 4 and line["c4"]["views_per_gpu"] == 4
    ring = line["c4"]["all_reduce_algorithmic_ms"]
    assert ring["bytes_on_the_wire_per_gpu"] == F * 5 * 4  # 2 (N - 1) / N = 1 at N = 2
    assert line["c4"]["face_observations_after_reduce"] > 0 and line["aggregate"]["faces_observed"] > 0
    assert line["c5"]["views_per_gpu"] == 4 and line["c5"]["all_reduce_bytes"] > 0
    assert set(line["rooflines"]) == {"k_setup_cull", "k_raster_tile_fused", "k_vote_labels"}
    for r in [line["roofline"], *line["rooflines"].values()]:
        assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["frac"] == pytest.approx(r["achieved"] / r["peak"], abs=1e-4)


@pytest.mark.timeout(600)
def test_one_rank_equals_the_sum_of_its_views(tmp_path):
    """World size 1 through the same entry point: the all-reduce sites are no-ops, the observation counts of the sharded run
    above can only be larger (twice the views)."""
    res = _run(tmp_path, 1, 1)
    assert res[0]["rc"] == 0
    line = json.loads(res[0]["stdout"].strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["ranks_seen"] == 1 and line["c4"]["all_reduce_algorithmic_ms"]["bytes_on_the_wire_per_gpu"] == 0
    assert line["cpu_baseline"] is not None and line["cpu_baseline"]["kind"] == "port"   # the CPU leg runs at N == 1
    assert "equal the CPU oracle's: True" in line["aggregate"]["oracle_check"]


@pytest.mark.timeout(600)
def test_a_line_that_claims_more_gpus_than_answered_is_refused(tmp_path):
    res = _run(tmp_path, 2, 3)
    assert [r["rc"] for r in res] == [3, 3] and all(r["stdout"].strip() == "" for r in res)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world", [2, 3])
def test_a_rank_without_views_still_joins_every_collective(tmp_path, world):
    """Fewer views than ranks in the c4 and c5 legs (one view each): the ranks whose shard is empty must still walk through
    every barrier, every MAX reduction and the packed vote reduce -- a rank that skipped one would hang the job (here: the
    test would time out) -- and the reduced votes are those of the one view."""
    res = _run(tmp_path, world, world, overrides=dict(c4_views_total=1, c5_views_total=1))
    assert [r["rc"] for r in res] == [0] * world
    line = json.loads(res[0]["stdout"].strip().splitlines()[-1])
    assert line["n_gpus"] == world and line["ranks_seen"] == world
    assert [r["rank"] for r in line["ranks"]] == list(range(world))        # the rank -> device map of the line
    assert line["c4"]["views_per_gpu"] == 1 and line["c5"]["views_per_gpu"] == 1      # rank 0 holds the one view
    assert line["c4"]["face_observations_after_reduce"] > 0 and line["c5"]["face_observations_after_reduce"] > 0
    assert line["c4"]["reps"] >= 3 and line["c5"]["aggregate_reps"] >= 2
