"""the CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer.  `make -C oracle asan` builds
oracle_raster.c and oracle_envelope.c with -fsanitize=address,undefined into a stand-alone driver (oracle/asan_driver.c)
that runs the edge-case suite -- 1x1 to ragged images, faces behind / across the near plane (the clipper's polygon
buffers), zero-area, coincident, NaN / inf and astronomically large faces, a camera with near <= 0, labels >= C, the -1
aliasing of the projection stage -- and checks spec == fast on every scene.  GPU sanitizers are not available on the pool;
the host-side C that the parity tests trust is checked here."""
import subprocess
from pathlib import Path

ORACLE = Path(__file__).resolve().parents[1] / "oracle"


def test_oracle_runs_clean_under_asan_and_ubsan():
    build = subprocess.run(["make", "-C", str(ORACLE), "-s", "asan"], capture_output=True, text=True)
    assert build.returncode == 0, build.stderr
    run = subprocess.run([str(ORACLE / "_build" / "oracle_asan")], capture_output=True, text=True, timeout=600,
                         env={"ASAN_OPTIONS": "detect_leaks=1:abort_on_error=0", "UBSAN_OPTIONS": "print_stacktrace=1"})
    assert run.returncode == 0, run.stdout + run.stderr
    assert run.stdout.startswith("ok:") and "runtime error" not in run.stderr and "AddressSanitizer" not in run.stderr
