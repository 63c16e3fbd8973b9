"""The tile kernel's `s_waitcnt vmcnt` instructions, counted in the ISA the build produces.

Loads and stores of a wave share ONE in-order memory counter on gfx950, and the compiler's bookkeeping of it across the loop
over the tiles of a chain is conservative: with a small change of the source (the form of the chunk requests, a register more
or less) it puts a wait for an already finished chunk load into the tile loop, where it waits for the previous tile's id
STORES instead -- 5 % of the kernel, invisible in any functional test (measured in round 3, geograster.hip k_raster_tile).
This test compiles the device code to assembly (no GPU needed) and holds the default kernels to the known-good numbers."""
import re
import subprocess

import pytest

from geograypher_amd import build as gbuild

# The kernels of the usual calls at 64x32 tiles (round 5) -> their number of `s_waitcnt vmcnt` instructions:
#   "ids"   k_raster_tile<6, 5, 256, false, 4, PAD, short, PLAIN, micro>: chains of 4 tiles -- ONE wait for all requests of the
#           chain before its first tile, and the waits for later chunks and their row counts inside a tile (2 per copy of the
#           tile code, 2 copies); none for the empty-tile path, none between or inside the tiles of the chain;
#   "fused" k_raster_tile_roll<6, 5, 256, true, PAD, short, false, 16, micro>: rolling chains of 16 tiles -- the counters of
#           the chain, the first tile's request, the chunk-ahead wait and the wait in front of the epilogue of ONE inlined tile
#           body (the request of the next tile stays in flight over the epilogue: that is the point of the kernel).
#   With micro lists (the MICRO builds, used only for meshes the learned table marks) a wave also reads its own micro chunks.
KNOWN_GOOD = {("ids", False, False): 5, ("ids", True, False): 5, ("fused", False, False): 5, ("fused", True, False): 5,
              ("ids", True, True): 9, ("fused", True, True): 9}


def _device_asm(src, out):
    flags = [f for f in gbuild.HIPCC_FLAGS if f not in ("-shared", "-fPIC")]
    cmd = [gbuild.hipcc_path(), *flags, "-S", "--cuda-device-only", f"-I{gbuild.INCLUDE}", f"-I{gbuild.CSRC}", "-o", str(out), str(src)]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr[-2000:]
    return out.read_text().splitlines()


@pytest.fixture(scope="module")
def device_asm(tmp_path_factory):
    """ISA of the tile kernel's translation unit (csrc/raster_tile.hip)."""
    return _device_asm(gbuild.SRC, tmp_path_factory.mktemp("isa") / "raster_tile.s")


def _kernel_body(lines, kind, short, micro):
    tail = "EEvN6grimpl7BinArgsENS1_9RasterOutE:"
    if kind == "ids":
        name = "_ZN12_GLOBAL__N_113k_raster_tileILi6ELi5ELi256ELb0ELi4ELi%dELb%dELb1ELb%dE" % (_lds_pad(short), int(short), int(micro))
    else:
        name = "_ZN12_GLOBAL__N_118k_raster_tile_rollILi6ELi5ELi256ELb1ELi%dELb%dELb0ELi%dELb%dE" % (
            _lds_pad(short), int(short), _roll_kt(), int(micro))
    start = [i for i, l in enumerate(lines) if l.startswith(name + tail)]
    assert len(start) == 1, f"kernel symbol not found: {name}"
    end = next(i for i in range(start[0], len(lines)) if lines[i].startswith(".Lfunc_end"))
    return lines[start[0]:end]


def _lds_pad(short=True):
    m = re.search(r"#define GR_LDS_PAD%s (\d+)" % ("" if short else "48"), gbuild.SRC.read_text())
    return int(m.group(1))


def _roll_kt():
    return int(re.search(r"#define GR_ROLL_KT (\d+)", gbuild.SRC.read_text()).group(1))


@pytest.mark.parametrize("kind,short,micro", sorted(KNOWN_GOOD))
def test_tile_kernel_has_no_new_memory_waits(device_asm, kind, short, micro):
    body = _kernel_body(device_asm, kind, short, micro)
    waits = [(i, l.strip()) for i, l in enumerate(body) if "s_waitcnt" in l and "vmcnt" in l]
    assert len(waits) <= KNOWN_GOOD[(kind, short, micro)], (
        f"{kind} kernel (short={short}, micro={micro}) has {len(waits)} vmcnt waits, known-good build has "
        f"{KNOWN_GOOD[(kind, short, micro)]}: {waits} -- check that none of them sits in the loop over the chain's tiles")
    if kind == "ids":
        # the chain's four chunk requests are issued back to back: no wait between the first and the last of them
        loads = [i for i, l in enumerate(body) if "global_load_dwordx4" in l]
        assert len(loads) >= 4
        assert waits[0][0] > loads[3], (waits[:3], loads[:5])


@pytest.mark.parametrize("unit", [p.name for p in gbuild.SOURCES])
def test_no_kernel_spills(unit, tmp_path):
    """No kernel of this library uses scratch memory (rocPRIM's radix sort, instantiated for the sparse path and the mesh
    upload, does: only kernels of the library's own anonymous namespaces are held to it)."""
    lines = _device_asm(gbuild.CSRC / unit, tmp_path / (unit + ".s"))
    cur, spills, own = None, [], 0
    for l in lines:
        if l.startswith("_Z") and ":" in l:
            cur = l.split(":")[0]
        if "ScratchSize:" in l and cur and cur.startswith("_ZN12_GLOBAL__N_1"):
            own += 1
            if not l.strip().endswith(" 0"):
                spills.append((cur, l.strip()))
    assert not spills, spills[:5]
    assert own > 0 or unit == "geograster.hip"  # the core unit launches no kernel of its own
