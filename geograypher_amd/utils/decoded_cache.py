"""Decoded-input cache: what a pass over a camera set pays per view on the HOST is the decode of its files -- PNG / JPEG inflate,
211 label views/s and 25 photo views/s on a 16-core quota against 50 000 views/s in the kernels (bench `io`) -- and the reference
pays it again in every pass too.  It caches the stage it finds slow on disk (`save_to_cache` / `cache_folder` of pix2face,
meshes.py:1759-1770, 1838-1840, under constants.py:18 CACHE_FOLDER); this module does the same for the stage that is slow HERE:
a decoded image is kept as an uncompressed `.npy` under `<cache folder>/decoded/`, keyed by the file's path, modification time
and size (plus a tag: the scale a label image was resized to), and later passes memory-map it -- reads at page-cache / disk
rate instead of inflate rate.  Opt-in (`decoded_cache=` of LookUpSegmentor, PhotogrammetryCameraSet and
aggregate_projected_images); without it nothing is written and behaviour is unchanged.  A file that changes gets a new key: its
old entry is never read again (and is removed when the new one is written)."""
from __future__ import annotations

import hashlib
import os
import tempfile
from pathlib import Path
from typing import Callable, Optional, Union

import numpy as np

from geograypher_amd.constants import CACHE_FOLDER

CacheSpec = Union[None, bool, str, os.PathLike]


def resolve_folder(spec: CacheSpec) -> Optional[Path]:
    """None / False: no cache.  True: `CACHE_FOLDER/decoded` (the reference's cache root).  A path: that folder."""
    if spec is None or spec is False:
        return None
    return Path(CACHE_FOLDER, "decoded") if spec is True else Path(spec)


def _entry_names(path: Path, tag: str):
    """(name of the entry for the file as it is now, prefix shared by the entries of every version of this file + tag)"""
    st = os.stat(path)
    ident = hashlib.sha256(f"{Path(path).resolve()}|{tag}".encode()).hexdigest()[:24]
    version = hashlib.sha256(f"{st.st_mtime_ns}|{st.st_size}".encode()).hexdigest()[:16]
    return f"{ident}-{version}.npy", f"{ident}-"


def cached_decode(path, folder: CacheSpec, decode: Callable[[], np.ndarray], tag: str = "") -> np.ndarray:
    """The array `decode()` returns for the file at `path`, from the cache when an entry for the file's present (mtime, size)
    exists -- a read-only memory map --, decoded and stored otherwise.  `folder` None: just `decode()`."""
    folder = resolve_folder(folder)
    if folder is None:
        return decode()
    name, prefix = _entry_names(Path(path), tag)
    entry = folder / name
    try:
        return np.load(entry, mmap_mode="r", allow_pickle=False)
    except (FileNotFoundError, ValueError, OSError):
        pass
    array = np.ascontiguousarray(decode())
    try:
        folder.mkdir(parents=True, exist_ok=True)
        for old in folder.glob(prefix + "*.npy"):   # entries of earlier versions of the file
            old.unlink(missing_ok=True)
        fd, tmp = tempfile.mkstemp(dir=folder, suffix=".tmp")
        with os.fdopen(fd, "wb") as f:
            np.save(f, array, allow_pickle=False)
        os.replace(tmp, entry)                       # atomic: a concurrent reader sees the old state or the whole file
    except OSError:
        pass                                         # a cache that cannot be written is not an error
    return array
