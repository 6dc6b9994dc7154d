"""Host-side checks that need no GPU: the C-ABI library loads, exports every entry point the header declares and the
ctypes table binds each of them; the native restatement of the reference's host RNG draw is bit-exact."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from geoformer_amd import _build, _lib

    if not os.path.exists(_lib.LIB_PATH):
        _build.build_hip()  # cross-compiles for gfx950 without a GPU
    return _lib.load()


def _declared():
    """Every entry point declared in include/*.h (the product ABI and the dev-hook header)."""
    names = set()
    inc = os.path.join(ROOT, "include")
    for f in sorted(os.listdir(inc)):
        if f.endswith(".h"):
            text = re.sub(r"/\*.*?\*/", "", open(os.path.join(inc, f)).read(), flags=re.S)
            names |= set(re.findall(r"\b(gf_[a-z0-9_]+)\s*\(", text))
    return sorted(names)


def test_library_exports_every_declared_symbol(lib):
    from geoformer_amd import _lib

    names = _declared()
    assert len(names) > 60
    raw = ctypes.CDLL(_lib.LIB_PATH)
    missing = [n for n in names if not hasattr(raw, n)]
    assert not missing, f"declared in include/*.h but not exported: {missing}"
    # every declared entry point has argument/return types bound on the handle the package uses
    unbound = [n for n in names if getattr(lib, n).argtypes is None]
    assert not unbound, f"no ctypes signature in geoformer_amd/_lib.py: {unbound}"


def test_abi_version_and_error_channel(lib):
    text = open(os.path.join(ROOT, "include", "geoformer_hip.h")).read()
    assert lib.gf_abi_version() == int(re.search(r"#define GF_ABI_VERSION (\d+)", text).group(1))
    # argument checking happens before any device call: usable without a GPU
    assert lib.gf_host_legacy_choice(None, None, 10, 5, None) < 0
    assert b"gf_host_legacy_choice" in lib.gf_last_error()


@pytest.mark.parametrize("seed,n,k", [(0, 60108, 50000), (1, 10, 10), (2, 1, 1), (3, 2, 1), (4, 65536, 100),
                                      (5, 65537, 65537), (6, 100000, 50000), (7, 3248, 3248), (8, 624, 3)])
def test_legacy_choice_matches_numpy(lib, seed, n, k):
    """gf_host_legacy_choice == np.random.choice(n, k, replace=False) on the global legacy generator (the draw of
    geoformer.py:575-577): same indices, same generator state afterwards, wherever the generator stands."""
    from geoformer_amd import pointops

    def prime():
        np.random.seed(seed)
        np.random.rand(seed * 97)  # move the position inside the 624-word block
        np.random.randn(seed % 2)  # and, for odd seeds, leave a cached gaussian behind

    prime()
    ref = np.random.choice(n, k, replace=False)
    ref_after = (np.random.randn(3), np.random.get_state())
    prime()
    got = pointops.legacy_choice(n, k)
    got_after = (np.random.randn(3), np.random.get_state())
    assert got.dtype == ref.dtype and (got == ref).all()
    assert (ref_after[0] == got_after[0]).all()
    assert (ref_after[1][1] == got_after[1][1]).all() and ref_after[1][2:] == got_after[1][2:]


def test_legacy_choice_sequence(lib):
    from geoformer_amd import pointops

    np.random.seed(11)
    ref = [np.random.choice(5000 + i, 2048, replace=False) for i in range(5)]
    np.random.seed(11)
    got = [pointops.legacy_choice(5000 + i, 2048) for i in range(5)]
    assert all((a == b).all() for a, b in zip(ref, got))
    with pytest.raises(ValueError):
        np.random.choice(5, 6, replace=False)
    with pytest.raises(ValueError):
        pointops.legacy_choice(5, 6)
