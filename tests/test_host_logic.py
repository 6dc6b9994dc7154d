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


@pytest.mark.parametrize("route", ["in-place", "get/set_state"])
@pytest.mark.parametrize("seed,n,k", [(0, 60108, 50000), (1, 10, 10), (2, 1, 1), (3, 2, 1), (4, 65536, 100),
                                      (5, 65537, 65537), (6, 100000, 50000), (7, 3248, 3248), (8, 624, 3),
                                      (9, 31, 31), (10, 32, 5), (11, 33, 33), (12, 64, 64), (13, 131_073, 9),
                                      (14, 300_001, 50_000)])
def test_legacy_choice_matches_numpy(lib, seed, n, k, route, monkeypatch):
    """gf_host_legacy_choice == np.random.choice(n, k, replace=False) on the global legacy generator (the draw of
    geoformer.py:575-577): same indices, same generator state afterwards, wherever the generator stands."""
    from geoformer_amd import pointops

    if route == "in-place":  # numpy's MT19937 state driven where it lives (no get_state / set_state round trip)
        assert pointops._legacy_state() is not None
    else:                    # ... and the portable route for a numpy whose state cannot be reached
        monkeypatch.setattr(pointops, "_legacy_state", lambda: None)

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


@pytest.mark.parametrize("seed,n,k,ahead", [(1, 60_000, 50_000, 90_000), (2, 60_000, 50_000, 30_000), (3, 5000, 5000, 100),
                                             (4, 70_001, 3, 200_000), (5, 1234, 1234, 1), (6, 624, 100, 0)])
def test_legacy_choice_with_words_drawn_ahead(lib, seed, n, k, ahead):
    """gf_host_legacy_prefetch: the generator's outputs drawn ahead of the draw (enough, too few -- the draw then runs on
    the generator itself --, none): same indices, same generator state as numpy's own draw, and a prefetch that no draw
    picks up (another state in between) changes nothing."""
    from geoformer_amd import pointops

    def prime():
        np.random.seed(seed)
        np.random.rand(seed * 131)

    prime()
    ref = np.random.choice(n, k, replace=False)
    ref_after = (np.random.randn(3), np.random.get_state())
    prime()
    pointops.legacy_prefetch(ahead)
    got = pointops.legacy_choice(n, k)
    got_after = (np.random.randn(3), np.random.get_state())
    assert (got == ref).all() and (ref_after[0] == got_after[0]).all()
    assert (ref_after[1][1] == got_after[1][1]).all() and ref_after[1][2:] == got_after[1][2:]
    # a stale prefetch: the generator moves on before the draw
    prime()
    pointops.legacy_prefetch(ahead)
    np.random.rand(5)
    ref2_state = np.random.get_state()
    ref2 = np.random.choice(n, k, replace=False)
    np.random.set_state(ref2_state)
    assert (pointops.legacy_choice(n, k) == ref2).all()


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


def test_dropout_keep_reference_is_the_hash_the_header_states():
    """pointops.dropout_keep_reference (what the GPU tests feed to the float64 references of the training transformer
    kernels) against a scalar transcription of the rule in include/geoformer_hip.h: kept iff (h >> 8) >= (uint32)(p 2^24),
    h = fmix32(fmix32(seed ^ (row * 64 + site)) + col * 0x9E3779B1), fmix32 = MurmurHash3's finaliser."""
    import numpy as np
    import torch

    from geoformer_amd import pointops

    M = 0xFFFFFFFF

    def fmix(x):
        x ^= x >> 16
        x = (x * 0x85EBCA6B) & M
        x ^= x >> 13
        x = (x * 0xC2B2AE35) & M
        return x ^ (x >> 16)

    def keep(seed, p, site, row, col):
        h = fmix((seed & M) ^ ((row * 64 + site) & M))
        h = fmix((h + ((col * 0x9E3779B1) & M)) & M)
        return (h >> 8) >= int(float(p) * 16777216.0)

    # MurmurHash3 fmix32 known answers (h = 1, 0xdeadbeef)
    assert fmix(1) == 0x514E28B7 and fmix(0) == 0
    rng = np.random.default_rng(3)
    for seed, p, site in ((1234567, 0.1, 0), (2 ** 31 - 2, 0.1, 13), (42, 0.5, 7)):
        rows = torch.from_numpy(rng.integers(0, 1 << 20, (5, 1)))
        cols = torch.from_numpy(rng.integers(0, 1 << 12, (1, 7)))
        got = pointops.dropout_keep_reference(seed, p, site, rows.expand(5, 7), cols.expand(5, 7))
        for i in range(5):
            for j in range(7):
                k = keep(seed, p, site, int(rows[i, 0]), int(cols[0, j]))
                assert (float(got[i, j]) != 0.0) == k
                if k:
                    assert abs(float(got[i, j]) - 1.0 / (1.0 - p)) < 1e-6
    big = pointops.dropout_keep_reference(7, 0.1, 3, torch.arange(4096).view(-1, 1).expand(4096, 64),
                                          torch.arange(64).view(1, -1).expand(4096, 64))
    assert abs(float((big == 0).float().mean()) - 0.1) < 0.005
    assert float(pointops.dropout_keep_reference(7, 0.0, 3, torch.arange(8), torch.arange(8)).min()) == 1.0


def test_host_wait_word_returns_the_stored_value_or_times_out(lib):
    """gf_host_wait_word (the host side of gf_fg_select's polled count): a word that already differs from `pending` comes
    back at once; one that never changes comes back as `pending` after the time-out; a word another thread stores to
    ends the wait."""
    import ctypes
    import threading
    import time

    w = (ctypes.c_int32 * 1)(42)
    assert lib.gf_host_wait_word(ctypes.addressof(w), -1, 1_000_000) == 42
    w[0] = -1
    t0 = time.perf_counter()
    assert lib.gf_host_wait_word(ctypes.addressof(w), -1, 20_000) == -1
    assert 0.015 < time.perf_counter() - t0 < 1.0

    def store():
        time.sleep(0.01)
        w[0] = 7

    th = threading.Thread(target=store)
    th.start()
    assert lib.gf_host_wait_word(ctypes.addressof(w), -1, 5_000_000) == 7
    th.join()
    assert lib.gf_host_wait_word(None, -1, 10) == -1
