"""CPU-side checks of the drop-in boundary: the C-ABI library loads without a GPU, exports every symbol that
include/imageclust.h declares, the ctypes table covers all of them, and GPU-less calls fail loudly."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "imageclust.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(icl_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported_and_bound():
    from imageclust_amd import _lib

    L = _lib.load()
    names = declared_symbols()
    assert len(names) >= 30
    bound = {s[0] for s in _lib.SYMBOLS}
    raw = ctypes.CDLL(_lib.SO_PATH)
    for n in names:
        assert hasattr(raw, n), "library does not export %s" % n
        assert n in bound, "ctypes table does not bind %s" % n
    assert bound <= set(names), "ctypes binds symbols the header does not declare: %s" % (bound - set(names))
    assert L.icl_version().startswith(b"imageclust_hip")


def test_no_cpu_fallback_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from imageclust_amd import _lib, clustering

    with pytest.raises(_lib.ICLError) as ei:
        _lib.Context(0)
    assert ei.value.code == _lib.ICL_ERR_HIP and "no CPU fallback" in str(ei.value)
    clustering.set_default_context(None)
    with pytest.raises(_lib.ICLError):
        clustering.PerformClusteringWithConstraints([[0.0], [1.0]], ["a", "b"], 1, 2)


def test_host_only_entry_points():
    from imageclust_amd import _lib, clustering

    assert _lib.calc_optimal_clusters(64, 3, 6) == (16, None)
    assert _lib.calc_optimal_clusters(10, 4, 4)[1] == _lib.ICL_ERR_CONSTRAINT
    assert clustering.CalculateOptimalClusters(2, 3, 5) == (0, "total items (2) less than minimum cluster size (3)")
    a = _lib.synth_images(20250217, 5, 2, _lib.SYNTH_NOISE)
    b = _lib.synth_images(20250217, 6, 1, _lib.SYNTH_NOISE)
    assert a.shape == (2, 224, 224, 3) and (a[1] == b[0]).all() and a.std() > 50
    s = _lib.synth_images(20250217, 1003, 1, _lib.SYNTH_STRUCTURED)
    s2 = _lib.synth_images(20250217, 3, 1, _lib.SYNTH_STRUCTURED)
    assert abs(s.astype(int) - s2.astype(int)).max() <= 16  # same class (n mod 1000), different noise
