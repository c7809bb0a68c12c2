"""CPU tests of the host-side glue mirrored from embeddings.go:166-236 (label vectors) and of the embedding cache
format.  No GPU, no oracle."""
import os

import numpy as np
import pytest

from imageclust_amd import embeddings as EM


def test_generate_label_vector_matches_reference_semantics():
    # embeddings.go:166-174: one-hot over the label set, unknown labels ignored, duplicates harmless
    ls = {"Shoe": 0, "Boot": 1, "Red": 2}
    v = EM.GenerateLabelVector(["Red", "Shoe", "Red", "Unknown"], ls)
    assert v.dtype == np.float32 and v.tolist() == [1.0, 0.0, 1.0]
    assert EM.GenerateLabelVector([], ls).tolist() == [0.0, 0.0, 0.0]
    assert EM.GenerateLabelVector(["x"], {}).shape == (0,)


def test_combine_embeddings_is_concatenation():
    e = np.arange(5, dtype=np.float32)
    c = EM.CombineEmbeddings(e, [0.0, 1.0])
    assert c.dtype == np.float32 and c.tolist() == [0, 1, 2, 3, 4, 0, 1]
    c[0] = 99  # embeddings.go:179 allocates: the input is not aliased
    assert e[0] == 0


def test_build_label_set_order_and_mapping(tmp_path):
    for name in ["b.jpg", "a.jpg", "c.jpg"]:
        (tmp_path / name).write_bytes(b"x")
    (tmp_path / "sub").mkdir()
    labels = {"a.jpg": ["Shoe", "Red"], "b.jpg": ["Red", "Boot"], "c.jpg": []}
    app = EM.AppContext(ImageDir=str(tmp_path))
    err = EM.BuildLabelSet(app, lambda p: labels[os.path.basename(p)])
    assert err is None
    # os.ReadDir order (sorted), first appearance defines the index (embeddings.go:214-221)
    assert app.LabelSet == {"Shoe": 0, "Red": 1, "Boot": 2}
    assert app.LabelsMapping == labels
    bad = EM.BuildLabelSet(EM.AppContext(ImageDir=str(tmp_path / "missing")), lambda p: [])
    assert bad is not None and "failed to read image directory" in str(bad)

    def boom(p):
        raise RuntimeError("throttled")

    e2 = EM.BuildLabelSet(EM.AppContext(ImageDir=str(tmp_path)), boom)
    assert e2 is not None and "failed to detect labels for image a.jpg" in str(e2)


@pytest.mark.parametrize("n,d", [(0, 8), (1, 1), (37, 2048)])
def test_embedding_cache_round_trip(tmp_path, n, d):
    rng = np.random.default_rng(n + d)
    E = rng.standard_normal((n, d)).astype(np.float32)
    ids = ["prod-%d" % i for i in range(n)]
    p = str(tmp_path / "e.icle")
    EM.SaveEmbeddings(p, ids, E)
    i2, E2 = EM.LoadEmbeddings(p)
    assert i2 == ids and E2.dtype == np.float32 and np.array_equal(E2.view(np.uint32), E.view(np.uint32))
    if n:
        i3, E3 = EM.LoadEmbeddings(p, mmap=True)
        assert i3 == ids and np.array_equal(np.asarray(E3), E)


def test_embedding_cache_rejects_bad_input(tmp_path):
    with pytest.raises(ValueError):
        EM.SaveEmbeddings(str(tmp_path / "x"), ["a"], np.zeros((2, 3), np.float32))
    with pytest.raises(ValueError):
        EM.SaveEmbeddings(str(tmp_path / "x"), ["a\nb"], np.zeros((1, 3), np.float32))
    (tmp_path / "junk").write_bytes(b"not a cache file at all......................")
    with pytest.raises(ValueError):
        EM.LoadEmbeddings(str(tmp_path / "junk"))
