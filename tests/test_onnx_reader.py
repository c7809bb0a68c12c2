"""icl_model_load_onnx (LoadPretrainedModelONNX, embeddings.go:28-43): the dependency-free ONNX initializer reader.
The real resnet50-v1-7.onnx is absent (.MISSING_LARGE_BLOBS) and cannot be fetched, so the reader is exercised on
ONNX files written by tests/onnx_writer.py from the synthetic ICLW blob; on the GPU the model loaded from the .onnx
must embed bit-identically to the model loaded from the blob."""
import numpy as np
import pytest

from tests import onnx_writer as W


@pytest.fixture(scope="module")
def blob():
    from imageclust_amd import _lib

    return _lib.synthetic_blob(1)


@pytest.mark.gpu
@pytest.mark.parametrize("raw,trans_b", [(True, 1), (False, 0)])
def test_onnx_model_matches_blob_model(blob, tmp_path, raw, trans_b):
    from imageclust_amd import _lib

    path = str(tmp_path / "resnet50-v1-7.onnx")
    W.blob_to_onnx(blob, path, raw=raw, trans_b=trans_b)
    a, b = _lib.Context(0), _lib.Context(0)
    a.load_onnx(path)
    b.load_blob(blob)
    imgs = _lib.synth_images(20250217, 0, 3, _lib.SYNTH_STRUCTURED)
    for head in (_lib.HEAD_POOLED, _lib.HEAD_DENSE0):
        assert np.array_equal(a.embed_u8(imgs, head, _lib.PREC_FP32), b.embed_u8(imgs, head, _lib.PREC_FP32))
    a.close()
    b.close()


@pytest.mark.gpu
def test_onnx_errors(blob, tmp_path):
    from imageclust_amd import _lib

    c = _lib.Context(0)
    with pytest.raises(_lib.ICLError) as ei:
        c.load_onnx(str(tmp_path / "missing.onnx"))
    assert ei.value.code == _lib.ICL_ERR_IO and "failed to load ResNet50 ONNX model from" in str(ei.value)
    bad = tmp_path / "bad.onnx"
    bad.write_bytes(b"\x00\x01garbage")
    with pytest.raises(_lib.ICLError) as ei:
        c.load_onnx(str(bad))
    assert ei.value.code == _lib.ICL_ERR_IO
    # a graph that is not ResNet50-v1 (first conv has the wrong stride)
    path = str(tmp_path / "wrong.onnx")
    W.blob_to_onnx(blob, path)
    data = open(path, "rb").read()
    # flip the stem's strides attribute [2,2] -> [1,1]
    needle = b"\x0a\x07strides\x40\x02\x40\x02"
    assert needle in data
    open(path, "wb").write(data.replace(needle, b"\x0a\x07strides\x40\x01\x40\x01", 1))
    with pytest.raises(_lib.ICLError) as ei:
        c.load_onnx(path)
    assert "ResNet50-v1 expects" in str(ei.value)
    c.close()
