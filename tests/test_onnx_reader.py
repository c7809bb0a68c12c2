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


# ---- independent check of the reader (no GPU): Google's protobuf runtime parses the same file --------------------------
def _tensor_values(t):
    import numpy as np

    if t.raw_data:
        a = np.frombuffer(t.raw_data, np.float32)
    else:
        a = np.array(t.float_data, np.float32)
    return a.reshape(tuple(t.dims))


def _expected_blob_payload(model):
    """What LoadPretrainedModelONNX must extract, found by an INDEPENDENT walk of the graph Google-protobuf parsed: follow the
    data edges from the graph input; per convolution W, [bias], gamma, beta, mean, var in block order c1, c2, c3, downsample."""
    import numpy as np

    g = model.graph
    init = {t.name: t for t in g.initializer}
    produced = {o for n in g.node for o in n.output}
    readers = {}
    for n in g.node:
        for k, name in enumerate(n.input):
            if name not in init and (k == 0 or n.op_type == "Add"):
                readers.setdefault(name, []).append(n)

    def nxt(t, op):
        r = [n for n in readers.get(t, []) if n.op_type == op]
        assert len(r) == 1, (t, op, [n.op_type for n in readers.get(t, [])])
        return r[0]

    out, bias_flags = [], []

    def emit(conv):
        bn = nxt(conv.output[0], "BatchNormalization")
        out.append(_tensor_values(init[conv.input[1]]).ravel())
        has_b = len(conv.input) > 2 and conv.input[2] != ""
        bias_flags.append(has_b)
        if has_b:
            out.append(_tensor_values(init[conv.input[2]]).ravel())
        for k in range(1, 5):
            out.append(_tensor_values(init[bn.input[k]]).ravel())
        return bn

    stem = [n for n in g.node if n.op_type == "Conv" and n.input[0] not in produced]
    assert len(stem) == 1
    x = nxt(nxt(emit(stem[0]).output[0], "Relu").output[0], "MaxPool").output[0]
    for blk in range(16):
        convs = [n for n in readers[x] if n.op_type == "Conv"]
        couts = {n.name: init[n.input[1]].dims[0] for n in convs}
        c1 = min(convs, key=lambda n: couts[n.name])
        ds = [n for n in convs if n is not c1]
        assert len(ds) <= 1 and (not ds or couts[ds[0].name] == 4 * couts[c1.name])
        b1 = emit(c1)
        c2 = nxt(nxt(b1.output[0], "Relu").output[0], "Conv")
        b2 = emit(c2)
        c3 = nxt(nxt(b2.output[0], "Relu").output[0], "Conv")
        b3 = emit(c3)
        if ds:
            emit(ds[0])
        x = nxt(nxt(b3.output[0], "Add").output[0], "Relu").output[0]
    head = nxt(x, "GlobalAveragePool").output[0]
    flat = [n for n in readers[head] if n.op_type in ("Flatten", "Reshape")][0].output[0]
    gemm = nxt(flat, "Gemm")
    W_ = _tensor_values(init[gemm.input[1]])
    trans_b = [a.i for a in gemm.attribute if a.name == "transB"]
    out.append((W_ if (trans_b and trans_b[0] == 1) else W_.T).ravel())
    out.append(_tensor_values(init[gemm.input[2]]).ravel())
    return np.concatenate(out), bias_flags


@pytest.mark.parametrize("variant", ["zoo_order", "gluon_names", "downsample_first", "shuffled_everything"])
def test_reader_against_an_independent_protobuf_parser(blob, tmp_path, variant):
    """The engine's ONNX reader (its own wire-format walker + graph walk) must extract exactly the tensors that Google's
    protobuf runtime + an independent graph walk find in the same file -- by name and data flow, in any node order, with
    Gluon-style names, raw or float_data storage, transB 0/1, Flatten or Reshape in front of the Gemm."""
    from imageclust_amd import _lib
    from tests import onnx_proto as P

    kw = {"zoo_order": dict(),
          "gluon_names": dict(gluon_names=True, raw=False),
          "downsample_first": dict(ds_first=True, trans_b=0),
          "shuffled_everything": dict(gluon_names=True, ds_first=True, shuffle=np.random.default_rng(5), reshape_head=True, raw=False, trans_b=0)}[variant]
    path = str(tmp_path / "resnet50-v1-7.onnx")
    W.blob_to_onnx(blob, path, **kw)
    model = P.load(path)
    assert model.graph.name == "mxnet_converted_model" and model.opset_import[0].version == 7
    ops = [n.op_type for n in model.graph.node]
    assert ops.count("Conv") == 53 and ops.count("BatchNormalization") == 53 and ops.count("Gemm") == 1 and ops.count("Add") == 16
    want, bias_flags = _expected_blob_payload(model)
    got = _lib.onnx_to_blob(path)
    hdr, payload = got[:80], np.frombuffer(got[80:].tobytes(), np.float32)
    assert payload.shape == want.shape
    assert np.array_equal(payload.view(np.uint32), want.view(np.uint32))
    assert [bool(b) for b in hdr[16:16 + 53]] == bias_flags
    if variant == "gluon_names":
        names = {t.name for t in model.graph.initializer}
        assert "resnetv17_stage1_conv0_weight" in names and "resnetv17_stage4_batchnorm9_gamma" in names


def test_reader_rejects_graphs_that_are_not_resnet50(blob, tmp_path):
    from imageclust_amd import _lib
    from tests import onnx_proto as P

    path = str(tmp_path / "m.onnx")
    W.blob_to_onnx(blob, path)
    model = P.load(path)
    # (a) cut the identity edge of one block: its Add reads the block's own c1 output instead of the block input
    m2 = P.ModelProto()
    m2.CopyFrom(model)
    adds = [n for n in m2.graph.node if n.op_type == "Add"]
    relus = {n.output[0]: n for n in m2.graph.node if n.op_type == "Relu"}
    victim = adds[1]  # block 1 of stage 1 has no downsample: input[1] is the block input
    victim.input[1] = [n for n in m2.graph.node if n.op_type == "Relu"][2].output[0]  # some other tensor than the block input
    bad = str(tmp_path / "cut.onnx")
    open(bad, "wb").write(m2.SerializeToString())
    with pytest.raises(_lib.ICLError, match="not a ResNet50-v1 graph"):
        _lib.onnx_to_blob(bad)
    # (b) a tensor whose dims promise more data than the file holds
    m3 = P.ModelProto()
    m3.CopyFrom(model)
    m3.graph.initializer[0].dims[0] = 1 << 40
    bad = str(tmp_path / "dims.onnx")
    open(bad, "wb").write(m3.SerializeToString())
    with pytest.raises(_lib.ICLError):
        _lib.onnx_to_blob(bad)
    # (c) a hostile graph whose block Add has ONE input (the bn3 output): rejected, not indexed out of bounds
    m4 = P.ModelProto()
    m4.CopyFrom(model)
    victim = [n for n in m4.graph.node if n.op_type == "Add"][1]
    bn3_out = victim.input[0] if victim.input[1] in relus or victim.input[1] == "x" else victim.input[1]
    bns = {n.output[0] for n in m4.graph.node if n.op_type == "BatchNormalization"}
    keep = [t for t in victim.input if t in bns][:1] or [bn3_out]
    del victim.input[:]
    victim.input.extend(keep)
    bad = str(tmp_path / "add1.onnx")
    open(bad, "wb").write(m4.SerializeToString())
    with pytest.raises(_lib.ICLError, match="not a ResNet50-v1 graph"):
        _lib.onnx_to_blob(bad)
    assert relus
