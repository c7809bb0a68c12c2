"""Test helper: writes a minimal ONNX ModelProto (protobuf wire format by hand -- the onnx package is not installed)
holding a ResNet50-v1 graph with the tensors of an ICLW blob, laid out like the model-zoo export: Conv ->
BatchNormalization -> Relu chains, MaxPool, Add, GlobalAveragePool, Flatten, Gemm.  Only used to exercise
icl_model_load_onnx (imageclust_amd/csrc/onnx_reader.hip)."""
import struct

import numpy as np


def _varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _ld(num, payload):  # length-delimited field
    return _varint((num << 3) | 2) + _varint(len(payload)) + payload


def _vi(num, v):
    return _varint((num << 3) | 0) + _varint(v)


def tensor(name, arr, raw=True, packed_dims=False):
    arr = np.ascontiguousarray(arr, np.float32)
    b = b""
    if packed_dims:
        b += _ld(1, b"".join(_varint(d) for d in arr.shape))
    else:
        for d in arr.shape:
            b += _vi(1, d)
    b += _vi(2, 1)  # FLOAT
    b += _ld(9, arr.tobytes()) if raw else _ld(4, arr.tobytes())
    b += _ld(8, name.encode())
    return b


def attr_ints(name, vals):
    return _ld(1, name.encode()) + b"".join(_vi(8, v) for v in vals) + _vi(20, 7)


def attr_int(name, v):
    return _ld(1, name.encode()) + _vi(3, v) + _vi(20, 2)


def attr_float(name, v):
    return _ld(1, name.encode()) + _varint((2 << 3) | 5) + struct.pack("<f", v) + _vi(20, 1)


def node(op, ins, outs, attrs=()):
    b = b"".join(_ld(1, i.encode()) for i in ins) + b"".join(_ld(2, o.encode()) for o in outs)
    b += _ld(3, (op + "_" + outs[0]).encode()) + _ld(4, op.encode())
    for a in attrs:
        b += _ld(5, a)
    return b


def _fields(buf):
    """(field number, wire type, value bytes / int) of one message level -- just enough to re-order NodeProtos."""
    i = 0
    while i < len(buf):
        key = 0
        sh = 0
        while True:
            b = buf[i]
            i += 1
            key |= (b & 0x7F) << sh
            sh += 7
            if not b & 0x80:
                break
        num, wt = key >> 3, key & 7
        if wt == 2:
            ln = 0
            sh = 0
            while True:
                b = buf[i]
                i += 1
                ln |= (b & 0x7F) << sh
                sh += 7
                if not b & 0x80:
                    break
            yield num, wt, bytes(buf[i:i + ln])
            i += ln
        elif wt == 0:
            while buf[i] & 0x80:
                i += 1
            i += 1
            yield num, wt, None
        elif wt == 5:
            i += 4
            yield num, wt, None
        else:
            raise ValueError("wire type %d" % wt)


def _random_topological_order(nodes, rng):
    ins, outs = [], []
    for nb in nodes:
        f = list(_fields(nb))
        ins.append([v.decode() for n, w, v in f if n == 1])
        outs.append([v.decode() for n, w, v in f if n == 2])
    produced_by = {o: k for k, os_ in enumerate(outs) for o in os_}
    deps = [{produced_by[i] for i in ins[k] if i in produced_by} for k in range(len(nodes))]
    done, order = set(), []
    while len(order) < len(nodes):
        ready = [k for k in range(len(nodes)) if k not in done and deps[k] <= done]
        k = ready[int(rng.integers(len(ready)))]
        done.add(k)
        order.append(nodes[k])
    return order


def topology():
    t = [(3, 64, 7, 2, 3, 0, 0)]
    cin = 64
    for s, nb in enumerate([3, 4, 6, 3]):
        cout, mid = 256 << s, (256 << s) // 4
        for b in range(nb):
            stride = 2 if (b == 0 and s > 0) else 1
            t += [(cin, mid, 1, stride, 0, 1, b), (mid, mid, 3, 1, 1, 2, b), (mid, cout, 1, 1, 0, 3, b)]
            if b == 0:
                t.append((cin, cout, 1, stride, 0, 4, b))
            cin = cout
    return t


def blob_to_onnx(blob: np.ndarray, path: str, raw=True, trans_b=1, gluon_names=False, ds_first=False, shuffle=None, reshape_head=False):
    """gluon_names: Gluon-style initializer names (stageN_convK / stageN_batchnormK) instead of one running conv index;
    ds_first: every block lists its downsample Conv/BN BEFORE the main branch (a different, equally valid topological order);
    shuffle: a numpy Generator -- initializers in random order, nodes in a random TOPOLOGICAL order; reshape_head: Reshape
    instead of Flatten in front of the Gemm.  The loader must find every tensor by walking the graph, not by position."""
    hdr = blob[:80]
    eps = float(np.frombuffer(hdr[8:12].tobytes(), np.float32)[0])
    has_bias = hdr[16:80]
    p = np.frombuffer(blob[80:].tobytes(), np.float32)
    pos = 0
    inits, nodes = [], []

    def take(shape):
        nonlocal pos
        n = int(np.prod(shape))
        v = p[pos:pos + n].reshape(shape)
        pos += n
        return v

    gl = {"stage": 0, "conv": 0, "bn": 0}

    def conv_bn(i, x, cin, cout, k, s, pad, out_nodes=None):
        out_nodes = nodes if out_nodes is None else out_nodes
        if gluon_names:
            pre = "resnetv17_" if gl["stage"] == 0 else "resnetv17_stage%d_" % gl["stage"]
            names = [pre + "conv%d_weight" % gl["conv"], pre + "conv%d_bias" % gl["conv"]] + [pre + "batchnorm%d_%s" % (gl["bn"], q) for q in ("gamma", "beta", "running_mean", "running_var")]
            gl["conv"] += 1
            gl["bn"] += 1
        else:
            names = ["resnetv17_conv%d_%s" % (i, q) for q in ("weight", "bias", "gamma", "beta", "running_mean", "running_var")]
        inits.append(tensor(names[0], take((cout, cin, k, k)), raw, packed_dims=(i % 2 == 0)))
        ins = [x, names[0]]
        if has_bias[i]:
            inits.append(tensor(names[1], take((cout,)), raw))
            ins.append(names[1])
        for q in range(2, 6):
            inits.append(tensor(names[q], take((cout,)), raw))
        y, z = "conv%d_fwd" % i, "bn%d_fwd" % i
        out_nodes.append(node("Conv", ins, [y], [attr_ints("dilations", [1, 1]), attr_int("group", 1), attr_ints("kernel_shape", [k, k]),
                                                 attr_ints("pads", [pad] * 4), attr_ints("strides", [s, s])]))
        out_nodes.append(node("BatchNormalization", [y] + names[2:6], [z], [attr_float("epsilon", eps), attr_float("momentum", 0.9)]))
        return z

    def relu(x, tag):
        nodes.append(node("Relu", [x], [tag]))
        return tag

    topo = topology()
    x = relu(conv_bn(0, "data", *topo[0][:5]), "relu0")
    nodes.append(node("MaxPool", [x], ["pool0"], [attr_ints("kernel_shape", [3, 3]), attr_ints("pads", [1] * 4), attr_ints("strides", [2, 2])]))
    x = "pool0"
    i = 1
    while i < len(topo):
        has_ds = topo[i][6] == 0
        if has_ds:
            gl["stage"] += 1
            gl["conv"] = gl["bn"] = 0
        at = len(nodes)
        t = relu(conv_bn(i, x, *topo[i][:5]), "r%da" % i)
        t = relu(conv_bn(i + 1, t, *topo[i + 1][:5]), "r%db" % i)
        t = conv_bn(i + 2, t, *topo[i + 2][:5])
        if has_ds:  # the blob stores the downsample tensors AFTER the block's c3 either way; only the node order changes
            ds_nodes = []
            res = conv_bn(i + 3, x, *topo[i + 3][:5], out_nodes=ds_nodes)
            nodes[at:at] = ds_nodes if ds_first else []
            if not ds_first:
                nodes.extend(ds_nodes)
        else:
            res = x
        nodes.append(node("Add", [res, t] if ds_first else [t, res], ["add%d" % i]))
        x = relu("add%d" % i, "out%d" % i)
        i += 4 if has_ds else 3
    nodes.append(node("GlobalAveragePool", [x], ["pool1"]))
    nodes.append(node("Reshape" if reshape_head else "Flatten", ["pool1"], ["flat"]))
    W = take((1000, 2048))
    inits.append(tensor("resnetv17_dense0_weight", W if trans_b else np.ascontiguousarray(W.T), raw))
    inits.append(tensor("resnetv17_dense0_bias", take((1000,)), raw))
    nodes.append(node("Gemm", ["flat", "resnetv17_dense0_weight", "resnetv17_dense0_bias"], ["resnetv17_dense0_fwd"],
                      [attr_float("alpha", 1.0), attr_float("beta", 1.0), attr_int("transA", 0), attr_int("transB", trans_b)]))
    assert pos == len(p)
    if shuffle is not None:
        inits = [inits[k] for k in shuffle.permutation(len(inits))]
        nodes = _random_topological_order(nodes, shuffle)
    graph = b"".join(_ld(1, n) for n in nodes) + _ld(2, b"mxnet_converted_model") + b"".join(_ld(5, t) for t in inits)
    model = _vi(1, 3) + _ld(2, b"onnx-mxnet") + _ld(7, graph) + _ld(8, _ld(1, b"") + _vi(2, 7))
    with open(path, "wb") as f:
        f.write(model)
