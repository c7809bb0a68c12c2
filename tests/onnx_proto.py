"""Test helper: the subset of onnx.proto (ONNX IR, public spec) needed to read a model's graph, built at run time as
google.protobuf descriptors -- the `onnx` package is not installed, the protobuf RUNTIME is.  This gives the test-suite a
parser for .onnx files that shares no code with the engine's reader (imageclust_amd/csrc/onnx_reader.hip) nor with the
test writer (tests/onnx_writer.py): Google's protobuf implementation decodes the bytes."""
from google.protobuf import descriptor_pb2, descriptor_pool, message_factory

_T = descriptor_pb2.FieldDescriptorProto


def _build():
    f = descriptor_pb2.FileDescriptorProto()
    f.name = "onnx_subset.proto"
    f.package = "onnx"
    f.syntax = "proto2"

    def msg(name, fields):
        m = f.message_type.add()
        m.name = name
        for (fname, num, typ, label, tname) in fields:
            fd = m.field.add()
            fd.name, fd.number, fd.type, fd.label = fname, num, typ, label
            if tname:
                fd.type_name = ".onnx." + tname
        return m

    O, R = _T.LABEL_OPTIONAL, _T.LABEL_REPEATED
    msg("TensorProto", [("dims", 1, _T.TYPE_INT64, R, None), ("data_type", 2, _T.TYPE_INT32, O, None),
                        ("float_data", 4, _T.TYPE_FLOAT, R, None), ("int32_data", 5, _T.TYPE_INT32, R, None),
                        ("int64_data", 7, _T.TYPE_INT64, R, None), ("name", 8, _T.TYPE_STRING, O, None),
                        ("raw_data", 9, _T.TYPE_BYTES, O, None), ("doc_string", 12, _T.TYPE_STRING, O, None)])
    msg("AttributeProto", [("name", 1, _T.TYPE_STRING, O, None), ("f", 2, _T.TYPE_FLOAT, O, None), ("i", 3, _T.TYPE_INT64, O, None),
                           ("s", 4, _T.TYPE_BYTES, O, None), ("t", 5, _T.TYPE_MESSAGE, O, "TensorProto"),
                           ("floats", 7, _T.TYPE_FLOAT, R, None), ("ints", 8, _T.TYPE_INT64, R, None), ("strings", 9, _T.TYPE_BYTES, R, None),
                           ("type", 20, _T.TYPE_INT32, O, None)])
    msg("NodeProto", [("input", 1, _T.TYPE_STRING, R, None), ("output", 2, _T.TYPE_STRING, R, None), ("name", 3, _T.TYPE_STRING, O, None),
                      ("op_type", 4, _T.TYPE_STRING, O, None), ("attribute", 5, _T.TYPE_MESSAGE, R, "AttributeProto"),
                      ("doc_string", 6, _T.TYPE_STRING, O, None), ("domain", 7, _T.TYPE_STRING, O, None)])
    msg("GraphProto", [("node", 1, _T.TYPE_MESSAGE, R, "NodeProto"), ("name", 2, _T.TYPE_STRING, O, None),
                       ("initializer", 5, _T.TYPE_MESSAGE, R, "TensorProto"), ("doc_string", 10, _T.TYPE_STRING, O, None)])
    msg("OperatorSetIdProto", [("domain", 1, _T.TYPE_STRING, O, None), ("version", 2, _T.TYPE_INT64, O, None)])
    msg("ModelProto", [("ir_version", 1, _T.TYPE_INT64, O, None), ("producer_name", 2, _T.TYPE_STRING, O, None),
                       ("producer_version", 3, _T.TYPE_STRING, O, None), ("domain", 4, _T.TYPE_STRING, O, None),
                       ("model_version", 5, _T.TYPE_INT64, O, None), ("doc_string", 6, _T.TYPE_STRING, O, None),
                       ("graph", 7, _T.TYPE_MESSAGE, O, "GraphProto"), ("opset_import", 8, _T.TYPE_MESSAGE, R, "OperatorSetIdProto")])
    pool = descriptor_pool.DescriptorPool()
    pool.Add(f)
    desc = pool.FindMessageTypeByName("onnx.ModelProto")
    if hasattr(message_factory, "GetMessageClass"):
        return message_factory.GetMessageClass(desc)
    return message_factory.MessageFactory(pool).GetPrototype(desc)


ModelProto = _build()


def load(path):
    m = ModelProto()
    with open(path, "rb") as fh:
        m.ParseFromString(fh.read())
    return m
