"""Host-side mirror of /root/reference/internal/embeddings/embeddings.go (lines 17-163) over the HIP engine.

Same names and error behaviour as the Go package; gocv.Net / gocv.Mat become this module's opaque Net / Mat.
GenerateLabelVector / CombineEmbeddings / BuildLabelSet (embeddings.go:166-236) are Rekognition-side glue and are
out of scope.  No CPU fallback: every forward pass runs in libimageclust_hip.so.
"""
import threading
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np

from . import _lib


class Net:
    """Stands in for gocv.Net (embeddings.go:23): a GPU context with ResNet50-v1 weights resident in HBM."""

    def __init__(self, ctx: _lib.Context):
        self.ctx = ctx

    def Empty(self) -> bool:
        return self.ctx is None

    def Close(self):
        if self.ctx is not None:
            self.ctx.close()
            self.ctx = None


class Mat:
    """Stands in for gocv.Mat (embeddings.go:46): the preprocessed 224x224x3 u8 RGB image + its fp32 NCHW blob."""

    def __init__(self, rgb_u8: np.ndarray):
        self.rgb = np.ascontiguousarray(rgb_u8, np.uint8).reshape(224, 224, 3)

    def Size(self):
        return [1, 3, 224, 224]

    def Blob(self) -> np.ndarray:
        out = np.empty((1, 3, 224, 224), np.float32)
        rc = _lib.load().icl_preprocess_u8(self.rgb.ctypes.data, out.ctypes.data)
        if rc:
            raise _lib.ICLError(rc, "icl_preprocess_u8")
        return out


@dataclass
class AppContext:
    """embeddings.go:17-25"""
    ImageDir: str = ""
    CacheDir: str = ""
    LabelSet: Dict[str, int] = field(default_factory=dict)
    LabelsMapping: Dict[str, List[str]] = field(default_factory=dict)
    Net: Optional[Net] = None
    NetMutex: threading.Lock = field(default_factory=threading.Lock)
    Head: int = _lib.HEAD_DENSE0  # "resnetv17_dense0_fwd" (embeddings.go:140); HEAD_POOLED = the 2048-d vector


def LoadPretrainedModelONNX(modelPath: str, device: int = 0):
    """embeddings.go:28-43 -> (net, err).  `synthetic:<seed>` loads the seeded synthetic weights instead of a file;
    a `.iclw` path loads an ICLW blob (include/icl_model_format.h)."""
    try:
        ctx = _lib.Context(device)
        if modelPath.startswith("synthetic:"):
            ctx.load_synthetic(int(modelPath.split(":", 1)[1]))
        elif modelPath.endswith(".iclw"):
            with open(modelPath, "rb") as f:
                ctx.load_blob(f.read())
        else:
            ctx.load_onnx(modelPath)
        return Net(ctx), None
    except _lib.ICLError as e:
        return Net(None), "failed to load ResNet50 ONNX model from: %s (%s)" % (modelPath, e)
    except OSError as e:
        return Net(None), "failed to load ResNet50 ONNX model from: %s (%s)" % (modelPath, e)


def PreprocessImage(imagePath: str):
    """embeddings.go:46-116 -> (Mat, err): IMRead (baseline JPEG / binary PPM, decoded bit-identically to libjpeg-turbo) ->
    Resize 224x224 INTER_LINEAR -> RGB; Mat.Blob() gives the 1x3x224x224 fp32 blob scaled by 1/255."""
    try:
        rgb = _lib.load_image_224(imagePath)
    except _lib.ICLError as e:
        return None, str(e).split(": ", 1)[-1]
    return Mat(rgb), None


def GetImageEmbedding(appCtx: AppContext, imagePath: str):
    """embeddings.go:119-163 -> (embedding []float32, err)."""
    if appCtx.Net is None or appCtx.Net.Empty():
        return None, "failed to generate embedding for image: %s" % imagePath
    with appCtx.NetMutex:  # embeddings.go:133 (the engine is itself thread-safe; kept for signature parity)
        try:
            emb = appCtx.Net.ctx.embed_file(imagePath, appCtx.Head)
        except _lib.ICLError as e:
            return None, str(e)
    if emb.size == 0:
        return None, "embedding is empty for image: %s" % imagePath
    return emb, None


GenerateEmbedding = GetImageEmbedding  # the name BASELINE.json's north_star uses for the same function


def GetImageEmbeddingsBatch(appCtx: AppContext, images_u8: np.ndarray, prec: int = _lib.PREC_BF16) -> np.ndarray:
    """Batched fast path behind the same Net: n x 224 x 224 x 3 u8 RGB -> n x Head fp32."""
    return appCtx.Net.ctx.embed_u8(images_u8, appCtx.Head, prec)
