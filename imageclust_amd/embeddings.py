"""Host-side mirror of /root/reference/internal/embeddings/embeddings.go (lines 17-163) over the HIP engine.

Same names and error behaviour as the Go package; gocv.Net / gocv.Mat become this module's opaque Net / Mat.
GenerateLabelVector / CombineEmbeddings (embeddings.go:166-183) are mirrored as plain host code; BuildLabelSet
(:188-236) takes the label detector as a callable (the Rekognition client itself is out of scope).
SaveEmbeddings / LoadEmbeddings give the unused CacheDir (embeddings.go:19) a flat on-disk E format (SURVEY.md 8f
rank 4).  No CPU fallback: every forward pass runs in libimageclust_hip.so.
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
    """embeddings.go:46-116 -> (Mat, err): IMRead (baseline or progressive JPEG decoded bit-identically to libjpeg-turbo, PNG bit-identically to Pillow / libpng, binary PPM) ->
    Resize 224x224 INTER_LINEAR -> RGB; Mat.Blob() gives the 1x3x224x224 fp32 blob scaled by 1/255."""
    try:
        rgb = _lib.load_image_224(imagePath)  # IMRead (+ EXIF orientation) -> cv::resize -> RGB; icl_preprocess_file = this + Blob()
    except _lib.ICLError as e:
        return None, str(e).split(": ", 1)[-1]
    return Mat(rgb), None


def GetImageEmbedding(appCtx: AppContext, imagePath: str):
    """embeddings.go:119-163 -> (embedding []float32, err)."""
    if appCtx.Net is None or appCtx.Net.Empty():
        return None, "failed to generate embedding for image: %s" % imagePath
    # embeddings.go:133 serialises batch-1 forwards behind NetMutex; the engine instead COALESCES concurrent callers into one
    # batched forward pass (icl_embed_file), so the mutex is kept as a field for signature parity but not taken here
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


# ---- label vectors (embeddings.go:166-236): host-side glue, no GPU work -------------------------------------------
def GenerateLabelVector(labels: List[str], labelSet: Dict[str, int]) -> np.ndarray:
    """embeddings.go:166-174: one-hot over the full label set; unknown labels are ignored."""
    v = np.zeros(len(labelSet), np.float32)
    for label in labels:
        idx = labelSet.get(label)
        if idx is not None:
            v[idx] = 1.0
    return v


def CombineEmbeddings(embedding, labelVector) -> np.ndarray:
    """embeddings.go:177-183: concatenation [embedding | labelVector] (D becomes 1000 + |labels|, or 2048 + |labels|)."""
    return np.concatenate([np.asarray(embedding, np.float32).ravel(), np.asarray(labelVector, np.float32).ravel()])


def BuildLabelSet(appCtx: AppContext, detect_labels) -> Optional[Exception]:
    """embeddings.go:188-236 with the Rekognition client replaced by `detect_labels(imagePath) -> [label names]`:
    files of ImageDir in os.ReadDir order (sorted by name), labels indexed in order of first appearance, per-file
    label lists stored in LabelsMapping under the NetMutex-free Mutex of the reference (here: NetMutex)."""
    import os

    try:
        names = sorted(os.listdir(appCtx.ImageDir))
    except OSError as e:
        return RuntimeError("failed to read image directory: %s" % e)
    labelSet: Dict[str, int] = {}
    for name in names:
        path = os.path.join(appCtx.ImageDir, name)
        if os.path.isdir(path):
            continue
        try:
            labels = list(detect_labels(path))
        except Exception as e:  # noqa: BLE001 - mirrored error path
            return RuntimeError("failed to detect labels for image %s: %s" % (name, e))
        for lab in labels:
            if lab not in labelSet:
                labelSet[lab] = len(labelSet)
        with appCtx.NetMutex:
            appCtx.LabelsMapping[name] = labels
    appCtx.LabelSet = labelSet
    return None


# ---- embedding cache: flat N x D fp32 + ids --------------------------------------------------
_CACHE_MAGIC = b"ICLE0001"


def SaveEmbeddings(path: str, ids: List[str], E) -> None:
    """File = magic(8) | N(i64) | D(i64) | ids_bytes(i64) | utf-8 ids joined by '\n' | N*D little-endian fp32."""
    E = np.ascontiguousarray(E, dtype="<f4")
    if E.ndim != 2 or len(ids) != E.shape[0]:
        raise ValueError("SaveEmbeddings: %d ids for an array of shape %s" % (len(ids), E.shape))
    if any("\n" in s for s in ids):
        raise ValueError("SaveEmbeddings: ids must not contain newlines")
    blob = "\n".join(ids).encode("utf-8")
    with open(path, "wb") as f:
        f.write(_CACHE_MAGIC)
        f.write(np.array([E.shape[0], E.shape[1], len(blob)], "<i8").tobytes())
        f.write(blob)
        f.write(E.tobytes())


def LoadEmbeddings(path: str, mmap: bool = False):
    """Returns (ids, E[N][D] fp32).  mmap=True maps the matrix instead of reading it (large N)."""
    with open(path, "rb") as f:
        if f.read(8) != _CACHE_MAGIC:
            raise ValueError("%s is not an imageclust embedding cache" % path)
        n, d, nb = (int(x) for x in np.frombuffer(f.read(24), "<i8"))
        blob = f.read(nb)
        off = f.tell()
        ids = blob.decode("utf-8").split("\n") if n else []
        if len(ids) != n:
            raise ValueError("%s: %d ids for %d rows" % (path, len(ids), n))
        if mmap:
            E = np.memmap(path, dtype="<f4", mode="r", offset=off, shape=(n, d))
        else:
            E = np.frombuffer(f.read(n * d * 4), "<f4").reshape(n, d).copy()
            if E.shape != (n, d):
                raise ValueError("%s is truncated" % path)
    return ids, E
