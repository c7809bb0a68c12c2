/*
 * icl_model_format.h -- the "ICLW" weight blob: the one interchange format for ResNet50-v1 weights
 * between a weight source (synthetic generator, ONNX initializer reader) and a consumer
 * (libimageclust_hip's model loader; the test oracle).
 *
 * Replaces: the in-memory cv::dnn::Net built by gocv.ReadNetFromONNX
 *           (/root/reference/internal/embeddings/embeddings.go:28-43) for the single graph the
 *           reference ever loads, "resnet50-v1-7.onnx" (/root/reference/internal/workflow/workflow.go:49).
 *
 * Layout (little endian):
 *   icl_blob_header (80 bytes)
 *   float32 payload, tensors in CANONICAL ORDER:
 *     for each conv c in topology order (conv0; then for stage 1..4, block 0..nb-1: c1, c2, c3, [ds if block 0]):
 *        W      [cout][cin][kh][kw]        (OIHW, as an ONNX Conv initializer)
 *        bias   [cout]                     only if header.has_bias[c] != 0
 *        gamma, beta, mean, var  [cout] each   (the BatchNormalization that follows the conv)
 *     fc W [1000][2048]  (ONNX Gemm with transB=1 layout), fc b [1000]
 *
 * Topology (ONNX model zoo ResNet50-v1-7 == MXNet-Gluon resnet50_v1; stride on the FIRST 1x1 of a
 * bottleneck): conv0 7x7/2 p3 3->64, BN, ReLU, maxpool 3x3/2 p1, stages of [3,4,6,3]
 * bottlenecks with widths 256/512/1024/2048, global average pool (2048-d "pooled" head), flatten,
 * dense0 2048->1000 ("resnetv17_dense0_fwd" head, embeddings.go:140).
 */
#ifndef ICL_MODEL_FORMAT_H
#define ICL_MODEL_FORMAT_H
#include <stdint.h>

#define ICL_BLOB_MAGIC 0x574C4349u /* 'I''C''L''W' */
#define ICL_BLOB_VERSION 1u
#define ICL_RESNET50_NCONV 53
#define ICL_FEAT_DIM 2048
#define ICL_FC_OUT 1000
#define ICL_IMG_H 224
#define ICL_IMG_W 224
#define ICL_IMG_C 3
#define ICL_IMG_BYTES (ICL_IMG_H * ICL_IMG_W * ICL_IMG_C)

typedef struct icl_blob_header {
    uint32_t magic;       /* ICL_BLOB_MAGIC */
    uint32_t version;     /* ICL_BLOB_VERSION */
    float bn_eps;         /* BatchNormalization epsilon (Gluon default 1e-5) */
    uint32_t n_conv;      /* ICL_RESNET50_NCONV */
    uint8_t has_bias[64]; /* per conv, canonical order */
} icl_blob_header;

/* One conv of the canonical topology. */
typedef struct icl_conv_rec {
    int32_t cin, cout, k, stride, pad;
    int32_t hin, hout;    /* square spatial size in/out */
    int32_t role;         /* 0 conv0, 1 c1, 2 c2, 3 c3, 4 downsample */
    int32_t stage, block; /* stage 1..4 (0 for conv0), block index inside the stage */
} icl_conv_rec;

#endif
