/*
 * oracle/resnet_ref.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU fp32 restatement of the embedding half of the reference hot path,
 *     /root/reference/internal/embeddings/embeddings.go:46-116  (PreprocessImage: RGB/255, NCHW)
 *     /root/reference/internal/embeddings/embeddings.go:119-163 (GetImageEmbedding: forward to
 *                                                                "resnetv17_dense0_fwd", batch 1, fp32)
 * The arithmetic itself lives in a third-party dependency that is NOT under /root/reference:
 *     gocv.io/x/gocv v0.40.0 (go.mod:11) -> OpenCV dnn/imgproc (Dockerfile:24-25 pins 4.6.0),
 * running the ONNX-model-zoo graph resnet50-v1-7.onnx (workflow.go:49; blob absent:
 * .MISSING_LARGE_BLOBS:1).  This file therefore restates the PUBLISHED algorithm of that graph
 * (ONNX Conv / BatchNormalization / Relu / MaxPool / GlobalAveragePool / Gemm operator definitions,
 * ResNet50-v1 topology with the stride on the first 1x1 of each bottleneck) in naive NCHW fp32.
 *
 * PARITY UNPINNED: the reference holds no test, fixture or golden vector at this boundary and the
 * model file cannot be obtained offline, so nothing pins this restatement to the reference's bits.
 * It is cross-checked only against torch-CPU conv2d/batch_norm/max_pool2d (tests/test_oracle_resnet.py).
 *
 * Weights arrive as an "ICLW" blob (include/icl_model_format.h).
 * Build: gcc -O2 -ffp-contract=off -fopenmp (see oracle/Makefile).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/icl_model_format.h"

/* ---- topology (restated independently of the product library) ---------------------------------- */
int icl_ref_resnet50_topology(icl_conv_rec *out /* [53] */)
{
    static const int nblocks[4] = {3, 4, 6, 3};
    int n = 0, h = 224, cin = 3;
    icl_conv_rec r;
    memset(&r, 0, sizeof r);
    r.cin = 3; r.cout = 64; r.k = 7; r.stride = 2; r.pad = 3; r.hin = 224; r.hout = 112; r.role = 0;
    out[n++] = r;
    h = 56; /* after maxpool 3x3/2 p1 on 112 */
    cin = 64;
    for (int s = 0; s < 4; ++s) {
        int cout = 256 << s, mid = cout / 4;
        for (int b = 0; b < nblocks[s]; ++b) {
            int stride = (b == 0 && s > 0) ? 2 : 1;
            int hout = h / stride;
            icl_conv_rec c1 = {cin, mid, 1, stride, 0, h, hout, 1, s + 1, b};
            icl_conv_rec c2 = {mid, mid, 3, 1, 1, hout, hout, 2, s + 1, b};
            icl_conv_rec c3 = {mid, cout, 1, 1, 0, hout, hout, 3, s + 1, b};
            out[n++] = c1; out[n++] = c2; out[n++] = c3;
            if (b == 0) {
                icl_conv_rec ds = {cin, cout, 1, stride, 0, h, hout, 4, s + 1, b};
                out[n++] = ds;
            }
            cin = cout;
            h = hout;
        }
    }
    return n; /* 53 */
}

/* Gluon resnet50_v1: the bottleneck's two 1x1 convs carry a bias; 3x3, conv0 and downsample do not. */
void icl_ref_default_bias_flags(uint8_t *flags /* [64] */)
{
    icl_conv_rec t[ICL_RESNET50_NCONV];
    int n = icl_ref_resnet50_topology(t);
    memset(flags, 0, 64);
    for (int i = 0; i < n; ++i) flags[i] = (t[i].role == 1 || t[i].role == 3) ? 1 : 0;
}

int64_t icl_ref_blob_floats(const icl_blob_header *h)
{
    icl_conv_rec t[ICL_RESNET50_NCONV];
    int n = icl_ref_resnet50_topology(t);
    int64_t tot = 0;
    for (int i = 0; i < n; ++i) {
        tot += (int64_t)t[i].cout * t[i].cin * t[i].k * t[i].k;
        if (h->has_bias[i]) tot += t[i].cout;
        tot += 4 * (int64_t)t[i].cout;
    }
    tot += (int64_t)ICL_FC_OUT * ICL_FEAT_DIM + ICL_FC_OUT;
    return tot;
}

/* ---- operators (NCHW, one image) -------------------------------------------------------------- */

/* ONNX Conv, group 1, square kernel, symmetric zero padding.  out[co][oy][ox] = bias[co] +
 * sum over (ci, kh, kw) in that order of in[ci][oy*s-p+kh][ox*s-p+kw] * w[co][ci][kh][kw]; fp32,
 * product rounded then sum rounded.  Row-axpy form: identical per-element accumulation order. */
void icl_ref_conv2d(const float *in, int cin, int hin, int win, const float *w, const float *bias, int cout,
                    int k, int stride, int pad, float *out, int hout, int wout)
{
#pragma omp parallel for schedule(static)
    for (int co = 0; co < cout; ++co) {
        float *o = out + (int64_t)co * hout * wout;
        float b = bias ? bias[co] : 0.0f;
        for (int i = 0; i < hout * wout; ++i) o[i] = b;
        for (int ci = 0; ci < cin; ++ci) {
            const float *ip = in + (int64_t)ci * hin * win;
            const float *wp = w + ((int64_t)co * cin + ci) * k * k;
            for (int kh = 0; kh < k; ++kh)
                for (int kw = 0; kw < k; ++kw) {
                    float wv = wp[kh * k + kw];
                    for (int oy = 0; oy < hout; ++oy) {
                        int iy = oy * stride - pad + kh;
                        if (iy < 0 || iy >= hin) continue; /* zero padding contributes +0 exactly */
                        const float *irow = ip + (int64_t)iy * win;
                        float *orow = o + (int64_t)oy * wout;
                        /* valid ox range: 0 <= ox*stride - pad + kw < win */
                        int ox0 = 0, ox1 = wout;
                        while (ox0 < wout && ox0 * stride - pad + kw < 0) ++ox0;
                        while (ox1 > ox0 && (ox1 - 1) * stride - pad + kw >= win) --ox1;
                        if (stride == 1) {
                            const float *ir = irow + (kw - pad);
                            for (int ox = ox0; ox < ox1; ++ox) {
                                float p = ir[ox] * wv;
                                orow[ox] = orow[ox] + p;
                            }
                        } else {
                            for (int ox = ox0; ox < ox1; ++ox) {
                                float p = irow[ox * stride - pad + kw] * wv;
                                orow[ox] = orow[ox] + p;
                            }
                        }
                    }
                }
        }
    }
}

/* ONNX BatchNormalization (inference): y = gamma * (x - mean) / sqrt(var + eps) + beta.
 * Optional fused residual add and ReLU applied AFTER it, in graph order (add, then relu). */
void icl_ref_bn_act(float *x, int c, int hw, const float *gamma, const float *beta, const float *mean,
                    const float *var, float eps, const float *residual, int relu)
{
#pragma omp parallel for schedule(static)
    for (int ch = 0; ch < c; ++ch) {
        float inv = 1.0f / sqrtf(var[ch] + eps);
        float *p = x + (int64_t)ch * hw;
        const float *r = residual ? residual + (int64_t)ch * hw : 0;
        for (int i = 0; i < hw; ++i) {
            float v = (p[i] - mean[ch]) * inv;
            v = v * gamma[ch] + beta[ch];
            if (r) v = v + r[i];
            if (relu && !(v > 0.0f)) v = 0.0f;
            p[i] = v;
        }
    }
}

/* ONNX MaxPool k3 s2 p1: padding never wins (treated as -inf). */
void icl_ref_maxpool3x3s2(const float *in, int c, int hin, int win, float *out, int hout, int wout)
{
#pragma omp parallel for schedule(static)
    for (int ch = 0; ch < c; ++ch)
        for (int oy = 0; oy < hout; ++oy)
            for (int ox = 0; ox < wout; ++ox) {
                float m = -INFINITY;
                for (int kh = 0; kh < 3; ++kh)
                    for (int kw = 0; kw < 3; ++kw) {
                        int iy = oy * 2 - 1 + kh, ix = ox * 2 - 1 + kw;
                        if (iy < 0 || iy >= hin || ix < 0 || ix >= win) continue;
                        float v = in[((int64_t)ch * hin + iy) * win + ix];
                        if (v > m) m = v;
                    }
                out[((int64_t)ch * hout + oy) * wout + ox] = m;
            }
}

/* ONNX GlobalAveragePool: sequential fp32 sum over the map, divided by its size. */
void icl_ref_global_avgpool(const float *in, int c, int hw, float *out)
{
    for (int ch = 0; ch < c; ++ch) {
        float s = 0.0f;
        for (int i = 0; i < hw; ++i) s = s + in[(int64_t)ch * hw + i];
        out[ch] = s / (float)hw;
    }
}

/* ONNX Gemm (transB=1): y[o] = sum_i x[i]*W[o][i] (sequential fp32) + b[o]. */
void icl_ref_fc(const float *x, int nin, const float *W, const float *b, int nout, float *y)
{
    for (int o = 0; o < nout; ++o) {
        float s = 0.0f;
        for (int i = 0; i < nin; ++i) {
            float p = x[i] * W[(int64_t)o * nin + i];
            s = s + p;
        }
        y[o] = s + b[o];
    }
}

/* embeddings.go:82,96: u8 HWC RGB -> fp32 NCHW scaled by 1/255 (BlobFromImage scalefactor 1.0/255.0,
 * mean 0, swapRB=false, crop=false).  OpenCV applies the scale as an fp32 multiply by float(1/255). */
void icl_ref_preprocess_rgb_u8(const uint8_t *hwc, float *nchw /* [3][224][224] */)
{
    const float sc = (float)(1.0 / 255.0);
    for (int y = 0; y < ICL_IMG_H; ++y)
        for (int x = 0; x < ICL_IMG_W; ++x)
            for (int c = 0; c < 3; ++c)
                nchw[((int64_t)c * ICL_IMG_H + y) * ICL_IMG_W + x] = (float)hwc[((int64_t)y * ICL_IMG_W + x) * 3 + c] * sc;
}

typedef struct {
    const float *w, *bias, *gamma, *beta, *mean, *var;
} conv_params;

/* embeddings.go:137-152 for ONE image: forward to the pooled 2048-d vector and the dense0 1000-d vector.
 * Either output pointer may be NULL.  Returns 0, or nonzero for a malformed blob. */
int icl_ref_resnet50_forward(const void *blob, int64_t blob_bytes, const uint8_t *img_hwc_rgb, float *pooled2048,
                             float *dense1000)
{
    const icl_blob_header *h = (const icl_blob_header *)blob;
    if (blob_bytes < (int64_t)sizeof *h || h->magic != ICL_BLOB_MAGIC || h->n_conv != ICL_RESNET50_NCONV) return 1;
    if (blob_bytes != (int64_t)sizeof *h + 4 * icl_ref_blob_floats(h)) return 2;
    icl_conv_rec t[ICL_RESNET50_NCONV];
    int n = icl_ref_resnet50_topology(t);
    conv_params cp[ICL_RESNET50_NCONV];
    const float *p = (const float *)((const char *)blob + sizeof *h);
    for (int i = 0; i < n; ++i) {
        cp[i].w = p; p += (int64_t)t[i].cout * t[i].cin * t[i].k * t[i].k;
        cp[i].bias = 0;
        if (h->has_bias[i]) { cp[i].bias = p; p += t[i].cout; }
        cp[i].gamma = p; p += t[i].cout;
        cp[i].beta = p; p += t[i].cout;
        cp[i].mean = p; p += t[i].cout;
        cp[i].var = p; p += t[i].cout;
    }
    const float *fcw = p; p += (int64_t)ICL_FC_OUT * ICL_FEAT_DIM;
    const float *fcb = p;

    const int64_t maxact = (int64_t)64 * 112 * 112; /* == 256*56*56, the largest activation */
    float *bufs[4];
    for (int i = 0; i < 4; ++i) bufs[i] = (float *)malloc((size_t)maxact * sizeof(float));
    float *x = bufs[0], *a = bufs[1], *b = bufs[2], *idn = bufs[3];

    icl_ref_preprocess_rgb_u8(img_hwc_rgb, a);
    icl_ref_conv2d(a, 3, 224, 224, cp[0].w, cp[0].bias, 64, 7, 2, 3, b, 112, 112);
    icl_ref_bn_act(b, 64, 112 * 112, cp[0].gamma, cp[0].beta, cp[0].mean, cp[0].var, h->bn_eps, 0, 1);
    icl_ref_maxpool3x3s2(b, 64, 112, 112, x, 56, 56);

    int ci = 1, cur_c = 64, cur_h = 56;
    while (ci < n) {
        const icl_conv_rec *c1 = &t[ci], *c2 = &t[ci + 1], *c3 = &t[ci + 2];
        int has_ds = (c1->block == 0);
        int ho = c1->hout;
        /* body */
        icl_ref_conv2d(x, c1->cin, cur_h, cur_h, cp[ci].w, cp[ci].bias, c1->cout, 1, c1->stride, 0, a, ho, ho);
        icl_ref_bn_act(a, c1->cout, ho * ho, cp[ci].gamma, cp[ci].beta, cp[ci].mean, cp[ci].var, h->bn_eps, 0, 1);
        icl_ref_conv2d(a, c2->cin, ho, ho, cp[ci + 1].w, cp[ci + 1].bias, c2->cout, 3, 1, 1, b, ho, ho);
        icl_ref_bn_act(b, c2->cout, ho * ho, cp[ci + 1].gamma, cp[ci + 1].beta, cp[ci + 1].mean, cp[ci + 1].var,
                       h->bn_eps, 0, 1);
        icl_ref_conv2d(b, c3->cin, ho, ho, cp[ci + 2].w, cp[ci + 2].bias, c3->cout, 1, 1, 0, a, ho, ho);
        const float *res = x;
        if (has_ds) {
            const icl_conv_rec *ds = &t[ci + 3];
            icl_ref_conv2d(x, ds->cin, cur_h, cur_h, cp[ci + 3].w, cp[ci + 3].bias, ds->cout, 1, ds->stride, 0, idn, ho,
                           ho);
            icl_ref_bn_act(idn, ds->cout, ho * ho, cp[ci + 3].gamma, cp[ci + 3].beta, cp[ci + 3].mean, cp[ci + 3].var,
                           h->bn_eps, 0, 0);
            res = idn;
        }
        /* bn(c3) + residual, then relu */
        icl_ref_bn_act(a, c3->cout, ho * ho, cp[ci + 2].gamma, cp[ci + 2].beta, cp[ci + 2].mean, cp[ci + 2].var,
                       h->bn_eps, res, 1);
        float *tmp = x; x = a; a = tmp;
        cur_c = c3->cout;
        cur_h = ho;
        ci += has_ds ? 4 : 3;
    }
    float pooled[ICL_FEAT_DIM];
    icl_ref_global_avgpool(x, cur_c, cur_h * cur_h, pooled);
    if (pooled2048) memcpy(pooled2048, pooled, sizeof pooled);
    if (dense1000) icl_ref_fc(pooled, ICL_FEAT_DIM, fcw, fcb, ICL_FC_OUT, dense1000);
    for (int i = 0; i < 4; ++i) free(bufs[i]);
    return 0;
}
