// multi_gpu.hip -- several MI355X behind ONE handle of the C-ABI ("icl_create(const int* devs, int ndev, ...)",
// 8e rows 1-2).  A Go service is one process: internal/workflow calls GetImageEmbedding / PerformClusteringWithConstraints
// (workflow.go:161,89) and cannot start one process per GPU, so the group drives N contexts from N host threads:
//   * embed: images shard by contiguous index ranges (no exchange: every GPU holds the weights);
//   * cluster: every GPU gets E, computes the distance rows of an area-balanced run of 128-row tile rows
//     (icl_ward_rows_partition) and its span is copied device-to-device into GPU 0's packed triangle (hipMemcpyPeerAsync:
//     xGMI when the devices are peers); GPU 0 runs the exact merge loop.  Results are bit-identical to one GPU: the same
//     kernel computes every row, only where it runs changes.
// Entries of `devices` may repeat (a test on a 1-GPU box builds a group of two contexts on device 0: the same code path,
// the peer copy degenerates to a device-to-device copy).
// The process-per-GPU path of bench.py uses the same building blocks (icl_ward_distance_rows_dev / icl_ward_span_ptr /
// icl_cluster_prefilled_dev) with RCCL send/recv as the transport.
#include "icl_common.h"

#include <new>
#include <thread>

struct icl_group {
    std::vector<icl_ctx *> ctx;
    std::string err;
    std::mutex mu;
};

static int group_fail(icl_group *g, int code, const std::string &msg)
{
    if (g) {
        std::lock_guard<std::mutex> lk(g->mu);
        g->err = msg;
    }
    return code;
}

extern "C" int icl_group_create(const int32_t *devices, int32_t ndev, icl_group **out)
{
    if (!devices || ndev < 1 || ndev > 64 || !out) return icl_fail(nullptr, ICL_ERR_ARG, "icl_group_create: bad argument");
    icl_group *g = new (std::nothrow) icl_group();
    if (!g) return icl_fail(nullptr, ICL_ERR_NOMEM, "icl_group_create: out of memory");
    for (int i = 0; i < ndev; ++i) {
        icl_ctx *c = nullptr;
        const int rc = icl_create(devices[i], &c);
        if (rc != ICL_OK) {
            for (icl_ctx *p : g->ctx) icl_destroy(p);
            delete g;
            return rc; // icl_create left its message in the thread-local error string
        }
        g->ctx.push_back(c);
    }
    // peer access between distinct devices (xGMI inside a node); failure is not fatal: copies then stage through the host
    for (int i = 0; i < ndev; ++i)
        for (int j = 0; j < ndev; ++j)
            if (devices[i] != devices[j]) {
                icl_device_guard dg(devices[i]);
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, devices[i], devices[j]) == hipSuccess && can) (void)hipDeviceEnablePeerAccess(devices[j], 0);
                (void)hipGetLastError(); // "already enabled" is fine
            }
    *out = g;
    return ICL_OK;
}

extern "C" void icl_group_destroy(icl_group *g)
{
    if (!g) return;
    for (icl_ctx *c : g->ctx) icl_destroy(c);
    delete g;
}

extern "C" int32_t icl_group_size(icl_group *g) { return g ? (int32_t)g->ctx.size() : 0; }
extern "C" icl_ctx *icl_group_ctx(icl_group *g, int32_t i) { return (g && i >= 0 && i < (int32_t)g->ctx.size()) ? g->ctx[(size_t)i] : nullptr; }
extern "C" const char *icl_group_last_error(icl_group *g) { return g ? g->err.c_str() : ""; }

// run f(i, ctx_i) on one host thread per context; first failure wins
template <typename F>
static int for_each_ctx(icl_group *g, F &&f)
{
    const int n = (int)g->ctx.size();
    std::vector<int> rc((size_t)n, ICL_OK);
    std::vector<std::thread> th;
    try {
        for (int i = 1; i < n; ++i) th.emplace_back([&, i] { rc[(size_t)i] = f(i, g->ctx[(size_t)i]); });
    } catch (...) {
        for (auto &t : th) t.join();
        return group_fail(g, ICL_ERR_NOMEM, "could not start a host thread per GPU");
    }
    rc[0] = f(0, g->ctx[0]);
    for (auto &t : th) t.join();
    for (int i = 0; i < n; ++i)
        if (rc[(size_t)i] != ICL_OK) return group_fail(g, rc[(size_t)i], std::string("GPU ") + std::to_string(i) + ": " + icl_last_error(g->ctx[(size_t)i]));
    return ICL_OK;
}

extern "C" int icl_group_load_synthetic(icl_group *g, uint64_t seed)
{
    if (!g) return ICL_ERR_ARG;
    return for_each_ctx(g, [&](int, icl_ctx *c) { return icl_model_load_synthetic(c, seed); });
}
extern "C" int icl_group_load_onnx(icl_group *g, const char *path)
{
    if (!g || !path) return ICL_ERR_ARG;
    return for_each_ctx(g, [&](int, icl_ctx *c) { return icl_model_load_onnx(c, path); });
}
extern "C" int icl_group_load_blob(icl_group *g, const void *blob, int64_t bytes)
{
    if (!g || !blob) return ICL_ERR_ARG;
    return for_each_ctx(g, [&](int, icl_ctx *c) { return icl_model_load_blob(c, blob, bytes); });
}

// contiguous index range of part i of n items over `parts` (the first n % parts parts hold one more)
static void shard_range(int64_t n, int parts, int i, int64_t &lo, int64_t &hi)
{
    const int64_t base = n / parts, extra = n % parts;
    lo = i * base + std::min<int64_t>(i, extra);
    hi = lo + base + (i < extra ? 1 : 0);
}

extern "C" int icl_group_embed_u8(icl_group *g, const uint8_t *hwc_rgb, int64_t n, int head, int prec, float *out)
{
    if (!g || n < 0 || (n && (!hwc_rgb || !out))) return group_fail(g, ICL_ERR_ARG, "icl_group_embed_u8: bad argument");
    const int parts = (int)g->ctx.size();
    return for_each_ctx(g, [&](int i, icl_ctx *c) {
        int64_t lo, hi;
        shard_range(n, parts, i, lo, hi);
        if (hi == lo) return (int)ICL_OK;
        return icl_embed_u8(c, hwc_rgb + lo * (int64_t)ICL_IMG_BYTES, hi - lo, head, prec, out + lo * head);
    });
}

extern "C" int icl_group_cluster(icl_group *g, const float *E, int64_t n, int32_t d, int32_t min_size, int32_t max_size, int update,
                                 int32_t *cluster_id, int32_t *member_rank, int32_t *n_clusters)
{
    if (!g || n < 0 || d < 0 || !n_clusters || (n && (!E || !cluster_id || !member_rank))) return group_fail(g, ICL_ERR_ARG, "icl_group_cluster: bad argument");
    const int parts = (int)g->ctx.size();
    icl_ctx *c0 = g->ctx[0];
    int64_t kk = 0;
    if (parts == 1 || update != ICL_UPDATE_EXACT || n < 2 * 128 * parts || icl_calc_optimal_clusters(n, min_size, max_size, &kk) != ICL_OK) {
        // one GPU, FAST mode (its MFMA tile is built on one GPU), inputs too small to deal out, or constraint errors: plain call
        const int rc = icl_cluster(c0, E, n, d, min_size, max_size, update, cluster_id, member_rank, n_clusters);
        return rc == ICL_OK ? rc : group_fail(g, rc, icl_last_error(c0));
    }
    int rc = icl_ward_prepare(c0, n, d);
    if (rc != ICL_OK) return group_fail(g, rc, icl_last_error(c0));
    std::vector<float *> dE((size_t)parts, nullptr), dSpan((size_t)parts, nullptr);
    auto cleanup = [&] {
        for (int i = 0; i < parts; ++i) {
            if (dE[(size_t)i]) (void)icl_dev_free(g->ctx[(size_t)i], dE[(size_t)i]);
            if (dSpan[(size_t)i]) (void)icl_dev_free(g->ctx[(size_t)i], dSpan[(size_t)i]);
        }
    };
    rc = for_each_ctx(g, [&](int i, icl_ctx *c) -> int {
        int64_t lo = 0, hi = 0, off = 0, cnt = 0;
        ICL_TRY(icl_ward_rows_partition(n, parts, i, &lo, &hi));
        ICL_TRY(icl_ward_span(lo, hi, &off, &cnt));
        void *p = nullptr;
        ICL_TRY(icl_dev_malloc(c, std::max<int64_t>(n * d * 4, 16), &p));
        dE[(size_t)i] = (float *)p;
        ICL_TRY(icl_memcpy_h2d(c, p, E, n * (int64_t)d * 4));
        if (hi == lo) return ICL_OK;
        if (i == 0) { // GPU 0 writes its own rows straight into its triangle
            void *dst = nullptr;
            int64_t c2 = 0;
            ICL_TRY(icl_ward_span_ptr(c, lo, hi, &dst, &c2));
            return icl_ward_distance_rows_dev(c, dE[0], n, d, lo, hi, (float *)dst);
        }
        ICL_TRY(icl_dev_malloc(c, std::max<int64_t>(cnt * 4, 16), &p));
        dSpan[(size_t)i] = (float *)p;
        ICL_TRY(icl_ward_distance_rows_dev(c, dE[(size_t)i], n, d, lo, hi, dSpan[(size_t)i]));
        return icl_ward_deposit_dev(c0, lo, hi, dSpan[(size_t)i]); // device i -> GPU 0's triangle (peer copy over xGMI)
    });
    if (rc == ICL_OK) {
        rc = icl_cluster_prefilled_dev(c0, dE[0], n, d, min_size, max_size, update, 0, 0, cluster_id, member_rank, n_clusters);
        if (rc != ICL_OK) group_fail(g, rc, icl_last_error(c0));
    }
    cleanup();
    return rc;
}
