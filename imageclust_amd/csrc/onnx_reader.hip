// onnx_reader.hip -- host-only: a dependency-free reader of ONNX ResNet50-v1 files (no protobuf, no onnx package).
//
// Replaces gocv.ReadNetFromONNX as used by LoadPretrainedModelONNX
//   (/root/reference/internal/embeddings/embeddings.go:28-43; model path literal "resnet50-v1-7.onnx" at
//    /root/reference/internal/workflow/workflow.go:49).
// It walks the protobuf wire format of ModelProto.graph.{node,initializer}, follows the GRAPH (never a hard-coded
// tensor naming scheme): Conv nodes in file order, the BatchNormalization that consumes each Conv's output, the Gemm
// at the end; validates every Conv against the ResNet50-v1 topology (stride on the first 1x1 of a bottleneck) and
// emits the "ICLW" blob of include/icl_model_format.h, which icl_model_load_blob then uploads.
//
// Field numbers (onnx.proto3): ModelProto.graph=7; GraphProto.node=1, .initializer=5; NodeProto.input=1, .output=2,
// .op_type=4, .attribute=5; AttributeProto.name=1, .f=2, .i=3, .floats=7, .ints=8; TensorProto.dims=1, .data_type=2,
// .float_data=4, .name=8, .raw_data=9.
#include "icl_common.h"

#include <cstring>
#include <map>
#include <new>
#include <set>
#include <string>
#include <vector>

namespace {

struct span {
    const uint8_t *p = nullptr;
    size_t n = 0;
};

struct pb_reader {
    const uint8_t *p, *end;
    bool ok = true;
    pb_reader(const uint8_t *b, size_t n) : p(b), end(b + n) {}
    bool more() const { return ok && p < end; }
    uint64_t varint()
    {
        uint64_t v = 0;
        int shift = 0;
        while (p < end && shift < 64) {
            const uint8_t b = *p++;
            v |= (uint64_t)(b & 0x7f) << shift;
            if (!(b & 0x80)) return v;
            shift += 7;
        }
        ok = false;
        return 0;
    }
    // reads one field header; for length-delimited fields `s` receives the payload; for varint `v`; fixed32/64 in v
    bool field(int &num, int &wt, uint64_t &v, span &s)
    {
        const uint64_t key = varint();
        if (!ok) return false;
        num = (int)(key >> 3);
        wt = (int)(key & 7);
        s = span();
        v = 0;
        switch (wt) {
        case 0: v = varint(); break;
        case 1:
            if (end - p < 8) { ok = false; return false; }
            memcpy(&v, p, 8);
            p += 8;
            break;
        case 2: {
            const uint64_t len = varint();
            if (!ok || (uint64_t)(end - p) < len) { ok = false; return false; }
            s.p = p;
            s.n = (size_t)len;
            p += len;
            break;
        }
        case 5: {
            if (end - p < 4) { ok = false; return false; }
            uint32_t w;
            memcpy(&w, p, 4);
            v = w;
            p += 4;
            break;
        }
        default: ok = false; return false;
        }
        return ok;
    }
};

struct onnx_tensor {
    std::vector<int64_t> dims;
    int dtype = 0;
    span raw, fdata; // raw_data bytes, or packed float_data
    std::vector<float> loose; // unpacked float_data entries (wire type 5), rare
    int64_t count() const // -1: a dimension is negative or the product leaves the range a file of this size can hold
    {
        int64_t c = 1;
        for (auto d : dims) {
            if (d < 0 || (d > 0 && c > ((int64_t)1 << 40) / d)) return -1;
            c *= d;
        }
        return c;
    }
    bool floats(std::vector<float> &out) const
    {
        const int64_t c = count();
        // dims come from the file: the payload (raw_data, float_data or loose entries) must hold exactly c floats, so c
        // is bounded by what was actually read before anything is allocated
        if (dtype != 1 || c < 0) return false; // FLOAT
        if (!(raw.n == (size_t)c * 4 || fdata.n == (size_t)c * 4 || (int64_t)loose.size() == c)) return false;
        out.resize((size_t)c);
        if (raw.n == (size_t)c * 4) { memcpy(out.data(), raw.p, raw.n); return true; }
        if (fdata.n == (size_t)c * 4) { memcpy(out.data(), fdata.p, fdata.n); return true; }
        if ((int64_t)loose.size() == c) { out = loose; return true; }
        return false;
    }
};

struct onnx_node {
    std::string op;
    std::vector<std::string> in, out;
    std::map<std::string, std::vector<int64_t>> ints;
    std::map<std::string, float> f;
};

static std::string str(const span &s) { return std::string((const char *)s.p, s.n); }

static void parse_ints(const span &s, std::vector<int64_t> &out)
{
    pb_reader r(s.p, s.n);
    while (r.more()) out.push_back((int64_t)r.varint());
}

static bool parse_tensor(const span &s, std::string &name, onnx_tensor &t)
{
    pb_reader r(s.p, s.n);
    int num, wt;
    uint64_t v;
    span f;
    while (r.more()) {
        if (!r.field(num, wt, v, f)) return false;
        if (num == 1) {
            if (wt == 0) t.dims.push_back((int64_t)v);
            else if (wt == 2) parse_ints(f, t.dims);
        } else if (num == 2 && wt == 0) t.dtype = (int)v;
        else if (num == 4) {
            if (wt == 2) t.fdata = f;
            else if (wt == 5) { uint32_t w = (uint32_t)v; float x; memcpy(&x, &w, 4); t.loose.push_back(x); }
        } else if (num == 8 && wt == 2) name = str(f);
        else if (num == 9 && wt == 2) t.raw = f;
    }
    return r.ok;
}

static bool parse_attr(const span &s, onnx_node &n)
{
    pb_reader r(s.p, s.n);
    int num, wt;
    uint64_t v;
    span f;
    std::string name;
    std::vector<int64_t> ints;
    bool has_i = false, has_f = false;
    int64_t iv = 0;
    float fv = 0;
    while (r.more()) {
        if (!r.field(num, wt, v, f)) return false;
        if (num == 1 && wt == 2) name = str(f);
        else if (num == 2 && wt == 5) { uint32_t w = (uint32_t)v; memcpy(&fv, &w, 4); has_f = true; }
        else if (num == 3 && wt == 0) { iv = (int64_t)v; has_i = true; }
        else if (num == 8) {
            if (wt == 0) ints.push_back((int64_t)v);
            else if (wt == 2) parse_ints(f, ints);
        }
    }
    if (has_i) ints.push_back(iv);
    if (!ints.empty()) n.ints[name] = ints;
    if (has_f) n.f[name] = fv;
    return r.ok;
}

static bool parse_node(const span &s, onnx_node &n)
{
    pb_reader r(s.p, s.n);
    int num, wt;
    uint64_t v;
    span f;
    while (r.more()) {
        if (!r.field(num, wt, v, f)) return false;
        if (wt != 2) continue;
        if (num == 1) n.in.push_back(str(f));
        else if (num == 2) n.out.push_back(str(f));
        else if (num == 4) n.op = str(f);
        else if (num == 5 && !parse_attr(f, n)) return false;
    }
    return r.ok;
}

static int resnet50_topology_onnx(icl_conv_rec *out)
{
    static const int nblocks[4] = {3, 4, 6, 3};
    int n = 0;
    out[n++] = icl_conv_rec{3, 64, 7, 2, 3, 224, 112, 0, 0, 0};
    int h = 56, cin = 64;
    for (int s = 0; s < 4; ++s) {
        const int cout = 256 << s, mid = cout / 4;
        for (int b = 0; b < nblocks[s]; ++b) {
            const int stride = (b == 0 && s > 0) ? 2 : 1, ho = h / stride;
            out[n++] = icl_conv_rec{cin, mid, 1, stride, 0, h, ho, 1, s + 1, b};
            out[n++] = icl_conv_rec{mid, mid, 3, 1, 1, ho, ho, 2, s + 1, b};
            out[n++] = icl_conv_rec{mid, cout, 1, 1, 0, ho, ho, 3, s + 1, b};
            if (b == 0) out[n++] = icl_conv_rec{cin, cout, 1, stride, 0, h, ho, 4, s + 1, b};
            cin = cout;
            h = ho;
        }
    }
    return n;
}

} // namespace

static int onnx_to_blob_impl(icl_ctx *ctx, const char *path, std::vector<char> &blob);

// Parses an ONNX file into an ICLW blob.  Returns ICL_OK or ICL_ERR_IO with a message in ctx.  No C++ exception may cross
// the C ABI above this call: allocation failures on hostile sizes become status codes here.
int icl_onnx_to_blob(icl_ctx *ctx, const char *path, std::vector<char> &blob)
{
    try {
        return onnx_to_blob_impl(ctx, path, blob);
    } catch (const std::bad_alloc &) {
        return icl_fail(ctx, ICL_ERR_NOMEM, "failed to load ResNet50 ONNX model from: %s (out of host memory)", path);
    } catch (...) {
        return icl_fail(ctx, ICL_ERR_IO, "failed to load ResNet50 ONNX model from: %s (malformed file)", path);
    }
}

static int onnx_to_blob_impl(icl_ctx *ctx, const char *path, std::vector<char> &blob)
{
    FILE *fp = fopen(path, "rb");
    if (!fp) return icl_fail(ctx, ICL_ERR_IO, "failed to load ResNet50 ONNX model from: %s", path); // embeddings.go:32
    fseek(fp, 0, SEEK_END);
    const long fsz = ftell(fp);
    fseek(fp, 0, SEEK_SET);
    std::vector<uint8_t> file((size_t)std::max<long>(fsz, 0));
    const bool rd = fsz > 0 && fread(file.data(), 1, file.size(), fp) == file.size();
    fclose(fp);
    if (!rd) return icl_fail(ctx, ICL_ERR_IO, "failed to load ResNet50 ONNX model from: %s (empty or unreadable)", path);

    // ModelProto -> graph
    span graph;
    {
        pb_reader r(file.data(), file.size());
        int num, wt;
        uint64_t v;
        span f;
        while (r.more()) {
            if (!r.field(num, wt, v, f)) break;
            if (num == 7 && wt == 2) graph = f;
        }
        if (!r.ok || !graph.p) return icl_fail(ctx, ICL_ERR_IO, "%s: not an ONNX ModelProto (no graph)", path);
    }
    std::map<std::string, onnx_tensor> init;
    std::vector<onnx_node> nodes;
    {
        pb_reader r(graph.p, graph.n);
        int num, wt;
        uint64_t v;
        span f;
        while (r.more()) {
            if (!r.field(num, wt, v, f)) break;
            if (wt != 2) continue;
            if (num == 1) {
                onnx_node n;
                if (!parse_node(f, n)) return icl_fail(ctx, ICL_ERR_IO, "%s: malformed NodeProto", path);
                nodes.push_back(std::move(n));
            } else if (num == 5) {
                std::string name;
                onnx_tensor t;
                if (!parse_tensor(f, name, t)) return icl_fail(ctx, ICL_ERR_IO, "%s: malformed TensorProto", path);
                init[name] = std::move(t);
            }
        }
        if (!r.ok) return icl_fail(ctx, ICL_ERR_IO, "%s: malformed GraphProto", path);
    }
    // The graph is WALKED by data flow (producer -> consumer), never trusted to list its nodes in any particular order:
    // stem Conv -> BN -> Relu -> MaxPool -> 16 bottleneck blocks { x -> c1 1x1 -> BN -> Relu -> c2 3x3 -> BN -> Relu -> c3 1x1
    // -> BN ; [x -> downsample 1x1 -> BN] ; Add ; Relu } -> GlobalAveragePool -> Flatten/Reshape -> Gemm.  The convolutions
    // come out in the blob's order (c1, c2, c3, downsample per block), whatever order the exporter wrote them in.
    std::map<std::string, const onnx_node *> bn_of;
    std::multimap<std::string, const onnx_node *> cons; // tensor -> nodes that read it as DATA (not as a parameter)
    std::set<std::string> produced;
    const onnx_node *gemm = nullptr;
    size_t n_conv_nodes = 0;
    for (const auto &n : nodes) {
        for (const auto &o : n.out) produced.insert(o);
        if (n.op == "Conv") ++n_conv_nodes;
        if (n.op == "Gemm") gemm = &n;
        if (n.op == "BatchNormalization" && n.in.size() >= 5) bn_of[n.in[0]] = &n;
        const size_t ndata = n.op == "Add" ? std::min<size_t>(2, n.in.size()) : std::min<size_t>(1, n.in.size());
        for (size_t q = 0; q < ndata; ++q) cons.insert({n.in[q], &n});
    }
    icl_conv_rec topo[ICL_RESNET50_NCONV];
    const int nconv = resnet50_topology_onnx(topo);
    if ((int)n_conv_nodes != nconv || !gemm)
        return icl_fail(ctx, ICL_ERR_IO, "%s: expected a ResNet50-v1 graph (53 Conv + 1 Gemm), found %zu Conv%s", path, n_conv_nodes, gemm ? "" : ", no Gemm");
    auto readers = [&](const std::string &t, const char *op) {
        std::vector<const onnx_node *> r;
        auto range = cons.equal_range(t);
        for (auto it = range.first; it != range.second; ++it)
            if (it->second->op == op) r.push_back(it->second);
        return r;
    };
    auto only = [&](const std::string &t, const char *op) -> const onnx_node * {
        auto r = readers(t, op);
        return (r.size() == 1 && !r[0]->out.empty()) ? r[0] : nullptr;
    };
    auto cout_of = [&](const onnx_node *c) -> int64_t {
        if (c->in.size() < 2) return -1;
        auto it = init.find(c->in[1]);
        return (it == init.end() || it->second.dims.size() != 4) ? -1 : it->second.dims[0];
    };
    auto walk_fail = [&](int idx, const char *what) {
        return icl_fail(ctx, ICL_ERR_IO, "%s: not a ResNet50-v1 graph: conv %d: %s", path, idx, what);
    };
    std::vector<const onnx_node *> convs;
    std::string x;
    { // stem: the Conv whose data input no node produces (the graph input)
        const onnx_node *stem = nullptr;
        for (const auto &n : nodes)
            if (n.op == "Conv" && !n.in.empty() && !produced.count(n.in[0]) && !init.count(n.in[0])) stem = stem ? nullptr : &n;
        if (!stem || stem->out.empty()) return walk_fail(0, "no unique convolution reads the graph input");
        convs.push_back(stem);
        const onnx_node *bn = only(stem->out[0], "BatchNormalization");
        const onnx_node *re = bn ? only(bn->out[0], "Relu") : nullptr;
        const onnx_node *mp = re ? only(re->out[0], "MaxPool") : nullptr;
        if (!mp) return walk_fail(0, "expected Conv -> BatchNormalization -> Relu -> MaxPool");
        x = mp->out[0];
    }
    for (int i = 1; i < nconv;) {
        const bool has_ds = i + 3 < nconv && topo[i + 3].role == 4; // the blob's order: c1, c2, c3, then the block's downsample if it has one
        auto cs = readers(x, "Conv");
        const onnx_node *c1 = nullptr, *ds = nullptr;
        for (const onnx_node *c : cs) {
            if (cout_of(c) == topo[i].cout && !c1) c1 = c;
            else if (has_ds && cout_of(c) == topo[i + 3].cout && !ds) ds = c;
            else return walk_fail(i, "unexpected convolution on the block input");
        }
        if (!c1 || (has_ds && !ds) || c1->out.empty()) return walk_fail(i, "block input does not feed the expected 1x1 convolution(s)");
        const onnx_node *b1 = only(c1->out[0], "BatchNormalization"), *r1 = b1 ? only(b1->out[0], "Relu") : nullptr;
        const onnx_node *c2 = r1 ? only(r1->out[0], "Conv") : nullptr;
        const onnx_node *b2 = c2 ? only(c2->out[0], "BatchNormalization") : nullptr, *r2 = b2 ? only(b2->out[0], "Relu") : nullptr;
        const onnx_node *c3 = r2 ? only(r2->out[0], "Conv") : nullptr;
        const onnx_node *b3 = c3 ? only(c3->out[0], "BatchNormalization") : nullptr;
        const onnx_node *add = b3 ? only(b3->out[0], "Add") : nullptr;
        if (!add) return walk_fail(i, "expected 1x1 -> BN -> Relu -> 3x3 -> BN -> Relu -> 1x1 -> BN -> Add");
        if (add->in.size() != 2 || add->out.empty()) return walk_fail(i, "the block's Add must have two inputs and one output"); // (a hostile graph: a one-input Add would be indexed out of bounds below)
        const std::string &other = add->in[0] == b3->out[0] ? add->in[1] : add->in[0];
        if (has_ds) {
            const onnx_node *bd = ds->out.empty() ? nullptr : only(ds->out[0], "BatchNormalization");
            if (!bd || bd->out[0] != other) return walk_fail(i + 3, "the downsample branch does not reach the block's Add");
        } else if (other != x)
            return walk_fail(i, "the identity branch does not reach the block's Add");
        const onnx_node *ro = only(add->out[0], "Relu");
        if (!ro) return walk_fail(i, "expected Add -> Relu");
        convs.push_back(c1);
        convs.push_back(c2);
        convs.push_back(c3);
        if (has_ds) convs.push_back(ds);
        x = ro->out[0];
        i += has_ds ? 4 : 3;
    }
    { // head: GlobalAveragePool -> (Flatten | Reshape) -> Gemm
        const onnx_node *gp = only(x, "GlobalAveragePool");
        const onnx_node *fl = gp ? only(gp->out[0], "Flatten") : nullptr;
        if (gp && !fl) fl = only(gp->out[0], "Reshape");
        if (!fl || gemm->in.empty() || gemm->in[0] != fl->out[0]) return walk_fail(nconv, "expected GlobalAveragePool -> Flatten -> Gemm after the last block");
    }
    if ((int)convs.size() != nconv) return walk_fail((int)convs.size(), "the walk did not visit every convolution");

    icl_blob_header h;
    memset(&h, 0, sizeof h);
    h.magic = ICL_BLOB_MAGIC;
    h.version = ICL_BLOB_VERSION;
    h.n_conv = ICL_RESNET50_NCONV;
    h.bn_eps = -1.0f;
    std::vector<float> payload, tmp;
    auto need = [&](const std::string &name, std::vector<int64_t> dims, const char *what, int idx) -> int {
        auto it = init.find(name);
        if (it == init.end()) return icl_fail(ctx, ICL_ERR_IO, "%s: conv %d: initializer '%s' (%s) not found", path, idx, name.c_str(), what);
        if (it->second.dims != dims || !it->second.floats(tmp))
            return icl_fail(ctx, ICL_ERR_IO, "%s: conv %d: initializer '%s' (%s) has the wrong shape or dtype", path, idx, name.c_str(), what);
        payload.insert(payload.end(), tmp.begin(), tmp.end());
        return ICL_OK;
    };
    for (int i = 0; i < nconv; ++i) {
        const onnx_node &c = *convs[i];
        const icl_conv_rec &t = topo[i];
        if (c.in.size() < 2 || c.out.empty()) return icl_fail(ctx, ICL_ERR_IO, "%s: conv %d has no weight input", path, i);
        auto geti = [&](const char *k, size_t j, int64_t dflt) {
            auto it = c.ints.find(k);
            return (it == c.ints.end() || it->second.size() <= j) ? dflt : it->second[j];
        };
        const auto wit = init.find(c.in[1]);
        if (wit == init.end() || wit->second.dims.size() != 4) return icl_fail(ctx, ICL_ERR_IO, "%s: conv %d: weight initializer missing", path, i);
        const auto &wd = wit->second.dims;
        const int64_t k = wd[2];
        if (wd[0] != t.cout || wd[1] != t.cin || wd[2] != t.k || wd[3] != t.k || geti("strides", 0, 1) != t.stride || geti("pads", 0, 0) != t.pad ||
            geti("group", 0, 1) != 1 || geti("dilations", 0, 1) != 1 || geti("kernel_shape", 0, k) != t.k)
            return icl_fail(ctx, ICL_ERR_IO, "%s: conv %d is %lldx%lld k%lld s%lld p%lld, ResNet50-v1 expects %dx%d k%d s%d p%d", path, i,
                            (long long)wd[1], (long long)wd[0], (long long)wd[2], (long long)geti("strides", 0, 1), (long long)geti("pads", 0, 0), t.cin,
                            t.cout, t.k, t.stride, t.pad);
        ICL_TRY(need(c.in[1], {t.cout, t.cin, t.k, t.k}, "W", i));
        h.has_bias[i] = c.in.size() >= 3 && !c.in[2].empty();
        if (h.has_bias[i]) ICL_TRY(need(c.in[2], {t.cout}, "B", i));
        const auto bit = bn_of.find(c.out[0]);
        if (bit == bn_of.end()) return icl_fail(ctx, ICL_ERR_IO, "%s: conv %d is not followed by BatchNormalization", path, i);
        const onnx_node &bn = *bit->second;
        const float eps = bn.f.count("epsilon") ? bn.f.at("epsilon") : 1e-5f;
        if (h.bn_eps < 0) h.bn_eps = eps;
        else if (h.bn_eps != eps) return icl_fail(ctx, ICL_ERR_UNSUPPORTED, "%s: BatchNormalization layers use different epsilons", path);
        static const char *names[4] = {"scale", "B", "mean", "var"};
        for (int q = 0; q < 4; ++q) ICL_TRY(need(bn.in[1 + q], {t.cout}, names[q], i));
    }
    // dense0: Gemm(x, W, b) with transB = 1 (W is [1000][2048]) or 0 (W is [2048][1000])
    {
        if (gemm->in.size() < 3) return icl_fail(ctx, ICL_ERR_IO, "%s: Gemm needs W and b", path);
        const auto wit = init.find(gemm->in[1]);
        if (wit == init.end() || !wit->second.floats(tmp)) return icl_fail(ctx, ICL_ERR_IO, "%s: Gemm weight missing", path);
        const bool transB = gemm->ints.count("transB") && gemm->ints.at("transB")[0] == 1;
        const std::vector<int64_t> want = transB ? std::vector<int64_t>{ICL_FC_OUT, ICL_FEAT_DIM} : std::vector<int64_t>{ICL_FEAT_DIM, ICL_FC_OUT};
        if (wit->second.dims != want) return icl_fail(ctx, ICL_ERR_IO, "%s: Gemm weight has the wrong shape", path);
        if (transB) payload.insert(payload.end(), tmp.begin(), tmp.end());
        else
            for (int o = 0; o < ICL_FC_OUT; ++o)
                for (int i = 0; i < ICL_FEAT_DIM; ++i) payload.push_back(tmp[(size_t)i * ICL_FC_OUT + o]);
        ICL_TRY(need(gemm->in[2], {ICL_FC_OUT}, "fc bias", 53));
    }
    blob.resize(sizeof h + payload.size() * 4);
    memcpy(blob.data(), &h, sizeof h);
    memcpy(blob.data() + sizeof h, payload.data(), payload.size() * 4);
    return ICL_OK;
}

// The conversion alone, for hosts without a GPU: ONNX file -> ICLW blob bytes (include/icl_model_format.h).  With blob == NULL
// only *bytes is returned.  tests/test_onnx_reader.py compares the blob's tensors with what an independent protobuf parser
// finds in the same file.
extern "C" int icl_onnx_to_blob_file(const char *path, void *blob, int64_t cap_bytes, int64_t *bytes)
{
    if (!path || !bytes) return icl_fail(nullptr, ICL_ERR_ARG, "icl_onnx_to_blob_file: bad argument");
    std::vector<char> b;
    ICL_TRY(icl_onnx_to_blob(nullptr, path, b));
    *bytes = (int64_t)b.size();
    if (blob) {
        if (cap_bytes < (int64_t)b.size()) return icl_fail(nullptr, ICL_ERR_ARG, "icl_onnx_to_blob_file: buffer too small");
        memcpy(blob, b.data(), b.size());
    }
    return ICL_OK;
}
