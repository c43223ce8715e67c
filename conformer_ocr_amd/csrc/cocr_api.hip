// libcocr_hip.so -- C ABI (include/cocr.h) and host orchestration of the gfx950 kernels.
//
// A model is: the reference state dict (fp32, host) -> one packed device blob in the compute dtype
// (cocr_finalize) -> a workspace sized for (N, W) -> a fixed sequence of kernel launches on the
// caller's stream (cocr_forward).  Nothing here falls back to the CPU: without a GPU every compute
// entry point fails with COCR_EHIP.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <vector>

#include "../../include/cocr.h"
#include "attention.hip.h"
#include "common.hip.h"
#include "conv.hip.h"
#include "ctc.hip.h"
#include "ctc_loss.hip.h"
#include "train.hip.h"
#include "train_enc.hip.h"
#include "gemm.hip.h"
#include "ffn.hip.h"
#include "rowchain_args.hip.h"
#include "pack.hip.h"
#include "frontend.hip.h"
#include "norm.hip.h"
#include "preproc.hip.h"

// ------------------------------------------------------------------------------------ errors
static thread_local char g_err[512] = "";
static int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
#define HIP_TRY(expr)                                                                                      \
    do {                                                                                                   \
        hipError_t e_ = (expr);                                                                            \
        if (e_ != hipSuccess) return fail(COCR_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

extern "C" const char *cocr_last_error(void) { return g_err; }
#ifndef COCR_SRC_HASH
#define COCR_SRC_HASH "unknown"
#endif
extern "C" const char *cocr_version(void) { return "cocr-hip 0.2 (gfx950) src=" COCR_SRC_HASH; }

// ------------------------------------------------------------------------------------ model
struct HostTensor {
    std::vector<float> data;
    std::vector<int64_t> shape;
    bool set = false;
};

struct FfnW { size_t ln_g, ln_b, w1, b1, w2, b2; };
struct LayerW {
    FfnW ffn[2];
    size_t a_ln_g, a_ln_b, wqkv, bqkv, ub, vb, wpos, wo, bo;       // wpos: pos_proj weight, fp32 (D, D): the positional tables are DERIVED from it
    size_t c_ln_g, c_ln_b, wpw1, bpw1, dww, dwb, wpw2, bpw2;
    size_t f_ln_g, f_ln_b;
};
struct StageW { size_t dw_w, dw_b, pw_w, pw_b; };   // one (depthwise, pointwise) frontend stage
struct BlobPlan {
    size_t w0, b0;                 // frontend conv.0 taps [C][9], bias
    std::vector<StageW> stages;    // sampling_num - 1 stages; stage 0's depthwise is fused with conv.0
    size_t wout, bout;
    std::vector<LayerW> layers;
    size_t wdec, bdec;
    size_t total = 0;
};

struct DevBuf { void *p = nullptr; size_t bytes = 0; };

struct ProfRec { int fam; hipEvent_t a, b; };
struct TrainState;

struct cocr_model {
    cocr_hparams hp;
    int device = 0;
    // derived
    int D, C, L, heads, dh, dhp, ff, ksz, ncls, H, snum;      // D / ff / dh / dhp: the ENGINE's dimensions (zero-padded when `padded`)
    int rD = 0, rff = 0, rdh = 0;                      // the model's own encoder_dim, feed-forward width, d_head (tensor shapes, LayerNorm width, 1/sqrt(d_head))
    bool padded = false;                               // bf16, 128 <= encoder_dim < 512 other than 256: the model runs as a zero-padded 256- / 512-wide one (set_engine_dims)
    std::vector<int> feats;   // height after each stride-2 stage: feats[0] = F1, ...
    std::map<std::string, HostTensor> host;
    std::vector<std::string> names;
    // packed weights
    int dtype = -1;
    unsigned char *blob = nullptr;
    BlobPlan plan;
    // fragment-major copies of the row-chain kernels' weight matrices (rowchain.hip.h), at the blob's offsets; derived from the blob,
    // rebuilt before the next forward whenever the blob may have changed (finalize, import, cocr_weight_blob handed out)
    unsigned char *packed = nullptr;
    bf16_t *fpack = nullptr;     // fused frontend kernel (frontend.hip.h): conv.0 A-fragments, then the depthwise block-diagonal B-fragments
    bool packed_stale = true;
    // positional tables P_l = PE Wpos_l^T, [layer][9999][heads][dhp] in the compute dtype: derived from the blob's wpos matrices on this device
    // (finalize, import, handed-out blob pointer) -- 61 of the former 104 MB of the cfg2 blob, which every rank can compute for itself
    unsigned char *ptab = nullptr;
    size_t ptab_stride = 0;
    bool ptab_stale = true;
    // cocr_share_weights: this model reads `owner`'s blob / packed copies / tables instead of holding its own (several packed copies of one
    // model, each with its workspace, for callers that keep several batches in flight: four private copies are 4 x ~100 MB, more than the
    // 256 MB Infinity Cache -- every forward then streamed its weights from HBM).  `wgen`: bumped whenever the owner's buffers may have moved.
    cocr_model *owner = nullptr;
    unsigned long long wgen = 1, seen_wgen = 0;
    // workspace
    int capN = 0, capW = 0;
    std::vector<void *> ws_allocs;
    void *z_a = nullptr, *z_b = nullptr;
    float *x = nullptr;
    void *g_lines = nullptr;           // staged graph replay (cocr_forward): library-owned copies of the caller's lines / logits
    float *g_logits = nullptr;
    void *xn = nullptr, *hid = nullptr, *q = nullptr, *k = nullptr, *vt = nullptr, *ctx = nullptr, *glu = nullptr, *dwo = nullptr;
    size_t qkv_bytes = 0;
    int vtN = -1, vtT = -1;    // shape the q/k/vt buffers were last zeroed for
    int pos_maxlen = COCR_POS_MAXLEN;                  // relative positions the P tables cover: -(max_len - 1) .. max_len - 1
    unsigned char *pre_buf = nullptr;      // line pre-processing: descriptors, tap tables, intermediates
    size_t pre_cap = 0;
    int32_t *d_lens = nullptr, *h_lens = nullptr, *d_lens_cur = nullptr;      // device / pinned-host rings of per-line lengths (upload_lens)
    int lens_slot = 0;
    int32_t *ctc_lab = nullptr;
    const float *amax_logits = nullptr;     // the logits buffer whose per-frame argmax / maximum the last forward left in ctc_lab / ctc_val (decoder epilogue)
    int amax_rows = 0;
    bool amax_ok = false;                   // the forward's launch sequence (plain or captured) ends with the argmax epilogue
    float *ctc_val = nullptr;
    size_t ctc_cap = 0;
    float *tr_pad = nullptr;            // padded models: engine-layout staging of the output layer's gradient tensors
    size_t tr_pad_cap = 0;
    int32_t *beam_bp = nullptr;
    size_t beam_cap = 0;
    int lens_cap = 0;
    int32_t *loss_d = nullptr, *loss_h = nullptr;      // cocr_ctc_loss: device / pinned-host rings of [lens | label lens | label offsets | labels]
    size_t loss_ints = 0;
    int loss_slot = 0;
    float *loss_ws = nullptr;                          // log-softmax + alpha / beta tables
    size_t loss_ws_cap = 0;
    int lastN = 0, lastT = 0;                          // shape of the last forward: its encoder output is still in `xn`
    float *tr_part = nullptr;                          // decoder backward: per-chunk partial sums of dW | db
    size_t tr_part_cap = 0;
    float *tr_state = nullptr;                         // decoder AdamW: fp32 master [W | b], then exp_avg, then exp_avg_sq
    long tr_step = 0;
    // debug / profile
    // hipGraph replay of the forward's launch sequence, keyed by the call's shapes and buffers
    bool use_graph = false;
    struct GraphEntry { const void *lines; float *logits; int N, W, dtype; hipStream_t s; hipGraphExec_t exec; };
    std::vector<GraphEntry> graphs, graph_seen;
    bool debug = false;
    unsigned long long *stamps = nullptr;   // COCR_CHAIN_STAMPS=1 (dev builds): host-visible cycle stamps of the frontend / attention / beam kernels, printed at destroy
    bool no_pad = false;         // COCR_NO_PAD=1: never run a narrow model as a zero-padded 256-wide one
    bool beam_ref = false;       // COCR_BEAM_REF=1: the exhaustive beam kernel (all beam x C candidates per frame) also for <= 256 classes
    bool no_front96 = false;     // COCR_NO_FRONT96=1: frontend conv stages as separate kernels (A/B)
    bool no_front32 = false;     // COCR_NO_FRONT32=1: 32 conv channels: the pointwise conv as a GEMM launch of its own (A/B)
    bool no_conv_mfma = false;   // COCR_NO_CONV_MFMA=1: the all-VALU fp32 frontend conv kernel also in bf16 mode (A/B)
    bool no_dw_fuse = false;     // COCR_NO_DW_FUSE=1: depthwise conv as its own launch (A/B)
    int chain_rows = 0;          // rows per workgroup of the row-chain kernels (cocr_set_chain_rows / COCR_CHAIN_ROWS); 0 = by the number of rows
    bool no_front_chain = false; // COCR_NO_FRONT_CHAIN=1: the frontend's output linear as a split-K GEMM + reduction in front of the first chain launch (A/B)
    bool ffn_probe = false;      // COCR_FFN_PROBE=1: one extra FFN-only row-chain launch per forward (measurement; results discarded)
    int att_resident_min = 192;  // COCR_ATT_RESIDENT_MIN: fewest workgroups for which the LDS-resident attention kernel is chosen (tests: 1)
    bool att_resident_long = false;   // COCR_ATT_RESIDENT_LONG=1: the LDS-resident attention kernel also for lines of more than 320 frames (key passes; A/B)
    bool att_tiled = false;      // COCR_ATT_TILED=1: the tiled attention kernel also for lines of <= 320 frames (A/B against the LDS-resident one)
    bool no_kskip = false;       // COCR_NO_KSKIP=1: zero-padded narrow models multiply their zero k-steps too (A/B)
    bool chain_xcd = true;       // COCR_CHAIN_XCD=0: row blocks in plain workgroup order (A/B)
    bool no_chain = false;       // COCR_NO_CHAIN=1: one kernel per GEMM / FFN instead of the row-local chains (A/B measurements)
    bool no_fused_ffn = false;   // COCR_NO_FUSED_FFN=1: keep the two-GEMM feed-forward (A/B measurements)
    std::map<std::string, std::pair<float *, int64_t>> taps;
    float *tapbuf = nullptr;     // debug: 4 fp32 (M, D) tap targets of the chain kernels' TAPS instantiation + one bf16 (M, D)
    size_t tapbuf_rows = 0;
    TrainState *train = nullptr;   // cocr_train_begin .. cocr_train_end (train_api.hip.h)
    bool profile = false;
    std::vector<ProfRec> prof;
    std::vector<hipEvent_t> ev_pool;
};

static const char *FAMILIES[] = {"frontend_fused", "frontend_conv12", "frontend_dw", "gemm_front_pw", "gemm_front_out", "layernorm",
                                 "gemm_ffn_up", "gemm_ffn_down", "ffn_fused", "chain_ffn_qkv", "chain_front_ffn_qkv", "chain_attn_out_glu", "chain_pw2_ffn_ffn_qkv", "chain_pw2_ffn", "gemm_qkv", "attention", "gemm_attn_out", "gemm_glu",
                                 "dwconv", "gemm_pw2", "gemm_decoder", "ctc_greedy", "ctc_beam", "ctc_loss", "ffn_probe", "event_pair_overhead"};
enum { FAM_FRONT96, FAM_CONV12, FAM_FDW, FAM_FPW, FAM_FOUT, FAM_LN, FAM_FFN_UP, FAM_FFN_DOWN, FAM_FFN_FUSED, FAM_CH_FIRST, FAM_CH_FRONT, FAM_CH_A, FAM_CH_B, FAM_CH_LAST, FAM_QKV, FAM_ATTN, FAM_AOUT, FAM_GLU,
       FAM_DW, FAM_PW2, FAM_DEC, FAM_GREEDY, FAM_BEAM, FAM_LOSS, FAM_FFN_PROBE, FAM_EMPTY, FAM_COUNT };

static int out_len1(int l) { return l >= 1 ? (l - 1) / 2 + 1 : 0; }

extern "C" int32_t cocr_out_len(int32_t in_len, int32_t subsampling_factor) {
    int n = 0;
    for (int f = subsampling_factor; f > 1; f >>= 1) ++n;
    for (int i = 0; i < n; ++i) in_len = out_len1(in_len);
    return in_len;
}

static void add_name(cocr_model *m, const std::string &n) { m->names.push_back(n); m->host[n]; }

extern "C" int cocr_create(const cocr_hparams *hp, int device, cocr_model **out) {
    if (!hp || !out) return fail(COCR_EINVAL, "null argument");
    if (hp->num_classes < 1 || hp->height < 1 || hp->encoder_dim < 1 || hp->num_encoder_layers < 1 || hp->num_attention_heads < 1)
        return fail(COCR_EINVAL, "non-positive hyper-parameter");
    if (hp->encoder_dim % hp->num_attention_heads) return fail(COCR_EINVAL, "d_model %% num_heads should be zero.");
    if ((hp->conv_kernel_size - 1) % 2 || hp->conv_kernel_size < 1) return fail(COCR_EINVAL, "kernel_size should be a odd number for 'SAME' padding");
    if (hp->conv_expansion_factor != 2) return fail(COCR_EINVAL, "Currently, Only Supports expansion_factor 2");
    int sf = hp->subsampling_factor, snum = 0;
    if (sf < 2 || (sf & (sf - 1))) return fail(COCR_EINVAL, "Sampling factor should be a power of 2.");
    for (int f = sf; f > 1; f >>= 1) ++snum;
    if (snum < 1) return fail(COCR_EINVAL, "subsampling_factor must be at least 2");
    if (hp->encoder_dim % 16) return fail(COCR_EUNSUPPORTED, "encoder_dim must be a multiple of 16 (GLU tile pairing, 16-byte rows)");
    if (hp->subsampling_conv_channels % 8) return fail(COCR_EUNSUPPORTED, "subsampling_conv_channels must be a multiple of 8");
    if (hp->encoder_dim > COCR_LN_MAX_D) return fail(COCR_EUNSUPPORTED, "encoder_dim > 1024");
    const int dh = hp->encoder_dim / hp->num_attention_heads;
    if (dh > 128) return fail(COCR_EUNSUPPORTED, "d_head > 128");
    cocr_model *m = new cocr_model();
    m->hp = *hp;
    m->device = device;
    m->D = hp->encoder_dim; m->C = hp->subsampling_conv_channels; m->L = hp->num_encoder_layers;
    m->heads = hp->num_attention_heads; m->dh = dh; m->dhp = round_up(dh, 32);
    m->ff = hp->feed_forward_expansion_factor * hp->encoder_dim; m->ksz = hp->conv_kernel_size;
    m->rD = m->D; m->rff = m->ff; m->rdh = m->dh;
    m->ncls = hp->num_classes; m->H = hp->height; m->snum = snum;
    { const char *e = getenv("COCR_NO_PAD"); m->no_pad = e && e[0] == '1'; }
    { const char *e = getenv("COCR_NO_FUSED_FFN"); m->no_fused_ffn = e && e[0] == '1'; }
    { const char *e = getenv("COCR_NO_CHAIN"); m->no_chain = e && e[0] == '1'; }
    { const char *e = getenv("COCR_NO_FRONT_CHAIN"); m->no_front_chain = e && e[0] == '1'; }
    { const char *e = getenv("COCR_FFN_PROBE"); m->ffn_probe = e && e[0] == '1'; }
    { const char *e = getenv("COCR_ATT_TILED"); m->att_tiled = e && e[0] == '1'; }
    { const char *e = getenv("COCR_ATT_RESIDENT_LONG"); m->att_resident_long = e && e[0] == '1'; }
    { const char *e = getenv("COCR_CHAIN_XCD"); if (e) m->chain_xcd = e[0] != '0'; }
    { const char *e = getenv("COCR_NO_KSKIP"); m->no_kskip = e && e[0] == '1'; }
    { const char *e = getenv("COCR_ATT_RESIDENT_MIN"); if (e) m->att_resident_min = atoi(e); }
    { const char *e = getenv("COCR_CHAIN_ROWS"); m->chain_rows = e ? atoi(e) : 0; }
    { const char *e = getenv("COCR_NO_DW_FUSE"); m->no_dw_fuse = e && e[0] == '1'; }
    { const char *e = getenv("COCR_NO_CONV_MFMA"); m->no_conv_mfma = e && e[0] == '1'; }
    { const char *e = getenv("COCR_NO_FRONT96"); m->no_front96 = e && e[0] == '1'; }
    { const char *e = getenv("COCR_NO_FRONT32"); m->no_front32 = e && e[0] == '1'; }
    { const char *e = getenv("COCR_BEAM_REF"); m->beam_ref = e && e[0] == '1'; }
    { const char *e = getenv("COCR_CHAIN_STAMPS"); if (e && e[0] == '1') { (void)hipHostMalloc((void **)&m->stamps, 4096 * 8); memset(m->stamps, 0, 4096 * 8); } }
    int f = hp->height;
    for (int i = 0; i < snum; ++i) { f = out_len1(f); m->feats.push_back(f); }
    // expected state-dict entries, reference key names
    char buf[256];
    add_name(m, "encoder.conv_subsample.conv.0.weight");
    add_name(m, "encoder.conv_subsample.conv.0.bias");
    for (int s = 0, idx = 2; s < snum - 1; ++s, idx += 3) {
        for (int j = 0; j < 2; ++j) {
            snprintf(buf, sizeof buf, "encoder.conv_subsample.conv.%d.weight", idx + j); add_name(m, buf);
            snprintf(buf, sizeof buf, "encoder.conv_subsample.conv.%d.bias", idx + j); add_name(m, buf);
        }
    }
    add_name(m, "encoder.conv_subsample.out.0.weight");
    add_name(m, "encoder.conv_subsample.out.0.bias");
    for (int l = 0; l < m->L; ++l) {
        auto nm = [&](const char *suffix) { snprintf(buf, sizeof buf, "encoder.layers.%d.sequential.%s", l, suffix); add_name(m, buf); };
        for (int w = 0; w < 2; ++w) {
            const char *pre = w == 0 ? "0" : "3";
            for (const char *s : {"module.sequential.0.weight", "module.sequential.0.bias", "module.sequential.1.linear.weight",
                                  "module.sequential.1.linear.bias", "module.sequential.4.linear.weight", "module.sequential.4.linear.bias"}) {
                char b2[200]; snprintf(b2, sizeof b2, "%s.%s", pre, s); nm(b2);
            }
        }
        for (const char *s : {"1.module.layer_norm.weight", "1.module.layer_norm.bias", "1.module.attention.u_bias", "1.module.attention.v_bias",
                              "1.module.attention.query_proj.linear.weight", "1.module.attention.query_proj.linear.bias",
                              "1.module.attention.key_proj.linear.weight", "1.module.attention.key_proj.linear.bias",
                              "1.module.attention.value_proj.linear.weight", "1.module.attention.value_proj.linear.bias",
                              "1.module.attention.pos_proj.linear.weight", "1.module.attention.out_proj.linear.weight",
                              "1.module.attention.out_proj.linear.bias",
                              "2.module.sequential.0.weight", "2.module.sequential.0.bias", "2.module.sequential.2.conv.weight",
                              "2.module.sequential.2.conv.bias", "2.module.sequential.4.conv.weight", "2.module.sequential.5.weight",
                              "2.module.sequential.5.bias", "2.module.sequential.5.running_mean", "2.module.sequential.5.running_var",
                              "2.module.sequential.7.conv.weight", "2.module.sequential.7.conv.bias", "4.weight", "4.bias"})
            nm(s);
    }
    add_name(m, "decoder.weight");
    add_name(m, "decoder.bias");
    *out = m;
    return COCR_OK;
}

static void free_workspace(cocr_model *m) {
    for (void *p : m->ws_allocs) (void)hipFree(p);
    m->ws_allocs.clear();
    m->capN = m->capW = 0;
    m->vtN = m->vtT = -1;
}
static void clear_taps(cocr_model *m) {
    for (auto &kv : m->taps) (void)hipFree(kv.second.first);
    m->taps.clear();
    if (m->tapbuf) { (void)hipFree(m->tapbuf); m->tapbuf = nullptr; m->tapbuf_rows = 0; }
}

static void train_free(cocr_model *m);
extern "C" void cocr_destroy(cocr_model *m) {
    if (!m) return;
    (void)hipSetDevice(m->device);
    train_free(m);
    free_workspace(m);
    clear_taps(m);
    if (!m->owner) {
        if (m->blob) (void)hipFree(m->blob);
        if (m->packed) (void)hipFree(m->packed);
        if (m->ptab) (void)hipFree(m->ptab);
        if (m->fpack) (void)hipFree(m->fpack);
    }
    if (m->pre_buf) (void)hipFree(m->pre_buf);
    if (m->stamps) {
        (void)hipDeviceSynchronize();
        if (m->stamps[1024]) {                          // row-chain kernel (dominant form), workgroup 7: per wave, cycles between stamps
            for (int w = 0; w < 8; ++w) {
                const unsigned long long *q = m->stamps + 1024 + 64 * w;
                fprintf(stderr, "chain wave %d (start +%llu):", w, q[0] - m->stamps[1024]);
                for (int i = 1; i < 64 && q[i]; ++i) fprintf(stderr, " %llu", q[i] - q[i - 1]);
                fprintf(stderr, "  total %llu\n", [&] { int i = 1; while (i < 64 && q[i]) ++i; return q[i - 1] - q[0]; }());
            }
        }
        fprintf(stderr, "frontend stamps:");
        for (int i = 129; i < 192 && m->stamps[i]; ++i) fprintf(stderr, " %llu", m->stamps[i] - m->stamps[128]);
        fprintf(stderr, "\nbeam walk cycles of the stay wave, then of extension wave 0 (pairs + reads, keys, barrier + ranks, update, tail):");
        for (int i = 240; i < 250 && m->stamps[i]; ++i) fprintf(stderr, " %llu", m->stamps[i]);
        fprintf(stderr, "\nattention stamps:");
        for (int i = 193; i < 240 && m->stamps[i]; ++i) fprintf(stderr, " %llu", m->stamps[i] - m->stamps[192]);
        fprintf(stderr, "\n");
        {   // per-workgroup (start, end, hardware id) of the stamped attention launch: residency and tail
            const unsigned long long *w = m->stamps + 192 + 64;
            unsigned long long t0 = ~0ull, t1 = 0;
            int nwg = 0;
            for (int i = 0; i < 1200 && w[3 * i]; ++i) { t0 = std::min(t0, w[3 * i]); t1 = std::max(t1, w[3 * i + 1]); nwg = i + 1; }
            if (nwg) {
                fprintf(stderr, "attention workgroups %d, span %.2f us (100 MHz ticks)\n", nwg, (t1 - t0) * 0.01);
                std::map<unsigned long long, int> per_cu;
                int hist_start[32] = {0};
                double dur = 0;
                for (int i = 0; i < nwg; ++i) {
                    const unsigned long long hw = w[3 * i + 2];
                    const unsigned long long cu = ((hw >> 32) << 16) | ((hw >> 8) & 0xff) | (((hw >> 13) & 7) << 8);
                    per_cu[cu]++;
                    hist_start[std::min<unsigned long long>((w[3 * i] - t0) / 100, 31)]++;
                    dur += (w[3 * i + 1] - w[3 * i]) * 0.01;
                }
                fprintf(stderr, "  mean workgroup duration %.2f us; distinct CUs %zu; start-time histogram (1 us bins):", dur / nwg, per_cu.size());
                for (int b = 0; b < 32; ++b) fprintf(stderr, " %d", hist_start[b]);
                int cnt[8] = {0};
                for (auto &kv : per_cu) cnt[std::min(kv.second, 7)]++;
                fprintf(stderr, "\n  CUs by number of workgroups received (0..7+):");
                for (int b = 0; b < 8; ++b) fprintf(stderr, " %d", cnt[b]);
                fprintf(stderr, "\n");
            }
        }
        (void)hipHostFree(m->stamps);
    }
    if (m->d_lens) (void)hipFree(m->d_lens);
    if (m->h_lens) (void)hipHostFree(m->h_lens);
    if (m->ctc_lab) (void)hipFree(m->ctc_lab);
    if (m->ctc_val) (void)hipFree(m->ctc_val);
    if (m->beam_bp) (void)hipFree(m->beam_bp);
    if (m->tr_pad) (void)hipFree(m->tr_pad);
    if (m->loss_d) (void)hipFree(m->loss_d);
    if (m->loss_h) (void)hipHostFree(m->loss_h);
    if (m->loss_ws) (void)hipFree(m->loss_ws);
    if (m->tr_part) (void)hipFree(m->tr_part);
    if (m->tr_state) (void)hipFree(m->tr_state);
    for (auto &g : m->graphs) (void)hipGraphExecDestroy(g.exec);
    for (auto &r : m->prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    for (auto e : m->ev_pool) (void)hipEventDestroy(e);
    delete m;
}

extern "C" int cocr_set_tensor(cocr_model *m, const char *name, const void *host, int dtype, int ndim, const int64_t *shape) {
    if (!m || !name || !host) return fail(COCR_EINVAL, "null argument");
    std::string n(name);
    if (n.size() > 20 && n.compare(n.size() - 19, 19, "num_batches_tracked") == 0) return COCR_OK;   // BatchNorm counter: unused in eval
    auto it = m->host.find(n);
    if (it == m->host.end()) return fail(COCR_EINVAL, "unexpected key '%s'", name);
    if (dtype != COCR_F32) return fail(COCR_EINVAL, "tensor '%s': only float32 state is accepted", name);
    int64_t cnt = 1;
    for (int i = 0; i < ndim; ++i) cnt *= shape[i];
    it->second.shape.assign(shape, shape + ndim);
    it->second.data.assign((const float *)host, (const float *)host + cnt);
    it->second.set = true;
    return COCR_OK;
}

extern "C" int cocr_missing_tensors(cocr_model *m, char *buf, size_t buflen) {
    if (!m) return fail(COCR_EINVAL, "null argument");
    int cnt = 0;
    std::string s;
    for (auto &n : m->names)
        if (!m->host[n].set) { ++cnt; s += n; s += '\n'; }
    if (buf && buflen) { strncpy(buf, s.c_str(), buflen - 1); buf[buflen - 1] = 0; }
    return cnt;
}

// ------------------------------------------------------------------------------------ blob
static size_t esize(int dtype) { return dtype == COCR_BF16 ? 2 : 4; }

static BlobPlan make_plan(const cocr_model *m, int dtype) {
    BlobPlan p;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
    const size_t es = esize(dtype);
    const int D = m->D, C = m->C, ff = m->ff;
    p.w0 = take((size_t)C * 9 * 4); p.b0 = take((size_t)C * 4);
    for (int s = 0; s < m->snum - 1; ++s) {
        StageW st;
        st.dw_w = take((size_t)C * 9 * 4); st.dw_b = take((size_t)C * 4);
        st.pw_w = take((size_t)C * C * es); st.pw_b = take((size_t)C * 4);
        p.stages.push_back(st);
    }
    const int F = m->feats.back();
    p.wout = take((size_t)D * F * C * es); p.bout = take((size_t)D * 4);
    for (int l = 0; l < m->L; ++l) {
        LayerW w;
        for (int i = 0; i < 2; ++i) {
            w.ffn[i].ln_g = take(D * 4); w.ffn[i].ln_b = take(D * 4);
            w.ffn[i].w1 = take((size_t)ff * D * es); w.ffn[i].b1 = take((size_t)ff * 4);
            w.ffn[i].w2 = take((size_t)D * ff * es); w.ffn[i].b2 = take(D * 4);
        }
        w.a_ln_g = take(D * 4); w.a_ln_b = take(D * 4);
        w.wqkv = take((size_t)3 * D * D * es); w.bqkv = take((size_t)3 * D * 4);
        w.ub = take(D * 4); w.vb = take(D * 4);
        w.wpos = take((size_t)D * D * 4);
        w.wo = take((size_t)D * D * es); w.bo = take(D * 4);
        w.c_ln_g = take(D * 4); w.c_ln_b = take(D * 4);
        w.wpw1 = take((size_t)2 * D * D * es); w.bpw1 = take((size_t)2 * D * 4);
        w.dww = take((size_t)m->ksz * D * 4); w.dwb = take(D * 4);
        w.wpw2 = take((size_t)D * D * es); w.bpw2 = take(D * 4);
        w.f_ln_g = take(D * 4); w.f_ln_b = take(D * 4);
        p.layers.push_back(w);
    }
    p.wdec = take((size_t)m->ncls * D * es); p.bdec = take((size_t)m->ncls * 4);
    p.total = off;
    return p;
}

static uint16_t f32_to_bf16(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

// writes rows x cols of src (row i taken from src row map(i)) as compute dtype
static void put_matrix(unsigned char *dst, int dtype, const float *src, int rows, int cols, const std::vector<int> *rowmap = nullptr,
                       const std::vector<int> *colmap = nullptr) {
    for (int r = 0; r < rows; ++r) {
        const float *s = src + (size_t)(rowmap ? (*rowmap)[r] : r) * cols;
        for (int c = 0; c < cols; ++c) {
            const float v = s[colmap ? (*colmap)[c] : c];
            if (dtype == COCR_BF16) ((uint16_t *)dst)[(size_t)r * cols + c] = f32_to_bf16(v);
            else ((float *)dst)[(size_t)r * cols + c] = v;
        }
    }
}
static void put_f32(unsigned char *dst, const float *src, size_t n) { memcpy(dst, src, n * 4); }

static int expect_shape(const cocr_model *m, const std::string &name, std::initializer_list<int64_t> shape, const HostTensor **out) {
    auto it = m->host.find(name);
    if (it == m->host.end() || !it->second.set) return fail(COCR_ESTATE, "missing tensor '%s'", name.c_str());
    if (it->second.shape != std::vector<int64_t>(shape)) {
        std::string got;
        for (auto v : it->second.shape) got += std::to_string(v) + ",";
        return fail(COCR_EINVAL, "size mismatch for %s: got (%s)", name.c_str(), got.c_str());
    }
    *out = &it->second;
    return COCR_OK;
}

static int ensure_ptab(cocr_model *m, hipStream_t s);

// The row-chain kernels exist for encoder_dim 256 and 512.  A narrower model (the reference's default: encoder_dim 144, 4 heads of 36,
// feed-forward 576) ran one kernel per product and was SLOWER than the 256-wide model.  In bf16 mode such a model (and one between 256
// and 512 wide) is run as a zero-padded 256-wide (512-wide) one: every tensor is embedded in the 256 / 768-wide layout at pack time (model dimension: identity + zeros; head dimension:
// head h at columns [64 h, 64 h + d_head); feed-forward: identity + zeros), so every padded activation column is exactly zero at every
// stage (zero weights and biases, zero LayerNorm gain and shift, silu(0) = 0, 0 * sigmoid(0) = 0) and the real columns see the same
// sums.  What does not follow from the padding is stated separately: LayerNorm divides by the REAL width (the statistics are raw
// moments: zeros add nothing), the attention scale is 1 / sqrt(real d_head), the sinusoids use the real encoder_dim.
static void free_workspace(cocr_model *m);
static int set_engine_dims(cocr_model *m, int dtype) {
    const int wide = m->rD < 256 ? 256 : 512;            // 128 <= encoder_dim < 256 -> 256; 256 < encoder_dim < 512 -> 512
    const int slot = m->heads > 0 && wide % m->heads == 0 ? wide / m->heads : 0;
    const bool pad = dtype == COCR_BF16 && !m->no_pad && m->rD >= 128 && m->rD < 512 && m->rD != 256 && slot >= m->rdh && slot % 32 == 0 && slot <= 128 &&
                     round_up(m->rff, 256) <= (wide == 256 ? 1024 : 2048);
    const int D = pad ? wide : m->rD, ff = pad ? round_up(m->rff, 256) : m->rff, dh = pad ? slot : m->rdh, dhp = pad ? slot : round_up(m->rdh, 32);
    if (D != m->D || ff != m->ff || dh != m->dh || dhp != m->dhp) {      // workspace and captured launches belong to the old layout
        HIP_TRY(hipDeviceSynchronize());
        for (auto &g : m->graphs) (void)hipGraphExecDestroy(g.exec);
        m->graphs.clear(); m->graph_seen.clear();
        free_workspace(m);
        m->capN = m->capW = 0;
    }
    m->D = D; m->ff = ff; m->dh = dh; m->dhp = dhp; m->padded = pad;
    return COCR_OK;
}

static int alloc_blob(cocr_model *m, int dtype) {
    if (dtype != COCR_BF16 && dtype != COCR_F32) return fail(COCR_EINVAL, "compute dtype must be COCR_BF16 or COCR_F32");
    HIP_TRY(hipSetDevice(m->device));
    { const int rc = set_engine_dims(m, dtype); if (rc) return rc; }
    if (m->owner) { m->owner = nullptr; m->blob = m->packed = m->ptab = nullptr; m->fpack = nullptr; }      // weights of its own again
    m->wgen++;
    if (m->blob) { (void)hipFree(m->blob); m->blob = nullptr; }
    if (m->packed) { (void)hipFree(m->packed); m->packed = nullptr; }
    if (m->ptab) { (void)hipFree(m->ptab); m->ptab = nullptr; }
    m->ptab_stale = true;
    if (m->fpack) { (void)hipFree(m->fpack); m->fpack = nullptr; }
    if (m->tr_state) { (void)hipFree(m->tr_state); m->tr_state = nullptr; m->tr_step = 0; }      // optimizer state belongs to the old weights
    m->packed_stale = true;
    m->plan = make_plan(m, dtype);
    m->dtype = dtype;
    HIP_TRY(hipMalloc((void **)&m->blob, m->plan.total));
    HIP_TRY(hipMemset(m->blob, 0, m->plan.total));
    return COCR_OK;
}

extern "C" int cocr_finalize_empty(cocr_model *m, int compute_dtype) {
    if (!m) return fail(COCR_EINVAL, "null argument");
    int rc = alloc_blob(m, compute_dtype);
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    return COCR_OK;
}

extern "C" int cocr_share_weights(cocr_model *m, cocr_model *owner) {
    if (!m || !owner || m == owner) return fail(COCR_EINVAL, "two different models expected");
    if (owner->owner) return fail(COCR_EINVAL, "the owner itself shares another model's weights");
    if (owner->dtype < 0 || !owner->blob) return fail(COCR_ESTATE, "the owner is not finalized");
    if (m->device != owner->device || memcmp(&m->hp, &owner->hp, sizeof m->hp) != 0) return fail(COCR_EINVAL, "models of the same hyper-parameters on the same device expected");
    if (m->train || owner->train) return fail(COCR_ESTATE, "not while a training state exists");
    HIP_TRY(hipSetDevice(m->device));
    HIP_TRY(hipDeviceSynchronize());
    if (m->dtype != owner->dtype || !m->blob || m->owner) { const int rc = alloc_blob(m, owner->dtype); if (rc) return rc; }     // dims, plan
    (void)hipFree(m->blob);
    if (m->packed) (void)hipFree(m->packed);
    if (m->ptab) (void)hipFree(m->ptab);
    if (m->fpack) (void)hipFree(m->fpack);
    m->blob = m->packed = m->ptab = nullptr; m->fpack = nullptr;
    m->owner = owner;
    m->seen_wgen = 0;                               // the next forward adopts the owner's pointers
    m->blob = owner->blob;                          // ("finalized" tests look at it)
    for (auto &g : m->graphs) (void)hipGraphExecDestroy(g.exec);
    m->graphs.clear(); m->graph_seen.clear();
    return COCR_OK;
}

extern "C" int cocr_weight_blob(cocr_model *m, void **device_ptr, size_t *bytes) {
    if (!m || !device_ptr || !bytes) return fail(COCR_EINVAL, "null argument");
    if (!m->blob) return fail(COCR_ESTATE, "model not finalized");
    if (m->owner) return fail(COCR_ESTATE, "this model shares another model's weights: address the owner");
    *device_ptr = m->blob;
    *bytes = m->plan.total;
    m->packed_stale = m->ptab_stale = true;          // the caller may write through the pointer: derived copies are rebuilt by the next forward
    return COCR_OK;
}

// Copies between the packed blob and a caller-owned device buffer (a collective library's registered / framework-owned
// memory): rank 0 exports, broadcasts, the other ranks import.  Stream-ordered on `stream`.
extern "C" int cocr_blob_export(cocr_model *m, void *dst_device, size_t bytes, void *stream) {
    if (!m || !dst_device) return fail(COCR_EINVAL, "null argument");
    if (!m->blob) return fail(COCR_ESTATE, "model not finalized");
    if (m->owner) return fail(COCR_ESTATE, "this model shares another model's weights: address the owner");      // (its own `blob` is a view that the owner may have re-allocated)
    if (bytes != m->plan.total) return fail(COCR_EINVAL, "blob is %zu bytes, buffer %zu", m->plan.total, bytes);
    HIP_TRY(hipSetDevice(m->device));
    HIP_TRY(hipMemcpyAsync(dst_device, m->blob, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return COCR_OK;
}
extern "C" int cocr_blob_import(cocr_model *m, const void *src_device, size_t bytes, void *stream) {
    if (!m || !src_device) return fail(COCR_EINVAL, "null argument");
    if (!m->blob) return fail(COCR_ESTATE, "model not finalized");
    if (bytes != m->plan.total) return fail(COCR_EINVAL, "blob is %zu bytes, buffer %zu", m->plan.total, bytes);
    if (m->owner) return fail(COCR_ESTATE, "this model shares another model's weights: address the owner");
    HIP_TRY(hipSetDevice(m->device));
    HIP_TRY(hipMemcpyAsync(m->blob, src_device, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    m->packed_stale = m->ptab_stale = true;
    if (m->tr_state) { (void)hipFree(m->tr_state); m->tr_state = nullptr; m->tr_step = 0; }
    return COCR_OK;
}

// dst (vrows x vcols, compute dtype) <- src (.. x src_cols): element (r, c) = src[rmap[r]][cmap[c]], zero where a map says -1
static void put_mapped(unsigned char *dst, int dtype, const float *src, int src_cols, const std::vector<int> &rmap, const std::vector<int> &cmap) {
    const size_t vcols = cmap.size();
    for (size_t r = 0; r < rmap.size(); ++r)
        for (size_t c = 0; c < vcols; ++c) {
            const float v = (rmap[r] >= 0 && cmap[c] >= 0) ? src[(size_t)rmap[r] * src_cols + cmap[c]] : 0.0f;
            if (dtype == COCR_BF16) ((uint16_t *)dst)[r * vcols + c] = f32_to_bf16(v);
            else ((float *)dst)[r * vcols + c] = v;
        }
}
static void put_vec(unsigned char *dst, const float *src, const std::vector<int> &map) {
    for (size_t i = 0; i < map.size(); ++i) ((float *)dst)[i] = map[i] >= 0 ? src[map[i]] : 0.0f;
}

extern "C" int cocr_finalize(cocr_model *m, int dtype) {
    if (!m) return fail(COCR_EINVAL, "null argument");
    if (dtype != COCR_BF16 && dtype != COCR_F32) return fail(COCR_EINVAL, "compute dtype must be COCR_BF16 or COCR_F32");
    HIP_TRY(hipSetDevice(m->device));
    int rc;
    if ((rc = set_engine_dims(m, dtype))) return rc;
    // tensor shapes: the model's own (rD, rff, h x rdh); blob layout: the engine's (D, ff, h x dh) -- the same unless the model is padded
    const int rD = m->rD, rff = m->rff, rdh = m->rdh, D = m->D, C = m->C, ff = m->ff, k = m->ksz, h = m->heads, dh = m->dh;
    std::vector<int> Mm(D), Fm(ff), Am(D), Gm((size_t)2 * D), ident_cls(m->ncls);      // engine index -> model index (-1: zero)
    for (int c = 0; c < D; ++c) Mm[c] = c < rD ? c : -1;                                // model dimension
    for (int c = 0; c < ff; ++c) Fm[c] = c < rff ? c : -1;                              // feed-forward dimension
    for (int c = 0; c < D; ++c) Am[c] = (c / dh < h && c % dh < rdh) ? (c / dh) * rdh + c % dh : -1;      // head-major dimension: head hh at [dh hh, dh hh + rdh)
    for (int n = 0; n < 2 * D; ++n) {   // GLU interleave: packed row 32j + c <- value row 16j + c ; packed row 32j + 16 + c <- gate row D + 16j + c
        const int tn = n >> 4, c = n & 15, j = tn >> 1, v = 16 * j + c;
        Gm[n] = v < rD ? ((tn & 1) ? rD + v : v) : -1;
    }
    for (int c = 0; c < m->ncls; ++c) ident_cls[c] = c;
    const BlobPlan plan = make_plan(m, dtype);
    std::vector<unsigned char> stage(plan.total, 0);
    unsigned char *st = stage.data();
    const HostTensor *t = nullptr;
    char buf[256];
#define GET(NAME, ...)                                          \
    if ((rc = expect_shape(m, NAME, {__VA_ARGS__}, &t))) return rc;
    GET("encoder.conv_subsample.conv.0.weight", C, 1, 3, 3); put_f32(st + plan.w0, t->data.data(), (size_t)C * 9);
    GET("encoder.conv_subsample.conv.0.bias", C); put_f32(st + plan.b0, t->data.data(), C);
    for (int s = 0, idx = 2; s < m->snum - 1; ++s, idx += 3) {
        snprintf(buf, sizeof buf, "encoder.conv_subsample.conv.%d.weight", idx); GET(buf, C, 1, 3, 3); put_f32(st + plan.stages[s].dw_w, t->data.data(), (size_t)C * 9);
        snprintf(buf, sizeof buf, "encoder.conv_subsample.conv.%d.bias", idx); GET(buf, C); put_f32(st + plan.stages[s].dw_b, t->data.data(), C);
        snprintf(buf, sizeof buf, "encoder.conv_subsample.conv.%d.weight", idx + 1); GET(buf, C, C, 1, 1); put_matrix(st + plan.stages[s].pw_w, dtype, t->data.data(), C, C);
        snprintf(buf, sizeof buf, "encoder.conv_subsample.conv.%d.bias", idx + 1); GET(buf, C); put_f32(st + plan.stages[s].pw_b, t->data.data(), C);
    }
    {   // flatten order: the reference's feature index is c*F + f (convolution.py:235-236); ours is f*C + c
        const int F = m->feats.back();
        GET("encoder.conv_subsample.out.0.weight", rD, (int64_t)C * F);
        std::vector<int> colmap((size_t)F * C);
        for (int f = 0; f < F; ++f) for (int c = 0; c < C; ++c) colmap[(size_t)f * C + c] = c * F + f;
        put_mapped(st + plan.wout, dtype, t->data.data(), F * C, Mm, colmap);
        GET("encoder.conv_subsample.out.0.bias", rD); put_vec(st + plan.bout, t->data.data(), Mm);
    }
    for (int l = 0; l < m->L; ++l) {
        const LayerW &w = plan.layers[l];
        auto key = [&](const char *suffix) { snprintf(buf, sizeof buf, "encoder.layers.%d.sequential.%s", l, suffix); return std::string(buf); };
        for (int i = 0; i < 2; ++i) {
            const std::string pre = std::string(i == 0 ? "0" : "3") + ".module.sequential.";
            GET(key((pre + "0.weight").c_str()), rD); put_vec(st + w.ffn[i].ln_g, t->data.data(), Mm);
            GET(key((pre + "0.bias").c_str()), rD); put_vec(st + w.ffn[i].ln_b, t->data.data(), Mm);
            GET(key((pre + "1.linear.weight").c_str()), rff, rD); put_mapped(st + w.ffn[i].w1, dtype, t->data.data(), rD, Fm, Mm);
            GET(key((pre + "1.linear.bias").c_str()), rff); put_vec(st + w.ffn[i].b1, t->data.data(), Fm);
            GET(key((pre + "4.linear.weight").c_str()), rD, rff); put_mapped(st + w.ffn[i].w2, dtype, t->data.data(), rff, Mm, Fm);
            GET(key((pre + "4.linear.bias").c_str()), rD); put_vec(st + w.ffn[i].b2, t->data.data(), Mm);
        }
        GET(key("1.module.layer_norm.weight"), rD); put_vec(st + w.a_ln_g, t->data.data(), Mm);
        GET(key("1.module.layer_norm.bias"), rD); put_vec(st + w.a_ln_b, t->data.data(), Mm);
        const char *proj[3] = {"query", "key", "value"};
        for (int j = 0; j < 3; ++j) {
            snprintf(buf, sizeof buf, "encoder.layers.%d.sequential.1.module.attention.%s_proj.linear.weight", l, proj[j]);
            GET(std::string(buf), rD, rD); put_mapped(st + w.wqkv + (size_t)j * D * D * esize(dtype), dtype, t->data.data(), rD, Am, Mm);
            snprintf(buf, sizeof buf, "encoder.layers.%d.sequential.1.module.attention.%s_proj.linear.bias", l, proj[j]);
            GET(std::string(buf), rD); put_vec(st + w.bqkv + (size_t)j * D * 4, t->data.data(), Am);
        }
        GET(key("1.module.attention.u_bias"), h, rdh); put_vec(st + w.ub, t->data.data(), Am);
        GET(key("1.module.attention.v_bias"), h, rdh); put_vec(st + w.vb, t->data.data(), Am);
        GET(key("1.module.attention.pos_proj.linear.weight"), rD, rD); put_mapped(st + w.wpos, COCR_F32, t->data.data(), rD, Am, Mm);
        GET(key("1.module.attention.out_proj.linear.weight"), rD, rD); put_mapped(st + w.wo, dtype, t->data.data(), rD, Mm, Am);
        GET(key("1.module.attention.out_proj.linear.bias"), rD); put_vec(st + w.bo, t->data.data(), Mm);
        GET(key("2.module.sequential.0.weight"), rD); put_vec(st + w.c_ln_g, t->data.data(), Mm);
        GET(key("2.module.sequential.0.bias"), rD); put_vec(st + w.c_ln_b, t->data.data(), Mm);
        GET(key("2.module.sequential.2.conv.weight"), 2 * rD, rD, 1); put_mapped(st + w.wpw1, dtype, t->data.data(), rD, Gm, Mm);
        GET(key("2.module.sequential.2.conv.bias"), 2 * rD); put_vec(st + w.bpw1, t->data.data(), Gm);
        {   // BatchNorm (eval) folded into the depthwise taps: s = gamma / sqrt(var + eps)
            const HostTensor *wd, *g, *b, *mu, *var;
            if ((rc = expect_shape(m, key("2.module.sequential.4.conv.weight"), {rD, 1, k}, &wd))) return rc;
            if ((rc = expect_shape(m, key("2.module.sequential.5.weight"), {rD}, &g))) return rc;
            if ((rc = expect_shape(m, key("2.module.sequential.5.bias"), {rD}, &b))) return rc;
            if ((rc = expect_shape(m, key("2.module.sequential.5.running_mean"), {rD}, &mu))) return rc;
            if ((rc = expect_shape(m, key("2.module.sequential.5.running_var"), {rD}, &var))) return rc;
            float *tw = (float *)(st + w.dww), *tb = (float *)(st + w.dwb);      // (padded channels: taps and bias stay zero)
            for (int c = 0; c < rD; ++c) {
                const float s = g->data[c] / sqrtf(var->data[c] + 1e-5f);
                for (int tau = 0; tau < k; ++tau) tw[(size_t)tau * D + c] = wd->data[(size_t)c * k + tau] * s;
                tb[c] = b->data[c] - mu->data[c] * s;
            }
        }
        GET(key("2.module.sequential.7.conv.weight"), rD, rD, 1); put_mapped(st + w.wpw2, dtype, t->data.data(), rD, Mm, Mm);
        GET(key("2.module.sequential.7.conv.bias"), rD); put_vec(st + w.bpw2, t->data.data(), Mm);
        GET(key("4.weight"), rD); put_vec(st + w.f_ln_g, t->data.data(), Mm);
        GET(key("4.bias"), rD); put_vec(st + w.f_ln_b, t->data.data(), Mm);
    }
    GET("decoder.weight", m->ncls, rD); put_mapped(st + plan.wdec, dtype, t->data.data(), rD, ident_cls, Mm);
    GET("decoder.bias", m->ncls); put_f32(st + plan.bdec, t->data.data(), m->ncls);
#undef GET
    if ((rc = alloc_blob(m, dtype))) return rc;
    HIP_TRY(hipMemcpy(m->blob, st, plan.total, hipMemcpyHostToDevice));
    if ((rc = ensure_ptab(m, nullptr))) return rc;
    HIP_TRY(hipDeviceSynchronize());
    return COCR_OK;
}

// P_l = PE Wpos_l^T for all 2 max_len - 1 relative positions (max_len 5000 as in the reference, longer once a longer line has come) (embedding.py:35-56 table, attention.py:62,85 projection), computed on the device in
// fp32 from the blob's own wpos matrices and stored head-padded in the compute dtype.  Runs on `s` ahead of a forward's launches (stream
// order covers a blob import issued on the same stream), never inside a graph capture; the same kernel on the same inputs on every rank,
// so a rank that received the blob by broadcast holds bit-identical tables.
template <typename T> static int compute_pos_tables(cocr_model *m, hipStream_t s) {
    const int D = m->D, rD = m->rD, maxlen = m->pos_maxlen, R = 2 * maxlen - 1;      // (a padded model: rD sinusoids, zeros behind them)
    std::vector<float> pe((size_t)R * D, 0.0f);
    for (int r = 0; r < R; ++r) {
        const float pos = (float)(maxlen - 1 - r);           // +(max_len - 1) ... -(max_len - 1)
        for (int i = 0; i < rD; i += 2) {
            const float div = expf((float)i * (float)(-(log(10000.0) / rD)));
            const float ang = pos * div;
            pe[(size_t)r * D + i] = sinf(ang);
            if (i + 1 < rD) pe[(size_t)r * D + i + 1] = cosf(ang);
        }
    }
    float *d_pe = nullptr;
    HIP_TRY(hipMalloc((void **)&d_pe, pe.size() * 4));
    HIP_TRY(hipMemcpyAsync(d_pe, pe.data(), pe.size() * 4, hipMemcpyHostToDevice, s));
    for (int l = 0; l < m->L; ++l) {
        EpiPosTable<T> epi{(T *)(m->ptab + (size_t)l * m->ptab_stride), m->dh, m->dhp, m->heads};
        HIP_TRY(launch_gemm<float>(s, d_pe, D, (const float *)(m->blob + m->plan.layers[l].wpos), D, R, D, D, epi));
    }
    HIP_TRY(hipStreamSynchronize(s));            // (pe is host memory of this call; one-time start-up work)
    HIP_TRY(hipFree(d_pe));
    return COCR_OK;
}
static int ensure_ptab(cocr_model *m, hipStream_t s) {
    if (!m->ptab_stale) return COCR_OK;
    m->ptab_stride = (size_t)(2 * m->pos_maxlen - 1) * m->heads * m->dhp * esize(m->dtype);
    if (!m->ptab) {
        HIP_TRY(hipMalloc((void **)&m->ptab, m->ptab_stride * m->L));
        HIP_TRY(hipMemsetAsync(m->ptab, 0, m->ptab_stride * m->L, s));          // padded head dims read as zero
    }
    int rc = m->dtype == COCR_BF16 ? compute_pos_tables<bf16_t>(m, s) : compute_pos_tables<float>(m, s);
    if (rc) return rc;
    m->ptab_stale = false;
    return COCR_OK;
}

// ------------------------------------------------------------------------------------ workspace
static int ws_alloc(cocr_model *m, void **p, size_t bytes) {
    HIP_TRY(hipMalloc(p, bytes ? bytes : 256));
    m->ws_allocs.push_back(*p);
    return COCR_OK;
}

extern "C" int cocr_reserve(cocr_model *m, int N, int W) {
    if (!m) return fail(COCR_EINVAL, "null argument");
    if (m->dtype < 0) return fail(COCR_ESTATE, "model not finalized");
    if (N < 1 || W < 1) return fail(COCR_EINVAL, "empty batch");
    if (N <= m->capN && W <= m->capW) return COCR_OK;
    HIP_TRY(hipSetDevice(m->device));
    HIP_TRY(hipDeviceSynchronize());
    N = std::max(N, m->capN); W = std::max(W, m->capW);
    for (auto &g : m->graphs) (void)hipGraphExecDestroy(g.exec);     // captured launches point into the old workspace
    m->graphs.clear(); m->graph_seen.clear();
    free_workspace(m);
    const size_t es = esize(m->dtype);
    int T = W;
    for (int i = 0; i < m->snum; ++i) T = out_len1(T);
    const int Tz = m->snum >= 2 ? out_len1(out_len1(W)) : out_len1(W);      // frames after the fused first two stages (factor 2: after conv.0)
    const size_t M = (size_t)N * T, Tp = round_up(T, 64);      // q / k / v rows per (line, head): whole 64-key tiles (attention.hip.h reads them unclamped)
    int rc;
    const size_t zbytes = (size_t)N * Tz * m->feats[m->snum >= 2 ? 1 : 0] * m->C * es;
    if ((rc = ws_alloc(m, &m->z_a, zbytes))) return rc;
    if ((rc = ws_alloc(m, &m->z_b, zbytes))) return rc;
    if ((rc = ws_alloc(m, (void **)&m->x, (M + 128) * m->D * 4))) return rc;      // (+ 128 rows: the row-chain kernels keep the stream in whole row blocks of up to 96 rows)
    if ((rc = ws_alloc(m, &m->xn, M * m->D * es))) return rc;
    if ((rc = ws_alloc(m, &m->hid, M * m->ff * es))) return rc;
    m->qkv_bytes = (size_t)N * m->heads * Tp * m->dhp * es;
    if ((rc = ws_alloc(m, &m->q, m->qkv_bytes))) return rc;
    if ((rc = ws_alloc(m, &m->k, m->qkv_bytes))) return rc;
    if ((rc = ws_alloc(m, &m->vt, m->qkv_bytes))) return rc;
    if ((rc = ws_alloc(m, &m->ctx, M * m->D * es))) return rc;
    if ((rc = ws_alloc(m, &m->glu, M * m->D * es))) return rc;
    if ((rc = ws_alloc(m, &m->dwo, M * m->D * es))) return rc;
    if ((rc = ws_alloc(m, &m->g_lines, (size_t)N * m->H * W * 4))) return rc;
    if ((rc = ws_alloc(m, (void **)&m->g_logits, M * m->ncls * 4))) return rc;
    m->capN = N; m->capW = W;
    return COCR_OK;
}

// ------------------------------------------------------------------------------------ debug / profile
template <typename T> __global__ void to_f32_kernel(const T *in, float *out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = to_f32(in[i]);
}
// rows of the engine's (padded) width D -> rows of the model's own width rD: column c of the output is engine column c (model
// dimension) or, head-major, engine column (c / rdh) * dh + c % rdh
template <typename T> __global__ void tap_narrow_kernel(const T *in, float *out, size_t rows, int D, int rD, int dh, int rdh, int head_major) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < rows * rD; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / rD;
        const int c = (int)(i - r * rD);
        out[i] = to_f32(in[r * D + (head_major ? (c / rdh) * dh + c % rdh : c)]);
    }
}
template <typename T> static int tap(cocr_model *m, hipStream_t s, const std::string &name, const T *src, size_t n) {
    if (!m->debug) return COCR_OK;
    // a padded model (set_engine_dims): the taps show the model's own columns.  Which taps are rows of D, and in which column order,
    // follows from their names: q / k / v and the frontend's channel-last tensors have layouts of their own
    auto ends = [&](const char *suf) { const size_t l = strlen(suf); return name.size() >= l && name.compare(name.size() - l, l, suf) == 0; };
    const bool own_layout = ends(".q") || ends(".k") || ends(".v") || ends(".z2") || ends(".z3");
    const bool narrow = m->padded && !own_layout && n % (size_t)m->D == 0;
    const size_t nout = narrow ? n / m->D * m->rD : n;
    float *dst = nullptr;
    HIP_TRY(hipMalloc((void **)&dst, nout * 4));
    if (narrow) hipLaunchKernelGGL((tap_narrow_kernel<T>), dim3(256), dim3(256), 0, s, src, dst, n / m->D, m->D, m->rD, m->dh, m->rdh, ends(".ctx") ? 1 : 0);
    else hipLaunchKernelGGL((to_f32_kernel<T>), dim3(256), dim3(256), 0, s, src, dst, n);
    HIP_TRY(hipStreamSynchronize(s));
    auto it = m->taps.find(name);
    if (it != m->taps.end()) (void)hipFree(it->second.first);
    m->taps[name] = {dst, (int64_t)nout};
    return COCR_OK;
}

extern "C" int cocr_set_debug(cocr_model *m, int on) {
    if (!m) return fail(COCR_EINVAL, "null argument");
    m->debug = on != 0;
    if (!on) clear_taps(m);
    return COCR_OK;
}
extern "C" int cocr_debug_tap(cocr_model *m, const char *name, float *host_out, int64_t max_elems, int64_t *n_elems) {
    if (!m || !name) return fail(COCR_EINVAL, "null argument");
    auto it = m->taps.find(name);
    if (it == m->taps.end()) return fail(COCR_EINVAL, "no tap '%s' (debug off, or stage not run)", name);
    if (n_elems) *n_elems = it->second.second;
    if (host_out) {
        if (max_elems < it->second.second) return fail(COCR_EINVAL, "tap '%s' has %lld elements", name, (long long)it->second.second);
        HIP_TRY(hipMemcpy(host_out, it->second.first, it->second.second * 4, hipMemcpyDeviceToHost));
    }
    return COCR_OK;
}

struct ProfScope {
    cocr_model *m; hipStream_t s; ProfRec r; bool on;
    ProfScope(cocr_model *m_, hipStream_t s_, int fam) : m(m_), s(s_), on(m_->profile) {
        if (!on) return;
        r.fam = fam;
        auto get = [&]() { hipEvent_t e; if (m->ev_pool.empty()) { (void)hipEventCreate(&e); } else { e = m->ev_pool.back(); m->ev_pool.pop_back(); } return e; };
        r.a = get(); r.b = get();
        (void)hipEventRecord(r.a, s);
    }
    ~ProfScope() { if (on) { (void)hipEventRecord(r.b, s); m->prof.push_back(r); } }
};

extern "C" int cocr_profile(cocr_model *m, int on) {
    if (!m) return fail(COCR_EINVAL, "null argument");
    m->profile = on != 0;
    for (auto &r : m->prof) { m->ev_pool.push_back(r.a); m->ev_pool.push_back(r.b); }
    m->prof.clear();
    return COCR_OK;
}
extern "C" int cocr_profile_read(cocr_model *m, char *names, size_t names_len, double *ms, int64_t *launches, int max_entries) {
    if (!m) return fail(COCR_EINVAL, "null argument");
    HIP_TRY(hipSetDevice(m->device));
    HIP_TRY(hipDeviceSynchronize());
    double sum[FAM_COUNT] = {0};
    int64_t cnt[FAM_COUNT] = {0};
    for (auto &r : m->prof) {
        float t = 0.f;
        HIP_TRY(hipEventElapsedTime(&t, r.a, r.b));
        sum[r.fam] += t; cnt[r.fam]++;
    }
    std::string s;
    int n = 0;
    for (int f = 0; f < FAM_COUNT && n < max_entries; ++f) {
        if (!cnt[f]) continue;
        s += FAMILIES[f]; s += '\n';
        ms[n] = sum[f] / (double)cnt[f]; launches[n] = cnt[f];
        ++n;
    }
    if (names && names_len) { strncpy(names, s.c_str(), names_len - 1); names[names_len - 1] = 0; }
    return n;
}

// ------------------------------------------------------------------------------------ forward
#define LAUNCH_CHECK() HIP_TRY(hipGetLastError())

#define GEMM_TRY(call)                                                                                 \
    do {                                                                                               \
        hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess) return fail(COCR_EHIP, "kernel launch failed: %s (%s:%d)", hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

template <typename T, int DHP>
static hipError_t launch_attention(hipStream_t s, int N, const T *q, const T *k, const T *v, const T *ptab, const float *ub,
                                   const float *vb, T *ctx, int Tn, int Tp, int heads, int dh, float scale, int pos_center, unsigned long long *stamps = nullptr,
                                   bool tiled_only = false, int resident_min = 192, bool resident_long = false) {
    if constexpr (sizeof(T) == 2 && DHP == 64) {
        // lines of at most 320 frames (the metric's: 300): K, V and the band of a (line, head) resident in LDS, no barriers in the key loop
        // ... when there are enough (line, head) pairs to give most CUs one of its workgroups: a small batch is served faster by the tiled
        // kernel's five workgroups per (line, head) on CUs of their own (B = 1: 0.83 ms per forward with this kernel against 0.76 ms)
        const int ntiles = ceil_div(Tn, 16), nqb = ceil_div(ntiles, AF_QT);
        // Longer lines (`resident_long`, COCR_ATT_RESIDENT_LONG=1: measured, not the default) walk their keys in passes of at most AF_TK: at the wide
        // model's 600-frame lines the tiled kernel is the faster one (83 us against 91 us per launch of 32 lines x 8 heads: its staging is
        // amortised over ten key tiles there and three of its workgroups share a CU; 4.6 k against 4.2 k query-key pairs per us and CU).
        if (dh == DHP && (Tn <= AF_TK || resident_long) && !tiled_only && !stamps && nqb * N * heads >= resident_min) {
            // (lines of more than AF_TK frames: their keys in passes of at most AF_TK, balanced -- 600 frames = 2 x 320, 700 = 3 x 256)
            const int Tk0 = round_up(Tn, 64), npass = ceil_div(Tk0, AF_TK), Tk = round_up(ceil_div(Tk0, npass), 64);
            const int ntw = ceil_div(ntiles, nqb);
            const size_t lds = (size_t)(2 * Tk + 16 * ntw + Tk) * 128 + AF_WAVES * AF_SROWS * AF_SK * sizeof(float);
            auto kern = npass > 1 ? relpos_attention_full_kernel<true> : relpos_attention_full_kernel<false>;
            hipError_t e = raise_lds_limit((const void *)kern, lds);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(kern, dim3(nqb, N * heads), dim3(64 * AF_WAVES), lds, s, (const bf16_t *)q, (const bf16_t *)k, (const bf16_t *)v,
                               (const bf16_t *)ptab, ub, vb, (bf16_t *)ctx, Tn, Tp, heads, scale * 1.44269504088896340736f, pos_center, ntw, Tk);
            return hipGetLastError();
        }
    }
    const size_t lds = attention_lds_bytes<T, DHP>();
    auto kern = relpos_attention_kernel<T, DHP>;
    hipError_t e = raise_lds_limit((const void *)kern, lds);
    if (e != hipSuccess) return e;
    // log2(e) rides on the 1/sqrt(d_head) factor folded into the query operands: the kernel's scores are in log2 units and its
    // softmax uses v_exp_f32 (2^x) directly -- one multiply per score less.
    // (80-query tiles -- 5 waves, 512 workgroups = one round of 2 per CU at cfg2 instead of 1280 in two rounds -- measured SLOWER:
    // 19.4 us against 16.3 us per launch)
    hipLaunchKernelGGL(kern, dim3(ceil_div(Tn, 64), N * heads), dim3(256), lds, s, q, k, v, ptab, ub, vb, ctx, Tn, Tp, heads, dh,
                       scale * 1.44269504088896340736f, pos_center, stamps);
    return hipGetLastError();
}

static bool uses_chain96(const cocr_model *m);
static bool uses_frontend96(const cocr_model *m);
static int ensure_ctc_scratch(cocr_model *m, size_t rows);

template <typename T, typename TIn>
static int forward_impl(cocr_model *m, const TIn *lines, int N, int H, int W, float *logits, hipStream_t s) {
    const BlobPlan &P = m->plan;
    const unsigned char *B = m->blob;
    auto F32 = [&](size_t off) { return (const float *)(B + off); };
    auto WT = [&](size_t off) { return (const T *)(B + off); };
    const int D = m->D, C = m->C, ff = m->ff, heads = m->heads, dh = m->dh, dhp = m->dhp;
    const bool f2only = m->snum == 1;                    // subsampling_factor 2: conv.0 + ReLU, then the output linear
    const int T1 = out_len1(W), T2 = f2only ? T1 : out_len1(T1), F1 = m->feats[0], F2 = f2only ? F1 : m->feats[1];
    int rc;
    // profiling: one EMPTY event pair per forward = the fixed cost of a bracket (record -> record with nothing between), which
    // bench.py subtracts from every family's average so that the event timings line up with rocprofv3's dispatch durations
    { ProfScope ps(m, s, FAM_EMPTY); }

    // ---- frontend: conv.0 + ReLU + depthwise conv.2 fused, then pointwise conv.3 + ReLU as a GEMM over channels
    T *za = (T *)m->z_a, *zb = (T *)m->z_b;
    bool front_fused = false;
    if constexpr (sizeof(T) == 2) {
        if (uses_frontend96(m)) {                        // conv.0 + ReLU + depthwise conv.2 + pointwise conv.3 + ReLU in one kernel (frontend.hip.h)
            ProfScope ps(m, s, FAM_FRONT96);
            const size_t n0 = (size_t)(C / 16) * 64 * 4;
            GEMM_TRY(launch_frontend96<TIn>(s, lines, N, H, W, T1, F1, T2, m->fpack, F32(P.b0), m->fpack + n0, F32(P.stages[0].dw_b),
                                            (const bf16_t *)(m->packed + P.stages[0].pw_w), F32(P.stages[0].pw_b), (bf16_t *)zb, m->stamps ? m->stamps + 128 : nullptr));
            front_fused = true;
            if ((rc = tap<T>(m, s, "front.z3", zb, (size_t)N * T2 * F2 * C))) return rc;      // (Z2 does not exist on this path)
        }
    }
    if (f2only) {
        ProfScope ps(m, s, FAM_CONV12);
        const size_t npos = (size_t)N * T1 * F1;
        hipLaunchKernelGGL((frontend_conv0_kernel<T, TIn>), dim3((unsigned)((npos * (size_t)(C / 2) + 255) / 256)), dim3(256), 0, s, lines, H, W, T1, F1, C, npos,
                           F32(P.w0), F32(P.b0), zb);
        LAUNCH_CHECK();
    } else if (!front_fused) {
    bool pw_fused = false;
    if constexpr (sizeof(T) == 2) {
        // 32 conv channels (the reference's default model): conv.0 + ReLU + depthwise conv.2 + pointwise conv.3 + ReLU in one launch
        // (conv.hip.h: frontend_conv12pw32_kernel); with debug taps the separate kernels run (Z2 exists there; same arithmetic)
        const size_t lds32 = (size_t)(4 * 16 + 3) * ((H + 11) & ~3) * 4 + (size_t)16 * F2 * 80;      // line tile + Z2 rows of 16 frames
        if (C == 32 && F2 <= 32 && lds32 <= 160 * 1024 && !m->debug && !m->no_front32) {
            ProfScope ps(m, s, FAM_CONV12);
            const size_t lds = lds32;
            auto kern = frontend_conv12pw32_kernel<TIn>;
            GEMM_TRY(raise_lds_limit((const void *)kern, lds));
            hipLaunchKernelGGL(kern, dim3(ceil_div(T2, 16), N), dim3(256), lds, s, lines, H, W, T1, F1, T2, F2, F32(P.w0), F32(P.b0),
                               F32(P.stages[0].dw_w), F32(P.stages[0].dw_b), (const bf16_t *)WT(P.stages[0].pw_w), F32(P.stages[0].pw_b), (bf16_t *)zb);
            LAUNCH_CHECK();
            pw_fused = true;
        }
    }
    if (!pw_fused) {
    {
        ProfScope ps(m, s, FAM_CONV12);
        bool done = false;
        if constexpr (sizeof(T) == 2) {
            if (C % 64 == 0 && !m->no_conv_mfma) {               // conv.0 on the matrix cores (conv.hip.h)
                GEMM_TRY(launch_conv12_mfma<TIn>(s, lines, N, H, W, T1, F1, T2, F2, C, F32(P.w0), F32(P.b0), F32(P.stages[0].dw_w), F32(P.stages[0].dw_b),
                                                 (bf16_t *)za));
                done = true;
            }
        }
        if (!done) {
            const int TB = std::max(1, 512 / C);                 // 256 threads = TB time steps x C/2 channel pairs
            const size_t lds = (size_t)(4 * TB + 3) * ((H + 11) & ~3) * 4;
            hipLaunchKernelGGL((frontend_conv12_kernel<T, TIn>), dim3(ceil_div(T2, TB), N), dim3(256), lds, s, lines, H, W, T1, F1, T2, F2, C,
                               F32(P.w0), F32(P.b0), F32(P.stages[0].dw_w), F32(P.stages[0].dw_b), za, TB);
            LAUNCH_CHECK();
        }
    }
    if ((rc = tap<T>(m, s, "front.z2", za, (size_t)N * T2 * F2 * C))) return rc;
    {
        ProfScope ps(m, s, FAM_FPW);
        EpiBiasAct<T, ACT_RELU> epi{zb, C, F32(P.stages[0].pw_b), C};
        GEMM_TRY(launch_gemm<T>(s, za, C, WT(P.stages[0].pw_w), C, N * T2 * F2, C, C, epi));
    }
    }
    }
    if (!f2only && (rc = tap<T>(m, s, "front.z3", zb, (size_t)N * T2 * F2 * C))) return rc;
    int Tc = T2, Fc = F2;
    T *zcur = zb, *zoth = za;
    for (int st = 1; st < m->snum - 1; ++st) {           // further (depthwise s2, pointwise, ReLU) stages
        const int To = out_len1(Tc), Fo = m->feats[st + 1];
        {
            ProfScope ps(m, s, FAM_FDW);
            hipLaunchKernelGGL((dw3x3s2_kernel<T>), dim3(1024), dim3(256), 0, s, zcur, N, Tc, Fc, To, Fo, C, F32(P.stages[st].dw_w),
                               F32(P.stages[st].dw_b), zoth);
            LAUNCH_CHECK();
        }
        {
            ProfScope ps(m, s, FAM_FPW);
            EpiBiasAct<T, ACT_RELU> epi{zcur, C, F32(P.stages[st].pw_b), C};
            GEMM_TRY(launch_gemm<T>(s, zoth, C, WT(P.stages[st].pw_w), C, N * To * Fo, C, C, epi));
        }
        Tc = To; Fc = Fo;
    }
    const int Tn = Tc, F = Fc, M = N * Tn, Tp = round_up(Tn, 64);
    float *x = m->x;
    T *xn = (T *)m->xn, *hid = (T *)m->hid, *q = (T *)m->q, *k = (T *)m->k, *v = (T *)m->vt, *ctx = (T *)m->ctx, *glu = (T *)m->glu,
      *dwo = (T *)m->dwo;
    const float ffr = m->hp.half_step_residual ? 0.5f : 1.0f;
    const float scale = 1.0f / sqrtf((float)m->rdh);    // (the model's d_head: a padded model's engine d_head is its 64-wide slot)
    const bool rowln = gemm_rowln_supported<T>(D);       // N == D products own whole rows: residual + LayerNorm in their epilogue
    // bf16 row chains: the output linear (K = F C) is the first stage of the first chain launch
    const bool front_in_chain = sizeof(T) == 2 && uses_chain96(m) && (F * C) % 256 == 0 && !m->no_front_chain;
    auto ln = [&](size_t g1, size_t b1, bool write_f32, long g2, long b2) -> int {
        ProfScope ps(m, s, FAM_LN);
        launch_layernorm<T>(s, x, M, D, F32(g1), F32(b1), write_f32 ? x : nullptr, g2 >= 0 ? F32((size_t)g2) : nullptr,
                            b2 >= 0 ? F32((size_t)b2) : nullptr, xn, m->rD);
        LAUNCH_CHECK();
        return COCR_OK;
    };
    // x <- [x +] alpha (A W^T + bias); then the LayerNorm(s) that follow in the reference: xn <- LN1(x), or
    // x <- LN1(x), xn <- LN2(x) (g2 >= 0: block-final LayerNorm chained with the next block's first)
    auto gemm_to_stream = [&](int fam, const T *A, int K, size_t w, size_t bias, float alpha, bool resid, size_t g1, size_t b1, long g2,
                              long b2) -> int {
        if (rowln) {
            ProfScope ps(m, s, fam);
            EpiResidualLN<T, 1> e{x, D, F32(bias), alpha, D, resid ? 1 : 0, F32(g1), F32(b1), g2 >= 0 ? F32((size_t)g2) : nullptr,
                                  b2 >= 0 ? F32((size_t)b2) : nullptr, xn};
            e.Dn = m->rD;
            GEMM_TRY(launch_gemm_rowln<T>(s, A, K, WT(w), K, M, D, K, e));
            return COCR_OK;
        }
        {
            ProfScope ps(m, s, fam);
            if (resid) { EpiResidual e{x, D, F32(bias), alpha, D}; GEMM_TRY(launch_gemm<T>(s, A, K, WT(w), K, M, D, K, e)); }
            else { EpiStoreF32 e{x, D, F32(bias), D}; GEMM_TRY(launch_gemm<T>(s, A, K, WT(w), K, M, D, K, e)); }
        }
        return ln(g1, b1, g2 >= 0, g2, b2);
    };
    // feed-forward module + the LayerNorm(s) that follow it.  bf16, D == 256: one fused kernel (hidden stays on chip)
    auto ffn = [&](const FfnW &fw, size_t g1, size_t b1, long g2, long b2) -> int {
        if constexpr (sizeof(T) == 2) {
            if (rowln && ffn_fused_supported<int>(D, ff) && !m->no_fused_ffn) {
                ProfScope ps(m, s, FAM_FFN_FUSED);
                EpiResidualLN<T, 1> e{x, D, F32(fw.b2), ffr, D, 1, F32(g1), F32(b1), g2 >= 0 ? F32((size_t)g2) : nullptr,
                                      b2 >= 0 ? F32((size_t)b2) : nullptr, xn};
                GEMM_TRY(launch_ffn_fused(s, (const bf16_t *)xn, (const bf16_t *)WT(fw.w1), F32(fw.b1), (const bf16_t *)WT(fw.w2), M, D, ff, e));
                return COCR_OK;
            }
        }
        { ProfScope ps(m, s, FAM_FFN_UP); EpiBiasAct<T, ACT_SILU> e{hid, ff, F32(fw.b1), ff}; GEMM_TRY(launch_gemm<T>(s, xn, D, WT(fw.w1), D, M, ff, D, e)); }
        return gemm_to_stream(FAM_FFN_DOWN, hid, ff, fw.w2, fw.b2, ffr, true, g1, b1, g2, b2);
    };
    // flatten (b,t,(f,c)) is a view of the channel-last tensor; the output linear writes the fp32 residual stream.
    // K = F*C (6144) against N = D: split-K over 2 workgroup groups (2 measured best of 2, 4, 8: 63 vs 74 vs 91 us for product + reduction) (partials in the idle frontend buffer), then one pass
    // sums them, adds the bias and applies the first block's LayerNorm.
    {
#ifndef COCR_FO_SPLITS
#define COCR_FO_SPLITS 2
#endif
        constexpr int SPLITS = COCR_FO_SPLITS;
        const int Kf = F * C;
        if (front_in_chain) {
            // the first row-chain launch multiplies the frontend output by the output linear itself (rowchain.hip.h: FRONT stage)
        } else {
        const bool splitk = rowln && (Kf % (SPLITS * (128 / (int)sizeof(T))) == 0) && D <= 256 &&
                            (size_t)SPLITS * M * D * 4 <= (size_t)N * T2 * F2 * C * sizeof(T);
        if (splitk) {
            float *partial = reinterpret_cast<float *>(zcur == zb ? za : zb);          // the other frontend buffer is free now
            { ProfScope ps(m, s, FAM_FOUT); GEMM_TRY(launch_gemm_splitk<T>(s, zcur, Kf, WT(P.wout), Kf, M, D, Kf, SPLITS, partial)); }
            ProfScope ps(m, s, FAM_LN);
            hipLaunchKernelGGL((splitk_reduce_ln_kernel<T>), dim3(ceil_div(M, 16)), dim3(256), 0, s, partial, SPLITS, (size_t)M * D, F32(P.bout), M, D, m->rD,
                               F32(P.layers[0].ffn[0].ln_g), F32(P.layers[0].ffn[0].ln_b), x, xn);
            LAUNCH_CHECK();
        } else if ((rc = gemm_to_stream(FAM_FOUT, zcur, Kf, P.wout, P.bout, 1.0f, false, P.layers[0].ffn[0].ln_g, P.layers[0].ffn[0].ln_b, -1, -1))) {
            return rc;
        }
        }
    }
    if (!front_in_chain && (rc = tap<float>(m, s, "front.y", x, (size_t)M * D))) return rc;

    if (m->vtN != N || m->vtT != Tn) {   // pad dims of q, k, v must read as zero for this shape
        HIP_TRY(hipMemsetAsync(q, 0, m->qkv_bytes, s));
        HIP_TRY(hipMemsetAsync(k, 0, m->qkv_bytes, s));
        HIP_TRY(hipMemsetAsync(v, 0, m->qkv_bytes, s));
        m->vtN = N; m->vtT = Tn;
    }
    char nm[64];
    // decoder nn.Linear (pred.py:90,121): logits fp32.  Up to 128 classes the product's epilogue also leaves the greedy decoder's per-frame
    // argmax / maximum (ctc_lab / ctc_val): cocr_ctc_greedy on these logits then only merges runs.
    auto decoder = [&](const T *xin) -> int {
        ProfScope ps(m, s, FAM_DEC);
        constexpr int BK = 128 / (int)sizeof(T);
        if (m->ncls <= 128 && D % BK == 0 && (rc = ensure_ctc_scratch(m, (size_t)M)) == COCR_OK) {
            EpiLogitsArgmax e{logits, m->ncls, F32(P.bdec), m->ncls, m->ctc_lab, m->ctc_val};
            GemmArgs<T> a{xin, D, WT(P.wdec), D, M, m->ncls, D, 0};
            GEMM_TRY((launch_ring_cfg<T, 64, 128, 3, EpiLogitsArgmax>(s, a, e)));
            m->amax_ok = true; m->amax_rows = M;
            return COCR_OK;
        }
        if (rc) return rc;
        EpiStoreF32 e{logits, m->ncls, F32(P.bdec), m->ncls};
        GEMM_TRY(launch_gemm<T>(s, xin, D, WT(P.wdec), D, M, m->ncls, D, e));
        m->amax_ok = false;
        return COCR_OK;
    };
    if constexpr (sizeof(T) == 2) {
        if (rowchain_supported(D, ff, dh) && !m->no_chain) {      // (debug taps: the TAPS instantiation of every chain shape)
            // ---- row-local chains (rowchain.hip.h): 3 launches per block (attention core, chain A, chain B)
            const unsigned char *CW = m->packed;                 // chain weights: fragment-major copies
            auto CWT = [&](size_t off) { return (const bf16_t *)(CW + off); };
            // debug taps: the TAPS instantiation of the SAME kernels copies what never leaves the chip (or is overwritten inside the
            // launch) into tapbuf; everything else is read from the buffers the launches leave behind
            const bool taps = m->debug;
            float *tp[4] = {nullptr, nullptr, nullptr, nullptr};
            bf16_t *tap_dw = nullptr;
            if (taps) {
                if (m->tapbuf_rows < (size_t)M) {
                    if (m->tapbuf) (void)hipFree(m->tapbuf);
                    HIP_TRY(hipMalloc((void **)&m->tapbuf, (size_t)M * D * (4 * 4 + 2)));
                    m->tapbuf_rows = (size_t)M;
                }
                for (int i = 0; i < 4; ++i) tp[i] = m->tapbuf + (size_t)i * M * D;
                tap_dw = reinterpret_cast<bf16_t *>(m->tapbuf + (size_t)4 * M * D);
            }
            auto tapx = [&](int l, const char *what, const float *src) -> int { if (l < 0) snprintf(nm, sizeof nm, "%s", what); else snprintf(nm, sizeof nm, "l%d.%s", l, what); return tap<float>(m, s, nm, src, (size_t)M * D); };
            auto tapb = [&](int l, const char *what, const T *src, size_t n) -> int { snprintf(nm, sizeof nm, "l%d.%s", l, what); return tap<T>(m, s, nm, src, n); };
            auto tap_qkv = [&](int l) -> int {
                if (!taps) return COCR_OK;
                int r;
                if ((r = tapb(l, "q", q, m->qkv_bytes / sizeof(T))) || (r = tapb(l, "k", k, m->qkv_bytes / sizeof(T))) || (r = tapb(l, "v", v, m->qkv_bytes / sizeof(T)))) return r;
                return COCR_OK;
            };
            auto launch = [&](const ChainArgs &a) { return D == 256 ? launch_rowchain_256(s, a, taps, m->chain_rows) : launch_rowchain_512(s, a, taps, m->chain_rows); };
            // the fp32 stream between the chain launches: in the kernels' register order (ChainArgs::x_in_blocked); the first launch reads
            // the row-major stream the frontend's reduction wrote
            auto base = [&]() { ChainArgs a{}; a.x = x; a.x_in_blocked = 1; a.x_out_blocked = 1; a.xn = (bf16_t *)xn; a.M = M; a.dh = dh; a.dhp = dhp; a.heads = heads; a.T_ = Tn; a.Tp = Tp; a.inv_d = 1.0f / (float)m->rD; a.xcd_order = m->chain_xcd ? 1 : 0;
                                 a.kd = (m->padded && !m->no_kskip) ? ceil_div(m->rD, 32) : 8; a.kl = (m->padded && !m->no_kskip) ? ((m->rff - 1) % 256) / 32 + 1 : 8; return a; };
            auto st_rowln = [&](size_t wgt, size_t bias, float alpha, size_t g1, size_t b1) {
                ChainStage st{}; st.kind = ST_ROWLN; st.W = CWT(wgt); st.bias = F32(bias); st.N = D; st.alpha = alpha;
                st.g1 = F32(g1); st.b1 = F32(b1); return st; };
            auto st_ffn = [&](const FfnW &fw, size_t g1, size_t b1, long g2, long b2) {
                ChainStage st{}; st.kind = ST_FFN; st.W = CWT(fw.w1); st.W2 = CWT(fw.w2); st.bias = F32(fw.b1); st.bias2 = F32(fw.b2);
                st.N = ff; st.alpha = ffr; st.g1 = F32(g1); st.b1 = F32(b1);      // (W2's fragment-major copy is pre-scaled by ffr: ensure_packed)
                st.g2 = g2 >= 0 ? F32((size_t)g2) : nullptr; st.b2 = b2 >= 0 ? F32((size_t)b2) : nullptr; return st; };
            auto st_qkv = [&](const LayerW &lw) {
                ChainStage st{}; st.kind = ST_QKV; st.W = CWT(lw.wqkv); st.bias = F32(lw.bqkv); st.N = 3 * D;
                st.q = (bf16_t *)q; st.k = (bf16_t *)k; st.v = (bf16_t *)v; return st; };
            if (front_in_chain) {   // frontend output linear + first LayerNorm -> first block's FFN -> its q/k/v projection
                ChainArgs a = base(); a.A0 = (const bf16_t *)zcur; a.nstages = 3; a.x_in_blocked = 0;
                ChainStage f{}; f.kind = ST_FRONT; f.W = CWT(P.wout); f.bias = F32(P.bout); f.N = D; f.K = F * C; f.alpha = 1.0f;
                f.g1 = F32(P.layers[0].ffn[0].ln_g); f.b1 = F32(P.layers[0].ffn[0].ln_b);
                a.st[0] = f;
                a.st[1] = st_ffn(P.layers[0].ffn[0], P.layers[0].a_ln_g, P.layers[0].a_ln_b, -1, -1); a.st[1].store_x = 1;
                a.st[2] = st_qkv(P.layers[0]);
                a.st[0].tap_pre = tp[1]; a.st[1].tap_pre = tp[0];
                { ProfScope ps(m, s, FAM_CH_FRONT); GEMM_TRY(launch(a)); }
                if (taps && ((rc = tapx(-1, "front.y", tp[1])) || (rc = tapx(0, "ffn1", tp[0])) || (rc = tap_qkv(0)))) return rc;
            } else {   // first block's FFN + q/k/v projection on the frontend output
                ChainArgs a = base(); a.A0 = (const bf16_t *)xn; a.nstages = 2; a.x_in_blocked = 0;
                a.st[0] = st_ffn(P.layers[0].ffn[0], P.layers[0].a_ln_g, P.layers[0].a_ln_b, -1, -1); a.st[0].store_x = 1;
                a.st[1] = st_qkv(P.layers[0]);
                a.st[0].tap_pre = tp[0];
                { ProfScope ps(m, s, FAM_CH_FIRST); GEMM_TRY(launch(a)); }
                if (taps && ((rc = tapx(0, "ffn1", tp[0])) || (rc = tap_qkv(0)))) return rc;
            }
            if (m->ffn_probe) {
                // measurement only (COCR_FFN_PROBE=1): block 0's first feed-forward module as a launch of its own on real operands (the
                // frontend output's first M x D values, the stream the launch above left), results discarded (no store flags) -- so that
                // rocprofv3's matrix-pipe counters can be read for the FFN products alone (tools/ffn_probe.py, profiles/r03_ffn_probe_*)
                ChainArgs a = base(); a.A0 = (const bf16_t *)zcur; a.nstages = 1;
                a.st[0] = st_ffn(P.layers[0].ffn[0], P.layers[0].a_ln_g, P.layers[0].a_ln_b, -1, -1);
                { ProfScope ps(m, s, FAM_FFN_PROBE); GEMM_TRY(launch(a)); }
            }
            for (int l = 0; l < m->L; ++l) {
                const LayerW &w = P.layers[l];
                {
                    ProfScope ps(m, s, FAM_ATTN);
#define ATTN(DHP) GEMM_TRY((launch_attention<T, DHP>(s, N, q, k, v, (const T *)(m->ptab + (size_t)l * m->ptab_stride), F32(w.ub), F32(w.vb), ctx, Tn, Tp, heads, dh, scale, m->pos_maxlen - 1, (m->stamps && l == 5) ? m->stamps + 192 : nullptr, m->att_tiled, m->att_resident_min, m->att_resident_long)))
                    if (dhp == 32) ATTN(32); else if (dhp == 64) ATTN(64); else if (dhp == 96) ATTN(96); else ATTN(128);
#undef ATTN
                }
                {   // out-proj + residual + conv-module LayerNorm -> pointwise conv 1 + GLU
                    ChainArgs a = base(); a.A0 = (const bf16_t *)ctx; a.nstages = 2;
                    a.st[0] = st_rowln(w.wo, w.bo, 1.0f, w.c_ln_g, w.c_ln_b); a.st[0].store_x = 1;
                    ChainStage g{}; g.kind = ST_GLU; g.W = CWT(w.wpw1); g.bias = F32(w.bpw1); g.N = 2 * D; g.out = (bf16_t *)glu;
                    a.st[1] = g;
                    if (taps && (rc = tapb(l, "ctx", ctx, (size_t)M * D))) return rc;
                    a.st[0].tap_pre = tp[0];
                    { ProfScope ps(m, s, FAM_CH_A); GEMM_TRY(launch(a)); }
                    if (taps && ((rc = tapx(l, "mhsa", tp[0])) || (rc = tapb(l, "glu", glu, (size_t)M * D)))) return rc;
                }
                const bool dw_fused = m->ksz == 31 && (!m->no_dw_fuse || taps);      // depthwise conv as the chain's prologue
                if (!dw_fused) {
                    ProfScope ps(m, s, FAM_DW);
                    launch_dwconv<T>(s, glu, N, Tn, D, m->ksz, F32(w.dww), F32(w.dwb), dwo);
                    LAUNCH_CHECK();
                }
                {   // [depthwise conv + BN + SiLU ->] pointwise conv 2 + residual + LayerNorm -> FFN 2 (+ closing LayerNorm [+ next block's]) [-> next block's FFN 1 -> its q/k/v]
                    ChainArgs a = base(); a.A0 = (const bf16_t *)dwo;
                    if (dw_fused) { a.dw_in = (const bf16_t *)glu; a.dw_w = F32(w.dww); a.dw_b = F32(w.dwb); a.tap_dw = tap_dw; }
                    a.st[0] = st_rowln(w.wpw2, w.bpw2, 1.0f, w.ffn[1].ln_g, w.ffn[1].ln_b);
                    a.st[0].tap_pre = tp[0];
                    auto tap_common = [&]() -> int {          // the stages every chain B has: depthwise output, stream after the conv module and after FFN 2
                        int r;
                        if ((r = tapb(l, "dw", dw_fused ? (const T *)tap_dw : (const T *)dwo, (size_t)M * D)) || (r = tapx(l, "conv", tp[0])) || (r = tapx(l, "ffn2", tp[1]))) return r;
                        return COCR_OK;
                    };
                    if (l + 1 < m->L) {
                        const LayerW &nx = P.layers[l + 1];
                        a.st[1] = st_ffn(w.ffn[1], w.f_ln_g, w.f_ln_b, (long)nx.ffn[0].ln_g, (long)nx.ffn[0].ln_b);
                        a.st[2] = st_ffn(nx.ffn[0], nx.a_ln_g, nx.a_ln_b, -1, -1); a.st[2].store_x = 1;
                        a.st[3] = st_qkv(nx);
                        a.nstages = 4;
                        a.st[1].tap_pre = tp[1]; a.st[1].tap_post = tp[2]; a.st[2].tap_pre = tp[3];
                        a.stamps = (m->stamps && l == 5) ? m->stamps + 1024 : nullptr;
                        { ProfScope ps(m, s, FAM_CH_B); GEMM_TRY(launch(a)); }
                        if (taps && ((rc = tap_common()) || (rc = tapx(l, "out", tp[2])) || (rc = tapx(l + 1, "ffn1", tp[3])) || (rc = tap_qkv(l + 1)))) return rc;
                    } else {
                        a.st[1] = st_ffn(w.ffn[1], w.f_ln_g, w.f_ln_b, -1, -1); a.st[1].store_x = 1; a.st[1].store_xn = 1;
                        a.nstages = 2;
                        a.st[1].tap_pre = tp[1];
                        { ProfScope ps(m, s, FAM_CH_LAST); GEMM_TRY(launch(a)); }
                        // the closing LayerNorm's output exists only as the bf16 decoder operand xn on this path
                        if (taps && ((rc = tap_common()) || (rc = tapb(l, "out", xn, (size_t)M * D)))) return rc;
                    }
                }
            }
            return decoder(xn);
        }
    }
    for (int l = 0; l < m->L; ++l) {
        const LayerW &w = P.layers[l];
        // FFN, half-step residual (feed_forward.py:45-52, encoder.py:68-75); epilogue: LayerNorm of the attention module
        if ((rc = ffn(w.ffn[0], w.a_ln_g, w.a_ln_b, -1, -1))) return rc;
        snprintf(nm, sizeof nm, "l%d.ffn1", l); if ((rc = tap<float>(m, s, nm, x, (size_t)M * D))) return rc;
        // MHSA (attention.py:143-151)
        { ProfScope ps(m, s, FAM_QKV); EpiQKV<T> e{q, k, v, F32(w.bqkv), D, dh, dhp, heads, Tn, Tp, 3 * D}; GEMM_TRY(launch_gemm<T>(s, xn, D, WT(w.wqkv), D, M, 3 * D, D, e)); }
        {
            ProfScope ps(m, s, FAM_ATTN);
#define ATTN(DHP) GEMM_TRY((launch_attention<T, DHP>(s, N, q, k, v, (const T *)(m->ptab + (size_t)l * m->ptab_stride), F32(w.ub), F32(w.vb), ctx, Tn, Tp, heads, dh, scale, m->pos_maxlen - 1, nullptr, m->att_tiled, m->att_resident_min, m->att_resident_long)))
            if (dhp == 32) ATTN(32); else if (dhp == 64) ATTN(64); else if (dhp == 96) ATTN(96); else ATTN(128);
#undef ATTN
        }
        if (m->debug) {
            snprintf(nm, sizeof nm, "l%d.q", l); if ((rc = tap<T>(m, s, nm, q, m->qkv_bytes / sizeof(T)))) return rc;
            snprintf(nm, sizeof nm, "l%d.k", l); if ((rc = tap<T>(m, s, nm, k, m->qkv_bytes / sizeof(T)))) return rc;
            snprintf(nm, sizeof nm, "l%d.v", l); if ((rc = tap<T>(m, s, nm, v, m->qkv_bytes / sizeof(T)))) return rc;
            snprintf(nm, sizeof nm, "l%d.ctx", l); if ((rc = tap<T>(m, s, nm, ctx, (size_t)M * D))) return rc;
        }
        if ((rc = gemm_to_stream(FAM_AOUT, ctx, D, w.wo, w.bo, 1.0f, true, w.c_ln_g, w.c_ln_b, -1, -1))) return rc;
        snprintf(nm, sizeof nm, "l%d.mhsa", l); if ((rc = tap<float>(m, s, nm, x, (size_t)M * D))) return rc;
        // conv module (convolution.py:135-148)
        { ProfScope ps(m, s, FAM_GLU); EpiGLU<T> e{glu, D, F32(w.bpw1), 2 * D}; GEMM_TRY(launch_gemm<T>(s, xn, D, WT(w.wpw1), D, M, 2 * D, D, e)); }
        {
            ProfScope ps(m, s, FAM_DW);
            launch_dwconv<T>(s, glu, N, Tn, D, m->ksz, F32(w.dww), F32(w.dwb), dwo);
            LAUNCH_CHECK();
        }
        if (m->debug) {
            snprintf(nm, sizeof nm, "l%d.glu", l); if ((rc = tap<T>(m, s, nm, glu, (size_t)M * D))) return rc;
            snprintf(nm, sizeof nm, "l%d.dw", l); if ((rc = tap<T>(m, s, nm, dwo, (size_t)M * D))) return rc;
        }
        if ((rc = gemm_to_stream(FAM_PW2, dwo, D, w.wpw2, w.bpw2, 1.0f, true, w.ffn[1].ln_g, w.ffn[1].ln_b, -1, -1))) return rc;
        snprintf(nm, sizeof nm, "l%d.conv", l); if ((rc = tap<float>(m, s, nm, x, (size_t)M * D))) return rc;
        // second FFN; its epilogue applies the block-final LayerNorm (encoder.py:99) chained with the next block's first
        if (m->debug) {
            // taps want the stream before and after the closing LayerNorm separately: unfused in debug mode
            { ProfScope ps(m, s, FAM_FFN_UP); EpiBiasAct<T, ACT_SILU> e{hid, ff, F32(w.ffn[1].b1), ff}; GEMM_TRY(launch_gemm<T>(s, xn, D, WT(w.ffn[1].w1), D, M, ff, D, e)); }
            { ProfScope ps(m, s, FAM_FFN_DOWN); EpiResidual e{x, D, F32(w.ffn[1].b2), ffr, D}; GEMM_TRY(launch_gemm<T>(s, hid, ff, WT(w.ffn[1].w2), ff, M, D, ff, e)); }
            snprintf(nm, sizeof nm, "l%d.ffn2", l); if ((rc = tap<float>(m, s, nm, x, (size_t)M * D))) return rc;
            if (l + 1 < m->L) { if ((rc = ln(w.f_ln_g, w.f_ln_b, true, (long)P.layers[l + 1].ffn[0].ln_g, (long)P.layers[l + 1].ffn[0].ln_b))) return rc; }
            else if ((rc = ln(w.f_ln_g, w.f_ln_b, true, -1, -1))) return rc;
            snprintf(nm, sizeof nm, "l%d.out", l); if ((rc = tap<float>(m, s, nm, x, (size_t)M * D))) return rc;
        } else if (l + 1 < m->L) {
            const LayerW &nx = P.layers[l + 1];
            if ((rc = ffn(w.ffn[1], w.f_ln_g, w.f_ln_b, (long)nx.ffn[0].ln_g, (long)nx.ffn[0].ln_b))) return rc;
        } else {
            if ((rc = ffn(w.ffn[1], w.f_ln_g, w.f_ln_b, -1, -1))) return rc;
        }
    }
    return decoder(xn);
}

// (Re)builds the fragment-major weight copies the 96-row chain kernels read.  Runs on `s` ahead of the forward's launches
// (stream order covers a blob import issued on the same stream), never inside a graph capture.
static bool uses_chain96(const cocr_model *m) {
    return m->dtype == COCR_BF16 && rowchain_supported(m->D, m->ff, m->dh) && !m->no_chain;
}
static bool uses_frontend96(const cocr_model *m) {
    return m->dtype == COCR_BF16 && m->snum == 2 && frontend96_supported(m->C, m->feats[0], m->feats[1], m->H) && !m->no_front96;
}
static int ensure_packed(cocr_model *m, hipStream_t s) {
    if ((!uses_chain96(m) && !uses_frontend96(m)) || !m->packed_stale) return COCR_OK;
    if (!m->packed) HIP_TRY(hipMalloc((void **)&m->packed, m->plan.total));
    const int D = m->D, ff = m->ff, C = m->C;
    auto pack = [&](size_t off, int N, int K, float scale = 1.0f) {
        hipLaunchKernelGGL(pack_frag_kernel, dim3(std::min(1024, ceil_div(N * K / 8, 256))), dim3(256), 0, s, (const bf16_t *)(m->blob + off),
                           (bf16_t *)(m->packed + off), N, K, scale);
    };
    const float ffr = m->hp.half_step_residual ? 0.5f : 1.0f;       // the FFN's residual factor rides on the packed copy of its second matrix (exact)
    if (uses_chain96(m))
        for (const LayerW &w : m->plan.layers) {
            for (int i = 0; i < 2; ++i) { pack(w.ffn[i].w1, ff, D); pack(w.ffn[i].w2, D, ff, ffr); }
            pack(w.wqkv, 3 * D, D); pack(w.wo, D, D); pack(w.wpw1, 2 * D, D); pack(w.wpw2, D, D);
        }
    if (uses_chain96(m) && (m->feats[m->snum - 1] * C) % 256 == 0) pack(m->plan.wout, D, m->feats[m->snum - 1] * C);      // the FRONT stage's matrix
    if (uses_frontend96(m)) {
        pack(m->plan.stages[0].pw_w, C, C);
        const size_t n0 = (size_t)(C / 16) * 64 * 4, n2 = (size_t)(C / 16) * 5 * 64 * 8;
        if (!m->fpack) HIP_TRY(hipMalloc((void **)&m->fpack, (n0 + n2) * sizeof(bf16_t)));
        hipLaunchKernelGGL(frontend_pack_kernel, dim3(ceil_div((int)n2, 256)), dim3(256), 0, s, (const float *)(m->blob + m->plan.w0),
                           (const float *)(m->blob + m->plan.stages[0].dw_w), m->fpack, m->fpack + n0, C);
    }
    LAUNCH_CHECK();
    m->packed_stale = false;
    for (auto &g : m->graphs) (void)hipGraphExecDestroy(g.exec);     // (pointers unchanged, but keep replay and rebuild ordered simply)
    m->graphs.clear(); m->graph_seen.clear();
    return COCR_OK;
}

static int forward_entry(cocr_model *m, const void *lines, int line_dtype, int N, int H, int W, const int32_t *in_lens,
                         float *logits, int32_t *out_lens, void *stream);
extern "C" int cocr_forward(cocr_model *m, const void *lines, int line_dtype, int N, int H, int W, const int32_t *in_lens,
                            float *logits, int32_t *out_lens, void *stream) {
    if (m) m->amax_logits = nullptr;
    const int rc = forward_entry(m, lines, line_dtype, N, H, W, in_lens, logits, out_lens, stream);
    if (rc == COCR_OK && m->amax_ok) m->amax_logits = logits;      // (plain run, replay or staged replay: the sequence ended with the argmax epilogue)
    return rc;
}
extern "C" int cocr_forget_argmax(cocr_model *m) {
    if (!m) return fail(COCR_EINVAL, "null argument");
    m->amax_logits = nullptr;
    return COCR_OK;
}
static int forward_entry(cocr_model *m, const void *lines, int line_dtype, int N, int H, int W, const int32_t *in_lens,
                         float *logits, int32_t *out_lens, void *stream) {
    if (!m || !lines || !logits) return fail(COCR_EINVAL, "null argument");
    if (m->dtype < 0 || !m->blob) return fail(COCR_ESTATE, "model not finalized");
    if (H != m->H) return fail(COCR_EINVAL, "line height %d does not match the model's height %d", H, m->H);
    if (N < 1 || W < 1) return fail(COCR_EINVAL, "empty batch");
    HIP_TRY(hipSetDevice(m->device));
    {   // the attention core reads the band of whole 64-key tiles unclamped: the tables must cover |relative position| < round_up(T, 64).
        // A longer line than the tables hold: rebuild them (the reference's RelPositionalEncoding.extend_pe, embedding.py:35-41;
        // a position's encoding does not depend on the table length, so shorter lines keep their results)
        const int Tp64 = round_up(cocr_out_len(W, m->hp.subsampling_factor), 64);
        if (Tp64 > 65536) return fail(COCR_EUNSUPPORTED, "more than 65536 output frames");
        cocr_model *root = m->owner ? m->owner : m;      // whose tables these are
        if (Tp64 + 64 > root->pos_maxlen) {
            HIP_TRY(hipDeviceSynchronize());
            for (auto &g : m->graphs) (void)hipGraphExecDestroy(g.exec);      // captured launches point at the old tables
            m->graphs.clear(); m->graph_seen.clear();
            if (root->ptab) { (void)hipFree(root->ptab); root->ptab = nullptr; }
            root->pos_maxlen = round_up(Tp64 + 64, 1024);
            root->ptab_stale = true;
            root->wgen++;
        }
    }
    int rc;
    if (m->owner && (m->owner->dtype != m->dtype || m->owner->plan.total != m->plan.total)) {
        // the owner was finalized again in another compute dtype since cocr_share_weights: this model's layout, plan, workspace and
        // captured launches describe the old blob -- re-derived here, before anything is sized or launched from them
        cocr_model *o = m->owner;
        if (o->dtype < 0 || !o->blob) return fail(COCR_ESTATE, "the model whose weights this one shares is not finalized");
        HIP_TRY(hipDeviceSynchronize());
        if ((rc = set_engine_dims(m, o->dtype))) return rc;
        for (auto &g : m->graphs) (void)hipGraphExecDestroy(g.exec);
        m->graphs.clear(); m->graph_seen.clear();
        free_workspace(m);
        m->plan = make_plan(m, o->dtype);
        m->dtype = o->dtype;
        m->blob = o->blob;
        m->amax_logits = nullptr;
        m->seen_wgen = 0;
    }
    rc = cocr_reserve(m, N, W);
    if (rc) return rc;
    if (in_lens && out_lens)
        for (int i = 0; i < N; ++i) out_lens[i] = cocr_out_len(in_lens[i], m->hp.subsampling_factor);
    hipStream_t s = (hipStream_t)stream;
    if (m->owner) {
        // shared weights: the derived copies are the owner's; rebuilt (rarely: new weights, longer tables) with the device idle, because
        // other models that share them run on other streams; then this model's view of the pointers is refreshed
        cocr_model *o = m->owner;
        if (o->packed_stale || o->ptab_stale) {
            HIP_TRY(hipDeviceSynchronize());
            if ((rc = ensure_packed(o, s)) || (rc = ensure_ptab(o, s))) return rc;
            HIP_TRY(hipDeviceSynchronize());
            o->wgen++;
        }
        if (m->seen_wgen != o->wgen) {
            m->blob = o->blob; m->packed = o->packed; m->fpack = o->fpack; m->ptab = o->ptab; m->ptab_stride = o->ptab_stride; m->pos_maxlen = o->pos_maxlen;
            m->packed_stale = m->ptab_stale = false;
            m->seen_wgen = o->wgen;
            for (auto &g : m->graphs) (void)hipGraphExecDestroy(g.exec);      // captured launches may point at moved buffers
            m->graphs.clear(); m->graph_seen.clear();
        }
    } else {
        const bool rebuilt = m->packed_stale || m->ptab_stale;
        if (rebuilt && m->wgen > 1) HIP_TRY(hipDeviceSynchronize());       // (models may share these buffers: cocr_share_weights)
        if ((rc = ensure_packed(m, s)) || (rc = ensure_ptab(m, s))) return rc;
        if (rebuilt) { m->wgen++; HIP_TRY(hipStreamSynchronize(s)); }
        if (m->seen_wgen != m->wgen) {                  // (a model that shares these buffers may have regrown the tables)
            if (m->seen_wgen) { for (auto &g : m->graphs) (void)hipGraphExecDestroy(g.exec); m->graphs.clear(); m->graph_seen.clear(); }
            m->seen_wgen = m->wgen;
        }
    }
    m->lastN = N;
    m->lastT = cocr_out_len(W, m->hp.subsampling_factor);
    auto run_on = [&](const void *in, float *out) -> int {
        if (m->dtype == COCR_BF16) {
            if (line_dtype == COCR_F32) return forward_impl<bf16_t, float>(m, (const float *)in, N, H, W, out, s);
            if (line_dtype == COCR_U8) return forward_impl<bf16_t, uint8_t>(m, (const uint8_t *)in, N, H, W, out, s);
        } else {
            if (line_dtype == COCR_F32) return forward_impl<float, float>(m, (const float *)in, N, H, W, out, s);
            if (line_dtype == COCR_U8) return forward_impl<float, uint8_t>(m, (const uint8_t *)in, N, H, W, out, s);
        }
        return fail(COCR_EINVAL, "line dtype must be COCR_F32 or COCR_U8");
    };
    auto run = [&]() -> int { return run_on(lines, logits); };
    if (line_dtype != COCR_F32 && line_dtype != COCR_U8) return fail(COCR_EINVAL, "line dtype must be COCR_F32 or COCR_U8");
    if (!m->use_graph || m->debug || m->profile || s == nullptr) return run();
    // Launch-bound regime (~120 kernels of 10-40 us per forward): the second identical call captures the launch sequence
    // into a hipGraph, later identical calls replay it (one host call instead of ~120).
    // Two kinds of captured sequences:
    //   * keyed by the caller's buffers (lines, logits, N, W, dtype, stream): a loop that reuses its buffers replays with no extra copy;
    //   * STAGED, keyed by (N, W, dtype, stream) only: a caller that hands over fresh buffers every call (a data loader's batches, torch's
    //     allocator) gets one device-to-device copy of the lines into a library-owned staging buffer, the replay, and one copy of the
    //     logits out (~20 MB at 32 x 96 x 1200 f32: a few microseconds) instead of ~40 host-side launches.
    const int Tn = cocr_out_len(W, m->hp.subsampling_factor);
    const bool shape_ready = m->vtN == N && m->vtT == Tn;      // the first call of a shape runs plain: one-time attribute / zeroing work
    // (the stream is not part of either key: an instantiated graph launches on any stream, and one model serves one stream at a time anyway)
    auto same = [&](const cocr_model::GraphEntry &g) { return g.lines == lines && g.logits == logits && g.N == N && g.W == W && g.dtype == line_dtype; };
    auto same_shape = [&](const cocr_model::GraphEntry &g) { return g.lines == nullptr && g.N == N && g.W == W && g.dtype == line_dtype; };
    auto capture = [&](const void *in, float *out, hipGraphExec_t *exec) -> int {
        hipGraph_t graph = nullptr;
        HIP_TRY(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        int r = run_on(in, out);
        hipError_t ce = hipStreamEndCapture(s, &graph);
        if (r) { if (graph) (void)hipGraphDestroy(graph); return r; }
        if (ce != hipSuccess) return fail(COCR_EHIP, "hipStreamEndCapture failed: %s", hipGetErrorString(ce));
        HIP_TRY(hipGraphInstantiate(exec, graph, nullptr, nullptr, 0));
        (void)hipGraphDestroy(graph);
        if (m->graphs.size() >= 16) { (void)hipGraphExecDestroy(m->graphs.front().exec); m->graphs.erase(m->graphs.begin()); }
        return COCR_OK;
    };
    const size_t in_bytes = (size_t)N * H * W * (line_dtype == COCR_F32 ? 4 : 1), out_bytes = (size_t)N * Tn * m->ncls * 4;
    auto staged_launch = [&](hipGraphExec_t exec) -> int {
        HIP_TRY(hipMemcpyAsync(m->g_lines, lines, in_bytes, hipMemcpyDefault, s));
        HIP_TRY(hipGraphLaunch(exec, s));
        HIP_TRY(hipMemcpyAsync(logits, m->g_logits, out_bytes, hipMemcpyDeviceToDevice, s));
        return COCR_OK;
    };
    for (auto &g : m->graphs)
        if (same(g)) { HIP_TRY(hipGraphLaunch(g.exec, s)); return COCR_OK; }
    bool seen = false, seen_shape = false;
    for (auto &g : m->graph_seen) { seen = seen || same(g); seen_shape = seen_shape || same_shape(g); }
    if (seen && shape_ready) {                       // the caller reuses its buffers: capture on them
        hipGraphExec_t exec = nullptr;
        if ((rc = capture(lines, logits, &exec))) return rc;
        m->graphs.push_back({lines, logits, N, W, line_dtype, s, exec});
        HIP_TRY(hipGraphLaunch(exec, s));
        return COCR_OK;
    }
    if (m->graph_seen.size() >= 32) m->graph_seen.erase(m->graph_seen.begin());
    m->graph_seen.push_back({lines, logits, N, W, line_dtype, s, nullptr});
    for (auto &g : m->graphs)
        if (same_shape(g)) return staged_launch(g.exec);
    if (!seen_shape || !shape_ready) {               // first call of this shape: plain, on the caller's buffers
        m->graph_seen.push_back({nullptr, nullptr, N, W, line_dtype, s, nullptr});
        return run();
    }
    hipGraphExec_t exec = nullptr;                   // second call of the shape with other buffers: capture the staged sequence
    if ((rc = capture(m->g_lines, m->g_logits, &exec))) return rc;
    m->graphs.push_back({nullptr, nullptr, N, W, line_dtype, s, exec});
    return staged_launch(exec);
}

// ------------------------------------------------------------------------------------ host-side collation
extern "C" int cocr_collate_lines(const void *const *lines, const int32_t *widths, int N, int H, int elem_size, void *dst, int W, int threads) {
    if (N < 0 || H <= 0 || W <= 0 || (elem_size != 1 && elem_size != 4)) return fail(COCR_EINVAL, "collate: bad shape or element size");
    if (N == 0) return COCR_OK;
    if (!lines || !widths || !dst) return fail(COCR_EINVAL, "collate: null argument");
    for (int i = 0; i < N; ++i)
        if (!lines[i] || widths[i] < 0 || widths[i] > W) return fail(COCR_EINVAL, "collate: line %d is %d wide, the batch %d", i, widths[i], W);
    const long rows = (long)N * H;
    auto span = [=](long r0, long r1) {
        for (long r = r0; r < r1; ++r) {
            const int i = (int)(r / H), y = (int)(r % H);
            const size_t wb = (size_t)widths[i] * elem_size, Wb = (size_t)W * elem_size;
            char *d = (char *)dst + (size_t)r * Wb;
            memcpy(d, (const char *)lines[i] + (size_t)y * wb, wb);
            memset(d + wb, 0, Wb - wb);
        }
    };
    const int nt = (int)std::max(1L, std::min<long>(std::min(threads, 64), rows / 64));
    if (nt <= 1) { span(0, rows); return COCR_OK; }
    std::vector<std::thread> pool;
    pool.reserve(nt - 1);
    for (int t = 1; t < nt; ++t) pool.emplace_back(span, rows * t / nt, rows * (t + 1) / nt);
    span(0, rows / nt);
    for (auto &t : pool) t.join();
    return COCR_OK;
}

// ------------------------------------------------------------------------------------ CTC
// The per-line lengths reach the decode kernels through a PINNED host ring: an async copy from pageable memory goes through the
// runtime's staging buffers, and a third such copy in flight (three batches on three streams, each copy queued behind its
// forward) blocked the host until the first forward had finished -- 5.5 ms per run start.  Slots are reused after
// COCR_LENS_SLOTS further decode calls of this model (one model = one stream = a handful of calls in flight at most).
#define COCR_LENS_SLOTS 16
static int upload_lens(cocr_model *m, const int32_t *lens, int N, hipStream_t s) {
    if (N > m->lens_cap) {
        if (m->d_lens) (void)hipFree(m->d_lens);
        if (m->h_lens) (void)hipHostFree(m->h_lens);
        HIP_TRY(hipMalloc((void **)&m->d_lens, (size_t)N * 4 * COCR_LENS_SLOTS));
        HIP_TRY(hipHostMalloc((void **)&m->h_lens, (size_t)N * 4 * COCR_LENS_SLOTS));
        m->lens_cap = N;
        m->lens_slot = 0;
    }
    const int slot = m->lens_slot;
    m->lens_slot = (slot + 1) % COCR_LENS_SLOTS;
    int32_t *h = m->h_lens + (size_t)slot * m->lens_cap;
    memcpy(h, lens, (size_t)N * 4);
    m->d_lens_cur = m->d_lens + (size_t)slot * m->lens_cap;
    HIP_TRY(hipMemcpyAsync(m->d_lens_cur, h, (size_t)N * 4, hipMemcpyHostToDevice, s));
    return COCR_OK;
}

static int ensure_ctc_scratch(cocr_model *m, size_t rows) {
    if (rows > m->ctc_cap) {
        HIP_TRY(hipDeviceSynchronize());                      // (captured launches that point at the old scratch are dropped with it)
        for (auto &g : m->graphs) (void)hipGraphExecDestroy(g.exec);
        m->graphs.clear(); m->graph_seen.clear();
        if (m->ctc_lab) (void)hipFree(m->ctc_lab);
        if (m->ctc_val) (void)hipFree(m->ctc_val);
        HIP_TRY(hipMalloc((void **)&m->ctc_lab, rows * 4));
        HIP_TRY(hipMalloc((void **)&m->ctc_val, rows * 4));
        m->ctc_cap = rows;
        m->amax_logits = nullptr;
    }
    return COCR_OK;
}

extern "C" int cocr_ctc_greedy(cocr_model *m, const float *logits, int N, int T, int ncls, const int32_t *out_lens, int32_t *labels,
                               int32_t *starts, int32_t *ends, float *conf, int32_t *counts, int max_per_line, void *stream) {
    if (!m || !logits || !out_lens || !labels || !starts || !ends || !conf || !counts) return fail(COCR_EINVAL, "null argument");
    if (N < 1 || T < 1 || ncls < 1 || max_per_line < 1) return fail(COCR_EINVAL, "empty problem");
    if (T > 8000) return fail(COCR_EUNSUPPORTED, "more than 8000 frames per line");
    HIP_TRY(hipSetDevice(m->device));
    hipStream_t s = (hipStream_t)stream;
    int rc = upload_lens(m, out_lens, N, s);
    if (rc) return rc;
    const bool have_argmax = logits == m->amax_logits && N * T == m->amax_rows && ncls == m->ncls;      // the decoder's epilogue computed it for these logits
    if (!have_argmax && (rc = ensure_ctc_scratch(m, (size_t)N * T))) return rc;
    ProfScope ps(m, s, FAM_GREEDY);
    if (!have_argmax) {
        m->amax_logits = nullptr;                             // the scratch no longer belongs to the last forward's logits
        hipLaunchKernelGGL(ctc_argmax_kernel, dim3(ceil_div(N * T, 4)), dim3(256), 0, s, logits, T, ncls, N * T, m->d_lens_cur, m->ctc_lab, m->ctc_val);
        LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(ctc_collapse_kernel, dim3(N), dim3(64), (size_t)T * 8, s, T, m->d_lens_cur, m->ctc_lab, m->ctc_val, labels, starts, ends, conf, counts,
                       max_per_line);
    LAUNCH_CHECK();
    return COCR_OK;
}

extern "C" int cocr_ctc_beam(cocr_model *m, const float *logits, int N, int T, int ncls, const int32_t *out_lens, int32_t *labels,
                             int32_t *starts, int32_t *ends, float *conf, int32_t *counts, int max_per_line, int beam, void *stream) {
    if (!m || !logits || !out_lens || !labels || !starts || !ends || !conf || !counts) return fail(COCR_EINVAL, "null argument");
    if (N < 1 || T < 1 || ncls < 2 || max_per_line < 1) return fail(COCR_EINVAL, "empty problem");
    if (beam < 1 || beam > COCR_BEAM_MAX) return fail(COCR_EINVAL, "beam must be in 1..%d", COCR_BEAM_MAX);
    if (ncls > 65535) return fail(COCR_EUNSUPPORTED, "more than 65535 classes");
    HIP_TRY(hipSetDevice(m->device));
    hipStream_t s = (hipStream_t)stream;
    int rc = upload_lens(m, out_lens, N, s);
    if (rc) return rc;
    const bool fast = ncls <= 256 && !m->beam_ref;            // ctc_beam_rank_kernel + ctc_beam_walk_kernel: frames ranked in parallel, a static pruned candidate set per frame
    const int K = std::min(beam + 1, ncls - 1);
    // scratch: back-pointers [N][T][COCR_BEAM_MAX] i32, then log Z [N][T] (exhaustive kernel) or the frame records (fast)
    const size_t need = (size_t)N * T * ((size_t)COCR_BEAM_MAX * 4 + (fast ? (size_t)COCR_BEAM_REC : 4));
    if (need > m->beam_cap) {
        if (m->beam_bp) (void)hipFree(m->beam_bp);
        m->beam_bp = nullptr; m->beam_cap = 0;
        HIP_TRY(hipMalloc((void **)&m->beam_bp, need));
        m->beam_cap = need;
    }
    int32_t *bp = m->beam_bp;
    ProfScope ps(m, s, FAM_BEAM);
    if (fast) {
        unsigned char *rec = reinterpret_cast<unsigned char *>(bp + (size_t)N * T * COCR_BEAM_MAX);
        const size_t dyn = (size_t)T * ((size_t)beam * 4 + 8);                 // back-pointers + the label stack of the final walk, in LDS when they fit
        const int bp_in_lds = dyn <= 96 * 1024;
        hipLaunchKernelGGL(ctc_beam_rank_kernel, dim3(N * T), dim3(256), 0, s, logits, T, ncls, m->d_lens_cur, K, rec);
#define COCR_BEAM_WALK(SL)                                                                                                                          \
    {                                                                                                                                               \
        if (bp_in_lds && dyn > 48 * 1024) {      /* (the kernel also has ~12 KB of static LDS: the limit raised for the dynamic part stays below 160 KB - static) */ \
            static bool raised = false;                                                                                                             \
            if (!raised) { HIP_TRY(hipFuncSetAttribute((const void *)ctc_beam_walk_kernel<SL>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024)); raised = true; } \
        } \
        hipLaunchKernelGGL(ctc_beam_walk_kernel<SL>, dim3(N), dim3(256), bp_in_lds ? dyn : 0, s, logits, T, ncls, m->d_lens_cur, beam, K, labels, starts, \
                           ends, conf, counts, max_per_line, rec, bp, bp_in_lds, m->stamps ? m->stamps + 240 : nullptr);                            \
    }
        if (beam <= 16) COCR_BEAM_WALK(2) else COCR_BEAM_WALK(3)      // candidates per lane of a wave: <= 88 for beam <= 16, <= 184 for beam 32
#undef COCR_BEAM_WALK
    } else {
        float *logz = reinterpret_cast<float *>(bp + (size_t)N * T * COCR_BEAM_MAX);
        const size_t lds = ((size_t)ncls + (size_t)beam * ncls) * 4 + COCR_BEAM_MAX * (11 * 4 + 2 * 8) + 64;
        if (lds > 150 * 1024) return fail(COCR_EUNSUPPORTED, "beam x classes too large for the LDS candidate table");
        HIP_TRY(raise_lds_limit((const void *)ctc_beam_kernel, lds));
        hipLaunchKernelGGL(ctc_beam_kernel, dim3(N), dim3(64), lds, s, logits, T, ncls, m->d_lens_cur, beam, labels, starts, ends, conf, counts,
                           max_per_line, bp, logz);
    }
    LAUNCH_CHECK();
    return COCR_OK;
}

// ------------------------------------------------------------------------------------ CTC loss (ctc_loss.hip.h)
extern "C" int cocr_ctc_loss(cocr_model *m, const float *probits, int N, int T, int ncls, const int32_t *out_lens, const int32_t *targets,
                             const int32_t *label_lens, float *nll, float *grad, void *stream) {
    if (!m || !probits || !out_lens || !label_lens || !nll) return fail(COCR_EINVAL, "null argument");
    if (N < 1 || T < 1 || ncls < 2) return fail(COCR_EINVAL, "empty problem");
    if (ncls > 16384) return fail(COCR_EUNSUPPORTED, "more than 16384 classes");
    size_t total = 0;
    int max_l = 0;
    for (int n = 0; n < N; ++n) {
        if (label_lens[n] < 0) return fail(COCR_EINVAL, "negative target length (line %d)", n);
        if (label_lens[n] > COCR_CTCL_MAX_LABELS) return fail(COCR_EUNSUPPORTED, "line %d has %d labels; the kernel holds at most %d", n, label_lens[n], COCR_CTCL_MAX_LABELS);
        if (out_lens[n] < 0 || out_lens[n] > T) return fail(COCR_EINVAL, "input length %d outside [0, %d] (line %d)", out_lens[n], T, n);
        max_l = std::max(max_l, (int)label_lens[n]);
        total += (size_t)label_lens[n];
    }
    if (total && !targets) return fail(COCR_EINVAL, "null argument");
    for (size_t i = 0; i < total; ++i)
        if (targets[i] < 1 || targets[i] >= ncls) return fail(COCR_EINVAL, "target %d outside [1, %d) (blank is 0)", targets[i], ncls);
    HIP_TRY(hipSetDevice(m->device));
    hipStream_t s = (hipStream_t)stream;
    const size_t ints = (size_t)3 * N + total;
    if (ints > m->loss_ints) {
        if (m->loss_d) (void)hipFree(m->loss_d);
        if (m->loss_h) (void)hipHostFree(m->loss_h);
        m->loss_d = m->loss_h = nullptr;
        m->loss_ints = 0;
        const size_t cap = ints + ints / 2;
        HIP_TRY(hipMalloc((void **)&m->loss_d, cap * 4 * COCR_LENS_SLOTS));
        HIP_TRY(hipHostMalloc((void **)&m->loss_h, cap * 4 * COCR_LENS_SLOTS));
        m->loss_ints = cap;
        m->loss_slot = 0;
    }
    const int slot = m->loss_slot;
    m->loss_slot = (slot + 1) % COCR_LENS_SLOTS;
    int32_t *h = m->loss_h + (size_t)slot * m->loss_ints, *d = m->loss_d + (size_t)slot * m->loss_ints;
    int32_t off = 0;
    for (int n = 0; n < N; ++n) { h[n] = out_lens[n]; h[N + n] = label_lens[n]; h[2 * N + n] = off; off += label_lens[n]; }
    if (total) memcpy(h + 3 * N, targets, total * 4);
    HIP_TRY(hipMemcpyAsync(d, h, ints * 4, hipMemcpyHostToDevice, s));
    const int states = 2 * max_l + 1;
    const int sj = states <= 64 ? 1 : states <= 128 ? 2 : states <= 256 ? 4 : 8;
    const size_t need = (size_t)N * T * ((size_t)ncls + 2 * 64 * sj);
    if (need > m->loss_ws_cap) {
        if (m->loss_ws) (void)hipFree(m->loss_ws);
        m->loss_ws = nullptr;
        m->loss_ws_cap = 0;
        HIP_TRY(hipMalloc((void **)&m->loss_ws, need * 4));
        m->loss_ws_cap = need;
    }
    ProfScope ps(m, s, FAM_LOSS);
    launch_ctc_loss(s, sj, (size_t)ncls * 4, probits, N, T, ncls, d, d + N, d + 2 * N, d + 3 * N, nll, grad, m->loss_ws, m->loss_ws + (size_t)N * T * ncls);
    LAUNCH_CHECK();
    return COCR_OK;
}

// ------------------------------------------------------------------------------------ output-layer training step (train.hip.h)
extern "C" int cocr_decoder_backward(cocr_model *m, const float *grad_probits, int N, int T, float *grad_weight, float *grad_bias, float *grad_output,
                                     void *stream) {
    if (!m || !grad_probits || !grad_weight || !grad_bias) return fail(COCR_EINVAL, "null argument");
    if (m->dtype < 0 || !m->blob) return fail(COCR_ESTATE, "model not finalized");
    if (N != m->lastN || T != m->lastT || !m->xn) return fail(COCR_ESTATE, "no forward of shape (%d lines, %d frames) precedes this call (last forward: %d, %d)", N, T, m->lastN, m->lastT);
    HIP_TRY(hipSetDevice(m->device));
    hipStream_t s = (hipStream_t)stream;
    const int M = N * T, C = m->ncls, D = m->D, chunks = ceil_div(M, COCR_TR_ROWS);
    const size_t per = (size_t)C * D + C, need = per * chunks;
    if (need > m->tr_part_cap) {
        if (m->tr_part) (void)hipFree(m->tr_part);
        m->tr_part = nullptr;
        m->tr_part_cap = 0;
        HIP_TRY(hipMalloc((void **)&m->tr_part, need * 4));
        m->tr_part_cap = need;
    }
    float *part_w = m->tr_part, *part_b = m->tr_part + (size_t)chunks * C * D;
    // A padded model (set_engine_dims): the kernels work on the engine's D-wide rows (the padded columns of the encoder output and of
    // the weight are zero, so are their gradients); the caller's tensors have the model's own width rD: strided copies at the boundary.
    const int rD = m->rD;
    float *gw_dst = grad_weight, *go_dst = grad_output;
    if (m->padded) {
        const size_t need_pad = (size_t)C * D + (grad_output ? (size_t)M * D : 0);
        if (need_pad > m->tr_pad_cap) {
            if (m->tr_pad) (void)hipFree(m->tr_pad);
            m->tr_pad = nullptr; m->tr_pad_cap = 0;
            HIP_TRY(hipMalloc((void **)&m->tr_pad, need_pad * 4));
            m->tr_pad_cap = need_pad;
        }
        gw_dst = m->tr_pad;
        if (grad_output) go_dst = m->tr_pad + (size_t)C * D;
    }
    const dim3 grid(chunks, ceil_div(C, COCR_TR_CT));
    if (m->dtype == COCR_BF16) hipLaunchKernelGGL((decoder_wgrad_kernel<bf16_t>), grid, dim3(256), 0, s, grad_probits, (const bf16_t *)m->xn, M, C, D, part_w, part_b);
    else hipLaunchKernelGGL((decoder_wgrad_kernel<float>), grid, dim3(256), 0, s, grad_probits, (const float *)m->xn, M, C, D, part_w, part_b);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(chunk_reduce_kernel, dim3(ceil_div(C * D, 256)), dim3(256), 0, s, part_w, chunks, (size_t)C * D, gw_dst);
    hipLaunchKernelGGL(chunk_reduce_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, s, part_b, chunks, (size_t)C, grad_bias);
    LAUNCH_CHECK();
    if (grad_output) {
        if (m->dtype == COCR_BF16) hipLaunchKernelGGL((decoder_igrad_kernel<bf16_t>), dim3(ceil_div(M, 16)), dim3(256), 0, s, grad_probits, (const bf16_t *)(m->blob + m->plan.wdec), M, C, D, go_dst);
        else hipLaunchKernelGGL((decoder_igrad_kernel<float>), dim3(ceil_div(M, 16)), dim3(256), 0, s, grad_probits, (const float *)(m->blob + m->plan.wdec), M, C, D, go_dst);
        LAUNCH_CHECK();
    }
    if (m->padded) {
        HIP_TRY(hipMemcpy2DAsync(grad_weight, (size_t)rD * 4, gw_dst, (size_t)D * 4, (size_t)rD * 4, C, hipMemcpyDeviceToDevice, s));
        if (grad_output) HIP_TRY(hipMemcpy2DAsync(grad_output, (size_t)rD * 4, go_dst, (size_t)D * 4, (size_t)rD * 4, M, hipMemcpyDeviceToDevice, s));
    }
    return COCR_OK;
}

extern "C" int cocr_decoder_adamw(cocr_model *m, const float *grad_weight, const float *grad_bias, float lr, float beta1, float beta2, float eps,
                                  float weight_decay, void *stream) {
    if (!m || !grad_weight || !grad_bias) return fail(COCR_EINVAL, "null argument");
    if (m->dtype < 0 || !m->blob) return fail(COCR_ESTATE, "model not finalized");
    if (m->owner) return fail(COCR_ESTATE, "this model shares another model's weights: step the owner");
    if (!(lr >= 0.f) || !(beta1 >= 0.f && beta1 < 1.f) || !(beta2 >= 0.f && beta2 < 1.f) || !(eps >= 0.f) || !(weight_decay >= 0.f))
        return fail(COCR_EINVAL, "invalid AdamW hyper-parameters");            // torch.optim.AdamW's own checks
    HIP_TRY(hipSetDevice(m->device));
    hipStream_t s = (hipStream_t)stream;
    const size_t nw = (size_t)m->ncls * m->D, nb = (size_t)m->ncls, n = nw + nb, nw_model = (size_t)m->ncls * m->rD;
    const size_t row_e = (size_t)m->D * 4, row_m = (size_t)m->rD * 4;      // engine / model row bytes of the decoder weight (equal unless padded)
    if (m->padded) {    // the caller's (ncls, rD) gradient embedded in the engine's zero-padded rows
        if (nw > m->tr_pad_cap) {
            if (m->tr_pad) (void)hipFree(m->tr_pad);
            m->tr_pad = nullptr; m->tr_pad_cap = 0;
            HIP_TRY(hipMalloc((void **)&m->tr_pad, nw * 4));
            m->tr_pad_cap = nw;
        }
        HIP_TRY(hipMemsetAsync(m->tr_pad, 0, nw * 4, s));
        HIP_TRY(hipMemcpy2DAsync(m->tr_pad, row_e, grad_weight, row_m, row_m, m->ncls, hipMemcpyDeviceToDevice, s));
        grad_weight = m->tr_pad;
    }
    if (!m->tr_state) {
        // fp32 master copy: the state-dict tensors when this rank has them, else (weights received by broadcast) the blob's values
        HIP_TRY(hipMalloc((void **)&m->tr_state, 3 * n * 4));
        HIP_TRY(hipMemsetAsync(m->tr_state + n, 0, 2 * n * 4, s));
        auto w = m->host.find("decoder.weight"), b = m->host.find("decoder.bias");
        if (w != m->host.end() && w->second.set && w->second.data.size() == nw_model && b != m->host.end() && b->second.set && b->second.data.size() == nb) {
            HIP_TRY(hipMemsetAsync(m->tr_state, 0, nw * 4, s));
            HIP_TRY(hipMemcpy2DAsync(m->tr_state, row_e, w->second.data.data(), row_m, row_m, m->ncls, hipMemcpyHostToDevice, s));
            HIP_TRY(hipMemcpyAsync(m->tr_state + nw, b->second.data.data(), nb * 4, hipMemcpyHostToDevice, s));
            HIP_TRY(hipStreamSynchronize(s));                                   // pageable sources
        } else {
            if (m->dtype == COCR_BF16) hipLaunchKernelGGL((to_f32_kernel<bf16_t>), dim3(64), dim3(256), 0, s, (const bf16_t *)(m->blob + m->plan.wdec), m->tr_state, nw);
            else hipLaunchKernelGGL((to_f32_kernel<float>), dim3(64), dim3(256), 0, s, (const float *)(m->blob + m->plan.wdec), m->tr_state, nw);
            HIP_TRY(hipMemcpyAsync(m->tr_state + nw, m->blob + m->plan.bdec, nb * 4, hipMemcpyDeviceToDevice, s));
            LAUNCH_CHECK();
        }
        m->tr_step = 0;
    }
    const long t = ++m->tr_step;
    const float bc1 = 1.0f - (float)pow((double)beta1, (double)t), bc2s = (float)sqrt(1.0 - pow((double)beta2, (double)t));
    float *p = m->tr_state, *m1 = p + n, *m2 = m1 + n;
    if (m->dtype == COCR_BF16)
        hipLaunchKernelGGL((adamw_kernel<bf16_t>), dim3(ceil_div((int)nw, 256)), dim3(256), 0, s, p, grad_weight, m1, m2, nw, lr, beta1, beta2, eps, weight_decay, bc1, bc2s,
                           (bf16_t *)(m->blob + m->plan.wdec), (float *)nullptr);
    else
        hipLaunchKernelGGL((adamw_kernel<float>), dim3(ceil_div((int)nw, 256)), dim3(256), 0, s, p, grad_weight, m1, m2, nw, lr, beta1, beta2, eps, weight_decay, bc1, bc2s,
                           (float *)nullptr, (float *)(m->blob + m->plan.wdec));
    hipLaunchKernelGGL((adamw_kernel<float>), dim3(ceil_div((int)nb, 256)), dim3(256), 0, s, p + nw, grad_bias, m1 + nw, m2 + nw, nb, lr, beta1, beta2, eps, weight_decay, bc1, bc2s,
                       (float *)nullptr, (float *)(m->blob + m->plan.bdec));
    LAUNCH_CHECK();
    return COCR_OK;
}

extern "C" int cocr_get_tensor(cocr_model *m, const char *name, float *host_out, int64_t max_elems, void *stream) {
    if (!m || !name || !host_out) return fail(COCR_EINVAL, "null argument");
    const bool is_w = !strcmp(name, "decoder.weight"), is_b = !strcmp(name, "decoder.bias");
    if (!is_w && !is_b) return fail(COCR_EINVAL, "only decoder.weight / decoder.bias are trained by this library (got '%s')", name);
    const size_t nw = (size_t)m->ncls * m->D, nb = (size_t)m->ncls, n = is_w ? (size_t)m->ncls * m->rD : nb;      // nw: the engine's (maybe padded) copy
    if ((int64_t)n > max_elems) return fail(COCR_EINVAL, "%s has %zu elements, buffer %lld", name, n, (long long)max_elems);
    auto it = m->host.find(name);
    if (m->tr_state) {
        HIP_TRY(hipSetDevice(m->device));
        HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
        if (is_w) HIP_TRY(hipMemcpy2D(host_out, (size_t)m->rD * 4, m->tr_state, (size_t)m->D * 4, (size_t)m->rD * 4, m->ncls, hipMemcpyDeviceToHost));
        else HIP_TRY(hipMemcpy(host_out, m->tr_state + nw, n * 4, hipMemcpyDeviceToHost));
        if (it != m->host.end() && it->second.data.size() == n) memcpy(it->second.data.data(), host_out, n * 4);      // a later cocr_finalize keeps the trained values
        return COCR_OK;
    }
    if (it == m->host.end() || !it->second.set || it->second.data.size() != n) return fail(COCR_ESTATE, "%s was never set on this model", name);
    memcpy(host_out, it->second.data.data(), n * 4);
    return COCR_OK;
}

// ------------------------------------------------------------------------------------ line pre-processing (preproc.hip.h)
extern "C" int32_t cocr_preproc_width(int32_t h, int32_t w, int32_t out_h, int32_t pad) {
    if (h < 1 || w < 1 || out_h < 1 || pad < 0) return -1;
    return pre_scaled_width(h, w, out_h) + 2 * pad;
}

extern "C" int cocr_preproc_lines(cocr_model *m, const uint8_t *pixels, const int64_t *offsets, const int32_t *heights, const int32_t *widths,
                                  const int32_t *channels, int N, int out_h, int pad, int out_w, uint8_t *out, int32_t *out_widths, void *stream) {
    if (!m || !pixels || !offsets || !heights || !widths || !out || !out_widths) return fail(COCR_EINVAL, "null argument");
    if (N < 1 || out_h < 1 || pad < 0 || out_w < 1) return fail(COCR_EINVAL, "empty problem");
    std::vector<PreLine> lines((size_t)N);
    std::vector<int> tab;
    size_t tmp_bytes = 0;
    int max_h = 0, max_ow = 0;
    for (int i = 0; i < N; ++i) {
        const int h = heights[i], w = widths[i], cpp = channels ? channels[i] : 1;
        if (h < 1 || w < 1 || (cpp != 1 && cpp != 3)) return fail(COCR_EINVAL, "line %d: %d x %d pixels, %d channels", i, h, w, cpp);
        PreLine &L = lines[(size_t)i];
        L.in_off = offsets[i]; L.h = h; L.w = w; L.cpp = cpp; L.ow = pre_scaled_width(h, w, out_h);
        if (L.ow + 2 * pad > out_w) return fail(COCR_EINVAL, "size mismatch: line %d is %d px wide after scaling and padding, the batch %d", i, L.ow + 2 * pad, out_w);
        pre_coeffs(w, L.ow, tab, &L.hb, &L.hk, &L.hks);
        pre_coeffs(h, out_h, tab, &L.vb, &L.vk, &L.vks);
        L.tmp_off = (long long)tmp_bytes;
        tmp_bytes += (size_t)h * L.ow;
        out_widths[i] = L.ow + 2 * pad;
        max_h = std::max(max_h, h); max_ow = std::max(max_ow, L.ow);
    }
    HIP_TRY(hipSetDevice(m->device));
    hipStream_t s = (hipStream_t)stream;
    const size_t lines_b = round_up((int)(lines.size() * sizeof(PreLine)), 256), tab_b = (size_t)round_up((int)(tab.size() * 4), 256);
    const size_t need = lines_b + tab_b + tmp_bytes;
    if (need > m->pre_cap) {
        HIP_TRY(hipStreamSynchronize(s));                  // an earlier call on this stream may still read the old buffer
        if (m->pre_buf) (void)hipFree(m->pre_buf);
        HIP_TRY(hipMalloc((void **)&m->pre_buf, need + need / 4));
        m->pre_cap = need + need / 4;
    }
    PreLine *d_lines = reinterpret_cast<PreLine *>(m->pre_buf);
    int *d_tab = reinterpret_cast<int *>(m->pre_buf + lines_b);
    unsigned char *d_tmp = m->pre_buf + lines_b + tab_b;
    HIP_TRY(hipMemcpyAsync(d_lines, lines.data(), lines.size() * sizeof(PreLine), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d_tab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));                      // the host tables go out of scope; also orders reuse of pre_buf by the next call
    hipLaunchKernelGGL(preproc_h_kernel, dim3(ceil_div(max_ow, 256), max_h, N), dim3(256), 0, s, pixels, d_lines, d_tab, d_tmp);
    hipLaunchKernelGGL(preproc_v_kernel, dim3(ceil_div(out_w, 256), out_h, N), dim3(256), 0, s, d_tmp, d_lines, d_tab, out, out_h, out_w, pad);
    LAUNCH_CHECK();
    return COCR_OK;
}

// ------------------------------------------------------------------------------------ kernel micro-benchmarks (development hook)
__global__ void fill_kernel(unsigned short *p, size_t n, unsigned seed, int as_f32) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u + seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        const float v = ((float)(h & 0xffff) / 65536.0f - 0.5f);
        if (as_f32) ((float *)p)[i] = v; else ((bf16_t *)p)[i] = (bf16_t)v;
    }
}

struct EpiNull {   // ablation: keeps the accumulators alive, stores (almost) nothing
    typedef float stage_t;
    static constexpr bool GLU = false;
    static constexpr bool ROWWISE = false;
    float *sink;
    __device__ __forceinline__ void transform(int n, const float *v, float *r) const { for (int i = 0; i < 4; ++i) r[i] = v[i]; }
    __device__ __forceinline__ void store(int m, int c, const float *src, int cnt) const { if (src[0] == 123.456f) sink[0] = 1.0f; }
};

// Times one GEMM variant on random operands: returns the average device time per launch in microseconds.
extern "C" int cocr_dev_bench_gemm(int variant, int M, int N, int K, int iters, double *us_out) {
    typedef bf16_t T;
    void *A = nullptr, *W = nullptr, *O = nullptr;
    float *X = nullptr, *bias = nullptr, *gam = nullptr;
    HIP_TRY(hipMalloc(&A, (size_t)M * K * 2)); HIP_TRY(hipMalloc(&W, (size_t)N * K * 2)); HIP_TRY(hipMalloc(&O, (size_t)M * N * 4));
    HIP_TRY(hipMalloc((void **)&X, (size_t)M * K * 4)); HIP_TRY(hipMalloc((void **)&bias, (size_t)N * 4)); HIP_TRY(hipMalloc((void **)&gam, (size_t)K * 4));
    hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, 0, (unsigned short *)A, (size_t)M * K, 1u, 0);
    hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, 0, (unsigned short *)W, (size_t)N * K, 2u, 0);
    hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, 0, (unsigned short *)X, (size_t)M * K, 3u, 1);
    hipLaunchKernelGGL(fill_kernel, dim3(64), dim3(256), 0, 0, (unsigned short *)bias, (size_t)N, 4u, 1);
    hipLaunchKernelGGL(fill_kernel, dim3(64), dim3(256), 0, 0, (unsigned short *)gam, (size_t)K, 5u, 1);
    HIP_TRY(hipMemset(O, 0, (size_t)M * N * 4));
    GemmArgs<T> a{(const T *)A, K, (const T *)W, K, M, N, K, 0};
    EpiBiasAct<T, ACT_SILU> eh{(T *)O, N, bias, N};
    EpiResidual er{(float *)O, N, bias, 0.5f, N};
    EpiNull en{(float *)O};
    const bool resid = N <= 256;
    auto run = [&]() -> hipError_t {
#define RING(BM, BN, NST) (resid ? launch_ring_cfg<T, BM, BN, NST>(0, a, er) : launch_ring_cfg<T, BM, BN, NST>(0, a, eh))
        switch (variant) {
            case 0: return resid ? launch_gemm<T>(0, a.A, K, a.W, K, M, N, K, er) : launch_gemm<T>(0, a.A, K, a.W, K, M, N, K, eh);
            case 1: return RING(64, 64, 3);
            case 2: return RING(64, 64, 4);
            case 3: return RING(64, 128, 3);
            case 4: return RING(128, 128, 2);
            case 5: return RING(128, 128, 3);
            case 6: return RING(128, 64, 3);
            case 7: return RING(64, 128, 2);
            case 8: return RING(64, 64, 2);
            case 9: return resid ? launch_stream_cfg<T, 64, 64>(0, a, er) : launch_stream_cfg<T, 64, 64>(0, a, eh);
            case 30: return launch_ring_cfg<T, 128, 128, 2>(0, a, en);
            case 31: return launch_ring_cfg<T, 64, 64, 3>(0, a, en);
            case 40: {   // fused FFN on (M, D=256, FF=N): A = xn [M][256], W = W1 [N][256], O reused as W2 [256][N]
                EpiResidualLN<T, 1> e{X, 256, bias, 0.5f, 256, 1, gam, gam, nullptr, nullptr, (T *)A};
                return launch_ffn_fused<EpiResidualLN<T, 1>>(0, (const T *)A, (const T *)W, bias, (const T *)O, M, 256, N, e);
            }
            case 20: launch_layernorm<T>(0, X, M, K, gam, gam, nullptr, nullptr, nullptr, (T *)A); return hipGetLastError();
            case 21: launch_layernorm<T>(0, X, M, K, gam, gam, X, gam, gam, (T *)A); return hipGetLastError();
            default: return hipErrorInvalidValue;
        }
#undef RING
    };
    for (int i = 0; i < 3; ++i) HIP_TRY(run());
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
    HIP_TRY(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) HIP_TRY(run());
    HIP_TRY(hipEventRecord(e1, 0));
    HIP_TRY(hipEventSynchronize(e1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    *us_out = (double)ms * 1e3 / iters;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(A); (void)hipFree(W); (void)hipFree(O); (void)hipFree(X); (void)hipFree(bias); (void)hipFree(gam);
    return COCR_OK;
}

extern "C" int cocr_set_chain_rows(cocr_model *m, int rows) {
    if (!m) return fail(COCR_EINVAL, "null argument");
    if (rows < 0 || rows > 96) return fail(COCR_EINVAL, "rows per workgroup must be in 0..96");
    if (rows != m->chain_rows) {          // captured launch sequences use the old grid
        for (auto &g : m->graphs) (void)hipGraphExecDestroy(g.exec);
        m->graphs.clear(); m->graph_seen.clear();
    }
    m->chain_rows = rows;
    return COCR_OK;
}

extern "C" int cocr_set_graph(cocr_model *m, int on) {
    if (!m) return fail(COCR_EINVAL, "null argument");
    m->use_graph = on != 0;
    return COCR_OK;
}

#include "train_api.hip.h"
