// Training step of the whole network behind the C ABI (included at the end of cocr_api.hip):
//     cocr_train_begin      fp32 master copy of every parameter / buffer (reference state-dict names) on the device, zeroed AdamW state
//     cocr_train_step       RecognitionModel.training_step (model.py:129-152): train-mode forward (batch-statistics BatchNorm, dropout at the
//                           reference's six sites), CTC criterion, backward through decoder AND encoder -> the gradient of every parameter
//     cocr_train_adamw      torch.optim.AdamW over all parameters (model.py:283-284)
//     cocr_train_get        a parameter / buffer / gradient by name (checkpointing, tests)
//     cocr_train_end        the trained values back into the model's state (re-finalize to serve them)
// fp32 and correctness-first (train_enc.hip.h); every matrix product -- forward, input gradient, weight gradient -- is the exact-fp32 MFMA GEMM
// of gemm.hip.h: Y = X W^T directly, dX = dY (W^T)^T and dW = dY^T (X^T)^T through explicit transposes.  The inference path (bf16 row
// chains, fused frontend) is not touched: training keeps its own activations (everything the backward needs is stored; nothing is recomputed
// except dropout masks, which are regenerated from (seed, site, index)).
#pragma once

struct TrainEntry { size_t off = 0, n = 0; bool param = false; };

struct TrainState {
    std::map<std::string, TrainEntry> idx;
    std::vector<std::string> order;
    size_t nparam = 0, ntotal = 0;            // floats: parameters first (the optimizer's range), then buffers (BatchNorm running statistics)
    float *P = nullptr, *G = nullptr, *Mo = nullptr, *Vo = nullptr;
    long step = 0;
    unsigned char *ws = nullptr;              // activations + scratch of one step
    size_t ws_bytes = 0;
    float *pe = nullptr;                      // sinusoid rows for relative positions T-1 ... -(T-1), (2T-1, D)
    int peT = 0;
    bool matmul_bf16 = false;                 // cocr_train_set_matmul: the Linear / pointwise-conv products on bf16-rounded operands (fp32 accumulate)
    float *parts = nullptr;                        // partial column sums of the step's deferred finals (k_colsum_final_jobs), bump-allocated per step
    size_t parts_floats = 0, parts_used = 0;
    std::vector<ColsumJob> jobs;
    ColsumJob *jobs_host = nullptr, *jobs_dev = nullptr;      // pinned staging + device copy of the job table (COCR_MAX_COLSUM_JOBS entries)
    bool no_tn = false;                            // COCR_TRAIN_NO_TN=1 (read at cocr_train_set_matmul): weight gradients on transposed copies (A/B)
    unsigned char *Xb = nullptr;                   // 'medium': the bf16 copy of every Linear's input (rows zero-padded to the weight-gradient product's depth), written by the
    size_t Xb_bytes = 0, Xb_used = 0;              // forward, read by the backward as a K-major operand (gemm_tn_kernel): bump-allocated per step, offsets by weight name
    std::map<std::string, size_t> Xb_off;
    unsigned char *Wb = nullptr, *WTb = nullptr;   // 'medium': bf16 copies of every Linear weight (N, K) and of its transpose (K, N), written by the forward, read by the
                                              // backward (byte offset of a tensor = its float offset x 4: 16-byte aligned like the fp32 tensors)
};

static void train_free(cocr_model *m) {
    TrainState *t = m->train;
    if (!t) return;
    for (void *p : {(void *)t->P, (void *)t->G, (void *)t->Mo, (void *)t->Vo, (void *)t->ws, (void *)t->pe, (void *)t->Wb, (void *)t->WTb, (void *)t->Xb, (void *)t->parts,
                    (void *)t->jobs_dev})
        if (p) (void)hipFree(p);
    if (t->jobs_host) (void)hipHostFree(t->jobs_host);
    delete t;
    m->train = nullptr;
}

static bool train_is_buffer(const std::string &n) { return n.find("running_mean") != std::string::npos || n.find("running_var") != std::string::npos; }

extern "C" int cocr_train_set_matmul(cocr_model *m, int bf16_operands) {
    if (!m) return fail(COCR_EINVAL, "null argument");
    if (!m->train) return fail(COCR_ESTATE, "cocr_train_begin first");
    TrainState *t = m->train;
    t->matmul_bf16 = bf16_operands != 0;
    { const char *e = getenv("COCR_TRAIN_NO_TN"); t->no_tn = e && e[0] == '1'; }
    if (t->matmul_bf16 && !t->Wb) {
        HIP_TRY(hipSetDevice(m->device));
        HIP_TRY(hipMalloc((void **)&t->Wb, t->nparam * 4 + 256));
        HIP_TRY(hipMalloc((void **)&t->WTb, t->nparam * 4 + 256));
    }
    return COCR_OK;
}

extern "C" int cocr_train_begin(cocr_model *m) {
    if (!m) return fail(COCR_EINVAL, "null argument");
    if (m->hp.subsampling_conv_channels % 4 != 0 || m->hp.subsampling_conv_channels > 1024)
        return fail(COCR_EUNSUPPORTED, "training: subsampling_conv_channels must be a multiple of 4 and at most 1024 (is %d)", m->hp.subsampling_conv_channels);
    HIP_TRY(hipSetDevice(m->device));
    train_free(m);
    TrainState *t = new TrainState();
    m->train = t;
    for (int pass = 0; pass < 2; ++pass) {        // parameters, then buffers
        for (auto &n : m->names) {
            const HostTensor &h = m->host[n];
            if (!h.set) { train_free(m); return fail(COCR_ESTATE, "missing tensor '%s'", n.c_str()); }
            if (train_is_buffer(n) != (pass == 1)) continue;
            TrainEntry e;
            e.off = t->ntotal; e.n = h.data.size(); e.param = pass == 0;
            t->ntotal += (e.n + 3) / 4 * 4;              // 16-byte aligned tensors
            t->idx[n] = e;
            t->order.push_back(n);
        }
        if (pass == 0) t->nparam = t->ntotal;
    }
    std::vector<float> flat(t->ntotal, 0.f);
    for (auto &kv : t->idx) memcpy(flat.data() + kv.second.off, m->host[kv.first].data.data(), kv.second.n * 4);
    HIP_TRY(hipMalloc((void **)&t->P, t->ntotal * 4));
    HIP_TRY(hipMalloc((void **)&t->G, t->nparam * 4));
    HIP_TRY(hipMalloc((void **)&t->Mo, t->nparam * 4));
    HIP_TRY(hipMalloc((void **)&t->Vo, t->nparam * 4));
    HIP_TRY(hipMemcpy(t->P, flat.data(), t->ntotal * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(t->G, 0, t->nparam * 4));
    HIP_TRY(hipMemset(t->Mo, 0, t->nparam * 4));
    HIP_TRY(hipMemset(t->Vo, 0, t->nparam * 4));
    return COCR_OK;
}

// kind: 0 = value (parameter or buffer), 1 = gradient of the last cocr_train_step
extern "C" int cocr_train_get(cocr_model *m, const char *name, int kind, float *host_out, int64_t n_elems, void *stream) {
    if (!m || !name || !host_out) return fail(COCR_EINVAL, "null argument");
    TrainState *t = m->train;
    if (!t) return fail(COCR_ESTATE, "cocr_train_begin has not been called");
    auto it = t->idx.find(name);
    if (it == t->idx.end()) return fail(COCR_EINVAL, "unknown tensor '%s'", name);
    if ((size_t)n_elems != it->second.n) return fail(COCR_EINVAL, "tensor '%s' has %zu elements", name, it->second.n);
    if (kind == 1 && !it->second.param) return fail(COCR_EINVAL, "'%s' is a buffer: no gradient", name);
    HIP_TRY(hipSetDevice(m->device));
    HIP_TRY(hipMemcpyAsync(host_out, (kind == 1 ? t->G : t->P) + it->second.off, it->second.n * 4, hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return COCR_OK;
}

// the flat device gradient vector (all parameters, the order of cocr_train_begin): what a data-parallel job all-reduces between
// cocr_train_step and cocr_train_adamw
extern "C" int cocr_train_grad_buffer(cocr_model *m, void **device_ptr, size_t *n_floats) {
    if (!m || !device_ptr || !n_floats) return fail(COCR_EINVAL, "null argument");
    if (!m->train) return fail(COCR_ESTATE, "cocr_train_begin has not been called");
    *device_ptr = m->train->G;
    *n_floats = m->train->nparam;
    return COCR_OK;
}

// the flat device VALUE vector in the same layout: parameters [0, n_params), then buffers (BatchNorm running statistics) up to n_total.
// A caller that keeps the parameters elsewhere (torch: `net.nn.parameters()`, updated by a torch optimizer) writes them here before
// cocr_train_step and reads the running statistics back afterwards (device-to-device, stream-ordered).
extern "C" int cocr_train_param_buffer(cocr_model *m, void **device_ptr, size_t *n_total, size_t *n_params) {
    if (!m || !device_ptr || !n_total || !n_params) return fail(COCR_EINVAL, "null argument");
    if (!m->train) return fail(COCR_ESTATE, "cocr_train_begin has not been called");
    *device_ptr = m->train->P;
    *n_total = m->train->ntotal;
    *n_params = m->train->nparam;
    return COCR_OK;
}
// where a reference state-dict name lives in those vectors: float offset and element count; *is_param = 0 for a buffer (no gradient)
extern "C" int cocr_train_layout(cocr_model *m, const char *name, int64_t *offset, int64_t *n_elems, int *is_param) {
    if (!m || !name || !offset || !n_elems || !is_param) return fail(COCR_EINVAL, "null argument");
    if (!m->train) return fail(COCR_ESTATE, "cocr_train_begin has not been called");
    auto it = m->train->idx.find(name);
    if (it == m->train->idx.end()) return fail(COCR_EINVAL, "unknown tensor '%s'", name);
    *offset = (int64_t)it->second.off;
    *n_elems = (int64_t)it->second.n;
    *is_param = it->second.param ? 1 : 0;
    return COCR_OK;
}

extern "C" int cocr_train_end(cocr_model *m) {
    if (!m) return fail(COCR_EINVAL, "null argument");
    TrainState *t = m->train;
    if (!t) return fail(COCR_ESTATE, "cocr_train_begin has not been called");
    HIP_TRY(hipSetDevice(m->device));
    HIP_TRY(hipDeviceSynchronize());
    std::vector<float> flat(t->ntotal);
    HIP_TRY(hipMemcpy(flat.data(), t->P, t->ntotal * 4, hipMemcpyDeviceToHost));
    for (auto &kv : t->idx) memcpy(m->host[kv.first].data.data(), flat.data() + kv.second.off, kv.second.n * 4);
    train_free(m);
    return COCR_OK;
}

extern "C" int cocr_train_adamw(cocr_model *m, float lr, float beta1, float beta2, float eps, float weight_decay, void *stream) {
    if (!m) return fail(COCR_EINVAL, "null argument");
    TrainState *t = m->train;
    if (!t) return fail(COCR_ESTATE, "cocr_train_begin has not been called");
    if (!(lr >= 0.f) || !(beta1 >= 0.f && beta1 < 1.f) || !(beta2 >= 0.f && beta2 < 1.f) || !(eps >= 0.f) || !(weight_decay >= 0.f))
        return fail(COCR_EINVAL, "invalid AdamW hyper-parameters");
    HIP_TRY(hipSetDevice(m->device));
    t->step += 1;
    const float bc1 = 1.0f - powf(beta1, (float)t->step), bc2 = 1.0f - powf(beta2, (float)t->step);
    hipLaunchKernelGGL(k_adamw_flat, dim3(1024), dim3(256), 0, (hipStream_t)stream, t->P, t->G, t->Mo, t->Vo, t->nparam, lr, beta1, beta2, eps, weight_decay, bc1, bc2);
    LAUNCH_CHECK();
    return COCR_OK;
}

// dropout_p: {input, feed_forward, attention, conv} (the reference's four probabilities, encoder.py:144-147); loss_out (host): the summed CTC loss
extern "C" int cocr_train_step(cocr_model *m, const void *lines, int line_dtype, int N, int H, int W, const int32_t *in_lens, const int32_t *targets,
                               const int32_t *label_lens, const float *dropout_p, uint64_t seed, float *loss_out, void *stream) {
    if (!m || !lines || !in_lens || !label_lens || !loss_out) return fail(COCR_EINVAL, "null argument");
    TrainState *t = m->train;
    if (!t) return fail(COCR_ESTATE, "cocr_train_begin has not been called");
    if (H != m->H) return fail(COCR_EINVAL, "line height %d does not match the model's height %d", H, m->H);
    if (N < 1 || W < 1) return fail(COCR_EINVAL, "empty batch");
    if (line_dtype != COCR_F32 && line_dtype != COCR_U8) return fail(COCR_EINVAL, "line dtype must be COCR_F32 or COCR_U8");
    const float p_in = dropout_p ? dropout_p[0] : 0.f, p_ff = dropout_p ? dropout_p[1] : 0.f, p_at = dropout_p ? dropout_p[2] : 0.f, p_cv = dropout_p ? dropout_p[3] : 0.f;
    for (float p : {p_in, p_ff, p_at, p_cv}) if (!(p >= 0.f && p < 1.f)) return fail(COCR_EINVAL, "dropout probability outside [0, 1)");
    HIP_TRY(hipSetDevice(m->device));
    hipStream_t s = (hipStream_t)stream;
    const int D = m->rD, C = m->C, L = m->L, Hh = m->heads, dh = m->rdh, ff = m->rff, K = m->ksz, ncls = m->ncls, snum = m->snum;      // (the model's own dimensions: the engine's may be padded)
    const float ffr = m->hp.half_step_residual ? 0.5f : 1.0f;

    // ---- shapes of the frontend stages
    std::vector<int> Ts(snum), Fs(snum);
    { int tt = W, f = H; for (int i = 0; i < snum; ++i) { tt = out_len1(tt); f = out_len1(f); Ts[i] = tt; Fs[i] = f; } }
    const int T = Ts.back(), F = Fs.back(), M = N * T, Mp = round_up(M, 2048), R = 2 * T - 1, Rp = round_up(R, 2048);      // (row counts of split-K weight gradients: 32 or 64 x <= 32 splits)
    const int nclp = round_up(ncls, 4);
    // attention as batched exact-fp32 GEMMs (train_enc.hip.h) when d_head is a whole number of 32-wide k-chunks; else one wave per row
    const int Tk = round_up(T, 32), Rk = round_up(R, 32), Z = N * Hh;
    const bool attn_gemm = dh % 32 == 0 && !getenv("COCR_TRAIN_ATTN_NAIVE");
    std::vector<int32_t> out_lens(N);
    for (int i = 0; i < N; ++i) out_lens[i] = cocr_out_len(in_lens[i], m->hp.subsampling_factor);

    // ---- workspace: one bump allocation, (re)sized for this shape
    size_t need = 0;
    auto rsv = [&](size_t floats) { size_t o = need; need += (floats * 4 + 255) / 256 * 256; return o; };
    struct Stage { size_t z2, z3; };
    struct Lay {
        size_t x_in, xn1, mu1, rs1, h1, a1, x1, xn2, mu2, rs2, q, k, v, P, attn, ctx, x2, xn3, mu3, rs3, ga, g, dwo, bnm, bnr, xhat, bny, sact, x3, xn4, mu4, rs4, h4, a4, x4, mu5, rs5;
    };
    const size_t oX = rsv((size_t)N * H * W);
    const size_t oZ1 = rsv((size_t)N * Ts[0] * Fs[0] * C);
    std::vector<Stage> stg(snum - 1);
    for (int i = 0; i + 1 < snum; ++i) { const size_t rows = (size_t)N * Ts[i + 1] * Fs[i + 1]; stg[i].z2 = rsv(rows * C); stg[i].z3 = rsv(rows * C); }
    const size_t oZt = rsv((size_t)M * C * F);
    std::vector<Lay> lay(L);
    const size_t MD = (size_t)M * D;
    for (int l = 0; l < L; ++l) {
        Lay &a = lay[l];
        a.x_in = rsv(MD); a.xn1 = rsv(MD); a.mu1 = rsv(M); a.rs1 = rsv(M); a.h1 = rsv((size_t)M * ff); a.a1 = rsv((size_t)M * ff); a.x1 = rsv(MD);
        a.xn2 = rsv(MD); a.mu2 = rsv(M); a.rs2 = rsv(M); a.q = rsv(MD); a.k = rsv(MD); a.v = rsv(MD); a.P = rsv((size_t)R * D);
        a.attn = rsv((size_t)N * Hh * T * (attn_gemm ? Tk : T)); a.ctx = rsv(MD); a.x2 = rsv(MD);
        a.xn3 = rsv(MD); a.mu3 = rsv(M); a.rs3 = rsv(M); a.ga = rsv(2 * MD); a.g = rsv(MD); a.dwo = rsv(MD); a.bnm = rsv(D); a.bnr = rsv(D); a.xhat = rsv(MD);
        a.bny = rsv(MD); a.sact = rsv(MD); a.x3 = rsv(MD);
        a.xn4 = rsv(MD); a.mu4 = rsv(M); a.rs4 = rsv(M); a.h4 = rsv((size_t)M * ff); a.a4 = rsv((size_t)M * ff); a.x4 = rsv(MD); a.mu5 = rsv(M); a.rs5 = rsv(M);
    }
    const size_t oXout = rsv(MD);                              // encoder output (after the last block's LayerNorm)
    const size_t oLogits = rsv((size_t)M * ncls), oDlog = rsv((size_t)M * ncls), oDlogP = rsv((size_t)M * nclp), oNll = rsv(N);
    // backward scratch
    size_t big_rows = (size_t)M;
    for (int i = 0; i + 1 < snum; ++i) big_rows = std::max(big_rows, (size_t)N * Ts[i + 1] * Fs[i + 1]);
    const size_t big_rows_p = (big_rows + 2047) / 2048 * 2048;
    const int wide = std::max(std::max(ff, 3 * D), std::max(C * F, std::max(2 * D, nclp)));
    const size_t tr_floats = std::max((size_t)wide * Mp, big_rows_p * (size_t)C);
    const size_t oTA = rsv(tr_floats), oTB = rsv(tr_floats), oTW = rsv((size_t)std::max(std::max((size_t)ff * D, (size_t)C * F * D), (size_t)std::max(C * C, D * nclp)) + 1024);
    const size_t oBfA = rsv(t->matmul_bf16 ? tr_floats / 2 + 64 : 0), oBfW = rsv(t->matmul_bf16 ? tr_floats / 2 + 64 : 0);      // bf16 copies of a product's two operands
    const size_t oDa = rsv(MD), oDb = rsv(MD), oDc = rsv(MD), oDd = rsv(MD), oDe = rsv(MD), oDwide = rsv((size_t)M * std::max(ff, 2 * D)), oDwide2 = rsv((size_t)M * std::max(ff, 2 * D));
    const size_t oDsb = rsv((size_t)N * Hh * T * (attn_gemm ? Tk : T)), oDP = rsv((size_t)Rp * D);
    size_t oQu = 0, oQv = 0, oRm = 0, oAd = 0, oHT = 0, oTT = 0, oDRT = 0, oPmT = 0;
    if (attn_gemm) {
        oQu = rsv(MD); oQv = rsv(MD); oRm = rsv((size_t)Z * T * Rk); oAd = rsv((size_t)Z * T * Tk); oHT = rsv((size_t)Z * dh * Tk);
        oTT = rsv((size_t)Z * T * Tk); oDRT = rsv((size_t)Z * R * Tk); oPmT = rsv((size_t)Hh * dh * Rk);
    }
    const size_t oZg = rsv((size_t)M * C * F), oZa = rsv(big_rows * C), oZb = rsv(big_rows * C), oZ1g = rsv((size_t)N * Ts[0] * Fs[0] * C);
    const size_t part_floats = std::max(std::max<size_t>(1024, (size_t)ceil_div((int)std::min<size_t>(big_rows, 1u << 30), 256)) * (size_t)std::max(wide, C * 10),
                                        (size_t)ceil_div(N * Ts[0], COCR_CV_ROWS) * 10 * (size_t)C);
    const size_t oPart = rsv(part_floats + 4096), oVec = rsv(4 * (size_t)std::max(D, C) + 64);
    // weight gradients are tall-K products (K = rows): split-K partial sums [splits][out x in]
    auto wg_splits = [](int Nc, int Kr) { const int tiles = ceil_div(Nc, COCR_FO_BM) * ceil_div(Kr, COCR_FO_BN); return std::max(1, std::min(32, 512 / tiles)); };
    size_t split_floats = 0;
    for (auto nk : {std::pair<int, int>{ff, D}, {D, ff}, {D, D}, {2 * D, D}, {C, C}, {D, C * F}, {ncls, D}})
        split_floats = std::max(split_floats, (size_t)wg_splits(nk.first, nk.second) * nk.first * nk.second);
    const size_t oSplit = rsv(split_floats), oLinePart = rsv((size_t)N * std::max((size_t)R * D, (size_t)ceil_div(T, COCR_DW_WC) * D * K));
    {   // deferred column-sum finals: at most 12 jobs per block + the frontend's and the decoder's, each up to ceil(M / 32) x (widest matrix) partial sums
        const size_t pf = (size_t)(12 * L + 16) * (size_t)ceil_div(M, 32) * (size_t)std::max(wide, 2 * D) + 4096;
        if (pf > t->parts_floats) {
            HIP_TRY(hipDeviceSynchronize());
            if (t->parts) (void)hipFree(t->parts);
            t->parts = nullptr; t->parts_floats = 0;
            HIP_TRY(hipMalloc((void **)&t->parts, pf * 4));
            t->parts_floats = pf;
        }
        if (!t->jobs_host) {
            HIP_TRY(hipHostMalloc((void **)&t->jobs_host, COCR_MAX_COLSUM_JOBS * sizeof(ColsumJob)));
            HIP_TRY(hipMalloc((void **)&t->jobs_dev, COCR_MAX_COLSUM_JOBS * sizeof(ColsumJob)));
        }
        t->parts_used = 0;
        t->jobs.clear();
    }
    if (t->matmul_bf16) {
        // upper bound of the Linear inputs of one step (every row count padded by at most 64 x 32 rows)
        const size_t pad = 2048;
        size_t elems = (size_t)L * ((M + pad) * (size_t)(2 * (D + ff) + 6 * D) + (R + pad) * (size_t)D) + (size_t)snum * (big_rows + pad) * C + (M + pad) * ((size_t)C * F + D);
        const size_t bytes = elems * 2 + (size_t)(8 * L + 16) * 256;
        if (bytes > t->Xb_bytes) {
            HIP_TRY(hipDeviceSynchronize());
            if (t->Xb) (void)hipFree(t->Xb);
            t->Xb = nullptr; t->Xb_bytes = 0;
            HIP_TRY(hipMalloc((void **)&t->Xb, bytes));
            t->Xb_bytes = bytes;
        }
        t->Xb_used = 0;
        t->Xb_off.clear();
    }
    if (need > t->ws_bytes) {
        HIP_TRY(hipDeviceSynchronize());
        if (t->ws) (void)hipFree(t->ws);
        t->ws = nullptr; t->ws_bytes = 0;
        HIP_TRY(hipMalloc((void **)&t->ws, need));
        t->ws_bytes = need;
    }
    auto WS = [&](size_t off) { return reinterpret_cast<float *>(t->ws + off); };
    auto Pp = [&](const std::string &n) -> float * { return t->P + t->idx.at(n).off; };
    auto Gp = [&](const std::string &n) -> float * { return t->G + t->idx.at(n).off; };
    auto grid1 = [](size_t n) { return dim3((unsigned)std::min<size_t>((n + 255) / 256, 8192)); };
    char nb[256];
    auto key = [&](int l, const char *suffix) { snprintf(nb, sizeof nb, "encoder.layers.%d.sequential.%s", l, suffix); return std::string(nb); };

    // ---- primitives
    // 'medium' matmul precision (cocr_train_set_matmul; the reference trains under torch.set_float32_matmul_precision('medium'),
    // cli/train.py:252): both operands rounded to bf16, products on the bf16 matrix cores, fp32 accumulation and output
    auto to_bf16 = [&](const float *in, size_t off, size_t n) -> const bf16_t * {
        bf16_t *dst = reinterpret_cast<bf16_t *>(t->ws + off);
        hipLaunchKernelGGL(k_f32_to_bf16, grid1((n + 3) / 4), dim3(256), 0, s, in, dst, (n + 3) / 4);
        return dst;
    };
    auto gemm = [&](const float *A, int lda, const float *Wt, int ldw, int Mr, int Nc, int Kr, float *out, int ldo, const float *bias) -> int {
        EpiStoreF32 e{out, ldo, bias, Nc};
        if (t->matmul_bf16 && (lda & 7) == 0 && (ldw & 7) == 0 && (Kr & 7) == 0) {
            const bf16_t *Ab = to_bf16(A, oBfA, (size_t)Mr * lda), *Wb = to_bf16(Wt, oBfW, (size_t)Nc * ldw);
            GEMM_TRY(launch_gemm<bf16_t>(s, Ab, lda, Wb, ldw, Mr, Nc, Kr, e));
            return COCR_OK;
        }
        GEMM_TRY(launch_gemm<float>(s, A, lda, Wt, ldw, Mr, Nc, Kr, e));
        return COCR_OK;
    };
    auto transpose = [&](const float *in, float *out, int Rr, int Cc, int ldo) {      // out (Cc, ldo) zero-padded beyond Rr
        hipLaunchKernelGGL(k_transpose, dim3(ceil_div(Cc, 32), ceil_div(ldo, 32)), dim3(256), 0, s, in, out, Rr, Cc, ldo);
    };
    // out_z (Mr x Nc, stride ldo) = A_z (Mr x Kr) W_z (Nc x Kr)^T over the Z = N * heads (line, head) batches: offsets per (line, head)
    auto bgemm = [&](const float *A, int lda, long long azb, long long azh, const float *Wm, int ldw, long long wzb, long long wzh, int Mr, int Nc, int Kr,
                     float *out, int ldo, long long ozb, long long ozh) -> int {
        GemmArgs<float> a{A, lda, Wm, ldw, Mr, Nc, Kr, 0};
        a.z_div = Hh; a.a_zb = azb; a.a_zh = azh; a.w_zb = wzb; a.w_zh = wzh; a.o_zb = ozb; a.o_zh = ozh;
        GEMM_TRY(launch_gemm_batched_f32(s, a, out, ldo, Z));
        return COCR_OK;
    };
    const long long sTD = (long long)T * D, sTT = (long long)T * Tk, sTR = (long long)T * Rk, sHT = (long long)dh * Tk;
    // head h of an (M, D) activation as (T x dh) matrices -> [z][dh][Tk] (transposed, zero-padded)
    auto head_T = [&](const float *in, float *out) {
        hipLaunchKernelGGL(k_btranspose, dim3(ceil_div(Tk, 32), ceil_div(dh, 32), Z), dim3(256), 0, s, in, out, T, dh, (long long)D, (long long)Tk, Tk, Hh, sTD, (long long)dh, sHT,
                           0, 0.f, 0ull, 0u);
    };
    // [z][T][Tk] -> its transpose [z][T][Tk]; drop: the attention weights' dropout applied to the input
    auto square_T = [&](const float *in, float *out, bool drop, float p, unsigned site) {
        hipLaunchKernelGGL(k_btranspose, dim3(ceil_div(Tk, 32), ceil_div(T, 32), Z), dim3(256), 0, s, in, out, T, T, (long long)Tk, (long long)Tk, Tk, 1, sTT, 0ll, sTT,
                           drop ? T : 0, p, (unsigned long long)seed, site);
    };
    // Deferred finals: `part_alloc` hands out a region of this step's partial-sum arena (null: arena or job table full -> the caller does the
    // final at once, as before), `defer_final` queues "out[n] = sum over chunks of part[chunk * stride + n]"; `flush_finals` (end of the
    // backward pass) runs them all in one launch.
    auto part_alloc = [&](size_t n) -> float * {
        n = (n + 63) / 64 * 64;
        if (t->parts_used + n > t->parts_floats || t->jobs.size() + 2 > COCR_MAX_COLSUM_JOBS) return nullptr;
        float *p0 = t->parts + t->parts_used;
        t->parts_used += n;
        return p0;
    };
    auto defer_final = [&](const float *part, int stride, int chunks, int Nc, float *out) {
        const int fb = t->jobs.empty() ? 0 : t->jobs.back().first_block + ceil_div(t->jobs.back().N, 64);
        t->jobs.push_back(ColsumJob{part, out, stride, chunks, Nc, fb});
    };
    auto flush_finals = [&]() {
        if (t->jobs.empty()) return;
        const int total = t->jobs.back().first_block + ceil_div(t->jobs.back().N, 64);
        memcpy(t->jobs_host, t->jobs.data(), t->jobs.size() * sizeof(ColsumJob));
        (void)hipMemcpyAsync(t->jobs_dev, t->jobs_host, t->jobs.size() * sizeof(ColsumJob), hipMemcpyHostToDevice, s);
        hipLaunchKernelGGL(k_colsum_final_jobs, dim3(total), dim3(256), 0, s, t->jobs_dev, (int)t->jobs.size());
        t->jobs.clear();
    };
    auto colsum = [&](const float *a, const float *b, int Mr, int Nc, float *out, int accumulate) {
        const int rows = colsum_chunk_rows(Mr), chunks = ceil_div(Mr, rows);
        const bool vec = Nc % 4 == 0 && ((uintptr_t)a & 15) == 0 && ((uintptr_t)b & 15) == 0;
        if (vec) hipLaunchKernelGGL(k_colsum_partial4, dim3(ceil_div(Nc, 256), chunks), dim3(256), 0, s, a, b, WS(oPart), Mr, Nc, rows);
        else hipLaunchKernelGGL(k_colsum_partial, dim3(ceil_div(Nc, 64), chunks), dim3(256), 0, s, a, b, WS(oPart), Mr, Nc, rows);
        if (vec && chunks > 32) hipLaunchKernelGGL(k_colsum_final4, dim3(ceil_div(Nc, 64)), dim3(256), 0, s, WS(oPart), out, chunks, Nc, accumulate);
        else hipLaunchKernelGGL(k_colsum_final, dim3(ceil_div(Nc, 64)), dim3(256), 0, s, WS(oPart), out, chunks, Nc, accumulate);
    };
    // a gradient accumulator's column sums (bias gradients): the partial sums now, the final with the step's other finals
    auto colsum_grad = [&](const float *a, int Mr, int Nc, float *out) {
        const int rows = colsum_chunk_rows(Mr), chunks = ceil_div(Mr, rows);
        float *gp = (Nc % 4 == 0 && ((uintptr_t)a & 15) == 0 && ((uintptr_t)out & 15) == 0) ? part_alloc((size_t)chunks * Nc) : nullptr;
        if (!gp) { colsum(a, nullptr, Mr, Nc, out, 0); return; }
        hipLaunchKernelGGL(k_colsum_partial4, dim3(ceil_div(Nc, 256), chunks), dim3(256), 0, s, a, (const float *)nullptr, gp, Mr, Nc, rows);
        defer_final(gp, Nc, chunks, Nc, out);
    };
    const float *last_x = nullptr;                 // 'medium': the input of the last lin_fwd and where its bf16 copy lies
    int last_rows = 0, last_k = 0, last_rp = 0;
    size_t last_xo = 0;
    // Y (rows, Nc) = X (rows, Kr) W(Nc, Kr)^T + b
    auto lin_fwd = [&](const float *X, const std::string &w, const std::string &b, int rows, int Nc, int Kr, float *Y) -> int {
        if (t->matmul_bf16 && t->Wb && Nc % 8 == 0 && Kr % 8 == 0) {
            // 'medium': the weight's bf16 copy AND its bf16 transpose (the input-gradient product's operand) in one pass, kept for the backward
            const size_t wo = t->idx.at(w).off * 4;
            bf16_t *Wb = reinterpret_cast<bf16_t *>(t->Wb + wo), *WT = reinterpret_cast<bf16_t *>(t->WTb + wo);
            hipLaunchKernelGGL(k_transpose_bf16, dim3(ceil_div(Kr, 32), ceil_div(Nc, 32)), dim3(256), 0, s, Pp(w), WT, Wb, nullptr, Nc, Kr, Nc);
            // the input's bf16 copy is kept for the backward (K-major operand of dW = dY^T X: rows zero-padded to that product's depth)
            const int rp = round_up(rows, 64 * wg_splits(Nc, Kr));
            bf16_t *Ab;
            if (X == last_x && rows == last_rows && Kr == last_k && rp <= last_rp) {
                // the same input as the Linear just before (the query / key / value projections read one LayerNorm output): one copy serves both
                t->Xb_off[w] = last_xo;
                Ab = reinterpret_cast<bf16_t *>(t->Xb + last_xo);
            } else {
                const size_t xo = t->Xb_used, xbytes = ((size_t)rp * Kr * 2 + 255) / 256 * 256;
                if (xo + xbytes > t->Xb_bytes) {
                    // (the arena's size is an estimate over the model's Linears: a shape it did not foresee keeps no copy -- the backward
                    // of this Linear then converts and transposes its operands itself, as before gemm_tn_kernel)
                    t->Xb_off.erase(w);
                    last_x = nullptr;
                    const bf16_t *A1 = to_bf16(X, oBfA, (size_t)rows * Kr);
                    EpiStoreF32 e1{Y, Nc, b.empty() ? nullptr : Pp(b), Nc};
                    GEMM_TRY(launch_gemm<bf16_t>(s, A1, Kr, Wb, Kr, rows, Nc, Kr, e1));
                    return COCR_OK;
                }
                t->Xb_used += xbytes;
                t->Xb_off[w] = xo;
                Ab = reinterpret_cast<bf16_t *>(t->Xb + xo);
                hipLaunchKernelGGL(k_rows_bf16, dim3(ceil_div(Kr, 256), ceil_div(rp, 32)), dim3(256), 0, s, X, Ab, nullptr, rows, Kr, rp);
                last_x = X; last_rows = rows; last_k = Kr; last_rp = rp; last_xo = xo;
            }
            EpiStoreF32 e{Y, Nc, b.empty() ? nullptr : Pp(b), Nc};
            GEMM_TRY(launch_gemm<bf16_t>(s, Ab, Kr, Wb, Kr, rows, Nc, Kr, e));
            return COCR_OK;
        }
        return gemm(X, Kr, Pp(w), Kr, rows, Nc, Kr, Y, Nc, b.empty() ? nullptr : Pp(b));
    };
    // dW += dY^T X, db += colsum(dY), dX = dY W   (dX null: not wanted).  dY (rows, Nc), X (rows, Kr)
    auto lin_bwd = [&](const float *dY, const float *X, const std::string &w, const std::string &b, int rows, int Nc, int Kr, float *dX) -> int {
        const int splits = wg_splits(Nc, Kr), rp = round_up(rows, (t->matmul_bf16 ? 64 : 32) * splits);
        int r;
        if (t->matmul_bf16 && t->Wb && Nc % 8 == 0 && Kr % 8 == 0) {
            // 'medium': dY is read ONCE in fp32 and leaves as the bf16 row-major copy both products take (k_rows_bf16: rows zero-padded to the
            // weight-gradient product's depth, the bias gradient's partial sums on the way); X's copy is the forward's; the weight gradient
            // dW = dY^T X reads both K-major (gemm_tn_kernel: no transposed copies), the input gradient takes the forward's W^T.
            bf16_t *dYR = reinterpret_cast<bf16_t *>(WS(oTA));
            const auto xit = t->Xb_off.find(w);
            const bool have_x = xit != t->Xb_off.end();          // (no copy kept: the forward's arena was full)
            const bf16_t *XR = have_x ? reinterpret_cast<const bf16_t *>(t->Xb + xit->second) : nullptr;
            const bf16_t *WT = reinterpret_cast<const bf16_t *>(t->WTb + t->idx.at(w).off * 4);          // written by lin_fwd of this step
            float *bpart = (!b.empty() && colsum_chunk_rows(rows) == 32 && ((uintptr_t)Gp(b) & 15) == 0) ? part_alloc((size_t)ceil_div(rows, 32) * Nc) : nullptr;
            const bool fuse_bias = bpart != nullptr;
            if (t->no_tn || !have_x) {
                // COCR_TRAIN_NO_TN=1 (A/B of the test): the weight-gradient product on transposed bf16 copies, as before gemm_tn_kernel existed
                bf16_t *dYT = reinterpret_cast<bf16_t *>(t->ws + oBfA), *XT = reinterpret_cast<bf16_t *>(t->ws + oBfW);
                hipLaunchKernelGGL(k_transpose_bf16, dim3(ceil_div(Nc, 32), ceil_div(rp, 32)), dim3(256), 0, s, dY, dYT, dX ? dYR : nullptr, bpart, rows, Nc, rp);
                hipLaunchKernelGGL(k_transpose_bf16, dim3(ceil_div(Kr, 32), ceil_div(rp, 32)), dim3(256), 0, s, X, XT, nullptr, nullptr, rows, Kr, rp);
                if (splits == 1) {
                    EpiStoreF32 e{Gp(w), Kr, nullptr, Kr};
                    GEMM_TRY(launch_gemm<bf16_t>(s, dYT, rp, XT, rp, Nc, Kr, rp, e));
                } else {
                    GEMM_TRY(launch_gemm_splitk<bf16_t>(s, dYT, rp, XT, rp, Nc, Kr, rp, splits, WS(oSplit)));
                }
            } else {
                hipLaunchKernelGGL(k_rows_bf16, dim3(ceil_div(Nc, 256), ceil_div(rp, 32)), dim3(256), 0, s, dY, dYR, bpart, rows, Nc, rp);
                GEMM_TRY(launch_gemm_tn(s, dYR, Nc, XR, Kr, Nc, Kr, rp, splits, splits == 1 ? Gp(w) : WS(oSplit)));
            }
            if (splits > 1) hipLaunchKernelGGL(k_colsum_final, dim3(ceil_div(Nc * Kr, 64)), dim3(256), 0, s, WS(oSplit), Gp(w), splits, Nc * Kr, 0);
            if (fuse_bias) defer_final(bpart, Nc, ceil_div(rows, 32), Nc, Gp(b));
            else if (!b.empty()) colsum_grad(dY, rows, Nc, Gp(b));
            if (dX) {
                EpiStoreF32 e{dX, Kr, nullptr, Kr};
                GEMM_TRY(launch_gemm<bf16_t>(s, dYR, Nc, WT, Nc, rows, Kr, Nc, e));
            }
            return COCR_OK;
        }
        transpose(dY, WS(oTA), rows, Nc, rp);
        transpose(X, WS(oTB), rows, Kr, rp);
        if (splits == 1) {
            if ((r = gemm(WS(oTA), rp, WS(oTB), rp, Nc, Kr, rp, Gp(w), Kr, nullptr))) return r;
        } else {
            if (t->matmul_bf16) {
                const bf16_t *Ab = to_bf16(WS(oTA), oBfA, (size_t)Nc * rp), *Wb = to_bf16(WS(oTB), oBfW, (size_t)Kr * rp);
                GEMM_TRY(launch_gemm_splitk<bf16_t>(s, Ab, rp, Wb, rp, Nc, Kr, rp, splits, WS(oSplit)));
            } else {
                GEMM_TRY(launch_gemm_splitk<float>(s, WS(oTA), rp, WS(oTB), rp, Nc, Kr, rp, splits, WS(oSplit)));
            }
            hipLaunchKernelGGL(k_colsum_final, dim3(ceil_div(Nc * Kr, 64)), dim3(256), 0, s, WS(oSplit), Gp(w), splits, Nc * Kr, 0);
        }
        if (!b.empty()) colsum_grad(dY, rows, Nc, Gp(b));
        if (dX) {
            const int np = round_up(Nc, 4);
            const float *dYp = dY;
            if (np != Nc) { hipLaunchKernelGGL(k_pad_cols, grid1((size_t)rows * np), dim3(256), 0, s, dY, WS(oDlogP), rows, Nc, np); dYp = WS(oDlogP); }
            transpose(Pp(w), WS(oTW), Nc, Kr, np);                      // W^T (Kr, np)
            if ((r = gemm(dYp, np, WS(oTW), np, rows, Kr, np, dX, Kr, nullptr))) return r;
        }
        return COCR_OK;
    };
    auto ln_fwd = [&](const float *x, const std::string &g, const std::string &b, float *y, float *mu, float *rs) {
        hipLaunchKernelGGL(k_ln_fwd, dim3(ceil_div(M, 4)), dim3(256), 0, s, x, Pp(g), Pp(b), y, mu, rs, M, D);
    };
    // dx (+)= LayerNorm backward of dy; d gamma, d beta
    auto ln_bwd = [&](const float *dy, const float *x, const float *mu, const float *rs, const std::string &g, const std::string &b, float *dx, int accumulate) {
        hipLaunchKernelGGL(k_ln_bwd, dim3(ceil_div(M, 4)), dim3(256), 0, s, dy, x, mu, rs, Pp(g), dx, WS(oDwide2), M, D, accumulate);
        if (D % 4 == 0 && ((uintptr_t)dy & 15) == 0 && ((uintptr_t)Gp(g) & 15) == 0 && ((uintptr_t)Gp(b) & 15) == 0) {      // d gamma and d beta in one pair of launches
            const int rows = colsum_chunk_rows(M), chunks = ceil_div(M, rows);
            float *lp = part_alloc((size_t)chunks * 2 * D);
            hipLaunchKernelGGL(k_colsum_partial4_2, dim3(ceil_div(2 * D, 256), chunks), dim3(256), 0, s, WS(oDwide2), dy, lp ? lp : WS(oPart), M, D, rows);
            if (lp) { defer_final(lp, 2 * D, chunks, D, Gp(g)); defer_final(lp + D, 2 * D, chunks, D, Gp(b)); }
            else hipLaunchKernelGGL(k_colsum_final_2, dim3(ceil_div(2 * D, 64)), dim3(256), 0, s, WS(oPart), Gp(g), Gp(b), chunks, D);
            return;
        }
        colsum(WS(oDwide2), nullptr, M, D, Gp(g), 0);
        colsum(dy, nullptr, M, D, Gp(b), 0);
    };
    auto dropout = [&](float *x, size_t n, float p, unsigned site) {
        if (p > 0.f) hipLaunchKernelGGL(k_dropout, grid1(n), dim3(256), 0, s, x, n, p, (unsigned long long)seed, site);
    };
    auto copy = [&](float *dst, const float *src, size_t n) { (void)hipMemcpyAsync(dst, src, n * 4, hipMemcpyDeviceToDevice, s); };
    int rc;

    // ---- positional rows PE(p), p = T-1 ... -(T-1) (embedding.py:35-56,66), cached per T
    if (t->peT != T) {
        if (t->pe) (void)hipFree(t->pe);
        t->pe = nullptr;
        std::vector<float> pe((size_t)R * D);
        for (int r = 0; r < R; ++r) {
            const float pos = (float)(T - 1 - r);
            for (int i = 0; i < D; i += 2) {
                const float div = expf((float)i * (float)(-(log(10000.0) / D)));
                pe[(size_t)r * D + i] = sinf(pos * div);
                if (i + 1 < D) pe[(size_t)r * D + i + 1] = cosf(pos * div);
            }
        }
        HIP_TRY(hipMalloc((void **)&t->pe, pe.size() * 4));
        HIP_TRY(hipMemcpy(t->pe, pe.data(), pe.size() * 4, hipMemcpyHostToDevice));
        t->peT = T;
    }
    HIP_TRY(hipMemsetAsync(t->G, 0, t->nparam * 4, s));

    // =========================================================================================== forward (train mode)
    float *X = WS(oX);
    if (line_dtype == COCR_U8) hipLaunchKernelGGL(k_u8_to_f32, grid1((size_t)N * H * W), dim3(256), 0, s, (const uint8_t *)lines, X, (size_t)N * H * W);
    else copy(X, (const float *)lines, (size_t)N * H * W);
    hipLaunchKernelGGL(k_conv0_fwd, dim3(N * Ts[0]), dim3(256), 0, s, X, Pp("encoder.conv_subsample.conv.0.weight"),
                       Pp("encoder.conv_subsample.conv.0.bias"), WS(oZ1), N, H, W, Ts[0], Fs[0], C);
    const size_t cw_lds = (size_t)(256 / (C / 4)) * 10 * C * sizeof(float);          // conv_w_block_sum: [position phases][10][C]
    auto tfc = [&](const float *in, float *out, int reverse) {                       // (n, t, f, c) <-> (n, t, c, f), a row per block through LDS
        const size_t lds = (size_t)F * (C + 1) * sizeof(float);
        if (lds <= 64 * 1024) hipLaunchKernelGGL(k_tfc_to_tcf, dim3(M), dim3(256), lds, s, in, out, (size_t)M, F, C, reverse);
        else hipLaunchKernelGGL(k_tfc_to_tcf_flat, grid1((size_t)M * F * C), dim3(256), 0, s, in, out, (size_t)M, F, C, reverse);
    };
    auto conv_name = [&](int idx, const char *leaf) { snprintf(nb, sizeof nb, "encoder.conv_subsample.conv.%d.%s", idx, leaf); return std::string(nb); };
    {
        const float *zin = WS(oZ1);
        for (int i = 0, idx = 2; i + 1 < snum; ++i, idx += 3) {
            const size_t rows = (size_t)N * Ts[i + 1] * Fs[i + 1];
            hipLaunchKernelGGL(k_dw3_fwd, dim3(N * Ts[i + 1]), dim3(256), 0, s, zin, Pp(conv_name(idx, "weight")), Pp(conv_name(idx, "bias")), WS(stg[i].z2),
                               N, Ts[i], Fs[i], Ts[i + 1], Fs[i + 1], C);
            if ((rc = lin_fwd(WS(stg[i].z2), conv_name(idx + 1, "weight"), conv_name(idx + 1, "bias"), (int)rows, C, C, WS(stg[i].z3)))) return rc;
            hipLaunchKernelGGL(k_relu, grid1(rows * C), dim3(256), 0, s, WS(stg[i].z3), rows * C);
            zin = WS(stg[i].z3);
        }
        tfc(zin, WS(oZt), 0);
    }
    if ((rc = lin_fwd(WS(oZt), "encoder.conv_subsample.out.0.weight", "encoder.conv_subsample.out.0.bias", M, D, C * F, WS(lay[0].x_in)))) return rc;
    dropout(WS(lay[0].x_in), MD, p_in, 1);
    const float scale = 1.0f / sqrtf((float)dh);
    const long long arows = (long long)N * Hh * T;
    auto ffn_fwd = [&](int l, int which, const float *xin, size_t oxn, size_t omu, size_t ors, size_t oh, size_t oa, float *xout) -> int {
        const std::string pre = std::string(which == 0 ? "0" : "3") + ".module.sequential.";
        ln_fwd(xin, key(l, (pre + "0.weight").c_str()), key(l, (pre + "0.bias").c_str()), WS(oxn), WS(omu), WS(ors));
        int r;
        if ((r = lin_fwd(WS(oxn), key(l, (pre + "1.linear.weight").c_str()), key(l, (pre + "1.linear.bias").c_str()), M, ff, D, WS(oh)))) return r;
        hipLaunchKernelGGL(k_silu_fwd_drop, grid1((size_t)M * ff), dim3(256), 0, s, WS(oh), WS(oa), (size_t)M * ff, p_ff, (unsigned long long)seed, (unsigned)(16 * l + 2 + 8 * which));
        if ((r = lin_fwd(WS(oa), key(l, (pre + "4.linear.weight").c_str()), key(l, (pre + "4.linear.bias").c_str()), M, D, ff, WS(oDa)))) return r;
        hipLaunchKernelGGL(k_add3_drop, grid1(MD), dim3(256), 0, s, xout, xin, WS(oDa), ffr, MD, p_ff, (unsigned long long)seed, (unsigned)(16 * l + 3 + 8 * which));
        return COCR_OK;
    };
    for (int l = 0; l < L; ++l) {
        Lay &a = lay[l];
        if ((rc = ffn_fwd(l, 0, WS(a.x_in), a.xn1, a.mu1, a.rs1, a.h1, a.a1, WS(a.x1)))) return rc;
        // MHSA
        ln_fwd(WS(a.x1), key(l, "1.module.layer_norm.weight"), key(l, "1.module.layer_norm.bias"), WS(a.xn2), WS(a.mu2), WS(a.rs2));
        if ((rc = lin_fwd(WS(a.xn2), key(l, "1.module.attention.query_proj.linear.weight"), key(l, "1.module.attention.query_proj.linear.bias"), M, D, D, WS(a.q)))) return rc;
        if ((rc = lin_fwd(WS(a.xn2), key(l, "1.module.attention.key_proj.linear.weight"), key(l, "1.module.attention.key_proj.linear.bias"), M, D, D, WS(a.k)))) return rc;
        if ((rc = lin_fwd(WS(a.xn2), key(l, "1.module.attention.value_proj.linear.weight"), key(l, "1.module.attention.value_proj.linear.bias"), M, D, D, WS(a.v)))) return rc;
        if ((rc = lin_fwd(t->pe, key(l, "1.module.attention.pos_proj.linear.weight"), "", R, D, D, WS(a.P)))) return rc;
        if (attn_gemm) {
            hipLaunchKernelGGL(k_attn_qu_qv, grid1(MD), dim3(256), 0, s, WS(a.q), Pp(key(l, "1.module.attention.u_bias")), Pp(key(l, "1.module.attention.v_bias")),
                               WS(oQu), WS(oQv), MD, D);
            // S = (q + u) K^T -> attn buffer; Rm = (q + vb) P_h^T for all 2T - 1 relative positions
            if ((rc = bgemm(WS(oQu), D, sTD, dh, WS(a.k), D, sTD, dh, T, T, dh, WS(a.attn), Tk, (long long)Hh * sTT, sTT))) return rc;
            if ((rc = bgemm(WS(oQv), D, sTD, dh, WS(a.P), D, 0, dh, T, R, dh, WS(oRm), Rk, (long long)Hh * sTR, sTR))) return rc;
            hipLaunchKernelGGL(k_attn_softmax, dim3((unsigned)((arows + 3) / 4)), dim3(256), 0, s, WS(a.attn), WS(oRm), p_at > 0.f ? WS(oAd) : (float *)nullptr, arows, T, Tk,
                               Rk, scale, p_at, (unsigned long long)seed, (unsigned)(16 * l + 4));
            head_T(WS(a.v), WS(oHT));                                   // V^T per (line, head)
            if ((rc = bgemm(p_at > 0.f ? WS(oAd) : WS(a.attn), Tk, (long long)Hh * sTT, sTT, WS(oHT), Tk, (long long)Hh * sHT, sHT, T, dh, Tk, WS(a.ctx), D, sTD, dh)))
                return rc;
        } else {
            hipLaunchKernelGGL(k_attn_fwd, dim3((unsigned)((arows + 3) / 4)), dim3(256), 4 * 2 * dh * 4, s, WS(a.q), WS(a.k), WS(a.v), WS(a.P),
                               Pp(key(l, "1.module.attention.u_bias")), Pp(key(l, "1.module.attention.v_bias")), WS(a.attn), WS(a.ctx), arows, T, Hh, dh, scale,
                               p_at, (unsigned long long)seed, (unsigned)(16 * l + 4));
        }
        if ((rc = lin_fwd(WS(a.ctx), key(l, "1.module.attention.out_proj.linear.weight"), key(l, "1.module.attention.out_proj.linear.bias"), M, D, D, WS(oDa)))) return rc;
        hipLaunchKernelGGL(k_add3_drop, grid1(MD), dim3(256), 0, s, WS(a.x2), WS(a.x1), WS(oDa), 1.0f, MD, p_at, (unsigned long long)seed, (unsigned)(16 * l + 5));
        // conv module
        ln_fwd(WS(a.x2), key(l, "2.module.sequential.0.weight"), key(l, "2.module.sequential.0.bias"), WS(a.xn3), WS(a.mu3), WS(a.rs3));
        if ((rc = lin_fwd(WS(a.xn3), key(l, "2.module.sequential.2.conv.weight"), key(l, "2.module.sequential.2.conv.bias"), M, 2 * D, D, WS(a.ga)))) return rc;
        hipLaunchKernelGGL(k_glu_fwd, grid1(MD), dim3(256), 0, s, WS(a.ga), WS(a.g), M, D);
        if (K == 31) hipLaunchKernelGGL((k_dw1d_rows<false, 31>), dim3(ceil_div(D, 256), ceil_div(T, COCR_DW_TC), N), dim3(256), 0, s, WS(a.g), Pp(key(l, "2.module.sequential.4.conv.weight")), WS(a.dwo), N, T, D, K);
        else if (K <= 32) hipLaunchKernelGGL((k_dw1d_rows<false, 0>), dim3(ceil_div(D, 256), ceil_div(T, COCR_DW_TC), N), dim3(256), 0, s, WS(a.g), Pp(key(l, "2.module.sequential.4.conv.weight")), WS(a.dwo), N, T, D, K);
        else hipLaunchKernelGGL(k_dw1d_fwd_flat, grid1(MD), dim3(256), 0, s, WS(a.g), Pp(key(l, "2.module.sequential.4.conv.weight")), WS(a.dwo), N, T, D, K, 0);
        if (D % 4 == 0) {                                  // sum x and sum x^2 in one pass
            const int rows = colsum_chunk_rows(M), chunks = ceil_div(M, rows);
            hipLaunchKernelGGL(k_colsum_partial4_sq, dim3(ceil_div(2 * D, 256), chunks), dim3(256), 0, s, WS(a.dwo), WS(oPart), M, D, rows);
            hipLaunchKernelGGL(k_colsum_final_2, dim3(ceil_div(2 * D, 64)), dim3(256), 0, s, WS(oPart), WS(oVec), WS(oVec) + D, chunks, D);
        } else {
            colsum(WS(a.dwo), nullptr, M, D, WS(oVec), 0);
            colsum(WS(a.dwo), WS(a.dwo), M, D, WS(oVec) + D, 0);
        }
        hipLaunchKernelGGL(k_bn_finalize, dim3(ceil_div(D, 256)), dim3(256), 0, s, WS(oVec), WS(oVec) + D, M, D, WS(a.bnm), WS(a.bnr),
                           Pp(key(l, "2.module.sequential.5.running_mean")), Pp(key(l, "2.module.sequential.5.running_var")), 0.1f);
        hipLaunchKernelGGL(k_bn_apply_silu, grid1(MD), dim3(256), 0, s, WS(a.dwo), WS(a.bnm), WS(a.bnr), Pp(key(l, "2.module.sequential.5.weight")),
                           Pp(key(l, "2.module.sequential.5.bias")), WS(a.xhat), WS(a.bny), WS(a.sact), M, D);
        if ((rc = lin_fwd(WS(a.sact), key(l, "2.module.sequential.7.conv.weight"), key(l, "2.module.sequential.7.conv.bias"), M, D, D, WS(oDa)))) return rc;
        hipLaunchKernelGGL(k_add3_drop, grid1(MD), dim3(256), 0, s, WS(a.x3), WS(a.x2), WS(oDa), 1.0f, MD, p_cv, (unsigned long long)seed, (unsigned)(16 * l + 6));
        if ((rc = ffn_fwd(l, 1, WS(a.x3), a.xn4, a.mu4, a.rs4, a.h4, a.a4, WS(a.x4)))) return rc;
        float *xnext = l + 1 < L ? WS(lay[l + 1].x_in) : WS(oXout);
        ln_fwd(WS(a.x4), key(l, "4.weight"), key(l, "4.bias"), xnext, WS(a.mu5), WS(a.rs5));
    }
    if ((rc = lin_fwd(WS(oXout), "decoder.weight", "decoder.bias", M, ncls, D, WS(oLogits)))) return rc;
    LAUNCH_CHECK();
    // ---- criterion (model.py:119,136-142): summed CTC loss and d loss / d probits
    if ((rc = cocr_ctc_loss(m, WS(oLogits), N, T, ncls, out_lens.data(), targets, label_lens, WS(oNll), WS(oDlog), stream))) return rc;
    {
        std::vector<float> nll(N);
        HIP_TRY(hipMemcpyAsync(nll.data(), WS(oNll), (size_t)N * 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        double sum = 0.0;
        for (float v : nll) sum += v;
        *loss_out = (float)sum;
    }

    // =========================================================================================== backward
    float *dx = WS(oDb);                                    // gradient of the stream entering the current point
    if ((rc = lin_bwd(WS(oDlog), WS(oXout), "decoder.weight", "decoder.bias", M, ncls, D, dx))) return rc;
    auto ffn_bwd = [&](int l, int which, const float *xin, size_t oxn, size_t omu, size_t ors, size_t oh, size_t oa, float *dxio) -> int {
        // x_out = x_in + ffr drop(W2 drop(silu(W1 LN(x_in) + b1)) + b2): dxio holds d x_out on entry, d x_in on exit
        const std::string pre = std::string(which == 0 ? "0" : "3") + ".module.sequential.";
        float *dob = WS(oDa);
        hipLaunchKernelGGL(k_scale_drop, grid1(MD), dim3(256), 0, s, dob, dxio, ffr, MD, p_ff, (unsigned long long)seed, (unsigned)(16 * l + 3 + 8 * which));
        int r;
        if ((r = lin_bwd(dob, WS(oa), key(l, (pre + "4.linear.weight").c_str()), key(l, (pre + "4.linear.bias").c_str()), M, D, ff, WS(oDwide)))) return r;
        hipLaunchKernelGGL(k_silu_bwd_drop, grid1((size_t)M * ff), dim3(256), 0, s, WS(oh), WS(oDwide), (size_t)M * ff, p_ff, (unsigned long long)seed, (unsigned)(16 * l + 2 + 8 * which));
        if ((r = lin_bwd(WS(oDwide), WS(oxn), key(l, (pre + "1.linear.weight").c_str()), key(l, (pre + "1.linear.bias").c_str()), M, ff, D, WS(oDc)))) return r;
        ln_bwd(WS(oDc), xin, WS(omu), WS(ors), key(l, (pre + "0.weight").c_str()), key(l, (pre + "0.bias").c_str()), dxio, 1);
        return COCR_OK;
    };
    for (int l = L - 1; l >= 0; --l) {
        Lay &a = lay[l];
        // block-final LayerNorm (encoder.py:99)
        ln_bwd(dx, WS(a.x4), WS(a.mu5), WS(a.rs5), key(l, "4.weight"), key(l, "4.bias"), WS(oDd), 0);
        dx = WS(oDd);
        if ((rc = ffn_bwd(l, 1, WS(a.x3), a.xn4, a.mu4, a.rs4, a.h4, a.a4, dx))) return rc;
        // conv module: x3 = x2 + drop(pw2(silu(bn(dw(glu(pw1(LN(x2))))))))
        {
            float *dob = WS(oDa);
            hipLaunchKernelGGL(k_scale_drop, grid1(MD), dim3(256), 0, s, dob, dx, 1.0f, MD, p_cv, (unsigned long long)seed, (unsigned)(16 * l + 6));
            if ((rc = lin_bwd(dob, WS(a.sact), key(l, "2.module.sequential.7.conv.weight"), key(l, "2.module.sequential.7.conv.bias"), M, D, D, WS(oDc)))) return rc;
            hipLaunchKernelGGL(k_silu_bwd, grid1(MD), dim3(256), 0, s, WS(a.bny), WS(oDc), MD);               // d bn_y
            float *gbeta = Gp(key(l, "2.module.sequential.5.bias")), *ggamma = Gp(key(l, "2.module.sequential.5.weight"));
            if (D % 4 == 0 && ((uintptr_t)gbeta & 15) == 0 && ((uintptr_t)ggamma & 15) == 0) {
                // sum dy (= d beta) and sum dy xhat (= d gamma) in one pass, straight into the gradient vector; the input gradient reads them there
                const int rows = colsum_chunk_rows(M), chunks = ceil_div(M, rows);
                hipLaunchKernelGGL(k_colsum_partial4_ab, dim3(ceil_div(2 * D, 256), chunks), dim3(256), 0, s, WS(oDc), WS(a.xhat), WS(oPart), M, D, rows);
                hipLaunchKernelGGL(k_colsum_final_2, dim3(ceil_div(2 * D, 64)), dim3(256), 0, s, WS(oPart), gbeta, ggamma, chunks, D);
            } else {
                colsum(WS(oDc), nullptr, M, D, WS(oVec), 0);                                                      // sum dy   = d beta
                colsum(WS(oDc), WS(a.xhat), M, D, WS(oVec) + D, 0);                                               // sum dy xhat = d gamma
                copy(gbeta, WS(oVec), D);
                copy(ggamma, WS(oVec) + D, D);
            }
            hipLaunchKernelGGL(k_bn_bwd, grid1(MD), dim3(256), 0, s, WS(oDc), WS(a.xhat), Pp(key(l, "2.module.sequential.5.weight")), WS(a.bnr), gbeta,
                               ggamma, WS(oDe), M, D);                                                        // d dwo
            if (K <= 32) {
                const int nch = ceil_div(T, COCR_DW_WC);
                const dim3 gw(ceil_div(D, 256), nch, N), gr(ceil_div(D, 256), ceil_div(T, COCR_DW_TC), N);
                const float *wdw = Pp(key(l, "2.module.sequential.4.conv.weight"));
                float *gdw = Gp(key(l, "2.module.sequential.4.conv.weight"));
                if (K == 31) hipLaunchKernelGGL(k_dw1d_bwd_w<31>, gw, dim3(256), 0, s, WS(oDe), WS(a.g), WS(oLinePart), N, T, D, K);
                else hipLaunchKernelGGL(k_dw1d_bwd_w<0>, gw, dim3(256), 0, s, WS(oDe), WS(a.g), WS(oLinePart), N, T, D, K);
                if ((D * K) % 4 == 0) hipLaunchKernelGGL(k_colsum_final4, dim3(ceil_div(D * K, 64)), dim3(256), 0, s, WS(oLinePart), gdw, N * nch, D * K, 0);
                else hipLaunchKernelGGL(k_colsum_final, dim3(ceil_div(D * K, 64)), dim3(256), 0, s, WS(oLinePart), gdw, N * nch, D * K, 0);
                if (K == 31) hipLaunchKernelGGL((k_dw1d_rows<true, 31>), gr, dim3(256), 0, s, WS(oDe), wdw, WS(oDc), N, T, D, K);   // d g
                else hipLaunchKernelGGL((k_dw1d_rows<true, 0>), gr, dim3(256), 0, s, WS(oDe), wdw, WS(oDc), N, T, D, K);
            } else {
                hipLaunchKernelGGL(k_dw1d_bwd_w_flat, dim3(ceil_div(D, 64), K, N), dim3(64), 0, s, WS(oDe), WS(a.g), WS(oLinePart), N, T, D, K);
                hipLaunchKernelGGL(k_colsum_final, dim3(ceil_div(D * K, 64)), dim3(256), 0, s, WS(oLinePart), Gp(key(l, "2.module.sequential.4.conv.weight")), N, D * K, 0);
                hipLaunchKernelGGL(k_dw1d_fwd_flat, grid1(MD), dim3(256), 0, s, WS(oDe), Pp(key(l, "2.module.sequential.4.conv.weight")), WS(oDc), N, T, D, K, 1);   // d g
            }
            hipLaunchKernelGGL(k_glu_bwd, grid1(MD), dim3(256), 0, s, WS(a.ga), WS(oDc), WS(oDwide), M, D);                                              // d a (M, 2D)
            if ((rc = lin_bwd(WS(oDwide), WS(a.xn3), key(l, "2.module.sequential.2.conv.weight"), key(l, "2.module.sequential.2.conv.bias"), M, 2 * D, D, WS(oDc)))) return rc;
            ln_bwd(WS(oDc), WS(a.x2), WS(a.mu3), WS(a.rs3), key(l, "2.module.sequential.0.weight"), key(l, "2.module.sequential.0.bias"), dx, 1);
        }
        // MHSA: x2 = x1 + drop(out_proj(attention(LN(x1))))
        {
            float *dob = WS(oDa);
            hipLaunchKernelGGL(k_scale_drop, grid1(MD), dim3(256), 0, s, dob, dx, 1.0f, MD, p_at, (unsigned long long)seed, (unsigned)(16 * l + 5));
            if ((rc = lin_bwd(dob, WS(a.ctx), key(l, "1.module.attention.out_proj.linear.weight"), key(l, "1.module.attention.out_proj.linear.bias"), M, D, D, WS(oDc)))) return rc;   // d ctx
            float *du_part = WS(oDwide), *dvb_part = WS(oDwide) + MD;
            const float *ub = Pp(key(l, "1.module.attention.u_bias")), *vbp = Pp(key(l, "1.module.attention.v_bias"));
            if (attn_gemm) {
                const unsigned site = (unsigned)(16 * l + 4);
                const long long zTT = (long long)Hh * sTT, zHT = (long long)Hh * sHT;
                hipLaunchKernelGGL(k_attn_qu_qv, grid1(MD), dim3(256), 0, s, WS(a.q), ub, vbp, WS(oQu), WS(oQv), MD, D);
                // dA = dctx V^T, then ds (in place): oDsb
                if ((rc = bgemm(WS(oDc), D, sTD, dh, WS(a.v), D, sTD, dh, T, T, dh, WS(oDsb), Tk, zTT, sTT))) return rc;
                hipLaunchKernelGGL(k_attn_softmax_bwd, dim3((unsigned)((arows + 3) / 4)), dim3(256), 0, s, WS(oDsb), WS(a.attn), arows, T, Tk, scale, p_at,
                                   (unsigned long long)seed, site);
                // dV = drop(attn)^T dctx -> oDa
                square_T(WS(a.attn), WS(oTT), true, p_at, site);
                head_T(WS(oDc), WS(oHT));
                if ((rc = bgemm(WS(oTT), Tk, zTT, sTT, WS(oHT), Tk, zHT, sHT, T, dh, Tk, WS(oDa), D, sTD, dh))) return rc;
                // d(q + u) = ds K -> du_part
                head_T(WS(a.k), WS(oHT));
                if ((rc = bgemm(WS(oDsb), Tk, zTT, sTT, WS(oHT), Tk, zHT, sHT, T, dh, Tk, du_part, D, sTD, dh))) return rc;
                // dK = ds^T (q + u) -> oDe
                square_T(WS(oDsb), WS(oTT), false, 0.f, 0u);
                head_T(WS(oQu), WS(oHT));
                if ((rc = bgemm(WS(oTT), Tk, zTT, sTT, WS(oHT), Tk, zHT, sHT, T, dh, Tk, WS(oDe), D, sTD, dh))) return rc;
                // dR = shift^-1(ds) -> oRm;  d(q + vb) = dR P_h -> dvb_part
                hipLaunchKernelGGL(k_attn_unshift, dim3((unsigned)((arows + 3) / 4)), dim3(256), 0, s, WS(oDsb), WS(oRm), arows, T, Tk, Rk);
                hipLaunchKernelGGL(k_btranspose, dim3(ceil_div(Rk, 32), ceil_div(dh, 32), Hh), dim3(256), 0, s, WS(a.P), WS(oPmT), R, dh, (long long)D, (long long)Rk, Rk, Hh, 0ll,
                                   (long long)dh, (long long)dh * Rk, 0, 0.f, 0ull, 0u);
                if ((rc = bgemm(WS(oRm), Rk, (long long)Hh * sTR, sTR, WS(oPmT), Rk, 0, (long long)dh * Rk, T, dh, Rk, dvb_part, D, sTD, dh))) return rc;
                // dP_h = sum over the lines of dR^T (q + vb): per line into oLinePart [line][R][D], summed below
                hipLaunchKernelGGL(k_btranspose, dim3(ceil_div(Tk, 32), ceil_div(R, 32), Z), dim3(256), 0, s, WS(oRm), WS(oDRT), T, R, (long long)Rk, (long long)Tk, Tk, 1, sTR, 0ll,
                                   (long long)R * Tk, 0, 0.f, 0ull, 0u);
                head_T(WS(oQv), WS(oHT));
                if ((rc = bgemm(WS(oDRT), Tk, (long long)Hh * R * Tk, (long long)R * Tk, WS(oHT), Tk, zHT, sHT, R, dh, Tk, WS(oLinePart), D, (long long)R * D, dh))) return rc;
            } else {
                hipLaunchKernelGGL(k_attn_bwd_rows, dim3((unsigned)((arows + 3) / 4)), dim3(256), 4 * dh * 4, s, WS(oDc), WS(a.k), WS(a.v), WS(a.P), WS(a.attn), WS(oDsb),
                                   du_part, dvb_part, arows, T, Hh, dh, scale, p_at, (unsigned long long)seed, (unsigned)(16 * l + 4));
                hipLaunchKernelGGL(k_attn_bwd_cols, dim3((unsigned)((arows + 3) / 4)), dim3(256), 0, s, WS(oDc), WS(a.q), ub, WS(a.attn), WS(oDsb), WS(oDe), WS(oDa),
                                   arows, T, Hh, dh, p_at, (unsigned long long)seed, (unsigned)(16 * l + 4));            // d k -> oDe, d v -> oDa
                hipLaunchKernelGGL(k_attn_bwd_pos, dim3(ceil_div(R * Hh, 4), N), dim3(256), 0, s, WS(a.q), vbp, WS(oDsb), WS(oLinePart), N, T, Hh, dh);
            }
            (void)hipMemsetAsync(WS(oDP), 0, (size_t)Rp * D * 4, s);
            hipLaunchKernelGGL(k_colsum_final, dim3(ceil_div(R * D, 64)), dim3(256), 0, s, WS(oLinePart), WS(oDP), N, R * D, 0);
            colsum_grad(du_part, M, D, Gp(key(l, "1.module.attention.u_bias")));
            colsum_grad(dvb_part, M, D, Gp(key(l, "1.module.attention.v_bias")));
            hipLaunchKernelGGL(k_axpy, grid1(MD), dim3(256), 0, s, du_part, dvb_part, 1.0f, MD);                 // d q
            // pos_proj weight: P = PE Wpos^T  ->  d Wpos = dP^T PE
            if ((rc = lin_bwd(WS(oDP), t->pe, key(l, "1.module.attention.pos_proj.linear.weight"), "", R, D, D, nullptr))) return rc;
            float *dxn = WS(oDc);
            if ((rc = lin_bwd(du_part, WS(a.xn2), key(l, "1.module.attention.query_proj.linear.weight"), key(l, "1.module.attention.query_proj.linear.bias"), M, D, D, dxn))) return rc;
            if ((rc = lin_bwd(WS(oDe), WS(a.xn2), key(l, "1.module.attention.key_proj.linear.weight"), key(l, "1.module.attention.key_proj.linear.bias"), M, D, D, WS(oDwide2)))) return rc;
            hipLaunchKernelGGL(k_axpy, grid1(MD), dim3(256), 0, s, dxn, WS(oDwide2), 1.0f, MD);
            if ((rc = lin_bwd(WS(oDa), WS(a.xn2), key(l, "1.module.attention.value_proj.linear.weight"), key(l, "1.module.attention.value_proj.linear.bias"), M, D, D, WS(oDwide2)))) return rc;
            hipLaunchKernelGGL(k_axpy, grid1(MD), dim3(256), 0, s, dxn, WS(oDwide2), 1.0f, MD);
            // (ln_bwd uses oDwide2 as its product scratch: dxn lives in oDc)
            ln_bwd(dxn, WS(a.x1), WS(a.mu2), WS(a.rs2), key(l, "1.module.layer_norm.weight"), key(l, "1.module.layer_norm.bias"), dx, 1);
        }
        if ((rc = ffn_bwd(l, 0, WS(a.x_in), a.xn1, a.mu1, a.rs1, a.h1, a.a1, dx))) return rc;
        // dx is now d x_in of block l = d (output of block l-1's LayerNorm): keep it out of the buffers the next iteration overwrites first
        copy(WS(oDb), dx, MD);
        dx = WS(oDb);
    }
    // ---- frontend
    dropout(dx, MD, p_in, 1);
    if ((rc = lin_bwd(dx, WS(oZt), "encoder.conv_subsample.out.0.weight", "encoder.conv_subsample.out.0.bias", M, D, C * F, WS(oZg)))) return rc;
    {
        float *dz3 = snum == 1 ? WS(oZ1g) : WS(oZa), *dz2 = WS(oZb);      // (factor 2: the flattened tensor IS conv.0's output)
        tfc(WS(oZg), dz3, 1);
        for (int i = snum - 2, idx = 2 + 3 * (snum - 2); i >= 0; --i, idx -= 3) {
            const size_t rows = (size_t)N * Ts[i + 1] * Fs[i + 1];
            if (i == snum - 2) hipLaunchKernelGGL(k_relu_bwd, grid1(rows * C), dim3(256), 0, s, WS(stg[i].z3), dz3, rows * C);     // (later stages: masked where it was produced)
            if ((rc = lin_bwd(dz3, WS(stg[i].z2), conv_name(idx + 1, "weight"), conv_name(idx + 1, "bias"), (int)rows, C, C, dz2))) return rc;
            const float *zin = i == 0 ? WS(oZ1) : WS(stg[i - 1].z3);
            const int chunks = ceil_div(N * Ts[i + 1], COCR_CV_ROWS);
            hipLaunchKernelGGL(k_dw3_bwd_w, dim3(chunks), dim3(256), cw_lds, s, dz2, zin, WS(oPart), N, Ts[i], Fs[i], Ts[i + 1], Fs[i + 1], C);
            hipLaunchKernelGGL(k_conv_w_final, dim3(ceil_div(C * 10, 64)), dim3(256), 0, s, WS(oPart), chunks, C, Gp(conv_name(idx, "weight")), Gp(conv_name(idx, "bias")));
            // d (stage input), masked by the ReLU that produced that input (conv.0's for i == 0, the previous stage's conv.3's otherwise)
            float *dzin = i == 0 ? WS(oZ1g) : dz3;              // (for i > 0 the previous stage's d z3 has the shape of z3[i-1] <= big_rows x C)
            hipLaunchKernelGGL(k_dw3_bwd_in, dim3(N * Ts[i]), dim3(256), 0, s, dz2, Pp(conv_name(idx, "weight")), dzin, zin, N, Ts[i], Fs[i], Ts[i + 1], Fs[i + 1], C);
        }
        if (snum == 1) hipLaunchKernelGGL(k_relu_bwd, grid1((size_t)N * Ts[0] * Fs[0] * C), dim3(256), 0, s, WS(oZ1), WS(oZ1g), (size_t)N * Ts[0] * Fs[0] * C);
        const int chunks = ceil_div(N * Ts[0], COCR_CV_ROWS);
        hipLaunchKernelGGL(k_conv0_bwd_w, dim3(chunks), dim3(256), cw_lds, s, WS(oZ1g), X, WS(oPart), N, H, W, Ts[0], Fs[0], C);
        hipLaunchKernelGGL(k_conv_w_final, dim3(ceil_div(C * 10, 64)), dim3(256), 0, s, WS(oPart), chunks, C, Gp("encoder.conv_subsample.conv.0.weight"),
                           Gp("encoder.conv_subsample.conv.0.bias"));
    }
    flush_finals();
    LAUNCH_CHECK();
    return COCR_OK;
}
