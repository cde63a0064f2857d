// commit.hip - host side of specmi_commit: weight packing into the kernels' layouts, eval-mode BatchNorm folding (incl. the
// downsample branch folded into conv3), FC stacking, the float64 composition of the HMR regressor (+ its variance rows), the
// ResNet layer tables and the SMPL constants.  Split out of api.hip in round 6; reference interfaces replaced: load_state_dict /
// load_pretrained_model (spec/tester.py:63-71, scripts/camcalib_demo.py:80-81, spec/models/hmr.py:124-136).
#include <cmath>
#include <cstdarg>
#include <climits>
#include <cstdlib>
#include <cstring>

#include "handle.h"

using namespace specmi;

// ------------------------------------------------------------------------------------------
// packing
// ------------------------------------------------------------------------------------------
// OIHW -> [Kp/4][Npad][4], k = (ky*KW + kx)*Cin + ci  (zero padded)
void pack_gemm_weights(const float* w, int cout, int cin, int kh, int kw, int Kp, int Npad,
                              std::vector<float>& out) {
    out.assign((size_t)Kp * Npad, 0.f);
    for (int n = 0; n < cout; ++n)
        for (int ci = 0; ci < cin; ++ci)
            for (int ky = 0; ky < kh; ++ky)
                for (int kx = 0; kx < kw; ++kx) {
                    const int k = (ky * kw + kx) * cin + ci;
                    out[((size_t)(k / 4) * Npad + n) * 4 + (k % 4)] = w[(((size_t)n * cin + ci) * kh + ky) * kw + kx];
                }
}

// eval-mode BatchNorm as y = x*alpha + beta, computed like ATen's CPU kernel (fp32):
// invstd = 1/sqrt(var+eps); alpha = invstd*gamma; beta = bias - mean*alpha
static void fold_bn(const float* gamma, const float* beta, const float* mean, const float* var, int n, float eps,
                    int npad, std::vector<float>& scale, std::vector<float>& shift) {
    scale.assign(npad, 0.f);
    shift.assign(npad, 0.f);
    for (int i = 0; i < n; ++i) {
        const float invstd = 1.0f / std::sqrt(var[i] + eps);
        const float a = invstd * gamma[i];
        scale[i] = a;
        shift[i] = beta[i] - mean[i] * a;
    }
}

const HostTensor* find(specmi_handle* h, const std::string& name) {
    auto it = h->staged.find(name);
    return it == h->staged.end() ? nullptr : &it->second;
}

int need(specmi_handle* h, const std::string& name, std::initializer_list<int64_t> shape, bool is_int,
                const HostTensor** out) {
    const HostTensor* t = find(h, name);
    if (!t) return fail(h, SPECMI_ERR_MISSING, "missing tensor '%s'", name.c_str());
    if (t->is_int != is_int) return fail(h, SPECMI_ERR_ARG, "tensor '%s' has the wrong dtype", name.c_str());
    size_t n = 1;
    for (int64_t s : shape) n *= (size_t)s;
    if (t->numel() != n) {
        return fail(h, SPECMI_ERR_ARG, "tensor '%s' has %zu elements, expected %zu", name.c_str(), t->numel(), n);
    }
    *out = t;
    return SPECMI_OK;
}

int commit_conv(specmi_handle* h, const std::string& prefix, ConvW& c) {
    const HostTensor *w, *g, *b, *m, *v;
    int rc;
    if ((rc = need(h, prefix + c.name + ".weight", {c.cout, c.cin, c.k, c.k}, false, &w))) return rc;
    if ((rc = need(h, prefix + c.bn_name + ".weight", {c.cout}, false, &g))) return rc;
    if ((rc = need(h, prefix + c.bn_name + ".bias", {c.cout}, false, &b))) return rc;
    if ((rc = need(h, prefix + c.bn_name + ".running_mean", {c.cout}, false, &m))) return rc;
    if ((rc = need(h, prefix + c.bn_name + ".running_var", {c.cout}, false, &v))) return rc;
    std::vector<float> packed, scale, shift;
    const int cin = c.cin_p > 0 ? c.cin_p : c.cin, cout = c.cout_p > 0 ? c.cout_p : c.cout;
    const float* wsrc = w->f.data();
    std::vector<float> wpad;
    if (cin != c.cin || cout != c.cout) {   // zero-padded OIHW copy: the layer is then an ordinary (cin_p -> cout_p) conv
        wpad.assign((size_t)cout * cin * c.k * c.k, 0.f);
        for (int n = 0; n < c.cout; ++n)
            for (int ci = 0; ci < c.cin; ++ci)
                std::memcpy(wpad.data() + ((size_t)n * cin + ci) * c.k * c.k, wsrc + ((size_t)n * c.cin + ci) * c.k * c.k,
                            (size_t)c.k * c.k * 4);
        wsrc = wpad.data();
    }
    if (c.cin == 3 && c.k == 7) {
        c.Kp = 148;
        c.Npad = 64;
        pack_stem_weights(wsrc, packed);
    } else if (c.cin == 3) {                // small-Cin direct convolution (hrnet.hip): plain [k = (ci*KH + ky)*KW + kx][cout]
        c.Kp = 3 * c.k * c.k;
        c.Npad = cout;
        packed.assign((size_t)c.Kp * cout, 0.f);
        for (int n = 0; n < cout; ++n)
            for (int k = 0; k < c.Kp; ++k) packed[(size_t)k * cout + n] = wsrc[(size_t)n * c.Kp + k];
    } else {
        c.Kp = cin * c.k * c.k;
        c.Npad = round_up(cout, 64);
        pack_gemm_weights(wsrc, cout, cin, c.k, c.k, c.Kp, c.Npad, packed);
    }
    fold_bn(g->f.data(), b->f.data(), m->f.data(), v->f.data(), c.cout, 1e-5f, c.Npad > cout ? c.Npad : cout, scale, shift);
    if ((rc = dev_upload(h, packed.data(), packed.size() * 4, (void**)&c.w, h->param_allocs))) return rc;
    if ((rc = dev_upload(h, scale.data(), scale.size() * 4, (void**)&c.scale, h->param_allocs))) return rc;
    if ((rc = dev_upload(h, shift.data(), shift.size() * 4, (void**)&c.shift, h->param_allocs))) return rc;
    c.wino = nullptr;
    if (c.k == 3 && c.stride == 1 && c.pad == 1 && cin % 16 == 0 && (cout % 64 == 0 || (cout % 32 == 0 && cout > 64))) {
        std::vector<float> u;
        pack_wino_weights(wsrc, cout, cin, u);
        if ((rc = dev_upload(h, u.data(), u.size() * 4, (void**)&c.wino, h->param_allocs))) return rc;
    }
    // optional split-bf16 path (conv_bf16s.hip): the weights of the plain 1x1 / stride-1 layers as three bf16 pieces
    c.wsplit = nullptr;
    if (opt_i(h, "conv_precision", 0) != 0 && (c.k == 1 || c.k == 3) && cin % 16 == 0 && cout % 4 == 0) {
        std::vector<unsigned short> pieces;
        pack_bf16_split_weights_oihw(wsrc, cout, cin, c.k, c.k, c.Npad, pieces);
        if ((rc = dev_upload(h, pieces.data(), pieces.size() * 2, &c.wsplit, h->param_allocs))) return rc;
    }
    return SPECMI_OK;
}

int commit_fused_ds(specmi_handle* h, const std::string& prefix, Bneck& b) {
    const ConvW &c3 = b.c3, &ds = b.ds;
    const HostTensor *w3, *wd, *g3, *b3, *m3, *v3, *gd, *bd, *md, *vd;
    int rc;
    if ((rc = need(h, prefix + c3.name + ".weight", {c3.cout, c3.cin, 1, 1}, false, &w3))) return rc;
    if ((rc = need(h, prefix + ds.name + ".weight", {ds.cout, ds.cin, 1, 1}, false, &wd))) return rc;
    if ((rc = need(h, prefix + c3.bn_name + ".weight", {c3.cout}, false, &g3))) return rc;
    if ((rc = need(h, prefix + c3.bn_name + ".bias", {c3.cout}, false, &b3))) return rc;
    if ((rc = need(h, prefix + c3.bn_name + ".running_mean", {c3.cout}, false, &m3))) return rc;
    if ((rc = need(h, prefix + c3.bn_name + ".running_var", {c3.cout}, false, &v3))) return rc;
    if ((rc = need(h, prefix + ds.bn_name + ".weight", {ds.cout}, false, &gd))) return rc;
    if ((rc = need(h, prefix + ds.bn_name + ".bias", {ds.cout}, false, &bd))) return rc;
    if ((rc = need(h, prefix + ds.bn_name + ".running_mean", {ds.cout}, false, &md))) return rc;
    if ((rc = need(h, prefix + ds.bn_name + ".running_var", {ds.cout}, false, &vd))) return rc;
    const int N = c3.cout, K1 = c3.cin, K2 = ds.cin, K = K1 + K2;
    const int Npad = round_up(N, 64);
    std::vector<float> s3, h3, sd, hd;
    fold_bn(g3->f.data(), b3->f.data(), m3->f.data(), v3->f.data(), N, 1e-5f, Npad, s3, h3);
    fold_bn(gd->f.data(), bd->f.data(), md->f.data(), vd->f.data(), N, 1e-5f, Npad, sd, hd);
    std::vector<float> wcat((size_t)N * K), ones(Npad, 1.f), shift(Npad, 0.f), packed;
    for (int n = 0; n < N; ++n) {
        for (int k = 0; k < K1; ++k) wcat[(size_t)n * K + k] = (float)((double)w3->f[(size_t)n * K1 + k] * (double)s3[n]);
        for (int k = 0; k < K2; ++k) wcat[(size_t)n * K + K1 + k] = (float)((double)wd->f[(size_t)n * K2 + k] * (double)sd[n]);
        shift[n] = h3[n] + hd[n];
    }
    pack_gemm_weights(wcat.data(), N, K, 1, 1, K, Npad, packed);
    b.f_Npad = Npad;
    b.f_wsplit = nullptr;
    if (opt_i(h, "conv_precision", 0) != 0 && K1 % 16 == 0 && K2 % 16 == 0 && N % 4 == 0) {
        std::vector<unsigned short> pieces;
        pack_bf16_split_weights(wcat.data(), N, K, Npad, pieces);
        if ((rc = dev_upload(h, pieces.data(), pieces.size() * 2, &b.f_wsplit, h->param_allocs))) return rc;
    }
    if ((rc = dev_upload(h, packed.data(), packed.size() * 4, (void**)&b.f_w, h->param_allocs))) return rc;
    if ((rc = dev_upload(h, ones.data(), ones.size() * 4, (void**)&b.f_scale, h->param_allocs))) return rc;
    if ((rc = dev_upload(h, shift.data(), shift.size() * 4, (void**)&b.f_shift, h->param_allocs))) return rc;
    return SPECMI_OK;
}

// One or several Linear layers stacked along the output dimension
// (nout, nin) -> (nout, Kp) row-major with zero padding: the operand of the small-batch GEMV kernel (head.hip)
static int upload_row_major(specmi_handle* h, const float* w, int nout, int nin, FcW& fc) {
    std::vector<float> rm((size_t)nout * fc.Kp, 0.f);
    for (int n = 0; n < nout; ++n) std::memcpy(rm.data() + (size_t)n * fc.Kp, w + (size_t)n * nin, (size_t)nin * 4);
    return dev_upload(h, rm.data(), rm.size() * 4, (void**)&fc.w_rm, h->param_allocs);
}

int commit_fc(specmi_handle* h, const std::vector<std::string>& names, const std::vector<int>& nouts, int nin,
                     FcW& fc) {
    int ntot = 0;
    for (int n : nouts) ntot += n;
    fc.nin = nin;
    fc.nout = ntot;
    fc.Kp = round_up(nin, 32);
    fc.Npad = round_up(ntot, 64);
    std::vector<float> wcat((size_t)ntot * nin), bias(fc.Npad, 0.f), ones(fc.Npad, 1.f);
    int row = 0, rc;
    for (size_t i = 0; i < names.size(); ++i) {
        const HostTensor *w, *b;
        if ((rc = need(h, names[i] + ".weight", {nouts[i], nin}, false, &w))) return rc;
        if ((rc = need(h, names[i] + ".bias", {nouts[i]}, false, &b))) return rc;
        std::memcpy(wcat.data() + (size_t)row * nin, w->f.data(), (size_t)nouts[i] * nin * 4);
        std::memcpy(bias.data() + row, b->f.data(), (size_t)nouts[i] * 4);
        row += nouts[i];
    }
    std::vector<float> packed;
    pack_gemm_weights(wcat.data(), ntot, nin, 1, 1, fc.Kp, fc.Npad, packed);
    if ((rc = dev_upload(h, packed.data(), packed.size() * 4, (void**)&fc.w, h->param_allocs))) return rc;
    if ((rc = dev_upload(h, ones.data(), ones.size() * 4, (void**)&fc.scale, h->param_allocs))) return rc;
    if ((rc = dev_upload(h, bias.data(), bias.size() * 4, (void**)&fc.shift, h->param_allocs))) return rc;
    return upload_row_major(h, wcat.data(), ntot, nin, fc);
}

// HMRHead in eval mode is an affine map: there is no activation between fc1, fc2 and the decoders and dropout is
// the identity (pare HMRHead = SPIN's regressor, call site spec/models/hmr.py:96-98).  With the state s = [pose6d |
// shape | cam] (157), features xf (F) and camera features c (7, use_cam_feats) one iteration is
//     s' = s + Wd (W2 (W1 [xf | s | c] + b1) + b2) + bd = (I + Q) s + P xf + T c + r,
// [P | Q | T] = Wd W2 W1, r = Wd (W2 b1 + b2) + bd, so three iterations from s0 = init_{pose,shape,cam} give
//     s3 = G (P xf + T c) + (E^3 s0 + G r),   E = I + Q,  G = I + E + E^2.
// The composition runs in float64 and is rounded once; the forward is then ONE (F+164 -> 157) GEMM instead of 9 GEMMs.
int commit_head_collapsed(specmi_handle* h, int F, int ucf) {
    const int nin = F + 157 + (ucf ? 7 : 0), NS = 157, NH = 1024;
    const HostTensor *w1, *b1, *w2, *b2, *wp, *bp, *wsh, *bsh, *wc, *bc, *ip, *is, *ic;
    int rc;
    if ((rc = need(h, "head.fc1.weight", {NH, nin}, false, &w1)) || (rc = need(h, "head.fc1.bias", {NH}, false, &b1)) ||
        (rc = need(h, "head.fc2.weight", {NH, NH}, false, &w2)) || (rc = need(h, "head.fc2.bias", {NH}, false, &b2)) ||
        (rc = need(h, "head.decpose.weight", {144, NH}, false, &wp)) || (rc = need(h, "head.decpose.bias", {144}, false, &bp)) ||
        (rc = need(h, "head.decshape.weight", {10, NH}, false, &wsh)) || (rc = need(h, "head.decshape.bias", {10}, false, &bsh)) ||
        (rc = need(h, "head.deccam.weight", {3, NH}, false, &wc)) || (rc = need(h, "head.deccam.bias", {3}, false, &bc)) ||
        (rc = need(h, "head.init_pose", {144}, false, &ip)) || (rc = need(h, "head.init_shape", {10}, false, &is)) ||
        (rc = need(h, "head.init_cam", {3}, false, &ic)))
        return rc;
    std::vector<double> Wd((size_t)NS * NH), bd(NS), s0(NS);
    for (int i = 0; i < NS; ++i) {
        const float* src = i < 144 ? wp->f.data() + (size_t)i * NH : i < 154 ? wsh->f.data() + (size_t)(i - 144) * NH
                                                                             : wc->f.data() + (size_t)(i - 154) * NH;
        for (int k = 0; k < NH; ++k) Wd[(size_t)i * NH + k] = src[k];
        bd[i] = i < 144 ? bp->f[i] : i < 154 ? bsh->f[i - 144] : bc->f[i - 154];
        s0[i] = i < 144 ? ip->f[i] : i < 154 ? is->f[i - 144] : ic->f[i - 154];
    }
    // M = Wd W2 (157 x 1024), PQT = M W1 (157 x nin), r = M b1 + Wd b2 + bd
    std::vector<double> M((size_t)NS * NH, 0.0), PQT((size_t)NS * nin, 0.0), r(NS);
    for (int i = 0; i < NS; ++i) {
        double* mi = M.data() + (size_t)i * NH;
        for (int k = 0; k < NH; ++k) {
            const double a = Wd[(size_t)i * NH + k];
            const float* w2k = w2->f.data() + (size_t)k * NH;
            for (int j = 0; j < NH; ++j) mi[j] += a * (double)w2k[j];
        }
        double* pi = PQT.data() + (size_t)i * nin;
        double ri = bd[i];
        for (int k = 0; k < NH; ++k) {
            const double a = mi[k];
            const float* w1k = w1->f.data() + (size_t)k * nin;
            for (int j = 0; j < nin; ++j) pi[j] += a * (double)w1k[j];
            ri += a * (double)b1->f[k] + Wd[(size_t)i * NH + k] * (double)b2->f[k];
        }
        r[i] = ri;
    }
    // E = I + Q, E2 = E E, G = I + E + E2, E3 = E2 E
    auto matmul = [&](const std::vector<double>& A, const std::vector<double>& Bm, std::vector<double>& C) {
        C.assign((size_t)NS * NS, 0.0);
        for (int i = 0; i < NS; ++i)
            for (int k = 0; k < NS; ++k) {
                const double a = A[(size_t)i * NS + k];
                for (int j = 0; j < NS; ++j) C[(size_t)i * NS + j] += a * Bm[(size_t)k * NS + j];
            }
    };
    std::vector<double> E((size_t)NS * NS), E2, E3, G((size_t)NS * NS);
    for (int i = 0; i < NS; ++i)
        for (int j = 0; j < NS; ++j) E[(size_t)i * NS + j] = PQT[(size_t)i * nin + F + j] + (i == j ? 1.0 : 0.0);
    matmul(E, E, E2);
    matmul(E2, E, E3);
    for (int i = 0; i < NS; ++i)
        for (int j = 0; j < NS; ++j) G[(size_t)i * NS + j] = (i == j ? 1.0 : 0.0) + E[(size_t)i * NS + j] + E2[(size_t)i * NS + j];
    // option "estimate_var": the variance decoders read xc of the LAST iteration, i.e. the state s2 after two iterations:
    //     var = Wv (W2 (W1 [xf | s2 | c] + b1) + b2) + bv = [Pv | Qv | Tv] [xf | s2 | c] + rv,   s2 = G2 (P xf + T c) + (E2 s0 + G2 r),  G2 = I + E
    // -> 154 more rows of the same affine map (composed in float64 like the rest)
    const int NV = h->has_var ? 154 : 0, NO = NS + NV;
    std::vector<double> PQTv((size_t)NV * nin, 0.0), rv(NV, 0.0);
    if (NV) {
        const HostTensor *wpv, *bpv, *wsv, *bsv;
        if ((rc = need(h, "head.decpose_var.weight", {144, NH}, false, &wpv)) || (rc = need(h, "head.decpose_var.bias", {144}, false, &bpv)) ||
            (rc = need(h, "head.decshape_var.weight", {10, NH}, false, &wsv)) || (rc = need(h, "head.decshape_var.bias", {10}, false, &bsv)))
            return rc;
        std::vector<double> mv(NH);
        for (int i = 0; i < NV; ++i) {
            const float* wrow = i < 144 ? wpv->f.data() + (size_t)i * NH : wsv->f.data() + (size_t)(i - 144) * NH;
            std::fill(mv.begin(), mv.end(), 0.0);
            for (int k = 0; k < NH; ++k) {
                const double a = wrow[k];
                const float* w2k = w2->f.data() + (size_t)k * NH;
                for (int j = 0; j < NH; ++j) mv[j] += a * (double)w2k[j];
            }
            double* pi = PQTv.data() + (size_t)i * nin;
            double ri = i < 144 ? bpv->f[i] : bsv->f[i - 144];
            for (int k = 0; k < NH; ++k) {
                const double a = mv[k];
                const float* w1k = w1->f.data() + (size_t)k * nin;
                for (int j = 0; j < nin; ++j) pi[j] += a * (double)w1k[j];
                ri += a * (double)b1->f[k] + (double)wrow[k] * (double)b2->f[k];
            }
            rv[i] = ri;
        }
    }
    // composed weight over the xc row layout [xf | state (zero columns) | c], bias = E3 s0 + G r
    FcW& fc = h->head_c;
    fc.nin = nin; fc.nout = NO; fc.Kp = round_up(nin, 32); fc.Npad = round_up(NO, 64);
    std::vector<float> wcat((size_t)NO * nin, 0.f), bias(fc.Npad, 0.f), ones(fc.Npad, 1.f), packed;
    for (int i = 0; i < NS; ++i) {
        double bi = 0.0;
        for (int k = 0; k < NS; ++k) bi += E3[(size_t)i * NS + k] * s0[k] + G[(size_t)i * NS + k] * r[k];
        bias[i] = (float)bi;
        std::vector<double> row(nin, 0.0);
        for (int k = 0; k < NS; ++k) {
            const double g = G[(size_t)i * NS + k];
            const double* pk = PQT.data() + (size_t)k * nin;
            for (int j = 0; j < F; ++j) row[j] += g * pk[j];
            for (int j = F + NS; j < nin; ++j) row[j] += g * pk[j];
        }
        for (int j = 0; j < nin; ++j) wcat[(size_t)i * nin + j] = (float)row[j];
    }
    if (NV) {
        // s2 = G2 (P xf + T c) + c2,  c2 = E2 s0 + G2 r
        std::vector<double> G2((size_t)NS * NS), c2(NS, 0.0);
        for (int i = 0; i < NS; ++i)
            for (int j = 0; j < NS; ++j) G2[(size_t)i * NS + j] = (i == j ? 1.0 : 0.0) + E[(size_t)i * NS + j];
        for (int i = 0; i < NS; ++i)
            for (int k = 0; k < NS; ++k) c2[i] += E2[(size_t)i * NS + k] * s0[k] + G2[(size_t)i * NS + k] * r[k];
        std::vector<double> qg(NS), row(nin);
        for (int i = 0; i < NV; ++i) {
            const double* pv = PQTv.data() + (size_t)i * nin;
            // qg = Qv[i] G2 (1 x 157)
            std::fill(qg.begin(), qg.end(), 0.0);
            double bi = rv[i];
            for (int k = 0; k < NS; ++k) {
                const double q = pv[F + k];
                bi += q * c2[k];
                for (int j = 0; j < NS; ++j) qg[j] += q * G2[(size_t)k * NS + j];
            }
            for (int j = 0; j < nin; ++j) row[j] = (j < F || j >= F + NS) ? pv[j] : 0.0;
            for (int k = 0; k < NS; ++k) {
                const double g = qg[k];
                const double* pk = PQT.data() + (size_t)k * nin;
                for (int j = 0; j < F; ++j) row[j] += g * pk[j];
                for (int j = F + NS; j < nin; ++j) row[j] += g * pk[j];
            }
            bias[NS + i] = (float)bi;
            for (int j = 0; j < nin; ++j) wcat[(size_t)(NS + i) * nin + j] = (float)row[j];
        }
    }
    pack_gemm_weights(wcat.data(), NO, nin, 1, 1, fc.Kp, fc.Npad, packed);
    if ((rc = dev_upload(h, packed.data(), packed.size() * 4, (void**)&fc.w, h->param_allocs))) return rc;
    if ((rc = dev_upload(h, ones.data(), ones.size() * 4, (void**)&fc.scale, h->param_allocs))) return rc;
    if ((rc = dev_upload(h, bias.data(), bias.size() * 4, (void**)&fc.shift, h->param_allocs))) return rc;
    if ((rc = upload_row_major(h, wcat.data(), NO, nin, fc))) return rc;
    h->has_head_c = true;
    return SPECMI_OK;
}

// torchvision ResNet-50 (Bottleneck [3,4,6,3], v1.5: stride on the 3x3) or ResNet-34 (BasicBlock [3,4,6,3]) trunk
void build_resnet(specmi_handle* h, int depth) {
    h->stem = ConvW();
    h->stem.name = "conv1"; h->stem.bn_name = "bn1";
    h->stem.cin = 3; h->stem.cout = 64; h->stem.k = 7; h->stem.stride = 2; h->stem.pad = 3;
    h->blocks.clear();
    // torchvision's family: BasicBlock [2,2,2,2] (18) / [3,4,6,3] (34), Bottleneck [3,4,6,3] (50) / [3,4,23,3] (101) / [3,8,36,3] (152)
    const bool basic = depth == 34 || depth == 18;
    const int nb18[4] = {2, 2, 2, 2}, nb50[4] = {3, 4, 6, 3}, nb101[4] = {3, 4, 23, 3}, nb152[4] = {3, 8, 36, 3};
    const int* nblocks = depth == 18 ? nb18 : depth == 101 ? nb101 : depth == 152 ? nb152 : nb50;
    const int planes[4] = {64, 128, 256, 512};
    const int expansion = basic ? 1 : 4;
    int inplanes = 64;
    for (int li = 0; li < 4; ++li) {
        for (int b = 0; b < nblocks[li]; ++b) {
            const int stride = (b == 0 && li > 0) ? 2 : 1;
            const std::string p = "layer" + std::to_string(li + 1) + "." + std::to_string(b);
            Bneck bn;
            bn.basic = basic;
            auto mk = [&](ConvW& c, const std::string& cn, const std::string& bnn, int cin, int cout, int k, int s,
                          int pad) {
                c.name = p + "." + cn; c.bn_name = p + "." + bnn;
                c.cin = cin; c.cout = cout; c.k = k; c.stride = s; c.pad = pad;
            };
            if (basic) {
                mk(bn.c1, "conv1", "bn1", inplanes, planes[li], 3, stride, 1);
                mk(bn.c2, "conv2", "bn2", planes[li], planes[li], 3, 1, 1);
            } else {
                mk(bn.c1, "conv1", "bn1", inplanes, planes[li], 1, 1, 0);
                mk(bn.c2, "conv2", "bn2", planes[li], planes[li], 3, stride, 1);
                mk(bn.c3, "conv3", "bn3", planes[li], planes[li] * 4, 1, 1, 0);
            }
            bn.has_ds = (stride != 1 || inplanes != planes[li] * expansion);
            if (bn.has_ds) mk(bn.ds, "downsample.0", "downsample.1", inplanes, planes[li] * expansion, 1, stride, 0);
            inplanes = planes[li] * expansion;
            h->blocks.push_back(bn);
        }
    }
    h->feat_ch = inplanes;
}


int commit_smpl(specmi_handle* h) {
    const HostTensor* vt = find(h, "smpl.v_template");
    if (!vt) return fail(h, SPECMI_ERR_MISSING, "missing tensor 'smpl.v_template'");
    if (vt->is_int || vt->numel() % 3) return fail(h, SPECMI_ERR_ARG, "smpl.v_template must be (V,3) fp32");
    const int V = (int)(vt->numel() / 3);
    const HostTensor *sd, *pd, *jr, *lw, *jx, *par, *eid, *jm;
    int rc;
    if ((rc = need(h, "smpl.shapedirs", {V, 3, 10}, false, &sd))) return rc;
    if ((rc = need(h, "smpl.posedirs", {207, (int64_t)V * 3}, false, &pd))) return rc;
    if ((rc = need(h, "smpl.J_regressor", {24, V}, false, &jr))) return rc;
    if ((rc = need(h, "smpl.lbs_weights", {V, 24}, false, &lw))) return rc;
    if ((rc = need(h, "smpl.J_regressor_extra", {9, V}, false, &jx))) return rc;
    if ((rc = need(h, "smpl.parents", {24}, true, &par))) return rc;
    if ((rc = need(h, "smpl.extra_vertex_ids", {21}, true, &eid))) return rc;
    if ((rc = need(h, "smpl.joint_map", {49}, true, &jm))) return rc;
    for (int j = 1; j < 24; ++j)
        if (par->i[j] < 0 || par->i[j] >= j)
            return fail(h, SPECMI_ERR_ARG, "smpl.parents[%d]=%d: a parent must precede its child", j, par->i[j]);
    for (int e = 0; e < 21; ++e)
        if (eid->i[e] < 0 || eid->i[e] >= V) return fail(h, SPECMI_ERR_ARG, "smpl.extra_vertex_ids[%d] out of range", e);
    for (int e = 0; e < 49; ++e)
        if (jm->i[e] < 0 || jm->i[e] >= 54) return fail(h, SPECMI_ERR_ARG, "smpl.joint_map[%d] out of range", e);
    // rest-pose joint regression is linear in beta: J = Jr@v_template + (Jr@shapedirs) beta,
    // folded once in float64 (24 x V x 33 MACs)
    std::vector<float> Jt(72), Jd(720);
    for (int j = 0; j < 24; ++j) {
        double acc[33] = {0};
        for (int v = 0; v < V; ++v) {
            const double wv = jr->f[(size_t)j * V + v];
            if (wv == 0.0) continue;
            for (int c = 0; c < 3; ++c) {
                acc[c] += wv * vt->f[(size_t)v * 3 + c];
                for (int l = 0; l < 10; ++l) acc[3 + c * 10 + l] += wv * sd->f[((size_t)v * 3 + c) * 10 + l];
            }
        }
        for (int c = 0; c < 3; ++c) {
            Jt[j * 3 + c] = (float)acc[c];
            for (int l = 0; l < 10; ++l) Jd[(j * 3 + c) * 10 + l] = (float)acc[3 + c * 10 + l];
        }
    }
    SmplDev& m = h->smpl;
    m.V = V;
    std::vector<int32_t> parents = par->i;
    parents[0] = -1;
    {   // the skinning kernel's operands in MFMA fragment order (smpl.hip): [posedirs ; shapedirs ; v_template] per
        // (vertex group, coordinate) and the skinning weights per vertex group; vertices past V are zero rows
        const int G = (V + 31) / 32;
        const size_t tile = (size_t)SMPL_KQ * 256;
        std::vector<float> dirs((size_t)G * 3 * tile, 0.f), wt((size_t)G * 768, 0.f);
        for (int v = 0; v < V; ++v) {
            const int g = v / 32, n = v % 32;
            for (int c = 0; c < 3; ++c) {
                float* d = dirs.data() + ((size_t)g * 3 + c) * tile;
                for (int k = 0; k < 207; ++k) d[frag_slot(k, n)] = pd->f[(size_t)k * V * 3 + (size_t)v * 3 + c];
                for (int l = 0; l < 10; ++l) d[frag_slot(207 + l, n)] = sd->f[((size_t)v * 3 + c) * 10 + l];
                d[frag_slot(217, n)] = vt->f[(size_t)v * 3 + c];
            }
            for (int j = 0; j < 24; ++j) wt[(size_t)g * 768 + frag_slot(j, n)] = lw->f[(size_t)v * 24 + j];
        }
        if ((rc = dev_upload(h, dirs.data(), dirs.size() * 4, (void**)&m.dirsT, h->param_allocs))) return rc;
        if ((rc = dev_upload(h, wt.data(), wt.size() * 4, (void**)&m.wT, h->param_allocs))) return rc;
    }
    if ((rc = dev_upload(h, jx->f.data(), jx->f.size() * 4, (void**)&m.J_extra, h->param_allocs))) return rc;
    if ((rc = dev_upload(h, Jt.data(), Jt.size() * 4, (void**)&m.J_template, h->param_allocs))) return rc;
    if ((rc = dev_upload(h, Jd.data(), Jd.size() * 4, (void**)&m.J_shapedirs, h->param_allocs))) return rc;
    if ((rc = dev_upload(h, parents.data(), 24 * 4, (void**)&m.parents, h->param_allocs))) return rc;
    if ((rc = dev_upload(h, eid->i.data(), 21 * 4, (void**)&m.extra_ids, h->param_allocs))) return rc;
    if ((rc = dev_upload(h, jm->i.data(), 49 * 4, (void**)&m.joint_map, h->param_allocs))) return rc;
    return SPECMI_OK;
}

