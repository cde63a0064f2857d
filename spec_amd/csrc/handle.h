// handle.h - the library handle and the host-side helpers shared by api.hip (ResNet trunks, heads, C ABI) and
// hrnet.hip (HRNet trunks).  Internal to libspecmi.so.
#pragma once
#include <cstdarg>
#include <initializer_list>

#include "specmi_internal.h"

using namespace specmi;

// ------------------------------------------------------------------------------------------
// handle
// ------------------------------------------------------------------------------------------
struct HostTensor {
    std::vector<float> f;
    std::vector<int32_t> i;
    std::vector<int64_t> shape;
    bool is_int = false;
    size_t numel() const { return is_int ? i.size() : f.size(); }
};

struct ConvW {
    std::string name;  // state-dict prefix of the conv ("layer1.0.conv1"), bn under bn_name
    std::string bn_name;
    int cin = 0, cout = 0, k = 1, stride = 1, pad = 0;
    // channel counts of the activation tensors this layer reads / writes when they are padded to a multiple of 32
    // (HRNet-W48: 48 -> 64); 0 = same as cin / cout.  Padded weights are zero, padded outputs come out as 0.
    int cin_p = 0, cout_p = 0;
    int Kp = 0, Npad = 0;
    float *w = nullptr, *scale = nullptr, *shift = nullptr;  // device
    float* wino = nullptr;  // device: Winograd-domain filters (3x3 stride-1 layers only)
    void* wsplit = nullptr; // device: bf16 pieces of the weights (1x1 stride-1 layers, only when option conv_precision != 0)
};

struct Bneck {
    ConvW c1, c2, c3, ds;
    bool has_ds = false;
    bool basic = false;   // torchvision BasicBlock (ResNet-18/34): c1 = 3x3 (stride), c2 = 3x3, no c3
    // conv3 + bn3 and downsample conv + bn folded into ONE 1x1 GEMM over [conv2 output | block input]:
    // weights pre-multiplied by the BN scales (fp64), shift = shift3 + shift_ds, scale = 1
    float *f_w = nullptr, *f_scale = nullptr, *f_shift = nullptr;
    void* f_wsplit = nullptr;   // bf16 pieces of the same folded matrix (option conv_precision != 0)
    int f_Npad = 0;
};

struct FcW {  // Linear layers as H=W=1 convolutions
    int nin = 0, nout = 0, Kp = 0, Npad = 0;
    float *w = nullptr, *scale = nullptr, *shift = nullptr;
    float* w_rm = nullptr;   // the same matrix as (nout, Kp) row-major rows, zero padded to Kp: the small-batch GEMV kernel's operand
};

struct HrNet;

struct specmi_handle {
    int device = 0;
    int kind = 0;
    std::map<std::string, HostTensor> staged;
    std::map<std::string, int> opt_i;
    std::map<std::string, float> opt_f;
    bool committed = false;
    std::string err;

    // packed parameters (device)
    ConvW stem;
    std::vector<Bneck> blocks;
    FcW fc_cam[3][3];          // CamCalib: vfov, pitch, roll x up to 3 stacked Linear layers (camcalib/model.py:59-70: no activation between them)
    int fc_layers = 1, feat_ch = 2048;
    FcW fc1, fc2, dec;         // HMR head (dec = decpose|decshape|deccam)
    FcW head_c;                // the 3 IEF iterations composed into ONE affine map [xf | cam feats] -> 157 (commit_head_collapsed)
    bool has_head_c = false;
    // option "estimate_var" (HMRHead's uncertainty outputs, spec/models/hmr.py:35-38,57-64): the variance decoders
    // [decpose_var (144) ; decshape_var (10)] for the nine-GEMM loop; the collapsed map carries them as 154 extra output rows
    FcW head_var;
    bool has_var = false;
    // where the last head forward left the regressed state [pose6d | shape | cam] and the raw variance columns (specmi_hmr_uncertainty)
    const float* last_state = nullptr; long last_ld_state = 0;
    const float* last_var = nullptr; long last_ld_var = 0;
    int last_B = 0;
    int xc_ld = 2240;          // row stride of the IEF state [xf (feat_ch) | pose6d | shape | cam | rot6d(R) | vfov | 0-pad]
    float *init_pose = nullptr, *init_shape = nullptr, *init_cam = nullptr;
    SmplDev smpl;
    HrNet* hrnet = nullptr;    // HRNet trunk (HMR option "backbone" = 32 / 48) instead of the ResNet blocks
    std::vector<void*> param_allocs;

    // workspace (device), grown on demand
    std::vector<void*> ws_allocs;
    std::vector<void*> ws_retired;      // outgrown workspaces, kept until destroy (captured graphs may still name them)
    float* act[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t act_elems = 0;
    float *xc = nullptr, *h1 = nullptr, *h2 = nullptr, *xf = nullptr;
    float* fc_hidden[2] = {nullptr, nullptr};   // (3, B, 1024) each: hidden rows of the three CamCalib Linear chains (latency plan)
    float *rot_ws = nullptr, *betas_ws = nullptr, *cam_ws = nullptr, *verts_ws = nullptr;
    float *pf_ws = nullptr, *A_ws = nullptr, *pj_ws = nullptr;
    int ws_B = 0;
    int* resize_tab = nullptr;          // device copy of the Pillow coefficient tables of the last resize geometry
    size_t resize_tab_ints = 0;
    std::vector<int> resize_host;       // host image of the same (kept alive for the async copy)
    int resize_geom[4] = {0, 0, 0, 0};  // H, W, OH, OW the tables were built for
    SkWs sk;                            // split-K partial tiles + arrival counters (ensure_sk; never allocated under graph capture:
                                        // the warm-up call of a shape sizes it)
    std::vector<void*> sk_retired;      // outgrown split-K buffers, kept until destroy (captured graphs may still name them)
    // persistent multi-layer launches (conv_persist.hip): device tables keyed by their host image (a table names workspace
    // buffers, weights and shapes - everything but the caller's feature buffer, which travels as a launch argument - so equal
    // images mean the cached device copy is still right; captured graphs keep naming superseded tables: never freed before destroy)
    struct PersistTable { std::vector<unsigned char> img; void* dev = nullptr; int nl = 0; };
    std::vector<PersistTable> persist_tables;
    PersistCtl* pctl = nullptr;         // device control block of this handle's persistent launches (one launch at a time per handle)
    unsigned* tail_ctl = nullptr;       // device: counters of the fused tails (head.hip: tail_gemv_kernel), kTailCtlWords zeroed words
    static constexpr int kTailCtlWords = 64;

    Profiler prof;
};

int fail(specmi_handle* h, int code, const char* fmt, ...);

#define HIPCHK(h, call)                                                                         \
    do {                                                                                        \
        hipError_t e__ = (call);                                                                \
        if (e__ != hipSuccess)                                                                  \
            return fail(h, SPECMI_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), \
                        __FILE__, __LINE__);                                                    \
    } while (0)

#define LAUNCHCHK(h, rc, what)                                                                   \
    do {                                                                                        \
        int rc__ = (rc);                                                                        \
        if (rc__ != 0)                                                                          \
            return fail(h, SPECMI_ERR_HIP, "launch %s failed: %s", what,                        \
                        hipGetErrorString((hipError_t)rc__));                                   \
    } while (0)


// ---- shared host helpers (api.hip; find / need: commit.hip; opt_i / opt_f: options.hip) -------------------------------------------------------------------------
int round_up(int x, int m);
int conv_out(int x, int k, int s, int p);
int dev_upload(specmi_handle* h, const void* src, size_t bytes, void** out, std::vector<void*>& pool);
int dev_alloc(specmi_handle* h, size_t bytes, void** out, std::vector<void*>& pool);
void free_pool(std::vector<void*>& pool);
const HostTensor* find(specmi_handle* h, const std::string& name);
int need(specmi_handle* h, const std::string& name, std::initializer_list<int64_t> shape, bool is_int, const HostTensor** out);
int opt_i(specmi_handle* h, const char* name, int dflt);
float opt_f(specmi_handle* h, const char* name, float dflt);
// ---- commit.hip: packing, BatchNorm folding, the composed regressor, layer tables, SMPL constants ----------
// conv weight + eval-mode BatchNorm under state-dict names prefix + c.name / c.bn_name -> packed device tensors
int commit_conv(specmi_handle* h, const std::string& prefix, ConvW& c);
// OIHW -> [Kp/4][Npad][4], k = (ky*KW + kx)*Cin + ci (zero padded): the MFMA B-fragment layout of the implicit-GEMM kernels
void pack_gemm_weights(const float* w, int cout, int cin, int kh, int kw, int Kp, int Npad, std::vector<float>& out);
int commit_fused_ds(specmi_handle* h, const std::string& prefix, Bneck& b);
int commit_fc(specmi_handle* h, const std::vector<std::string>& names, const std::vector<int>& nouts, int nin, FcW& fc);
int commit_head_collapsed(specmi_handle* h, int F, int ucf);
void build_resnet(specmi_handle* h, int depth);
int commit_smpl(specmi_handle* h);

// ---- HRNet trunks (hrnet.hip) ------------------------------------------------------------------------------
struct HrNet;
void hrnet_free(HrNet* n);
int hrnet_commit(specmi_handle* h, const std::string& prefix, int width, int use_conv);
// images NCHW -> (B, H/32, W/32, feat_ch) NHWC in feat_out
const float* hrnet_feat_ws(specmi_handle* h);   // the internal (B, fh, fw, feat_ch) buffer used when feat_out == NULL
int hrnet_forward(specmi_handle* h, const float* images, int B, int H, int W, float* feat_out, int* fh, int* fw, hipStream_t s);
