// Internal declarations shared by the HIP translation units of libspecmi.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/specmi.h"

namespace specmi {

// ----------------------------------------------------------------------------------------
// launch profiler: HIP events on the launch stream around every kernel
// ----------------------------------------------------------------------------------------
struct ProfRecord {
    const char* kernel;
    std::string label;
    double flops, bytes;
    hipEvent_t e0, e1;
};

struct Profiler {
    bool on = false;
    std::vector<ProfRecord> log;
    std::string scope;  // label prefix set by the caller ("backbone.layer1.0.conv1")
};

struct LaunchCtx {
    hipStream_t stream;
    Profiler* prof;
    const char* label;
};

struct ProfScope {  // RAII: records e0 at construction, e1 at destruction
    Profiler* p;
    hipStream_t s;
    ProfRecord r;
    bool active;
    ProfScope(const LaunchCtx& c, const char* kernel, double flops, double bytes)
        : p(c.prof), s(c.stream), active(c.prof && c.prof->on) {
        if (!active) return;
        r.kernel = kernel;
        r.label = c.label ? c.label : "";
        r.flops = flops;
        r.bytes = bytes;
        (void)hipEventCreate(&r.e0);
        (void)hipEventCreate(&r.e1);
        (void)hipEventRecord(r.e0, s);
    }
    ~ProfScope() {
        if (!active) return;
        (void)hipEventRecord(r.e1, s);
        p->log.push_back(r);
    }
};

// hipFuncSetAttribute is a per-DEVICE setting: remember per (kernel instantiation, device) whether the
// dynamic-LDS limit was raised (a second device in the same process must get its own call).
struct DevOnce {
    bool done[64] = {};
};
inline int set_dyn_lds_once(DevOnce& once, const void* fn, int bytes) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    if (dev >= 0 && dev < 64 && once.done[dev]) return 0;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return (int)e;
    if (dev >= 0 && dev < 64) once.done[dev] = true;
    return 0;
}

// ----------------------------------------------------------------------------------------
// implicit-GEMM convolution / linear layer on fp32 MFMA  (conv_igemm.hip)
// ----------------------------------------------------------------------------------------
struct ConvArgs {
    const float* x;      // NHWC activations; pixel stride ldx floats
    const float* w;      // packed weights [Kp/4][Npad][4], k = (ky*KW+kx)*Cin + ci
    const float* scale;  // [Npad]
    const float* shift;  // [Npad]
    const float* res;    // optional residual, indexed like out (may alias out)
    float* out;          // [M][ldo]
    int B, H, W, Cin, ldx;
    int OH, OW, Cout, Npad, ldo;
    int KH, KW, stride, pad;
    int relu;
    // optional second A source of a 1x1 layer: out = [x | x2(strided)] * w, K = Cin + Cin2 (rows k >= Cin of w);
    // used to fold a bottleneck's downsample branch into its conv3
    const float* x2 = nullptr;
    int H2 = 0, W2 = 0, ldx2 = 0, Cin2 = 0, stride2 = 1;
    // per-handle tuning overrides (tests / tools): 0 = automatic choice
    int force_variant = 0;   // conv_igemm tile (1: 128x128/4 waves, 2: 128x64/4, 3: 64x64/4, 4: 128x128/8 waves)
    int wino_variant = 0;    // conv_wino frequencies per wave (16 / 8)
};
// Cin % 32 == 0, Npad % 64 == 0.  Returns hipError as int.
// b != nullptr: the same layer shape of a SECOND network (own tensors) in the same launch (gridDim.z = 2)
int launch_conv_igemm(const ConvArgs& a, const LaunchCtx& ctx, const ConvArgs* b = nullptr);
// Split-K (conv_igemm.hip header): K slices as gridDim.y of ONE launch; the last slice of a tile to arrive adds the partial
// tiles in slice order and runs the epilogue (deterministic, batch-invariant).
struct SkWs {
    float* ws = nullptr;       // partial tiles: conv_igemm_sk_ws_floats() floats
    size_t floats = 0;
    unsigned* cnt = nullptr;   // one arrival counter per (tile, group), zero between launches
    int ncnt = 0;
};
// The canonical k-sum tree of a sliced layer and the share of it one workgroup computes (conv_igemm.hip header)
struct SkPlan {
    int leaves = 1;   // K = leaves x L chunks; a leaf is one MFMA chain from +0
    int G = 1;        // consecutive leaves that fold into a group (the groups fold into the result)
    int unit = 1;     // leaves per workgroup: 1, G or leaves (= no slabs) - a speed choice, the bits are the same
};
// leaves of the FC GEMMs (every plan; 1 = don't split)
int conv_igemm_splitk_plan(const ConvArgs& a);
// leaves of a convolution in the latency plan: depends on the layer's per-image shape only, never on the batch
int conv_igemm_sk_slices(const ConvArgs& a, int target_wgs, int min_chunks);
// the same + the unit for batch a.B (groups = 2: grouped launch of two networks)
SkPlan conv_igemm_sk_plan(const ConvArgs& a, int groups, int target_wgs, int min_chunks, int fill_wgs);
size_t conv_igemm_sk_ws_floats(const ConvArgs& a, int slabs, int groups);
int conv_igemm_sk_tiles(const ConvArgs& a, int groups);
// conv_igemm_sk_check / the sliced launchers: the tensors pass 32-bit buffer addressing and these launchers do not split the batch -
// the ONE condition on which the caller may take the throughput launcher instead (every other non-zero result is an error)
constexpr int SK_NEEDS_BATCH_SPLIT = -1001;
int launch_conv_igemm_sk(const ConvArgs& a, const SkPlan& pl, const SkWs& sk, const LaunchCtx& ctx, const ConvArgs* b = nullptr);
// the wave-split unit of the same tree (conv_wsplit.hip): a 32x32 tile per workgroup, the leaves of a group on its four waves;
// pl.unit = pl.leaves (no slabs) or pl.G (one group per workgroup, a 4 KB slab per group)
bool conv_wsplit_supported(const ConvArgs& a, const SkPlan& pl);
int conv_wsplit_tiles(const ConvArgs& a, int groups);
size_t conv_wsplit_ws_floats(const ConvArgs& a, int slabs, int groups);
int launch_conv_wsplit(const ConvArgs& a, const SkPlan& pl, const SkWs& sk, const LaunchCtx& ctx, const ConvArgs* b = nullptr);
// pick the tile the launcher would use (for tests / labels)
const char* conv_igemm_variant(const ConvArgs& a);

// ----------------------------------------------------------------------------------------
// a run of consecutive sliced convolutions as ONE persistent launch  (conv_persist.hip)
// ----------------------------------------------------------------------------------------
constexpr int kPersistMaxLayers = 64;
struct PersistCtl {                            // device; zero between launches (the kernel leaves it so)
    unsigned done[2 * kPersistMaxLayers];      // finished tiles of (layer, network)
    unsigned exit;                             // workgroups that have left
    unsigned err;                              // != 0: a bounded spin gave up (protocol error; results of that launch are garbage)
};
struct PersistLayerHost {
    ConvArgs a, b;       // the fused convolution of network 0 / network 1 (pair)
    bool pair = false;
    SkPlan pl;           // canonical tree + unit (leaves = 1: unsliced)
    int out_arg = 0;     // 1: the output pointer comes with the launch (caller-owned feature buffer), not from the table
};
size_t persist_table_bytes(int nl);
// img == nullptr: only sizes what sk must hold.  Otherwise writes the table (host image, persist_table_bytes(nl)) for a grid of nwg
int persist_fill_table(const PersistLayerHost* layers, int nl, const SkWs& sk, int nwg, int l2_prefetch, void* img,
                       size_t* ws_floats_needed, int* cnt_needed);
int launch_persist(const void* dev_table, int nl, PersistCtl* ctl, float* out0, float* out1, int nwg, unsigned spin_limit,
                   const LaunchCtx& ctx, double flops, double bytes, bool allow_full = false);

// ----------------------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1 convolution as Winograd F(2x2,3x3) on fp32 MFMA  (conv_wino.hip)
// ----------------------------------------------------------------------------------------
// a.w must point at pack_wino_weights() output; Cin % 16 == 0, Cout % 64 == 0, no residual.
bool conv_wino_supported(const ConvArgs& a);
void pack_wino_weights(const float* w_oihw, int cout, int cin, std::vector<float>& out);
int launch_conv_wino(const ConvArgs& a, const LaunchCtx& ctx, const ConvArgs* b = nullptr);   // b: see launch_conv_igemm

// ----------------------------------------------------------------------------------------
// OPTIONAL: 1x1 / stride-1 convolution with fp32-class results on the bf16 matrix cores  (conv_bf16s.hip)
// ----------------------------------------------------------------------------------------
// w_oi (cout, cin) fp32 -> bf16 pieces w0 + w1 + w2 packed [piece][cin/8][Npad][8]
void pack_bf16_split_weights(const float* w_oi, int cout, int cin, int Npad, std::vector<unsigned short>& out);
void pack_bf16_split_weights_oihw(const float* w_oihw, int cout, int cin, int kh, int kw, int Npad, std::vector<unsigned short>& out);
bool conv_bf16s_supported(const ConvArgs& a);
// terms = 6 (fp32-class) or 3 (~1e-5); a.w is ignored, wsplit = the packed pieces on the device
int launch_conv_bf16s(const ConvArgs& a, const void* wsplit, int terms, const LaunchCtx& ctx);

// ----------------------------------------------------------------------------------------
// stem + pooling  (stem.hip)
// ----------------------------------------------------------------------------------------
// x NCHW (B,3,H,W); w = pack_stem_weights() output; out NHWC (B,OH,OW,64), BN+ReLU.
void pack_stem_weights(const float* w_oihw, std::vector<float>& out);
// pair / (x1, out1): the same op of a SECOND network in the same launch (see launch_conv_igemm)
struct StemPair { const float *x, *w, *scale, *shift; float* out; };
int launch_stem(const float* x, const float* w, const float* scale, const float* shift,
                float* out, int B, int H, int W, int OH, int OW, int relu, const LaunchCtx& ctx, const StemPair* pair = nullptr);
int launch_maxpool3x3s2(const float* x, float* out, int B, int H, int W, int C, int OH, int OW,
                        const LaunchCtx& ctx, const float* x1 = nullptr, float* out1 = nullptr);
// IEF state row of image b at xc + b*ld: [xf (F) | pose6d 144 | shape 10 | cam 3 | rot6d(R) 6 | vfov 1 | 0-pad], F = trunk
// features (2048 for ResNet-50: ld = 2240), state_off = F  (head.hip: head_init_kernel; stem.hip: extra workgroups of the avg-pool)
struct HeadInit {
    float* xc; const float *init_pose, *init_shape, *init_cam, *R, *K, *img_h;
    int use_cam_feats, state_off, ld;
};
// columns state_off .. ld of row b, by `nthreads` threads.  WT: write-through (agent-scope) stores - the row is read later in the
// SAME launch by other workgroups (fused tail, head.hip)
template <bool WT = false>
__device__ __forceinline__ void head_init_row(const HeadInit& a, int b, int tid, int nthreads) {
    float* row = a.xc + (size_t)b * a.ld + a.state_off;
    for (int i = tid; i < a.ld - a.state_off; i += nthreads) {
        float v = 0.f;
        if (i < 144) v = a.init_pose[i];
        else if (i < 154) v = a.init_shape[i - 144];
        else if (i < 157) v = a.init_cam[i - 154];
        else if (a.use_cam_feats && i < 163) {
            const int e = i - 157;                   // rotmat[:, :, :2] row-major: (row, col) = (e/2, e%2)
            v = a.R[(size_t)b * 9 + (e >> 1) * 3 + (e & 1)];
        } else if (a.use_cam_feats && i == 163) {
            v = 2.0f * atanf(a.img_h[b] / (2.0f * a.K[(size_t)b * 9]));
        }
        if (WT) __hip_atomic_store(row + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else row[i] = v;
    }
}
// x (B,HW,C) -> out[b*ldo + c] = mean_hw.  init != nullptr: the same launch also writes the IEF state columns of every row
// (B more workgroups; one graph node less on the small-batch path) - returns 1 in *init_done if it did
int launch_avgpool(const float* x, float* out, int B, int HW, int C, int ldo, const LaunchCtx& ctx, const HeadInit* init = nullptr,
                   bool* init_done = nullptr);

// ----------------------------------------------------------------------------------------
// heads  (head.hip)
// ----------------------------------------------------------------------------------------
int launch_head_init(float* xc, const float* init_pose, const float* init_shape,
                     const float* init_cam, const float* cam_rotmat, const float* cam_intrinsics,
                     const float* img_h, int use_cam_feats, int B, int state_off, int ld, const LaunchCtx& ctx);
// rot6d_to_rotmat of one joint (pare: Gram-Schmidt, F.normalize eps 1e-12); p = the joint's 6 numbers viewed (3, 2):
// a1 = p[0], p[2], p[4]; a2 = p[1], p[3], p[5]; Rm row-major with columns b1 b2 b3
__device__ __forceinline__ void rot6d_joint(const float* p, float* Rm) {
    const float a1x = p[0], a1y = p[2], a1z = p[4];
    const float a2x = p[1], a2y = p[3], a2z = p[5];
    const float n1 = fmaxf(sqrtf(a1x * a1x + a1y * a1y + a1z * a1z), 1e-12f);
    const float b1x = a1x / n1, b1y = a1y / n1, b1z = a1z / n1;
    const float d = b1x * a2x + b1y * a2y + b1z * a2z;
    const float ux = a2x - d * b1x, uy = a2y - d * b1y, uz = a2z - d * b1z;
    const float n2 = fmaxf(sqrtf(ux * ux + uy * uy + uz * uz), 1e-12f);
    const float b2x = ux / n2, b2y = uy / n2, b2z = uz / n2;
    const float b3x = b1y * b2z - b1z * b2y;
    const float b3y = b1z * b2x - b1x * b2z;
    const float b3z = b1x * b2y - b1y * b2x;
    Rm[0] = b1x; Rm[1] = b2x; Rm[2] = b3x; Rm[3] = b1y; Rm[4] = b2y; Rm[5] = b3y; Rm[6] = b1z; Rm[7] = b2z; Rm[8] = b3z;
}
// state: 157 regressor outputs per image at stride ld_state; ld_* = per-image strides of pred_pose / pred_shape /
// pred_cam / pred_pose_6d (216 / 10 / 3 / 144 when dense); null outputs are skipped
struct HeadFinal {
    const float* state = nullptr; long ld_state = 0;
    float *pred_pose = nullptr, *pred_shape = nullptr, *pred_cam = nullptr, *pred_pose_6d = nullptr;
    long ld_pose = 216, ld_shape = 10, ld_cam = 3, ld_p6d = 144;
    float *rot_ws = nullptr, *betas_ws = nullptr, *cam_ws = nullptr;
};
int launch_head_final(const float* state, long ld_state, float* pred_pose, float* pred_shape, float* pred_cam,
                      float* pred_pose_6d, const long ld[4], float* rotmat_ws, float* betas_ws, float* cam_ws,
                      int B, const LaunchCtx& ctx);
// pred_pose_var (B, 288) = [pose6d | act(var_pose)], pred_shape_var (B, 20) = [shape | act(var_shape)] (pare HMRHead, estimate_var);
// act: 0 none, 1 relu, 2 softplus, 3 sigmoid, 4 tanh, 5 elu (torch.nn.functional defaults)
int launch_head_var(const float* state, long ld_state, const float* var, long ld_var, int act, float* pose_var, float* shape_var,
                    int B, const LaunchCtx& ctx);
// ld_ang: per-image stride of vfov / pitch / roll (1 when dense)
int launch_camcalib_decode(const float* lv, const float* lp, const float* lr, int B, int nbins,
                           const float* img_h, const float* img_w, float* vfov, float* pitch,
                           float* roll, float* f_pix, float* R, float* K, long ld_ang, const LaunchCtx& ctx);
// per-row argmax (first maximum, NumPy NaN semantics) and / or normalised soft-argmax of (rows, nbins) logits
int launch_bins_reduce(const float* x, int rows, int nbins, int* idx, float* soft, const LaunchCtx& ctx);

// FC layers at small batch: one launch for up to three heads, one wave per output column (head.hip).  w = (N, Kp) row-major,
// zero padded to Kp; x rows ldx floats apart with Kp floats readable; out[b * ldo + n] = w[n] . x[b] + bias[n] (+ res[b * ldo + n])
struct FcGemv { const float* x; const float* w; const float* bias; const float* res; float* out; };
int launch_fc_gemv(const FcGemv* heads, int nheads, int N, int Kp, int ldx, int ldo, int B, const LaunchCtx& ctx);

// fused tails of the small-batch step (head.hip: tail_gemv_kernel): pool [+ state init] -> GEMV heads -> CamCalib decode | HMR pose chain
struct DecodeArgs;
struct PoseTail;
int launch_tail_camcalib(const FcGemv* heads, int N, int Kp, int B, const float* map, float* pooled, int HW, int C, unsigned* ctl, int ctl_words,
                         const float* img_h, const float* img_w, float* vfov, float* pitch, float* roll, float* f_pix, float* R, float* K,
                         long ld_ang, const LaunchCtx& ctx);
struct SmplDev;
int launch_tail_hmr(const FcGemv& head, int N, int Kp, int ldx, int ldo, int B, const float* map, float* xc, int HW, int C, const HeadInit& init,
                    unsigned* ctl, int ctl_words, const SmplDev& m, float* feat, float* Afrag, float* posed_j, const HeadFinal& fin,
                    const LaunchCtx& ctx);

int launch_cam_params(const float* pitch, const float* roll, const float* f_pix, const float* img_w, const float* img_h,
                      int B, float* R, float* K, const LaunchCtx& ctx);

// ----------------------------------------------------------------------------------------
// SMPL  (smpl.hip)
// ----------------------------------------------------------------------------------------
// Operands of the skinning kernel live in MFMA fragment order.  Slot of element (k, n) of a K x 32 operand of
// v_mfma_f32_32x32x2_f32 (n = row of an A operand / column of a B operand) stored as [quad][lane][4]: MFMA step st = k / 2 reads
// register element st % 4 of the 16-byte quad st / 4; lane = (k % 2) * 32 + n.
constexpr int SMPL_KQ = 28;   // K = 224 = 207 pose features + 10 betas + 1 (v_template) + 6 zeros = 112 steps = 28 quads
__host__ __device__ inline size_t frag_slot(int k, int n) {
    const int st = k >> 1;
    return ((size_t)(st >> 2) * 64 + (k & 1) * 32 + n) * 4 + (st & 3);
}

struct SmplDev {
    int V = 0;
    // [group of 32 vertices][coordinate][frag_slot(k, vertex % 32)], k: 0..206 posedirs, 207..216 shapedirs, 217 v_template
    float* dirsT = nullptr;
    float* wT = nullptr;           // [group][frag_slot(joint, vertex % 32)]: lbs_weights, 768 floats per group
    float* J_template = nullptr;   // (24,3)   = J_regressor @ v_template      (fp64 on host)
    float* J_shapedirs = nullptr;  // (24,3,10) = J_regressor @ shapedirs      (fp64 on host)
    float* J_extra = nullptr;      // (9,V)
    int* parents = nullptr;        // (24)
    int* extra_ids = nullptr;      // (21)
    int* joint_map = nullptr;      // (49)
};
struct SmplArgs {
    const float* rotmat;  // (B,24,3,3)
    const float* betas;   // (B,10)
    const float* cam;     // (B,3)
    const float* cam_rotmat;      // (B,3,3) or null (mode 1)
    const float* cam_intrinsics;  // (B,3,3) or null (mode 1)
    const float* bbox_scale;
    const float* bbox_center;
    const float* img_w;
    const float* img_h;
    float* vertices;   // (B,V,3)
    float* joints3d;   // (B,49,3)
    float* joints2d;   // (B,49,2)
    float* cam_t;      // (B,3)
    long ld_verts = 0, ld_j3d = 147, ld_j2d = 98, ld_camt = 3;   // per-image strides (0: V*3)
    // workspace
    float* pose_feat;  // (B,208)
    float* A;          // (B,24,12)
    float* posed_j;    // (B,24,3)
    int B;
    int mode;          // 0: SMPLCamHead (full-image camera), 1: SMPLHead (weak perspective)
    float focal_length;
    float img_res;
    int normalize_joints2d;
    int skin_split = -1;   // -1: by batch, 0 / 1: never / always three waves per vertex group (same bits)
    // non-null: the pose kernel first does head_final's work for its image (rot6d -> rotmat, output gather) and takes rotmat /
    // betas from there instead of a.rotmat / a.betas (one graph node less; same bits)
    const HeadFinal* final_ = nullptr;
    bool pose_done = false;   // the pose chains of this batch already ran (fused HMR tail, head.hip): launch_smpl starts at the skinning
};
int launch_smpl(const SmplDev& m, const SmplArgs& a, const LaunchCtx& ctx);
// vertices (optional) + the 24 posed kinematic joints (optional; a.posed_j receives them otherwise)
int launch_smpl_native(const SmplDev& m, const SmplArgs& a, float* joints24, const LaunchCtx& ctx);
// axis-angle (n,3) -> rotation matrices (n,3,3), smplx batch_rodrigues
int launch_rodrigues(const float* aa, float* rot, int n, const LaunchCtx& ctx);

// ----------------------------------------------------------------------------------------
// evaluation metrics  (eval.hip)
// ----------------------------------------------------------------------------------------
int launch_eval_mesh(const float* pred, const float* gt, int B, int V, const float* Jr, int J, const int* sel, int nsel,
                     float* mpjpe, float* pampjpe, float* v2v, const LaunchCtx& ctx);
int launch_eval_joints(const float* pred, const float* gt, int B, int J, float* mpjpe, float* pampjpe,
                       const LaunchCtx& ctx);
int launch_regress_joints(const float* verts, int B, int V, const float* Jr, int J, float* out, const LaunchCtx& ctx);
int launch_rotate_points(const float* R, const float* x, int B, int N, float* out, const LaunchCtx& ctx);

// ----------------------------------------------------------------------------------------
// crop + normalise  (preprocess.hip)
// ----------------------------------------------------------------------------------------
// frame_of != nullptr: `frame` is a slab of nframes equal-sized frames and crop d is cut from frame frame_of[d] (device, (n) int32)
int launch_crop_normalize(const unsigned char* frame, int H, int W, const float* bboxes, int n, float scale, int S,
                          float* out, unsigned char* raw, float* bbox_scale, float* bbox_center, const LaunchCtx& ctx,
                          const int* frame_of = nullptr, int nframes = 1);

// dataset crop: integer boxes (n,4) [ulx, uly, brx, bry] -> cv2.resize-style bilinear to S x S + ToTensor + Normalize
int launch_crop_resize_normalize(const unsigned char* frame, int H, int W, const int* boxes, int n, int S, float* out,
                                 const LaunchCtx& ctx);

// Pillow-exact bilinear resize + ToTensor + Normalize (CamCalib frame transform)
int pillow_coeffs(int in_size, int out_size, std::vector<int>& bounds, std::vector<int>& kk);
int launch_resize_normalize(const unsigned char* frame, int H, int W, int OH, int OW, const int* hb, const int* hk, int ksh,
                            const int* vb, const int* vk, int ksv, float* out, unsigned char* raw, const LaunchCtx& ctx);

}  // namespace specmi
