/*
 * specmi.h - C ABI of the MI355X-native SPEC inference hot path (libspecmi.so).
 *
 * The reference (mkocabas/SPEC) has NO native/FFI seam for this path: its boundary is two
 * Python nn.Module classes whose leaf ops are stock torch ops,
 *
 *   spec/models/hmr.py:28-122      class HMR            (trunk -> HMRHead -> SMPLCamHead)
 *   camcalib/model.py:24-81        class CameraRegressorNetwork (trunk -> avgpool -> 3 FC)
 *   camcalib/cam_utils.py:110-145  soft-argmax decode of the 256-bin logits
 *   spec/utils/cam_params.py:24-50 (pitch, roll, f) -> cam_rotmat, cam_intrinsics
 *
 * so this header DEFINES the boundary a maintainer would bind (ctypes stub in
 * INTEGRATION.md).  Each entry point names the reference interface it replaces.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes only; no exceptions cross the ABI.
 *   - return 0 (SPECMI_OK) or an error code; the message is kept per handle
 *     (specmi_last_error).  Passing h == NULL to specmi_last_error returns the last error
 *     of a failed specmi_create.
 *   - parameters are handed over as HOST pointers in the canonical PyTorch layouts under
 *     their state_dict key names; the library folds BatchNorm into a
 *     per-channel scale/shift, re-lays the weights out for the MFMA kernels and uploads
 *     them at specmi_commit.
 *   - every forward I/O buffer is a caller-owned DEVICE pointer (torch.Tensor.data_ptr());
 *     the library owns only the packed weights and its workspace.  All work is enqueued on
 *     the caller's HIP stream (`stream` is a hipStream_t passed as void*; NULL = default
 *     stream); no call synchronises the device except specmi_commit, specmi_destroy and
 *     specmi_profile_read.
 *   - every call runs on the device the handle was created for and restores the caller's current HIP device before
 *     it returns (a second device in the same process is fine; kernel attributes are set per device).
 *   - one handle per model instance; a handle is not thread-safe (the reference is
 *     single-threaded) and is driven from ONE stream at a time (its workspaces, split-K slabs and hand-off counters are per
 *     handle: two streams in the same handle at once would share them); distinct handles are independent.
 *   - all arithmetic is IEEE fp32 (reference inference never enables AMP:
 *     spec/config.py:138, spec/tester.py:109-110); index tables are int32.
 */
#ifndef SPECMI_H
#define SPECMI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct specmi_handle specmi_handle;

enum {
    SPECMI_OK = 0,
    SPECMI_ERR_ARG = 1,     /* bad argument (null pointer, bad shape, unknown name)      */
    SPECMI_ERR_HIP = 2,     /* a HIP runtime call failed                                  */
    SPECMI_ERR_STATE = 3,   /* call order violated (forward before commit, ...)           */
    SPECMI_ERR_MISSING = 4  /* commit: a required tensor was never set                    */
};

enum {
    SPECMI_MODEL_CAMCALIB = 0, /* camcalib/model.py CameraRegressorNetwork(resnet50, 1 FC) */
    SPECMI_MODEL_HMR = 1,      /* spec/models/hmr.py HMR(resnet50)                         */
    SPECMI_MODEL_SMPL = 2      /* the SMPL body model alone ("smpl.*" tensors only): specmi_smpl_native /
                                  specmi_smpl_forward for ground-truth meshes in the evaluation code
                                  (spec/utils/compute_error.py:118-160, spec/trainer.py:71-86)             */
};

/* Output dict of HMR.forward (spec/models/hmr.py:113,122): device pointers, fp32. */
typedef struct specmi_hmr_outputs {
    float* smpl_vertices; /* (B, V, 3)      */
    float* smpl_joints3d; /* (B, 49, 3)     */
    float* smpl_joints2d; /* (B, 49, 2)     */
    float* pred_cam_t;    /* (B, 3)         */
    float* pred_pose;     /* (B, 24, 3, 3)  */
    float* pred_cam;      /* (B, 3)         */
    float* pred_shape;    /* (B, 10)        */
    float* pred_pose_6d;  /* (B, 144)       */
} specmi_hmr_outputs;

/* One record of the built-in launch profiler (HIP events around every kernel). */
typedef struct specmi_prof_entry {
    char   kernel[48];  /* kernel family, e.g. "conv_igemm_f32<128x128>"                   */
    char   label[48];   /* call site, e.g. "backbone.layer2.0.conv2"                       */
    double ms;          /* accumulated device time                                         */
    double flops;       /* algorithmic FLOPs of the accumulated launches (2*MACs)          */
    double bytes;       /* algorithmic HBM bytes (activations in+out+residual, weights)    */
    int    launches;
} specmi_prof_entry;

/* ---- lifetime ------------------------------------------------------------------------ */

/* Replaces the module constructors HMR.__init__ (spec/models/hmr.py:29-80) and
 * CameraRegressorNetwork.__init__ (camcalib/model.py:25-57). */
int specmi_create(specmi_handle** out, int device_id, int model_kind);
int specmi_destroy(specmi_handle* h);
const char* specmi_last_error(const specmi_handle* h);
const char* specmi_version(void);

/* ---- parameters (replaces load_state_dict / load_pretrained_model,
 *      spec/tester.py:63-71, scripts/camcalib_demo.py:80-81) ---------------------------- */

/* ---- options -------------------------------------------------------------------------------------------------------------
 * specmi_set_option_i32 / _f32 accept the names below and refuse anything else (SPECMI_ERR_ARG).  The option table of the library
 * (specmi_option_info) is the single list; tests/test_abi.py holds this comment and the code to it.
 *
 * STABLE (frozen in round 6: what a reference maintainer binds)
 *   Model shape, before specmi_commit.  CamCalib (camcalib/model.py:25-70): "backbone" (50 default | 18 | 34 | 101 | 152: the
 *   torchvision ResNet family the reference's eval(backbone) resolves, spec/models/hmr.py:53, camcalib/model.py:33; HMR also 32 / 48 =
 *   HRNet-W32 / W48, with "hrnet_use_conv" 1 = the '-conv' variant, hmr.py:44-51), "num_fc_layers" (1..3), "num_fc_channels"
 *   (<= 1024, multiple of 32).  HMR: "use_cam" (SMPLCamHead vs SMPLHead, hmr.py:66-74), "use_cam_feats" (hmr.py:55,94-98),
 *   "img_res" (224; hmr.py:69), "estimate_var" (0; 1 = HMRHead's uncertainty outputs, hmr.py:35-38,57-64: the library then also wants
 *   "head.decpose_var.*" (144 rows) and "head.decshape_var.*" (10 rows) - the separate-branch layout; a binding splits the doubled
 *   decoders of the other layout - and serves specmi_hmr_uncertainty), "uncertainty_activation" (0 none | 1 relu | 2 softplus |
 *   3 sigmoid | 4 tanh | 5 elu: the torch.nn.functional the reference evaluates by name).  Float option: "focal_length" (5000; hmr.py:31).
 *   Execution, any time (read at every forward):
 *   "plan" - the execution plan of the ResNet trunk:
 *     1 = throughput: the kernels the batch-256 benchmark runs (Winograd F(2x2,3x3) + 64x64 / 128x128 implicit GEMM);
 *     2 = latency: for the reference's own operating point - spec/tester.py:109-151 runs the path at batch = #detections of one
 *         frame, scripts/camcalib_demo.py:95-102 at batch 1 - every convolution with K >= 512 is cut into K slices that run as ONE
 *         launch (the last slice of a tile to arrive folds the partial tiles and applies BN / residual / ReLU), layer3 / layer4
 *         3x3 convolutions leave Winograd for the sliced direct kernel;
 *     3 = single: the latency plan with EVERY 3x3 convolution on the sliced direct kernels (no Winograd) - what batch 1-2 wants;
 *     0 = auto (default): single up to 2 images of 224 x 224 per call, latency up to 10 (trunk pair, FC heads) / 16 (single trunk;
 *         one CamCalib frame at 600 x 1066 counts as 12.7 images), throughput beyond.  specmi_trunk_plan reports the choice.
 *     WITHIN a plan an image's result is bit-identical whatever the batch size, the grouping (specmi_trunk_forward_pair) or the
 *     replay (every k sum has one association fixed by the layer's shape; 8 x 256 rank shards == 2048 unsharded).  BETWEEN plans the
 *     last bits differ (other association of the same products, other algorithm on layer3 / layer4 conv2); all meet the 1e-4
 *     contract on every reference fixture (tests/test_gpu_e2e.py, tests/test_gpu_pretrained_like.py).  Callers that need
 *     bit-reproducibility across batch sizes on both sides of a switch pin a plan (spec_amd.tester runs a whole folder under ONE plan
 *     whatever --frame_batch; bench.py pins 'throughput' for rank-sharded runs).
 *   "winograd" (default 1: 3x3 / stride-1 convolutions with Cin % 16 == 0 and Cout % 64 == 0 run as fused Winograd F(2x2,3x3) on the
 *     fp32 matrix cores under the throughput / latency plans; 0: always the direct implicit GEMM);
 *   "fuse_downsample" (default 1: a bottleneck's downsample conv + BN is folded into its conv3 as one 1x1 GEMM over
 *     [conv2 output | block input]; 0: separate launch + residual add, as torchvision writes it);
 *   "head_collapse" (default 1; HMR, read at commit and at forward: the three IEF iterations of HMRHead - an affine map in eval mode,
 *     no activation between fc1 / fc2 / dec* and dropout = identity - run as ONE (features + camera features) -> 157 GEMM whose
 *     matrix is composed in float64 at commit; 0: the nine GEMMs of the reference loop);
 *   "output_ld" (HMR, default 0 = dense outputs; n > 0: every per-image output pointer of specmi_hmr_forward /
 *     specmi_hmr_head_forward / specmi_smpl_forward addresses image 0 and image b lives at pointer + b*n floats, i.e. the outputs are
 *     columns of ONE caller-owned (B, n) record - the packed all-gather record of SURVEY.md 8e - written by the kernels directly);
 *   "angle_ld" (the same for vfov / pitch / roll of specmi_camcalib_decode / specmi_camcalib_head_decode);
 *   "experimental" (default 0; 1 = this handle accepts the names of the next list).
 *
 * EXPERIMENTAL (refused with SPECMI_ERR_STATE unless the environment has SPECMI_EXPERIMENTAL=1 or the handle's "experimental" is 1;
 * setting one to its default is a no-op and always accepted).  Tuning thresholds from single-box sweeps, debug pins, opt-ins that
 * measured slower or neutral on MI355X, and the narrower-arithmetic secondary mode: scaffolding of the experiments recorded in
 * docs/ROUNDS.md and profiles/, same bits as the defaults unless stated, no compatibility promise.
 *   secondary arithmetic: "conv_precision" (0; 3 / 6 = the 1x1 (6: and strided 3x3) convolutions as that many bf16 piece products on
 *     the bf16 matrix cores: NOT the reference's fp32 arithmetic per product, tests/test_gpu_bf16split.py), "conv_precision_3x3" (0);
 *   debug pins: "force_conv_variant" / "force_wino_variant" (0 = auto), "latency_force_unit" (0 = by batch, 1 / 2 / 3 = a leaf / a
 *     group / the whole K per workgroup), "conv2d_sk" (specmi_conv2d only: 0 = throughput kernel, -1 = the latency plan's rule,
 *     n > 1 = n leaves), "conv2d_wsplit" (specmi_conv2d only: 0 | 2 | 3), "smpl_skin_split" (-1 = by batch, 0 / 1 = never / always),
 *     "trunk_subbatch" (0) / "trunk_subbatch_layers" (2), "fc_splitk" (1), "fc_gemv" (1), "head_fuse" (3; bit 0: the regressor's state
 *     init rides in the pooling launch, bit 1: head_final's work is done by the SMPL pose kernel);
 *   plan thresholds: "single_max_batch" (2), "latency_max_batch" (10), "latency_max_batch_single" (16), "latency_target_wgs" (256),
 *     "latency_min_chunks" (4), "latency_wino_min_tiles" (128), "latency_fill_wgs" (240) / "latency_fill_wgs_large" (400),
 *     "latency_unit_model" (0; 1 = pick the unit by a round model with "latency_unit_slots" 256: measured equal or worse);
 *   wave-split unit of the latency / single plans (spec_amd/csrc/conv_wsplit.hip): "wsplit" (1; 0 = never, 2 / 3 = always with one
 *     group / all groups per workgroup), "wsplit_max_units" (1400, trunk pair), "wsplit_max_units_single" (500), "wsplit_slots" (256);
 *   "persist" (0, opt-in): every run of implicit-GEMM layers of the latency / single plan as ONE persistent launch
 *     (spec_amd/csrc/conv_persist.hip).  Bit-identical to the per-layer launches and measured SLOWER on MI355X (0.68 vs 0.55 ms,
 *     profiles/r05_a_persist_ab.jsonl); at most two such forwards may be in flight per device.  "persist_wgs" (0 = 512 pair / 256
 *     single), "persist_fill_wgs" (0 = the latency fill), "persist_l2_prefetch" (0), "persist_spin_limit" (400000), "persist_min_run"
 *     (2), "persist_max_run" (64; 1 = every layer its own walker launch: resident workgroups walk the layer's tiles, no in-launch
 *     waits - the round-6 tile-walking ablation, profiles/r06_*_walk_ablation.*), "persist_allow_full" (0); specmi_sync_status
 *     reports a spin that gave up;
 *   "tail_fuse" (0, opt-in): at small batches each network's tail as ONE launch (spec_amd/csrc/head.hip) - same bits, four graph
 *     nodes less per step and NOT faster on MI355X (profiles/r05_f_tail_check.jsonl). */
int specmi_set_option_i32(specmi_handle* h, const char* name, int value);
int specmi_set_option_f32(specmi_handle* h, const char* name, float value);
/* The effective value of an integer option on this handle: what was set, else the default of the table. */
int specmi_get_option_i32(specmi_handle* h, const char* name, int* value);
/* Entry `index` (0 .. n-1; SPECMI_ERR_ARG past the end) of the option table: name, default and whether it is on the STABLE list.
 * Needs no handle and no GPU. */
int specmi_option_info(int index, const char** name, int* default_value, int* is_stable);

/* Host tensors under state_dict key names, e.g. "backbone.layer1.0.conv1.weight" (OIHW),
 * "backbone.bn1.running_var", "fc_vfov.weight" (out,in), "head.fc1.weight",
 * "head.init_pose"; SMPL body model under "smpl.v_template" (V,3), "smpl.shapedirs"
 * (V,3,10), "smpl.posedirs" (207,3V), "smpl.J_regressor" (24,V), "smpl.lbs_weights" (V,24),
 * "smpl.J_regressor_extra" (9,V); int32: "smpl.parents" (24), "smpl.extra_vertex_ids" (21),
 * "smpl.joint_map" (49).  Data is copied; the caller may free it on return. */
int specmi_set_tensor_f32(specmi_handle* h, const char* name, const float* host_data,
                          const int64_t* shape, int ndim);
int specmi_set_tensor_i32(specmi_handle* h, const char* name, const int32_t* host_data,
                          const int64_t* shape, int ndim);
/* Validate, fold BN, pack and upload to HBM.  May be called again after further set_* calls. */
int specmi_commit(specmi_handle* h);

/* ---- forward: CamCalib ----------------------------------------------------------------- */

/* (The execution plan - option "plan" above - is chosen from the FIRST handle's options and B, H, W for both trunks.)
 * The ResNet trunks of TWO committed models (CamCalib + SPEC: camcalib/model.py:73 `self.backbone(images)` and
 * spec/models/hmr.py:92 `self.backbone(images)`, which the reference runs as two processes / two calls) walked in lockstep
 * with every layer of both as ONE grouped launch: images_a / images_b (B,3,H,W) NCHW -> feat_a / feat_b (B, H/32, W/32, C)
 * NHWC.  Both trunks must be ResNets of the same depth and see the same B, H, W; each keeps its own weights and workspaces.
 * Results are bit-identical to two specmi_trunk_forward calls. */
int specmi_trunk_forward_pair(specmi_handle* ha, specmi_handle* hb, const float* images_a, const float* images_b, int B,
                              int H, int W, float* feat_a, float* feat_b, void* stream);

/* The part of CameraRegressorNetwork.forward after the backbone (camcalib/model.py:74-80: avg-pool, flatten, the three
 * Linear chains) from a trunk feature map (B,fh,fw,C) NHWC -> vfov / pitch / roll logits (B,nbins) each. */
int specmi_camcalib_head_forward(specmi_handle* h, const float* feat_nhwc, int B, int fh, int fw, float* logits_vfov,
                                 float* logits_pitch, float* logits_roll, void* stream);

/* specmi_camcalib_head_forward + specmi_camcalib_decode in one call (round 5); with option "tail_fuse" = 1 and at small batches
 * (the GEMV path of the latency / single plans, one Linear layer per head) the avg-pool, the three heads and the decode run as ONE
 * launch (spec_amd/csrc/head.hip: tail_gemv_kernel - the code of the three kernels, same bits); otherwise the separate kernels.
 * Replaces camcalib/model.py:74-80 + camcalib/cam_utils.py:110-133 + scripts/camcalib_demo.py:129 + spec/utils/cam_params.py:37-46.
 * Outputs as in the two calls (any of vfov .. K may be NULL); "angle_ld" applies to vfov / pitch / roll. */
int specmi_camcalib_head_decode(specmi_handle* h, const float* feat_nhwc, int B, int fh, int fw, float* logits_vfov,
                                float* logits_pitch, float* logits_roll, const float* img_h, const float* img_w, float* vfov,
                                float* pitch, float* roll, float* f_pix, float* cam_rotmat, float* cam_intrinsics, void* stream);

/* CameraRegressorNetwork.forward (camcalib/model.py:72-81): images (B,3,H,W) NCHW fp32 ->
 * three (B,256) logit tensors [vfov, pitch, roll]. */
int specmi_camcalib_forward(specmi_handle* h, const float* images_nchw, int B, int H, int W,
                            float* logits_vfov, float* logits_pitch, float* logits_roll,
                            void* stream);

/* convert_preds_to_angles soft-argmax branch (camcalib/cam_utils.py:114-133), focal length
 * (scripts/camcalib_demo.py:129) and the CamCalib->SPEC hand-off read_cam_params
 * (spec/utils/cam_params.py:37-48), fused in one launch.  Any output pointer may be NULL.
 * img_h/img_w: (B,) full-image size in pixels.  R (B,3,3) = Rx(pitch) Ry(0) Rz(roll);
 * K (B,3,3) = [[f,0,w/2],[0,f,h/2],[0,0,0]] (K[2,2] stays 0 as in the reference). */
int specmi_camcalib_decode(specmi_handle* h, const float* logits_vfov, const float* logits_pitch,
                           const float* logits_roll, int B, int nbins, const float* img_h,
                           const float* img_w, float* vfov, float* pitch, float* roll,
                           float* f_pix, float* cam_rotmat, float* cam_intrinsics, void* stream);

/* The two per-row reductions of camcalib/cam_utils.py on a (rows, nbins) device logit tensor:
 *   argmax_idx[row] = np.argmax(row) - bins2vfov / bins2pitch / bins2roll / bins2horizon (cam_utils.py:66-91), the 'kl' / 'ce'
 *                     and legacy branches of convert_preds_to_angles (:123-133); first maximum, NaN counts as maximum;
 *                     the caller gathers the bin-centre table (host float64, as the reference returns it);
 *   soft_idx[row]   = get_softargmax (cam_utils.py:110-118): softmax expectation of the index, normalised to [-1, 1].
 * Either output may be NULL (not both). */
int specmi_camcalib_bins(specmi_handle* h, const float* logits, int rows, int nbins, int32_t* argmax_idx,
                         float* soft_idx, void* stream);

/* read_cam_params (spec/utils/cam_params.py:24-50) for angles that were decoded earlier, e.g.
 * read back from the CamCalib result pickle: (pitch, roll, f_pix, img_w, img_h) (B,) device ->
 * cam_rotmat (B,3,3), cam_intrinsics (B,3,3) (K[2,2] = 0).  Either output may be NULL. */
int specmi_cam_params(specmi_handle* h, const float* pitch, const float* roll, const float* f_pix,
                      const float* img_w, const float* img_h, int B, float* cam_rotmat,
                      float* cam_intrinsics, void* stream);

/* ---- forward: SPEC --------------------------------------------------------------------- */

/* HMR.forward (spec/models/hmr.py:82-122).  cam_* / bbox_* / img_* may be NULL when the
 * handle was built with use_cam = 0 (and use_cam_feats = 0). */
int specmi_hmr_forward(specmi_handle* h, const float* images_nchw, int B, int H, int W,
                       const float* cam_rotmat, const float* cam_intrinsics,
                       const float* bbox_scale, const float* bbox_center, const float* img_w,
                       const float* img_h, const specmi_hmr_outputs* out, void* stream);

/* The two extra outputs of HMR.forward when the model was built with estimate_var = True (spec/models/hmr.py:35-38,57-64; consumed
 * by spec/losses.py:61-62): pred_pose_var (B, 288) = [pred_pose_6d | var_pose], pred_shape_var (B, 20) = [pred_shape | var_shape],
 * the variances being the extra decoder outputs of the LAST regressor iteration through option "uncertainty_activation".  Call after
 * specmi_hmr_forward / specmi_hmr_regress / specmi_hmr_head_forward of the same batch on the same stream (it reads what that call
 * left in the handle's workspace); dense outputs.  SPECMI_ERR_STATE without option "estimate_var". */
int specmi_hmr_uncertainty(specmi_handle* h, int B, float* pred_pose_var, float* pred_shape_var, void* stream);

/* Everything of HMR.forward after `features = self.backbone(images)` (spec/models/hmr.py:94-122): regressor head +
 * SMPL head from an NHWC layer-4 map (B, fh, fw, C).  Lets a caller run the SPEC trunk beside the CamCalib network
 * (whose output the head needs) on another stream and join afterwards. */
int specmi_hmr_regress(specmi_handle* h, const float* feat_nhwc, int B, int fh, int fw,
                       const float* cam_rotmat, const float* cam_intrinsics, const float* bbox_scale,
                       const float* bbox_center, const float* img_w, const float* img_h,
                       const specmi_hmr_outputs* out, void* stream);

/* ---- stage-level entry points (parity tests, profiling, reuse) ------------------------- */

/* The trunk alone: `self.backbone(images)` (hmr.py:92, camcalib/model.py:73).  Output is the
 * layer4 map in NHWC: (B, H/32, W/32, 2048). */
int specmi_trunk_forward(specmi_handle* h, const float* images_nchw, int B, int H, int W,
                         float* feat_nhwc, void* stream);

/* HMRHead.forward on an NHWC feature map (hmr.py:96/98): avg-pool + 3 IEF iterations +
 * rot6d->rotmat.  Outputs any-NULL. */
int specmi_hmr_head_forward(specmi_handle* h, const float* feat_nhwc, int B, int fh, int fw,
                            const float* cam_rotmat, const float* cam_intrinsics,
                            const float* img_h, float* pred_pose, float* pred_shape,
                            float* pred_cam, float* pred_pose_6d, void* stream);

/* SMPLCamHead / SMPLHead forward (hmr.py:101-120): SMPL LBS (pose2rot=False) + 49 joints +
 * camera translation + perspective projection. */
int specmi_smpl_forward(specmi_handle* h, const float* rotmat, const float* betas,
                        const float* cam, int B, const float* cam_rotmat,
                        const float* cam_intrinsics, const float* bbox_scale,
                        const float* bbox_center, const float* img_w, const float* img_h,
                        float* vertices, float* joints3d, float* joints2d, float* cam_t,
                        void* stream);

/* The body model WITHOUT the 49-joint wrapper - smplx.SMPL.forward as the evaluation code calls it (`smpl_native` /
 * `body_model` / `body_model_orig`, spec/trainer.py:71-86,249-254, spec/utils/compute_error.py:118-160): `pose` is
 * (B,24,3,3) rotation matrices (pose2rot=False) or, with pose_is_axis_angle != 0, (B,72) axis-angle vectors
 * (global_orient | body_pose; smplx batch_rodrigues); outputs vertices (B,V,3) and / or joints24 (B,24,3) =
 * `.joints[:, :24]`, the posed kinematic-chain joints.  Either output may be NULL (not both). */
int specmi_smpl_native(specmi_handle* h, const float* pose, int pose_is_axis_angle, const float* betas, int B,
                       float* vertices, float* joints24, void* stream);

/* A single fused conv+BN(+residual)(+ReLU) layer, y = act(conv(x)*scale + shift [+ res]).
 * x (B,H,W,Cin) NHWC device; w (Cout,Cin,KH,KW) OIHW HOST; scale/shift (Cout) HOST;
 * residual/out (B,OH,OW,Cout) NHWC device.  Cin%32==0 unless (Cin==3,KH==7: stem path,
 * x is then NCHW).  Device pointers 16-byte aligned (the kernels move 16 bytes per lane; a misaligned x is refused by the
 * Winograd path with an error, never read wrongly).  Used by the per-layer parity tests. */
int specmi_conv2d(specmi_handle* h, const float* x, int B, int H, int W, int Cin,
                  const float* w_oihw_host, const float* scale_host, const float* shift_host,
                  int Cout, int KH, int KW, int stride, int pad, const float* residual,
                  int relu, float* out, void* stream);

/* MaxPool2d(3,2,1) and global average pool on NHWC device tensors (trunk building blocks). */
int specmi_maxpool3x3s2(specmi_handle* h, const float* x, int B, int H, int W, int C,
                        float* out, void* stream);
int specmi_avgpool(specmi_handle* h, const float* x, int B, int HW, int C, float* out,
                   void* stream);

/* ---- crop + normalise in front of the path ------------------------------------ */

/* The detection loop of spec/tester.py:116-128: for each bbox (cx, cy, w, h) [device, (n,4)]
 * get_single_image_crop_demo(frame, bbox, scale, crop_size) - 3-point affine (rot 0) +
 * cv2.warpAffine(INTER_LINEAR, BORDER_CONSTANT) fixed-point bilinear + ToTensor + ImageNet
 * Normalize (spec/constants.py:20-21) - from a uint8 RGB HWC frame in device memory to
 * (n,3,S,S) fp32 NCHW (frames of 4 GiB or more, or with a side of 2^24 pixels, are refused: 32-bit offsets).
 * Optional outputs: raw (n,S,S,3) uint8 crop, bbox_scale = w/200 (n),
 * bbox_center (n,2). */
int specmi_crop_normalize(specmi_handle* h, const uint8_t* frame_rgb_hwc, int H, int W,
                          const float* bboxes, int n, float scale, int crop_size, float* out_nchw,
                          uint8_t* raw_hwc, float* bbox_scale, float* bbox_center, void* stream);

/* The same crops for the detections of MANY frames in one launch - the loop over images of spec/tester.py:109-128 (one
 * frame, its detections, one crop each) flattened: `frames` is a slab of nframes equal-sized uint8 RGB HWC frames in device
 * memory (frame f at frames + f*H*W*3), crop d is cut from frame frame_index[d] (device, (n) int32, values in [0, nframes);
 * the index lives in caller memory the library cannot inspect without a synchronisation: an out-of-range value is CLAMPED into
 * the slab - a wrong crop, never an out-of-bounds read; validate on the host where the index is produced) with bbox d.  Same arithmetic, same outputs per crop as specmi_crop_normalize (bit-identical); what it removes is one
 * launch, one host synchronisation and one small batch per frame. */
int specmi_crop_normalize_batch(specmi_handle* h, const uint8_t* frames_rgb_hwc, int nframes, int H, int W,
                                const int32_t* frame_index, const float* bboxes, int n, float scale, int crop_size,
                                float* out_nchw, uint8_t* raw_hwc, float* bbox_scale, float* bbox_center, void* stream);

/* The evaluation dataset's image path (spec/dataset/cam_dataset.py:253-287 rgb_processing with flip 0 / rot 0 / pn 1, :367-377):
 * pare `crop(img, center, scale, [res, res])` = copy of the integer box [ul, br) (zero outside the frame) scaled to res x res
 * with cv2.resize (bilinear, half-pixel centres, replicated border), clip to [0, 255], float32 / 255, ImageNet Normalize.
 * boxes: (n,4) int32 device [ul_x, ul_y, br_x, br_y] as the reference's `transform(..., invert=1)` yields them (computed on
 * the host: spec_amd.preprocess.pare_crop_boxes); frame uint8 RGB HWC device; out (n,3,S,S) fp32 NCHW. */
int specmi_crop_resize_normalize(specmi_handle* h, const uint8_t* frame_rgb_hwc, int H, int W, const int32_t* boxes, int n,
                                 int crop_size, float* out_nchw, void* stream);

/* The CamCalib frame transform (camcalib/pano_dataset.py:156-162, scripts/camcalib_demo.py:100): torchvision
 * Resize(600) on a PIL image = Pillow's antialiased bilinear resample (Image.resize((OW, OH), BILINEAR): separable
 * triangle filter, 22-bit fixed-point coefficients, uint8 between the passes) + ToTensor + ImageNet Normalize, from a
 * uint8 RGB HWC frame in device memory to (3,OH,OW) fp32; bit-identical to Pillow.  The caller picks (OH, OW)
 * (shorter side 600, longer int(600*long/short)).  Optional raw_hwc: the resized uint8 image (OH,OW,3). */
int specmi_resize_normalize(specmi_handle* h, const uint8_t* frame_rgb_hwc, int H, int W, int OH, int OW,
                            float* out_chw, uint8_t* raw_hwc, void* stream);

/* ---- evaluation metrics on the path's outputs ---------------------------------- */

/* eval_single (spec/utils/compute_error.py:52-86, spec/trainer.py:272-316): joints =
 * J_regressor (J,V) @ vertices for prediction and ground truth, pelvis (joint 0) alignment,
 * selection of `nsel` joints (`joint_sel` device int32, NULL = the first nsel), then per image
 * MPJPE, PA-MPJPE (similarity Procrustes) and pelvis-aligned V2V, all in millimetres.  All
 * pointers are device pointers; any output may be NULL.  J, nsel <= 32. */
int specmi_eval_mesh(specmi_handle* h, const float* pred_vertices, const float* gt_vertices, int B,
                     int V, const float* J_regressor, int J, const int32_t* joint_sel, int nsel,
                     float* mpjpe_mm, float* pampjpe_mm, float* v2v_mm, void* stream);

/* eval_j_24 (spec/utils/compute_error.py:33-49): pelvis-aligned MPJPE / PA-MPJPE (mm) of two
 * (B,J,3) joint sets. */
int specmi_eval_joints(specmi_handle* h, const float* pred_joints, const float* gt_joints, int B,
                       int J, float* mpjpe_mm, float* pampjpe_mm, void* stream);

/* pred_joints = einsum('bik,ji->bjk', vertices, J_regressor) (spec/utils/compute_error.py:184,187): vertices (B,V,3),
 * J_regressor (J,V) device -> joints (B,J,3). */
int specmi_regress_joints(specmi_handle* h, const float* vertices, int B, int V, const float* J_regressor, int J,
                          float* joints, void* stream);

/* torch.bmm(R, x.transpose(2,1)).transpose(2,1) (spec/utils/compute_error.py:164-165,186): R (B,3,3), points (B,N,3)
 * -> out (B,N,3), out[b,n] = R[b] points[b,n]. */
int specmi_rotate_points(specmi_handle* h, const float* R, const float* points, int B, int N, float* out, void* stream);

/* Which execution plan a trunk forward of (B, 3, H, W) takes under the handle's current options ("plan" and its thresholds):
 * *mode = 0 throughput, 1 latency, 2 single; pair != 0: as specmi_trunk_forward_pair decides (the FIRST handle's options).  Callers
 * that describe or log what ran ask here instead of re-deriving the rule (bench.py, SpecPipeline). */
int specmi_trunk_plan(specmi_handle* h, int B, int H, int W, int pair, int32_t* mode);

/* ---- in-launch hand-off state (round 5) ---------------------------------------------------
 * The latency / single plans hand data between workgroups INSIDE a launch (split-K tile tickets, the completion counters of the
 * persistent multi-layer walker: spec_amd/csrc/conv_persist.hip).  Those protocols keep a few device counters that every launch
 * leaves at zero.  They have no counterpart in the reference (spec/tester.py:109-151 runs stock torch ops). */

/* Synchronises the device.  *persist_err: 0 = the handle's persistent launches all completed their hand-offs; 1 = a bounded spin
 * gave up (the results of that launch are garbage: a protocol error or a grid that was not co-resident - more than two persistent
 * forwards in flight on one device); 3 = the bounded wait inside a fused tail (option "tail_fuse") gave up; < 0 = a control block
 * was not left clean (-1 / -2 the walker's, -3 the fused tails'). */
int specmi_sync_status(specmi_handle* h, int32_t* persist_err);
/* Zeroes the hand-off counters on `stream`.  The library does this itself at specmi_commit and after any forward that returned
 * an error; a caller that destroyed a captured graph mid-replay (or killed a launch some other way) calls it before the next forward. */
int specmi_sync_reset(specmi_handle* h, void* stream);
/* Tests only: overwrites every hand-off counter with `value` (synchronises). */
int specmi_debug_poison_sync(specmi_handle* h, uint32_t value);

/* ---- profiling -------------------------------------------------------------------------- */

/* When on, every kernel launch is bracketed by HIP events on the launch stream. */
int specmi_profile_enable(specmi_handle* h, int on);
/* Synchronises, folds the recorded launches into entries keyed by (kernel,label) and
 * clears the log.  Returns the number of entries in *n (at most max_entries copied). */
int specmi_profile_read(specmi_handle* h, specmi_prof_entry* entries, int max_entries, int* n);

#ifdef __cplusplus
}
#endif
#endif /* SPECMI_H */
