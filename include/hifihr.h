/*
 * libhifihr.so -- C ABI of the MI355X-native HiFiHR training hot path.
 *
 * The reference (viridityzhu/HiFiHR) has no FFI layer: its hot path is reached through plain Python
 * call sites.  Every entry point below replaces one of those call sites and
 * cites it.  Conventions:
 *   - every function returns 0 on success, a negative HIFIHR_E* code on failure;
 *     hifihr_last_error() returns a thread-local message.  No exceptions cross the ABI.
 *   - pointers named *_d are DEVICE pointers (HBM), *_h are HOST pointers; fp32 unless stated.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  No hidden syncs, no
 *     allocation inside compute calls: the caller owns every buffer; handles own only their tables.
 *   - handles are immutable after create; compute calls are re-entrant on different streams.
 */
#ifndef HIFIHR_H
#define HIFIHR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HIFIHR_OK 0
#define HIFIHR_EINVAL (-1)   /* bad argument */
#define HIFIHR_EHIP (-2)     /* HIP runtime error (message has hipGetErrorString) */
#define HIFIHR_ENOMEM (-3)

#define HIFIHR_MANO_NV 778
#define HIFIHR_MANO_NF 1538
#define HIFIHR_MANO_NJ 16

int hifihr_version(void);
const char* hifihr_last_error(void);
/* number of HIP devices visible to the library (<=0: the library cannot be used for compute) */
int hifihr_device_count(void);

/* ------------------------------------------------------------------------------------------------
 * MANO linear-blend skinning.
 * Replaces ManoLayer.__init__/forward   reference utils/my_mano.py:231-313, 315-483
 *          (as configured by MyMANOLayer, utils/my_mano.py:35-36: center_idx=9, flat_hand_mean=False,
 *           side='right', use_pca=True, ncomps=48 => 45 effective PCA coefficients, axis-angle root)
 *          batch_rodrigues/quat2mat      reference utils/manopth/rodrigues_layer.py:43-54,15-40
 *          xyz_from_vertice              reference utils/Freihand_GNN_mano/Freihand_trainer_mano_fullsup.py:175-215
 * ---------------------------------------------------------------------------------------------- */
typedef struct hifihr_mano hifihr_mano_t;

/* Tables are host arrays in the layout ManoLayer registers them (my_mano.py:283-313):
 * v_template[778][3], shapedirs[778][3][10], posedirs[778][3][135], j_regressor[16][778] (dense),
 * weights[778][16], hands_components[45][45] (row k = component k), hands_mean[45]. */
int hifihr_mano_create(hifihr_mano_t** out, const float* v_template_h, const float* shapedirs_h,
                       const float* posedirs_h, const float* j_regressor_h, const float* weights_h,
                       const float* hands_components_h, const float* hands_mean_h);
int hifihr_mano_destroy(hifihr_mano_t* h);

/* ManoLayer.forward(th_pose_coeffs=pose[B][48], th_betas=beta[B][10]) -> th_verts[B][778][3],
 * th_jtr[B][21][3] (both centred on joint 9; th_trans==0 branch, my_mano.py:471-475).
 * saved_vposed_d[B][778][3] receives v_posed (my_mano.py:392) for the backward pass; may be NULL when
 * no backward follows. */
int hifihr_mano_lbs_fwd(const hifihr_mano_t* h, const float* pose_d, const float* beta_d, int B,
                        float* verts_d, float* jtr_d, float* saved_vposed_d, void* stream);

/* Gradient of the above: gverts[B][778][3], gjtr[B][21][3] (either may be NULL = zero) ->
 * gpose[B][48], gbeta[B][10] (overwritten).  Deterministic (no float atomics). */
int hifihr_mano_lbs_bwd(const hifihr_mano_t* h, const float* pose_d, const float* beta_d,
                        const float* saved_vposed_d, const float* gverts_d, const float* gjtr_d, int B,
                        float* gpose_d, float* gbeta_d, void* stream);

/* xyz_from_vertice(verts).permute(1,0,2) followed by the root-relative step of Model.forward
 * (reference models_res_nimble.py:153,160-166, training branch):
 *   joints21 = regress(verts); root = joints21[:, root_id]; joints_rel = joints21 - root;
 *   verts_rel = verts - root.   root_id < 0 skips the subtraction (root_d then receives zeros).
 * Outputs joints_rel_d[B][21][3], verts_rel_d[B][778][3] (may alias verts_d), root_d[B][3]. */
int hifihr_mano_joints_fwd(const hifihr_mano_t* h, const float* verts_d, int B, int root_id,
                           float* joints_rel_d, float* verts_rel_d, float* root_d, void* stream);
/* Gradient: gjoints_rel[B][21][3], gverts_rel[B][778][3], groot[B][3] (any may be NULL) ->
 * gverts[B][778][3] (overwritten). */
int hifihr_mano_joints_bwd(const hifihr_mano_t* h, const float* gjoints_rel_d, const float* gverts_rel_d,
                           const float* groot_d, int B, int root_id, float* gverts_d, void* stream);

/* The two steps above as ONE call per direction (round 5), with the mesh offset `skin_meshes.offset_verts_(-pred_root);
 * .offset_verts_(root_xyz)` that precedes the renderer folded in
 * (reference utils/my_mano.py:315-483; Freihand_trainer_mano_fullsup.py:175-215; models_res_nimble.py:153,160-166,203-205):
 *   verts = layer(pose, beta); joints21 = regress(verts); root = joints21[:, root_id];
 *   joints_rel = joints21 - root; verts_rel = verts - root; verts_cam = verts_rel + root_xyz.
 * Forward: two launches (the layer; regression + offsets); backward: ONE launch (the regression's backward is the head of the layer's).
 * verts_d[B][778][3] receives the layer's (absolute, centred) vertices.  root_xyz_d[B][3] / verts_cam_d / root_d may be NULL.
 * saved_vposed_d as in hifihr_mano_lbs_fwd.  Results are bit-identical to the two-call form. */
int hifihr_mano_full_fwd(const hifihr_mano_t* h, const float* pose_d, const float* beta_d, int B, int root_id,
                         const float* root_xyz_d, float* verts_d, float* joints_rel_d, float* verts_rel_d, float* verts_cam_d,
                         float* root_d, float* saved_vposed_d, void* stream);
/* Gradient: gjoints_rel[B][21][3], gverts_rel[B][778][3], gverts_cam[B][778][3], groot[B][3] (any may be NULL = zero) ->
 * gpose[B][48], gbeta[B][10] (overwritten).  gpose_add_d[B][48] / gbeta_add_d[B][10] (may be NULL) are ADDED to the results: the
 * gradient that reaches pose / beta through their other consumer (the mpose / mshape terms of losses.py:283-284), so that the sum
 * costs no launch of its own.  Deterministic (no float atomics). */
int hifihr_mano_full_bwd(const hifihr_mano_t* h, const float* pose_d, const float* beta_d, const float* saved_vposed_d,
                         const float* gjoints_rel_d, const float* gverts_rel_d, const float* gverts_cam_d,
                         const float* groot_d, const float* gpose_add_d, const float* gbeta_add_d, int B, int root_id,
                         float* gpose_d, float* gbeta_d, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Generic linear-blend skinning (any mesh size / kinematic tree): the NIMBLE-shaped hand layer.
 * Replaces the skinning step of  self.hand_layer(hand_params, handle_collision=False)  when `hand_model: "nimble"`
 *          (reference models_res_nimble.py:56-57,133-142; MyNIMBLELayer itself is an un-vendored submodule -- SURVEY.md section 8 A9 --
 *          so the formulation is the MANO one of utils/my_mano.py:386-451 without pose-corrective blend shapes:
 *          v_shaped = v_template + shapedirs . beta;  J = j_regressor . v_shaped;  R = Rodrigues(theta);
 *          global transforms down `parents`;  v = sum_j weights[v][j] (G_j [v_shaped; 1] - G_j [J_j; 0])).
 * ---------------------------------------------------------------------------------------------- */
typedef struct hifihr_lbs hifihr_lbs_t;

/* Host tables: v_template[V][3], shapedirs[V][3][S], j_regressor[J][V] (dense), weights[V][J] (dense; at most 8 non-zeros per
 * row, kept exactly), parents[J] (parents[0] = -1, parents[j] < j).  1 <= J <= 32, 0 <= S <= 32. */
int hifihr_lbs_create(hifihr_lbs_t** out, int V, int J, int S, const float* v_template_h, const float* shapedirs_h,
                      const float* j_regressor_h, const float* weights_h, const int* parents_h);
int hifihr_lbs_destroy(hifihr_lbs_t* h);

/* theta[B][J][3] (axis-angle per joint, joint 0 = global rotation), beta[B][S] -> verts[B][V][3], joints[B][J][3] (posed joint
 * positions; may be NULL).  No centring. */
int hifihr_lbs_fwd(const hifihr_lbs_t* h, const float* theta_d, const float* beta_d, int B, float* verts_d, float* joints_d, void* stream);

/* Gradient: gverts[B][V][3], gjoints[B][J][3] (may be NULL = zero) -> gtheta[B][J][3] (overwritten), gbeta[B][S] (ACCUMULATED: the
 * caller passes it zeroed).  scratch[B][J][12] must be zero on entry; it is left holding d(loss)/d(A_j).  Sums over vertices use float
 * atomics: results vary in the last bits from run to run. */
int hifihr_lbs_bwd(const hifihr_lbs_t* h, const float* theta_d, const float* beta_d, const float* gverts_d, const float* gjoints_d,
                   int B, float* scratch_zeroed_d, float* gtheta_d, float* gbeta_zeroed_d, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Hard rasteriser + Phong shader + anti-aliasing resolve.
 * Replaces  rendered = self.renderer_p3d(skin_meshes, cameras=cameras, lights=lighting)
 *           rendered = F.avg_pool2d(rendered.permute(0,3,1,2), aa, aa)      reference models_res_nimble.py:208-211
 * with the renderer built as in models_res_nimble.py:70-96 (MeshRasterizer(image_size*aa, blur 0, 1 face per
 * pixel) + HardPhongShader) and cameras/lights as in :184-190.  PyTorch3D semantics are restated in
 * oracle/raster_oracle.c and oracle/render_oracle.py (parity unpinned at that third-party boundary).
 * Meshes share one topology (faces) across the batch, as MyMANOLayer/MyNIMBLELayer produce them.
 * ---------------------------------------------------------------------------------------------- */
typedef struct hifihr_renderer hifihr_renderer_t;

/* faces_h[F][3] int32 (host); image_size = output edge H (224), aa = samples per pixel edge (3; 1..3 supported);
 * ambient3 = materials.ambient * lights.ambient, mat_diffuse3 = materials.diffuse,
 * specular3 = materials.specular * lights.specular, background3 = BlendParams.background_color. */
int hifihr_renderer_create(hifihr_renderer_t** out, const int32_t* faces_h, int V, int F, int image_size, int aa,
                           const float* ambient3, const float* mat_diffuse3, const float* specular3, float shininess,
                           const float* background3);
/* point_lights = 1: PointLights instead of DirectionalLights (the reference's light_estimation = false branch,
 * models_res_nimble.py:191-198): light_dir_d[B][3] of render_fwd / bwd is then the light's LOCATION, the direction of a sample is
 * location - point (normalised); the location gets no gradient (glight_dir_d comes back zero). */
int hifihr_renderer_set_light_mode(hifihr_renderer_t* h, int point_lights);
int hifihr_renderer_destroy(hifihr_renderer_t* h);
/* bytes of scratch the caller must pass to render_fwd/bwd for batch B; the SAME buffer, untouched in between,
 * must be passed to the backward call of a forward call (it carries the packed per-vertex records). */
size_t hifihr_render_workspace_bytes(const hifihr_renderer_t* h, int B);

/* verts_d[B][V][3] view-space vertices (R = I, T = 0); vcolors_d per-vertex RGB ([B][V][3] when vcolors_batched,
 * else [V][3] shared); cam_d[B][4] = (fx, fy, px, py) exactly as handed to PerspectiveCameras (i.e. the
 * NEGATED ndc focal lengths, models_res_nimble.py:184-186); light_color_d[B][3] = DirectionalLights.diffuse_color,
 * light_dir_d[B][3] = DirectionalLights.direction (un-normalised).
 * Outputs: rgba_d[B][4][H][H] (NCHW, after the aa x aa average pool; channel 3 = coverage),
 *          face_id_d[B][H*aa][H*aa] int32 = pix_to_face minus the b*F packing offset (-1 = background). */
int hifihr_render_fwd(const hifihr_renderer_t* h, const float* verts_d, const float* vcolors_d, int vcolors_batched,
                      const float* cam_d, const float* light_color_d, const float* light_dir_d, int B, float* rgba_d,
                      int32_t* face_id_d, void* workspace_d, void* stream);

/* grad_rgba_d[B][4][H][H] (the coverage channel carries no gradient, as in the reference: SURVEY.md F7) ->
 * gverts_d[B][V][3], gvcolors_d[B][V][3] (may be NULL), glight_color_d[B][3], glight_dir_d[B][3]; all overwritten.
 * Uses float atomics: results are reproducible to rounding, not bitwise. */
int hifihr_render_bwd(const hifihr_renderer_t* h, const float* verts_d, const float* cam_d, const float* light_color_d,
                      const float* light_dir_d, const int32_t* face_id_d, const float* grad_rgba_d, int B,
                      float* gverts_d, float* gvcolors_d, float* glight_color_d, float* glight_dir_d, void* workspace_d,
                      void* stream);
/* TexturesUV mode (PyTorch3D renderer/mesh/textures.py TexturesUV.sample_textures [recalled]; the reference hands NIMBLE's texture image to
 * the renderer this way, models_res_nimble.py:203-208): per sample uv = sum_k bary_k * verts_uvs[faces_uvs[f][k]] with the rasteriser's
 * perspective-corrected barycentrics, texel = grid_sample(flip(maps, vertical), 2 uv - 1, bilinear, align_corners = True, border padding),
 * shaded like a vertex colour -- inside the fused tile kernels' per-sample shading (a template flag of theirs); the backward returns
 * d loss / d maps (float atomics into gmaps_acc_d, which the caller zeroes) and d loss / d verts including the path through uv.
 *   hifihr_renderer_set_uv(h, faces_uvs_h[F][3], verts_uvs_h[n_uv][2], n_uv)   once per renderer
 *   maps_d[B][TH][TW][3]; texels_scratch_d / gtexels_scratch_d: reserved, pass NULL (hifihr_render_uv_scratch_bytes returns 0) */
int hifihr_renderer_set_uv(hifihr_renderer_t* h, const int32_t* faces_uvs_h, const float* verts_uvs_h, int n_uv);
size_t hifihr_render_uv_scratch_bytes(const hifihr_renderer_t* h, int B);
int hifihr_render_fwd_uv(const hifihr_renderer_t* h, const float* verts_d, const float* maps_d, int TH, int TW, const float* cam_d,
                         const float* light_color_d, const float* light_dir_d, int B, float* rgba_d, int32_t* face_id_d,
                         float* texels_scratch_d, void* ws_d, void* stream);
int hifihr_render_bwd_uv(const hifihr_renderer_t* h, const float* verts_d, const float* maps_d, int TH, int TW, const float* cam_d,
                         const float* light_color_d, const float* light_dir_d, const int32_t* face_id_d, const float* grad_rgba_d, int B,
                         const float* texels_scratch_d, float* gtexels_scratch_d, float* gverts_d, float* gmaps_acc_d,
                         float* glight_color_d, float* glight_dir_d, void* ws_d, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Texture-PCA decode (csrc/texpca.hip): tex[b][n] = mean[n] (or 0 when NULL) + sum_k coef[b][k] basis[k][n], K <= 32, n % 4 == 0.
 * The texture half of the NIMBLE layer as the reference consumes it (models_res_nimble.py:57,133-142: texture_params [B,10] -> the
 * hand's texture).  NIMBLE's own basis is not available: the caller supplies one (n = 778 * 3 vertex
 * colours for the declared stand-in, n = 1024 * 1024 * 3 for a UV map).  HBM-bound: 4 n (K + 1 + B) algorithmic bytes.
 * bwd: dcoef_zeroed_d[B][K] (ZERO on entry) += sum_n gtex[b][n] basis[k][n]   (float atomics: reproducible to rounding).
 * ---------------------------------------------------------------------------------------------- */
int hifihr_texture_pca_fwd(const float* coef_d, const float* basis_d, const float* mean_d /* or NULL */, int B, int K, long n,
                           float* tex_d, void* stream);
int hifihr_texture_pca_bwd(const float* gtex_d, const float* basis_d, int B, int K, long n, float* dcoef_zeroed_d, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Gradient exchange of the data-parallel step (RCCL over xGMI), for callers that do not go through torch.distributed.
 * Replaces nn.DataParallel (reference train_hrnet.py:560): one process per GPU, the ONE collective
 * of a step is a SUM all-reduce of the flat fp32 gradient buffer (the 1 / world scale is hifihr_adam_step's grad_scale),
 * parameters are broadcast once at start.  librccl is resolved with dlopen at the first call (no link-time dependency).
 *   rank 0: hifihr_comm_get_unique_id(&uid), ship the 128 bytes to the other ranks out of band;
 *   every rank (its device current): hifihr_comm_init(&comm, rank, world, &uid); per step hifihr_comm_allreduce_f32(comm, grads_d,
 *   n, stream) -- any number of calls on slices of the buffer (buckets) -- then hifihr_adam_step; hifihr_comm_destroy at the end.
 * Errors: HIFIHR_EHIP with the RCCL message in hifihr_comm_last_error().
 * ---------------------------------------------------------------------------------------------- */
typedef struct hifihr_comm hifihr_comm;
typedef struct { char bytes[128]; } hifihr_comm_uid;
const char* hifihr_comm_last_error(void);
int hifihr_comm_get_unique_id(hifihr_comm_uid* out);
int hifihr_comm_init(hifihr_comm** out, int rank, int world, const hifihr_comm_uid* uid);
int hifihr_comm_allreduce_f32(hifihr_comm* comm, float* buf_d /* in place */, size_t n, void* stream);
int hifihr_comm_broadcast_f32(hifihr_comm* comm, float* buf_d, size_t n, int root, void* stream);
int hifihr_comm_destroy(hifihr_comm* comm);

/* ------------------------------------------------------------------------------------------------
 * Fused Adam over ONE flat parameter buffer.
 * Replaces optimizer.step() of torch.optim.Adam(betas=(0.9,0.999), eps=1e-8, weight_decay=0|0.01)
 * reference train_hrnet.py:111-113, 546-551.  All four buffers are device fp32[n], 16-byte aligned.
 * grads are multiplied by grad_scale before use (1/world_size after a sum all-reduce).  `step` counts from 1.
 * ---------------------------------------------------------------------------------------------- */
int hifihr_adam_step(float* params_d, const float* grads_d, float* exp_avg_d, float* exp_avg_sq_d, size_t n,
                     float grad_scale, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                     void* stream);
/* Same update with the two per-step scalars read from DEVICE memory, dyn_d[2] = { lr / (1 - beta1^t),
 * 1 / sqrt(1 - beta2^t) }: the launch can be captured in a hipGraph and replayed while the host refreshes dyn_d
 * (a 8-byte async copy) outside the graph before each replay. */
/* The same update with the step counter and the learning rate in DEVICE memory (round 5): state_d is hifihr_adam_state_bytes() = 48 bytes,
 *   f64 lr, f64 beta1, f64 beta2, f64 beta1^step, f64 beta2^step, i32 step (completed steps), i32 0
 * written once by the caller (and again whenever lr or the counter changes: a scheduler step, a restored checkpoint).  The launch computes
 * lr / (1 - beta1^t) and 1 / sqrt(1 - beta2^t), t = step + 1, in double precision on the device (from the running products) and advances
 * `step` and the products when its last workgroup finishes: a captured training step replays with nothing to refresh from the host in between (hifihr_adam_step_dyn needs a 2-float upload
 * in front of every replay).  One launch at a time per state. */
size_t hifihr_adam_state_bytes(void);
int hifihr_adam_step_counted(float* params_d, const float* grads_d, float* exp_avg_d, float* exp_avg_sq_d, size_t n, float grad_scale,
                             float eps, float weight_decay, void* state_d, void* stream);
int hifihr_adam_step_dyn(float* params_d, const float* grads_d, float* exp_avg_d, float* exp_avg_sq_d, size_t n,
                         float grad_scale, float beta1, float beta2, float eps, float weight_decay, const float* dyn_d,
                         void* stream);

/* ------------------------------------------------------------------------------------------------
 * NHWC fp32 convolution on the f32 matrix cores (implicit GEMM, exact fp32 accumulate).
 * Replaces the cuDNN/MIOpen dispatches of the encoder's nn.Conv2d layers (forward, backward-data,
 * backward-weight): reference network/res_encoder.py:364-373 (ResNet trunk built at :345-362).
 * Layouts: x[N][H][W][C], w[K][R][S][C] (= a torch [K,C,R,S] tensor in channels_last memory format),
 * y[N][OH][OW][K] with OH = (H + 2 pad - R)/stride + 1.  C % 4 == 0; bwd_data needs K % 4 == 0 (K % 16 == 0 when stride > 1), bwd_weight K % 4 == 0.
 *
 * Workspace (ws_d / ws_bytes, may be NULL / 0): fwd, fwd_bnstats and bwd_data can run a BALANCED schedule -- exactly
 * 4 persistent workgroups per CU, each walking an equal share of the (tile, K-chunk) iterations, partial tiles summed
 * through ws_d -- instead of one workgroup per output tile, whose cost is quantised to whole rounds of 1024 workgroups
 * (784 tiles cost as much as 1024).  ws_d must hold hifihr_conv2d_workspace_bytes(...) bytes (0 = this shape never uses
 * one), be ALL ZERO before its first use and is returned all zero (self-cleaning); calls sharing a workspace must be
 * ordered on one stream.  Without a workspace the calls fall back to the data-parallel grid (same results up to fp32
 * summation order across K segments).
 * ---------------------------------------------------------------------------------------------- */
size_t hifihr_conv2d_workspace_bytes(int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int bwd_data);
/* act: 0 = none, 1 = ReLU after the bias (nn.Conv2d + nn.ReLU of the LightEstimator, network/res_encoder.py:150-210; its
 * backward first runs hifihr_bias_relu_bwd: g = dy * (y > 0) (g_d may alias dy_d), db_acc_d[K] += column sums of g). */
int hifihr_conv2d_fwd(const float* x_d, const float* w_d, const float* bias_d /* [K] or NULL */, int act, float* y_d, int N,
                      int H, int W, int C, int K, int R, int S, int stride, int pad, void* ws_d, size_t ws_bytes, void* stream);
int hifihr_bias_relu_bwd(const float* dy_d, const float* y_d, long M, int C, float* g_d, float* db_acc_d /* or NULL */,
                         void* stream);
/* LightEstimator.forward's last two lines (reference network/res_encoder.py:205-210): lights[B][6] ->
 * colors[B][3] = hardtanh(lights[:, :3]) (nn.Hardtanh: clamp to [-1, 1]), directions[B][3] = lights[:, 3:], contiguous, one launch;
 * bwd: glights[B][6] = [gcolors where -1 < lights < 1 else 0, gdirections] (either gradient may be NULL = zero). */
int hifihr_light_split_fwd(const float* lights_d, int B, float* colors_d, float* directions_d, void* stream);
int hifihr_light_split_bwd(const float* lights_d, const float* gcolors_d, const float* gdirections_d, int B, float* glights_d,
                           void* stream);
/* dx[N][H][W][C] (overwritten).  wt_scratch_d: K*R*S*C floats of scratch (receives the [C][R][S][K] transpose). */
int hifihr_conv2d_bwd_data(const float* dy_d, const float* w_d, float* dx_d, float* wt_scratch_d, int N, int H, int W, int C,
                           int K, int R, int S, int stride, int pad, void* ws_d, size_t ws_bytes, void* stream);
/* dw[K][R][S][C] += sum over pixels (ACCUMULATES with fp32 atomics: zero it, or pass the gradient buffer). */
int hifihr_conv2d_bwd_weight(const float* x_d, const float* dy_d, float* dw_d, int N, int H, int W, int C, int K, int R, int S,
                             int stride, int pad, void* stream);
/* The same with caller-provided scratch: layer 1's 3x3 and the stem's 7x7 weight gradients are summed from per-workgroup slabs in a
 * fixed order (bit-reproducible, no atomics; csrc/conv_halo.hip).  ws: hifihr_conv2d_wgrad_workspace_bytes(...) bytes, any contents,
 * not shared with a launch that may run concurrently on another stream; 0 bytes needed = the shape runs on the atomics kernel.
 * Without it (hifihr_conv2d_bwd_weight) those shapes use scratch the library allocates on first use outside a stream capture. */
size_t hifihr_conv2d_wgrad_workspace_bytes(int N, int H, int W, int C, int K, int R, int S, int stride, int pad);
int hifihr_conv2d_bwd_weight_ws(const float* x_d, const float* dy_d, float* dw_accumulate_d, int N, int H, int W, int C, int K, int R, int S,
                                int stride, int pad, void* ws_d, size_t ws_bytes, void* stream);
/* The stem as the reference declares it, nn.Conv2d(3, 64, 7, 2, 3) (vendored resnet.py conv1): the image arrives as NHWC4 with a
 * zero fourth plane (hifihr_image_to_nhwc4), the parameter and its gradient keep 3 channels, dw3_d[K][R][S][3] (accumulated into).
 * Served by the slab kernel only (7x7, stride 2, pad 3, K = 64; ask _supported); ws as hifihr_conv2d_wgrad_workspace_bytes(.., C = 4, ..).
 * The forward's padded filter [K][R][S][4] comes out of hifihr_weight_prep (job kind 5). */
int hifihr_conv2d_bwd_weight_c3_supported(int N, int H, int W, int K, int R, int S, int stride, int pad);
int hifihr_conv2d_bwd_weight_c3(const float* x4_d, const float* dy_d, float* dw3_d, int N, int H, int W, int K, int R, int S, int stride,
                                int pad, void* ws_d, size_t ws_bytes, void* stream);
/* Winograd F(2x2, 3x3) path for stride-1, pad-1 3x3 convolutions with many channels (ResNet-18 layers 3-4): 2.25x fewer
 * multiplications than the direct kernel, same results up to a few ulp.  Forward:
 *   hifihr_wino_weight_transform(w[K][3][3][C], U[16][K][C], K, C, flip = 0)
 *   hifihr_wino_input_transform (x[N][H][W][C], V[16][T][C]),  T = N * ceil(H/2) * ceil(W/2)
 *   hifihr_wino_gemm            (V, U, M[16][T][K])  -- 16 GEMMs on the MFMA kernel; ws as for hifihr_conv2d_fwd
 *   hifihr_wino_output_transform(M, y[N][H][W][K], stats_d or NULL)   -- stats_d: batch-norm slot buffer (zero on entry)
 * Backward-data is the same sequence on dy with the weights of the transposed, 180-degree rotated filter:
 *   weight_transform(wt[C][3][3][K] (= the [C][R][S][K] transpose hifihr_conv2d_bwd_data also builds), U[16][C][K], C, K, flip = 1).
 * Backward-weight: dU[16][K][C] (ZERO on entry) += the 16 batched reductions over the tiles of
 *   Y'[16][T][K] = hifihr_wino_dy_transform(dy)  and  V[16][T][C] (the forward's input transform of x), then
 *   hifihr_wino_dw_transform: dw[K][3][3][C] += G^T dU G  (accumulates, like hifihr_conv2d_bwd_weight). */
size_t hifihr_wino_gemm_workspace_bytes(int N, int H, int W, int C, int K);
int hifihr_wino_dy_transform(const float* dy_d, float* yt_d, int N, int H, int W, int K, void* stream);
int hifihr_wino_wgrad_gemm(const float* v_d, const float* yt_d, float* du_zeroed_d, int N, int H, int W, int C, int K, void* stream);
int hifihr_wino_dw_transform(float* du_d, float* dw_acc_d, int K, int C, int clear_du /* 1: leave du_d all zero (self-cleaning) */,
                             void* stream);
int hifihr_wino_weight_transform(const float* w_d, float* u_d, int K, int C, int flip, void* stream);
int hifihr_wino_input_transform(const float* x_d, float* v_d, int N, int H, int W, int C, void* stream);
int hifihr_wino_gemm(const float* v_d, const float* u_d, float* m_d, int N, int H, int W, int C, int K, void* ws_d, size_t ws_bytes,
                     void* stream);
int hifihr_wino_output_transform(const float* m_d, float* y_d, float* stats_d /* or NULL */, int N, int H, int W, int K, void* stream);
/* Backward of a Winograd layer reads dy twice (input transform for backward-data, hifihr_wino_dy_transform for backward-weight);
 * this does both from one read: v_d[16][T][K] = B^T d B of the padded 4x4 patches, yt_d[16][T][K] = A dy A^T of their central 2x2. */
int hifihr_wino_input_dy_transform(const float* dy_d, float* v_d, float* yt_d, int N, int H, int W, int K, void* stream);
/* Output transform with the act epilogue of hifihr_conv2d_fwd: y = act(A^T m A + bias[K] (or NULL)), act 0 = none, 1 = ReLU
 * (VGG19 layers of the perceptual loss, reference utils/perceptual_loss.py:27-36). */
int hifihr_wino_output_transform_act(const float* m_d, float* y_d, const float* bias_d /* or NULL */, int act, int N, int H, int W,
                                     int K, void* stream);
/* Winograd F(4x4, 3x3) (csrc/wino4.hip): the same pipeline with 4x4 output tiles -- 36 positions, T = N * ceil(H/4) * ceil(W/4), 4x fewer
 * multiplications than the direct convolution (F(2x2, 3x3): 2.25x), results within 1e-5 of it.  Every entry point above has an `_m`
 * form that takes the tile edge m (2 or 4; the plain names are m = 2); all calls of one layer must use the same m, and buffers are sized
 * with P = (m + 2)^2 positions: U[P][K][C], V[P][T][C], M[P][T][K], Y'[P][T][K], du_parts[parts][P][K][C].
 * hifihr_wino_tile(N, H, W, C, K) returns the m this library prefers for a layer: 4 where the batched GEMMs of csrc/gemm.hip take the
 * shape in all three directions (C % 64 == 0, K % 64 == 0, H, W >= 4), else 2.  m = 4 has the slab form of backward-weight only
 * (hifihr_wino_wgrad_parts_m > 0).  The per-step weight re-layout (hifihr_weight_prep) has job kinds 3 / 4 for U[36][K][C] / U'[36][C][K]. */
int hifihr_wino_tile(int N, int H, int W, int C, int K);
/* T, the tiles (rows per position of V / M / Y') a layer has at tile edge m: N * ceil(H / m) * ceil(W / m) -- or, at m = 4 on square maps
 * with H % 4 in {1, 2} and N % 16 == 0, the tiles of the 4 x 4-image MOSAICS the transforms cut (16 images share one map with single
 * lines of zeros between them: 225 tiles per 16 images of 14 x 14 instead of 256), rounded up to a multiple of 32.  Every buffer of the
 * pipeline is sized with it.  (HIFIHR_WINO_MOSAIC=0: always the first form.) */
long hifihr_wino_tiles(int N, int H, int W, int m);
/* ... and the rows of them the forward / backward-data products actually compute (the mosaic count before the rounding; else the same). */
long hifihr_wino_tiles_computed(int N, int H, int W, int m);
size_t hifihr_wino_gemm_workspace_bytes_m(int N, int H, int W, int C, int K, int m);
int hifihr_wino_weight_transform_m(const float* w_d, float* u_d, int K, int C, int flip, int m, void* stream);
int hifihr_wino_input_transform_m(const float* x_d, float* v_d, int N, int H, int W, int C, int m, void* stream);
int hifihr_wino_gemm_m(const float* v_d, const float* u_d, float* m_d, int N, int H, int W, int C, int K, int m, void* ws_d, size_t ws_bytes,
                       void* stream);
int hifihr_wino_output_transform_m(const float* m_d, float* y_d, float* stats_d /* or NULL */, int N, int H, int W, int K, int m, void* stream);
int hifihr_wino_output_transform_act_m(const float* m_d, float* y_d, const float* bias_d /* or NULL */, int act, int N, int H, int W, int K,
                                       int m, void* stream);
/* Output transform of a BACKWARD-DATA product whose layer input was a ReLU's output (VGG19 conv + ReLU -> conv): y_d = mask_d > 0 ? A^T m A : 0
 * with mask_d[N][H][W][K] = that ReLU output (the layer's saved input) -- the ReLU's backward where its gradient is produced, instead of a
 * hifihr_bias_relu_bwd pass in front of the previous layer's backward.  m == 4 only. */
int hifihr_wino_output_transform_mask_m(const float* m_d, float* y_d, const float* mask_d, int N, int H, int W, int K, int m, void* stream);
int hifihr_wino_dy_transform_m(const float* dy_d, float* yt_d, int N, int H, int W, int K, int m, void* stream);
/* 3x3 / stride 1 / pad 1 convolution with 64 input and 64 output channels as Winograd F(2x2, 3x3) in ONE launch, the transforms in
 * registers and nothing of the transform domain in HBM (round 3, csrc/conv_halo.hip: conv_wino2_kernel) -- ResNet layer 1 (reference
 * network/res_encoder.py:364-373 -> torchvision BasicBlock conv1 / conv2 of layer1) and VGG19 conv1_2 of the perceptual loss (reference
 * utils/perceptual_loss.py:27-36).  u_d = U[16][64][64]: hifihr_wino_weight_transform(w, u, 64, 64, 0) or hifihr_weight_prep kind 1 for
 * the forward; kind 2 (the transposed, rotated filter) with x_d = dy gives backward-data.  bias_d / relu: epilogue (VGG19); stats_d: the
 * forward statistics slots of the batch norm that follows (hifihr_bn_stats_floats(64) floats, zeroed) or NULL.  H even, W even and >= 14
 * (hifihr_conv3x3_c64_wino_supported); results within 1e-5 (relative to the output scale) of the direct convolution. */
int hifihr_conv3x3_c64_wino_supported(int N, int H, int W, int C, int K);
/* 1 when the library's page of zeros for this device exists (or could be allocated now: `stream` is not capturing).  The kernels whose
 * loaders read out-of-image taps from it (hifihr_conv3x3_c64_wino, the ragged / gathering row-share GEMM) cannot allocate it inside a stream
 * capture: a caller that is about to capture asks here first and otherwise picks the kernels that do not need it. */
int hifihr_zero_page_ready(void* stream);
int hifihr_conv3x3_c64_wino(const float* x_d, const float* u_d, const float* bias_d /* or NULL */, int relu, float* y_d,
                            float* stats_d /* or NULL */, int N, int H, int W, void* stream);
/* The same launch with y = product + res_d[N][H][W][64] (round 4).  Backward-data of a layer whose input has a SECOND consumer -- the
 * identity branch of a residual block (reference torchvision BasicBlock: `out += identity`, network/res_encoder.py:364-373): autograd adds
 * the two gradients of that input in an elementwise pass of its own; here the other consumer's gradient is added where this one is produced. */
int hifihr_conv3x3_c64_wino_res(const float* x_d, const float* u_d, const float* res_d, float* y_d, int N, int H, int W, void* stream);
/* Both gradients of one such layer in ONE launch (round 5): dx_d = hifihr_conv3x3_c64_wino[_res](dy_d, u_bwd_d [, res_d or NULL]) -- u_bwd_d the
 * Winograd-domain filters of the transposed convolution (hifihr_weight_prep kind 2) -- and dw_d[64][3][3][64] += the weight gradient of
 * hifihr_conv2d_bwd_weight(x_d, dy_d) (the pixel-reduction slab kernel + its fixed-order slab sum).  The two halves are independent,
 * ~50 and ~65 us at the config batch: separately each pays its own ramp-up and tail, here the workgroups of one launch are split between
 * them.  Results are those of the separate calls up to the summation order over workgroup shares (the weight gradient stays
 * bit-reproducible run to run).  ws_d: scratch of hifihr_conv2d_bwd_weight_ws_bytes or NULL (library-owned).  Replaces, for ResNet layer 1
 * (reference utils/Freihand_GNN_mano/network/resnet.py BasicBlock convs at 56 x 56 x 64), the pair of autograd nodes
 * cudnn_convolution_backward_input / _weight. */
int hifihr_conv3x3_c64_bwd_pair_supported(int N, int H, int W);
int hifihr_conv3x3_c64_bwd_pair(const float* dy_d, const float* u_bwd_d, const float* res_d /* or NULL */, float* dx_d, const float* x_d,
                                float* dw_d, void* ws_d, size_t ws_bytes, int N, int H, int W, void* stream);
/* The same pair launch with the slab sum LEFT to the caller: the weight-gradient slabs stay in slabs_d ([*nslab][9][64][64], slab_bytes >=
 * hifihr_conv2d_wgrad_workspace_bytes of the layer) and hifihr_conv_halo_wgrad_reduce_multi adds the slabs of several layers to their
 * dw_acc_d[64][3][3][64] in ONE launch, slabs in slab order (bit-identical to the sum hifihr_conv3x3_c64_bwd_pair makes itself).  A step
 * whose weight gradients are read by the optimizer only defers the sums of its 64 -> 64 layers to one call in front of it. */
int hifihr_conv3x3_c64_bwd_pair_slabs(const float* dy_d, const float* u_bwd_d, const float* res_d /* or NULL */, float* dx_d, const float* x_d,
                                      void* slabs_d, size_t slab_bytes, int N, int H, int W, int* nslab /* host, out */, void* stream);
typedef struct hifihr_halo_reduce_job {
  const float* slabs_d;
  float* dw_acc_d;
  int nslab;
} hifihr_halo_reduce_job;
int hifihr_conv_halo_wgrad_reduce_multi(const hifihr_halo_reduce_job* jobs /* host array */, int njobs, void* stream);
/* Batch-norm fused into the F(4x4, 3x3) input transform (round 3, csrc/wino4_bn.hip).  For a BatchNorm2d whose consumer is a Winograd
 * convolution (reference BasicBlock: conv1 -> bn1 -> relu -> conv2; bn2 -> += identity -> relu -> the next block's conv1,
 * network/res_encoder.py:364-373 + vendored utils/Freihand_GNN_mano/network/resnet.py) this ONE launch replaces hifihr_bn_act_fwd followed by
 * hifihr_wino_input_transform_m: x_d is the RAW output of the previous convolution, stats_d its batch statistics (consumed: zero on
 * return, as in hifihr_bn_act_fwd, same save_mean / save_invstd / running-statistics semantics); the activation a = relu(bn(x) + residual?)
 * is formed on the fly and only V = B^T a B is written.  With residual_d, out_d (same shape as x) receives a as well -- the next block's
 * identity branch reads it; without, both are NULL and a is never stored (the backward recomputes the ReLU mask from x, as
 * hifihr_bn_act_bwd does with y_d = NULL).  m = 4 and C % 4 == 0, C <= 512 (hifihr_wino_bn_input_supported). */
int hifihr_wino_bn_input_supported(int C, int m);
int hifihr_wino_bn_input_transform(const float* x_d, float* stats_d, const float* gamma_d, const float* beta_d, const float* residual_d,
                                   float* out_d, float* v_d, int N, int H, int W, int C, int m, float eps, float momentum,
                                   float* save_mean_d, float* save_invstd_d, float* running_mean_d, float* running_var_d, void* stream);
/* The backward of that fusion (round 3): the batch-norm backward no longer makes passes of its own over the gradient.
 *   hifihr_wino_output_transform_bnred: output transform of the backward-data product m_d[36][T][C] of the consuming convolution
 *     = d loss / d a; in its epilogue + gadd_d (the gradient that reached the block output through the next block's identity
 *     branch, or NULL), the ReLU mask (out_d > 0 when the forward added a residual, else recomputed from x_d), g_d = the masked
 *     gradient written once, and the batch-norm backward REDUCTION (sum g, sum g xhat) added into red_d (hifihr_bn_stats_floats(C)
 *     floats, all zero on entry).  Replaces hifihr_wino_output_transform_m + the autograd add + the reduction half of hifihr_bn_act_bwd.
 *   then EITHER hifihr_bn_bwd_apply: the apply half of hifihr_bn_act_bwd on (g_d, red_d): dx, dgamma_acc += , dbeta_acc +=, red_d zeroed;
 *   OR, when the batch-norm's input was itself produced by a Winograd convolution, hifihr_wino_bn_bwd_dual_transform in THAT
 *     convolution's backward: dy = gamma invstd (g - mean g - xhat mean(g xhat)) is evaluated on the fly and only V' = B^T dy B and
 *     Y' = A dy A^T are written (y_d = the convolution's raw output = the batch-norm's input; same dgamma / dbeta / red_d semantics).
 * Mirrors, in one direction each, torch.autograd through nn.BatchNorm2d + ReLU + `+= identity` of the reference's BasicBlock
 * (vendored utils/Freihand_GNN_mano/network/resnet.py).  m = 4, channels % 4 == 0 and <= 512. */
int hifihr_wino_output_transform_bnred(const float* m_d, const float* x_d, const float* out_d /* or NULL */, const float* gadd_d /* or NULL */,
                                       const float* save_mean_d, const float* save_invstd_d, const float* gamma_d, const float* beta_d,
                                       float* red_d, float* g_d, int N, int H, int W, int C, int m, void* stream);
int hifihr_wino_bn_bwd_dual_transform(const float* g_d, const float* y_d, const float* save_mean_d, const float* save_invstd_d,
                                      const float* gamma_d, float* red_d, float* v_d, float* yt_d, int N, int H, int W, int K, int m,
                                      float* dgamma_acc_d, float* dbeta_acc_d, void* stream);
int hifihr_bn_bwd_apply(const float* g_d, const float* x_d, const float* save_mean_d, const float* save_invstd_d, const float* gamma_d, long M,
                        int C, float* red_d, float* dx_d, float* dgamma_acc_d, float* dbeta_acc_d, void* stream);
int hifihr_wino_input_dy_transform_m(const float* dy_d, float* v_d, float* yt_d, int N, int H, int W, int K, int m, void* stream);
int hifihr_wino_wgrad_parts_m(int N, int H, int W, int C, int K, int m);
int hifihr_wino_wgrad_gemm_parts_m(const float* v_d, const float* yt_d, float* du_parts_d, int N, int H, int W, int C, int K, int parts, int m,
                                   void* stream);
/* The two products of a Winograd F(4x4, 3x3) layer's backward that do not depend on each other, in ONE launch (round 5):
 *   backward-data   M2[36][T][C] = V2[36][T][K] . U2[36][C][K]^T      (= hifihr_wino_gemm_m(V2, U2, M2, N, H, W, K, C, 4, ...))
 *   backward-weight dU_parts     = Yt[36][T][K]^T . Vx[36][T][C]      (= hifihr_wino_wgrad_gemm_parts_m(Vx, Yt, dU_parts, ..., parts, 4))
 * with V2 / Yt the two transforms of dy and Vx the forward's transformed input (the layer maps C -> K channels).  Workgroups of one
 * launch split between the two (a short launch of these kernels is shaped by its start and its end); falls back to the two launches
 * for shapes that are not on the row-share kernels.  Results identical to the separate calls. */
int hifihr_wino4_bwd_gemm_pair_supported(int N, int H, int W, int C, int K);   /* 1: the call below pairs; 0: it makes the two launches */
int hifihr_wino4_bwd_gemm_pair(const float* V2_d, const float* U2_d, float* M2_d, const float* Vx_d, const float* Yt_d, float* dU_parts_d,
                               int N, int H, int W, int C, int K, int parts, void* stream);
int hifihr_wino_dw_transform_parts_m(const float* du_parts_d, int parts, float* dw_acc_d, int K, int C, int m, void* stream);
/* The same F(4x4, 3x3) weight-gradient transform (m = 4) for SEVERAL layers in one launch: dw_acc_d[K][3][3][C] += G^T (sum of the `parts`
 * slabs du_parts_d[parts][36][K][C]) G per job, jobs independent of each other.  `jobs` is a HOST array (its pointers are device pointers);
 * the entries travel in the kernel arguments, so nothing is uploaded and a captured launch keeps them.  Results identical to one
 * hifihr_wino_dw_transform_parts_m(..., 4, ...) call per job.  A training step whose weight gradients are only read by the optimizer
 * collects its layers' jobs during backward and makes this one call in front of the optimizer (reference train_hrnet.py:104-105). */
typedef struct hifihr_wino_dw_job {
  const float* du_parts_d;
  float* dw_acc_d;
  int parts, K, C;
} hifihr_wino_dw_job;
int hifihr_wino4_dw_transform_multi(const hifihr_wino_dw_job* jobs, int njobs, void* stream);
/* Batched fp32 GEMM on the f32 matrix cores (csrc/gemm.hip): the plain products a Winograd layer consists of -- the GEMM half of
 * the vendor-library call behind one conv2d of the reference (network/res_encoder.py:364-373).  hifihr_wino_gemm dispatches here
 * when the shape allows (C % 32 == 0, K % 64 == 0; hifihr_wino_gemm_workspace_bytes then returns 0).
 *   hifihr_bgemm_nt: c[b][M][N] = a[b][M][K] . b[b][N][K]^T          (K % 32 == 0, N % 64 == 0, any M); with a workspace of
 *                    hifihr_bgemm_nt_workspace_bytes (zero-initialised once, handed back all zero) large shapes run on the persistent,
 *                    balanced kernel (one workgroup per CU, stream-K shares), else one workgroup per tile
 *   hifihr_bgemm_tn: c_parts[z][b][M][N] = sum over the t rows of slab z of a[b][t][M] (x) b[b][t][N]   (M, N % 64 == 0, any T);
 *                    `parts` = hifihr_bgemm_tn_parts(M, N, T, batch) slabs, to be summed by the consumer (no atomics).
 * Winograd backward-weight on it: parts = hifihr_wino_wgrad_parts(N, H, W, C, K) (0: shape unsupported, use hifihr_wino_wgrad_gemm),
 *   hifihr_wino_wgrad_gemm_parts(V, Y', du_parts[parts][16][K][C])  then
 *   hifihr_wino_dw_transform_parts: dw[K][3][3][C] += G^T (sum of the slabs) G   -- nothing zero-initialised, bit-reproducible. */
size_t hifihr_bgemm_nt_workspace_bytes(int M, int N, int K, int batch);
int hifihr_bgemm_nt(const float* a_d, const float* b_d, float* c_d, int M, int N, int K, int batch,
                    void* ws_d /* zero-initialised, self-cleaning; or NULL */, size_t ws_bytes, void* stream);
int hifihr_bgemm_tn_parts(int M, int N, int T, int batch);
/* Name of the kernel instantiation a shape runs on, as a profiler lists it ("" when the shape is not supported): measurement only. */
int hifihr_bgemm_describe(int tn, int M, int N, int K_or_T, char* out, int cap);                       /* batch = 16 */
int hifihr_bgemm_describe_batch(int tn, int M, int N, int K_or_T, int batch, char* out, int cap);
int hifihr_bgemm_tn(const float* a_d, const float* b_d, float* c_parts_d, int M, int N, int T, int batch, int parts, void* stream);
int hifihr_wino_wgrad_parts(int N, int H, int W, int C, int K);
int hifihr_wino_wgrad_gemm_parts(const float* v_d, const float* yt_d, float* du_parts_d, int N, int H, int W, int C, int K, int parts,
                                 void* stream);
int hifihr_wino_dw_transform_parts(const float* du_parts_d, int parts, float* dw_acc_d, int K, int C, void* stream);
/* [K][RS][C] -> [C][RS][K] (the transpose backward-data consumes). */
int hifihr_weight_transpose(const float* w_d, float* wt_d, int K, int RS, int C, void* stream);
/* hifihr_conv2d_bwd_data on weights that are ALREADY transposed to [C][R][S][K] (hifihr_weight_transpose / hifihr_weight_prep). */
int hifihr_conv2d_bwd_data_pre(const float* dy_d, const float* wt_d, float* dx_d, int N, int H, int W, int C, int K, int R, int S,
                               int stride, int pad, void* ws_d, size_t ws_bytes, void* stream);
/* dx = backward-data + res_d[N][H][W][C] (round 4; see hifihr_conv3x3_c64_wino_res): the first block of a ResNet stage sends its input to
 * the strided 3x3 convolution AND the 1x1 downsample convolution -- the second backward-data adds the first one's result as it stores. */
int hifihr_conv2d_bwd_data_pre_res(const float* dy_d, const float* wt_d, const float* res_d, float* dx_d, int N, int H, int W, int C, int K,
                                   int R, int S, int stride, int pad, void* ws_d, size_t ws_bytes, void* stream);
/* dx = backward-data of the strided convolution (dy_d, wt_d as in hifihr_conv2d_bwd_data_pre) + backward-data of a SECOND convolution of the same
 * input: 1x1, the same stride, pad 0, the same channel counts (dy2_d [N][OH][OW][K], wt2_d [C][K] = its transposed filter) -- the downsample
 * branch of a residual stage's first block (reference: vendored torchvision BasicBlock, network/res_encoder.py:364-373 runs both through
 * cuDNN / MIOpen and autograd adds the two gradients).  The 1x1 convolution's gradient lands on the pixels (stride i, stride j) only: it is one
 * more tap of that parity class of the strided launch instead of a launch of its own plus a residual pass (round 6).
 * _supported: 1 when the shape runs on that path (stride >= 2, K % 16 == 0, equal output grids); HIFIHR_DGRAD_PLUS1X1=0 switches it off. */
int hifihr_conv2d_bwd_data_pre_plus1x1_supported(int N, int H, int W, int C, int K, int R, int S, int stride, int pad);
int hifihr_conv2d_bwd_data_pre_plus1x1(const float* dy_d, const float* wt_d, const float* dy2_d, const float* wt2_d, float* dx_d, int N, int H,
                                       int W, int C, int K, int R, int S, int stride, int pad, void* stream);
/* dw_d[K][R][S][C] += weight gradient of the strided convolution, dw2_d[K][C] += weight gradient of the 1x1 / same stride / pad 0 convolution of
 * the same input with the same output channels (dy2_d [N][OH][OW][K]), in ONE launch: the second convolution's patch column is tap (pad, pad)
 * of the first (round 6; the downsample branch of a residual stage's first block, as hifihr_conv2d_bwd_data_pre_plus1x1).  Float atomics, like
 * hifihr_conv2d_bwd_weight on these shapes.  _supported: 1 when the shape runs on that path; HIFIHR_WGRAD_PLUS1X1=0 switches it off. */
int hifihr_conv2d_bwd_weight_plus1x1_supported(int N, int H, int W, int C, int K, int R, int S, int stride, int pad);
int hifihr_conv2d_bwd_weight_plus1x1(const float* x_d, const float* dy_d, float* dw_d, const float* dy2_d, float* dw2_d, int N, int H, int W,
                                     int C, int K, int R, int S, int stride, int pad, void* stream);
/* Every per-step weight re-layout of a model in ONE launch.  The weights change once per optimizer step; a ResNet-18 step
 * otherwise spends ~40 tiny launches (~5 us of launch floor each) on transposes and Winograd weight transforms.
 * jobs_d: DEVICE array of njobs descriptors (src / dst are device pointers);
 *   kind 0: dst[C][RS][K] = transpose of src[K][RS][C]                                   (= hifihr_weight_transpose)
 *   kind 1: dst = U[16][K][C] of src[K][3][3][C]                                         (= hifihr_wino_weight_transform, flip 0)
 *   kind 2: dst = U'[16][C][K], the backward-data weights of src[K][3][3][C]              (= weight_transpose + transform, flip 1)
 *   kind 3 / 4: the F(4x4, 3x3) forms of 1 / 2, U[36][K][C] / U'[36][C][K]
 *   kind 5: dst[K][RS][C4] = src[K][RS][C] with the channels zero-padded to C4 = the next multiple of 4 (the 3-channel stem filter)
 * blocks_per_job: workgroups per job (each job is a grid-stride loop). */
typedef struct hifihr_prep_job {
  const float* src;
  float* dst;
  int K, C, RS, kind;
} hifihr_prep_job;
int hifihr_weight_prep(const hifihr_prep_job* jobs_d, int njobs, int blocks_per_job, void* stream);

/* Name of the kernel a bias-free convolution of this shape runs on, as a profiler lists it: direction 0 forward, 1 backward-data
 * ("conv_halo_kernel": 3x3 / stride 1 / 64 -> 64 channels with the width a multiple of 14, csrc/conv_halo.hip; else
 * "conv_igemm_kernel"), 2 backward-weight ("conv_halo_wgrad_kernel" / "conv_wgrad_kernel").  Measurement only. */
int hifihr_conv2d_describe(int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int direction, char* out, int cap);

/* conv2d_fwd that also accumulates the per-channel sum and sum of squares of y into stats_d (hifihr_bn_stats_floats(K)
 * floats, ALL ZERO on entry: see the self-cleaning rule below) from the accumulator registers, so the batch-norm that
 * follows needs no pass over y. */
int hifihr_conv2d_fwd_bnstats(const float* x_d, const float* w_d, float* y_d, float* stats_d, int N, int H, int W, int C, int K,
                              int R, int S, int stride, int pad, void* ws_d, size_t ws_bytes, void* stream);
/* Two convolutions of the SAME input in one launch (round 6): y1 = conv(x, w1; R1 x R1, pad1), y2 = conv(x, w2; R2 x R2, pad2), both with
 * `stride` and with their batch-norm statistics (stats1_d / stats2_d: the slot-buffer contract of hifihr_conv2d_fwd_bnstats) -- the strided
 * 3x3 convolution of a residual stage's first block and the 1x1 convolution of its downsample branch (torchvision BasicBlock.conv1 +
 * downsample[0], reference network/res_encoder.py:364-373), whose short 1x1 launch otherwise pays its start and end alone.  Results
 * identical to two hifihr_conv2d_fwd_bnstats calls.  _supported: both shapes run on the gathering row-share GEMM (C % 32 == 0,
 * K % 128 == 0, R in {1, 3}, stride 2). */
int hifihr_conv2d_fwd_bnstats_pair_supported(int N, int H, int W, int C, int stride, int K1, int R1, int pad1, int K2, int R2, int pad2);
int hifihr_conv2d_fwd_bnstats_pair(const float* x_d, const float* w1_d, float* y1_d, float* stats1_d, int K1, int R1, int pad1, const float* w2_d,
                                   float* y2_d, float* stats2_d, int K2, int R2, int pad2, int N, int H, int W, int C, int stride, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Train-mode BatchNorm2d fused with the residual add and ReLU of a ResNet BasicBlock, NHWC: x[M][C], M = N*H*W.
 * Replaces nn.BatchNorm2d(training=True) + `out += identity` + nn.ReLU and their autograd in the trunk
 * (reference network/res_encoder.py:364-373; vendored BasicBlock utils/Freihand_GNN_mano/network/resnet.py).
 * C % 4 == 0, C <= 4096.  act: 0 = none, 1 = ReLU, 2 = swish (x * sigmoid(x); MemoryEfficientSwish of the reference's
 * EfficientNet, network/efficientnet_pt/utils.py:36-52; no residual with swish).  stats_d / red_scratch_d hold hifihr_bn_stats_floats(C) floats: partial (sum, sum of
 * squares) over the M rows, spread over several slots to keep atomic contention low (from
 * hifihr_conv2d_fwd_bnstats, or hifihr_bn_stats for any other producer).  Since round 3 the FORWARD partials are float64
 * (double[32][2][C] at the start of the buffer, which must be 8-byte aligned): producers accumulate sums shifted by a value of their own
 * in fp32 and hand them over unshifted in fp64, the consumer forms mean and variance in fp64 -- the batch variance of a channel with
 * |mean| >> std is as good as a Welford pass (PyTorch's nn.BatchNorm2d), which `E[x^2] - mean^2` in fp32 was not.  The layout is private
 * to the library: callers only allocate, zero once, and pass the buffer along.
 * SELF-CLEANING: producers (hifihr_conv2d_fwd_bnstats, hifihr_dwconv2d_fwd, hifihr_wino_output_transform, hifihr_bn_stats,
 * the reduction inside hifihr_bn_act_bwd) ADD into stats_d / red_scratch_d, which must be all zero on entry;
 * hifihr_bn_act_fwd / hifihr_bn_act_bwd fold the slots inside their apply kernel (no separate finalize launch) and the
 * last workgroup of that kernel to finish -- elected through arrival counters stored behind the slots -- writes the zeros
 * back, so one zero-initialised buffer serves every step without a memset launch.
 *   fwd: y = act( (x - mean) * invstd * gamma + beta + residual? ); writes save_mean/save_invstd[C] and updates
 *        running_mean/var (momentum, unbiased variance) when given.
 *   bwd: g = dy * act'(z) (ReLU: y > 0 from y_d -- or, with y_d NULL and no residual input in the forward, the mask recomputed from x, needs beta_d; swish: z recomputed from x, needs beta_d); dx = gamma*invstd*(g - mean(g) - xhat*mean(g*xhat)); dres (may be NULL) = g;
 *        dgamma_acc[C] += sum g*xhat, dbeta_acc[C] += sum g (either may be NULL).
 * ---------------------------------------------------------------------------------------------- */
int hifihr_bn_stats_floats(int C);
int hifihr_bn_stats(const float* x_d, long M, int C, float* stats_d, void* stream);
int hifihr_bn_act_fwd(const float* x_d, float* stats_d /* consumed: zero on return */, const float* gamma_d, const float* beta_d,
                      const float* residual_d /* or NULL */, int act, long M, int C, float eps, float momentum, float* y_d,
                      float* save_mean_d, float* save_invstd_d, float* running_mean_d, float* running_var_d, void* stream);
/* Evaluation mode (module.eval(), reference train_hrnet.py:119-161): the same fused apply with the running statistics; nothing
 * is updated, no statistics buffer is involved. */
int hifihr_bn_act_eval(const float* x_d, const float* running_mean_d, const float* running_var_d, const float* gamma_d, const float* beta_d,
                       const float* residual_d /* or NULL */, int act, long M, int C, float eps, float* y_d, void* stream);
int hifihr_bn_act_bwd(const float* dy_d, const float* y_d /* act 1; NULL: recompute the mask (no-residual layers) */, const float* x_d, const float* save_mean_d,
                      const float* save_invstd_d, const float* gamma_d, const float* beta_d /* act 2 */, int act, long M, int C,
                      float* red_scratch_d, float* dx_d, float* dres_d, float* dgamma_acc_d, float* dbeta_acc_d, void* stream);

/* The ResNet stem, bn1 -> relu -> maxpool fused: pooled = MaxPool2d(3, 2, 1)(ReLU(BN(x))) on x[N][H][W][C] (reference: the
 * vendored trunk utils/Freihand_GNN_mano/network/resnet.py forward, `x = self.maxpool(self.relu(self.bn1(self.conv1(x))))`, driven
 * from network/res_encoder.py:364-373).  The full-resolution activation and its gradient never reach HBM: the forward reads the
 * nine taps of x through scale / shift / ReLU (tie rule of nn.MaxPool2d: first tap in scan order), the backward gathers the
 * pool's gradient from pooled_grad + tap on the fly inside the batch-norm reduction and apply kernels.  Same statistics-buffer
 * contract (all zero on entry, all zero on return) and the same save_mean / save_invstd / running-statistics outputs as
 * hifihr_bn_act_fwd; C % 4 == 0, C <= 512, H, W >= 2.  pooled_d [N][OH][OW][C], tap_d one byte per pooled element,
 * OH = (H - 1) / 2 + 1. */
int hifihr_bn_relu_maxpool_supported(int N, int H, int W, int C);
int hifihr_bn_relu_maxpool_fwd(const float* x_d, float* stats_d, const float* gamma_d, const float* beta_d, int N, int H, int W, int C,
                               float eps, float momentum, float* pooled_d, unsigned char* tap_d, float* save_mean_d, float* save_invstd_d,
                               float* running_mean_d, float* running_var_d, void* stream);
int hifihr_bn_relu_maxpool_bwd(const float* pooled_grad_d, const unsigned char* tap_d, const float* x_d, const float* save_mean_d,
                               const float* save_invstd_d, const float* gamma_d, const float* beta_d, int N, int H, int W, int C,
                               float* red_scratch_d, float* dx_d, float* dgamma_acc_d, float* dbeta_acc_d, void* stream);
/* The same backward with the forward's pooled OUTPUT handed back (pooled_d [N][OH][OW][C]): the batch-norm reduction then walks the pooled
 * grid -- the pool's gradient lives at the winning taps only and the winner's normalised value follows from the pooled value itself,
 * xhat = (z - beta) / gamma (channels with |gamma| < 1e-3 fetch the winner's x through tap_d) -- a quarter of the bytes of the pass over
 * every input pixel.  Results equal hifihr_bn_relu_maxpool_bwd's to the rounding of the sums. */
int hifihr_bn_relu_maxpool_bwd_y(const float* pooled_grad_d, const float* pooled_d, const unsigned char* tap_d, const float* x_d,
                                 const float* save_mean_d, const float* save_invstd_d, const float* gamma_d, const float* beta_d, int N, int H, int W,
                                 int C, float* red_scratch_d, float* dx_d, float* dgamma_acc_d, float* dbeta_acc_d, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Depthwise convolution (groups == channels), NHWC fp32, k = 3 or 5, TensorFlow-style asymmetric zero padding.
 * Replaces the depthwise Conv2dStaticSamePadding of the reference's EfficientNet MBConv blocks
 * (reference network/efficientnet_pt/model.py:49-55,80; utils.py:122-145) and its autograd.
 * x[N][H][W][C], w[C][K][K] (= torch [C,1,K,K]), y[N][OH][OW][C]; stride 1 or 2; pad_top/pad_left explicit, bottom/right
 * implied by OH/OW.  bwd_weight ACCUMULATES into dw (fp32 atomics).  fwd: stats_d (may be NULL) = batch-norm slot buffer
 * (hifihr_bn_stats_floats(C) floats, all zero on entry, self-cleaning: see the batch-norm section) that receives the
 * per-channel sum / sum of squares of y, so the BatchNorm that follows needs no statistics pass.
 * ---------------------------------------------------------------------------------------------- */
int hifihr_dwconv2d_fwd(const float* x_d, const float* w_d, float* y_d, float* stats_d /* or NULL */, int N, int H, int W, int C,
                        int OH, int OW, int K, int stride, int pad_top, int pad_left, void* stream);
int hifihr_dwconv2d_bwd_data(const float* dy_d, const float* w_d, float* dx_d, int N, int H, int W, int C, int OH, int OW, int K,
                             int stride, int pad_top, int pad_left, void* stream);
int hifihr_dwconv2d_bwd_weight(const float* x_d, const float* dy_d, float* dw_d, int N, int H, int W, int C, int OH, int OW,
                               int K, int stride, int pad_top, int pad_left, void* stream);
/* The expand half of an MBConv block without its activated tensor (round 5): reference network/efficientnet_pt/model.py:73-80,
 *   x = swish(bn0(expand_conv(x)));  x = depthwise_conv(x)
 * x_d is the RAW output of the expand convolution; a = swish(x * gamma invstd + (beta - mean gamma invstd)) is formed as the depthwise
 * kernels load their rows (zero outside the image: the padding is of a) and never reaches HBM.  mean_d / invstd_d [C]: the batch
 * statistics of x (hifihr_bn_finalize_fwd below makes them from the convolution's slot buffer).  fwd: y, stats as hifihr_dwconv2d_fwd;
 * bwd_weight: dw += sum dy * a.  The gradient with respect to x is hifihr_dwconv2d_bwd_data followed by hifihr_bn_act_bwd(act = swish)
 * on (that, x): the batch-norm backward recomputes the activation's derivative from x alone. */
int hifihr_dwconv2d_fwd_bnswish(const float* x_d, const float* mean_d, const float* invstd_d, const float* gamma_d, const float* beta_d,
                                const float* w_d, float* y_d, float* stats_d /* or NULL */, int N, int H, int W, int C, int OH, int OW,
                                int K, int stride, int pad_top, int pad_left, void* stream);
int hifihr_dwconv2d_bwd_weight_bnswish(const float* x_d, const float* mean_d, const float* invstd_d, const float* gamma_d,
                                       const float* beta_d, const float* dy_d, float* dw_d, int N, int H, int W, int C, int OH, int OW,
                                       int K, int stride, int pad_top, int pad_left, void* stream);
/* The statistics half of a training-mode batch-norm on its own (what hifihr_bn_act_fwd does before it applies): the slot buffer of a
 * producer -> save_mean[C], save_invstd[C] (biased variance, eps inside the root), running statistics updated with `momentum`
 * (NULL: not kept), slots handed back all zero.  reference: nn.BatchNorm2d in training mode (model.py:73-76). */
int hifihr_bn_finalize_fwd(float* stats_d, long M, int C, float eps, float momentum, float* save_mean_d, float* save_invstd_d,
                           float* running_mean_d /* or NULL */, float* running_var_d /* or NULL */, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Pooling on NHWC fp32 activations (C % 4 == 0).
 * mmpool: MMPool((1,1)) of the reference encoder (network/res_encoder.py:247-265, called at :49):
 *   y[B][C] = max_hw(x) * w + mean_hw(x) * (1 - w), w = sigmoid(p_d[0]); x[B][HW][C].  fwd also writes argmax[B][C] (first
 *   maximum in scan order, like adaptive_max_pool2d), xmax[B][C], xavg[B][C] for the backward.
 *   bwd: dx[B][HW][C] (overwritten) and dp_acc_d[0] += d loss / d p (may be NULL).
 * maxpool2d: nn.MaxPool2d(k, s, p) for (k, s, p) = (3, 2, 1) (after the ResNet stem, network/res_encoder.py:345-373,
 *   torchvision resnet18.maxpool), (3, 1, 1) and (2, 2, 0) (LightEstimator, network/res_encoder.py:150-210):
 *   y[N][OH][OW][C], OH = (H + 2p - k)/s + 1; tap_d[N][OH][OW][C] bytes = winning tap (first maximum in scan order).
 *   bwd gathers: dx[N][H][W][C] is overwritten (no zero fill needed).
 * ---------------------------------------------------------------------------------------------- */
int hifihr_mmpool_fwd(const float* x_d, const float* p_d, int B, int HW, int C, float* y_d, int* argmax_d, float* xmax_d,
                      float* xavg_d, void* stream);
int hifihr_mmpool_bwd(const float* gy_d, const float* p_d, const int* argmax_d, const float* xmax_d, const float* xavg_d, int B,
                      int HW, int C, float* dx_d, float* dp_acc_d, void* stream);
int hifihr_maxpool2d_fwd(const float* x_d, int N, int H, int W, int C, int k, int s, int p, float* y_d, unsigned char* tap_d,
                         void* stream);
int hifihr_maxpool2d_bwd(const float* gy_d, const unsigned char* tap_d, int N, int H, int W, int C, int k, int s, int p,
                         float* dx_d, void* stream);
/* The same backward for a pool whose INPUT was a ReLU's output (VGG19: conv + ReLU -> MaxPool2d, reference utils/perceptual_loss.py:27-36
 * over torchvision's features): y_d = the pool's own output; a window's gradient passes only where y > 0, which IS the ReLU's backward at
 * the winning tap (the other taps receive nothing) -- dx_d is then the gradient of the convolution output, no pass over (dy, relu output)
 * of the four-times larger pre-pool tensor is needed. */
int hifihr_maxpool2d_bwd_relu(const float* gy_d, const unsigned char* tap_d, const float* y_d, int N, int H, int W, int C, int k, int s, int p,
                              float* dx_d, void* stream);
/* The pool whose output is FLATTENED next (LightEstimator: `base_layers(x).view(B, -1)` -> Linear, network/res_encoder.py:199-201): y_flat_d is
 * the [N][C * OH * OW] matrix in the reference's NCHW order, gy_flat_d its gradient; x / dx stay channels-last.  Saves the copy kernel a
 * reshape of the channels-last tensor costs in each direction. */
int hifihr_maxpool2d_fwd_flat(const float* x_d, int N, int H, int W, int C, int k, int s, int p, float* y_flat_d, unsigned char* tap_d,
                              void* stream);
int hifihr_maxpool2d_bwd_flat(const float* gy_flat_d, const unsigned char* tap_d, int N, int H, int W, int C, int k, int s, int p, float* dx_d,
                              void* stream);

/* normalize_batch_3C (reference network/res_encoder.py:212-216) fused with NCHW[B][3][H][W] -> NHWC4 [B][H][W][4]
 * (4th channel zero) for the first convolution. */
int hifihr_image_to_nhwc4(const float* images_d, float* out_d, int B, int H, int W, void* stream);
/* Same repack with an explicit zero border and optional normalisation: out[B][H + pt + pb][W + pl + pr][4].  Serves the
 * EfficientNet stem (reference network/efficientnet_pt/model.py:197-199: no input normalisation, Conv2dStaticSamePadding
 * k3 s2 pads (left 0, right 1, top 0, bottom 1)), which then runs on hifihr_conv2d_fwd with pad = 0. */
int hifihr_image_to_nhwc4_padded(const float* images_d, float* out_d, int B, int H, int W, int pad_top, int pad_left,
                                 int pad_bottom, int pad_right, int normalize, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fused SSIM (11x11 gaussian window sigma 1.5, zero padding, C1 = 0.01^2, C2 = 0.03^2, mean over all elements).
 * Replaces pytorch_ssim.ssim(img1, img2)   reference utils/pytorch_ssim/__init__.py:17-37,65-73
 * (called at reference losses.py:375 for the ssim_tex term) and its autograd w.r.t. img1.
 * img1/img2: [planes][H][W] with planes = B*C (contiguous NCHW); window11_h: the 11 normalised taps (HOST).
 * fwd writes one partial sum per 16x16 tile: SSIM = sum(partial[0..hifihr_ssim_partial_count)) / (planes*H*W)
 * (summed by the caller; deterministic).  dA/dB/dC ([planes][H][W] each, all three or none) receive the
 * derivative maps the backward call consumes.  bwd: gimg1 = grad_out[0] * d mean(SSIM) / d img1; grad_out_d is
 * a DEVICE scalar (no host sync).
 * ---------------------------------------------------------------------------------------------- */
int hifihr_ssim_partial_count(int planes, int H, int W);
int hifihr_ssim_fwd(const float* window11_h, const float* img1_d, const float* img2_d, int planes, int H, int W,
                    float* partial_d, float* dA_d, float* dB_d, float* dC_d, void* stream);
int hifihr_ssim_bwd(const float* window11_h, const float* img1_d, const float* img2_d, const float* dA_d, const float* dB_d,
                    const float* dC_d, const float* grad_out_d, int planes, int H, int W, float* gimg1_d, void* stream);
/* The scalar glue of the call site in two more entry points (reference losses.py:375-377, `lambda * (1 - ssim)`):
 *   hifihr_ssim_finish: out_d[0] = offset + scale * sum(partial)   (SSIM: scale = 1 / n, offset = 0; the loss term: scale = -lambda / n,
 *                       offset = lambda) -- one launch, fixed summation order;
 *   hifihr_ssim_bwd_scaled: hifihr_ssim_bwd with the incoming gradient multiplied by out_scale (= -lambda for the loss term). */
int hifihr_ssim_finish(const float* partial_d, int count, float scale, float offset, float* out_d, void* stream);
int hifihr_ssim_bwd_scaled(const float* window11_h, const float* img1_d, const float* img2_d, const float* dA_d, const float* dB_d,
                           const float* dC_d, const float* grad_out_d, float out_scale, int planes, int H, int W, float* gimg1_d, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fused training losses.  Replaces the torch-op evaluation of reference losses.py:226-453 (LossFunction.forward) and its
 * autograd for the terms below; every value comes back already multiplied by its lambda, as loss_dic holds it.
 *
 * geom_loss: out_d[5] = (joint_3d, vert_3d, edge_length, mshape, mpose)
 *   joint_3d / vert_3d = lambda * F.l1_loss or F.mse_loss (mse = 0 / 1: args.base_loss_fn, losses.py:259-266),
 *   edge_length = lambda * mean |edge(pred) - edge(gt)| over the 3 edges of every face (utils/losses_util.py:285-301),
 *   mshape / mpose = lambda * F.mse_loss(params, 0) (losses.py:398-406).
 *   joints[B][J][3], verts[B][V][3], shape[B][NS], pose[B][NP], faces_d[F][3] (F = 0 / NULL: no edge term);
 *   lambda5_h: HOST.  partial_d: B*5 floats of scratch.  bwd: gout_d[5] = gradient of each out element (DEVICE);
 *   vf_off_d[V+1] / vf_idx_d[3F] = vertex -> (face * 4 + corner) incidence lists in ascending face order (the edge
 *   gradient is gathered per vertex: deterministic, no atomics); any of gj/gv/gshape/gpose may be NULL.
 * photo_loss (the photometric block, losses.py:355-378, plus the `sil` term :388-390), from the renderer's rgba[B][4][H][W]:
 *   re_img_m = rgb * re_sil / 255 with re_sil = (alpha > 0 ? 255 : alpha); mask_rgbs = seg * imgs  (both written:
 *   the SSIM term consumes them); out_d[4] = (texture, mrgb, sil, mean(re_img_m) - mean(mask_rgbs)).
 *   seg_d: int64 [B][H][W] (segms_gt).  H*W % 4 == 0.  partial_d: hifihr_photo_loss_partial_floats() floats.
 *   bwd: grad_rgba[B][4][H][W] (overwritten; alpha channel 0: re_sil is detached in the reference) from gout_d[>=2]
 *   (texture, mrgb; may be NULL) and g_re_img_d (gradient arriving at re_img_m, e.g. from SSIM; may be NULL).
 * sil_post (models_res_nimble.py:219-220): re_sil[B][H][W] and maskRGBs[B][3][H][W] = images * (re_sil > 0) (may be NULL).
 * ---------------------------------------------------------------------------------------------- */
int hifihr_geom_loss_fwd(const float* joints_d, const float* joints_gt_d, const float* verts_d, const float* verts_gt_d,
                         const float* shape_d, const float* pose_d, const int32_t* faces_d, int B, int J, int V, int F, int NS,
                         int NP, int mse, const float* lambda5_h, float* partial_d, float* out_d, void* stream);
/* joint_2d, bone_direc, bone_direc_3d of LossFunction.__call__ (reference losses.py:267-282; bone_direction_loss of
 * utils/losses_util.py:217-283 with confidence 1) in ONE launch per direction: j2d / j2d_gt [B][21][2] (or both NULL), joints /
 * joints_gt [B][21][3] (or both NULL); lam3_host = (lambda_j2d_gt, lambda_bone_direc, lambda_bone_direc_3d), host memory; out3_d =
 * the three lambda-weighted terms (a term whose inputs are NULL is 0).  bwd: g_j2d_d / g_joints_d (either may be NULL) = gradient of
 * sum_k gout3_d[k] out3[k]. */
int hifihr_joint_terms_fwd(const float* j2d_d, const float* j2d_gt_d, const float* joints_d, const float* joints_gt_d, int B, int J, int mse,
                           const float* lam3_host, float* out3_d, void* stream);
int hifihr_joint_terms_bwd(const float* j2d_d, const float* j2d_gt_d, const float* joints_d, const float* joints_gt_d, int B, int J, int mse,
                           const float* lam3_host, const float* gout3_d, float* g_j2d_d, float* g_joints_d, void* stream);
int hifihr_geom_loss_bwd(const float* joints_d, const float* joints_gt_d, const float* verts_d, const float* verts_gt_d,
                         const float* shape_d, const float* pose_d, const int32_t* faces_d, const int32_t* vf_off_d,
                         const int32_t* vf_idx_d, int B, int J, int V, int F, int NS, int NP, int mse, const float* lambda5_h,
                         const float* gout_d, float* gj_d, float* gv_d, float* gshape_d, float* gpose_d, void* stream);
int hifihr_photo_loss_partial_floats(void);
int hifihr_photo_loss_fwd(const float* rgba_d, const float* imgs_d, const int64_t* seg_d, int B, int H, int W, float l_tex,
                          float l_mrgb, float l_sil, float* re_img_m_d, float* mask_rgbs_d, float* partial_d, float* out_d,
                          void* stream);
int hifihr_photo_loss_bwd(const float* rgba_d, const float* re_img_m_d, const float* mask_rgbs_d, const float* g_re_img_d,
                          const float* gout_d, const float* fwd_out_d, int B, int H, int W, float l_tex, float l_mrgb,
                          float* grad_rgba_d, void* stream);
int hifihr_sil_post(const float* rgba_d, const float* imgs_d, int B, int H, int W, float* re_sil_d, float* mask_rgbs_d,
                    void* stream);
/* loss = sum of the selected terms (reference train_hrnet.py:98-104: the sum over args.losses of loss_dic) when the terms are entries of
 * the fused loss kernels' small output vectors: total_d[0] = sum_i sum_{j < counts[i]} parts[i][j], parts in order (fixed summation
 * order); nparts <= 4, counts <= 64.  `parts` / `grads` are HOST arrays of DEVICE pointers.
 * bwd: grads[i][j] = gtotal_d[0] for j < counts[i], 0 for counts[i] <= j < lengths[i] (the vectors' full lengths). */
int hifihr_loss_total_fwd(const float* const* parts, const int* counts, int nparts, float* total_d, void* stream);
int hifihr_loss_total_bwd(const float* gtotal_d, float* const* grads, const int* counts, const int* lengths, int nparts,
                          void* stream);

/* ------------------------------------------------------------------------------------------------
 * Small-batch fully connected layer of the regression heads: y[B][O] = act( BN1d?( x[B][I] W[O][I]^T + b ) ).
 * Replaces nn.Linear (+ nn.BatchNorm1d, training mode) (+ nn.ReLU) and their autograd in the reference's HandEncoder /
 * LightEstimator heads (reference network/res_encoder.py:52-145, 150-210).  Any I (16-byte loads when I % 4 == 0); act 0 = none, 1 = ReLU,
 * 2 = swish (z_d receives the pre-activation, needed by the backward), 3 = sigmoid -- 2 / 3 serve the squeeze-excite
 * layers below and take no batch-norm.
 * Batch-norm (gamma_d != NULL, B <= 64): batch statistics over the B rows, running statistics updated with `momentum`
 * (unbiased variance), z_d[B][O] receives the pre-normalisation output, save_mean_d / save_invstd_d[O] the statistics.
 * bwd: dy_d = gradient of y; y_d (act 1) gives the ReLU mask; dz_scratch_d[B][O]; dW_acc_d[O][I], db_acc_d[O],
 *      dgamma_acc_d[O], dbeta_acc_d[O] ACCUMULATE (pass the gradient buffers); dx_d[B][I] is overwritten (NULL: skipped).
 * ---------------------------------------------------------------------------------------------- */
int hifihr_linear_fwd(const float* x_d, const float* w_d, const float* b_d /* or NULL */, int B, int I, int O, int act,
                      const float* gamma_d /* or NULL: no batch-norm */, const float* beta_d, float eps, float momentum,
                      float* running_mean_d, float* running_var_d, float* y_d, float* z_d, float* save_mean_d,
                      float* save_invstd_d, void* stream);
int hifihr_linear_bwd(const float* dy_d, const float* y_d, const float* x_d, const float* w_d, int B, int I, int O, int act,
                      const float* gamma_d, const float* z_d, const float* save_mean_d, const float* save_invstd_d,
                      float* dz_scratch_d, float* dW_acc_d, float* db_acc_d, float* dgamma_acc_d, float* dbeta_acc_d,
                      float* dx_d, void* stream);

/* Grouped launches for independent layers of the same depth (the HandEncoder's five to six heads, reference
 * network/res_encoder.py:112-131, 146-160): up to 6 members per call, no batch-norm, act 0 / 1; one launch forward, two backward
 * instead of one / two PER LAYER (each is a latency-bound ~8 us kernel on 8-32 workgroups).  `descs` is a HOST array.
 * bwd members: dy, y (act 1), dz_scratch[B][O] required; dW_acc / db_acc accumulate (NULL: skipped); dx overwritten (NULL: skipped). */
typedef struct hifihr_linear_desc {
  const float *x, *w, *b; /* b may be NULL */
  float* y;
  int B, I, O, act;
  const float* dy;
  float *dz_scratch, *dW_acc, *db_acc, *dx;
} hifihr_linear_desc;
int hifihr_linear_fwd_group(const hifihr_linear_desc* descs, int n, void* stream);
int hifihr_linear_bwd_group(const hifihr_linear_desc* descs, int n, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Squeeze-and-excitation of the EfficientNet MBConv block (reference network/efficientnet_pt/model.py:82-86):
 *   y = x * sigmoid( W2 swish( W1 mean_hw(x) + b1 ) + b2 ),  x[B][HW][C] NHWC, C % 4 == 0.
 * se_pool: mean_d[B][C] += mean over HW (mean_d must be ZERO on entry: partial sums are added with atomics).
 * The two layers are hifihr_linear_fwd / _bwd with act 2 and 3.  se_scale: y = x * gate[b][c] (+ add[b][c] * add_scale when
 * add_d != NULL).  Backward: se_bwd_gate: dgate_d[B][C] += sum_hw dy * x (zero on entry); then the linear backwards give
 * dmean; finally dx = se_scale(dy, gate, add = dmean, add_scale = 1 / HW) folds the pooling branch into the same pass.
 * ---------------------------------------------------------------------------------------------- */
int hifihr_se_pool(const float* x_d, int B, int HW, int C, float* mean_zeroed_d, void* stream);
int hifihr_se_scale(const float* x_d, const float* gate_d, const float* add_d /* or NULL */, float add_scale, int B, int HW, int C,
                    float* y_d, void* stream);
int hifihr_se_bwd_gate(const float* dy_d, const float* x_d, int B, int HW, int C, float* dgate_zeroed_d, void* stream);
/* The two layers as ONE launch forward and TWO backward (round 4; replaces the four hifihr_linear_* calls per block and direction of
 * reference network/efficientnet_pt/model.py:83-86).  w1_d[SQ][C], w2t_d[SQ][C] = the TRANSPOSE of the expand weight W2[C][SQ]
 * (hifihr_weight_prep kind 0 provides it once per step), C % 4 == 0, C <= 4096, SQ <= 256 (hifihr_se_mlp_supported).
 * fwd: reads the pooled means from mean_acc_d[B][C] (what hifihr_se_pool accumulated) and hands that buffer back ZEROED; writes
 *      mean_d[B][C], z1_d[B][SQ] (pre-activation), h1_d[B][SQ] = swish(z1), gate_d[B][C] = sigmoid(h1 W2^T + b2).
 * bwd: reads dgate_acc_d[B][C] (hifihr_se_bwd_gate's sums; handed back ZEROED); writes dz2_d[B][C], dz1_d[B][SQ] (scratch) and
 *      dmean_d[B][C]; ACCUMULATES (+=) dW1[SQ][C], db1[SQ], dW2[C][SQ], db2[C] -- each element summed over the batch by one thread in a
 *      fixed order (bit-reproducible). */
int hifihr_se_mlp_supported(int C, int SQ);
/* Drop-connect + skip connection of an MBConv block in one pass (reference network/efficientnet_pt/utils.py:82-91 `drop_connect`,
 * model.py:91-94): out_d = x_d / keep * floor(keep + u_d[b]) (+ skip_d when not NULL); x / skip / out [B][per_sample], per_sample % 4 == 0,
 * u_d[B] the per-sample uniform draws.  The backward is the same call on dy with skip_d = NULL. */
int hifihr_drop_connect_add(const float* x_d, const float* skip_d /* or NULL */, const float* u_d, float keep, int B, size_t per_sample,
                            float* out_d, void* stream);
int hifihr_se_mlp_fwd(float* mean_acc_d, const float* w1_d, const float* b1_d, const float* w2t_d, const float* b2_d, int B, int C, int SQ,
                      float* mean_d, float* z1_d, float* h1_d, float* gate_d, void* stream);
int hifihr_se_mlp_bwd(float* dgate_acc_d, const float* gate_d, const float* z1_d, const float* h1_d, const float* mean_d, const float* w1_d,
                      const float* w2t_d, int B, int C, int SQ, float* dz2_d, float* dz1_d, float* dmean_d, float* dw1_acc_d, float* db1_acc_d,
                      float* dw2_acc_d, float* db2_acc_d, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Evaluation: Procrustes-with-scale alignment of pred_d[B][N][3] to gt_d[B][N][3] and the
 * aligned error, one workgroup per sample, fp64 inside.  Replaces the per-sample numpy loop of reference
 * train_hrnet.py:227-243 over utils/train_utils.py:267-290 (align_w_scale: centre, Frobenius-normalise,
 * scipy.linalg.orthogonal_procrustes -- no determinant correction --, apply).
 *   aligned_d[B][N][3] (or NULL) = the aligned prediction;  err_sum_d[B] = sum_n || aligned_n - gt_n ||_2
 * (MPJPE / MPVPE = sum_b err_sum_d[b] / (B N), in the unit of the inputs).
 * ---------------------------------------------------------------------------------------------- */
int hifihr_procrustes_error(const float* pred_d, const float* gt_d, int B, int N, float* aligned_d /* or NULL */,
                            float* err_sum_d, void* stream);

/* ------------------------------------------------------------------------------------------------
 * FreiHAND training augmentation: gathers B samples from a uint8 dataset cache resident in
 * device memory and applies the reference's nearest-neighbour affine warp (reference data/dataset.py:223-270,
 * utils/handutils.py:48-60 = PIL Image.transform(AFFINE), zero fill) + to_tensor (+ torch.round for masks).
 *   img_rgbx_d[n][H][W]   uint32, bytes R, G, B, X     mask_d[n][H][W] uint8 (0 / 255)
 *   idx_d[B]              sample indices into the cache
 *   coef_fix_d[B][6]      PIL's 16.16 fixed-point terms of the six AFFINE coefficients (a b c; d e f):
 *                         {FIX(a), FIX(b), FIX(c + a/2 + b/2), FIX(d), FIX(e), FIX(f + d/2 + e/2)}, FIX(v) = floor(v * 65536 + 0.5)
 *   out_img_d[B][3][H][W] = u8 / 255,  out_mask_d[B][3][H][W] = round(u8 / 255) repeated over 3 channels (either may be NULL)
 * Bit-exact with PIL for the same coefficients.
 * ---------------------------------------------------------------------------------------------- */
int hifihr_freihand_augment(const uint32_t* img_rgbx_d, const uint8_t* mask_d, const int* idx_d, const int* coef_fix_d, int B, int H,
                            int W, float* out_img_d, float* out_mask_d, void* stream);

/* One FreiHAND training batch in two launches: the warp above (plus segms = mask channel 0 as int64, the reference's
 * `masks[:, 0].long()`, utils/traineval_util.py:104) and everything else the sample dict and data_dic hold
 * (reference data/dataset.py:256-275 per sample, utils/traineval_util.py:21-111 per batch):
 *   Ks = post_rot_trans . K[idx];  joints / verts = (R p^T)^T;  Ps = [Ks | 0];  j2d_gt = (Ks j)_xy / (Ks j)_z (fh_utils.py:30-39);
 *   scales[idx];  idxs as int64.
 * Cache (device): Ks_d[n][3][3], joints_d[n][J][3], verts_d[n][V][3], scales_d[n].
 * packed_d[25 B] int32, ONE host-to-device copy per batch: idx[B], coef_fix[B][6], post_rot_trans[B][3][3] (float bits),
 * rot_mat[B][3][3] (float bits).  Every output may be NULL.  Outputs: out_Ks[B][3][3], out_Ps[B][3][4], out_joints[B][J][3],
 * out_verts[B][V][3], out_j2d[B][J][2], out_scales[B], out_idxs[B] (int64), out_segm[B][H][W] (int64). */
int hifihr_freihand_batch(const uint32_t* img_rgbx_d, const uint8_t* mask_d, const float* Ks_d, const float* joints_d, const float* verts_d,
                          const float* scales_d, int J, int V, const int* packed_d, int B, int H, int W, float* out_img_d,
                          float* out_mask_d, long long* out_segm_d, float* out_Ks_d, float* out_Ps_d, float* out_joints_d,
                          float* out_verts_d, float* out_j2d_d, float* out_scales_d, long long* out_idxs_d, void* stream);
/* The same two launches, which additionally emit what every training iteration derives from the batch before the model runs
 * (reference train_hrnet.py:62-68: root_xyz = joints[:, ROOT]; joints -= root; verts -= root;  models_res_nimble.py:184-186,228-235:
 * the NDC camera terms of PerspectiveCameras(focal_length=-fcl, principal_point=prp)):
 *   out_root[B][3] = joints[:, root_id] (root_id < 0: zeros), out_joints_rel[B][J][3], out_verts_rel[B][V][3],
 *   out_cam_ndc[B][4] = (-2 fx / s, -2 fy / s, 1 - 2 cx / s, 1 - 2 cy / s) with s = image_size, from out_Ks.  Any may be NULL. */
int hifihr_freihand_batch_step(const uint32_t* img_rgbx_d, const uint8_t* mask_d, const float* Ks_d, const float* joints_d,
                               const float* verts_d, const float* scales_d, int J, int V, const int* packed_d, int B, int H, int W,
                               float* out_img_d, float* out_mask_d, long long* out_segm_d, float* out_Ks_d, float* out_Ps_d,
                               float* out_joints_d, float* out_verts_d, float* out_j2d_d, float* out_scales_d, long long* out_idxs_d,
                               int root_id, float image_size, float* out_root_d, float* out_joints_rel_d, float* out_verts_rel_d,
                               float* out_cam_ndc_d, void* stream);

/* ------------------------------------------------------------------------------------------------
 * HO-3D training sample assembly (the HO-3D half): the hand crop of reference data/dataset.py:1105-1215.
 * Frames live in device memory as for FreiHAND (img_rgbx_d[n][FH][FW] uint32 R,G,B,X; hand_mask_d[n][FH][FW] uint8 = channel 0 of the
 * reference's mask image); the host computes each sample's crop window from its projected joints (hifihr_amd/data.py:ho3d_crop_windows,
 * the reference's float32 arithmetic) and ships ONE packed int32 buffer per batch:
 *   packed_d[8 B]: idx[B], box[B][4] = the crop box (x0, y0, x1, y1) rounded as Pillow's Image.crop rounds it (half to even),
 *                  window[B][3] = crop centre u, v and scale (float bits)
 * Three launches: the resampling tables of Pillow's Image.resize (libImaging/Resample.c, evaluated in double on the device), the
 * crop + resize of frame (bilinear) and hand mask (bicubic) = torchvision's resized_crop on PIL images, and the small tensors:
 *   out_img_d[B][3][S][S] = u8 / 255 (`img_crop`), out_mask_d[B][1][S][S] = round(u8 / 255) (`hand_mask_crop`), S = out_size (224),
 *   out_K_d[B][3][3] = T . S . K (`K_crop`, :1206-1210), out_uv21_d[B][21][2] = (uv21 - centre) * scale + S / 2 (`uv21_crop`, :1186-1188),
 *   out_xyz21_d[B][21][3] = xyz21[idx].  Any output may be NULL.  Pixels bit-exact with Pillow (tests/golden/ho3d_path.npz).
 * ws_d: hifihr_ho3d_workspace_bytes(B, out_size) bytes of scratch (any contents).
 * ---------------------------------------------------------------------------------------------- */
size_t hifihr_ho3d_workspace_bytes(int B, int out_size);
int hifihr_ho3d_batch(const uint32_t* img_rgbx_d, const uint8_t* hand_mask_d, const float* Ks_d, const float* uv21_d, const float* xyz21_d,
                      int FH, int FW, const int* packed_d, int B, int out_size, void* ws_d, size_t ws_bytes, float* out_img_d,
                      float* out_mask_d, float* out_K_d, float* out_uv21_d, float* out_xyz21_d, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HIFIHR_H */
