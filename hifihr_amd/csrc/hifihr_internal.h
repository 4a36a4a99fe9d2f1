// Internal declarations shared by the HIP translation units of libhifihr.so (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <stdint.h>

#include "render_math.h"

namespace hifihr {

constexpr int kNVP = 800;   // padded vertex count of the SoA MANO tables (rows are 16-byte aligned)

// Device-resident MANO tables, structure-of-arrays over the vertex index so that a wave reads 256
// contiguous bytes per table row.  Padding entries are zero.
struct ManoDev {
  const float* tmpl;   // [3][kNVP]
  const float* sd;     // [10][3][kNVP]    shapedirs
  const float* pd;     // [135][3][kNVP]   posedirs
  const float* w;      // [16][kNVP]       skinning weights, transposed
  const float* jreg;   // [16][kNVP]       joint regressor (dense)
  const float* comps;  // [45][45]
  const float* mean;   // [45]
  const float* jt;     // [16*3]           J_regressor @ v_template
  const float* jsd;    // [16*3][10]       J_regressor @ shapedirs
};

hipError_t launch_mano_fwd(const ManoDev& t, const float* pose, const float* beta, int B, float* verts, float* jtr,
                           float* saved, hipStream_t st);
hipError_t launch_mano_bwd(const ManoDev& t, const float* pose, const float* beta, const float* saved,
                           const float* gverts, const float* gjtr, int B, float* gpose, float* gbeta, hipStream_t st);
hipError_t launch_mano_full_fwd(const ManoDev& t, const float* pose, const float* beta, int B, int root_id, const float* root_xyz,
                                float* verts, float* joints_rel, float* verts_rel, float* verts_cam, float* root_out, float* saved,
                                hipStream_t st);
hipError_t launch_mano_full_bwd(const ManoDev& t, const float* pose, const float* beta, const float* saved, const float* gjoints_rel,
                                const float* gverts_rel, const float* gverts_cam, const float* groot, const float* gpose_add,
                                const float* gbeta_add, int B, int root_id, float* gpose, float* gbeta, hipStream_t st);
hipError_t launch_mano_joints_fwd(const ManoDev& t, const float* verts, int B, int root_id, float* joints_rel,
                                  float* verts_rel, float* root, hipStream_t st);
hipError_t launch_mano_joints_bwd(const ManoDev& t, const float* gjoints_rel, const float* gverts_rel,
                                  const float* groot, int B, int root_id, float* gverts, hipStream_t st);

// Generic linear-blend skinning (csrc/lbs.hip): any vertex / joint / shape-component count, sparse skin weights.
constexpr int kLbsMaxJ = 32, kLbsMaxS = 32, kLbsMaxK = 8;
struct LbsDev {
  int V, Vp, J, S, K;   // vertices, padded vertices (multiple of 64), joints, shape components, skin weights kept per vertex
  const float* tmpl;    // [3][Vp]
  const float* sd;      // [S][3][Vp]
  const int* widx;      // [K][Vp]   joint index of the k-th weight
  const float* wval;    // [K][Vp]   (zero padded)
  const float* jt;      // [J*3]
  const float* jsd;     // [J*3][S]
  const int* parent;    // [J]       parent[0] = -1, parent[i] < i
};
hipError_t launch_lbs_fwd(const LbsDev& t, const float* theta, const float* beta, int B, float* verts, float* joints, hipStream_t st);
hipError_t launch_lbs_bwd(const LbsDev& t, const float* theta, const float* beta, const float* gverts, const float* gjoints, int B,
                          float* gA_zeroed, float* gtheta, float* gbeta_zeroed, hipStream_t st);

// Renderer handle contents (device pointers + constants); passed to kernels by value.
struct RenderDev {
  int V, F, H, aa;
  const int* faces;    // [F][3]
  const int* vf_off;   // [V+1]   CSR vertex -> incident faces
  const int* vf_idx;   // [3F]    face * 4 + role (0,1,2 = which corner of the face the vertex is)
  ShadeConsts sc;
  float bg[3];
};

size_t render_workspace_bytes(const RenderDev& r, int B);
// TexturesUV mode of the renderer (csrc/render.hip, template flag UV of the tile kernels): per-face UV indices, UV coordinates, the
// texture maps of the batch
struct TexUvPass {
  const int* faces_uvs;        // [F][3]
  const float* verts_uvs;      // [Vt][2]
  const float* maps;           // [B][TH][TW][3]
  float* gmaps;                // backward: [B][TH][TW][3], accumulated into; or null
  int TH, TW;
};
hipError_t launch_render_fwd(const RenderDev& r, const float* verts, const float* vcolors, long vcol_bstride, const float* cam,
                             const float* light_color, const float* light_dir, int B, float* rgba, int* face_id, void* ws,
                             hipStream_t st, const TexUvPass* uv = nullptr);
hipError_t launch_render_bwd(const RenderDev& r, const float* verts, const float* cam, const float* light_color,
                             const float* light_dir, const int* face_id, const float* grad_rgba, int B, float* gverts,
                             float* gvcolors, float* glight_color, float* glight_dir, void* ws, hipStream_t st, const TexUvPass* uv = nullptr);

// Implicit-GEMM convolution geometry.  src [N][IH][IW][IC] is gathered, dst [N][OH][OW][OC] is written.
// forward: src = x, dst = y (ih = oh*stride - pad + r); dgrad = 1: src = dy, dst = dx (ih = (oh + pad - r)/stride).
struct ConvGeom {
  int N, IH, IW, IC, OH, OW, OC, R, S, stride, pad, dgrad;
  int relu;      // forward only: clamp the output at 0 after the bias
  // batched GEMM use (Winograd): `batch` independent problems of this geometry, element strides between them
  int batch;
  long src_bs, wgt_bs, dst_bs;
  // backward-data only (conv_igemm_kernel's epilogue): dst = product + residual (dst's layout) -- the gradient of the OTHER consumer of the
  // layer's input (a block's identity branch, a second convolution of the same tensor) added where this one is produced, instead of an
  // elementwise pass over both afterwards (round 4; ops._Conv2dMFMA fork)
  const float* residual = nullptr;
  // backward-data only (conv_igemm_kernel, stride > 1, round 6): a SECOND convolution of the same input -- 1x1, the same stride, pad 0, the same
  // channel counts (a residual stage's downsample branch) -- whose data gradient lands on the pixels of parity class (0, 0) only: it rides in
  // that class as one more tap, src2 = its dy [N][OH][OW][IC], wgt2 = its transposed filter [OC][IC], instead of a launch of its own whose
  // stride-times-stride larger, mostly zero result would come back in as `residual`
  const float* src2 = nullptr;
  const float* wgt2 = nullptr;
};
// sk_ws (may be NULL): zero-initialised, self-cleaning workspace of conv_sk_workspace_bytes(g) bytes for the balanced schedule
hipError_t launch_conv_igemm(const ConvGeom& g, const float* src, const float* wgt, const float* bias, float* dst, float* stats,
                             void* sk_ws, size_t sk_ws_bytes, hipStream_t st);
size_t conv_sk_workspace_bytes(const ConvGeom& g);
bool conv_wgrad_plus1x1_supported(const ConvGeom& g);
hipError_t launch_conv_wgrad_plus1x1(const ConvGeom& g, const float* x, const float* dy, float* dw, const float* dy2, float* dw2, hipStream_t st);
// 3x3 / stride 1 / 64 -> 64 channels with the input halo staged once per tile (csrc/conv_halo.hip); forward (+ BN statistics) and dgrad
bool conv_halo_supported(const ConvGeom& g, const float* bias);
const float* conv_halo_zero_page(hipStream_t st);
// the stem (7x7 / stride 2 / NHWC4 -> 64 channels) with the filter resident in LDS (csrc/conv_halo.hip); forward (+ BN statistics)
bool conv_stem_supported(const ConvGeom& g, const float* bias);
hipError_t launch_conv_stem(const ConvGeom& g, const float* src, const float* wgt, float* dst, float* stats, const float* zeros, hipStream_t st);
bool conv_stem_wgrad_supported(const ConvGeom& g);
size_t conv_stem_wgrad_slab_bytes();
hipError_t launch_conv_stem_wgrad(const ConvGeom& g, const float* x, const float* dy, float* dw, float* slabs_or_null, hipStream_t st,
                                  int dw_channels = 4);
bool conv_halo_wgrad_supported(const ConvGeom& g);
// hipErrorNotReady: the scratch could not be set up now (first use inside a stream capture) -- use conv_wgrad_kernel for this launch
size_t conv_halo_wgrad_slab_bytes();
hipError_t launch_conv_halo_wgrad(const ConvGeom& g, const float* x, const float* dy, float* dw, float* slabs_or_null, hipStream_t st);
// the same layers as Winograd F(2x2, 3x3) with the transforms in registers (conv_halo.hip: conv_wino2_kernel); U[16][64][64] = kind 1 / 2
bool conv_wino2_supported(int N, int H, int W, int C, int K);
hipError_t launch_conv_wino2(const float* src, const float* U, const float* bias_or_null, int relu, float* dst, float* stats_or_null, int N,
                             int H, int W, hipStream_t st, const float* res_or_null = nullptr);
// data gradient (launch_conv_wino2 with the transposed filters, + res_or_null) and weight gradient (launch_conv_halo_wgrad) of one such layer in
// ONE launch + the slab reduction (conv_halo.hip: conv_c64_bwd_pair_kernel)
bool conv_c64_bwd_pair_supported(int N, int H, int W);
hipError_t launch_conv_c64_bwd_pair(const float* dy, const float* U_bwd, const float* res_or_null, float* dx, const float* x, float* dw,
                                    float* slabs_or_null, int N, int H, int W, hipStream_t st, int* nslab_out = nullptr);
struct HaloReduceJob { const float* slabs; float* dw; int nslab; };       // = hifihr_halo_reduce_job (include/hifihr.h)
hipError_t launch_conv_halo_wgrad_reduce_multi(const HaloReduceJob* jobs, int njobs, hipStream_t st);
hipError_t launch_conv_halo(const ConvGeom& g, const float* src, const float* wgt, const float* bias_or_null, float* dst, float* stats,
                            const float* zeros, hipStream_t st);
// batch-norm (+ residual add + ReLU) on NHWC activations, x[M][C].  Statistics buffers are [kStatSlots][2][C]: partial
// sums are spread over kStatSlots copies so that at most ~1/32 of the contributing workgroups hit one address with a
// float atomic (thousands of atomics on ONE address serialise at ~100 ns each: 200 us per reduction in round 1).
constexpr int kStatSlots = 32;
// Slots a layer actually USES (the others stay zero; the layout keeps all 32): a consumer folds the slots of every channel in the prologue
// of every workgroup (csrc/bn_fold.h) -- 512 B of L2 reads per channel and workgroup at 32 double slots, 67 MB per launch of a
// 512-channel layer's fused transform, as much as the transform itself moves -- while the same-address atomic traffic the slots exist
// to spread falls with the channel count (a wide layer has few rows per channel: <= 32 adders per address at 8 slots).
// (Tried: 8 slots from 64 channels up.  bn_bwd_reduce_kernel went from 16.9 to 25.2 us per launch: its <= 2 048 workgroups then meet 256
// ways on each address of a 64-channel layer.  The slot count follows the channel count because the adders per address fall with it.)
__host__ __device__ inline int stat_slots_used(int C) { return C >= 512 ? 8 : (C >= 256 ? 16 : 32); }
// FORWARD statistics (round 3): the slots hold DOUBLES, double S[kStatSlots][2][C] = (sum y, sum y^2), followed by 64 uint32 arrival
// counters.  A producer lane accumulates SHIFTED sums in fp32 -- d = y - k with k a value of its own (the first y it saw for that
// channel), s = sum d, q = sum d^2 over its n values: q stays of the order of n var however large the mean is -- and converts ONCE,
// when it hands its partial over, to the unshifted pair in fp64 (stat_unshift: exact identities sum y = n k + s, sum y^2 = q + 2 k s +
// n k^2); partials are then folded and added to the slots in fp64, and the consumer forms var = S2 / M - (S1 / M)^2 in fp64, where the
// cancellation costs 2^-53 mean^2 / var instead of 2^-24 mean^2 / var (round 2: E[y^2] - mean^2 in fp32 lost the variance of a channel
// with mean 50 / std 0.1 entirely; PyTorch's batch_norm uses Welford sums).  The BACKWARD reductions (sum g, sum g xhat: no
// cancellation) keep a float layout in the same buffer (below), which is sized for the larger.
constexpr int kStatCounters = 64;
__host__ __device__ inline size_t stat_fwd_doubles(int C) { return (size_t)kStatSlots * 2 * C; }
__host__ __device__ inline unsigned* stat_fwd_counters(float* stats, int C) {
  return reinterpret_cast<unsigned*>(reinterpret_cast<double*>(stats) + stat_fwd_doubles(C));
}
// backward layout inside the same buffer: float R[kStatSlots][2][C], 64 counters right behind, and -- wide layers only, written by the
// finalize launch and left dirty -- the totals float tot[2][C] BEHIND the forward layout, where no producer ever adds
__host__ __device__ inline unsigned* stat_bwd_counters(float* red, int C) { return reinterpret_cast<unsigned*>(red + (size_t)kStatSlots * 2 * C); }
__host__ __device__ inline float* stat_bwd_totals(float* red, int C) { return red + (size_t)kStatSlots * 4 * C + kStatCounters; }
__host__ __device__ inline int stat_buffer_floats(int C) { return kStatSlots * 4 * C + kStatCounters + 2 * C; }
#if defined(__HIPCC__) || defined(HIFIHR_HOSTSIM)
// (n, k, s, q) of one lane and channel -> its unshifted (sum y, sum y^2) in fp64
__device__ __forceinline__ void stat_unshift(int n, float k, float s, float q, double& S1, double& S2) {
  const double kd = (double)k, sd = (double)s, nd = (double)n;
  S1 = nd * kd + sd;
  S2 = (double)q + 2.0 * kd * sd + nd * kd * kd;
}
__device__ __forceinline__ void stat_atomic_add(double* p, double v) { unsafeAtomicAdd(p, v); }      // global_atomic_add_f64
// The same for a thread that walks rows of 4 consecutive channels (the elementwise producers: Winograd output transforms, depthwise
// convolution): add() per value, then fold16(): the 16 row lanes tl of a 16 x 16 thread arrangement (tl = tid >> 4, cl = tid & 15) meet in
// LDS in fp64 and row lane 0 adds the 4 channels' (sum, sum of squares) to slot[0 .. 3] / slot[C .. C + 3].
struct Stat4 {
  float4 k, s, q;
  int n;
  __device__ __forceinline__ Stat4() : k(make_float4(0.f, 0.f, 0.f, 0.f)), s(k), q(k), n(0) {}
  // seed(): call with (any) one value before the first add() -- the callers do it at ONE fixed position of their tile, so the check is
  // paid once per 16 values, not per value (the per-value form cost the Winograd output transform 0.9 us per launch)
  __device__ __forceinline__ void seed(const float4& v) {
    if (n == 0) k = v;
  }
  __device__ __forceinline__ void add(const float4& v) {
    const float dx = v.x - k.x, dy = v.y - k.y, dz = v.z - k.z, dw = v.w - k.w;
    s.x += dx; s.y += dy; s.z += dz; s.w += dw;
    q.x = fmaf(dx, dx, q.x); q.y = fmaf(dy, dy, q.y); q.z = fmaf(dz, dz, q.z); q.w = fmaf(dw, dw, q.w);
    ++n;
  }
  // red: __shared__ double[2][16][16][4]; every thread of the workgroup calls it (one barrier inside)
  __device__ __forceinline__ void fold16(double (*red)[16][16][4], int tl, int cl, bool mine, double* slot, int C) const {
    double S1[4], S2[4];
    stat_unshift(n, k.x, s.x, q.x, S1[0], S2[0]); stat_unshift(n, k.y, s.y, q.y, S1[1], S2[1]);
    stat_unshift(n, k.z, s.z, q.z, S1[2], S2[2]); stat_unshift(n, k.w, s.w, q.w, S1[3], S2[3]);
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[0][tl][cl][e] = S1[e]; red[1][tl][cl][e] = S2[e]; }
    __syncthreads();
    if (tl == 0 && mine) {
      for (int r = 1; r < 16; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e) { S1[e] += red[0][r][cl][e]; S2[e] += red[1][r][cl][e]; }
#pragma unroll
      for (int e = 0; e < 4; ++e) { stat_atomic_add(slot + e, S1[e]); stat_atomic_add(slot + C + e, S2[e]); }
    }
  }
};
#endif
hipError_t launch_bn_stats(const float* x, long M, int C, float* stats, hipStream_t st);
hipError_t launch_bn_act_fwd(const float* x, float* stats, const float* gamma, const float* beta, const float* residual,
                             int relu, long M, int C, float eps, float momentum, float* y, float* save_mean, float* save_invstd,
                             float* running_mean, float* running_var, hipStream_t st);
hipError_t launch_bn_act_eval(const float* x, const float* running_mean, const float* running_var, const float* gamma, const float* beta,
                              const float* residual, int act, long M, int C, float eps, float* y, hipStream_t st);
hipError_t launch_bn_act_bwd(const float* dy, const float* y, const float* x, const float* save_mean, const float* save_invstd,
                             const float* gamma, const float* beta, int act, long M, int C, float* red, float* dx, float* dres,
                             float* dgamma_acc, float* dbeta_acc, hipStream_t st);
// MaxPool2d(3, 2, 1)(ReLU(BN(x))) fused (the ResNet stem), C <= 512
bool bn_relu_maxpool_supported(int N, int H, int W, int C);
hipError_t launch_bn_relu_maxpool_fwd(const float* x, float* stats, const float* gamma, const float* beta, int N, int H, int W, int C,
                                      float eps, float momentum, float* y, unsigned char* tap, float* save_mean, float* save_invstd,
                                      float* running_mean, float* running_var, hipStream_t st);
hipError_t launch_bn_relu_maxpool_bwd(const float* gy, const unsigned char* tap, const float* x, const float* save_mean,
                                      const float* save_invstd, const float* gamma, const float* beta, int N, int H, int W, int C, float* red,
                                      float* dx, float* dgamma_acc, float* dbeta_acc, hipStream_t st, const float* pooled = nullptr);
// ws (may be NULL): scratch of conv_wgrad_workspace_bytes(g) bytes (any contents) for the shapes whose weight gradient is summed from
// per-workgroup slabs (layer 1, stem); without it those shapes use library-owned scratch or, inside a stream capture, the atomics kernel
size_t conv_wgrad_workspace_bytes(const ConvGeom& g);
hipError_t launch_conv_wgrad(const ConvGeom& g, const float* x, const float* dy, float* dw, void* ws, size_t ws_bytes, hipStream_t st);
bool conv_is_gemm(const ConvGeom& g);                 // 1x1 / stride 1 / no padding on shapes the kernels of gemm.hip take
bool conv_wgrad_c3_supported(const ConvGeom& g);
hipError_t launch_conv_wgrad_c3(const ConvGeom& g, const float* x, const float* dy, float* dw3, void* ws, size_t ws_bytes, hipStream_t st);
hipError_t launch_weight_transpose(const float* w, float* wt, int K, int RS, int C, hipStream_t st);
hipError_t launch_image_to_nhwc4(const float* img, float* out, int B, int H, int W, int OH, int OW, int pt, int pl, int normalize,
                                 hipStream_t st);

struct SsimWindow {
  float g[11];      // normalised 1-D gaussian (sigma 1.5), as pytorch_ssim.gaussian builds it
};
int ssim_tile_edge();           // output tile edge of the SSIM kernels: one partial sum per tile (hifihr_ssim_partial_count)
hipError_t launch_ssim_fwd(const SsimWindow& win, const float* img1, const float* img2, int planes, int H, int W, float* partial,
                           float* dA, float* dB, float* dC, hipStream_t st);
hipError_t launch_ssim_bwd(const SsimWindow& win, const float* img1, const float* img2, const float* dA, const float* dB,
                           const float* dC, const float* gscale, int planes, int H, int W, float* gimg1, hipStream_t st, float out_scale = 1.0f);
hipError_t launch_ssim_finish(const float* partial, int count, float scale, float offset, float* out, hipStream_t st);

// depthwise convolution geometry: x[N][H][W][C] -> y[N][OH][OW][C], k x k taps, padding top/left = pt/pl
struct DwGeom {
  int N, H, W, C, OH, OW, K, stride, pt, pl;
  int vorder = 0;      // 1 (HIFIHR_DW_VORDER=1): pixel blocks numbered row-fastest -- the 16 pixel lanes of a workgroup are 16 consecutive ROWS of one
                       // 4-pixel column block, whose windows share K - 1 of K input rows in L1.  Measured per EfficientNet-b3 step (tools/time_dwconv.py):
                       // forward 940 -> 913 us, backward-data 750 -> 732, backward-weight 1 306 -> 1 506; single shapes +-25 % either way: off
};
// pre_* (all four or none): x is the raw output of the preceding convolution; swish(batch_norm(x)) is applied as it is loaded
hipError_t launch_dwconv_fwd(const DwGeom& g, const float* x, const float* w, float* y, float* stats, hipStream_t st,
                             const float* pre_mean = nullptr, const float* pre_invstd = nullptr, const float* pre_gamma = nullptr,
                             const float* pre_beta = nullptr);
hipError_t launch_dwconv_bwd_data(const DwGeom& g, const float* dy, const float* w, float* dx, hipStream_t st);
hipError_t launch_dwconv_bwd_weight(const DwGeom& g, const float* x, const float* dy, float* dw, hipStream_t st,
                                    const float* pre_mean = nullptr, const float* pre_invstd = nullptr, const float* pre_gamma = nullptr,
                                    const float* pre_beta = nullptr);
// the statistics half of a training batch-norm alone: slots -> (save_mean, save_invstd), running statistics, slots handed back zeroed
hipError_t launch_bn_finalize_fwd(float* stats, long M, int C, float eps, float momentum, float* save_mean, float* save_invstd,
                                  float* running_mean, float* running_var, hipStream_t st);

hipError_t launch_mmpool_fwd(const float* x, const float* p, int B, int HW, int C, float* y, int* argmax, float* xmax, float* xavg,
                             hipStream_t st);
hipError_t launch_mmpool_bwd(const float* gy, const float* p, const int* argmax, const float* xmax, const float* xavg, int B, int HW,
                             int C, float* dx, float* dp_acc, hipStream_t st);
// nn.MaxPool2d(k, s, p) for (k, s, p) in {(3, 2, 1), (3, 1, 1), (2, 2, 0)}
hipError_t launch_maxpool_flat(const float* x_or_gy, unsigned char* tap, int N, int H, int W, int C, int k, int s, int p, float* y_or_dx, int bwd,
                               hipStream_t st);
hipError_t launch_maxpool_fwd(const float* x, int N, int H, int W, int C, int k, int s, int p, float* y, unsigned char* tap,
                              hipStream_t st);
hipError_t launch_maxpool_bwd(const float* gy, const unsigned char* tap, const float* ymask, int N, int H, int W, int C, int k, int s, int p, float* dx,
                              hipStream_t st);
// g = dy * (y > 0), db_acc[c] += sum over rows of g   (conv + bias + ReLU backward; g may alias dy)
hipError_t launch_bias_relu_bwd(const float* dy, const float* y, long M, int C, float* g, float* db_acc, hipStream_t st);
hipError_t launch_light_split_fwd(const float* lights, int B, float* colors, float* dirs, hipStream_t st);
hipError_t launch_light_split_bwd(const float* lights, const float* gcolors, const float* gdirs, int B, float* glights, hipStream_t st);

// geometry loss terms (losses.hip): k = 0 joint_3d, 1 vert_3d, 2 edge_length, 3 mshape, 4 mpose
struct GeomLossArgs {
  const float *joints, *joints_gt, *verts, *verts_gt, *shape, *pose;
  const int *faces, *vf_off, *vf_idx;     // faces [F][3]; vertex -> incident (face * 4 + corner) CSR (backward only)
  int B, J, V, F, NS, NP, mse;            // mse = 1: F.mse_loss, 0: F.l1_loss for the joint / vertex terms
  float lambda[5];
};
hipError_t launch_geom_loss_fwd(const GeomLossArgs& a, float* partial, float* out, hipStream_t st);
hipError_t launch_geom_loss_bwd(const GeomLossArgs& a, const float* gout, float* gj, float* gv, float* gshape, float* gpose,
                                hipStream_t st);
int photo_loss_partial_floats();
hipError_t launch_photo_loss_fwd(const float* rgba, const float* imgs, const long long* seg, int B, int HW, float l_tex, float l_mrgb,
                                 float l_sil, float* re_img_m, float* mask_rgbs, float* partial, float* out, hipStream_t st);
hipError_t launch_photo_loss_bwd(const float* rgba, const float* re_img_m, const float* mask_rgbs, const float* g_re_img,
                                 const float* gout, const float* fwd_out, int B, int HW, float l_tex, float l_mrgb, float* grad_rgba,
                                 hipStream_t st);
hipError_t launch_sil_post(const float* rgba, const float* imgs, int B, int HW, float* re_sil, float* mask_rgbs, hipStream_t st);
constexpr int kLossTotalParts = 4;
struct LossTotalParts { const float* v[kLossTotalParts]; int n[kLossTotalParts]; };          // sum of v[i][0 .. n[i])
struct LossTotalGrads { float* g[kLossTotalParts]; int n[kLossTotalParts]; int len[kLossTotalParts]; };
hipError_t launch_loss_total_fwd(const LossTotalParts& p, float* total, hipStream_t st);
hipError_t launch_loss_total_bwd(const float* gtotal, const LossTotalGrads& q, hipStream_t st);

// small-batch fully connected layer (mlp.hip): y[B][O] = act(BN1d?(x[B][I] W[O][I]^T + b)); gamma == nullptr: no batch-norm
// sigmoid(z) for the swish / sigmoid activations of EfficientNet (batch-norm + swish passes, squeeze-excite gates): v_exp_f32 of z log2(e) and
// v_rcp_f32 -- ~4 instructions where expf + an IEEE division are ~30, and the swish passes of csrc/bn.hip were bound by exactly those
// (bn_bwd_reduce at 0.45 of a 5 TB/s stream on EfficientNet-b3's 78 batch-norms).  Error <= ~3 ulp of the result (1 ulp each from the
// exponential, the reciprocal and the argument's rounding at |z| < 16): 3e-7 relative, against the 2e-4 / 2e-3 the EfficientNet golden test
// holds.  The emulator build keeps the libm expression.
__device__ __forceinline__ float fast_sigmoid(float z) {
#if defined(HIFIHR_HOSTSIM)
  return 1.0f / (1.0f + expf(-z));
#else
  return __builtin_amdgcn_rcpf(1.0f + __expf(-z));
#endif
}
struct LinearArgs {
  const float *x, *W, *b;
  float *y, *z;                         // z[B][O]: pre-batch-norm output (batch-norm layers only)
  const float *gamma, *beta;
  float *save_mean, *save_invstd, *running_mean, *running_var;
  float eps, momentum;
  int B, I, O, act;                     // act 0 none, 1 ReLU, 2 swish (z = pre-activation kept), 3 sigmoid (no batch-norm with 2 / 3)
};
struct LinearGrads {
  const float* dy;
  float *dz, *dW_acc, *db_acc, *dgamma_acc, *dbeta_acc, *dx;    // *_acc accumulate (+=); dz scratch [B][O]; dx overwritten
};
constexpr int kMaxLinearGroup = 6;
struct LinearGroup {
  LinearArgs a[kMaxLinearGroup];
  int n;
};
struct LinearGradsGroup {
  LinearGrads g[kMaxLinearGroup];
};
hipError_t launch_linear_fwd_group(const LinearGroup& grp, hipStream_t st);
hipError_t launch_linear_bwd_group(const LinearGroup& grp, const LinearGradsGroup& gg, hipStream_t st);
hipError_t launch_linear_fwd(const LinearArgs& a, hipStream_t st);
hipError_t launch_linear_bwd(const LinearArgs& a, const LinearGrads& g, hipStream_t st);

// squeeze-and-excitation pieces (se.hip); the two fully connected layers in between are launch_linear_* (act 2 = swish, 3 = sigmoid)
hipError_t launch_se_pool(const float* x, int B, int HW, int C, float* mean_zeroed, hipStream_t st);
hipError_t launch_se_bwd_gate(const float* dy, const float* x, int B, int HW, int C, float* dgate_zeroed, hipStream_t st);
hipError_t launch_se_scale(const float* x, const float* gate, const float* add, float ascale, int B, int HW, int C, float* y, hipStream_t st);
hipError_t launch_drop_connect_add(const float* x, const float* skip, const float* u, float keep, int B, size_t per_sample, float* out,
                                   hipStream_t st);
// the two layers between pooling and scaling, fused (se.hip): W1[SQ][C], W2T[SQ][C] = the transposed W2[C][SQ]
bool se_mlp_supported(int C, int SQ);
hipError_t launch_se_mlp_fwd(float* mean_acc, const float* W1, const float* b1, const float* W2T, const float* b2, int B, int C, int SQ,
                             float* mean_out, float* z1, float* h1, float* gate, hipStream_t st);
hipError_t launch_se_mlp_bwd(float* dgate_acc, const float* gate, const float* z1, const float* h1, const float* mean, const float* W1,
                             const float* W2T, int B, int C, int SQ, float* dz2, float* dz1, float* dmean, float* dW1_acc, float* db1_acc,
                             float* dW2_acc, float* db2_acc, hipStream_t st);

// Winograd F(2x2, 3x3) glue (wino.hip)
hipError_t launch_wino_weight_transform(const float* w, float* U, int K, int C, int flip, hipStream_t st);
hipError_t launch_wino_input_transform(const float* x, float* V, float* Y /* or nullptr */, int N, int H, int W, int C, hipStream_t st);
// one per-step weight re-layout job (include/hifihr.h: hifihr_weight_prep); kind 0: [K][RS][C] -> [C][RS][K], 1: Winograd U[16][K][C],
// 2: Winograd U'[16][C][K] of the transposed, rotated 3x3 filter (backward-data)
struct PrepJob {
  const float* src;
  float* dst;
  int K, C, RS, kind;
};
hipError_t launch_weight_prep(const PrepJob* jobs, int njobs, int blocks_per_job, hipStream_t st);
hipError_t launch_freihand_augment(const uint32_t* img, const uint8_t* mask, const int* idx, const int* coef, int B, int H, int W,
                                   float* out_img, float* out_mask, hipStream_t st);
struct BatchStepOut {          // hifihr_freihand_batch_step: per-iteration terms of the training step, all optional
  int root_id;                 // joint the ground truth is made relative to (args.ROOT); < 0: root = 0
  float image_size;            // s of get_ndc_fx_fy_cx_cy
  float *oroot, *ojoints_rel, *overts_rel, *ocam;
};
hipError_t launch_freihand_batch(const uint32_t* img, const uint8_t* mask, const float* Ks, const float* joints, const float* verts,
                                 const float* scales, int J, int V, const int* packed, int B, int H, int W, float* out_img, float* out_mask,
                                 long long* out_segm, float* oKs, float* oPs, float* ojoints, float* overts, float* oj2d, float* oscales,
                                 long long* oidx, const BatchStepOut& step, hipStream_t st);
size_t ho3d_workspace_bytes(int B, int out_size);
hipError_t launch_ho3d_batch(const uint32_t* img, const uint8_t* mask, const float* Ks, const float* uv21, const float* xyz21, int FH, int FW,
                             const int* packed, int B, int out_size, void* ws, float* out_img, float* out_mask, float* out_K,
                             float* out_uv, float* out_xyz, hipStream_t st);
hipError_t launch_procrustes(const float* pred, const float* gt, int B, int N, float* aligned, float* err_sum, hipStream_t st);
hipError_t launch_wino_output_transform(const float* Mm, float* y, float* stats, const float* bias, int relu, int N, int H, int W, int K,
                                        hipStream_t st);
hipError_t launch_wino_dy_transform(const float* dy, float* Y, int N, int H, int W, int K, hipStream_t st);
hipError_t launch_wino_dw_transform(float* dU, float* dw, int K, int C, int clear, hipStream_t st);

// batched fp32 GEMM on the 16x16x4 f32 MFMA (gemm.hip): the 16 products of a Winograd layer
struct BgemmArgs {
  const float* A; const float* B; float* C;
  int M, N, K;                 // NT: C[M][N] = A[M][K] B[N][K]^T;  TN: C[M][N] = sum_t A[t][M] B[t][N], K = number of t rows
  int lda, ldb, ldc;
  long sa, sb, sc;             // element strides between the problems of a batch
  int batch, tiles_m, tiles_n;
  int splits, cps;             // TN: slabs of the t range, K chunks (32 rows) per slab
  long sc_split;               // element stride between slabs of C
  const float* zeros = nullptr;   // bgemm_nt_rows_kernel<1 / 2> (ragged N / K; convolution gather): >= 16 zero bytes that out-of-range operand segments are read from
  // bgemm_nt_rows_kernel<2>: A is an NHWC image x[n][ih][iw][cC] and row m of the product is the (r, s, c)-ordered patch of output pixel
  // m = (n, oh, ow), gathered by the loader waves (K = cR * cS * cC, cC % 32 == 0; B = the filter [N][cR][cS][cC])
  int cIH = 0, cIW = 0, cC = 0, cOH = 0, cOW = 0, cR = 0, cS = 0, cStride = 1, cPad = 0;
  // bgemm_tn_rows_kernel, XCD-coherent schedule (round 6; 0: one contiguous share per workgroup).  The tiles of ONE problem read the same two
  // operand panels; with contiguous shares the 16 workgroups of an XCD sit in 4-5 different problems at any time (9 MB of operands against a
  // 4 MB L2: every panel was fetched ~4 x).  Here the chip walks the problems in `co_rounds` rounds of 8 * co_r: in a round XCD x owns co_r
  // consecutive problems and its workgroups split THEIR blocks evenly (whole tiles), so an XCD's L2 holds the operands of co_r problems;
  // the problems left over behind the last full round are shared out as before.
  int co_r = 0, co_rounds = 0;
  int k_valid = 0;             // bgemm_tn_rows_kernel: rows t >= k_valid of every problem are ZERO in both operands (the padding behind the last
                               // tile mosaic of an F(4x4) layer: 450 real rows in 480) -- their k-steps are skipped; 0: every row counts
  float* stats = nullptr;      // bgemm_nt_rows_kernel, batch 1 (a 1x1 convolution in front of a batch-norm): per-column sum / sum of squares of C
                               // added into the slot buffer [kStatSlots][2][N] (csrc/bn.hip), or null
};
void bgemm_describe(int tn, int M, int N, int K, char* out, int cap);
void bgemm_describe_batch(int tn, int M, int N, int K, int batch, char* out, int cap);
bool bgemm_nt_supported(int M, int N, int K);
bool bgemm_tn_supported(int M, int N, int T);
size_t bgemm_nt_workspace_bytes(int M, int N, int K, int batch);
bool bgemm_nt_stats_supported(int N);      // launch_bgemm_nt(..., stats != null) is available for this N
// forward convolution as the row-share GEMM with the patch gather in its loader waves (csrc/gemm.hip, bgemm_nt_rows_kernel<2>)
bool conv_rows_supported(const ConvGeom& g, const float* bias);
hipError_t launch_conv_rows(const ConvGeom& g, const float* src, const float* wgt, float* dst, float* stats, const float* zeros, hipStream_t st);
bool conv_rows_pair_supported(const ConvGeom& g1, const ConvGeom& g2);
hipError_t launch_conv_rows_pair(const ConvGeom& g1, const float* src, const float* w1, float* y1, float* stats1, const ConvGeom& g2, const float* w2,
                                 float* y2, float* stats2, const float* zeros, hipStream_t st);
bool bgemm_nt_ragged_supported(int M, int N, int K);   // bgemm_nt_rows_kernel<true>: N % 4 == 0, K % 4 == 0, N not a multiple of 128 or K not of 32
hipError_t launch_bgemm_nt(const float* A, const float* B, float* C, int M, int N, int K, int batch, void* ws, size_t ws_bytes, hipStream_t st,
                           float* stats_or_null = nullptr, int M_alloc = 0);      // stats: only with N % 128 == 0 and batch == 1 (else hipErrorInvalidValue)
int bgemm_tn_parts(int M, int N, int T, int batch);
hipError_t launch_bgemm_tn(const float* A, const float* B, float* Cparts, int M, int N, int T, int batch, int parts, hipStream_t st, int T_valid = 0);
bool bgemm_nt_tn_pair_supported(int M, int M_alloc, int N, int K, int batch, int M2, int N2, int T2, int batch2, int parts2);
// an NT and an independent TN product on the row-share kernels in ONE launch (csrc/gemm.hip); hipErrorNotSupported: launch them separately
hipError_t launch_bgemm_nt_tn_pair(const float* A, const float* B, float* C, int M, int M_alloc, int N, int K, int batch, const float* A2,
                                   const float* B2, float* C2parts, int M2, int N2, int T2, int batch2, int parts2, hipStream_t st, int T2_valid = 0);
hipError_t launch_wino_dw_transform_parts(const float* dU_parts, int parts, float* dw, int K, int C, hipStream_t st);
// Winograd F(4x4, 3x3) glue (wino4.hip): 36 positions, T = wino4_tiles(N, H, W) tiles (N * ceil(H / 4) * ceil(W / 4), or fewer where 16
// images share a mosaic: wino4_math.h TileGeo)
long wino4_tiles(int N, int H, int W);
long wino4_tiles_real(int N, int H, int W);               // without the rounding of the mosaic form: the rows the forward / backward-data products compute
hipError_t launch_wino4_weight_transform(const float* w, float* U, int K, int C, int flip, hipStream_t st);
hipError_t launch_wino4_input_transform(const float* x, float* V, float* Y /* or nullptr */, int N, int H, int W, int C, hipStream_t st);
// csrc/wino4_bn.hip: batch-norm (+ residual) + ReLU applied on the fly inside the F(4x4, 3x3) input transform (C <= 512)
bool wino4_bn_supported(int C);
hipError_t launch_wino4_bn_input_transform(const float* x, float* stats, const float* gamma, const float* beta, const float* res, float* out,
                                           float* V, int N, int H, int W, int C, float eps, float momentum, float* save_mean,
                                           float* save_invstd, float* running_mean, float* running_var, hipStream_t st);
hipError_t launch_wino4_output_transform_bnred(const float* Mm, const float* x, const float* outp, const float* gadd, const float* save_mean,
                                               const float* save_invstd, const float* gamma, const float* beta, float* red, float* g, int N,
                                               int H, int W, int C, hipStream_t st);
hipError_t launch_wino4_bn_bwd_dual_transform(const float* g, const float* y, const float* save_mean, const float* save_invstd, const float* gamma,
                                              float* red, float* V, float* Y, int N, int H, int W, int K, float* dgamma_acc, float* dbeta_acc,
                                              hipStream_t st);
hipError_t launch_bn_bwd_apply(const float* g, const float* x, const float* save_mean, const float* save_invstd, const float* gamma, long M,
                               int C, float* red, float* dx, float* dgamma_acc, float* dbeta_acc, hipStream_t st);
hipError_t launch_wino4_output_transform(const float* Mm, float* y, float* stats, const float* bias, int relu, const float* mask, int N, int H, int W, int K,
                                         hipStream_t st);
hipError_t launch_wino4_dy_transform(const float* dy, float* Y, int N, int H, int W, int K, hipStream_t st);
hipError_t launch_wino4_dw_transform_parts(const float* dU_parts, int parts, float* dw, int K, int C, hipStream_t st);
struct WinoDwJob { const float* dU; float* dw; int parts, K, C; };       // = hifihr_wino_dw_job (include/hifihr.h)
hipError_t launch_wino4_dw_transform_multi(const WinoDwJob* jobs, int njobs, hipStream_t st);

hipError_t launch_texpca_fwd(const float* coef, const float* basis, const float* mean, int B, int K, long n, float* out, hipStream_t st);
hipError_t launch_texpca_bwd(const float* g, const float* basis, int B, int K, long n, float* dcoef_zeroed, hipStream_t st);
size_t adam_state_bytes();
hipError_t launch_adam_counted(float* p, const float* g, float* m, float* v, size_t n, float grad_scale, float eps, float weight_decay,
                               void* state, hipStream_t st);
hipError_t launch_adam(float* p, const float* g, float* m, float* v, size_t n, float grad_scale, float lr, float beta1,
                       float beta2, float eps, float weight_decay, int step, const float* dyn, hipStream_t st);

// joint_2d / bone_direc / bone_direc_3d of LossFunction (csrc/losses.hip): J = 21; either the 2-D or the 3-D pair may be NULL
hipError_t launch_joint_terms_fwd(const float* j2d, const float* j2d_gt, const float* j3d, const float* j3d_gt, int B, int J, int mse,
                                  const float* lam3, float* out3, hipStream_t st);
hipError_t launch_joint_terms_bwd(const float* j2d, const float* j2d_gt, const float* j3d, const float* j3d_gt, int B, int J, int mse,
                                  const float* lam3, const float* gout3, float* g_j2d, float* g_j3d, hipStream_t st);

}  // namespace hifihr
