/*
 * surf_hip.h -- C ABI of the MI355X (gfx950) kernels behind the SuRF volume-rendering hot path.
 *
 * The reference (prstrive/SuRF) has no FFI: its "operator API" is the Python class
 * models/surf.py:15 `SuRF(nn.Module)`.  These entry points are what a binding for that path
 * would call instead of the PyTorch ops listed beside each one (file:line into the reference).
 * INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *   - every `d_` / unprefixed data pointer is a DEVICE pointer owned by the caller; the library
 *     never allocates, frees or synchronises.  `h_` pointers are small HOST arrays read at launch.
 *   - `stream` is a hipStream_t passed as void* (0 = default stream).
 *   - return value: 0 ok, <0 invalid argument (SURF_E_*), >0 a hipError_t from the launch.
 *   - all floating point is fp32; index tables are int32 (the reference's int64 tables are
 *     converted by the host wrapper), -1 = empty voxel.
 *
 * Layouts
 *   - dense volumes (matching logits, index tables): [x][y][z], z fastest (volume.py:99-132).
 *   - sparse feature volumes: rows of 8 floats = 7 channels + 1 pad (surf.py:119 `out_feats[:,1:]`).
 *   - image / feature maps: NHWC with 4 floats per texel ("texel4"): (nv, H, W, 4).
 *   - per-sample arrays are ray-major: index = ray * S + sample.
 */
#ifndef SURF_HIP_H
#define SURF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SURF_MAX_VIEWS 8
#define SURF_MAX_STAGES 4
#define SURF_MAX_SAMPLES 256

#define SURF_E_ARG (-1)      /* null pointer / bad size */
#define SURF_E_LIMIT (-2)    /* exceeds SURF_MAX_* */

/* ABI version, bumped whenever a signature below changes.  Defined here once: surf_abi_version() returns it and the
 * host binding (surf_amd/_lib.py ABI_VERSION) refuses a library that reports a different number. */
#define SURF_ABI_VERSION 39
int surf_abi_version(void);

/* Repack NCHW fp32 (n, C<=4, H, W) into texel4 NHWC (n, H, W, 4), zero padding channels >= C. */
int surf_pack_texel4(const float* src, int n, int C, int H, int W, float* dst, void* stream);

/*
 * Ray set-up: z-sampling guided by the matching volume + section mid-points + voxel mask.
 * Replaces ImplicitSurface.render's sampling block (implicit_surface.py:268-311),
 * the head of render_core (implicit_surface.py:72-86) and lookup_volume (projector.py:392-420).
 *   rays_o, rays_d (R,3); near, far (R)
 *   mvol        dense matching volume (Dm^3)
 *   lin_depth   device copy of torch.linspace(0,1,n_depth); lin_samples: the n_stage linspaces
 *               torch.linspace(0,1,n_samples[s]) concatenated (S floats)
 *   h_n_samples[n_stage], h_sample_ranges[n_stage]   (confs/surf.conf:118-119)
 *   jitter      NULL (render.perturb = 0) or device (R, n_stage): the per-ray, per-stage `torch.rand([R,1]) - 0.5`
 *               draws of render.perturb > 0 (implicit_surface.py:274-277, 304-306)
 *   tables[n_stage] (fine -> coarse, as surf.py:159 passes them), h_dims[n_stage]
 * outputs (any of z_vals may be NULL): z_vals, mid_z, dists (R,S); pts (R*S,3); vmask (R*S) uint8
 */
int surf_ray_setup(const float* rays_o, const float* rays_d, const float* near, const float* far, int n_rays,
                   const float* mvol, int Dm, const float* lin_depth, int n_depth, const float* lin_samples,
                   const int* h_n_samples, const float* h_sample_ranges, int n_stage, const float* jitter,
                   float sample_dist, const int32_t* const* h_tables, const int* h_dims, int n_vol,
                   float* z_vals, float* mid_z, float* dists, float* pts, uint8_t* vmask, void* stream);

/*
 * Size in floats of the packed SDF-MLP weight buffer, and the packer.
 * Replaces the weight_norm re-parameterisation + layer loop of SDFNetworkSparse
 * (sdf_network.py:28-121) for the shipped architecture (d_hidden 128, 6 hidden layers,
 * skip_in [3], multires 4, 28 feature channels, scale 1).  `h_W[l]`/`h_b[l]` are HOST pointers to
 * the EFFECTIVE (weight-normed) row-major matrices lin0..lin6 and biases.
 */
int64_t surf_sdf_packed_floats(void);
int surf_sdf_pack_weights(const float* const* h_W, const float* const* h_b, float* h_packed);

/* Bytes of scratch the SDF kernel needs for a launch of n points (gradient variant only). */
int64_t surf_sdf_scratch_bytes(int64_t n_points);

/*
 * SDF MLP forward (+ analytic gradient) at n points with sparse trilinear feature gather.
 * Replaces lookup_sparse_volume/grid_sample_3d_sparse (projector.py:217-390),
 * SDFNetworkSparse.forward/sdf (sdf_network.py:95-124) and the first-order part of
 * SDFNetworkSparse.gradient (sdf_network.py:129-141).
 *   pts (.,3); mask (.) uint8 or NULL (NULL = all points active)
 *   idx: NULL -> points 0..n-1 are evaluated; else the n point indices to evaluate (e.g. surf_compact of the mask):
 *        inputs are gathered and outputs scattered through idx, so wavefront tiles hold active points only
 *   h_vols[n_vol]: (N_s,8) rows; h_tables[n_vol]: (D_s^3) int32; fine -> coarse
 *   packed: device copy of surf_sdf_pack_weights output
 *   sdf (n); grad (n,3) or NULL (forward only); scratch: >= surf_sdf_scratch_bytes(n) or NULL if grad NULL
 * Points with mask 0 are not written.
 */
int surf_sdf_mlp(const float* pts, const uint8_t* mask, const int32_t* idx, int64_t n, const float* const* h_vols,
                 const int32_t* const* h_tables, const int* h_dims, int n_vol, const float* packed,
                 float* sdf, float* grad, void* scratch, void* stream);

/*
 * The same evaluation on the 16-bit matrix pipes with fp32 accumulation (surf_amd/csrc/sdf_mlp_split.hip).  Arguments
 * as surf_sdf_mlp; `packed` is the output of the matching surf_sdf_pack_weights_* (…_packed_bytes() bytes, device copy),
 * `scratch` >= …_scratch_bytes(n).
 *   bf16x3: every fp32 operand is split exactly into three bf16 pieces and six partial products are accumulated
 *           (dropped terms <= 2^-23 per product): fp32-equivalent results.  The packed image is opaque: the kernel works in
 *           units of the softplus exponent (biases and input columns x 100 log2 e, lin6's row 0 x ln 2 / 100, hidden matrices
 *           as they are), so an image is only meaningful to the kernel of the same library build.
 *   f16x2:  two fp16 pieces per operand (22 significant bits, second piece scaled by 2^11), three partial products:
 *           operand error <= 2^-22; activations must stay below 65504 in magnitude.  The fast path.
 */
int64_t surf_sdf_bf16_packed_bytes(void);
int64_t surf_sdf_bf16_scratch_bytes(int64_t n_points);
int surf_sdf_pack_weights_bf16(const float* const* h_W, const float* const* h_b, unsigned char* h_packed);
int surf_sdf_mlp_bf16x3(const float* pts, const uint8_t* mask, const int32_t* idx, int64_t n, const float* const* h_vols,
                        const int32_t* const* h_tables, const int* h_dims, int n_vol, const void* packed, float* sdf,
                        float* grad, void* scratch, void* stream);
int64_t surf_sdf_f16_packed_bytes(void);
int64_t surf_sdf_f16_scratch_bytes(int64_t n_points);
int surf_sdf_pack_weights_f16(const float* const* h_W, const float* const* h_b, unsigned char* h_packed);
int surf_sdf_mlp_f16x2(const float* pts, const uint8_t* mask, const int32_t* idx, int64_t n, const float* const* h_vols,
                       const int32_t* const* h_tables, const int* h_dims, int n_vol, const void* packed, float* sdf,
                       float* grad, void* scratch, void* stream);

/* Both split kernels with the number of `idx` entries read from DEVICE memory (d_n[0] <= n_capacity = entries allocated for idx):
 * the launch that follows surf_compact needs no host round trip for the count ("replaced by device-side counters").
 * Grid and scratch are sized by the capacity; rows beyond d_n[0] are not touched. */
int surf_sdf_mlp_bf16x3_dn(const float* pts, const int32_t* idx, int64_t n_capacity, const int32_t* d_n, const float* const* h_vols,
                           const int32_t* const* h_tables, const int* h_dims, int n_vol, const void* packed, float* sdf,
                           float* grad, void* scratch, void* stream);
int surf_sdf_mlp_f16x2_dn(const float* pts, const int32_t* idx, int64_t n_capacity, const int32_t* d_n, const float* const* h_vols,
                          const int32_t* const* h_tables, const int* h_dims, int n_vol, const void* packed, float* sdf,
                          float* grad, void* scratch, void* stream);

/* The SDF on a lattice WITHOUT point tensors (extract_geometry, models/modules/implicit_surface.py:337-351, replaces its
 * meshgrid + cat of points): out[(ix ny + iy) nz + iz] = sign * sdf(ax[ix], ay[iy], az[iz]) by the forward-only split kernels;
 * ax / ay / az: device arrays of nx / ny / nz coordinates (torch.linspace's values, :338-340), sign = -1 gives marching cubes'
 * input u directly (:350).  nx ny nz < 2^31 per call (the caller walks slabs of x). */
int surf_sdf_lattice_bf16x3(const float* ax, const float* ay, const float* az, int nx, int ny, int nz, const float* const* h_vols,
                            const int32_t* const* h_tables, const int* h_dims, int n_vol, const void* packed, float* out,
                            float sign, void* stream);
int surf_sdf_lattice_f16x2(const float* ax, const float* ay, const float* az, int nx, int ny, int nz, const float* const* h_vols,
                           const int32_t* const* h_tables, const int* h_dims, int n_vol, const void* packed, float* out,
                           float sign, void* stream);

/*
 * Second-order term of the SDF network (training): grad (n,3) (optional) and smooth (n,3) = H.1, the row sums of the
 * Hessian of the SDF wrt the point.  Replaces the two chained torch.autograd.grad calls of SDFNetworkSparse.gradient
 * (models/modules/sdf_network.py:129-152; `smooth` feeds smooth_error, implicit_surface.py:172) by their closed form
 * in plain fp32.  idx (optional): compacted list of the n point indices to evaluate; other rows are left untouched.
 * surf_sdf_smooth_pack_weights takes the same host matrices as surf_sdf_pack_weights.
 */
int64_t surf_sdf_smooth_packed_floats(void);
int surf_sdf_smooth_pack_weights(const float* const* h_W, const float* const* h_b, float* h_packed);
int surf_sdf_smooth(const float* pts, const int32_t* idx, int64_t n, const float* const* h_vols,
                    const int32_t* const* h_tables, const int* h_dims, int n_vol, const float* packed, float* grad,
                    float* smooth, void* stream);

/*
 * Backward of the multi-view feature-consistency term (mfc_loss, losses/loss.py:43-45) w.r.t. the SDF: the patches depend
 * on the network only through the surface point p = o + d z0 (normal and feature maps are detached, implicit_surface.py:
 * 224-235), so d ncc / d z0 is the forward-mode tangent of (surface_patch_warp2 -> compute_LNCC2) along the ray.
 *   surf_patch_warp_tangent  surf_patch_warp's outputs plus their derivatives along dirs (R,3) = d pts / d z0
 *   surf_lncc_jvp            ncc (optional) and d ncc / d z0 (n_rays) from patches and tangents
 *   surf_crossing_backward   d_sdf (R,S) += g_z0 dz0/d sdf at the two samples bracketing the first sign change
 *                            (implicit_surface.py:181-220; nothing where there is no crossing or z0 left [0, *zmax])
 */
int surf_patch_warp_tangent(const float* pts, const float* dirs, const float* grads, int n_rays, const float* const* h_maps_t4,
                            int nv, int H, int W, const float* h_intrs, const float* h_kinv_ref, const float* h_c2w,
                            int patch_size, float* ref_out, float* src_out, float* ref_tan, float* src_tan, void* stream);
int surf_lncc_jvp(const float* ref, const float* src, const float* ref_tan, const float* src_tan, int64_t n_rays, int n_src,
                  int patch_elems, int channels, float* ncc, float* dncc, void* stream);
int surf_crossing_backward(const float* sdf, const uint8_t* vmask, const float* mid_z, int n_rays, int S, const float* zmax,
                           const float* g_z0, float* d_sdf, void* stream);

/*
 * Tall-skinny reduction for the weight / bias gradients of the backward kernels:
 *   out (M, N + with_sum) = (accumulate ? out : 0) + A[:, :M]^T [ X[:, :N] | 1 ]      A (rows, ldA), X (rows, ldX) row-major
 * M <= 128, N + with_sum <= 160.  workspace: surf_colgram_workspace_floats(rows, M, N) device floats.  Deterministic
 * (slab partials + a second pass, no atomics).
 */
int64_t surf_colgram_workspace_floats(int64_t rows, int M, int N);
int surf_colgram(const float* A, int ldA, int M, const float* X, int ldX, int N, int64_t rows, int with_sum, int accumulate,
                 float* workspace, float* out, void* stream);
/* The same with a precision policy: 0 = fp32-equivalent (surf_colgram: fp32 FMAs for rows < 4096, otherwise the matrix cores with an
 * exact three-way bf16 operand split and fp32 accumulation), 1 = operands rounded to ONE bf16 piece, fp32 accumulation (the
 * weight-gradient reductions of `train.precision = bf16`, BASELINE configs[3]). */
int surf_colgram_p(const float* A, int ldA, int M, const float* X, int ldX, int N, int64_t rows, int with_sum, int accumulate,
                   int precision, float* workspace, float* out, void* stream);

/*
 * Backward of the blending network w.r.t. its parameters for an upstream gradient of the per-sample colour (gcolor, indexed
 * like pts): the autograd of BlendingNetwork.forward (blending_network.py:69-118) under loss.backward() in closed form.
 * rows (n, nv-1, surf_blend_backward_row_floats()): per (sample, view) the INPUT vector and the pre-activation ADJOINT of
 * each of the 11 linear layers (column layout in blend_bwd.hip); dW = adj^T in, db = sum adj.  ds (n): per-sample d/d s.
 * color (n,3), optional: the recomputed forward colour.  raw_weights: DEVICE copy of the raw parameter buffer
 * (surf_blend_raw_floats(), state_dict order).  h_gfeats_t4 (may be NULL): four texel4 maps like h_feats_t4 that ACCUMULATE
 * the gradient of the sampled feature channels (generalisation training: the FPN's share of the colour loss).
 */
int surf_blend_backward_row_floats(void);
int surf_blend_backward(const float* pts, const int32_t* idx, int64_t n, const float* gcolor, const float* const* h_feats_t4,
                        const int* h_feat_hw, const float* imgs_t4, int nv, const float* h_intrs, const float* h_w2c,
                        const float* h_c2w, const float* raw_weights, float* rows, float* ds, float* color,
                        float* const* h_gfeats_t4, void* stream);

/*
 * Local normalised cross-correlation of the surface patches: out (n_rays) = mean of the two smallest per-view values of
 * mean_c clamp(1 - cov^2 / (var_ref var_src + 1e-5), 0, 2).  Replaces compute_LNCC2 (models/losses/ncc.py:7-51), the
 * multi-view feature-consistency term of Loss.forward (losses/loss.py:43-45).  ref (1, n_rays, P, C), src (n_src, n_rays, P, C)
 * as written by surf_patch_warp; n_src >= 2.
 */
int surf_lncc(const float* ref, const float* src, int64_t n_rays, int n_src, int patch_elems, int channels, float* out,
              void* stream);

/*
 * Backward of surf_lncc: the autograd of compute_LNCC2 (models/losses/ncc.py:7-51) under loss.backward() (runner.py:163).
 * g_out (n_rays): upstream gradient of the per-ray value; g_ref (1, n_rays, P, C) and g_src (n_src, n_rays, P, C): gradients
 * of the two patch stacks (fully written: zero for the views outside the two smallest of a ray).
 */
int surf_lncc_backward(const float* ref, const float* src, const float* g_out, int64_t n_rays, int n_src, int patch_elems,
                       int channels, float* g_ref, float* g_src, void* stream);

/*
 * Per-pixel terms of the photometric loss of one depth map (training).  Replaces compute_ptloss + SSIM
 * (models/losses/photometric_loss.py:54-125, :6-33): the other nv-1 views are warped into view ref_idx through `depth` (H,W)
 * (bilinear, zeros, align_corners=True) into `warp` (nv-1,H,W,4: rgb + validity); terms (H,W,8) =
 * [l1 m, grad_x mx, grad_y my, ssim m | m, mx, my, m], every loss value being the SUM of its topk smallest source views
 * and m / mx / my the reference mask and its products with the right / lower neighbour.  The loss is
 * sum(col0)/(sum(col4)+1e-8) + sum(col1)/(sum(col5)+1e-8) + sum(col2)/(sum(col6)+1e-8) + sum(col3)/(sum(col7)+1e-8)
 * (col7 repeats col4 so that the four quotients are one vector division).
 * imgs_t4 (nv,H,W,4) texel4; h_intrs / h_c2w / h_w2c: host (nv,4,4) intrinsics, camera-to-world and their inverses.
 */
int surf_ptloss_terms(const float* imgs_t4, int nv, int H, int W, const float* depth, const float* mask, int ref_idx, int topk,
                      const float* h_intrs, const float* h_c2w, const float* h_w2c, float* warp, float* terms, void* stream);

/*
 * Backward of the SDF network for upstream gradients of the SDF value (ybar, n) and of its spatial gradient (gbar, n x 3):
 * gbar . grad is the forward-mode tangent of the SDF along gbar, so one reverse sweep over a (value, tangent) forward sweep
 * replaces the reference's double backward (sdf_network.py:129-141 under loss.backward(), runner.py:163).  The kernel writes
 * per-sample buffers - in_v / in_d (7, n, 160): every layer's inputs and their tangents; tb / tdb (6, n, 128): the adjoints
 * of lin0..lin5's pre-activations and of their tangents - from which dW_l = tb_l^T in_v_l + tdb_l^T in_d_l, db_l = sum tb_l
 * (lin6 row 0: sum ybar in_v_6 + in_d_6), and accumulates the gradient of the sparse feature rows into h_dvols[s] (N_s, 8)
 * with float atomics (caller zero-fills; NULL = skip).  packed: surf_sdf_smooth_pack_weights' image.
 */
int surf_sdf_backward(const float* pts, const float* ybar, const float* gbar, int64_t n, const float* const* h_vols,
                      const int32_t* const* h_tables, const int* h_dims, int n_vol, float* const* h_dvols, const float* packed,
                      float* in_v, float* in_d, float* tb, float* tdb, void* stream);

/* Backward of the smooth (H.1) loss term through the SDF network (the autograd of sdf_network.py:143-150 under loss.backward(),
 * losses/loss.py:40): for sbar (n,3) = d loss / d smooth, the gradients of sum_n sbar_n . (H_n 1) = the mixed second directional
 * derivative of the SDF along (1,1,1) and sbar_n - one reverse sweep over a forward sweep with four streams (value, tangent
 * along 1, tangent along sbar, mixed).  Per-sample buffers like surf_sdf_backward's with the four streams stacked per layer:
 * in (7, 4, n, 160), ab (6, 4, n, 128); dW_l = ab[l]^T in[l] over the 4 n rows, db_l = column sums of ab[l][0]; lin6: dW row 0 =
 * column sums of in[6][3].  h_dvols: the sparse feature rows' gradients (N_s, 8), accumulated (may be NULL). */
int surf_sdf_smooth_backward(const float* pts, const float* sbar, int64_t n, const float* const* h_vols,
                             const int32_t* const* h_tables, const int* h_dims, int n_vol, float* const* h_dvols,
                             const float* packed, float* in, float* ab, void* stream);

/*
 * Multi-view feature fetch + blending MLP.
 * Replaces lookup_feature/compute_angle (projector.py:485-556) and BlendingNetwork.forward
 * (blending_network.py:69-118).
 *   h_feats[n_level]: texel4 maps (nv,H_i,W_i,4) fine -> coarse, h_hw[2*n_level] their sizes
 *   imgs: texel4 (nv,H,W,4) rgb
 *   h_intrs (nv,4,4), h_w2c (nv,4,4) = inverse(c2w), h_c2w (nv,4,4): HOST row-major
 *   blend_w: device copy of surf_blend_pack_weights output
 *   color (n,3); n_valid (n) uint8 = number of source views the point projects into
 * Only the shipped colour network (d_feature 16 = 4 pyramid levels x 4 channels) is supported.
 */
int surf_blend_raw_floats(void);     /* floats of the concatenated state_dict tensors, order below   */
int surf_blend_packed_floats(void);  /* floats of the MFMA-ordered buffer the kernel reads            */
/* h_raw: HOST concatenation of color_network.{s, ray_dir_fc.0.weight, ray_dir_fc.0.bias, ray_dir_fc.2.*,
 * base_fc.0.*, base_fc.2.*, vis_fc.0.*, vis_fc.2.*, vis_fc2.0.*, vis_fc2.2.*, rgb_fc.0.*, rgb_fc.2.*,
 * rgb_fc.4.*} (weight then bias, row-major), i.e. blending_network.py:34-64 in declaration order. */
int surf_blend_pack_weights(const float* h_raw, float* h_packed);
int surf_blend(const float* pts, const uint8_t* mask, const int32_t* idx, int64_t n, const float* const* h_feats, const int* h_hw,
               int n_level, const float* imgs, int nv, const float* h_intrs, const float* h_w2c,
               const float* h_c2w, const float* blend_w, float* color, uint8_t* n_valid, void* stream);

/*
 * The same blending MLP on the 16-bit MFMA pipes (blend_split.hip): every fp32 operand split into 16-bit pieces, fp32
 * accumulation.  precision: SURF_BLEND_BF16X3 (exact three-way bf16 split, six products: fp32-equivalent) or
 * SURF_BLEND_F16X2 (two fp16 pieces, 22-bit operands, three products).  `blend_w` is the device copy of
 * surf_blend_pack_weights_split's output for the same precision (the kernel's LDS image); other arguments as surf_blend.
 */
#define SURF_BLEND_BF16X3 1
#define SURF_BLEND_F16X2 2
#define SURF_BLEND_F32 3      /* the same LDS-resident kernel on v_mfma_f32_32x32x2_f32, no operand split */
int64_t surf_blend_split_packed_bytes(int precision);
int surf_blend_pack_weights_split(const float* h_raw, unsigned char* h_packed, int precision);
/* Bytes of device scratch a launch over n points with nv views needs (per-wavefront staging of the per-view inputs of the
 * second pass; stays in L2 / Infinity Cache).  Contents need not be preserved between launches. */
int64_t surf_blend_split_scratch_bytes(int64_t n_points, int nv);
int surf_blend_split(const float* pts, const uint8_t* mask, const int32_t* idx, int64_t n, const float* const* h_feats,
                     const int* h_hw, int n_level, const float* imgs, int nv, const float* h_intrs, const float* h_w2c,
                     const float* h_c2w, const void* blend_w, int precision, float* color, uint8_t* n_valid, void* scratch,
                     void* stream);
/* ... with the number of idx entries read from device memory, as surf_sdf_mlp_bf16x3_dn. */
int surf_blend_split_dn(const float* pts, const int32_t* idx, int64_t n_capacity, const int32_t* d_n, const float* const* h_feats,
                        const int* h_hw, int n_level, const float* imgs, int nv, const float* h_intrs, const float* h_w2c,
                        const float* h_c2w, const void* blend_w, int precision, float* color, uint8_t* n_valid, void* scratch,
                        void* stream);

/*
 * NeuS SDF -> alpha compositing, zero-crossing depth and per-ray reductions.
 * Replaces render_core's tail (implicit_surface.py:126-166,181-216) and validate's normal sum (:380-382).
 *   per-sample inputs (R,S): sdf, grad(3), color(3), n_valid, mid_z, dists, pts(3), vmask
 *   h_rot_ref: inverse(c2w[0][:3,:3]) row-major (9 floats, HOST)
 * outputs (R rows each; any may be NULL): color(3), render_depth, sdf_depth, normal(3) (camera frame),
 *   normal_val(3) = sum_k grad*w*inside (world frame, for validate), valid_mask u8, mid_inside u8,
 *   weights (R,S), inside (R,S), eik (R,2) = per-ray [sum relax*(|g|-1)^2, sum relax],
 *   z_sdf0 (R) = the zero crossing's ray parameter before the validity factor and the cosine (feeds surf_surface_points)
 */
int surf_composite(const float* sdf, const float* grad, const float* color, const uint8_t* n_valid,
                   const float* mid_z, const float* dists, const float* pts, const uint8_t* vmask,
                   const float* rays_d, int n_rays, int S, float inv_s, float cos_anneal_ratio,
                   const float* h_rot_ref, float* out_color, float* out_depth, float* out_sdf_depth,
                   float* out_normal, float* out_normal_val, uint8_t* out_valid, uint8_t* out_mid_inside,
                   float* out_weights, float* out_inside, float* out_eik, float* out_z_sdf0, void* stream);
/*
 * Backward of surf_composite for the outputs the training loss differentiates (colour_fine, render_depth, the eikonal
 * sums): the autograd of render_core's tail (implicit_surface.py:126-166) in closed form.  g_color (R,3), g_depth (R, may
 * be NULL): upstream gradients; eik_scale = dL/d gradient_error / (sum relax + 1e-5).  Outputs: d_sdf (R,S), d_grad (R,S,3),
 * d_color (R,S,3), d_inv_s (R) per-ray partial sums.  First kernel of the training row's backward side.
 */
int surf_composite_backward(const float* sdf, const float* grad, const float* color, const float* mid_z, const float* dists,
                            const float* pts, const uint8_t* vmask, const float* rays_d, int n_rays, int S, float inv_s,
                            float cos_anneal_ratio, const float* h_rot_ref, const float* g_color, const float* g_depth,
                            float eik_scale, float* d_sdf, float* d_grad, float* d_color, float* d_inv_s, void* stream);
/* The same with the upstream gradient of gradient_error left ON THE DEVICE: eik_upstream (1 float, may be NULL = 1) multiplies
 * eik_scale inside the kernel, so a caller holding dL/d gradient_error as a device scalar (autograd) passes
 * eik_scale = 1 / (sum relax + 1e-5) and never reads the scalar back (a synchronising read at the head of the backward sweep). */
int surf_composite_backward_s(const float* sdf, const float* grad, const float* color, const float* mid_z, const float* dists,
                              const float* pts, const uint8_t* vmask, const float* rays_d, int n_rays, int S, float inv_s,
                              float cos_anneal_ratio, const float* h_rot_ref, const float* g_color, const float* g_depth,
                              float eik_scale, const float* eik_upstream, float* d_sdf, float* d_grad, float* d_color,
                              float* d_inv_s, void* stream);


/* =====================================================================================================
 * Volume build (surf.py:80-131).  Voxel coordinates are int32 triples; voxel_size = 2/(D-1), origin -1.
 * ===================================================================================================== */

/*
 * Children + depth-band test.  Replaces Volume.up_sample (volume.py:35-52) and the test of
 * Volume.depth_filtering (volume.py:134-165).  Child o of parent p has coordinate 2*parent + pos_list[o]
 * (order of volume.py:42).  flags[8 p + o] = 1 iff |warped depth - voxel depth| < depth_range in > 1 views.
 *   parents (n_parents,3) on the D/2 lattice; D = child lattice side; depths (nv,H,W) previous-stage maps
 */
int surf_upsample_filter(const int32_t* parents, int64_t n_parents, int D, const float* depths, int nv, int H, int W,
                         const float* h_intrs, const float* h_w2c, float depth_range, uint8_t* flags, void* stream);

/*
 * Homography warp + softmax mean/variance cost volume.  Replaces Volume.back_proj_multiscale
 * (volume.py:54-97).  Voxel i = site i of the full D^3 lattice (parents == idx == NULL, n == D^3, x slowest,
 * volume.py:21-33) or child (idx[i] & 7) of parent (idx[i] >> 3).
 *   h_feats[4]: texel4 pyramids COARSE -> fine (the reference's `features` order), h_hw[8] their (H,W);
 *   levels stage..3 are summed; h_agg: agg_mlp as [0.weight(8x4) | 0.bias(8) | 2.weight(8) | 2.bias(1)] (HOST)
 * outputs: coords (n,3), feat (n,8) = [mean | var], keep (n) = (#views in frustum > 1)
 */
int surf_costvol(const int32_t* parents, const int32_t* idx, int64_t n, int D, const float* const* h_feats,
                 const int* h_hw, int stage, int nv, const float* h_intrs, const float* h_w2c, const float* h_agg,
                 int32_t* coords, float* feat, uint8_t* keep, void* stream);

/* Stable stream compaction: idx_out = ascending indices i with flags[i] != 0, *total = their number
 * (device int).  Replaces the boolean-mask indexing of volume.py:165-166 / surf.py:104-108.
 * workspace: surf_compact_workspace_ints(n) int32; idx_out must hold up to n entries. */
int64_t surf_compact_workspace_ints(int64_t n);
int surf_compact(const uint8_t* flags, int64_t n, int32_t* workspace, int32_t* idx_out, int32_t* total, void* stream);

/* dst[i, off:off+row_words] = src[idx[i] >> idx_shift, :]  (rows of 32-bit words; idx_shift 3 = row of the parent) */
int surf_gather_rows(const void* src, const int32_t* idx, int64_t n, int row_words, int idx_shift, int dst_stride_words,
                     int dst_offset_words, void* dst, void* stream);
/* out[i] = a[b[i]] */
int surf_compose_index(const int32_t* a, const int32_t* b, int64_t n, int32_t* out, void* stream);

/*
 * Dense matching volume + index table.  Replaces Volume.sparse2dense and Volume.get_index
 * (volume.py:99-121, 123-132): dense = x2 trilinear upsample (align_corners=False) of prev (D/2)^3, or zeros
 * if prev == NULL; dense[coords[i]] = rows[i*row_stride]; table = -1 then table[coords[i]] = i.
 */
int surf_densify(const int32_t* coords, const float* rows, int row_stride, int64_t n, int D, const float* prev,
                 float* dense, int32_t* table, void* stream);

/*
 * Matching field: per-view softmax-expected depth.  Replaces MatchingField.forward / depth_render
 * (matching_field.py:73-141, 18-71, perturb False).
 *   h_kinv (nv,3,3) = inverse(intrs)[:, :3, :3]; h_c2w (nv,4,4); h_rinv (nv,3,3) = inverse(c2w[:, :3, :3]);
 *   h_near_fars (nv,2) -- HOST.  lin_x/lin_y/lin_n: device copies of torch.linspace(0,W-1,w), (0,H-1,h), (0,1,n).
 *   pre_depths (nv,H,W) or NULL (stage 0); ratio_cur/ratio_prev = range_ratios[stage], [stage-1]
 * jitter (optional, device, (nv, h*w, 2)): the train-mode `torch.rand([batch, 1]) - 0.5` of every ray and band
 * (matching_field.py:33-35; zeros for the views rendered without it, :129-133)
 * outputs: depth_lr (nv,h,w) and depth_full (nv,H,W) = bilinear upsample (align_corners=False);
 * stats (nv,h,w,4) or NULL: per ray (max logit, softmax denominator, expected z, 0) for surf_matching_depth_backward
 */
int surf_matching_depth(const float* mvol, int D, int nv, const float* h_kinv, const float* h_c2w, const float* h_rinv,
                        const float* h_near_fars, int H, int W, int h, int w, const float* lin_x, const float* lin_y,
                        const float* lin_n, int n, const float* pre_depths, float ratio_cur, float ratio_prev,
                        const float* jitter, float* depth_lr, float* depth_full, float* stats, void* stream);

/* =====================================================================================================
 * Sparse 3D U-Net pieces (reg_network.py:38-88 over torchsparse 2.1.0 -- third party, PARITY UNPINNED).
 * Kernel (27, C_in, C_out), offsets enumerated x fastest / z slowest; rows of C floats, C in {8,16,32,64}.
 * ===================================================================================================== */

/*
 * One sparse convolution with fused BatchNorm(eval) + ReLU (+ skip add after the ReLU, reg_network.py:79-83).
 *   mode 0 submanifold (stride 1): out site c gathers input sites c + o            (in_table on the same lattice)
 *   mode 1 down (k3, stride 2):    out site q gathers input sites 2q + o          (in_table on the fine lattice)
 *   mode 2 up (transposed, s2):    out site c gathers input sites q with 2q + o = c (in_table on the coarse lattice)
 *   bn_scale = gamma / sqrt(running_var + eps), bn_shift = beta - running_mean * bn_scale (device, C_out), or both NULL
 */
int surf_spconv(const float* in, int cin, const int32_t* in_table, int D_in, const int32_t* out_coords, int64_t n_out,
                int mode, const float* weight, int cout, const float* bn_scale, const float* bn_shift, const float* skip,
                float* out, void* stream);
/* bf16 ROW STORAGE of the sparse U-Net under the bf16 training policy (round 6; conf key train_precision = bf16).  Activations
 * keep their fp32 rows (BatchNorm statistics, skip sums, the backward's streaming reads) and get a bf16 SHADOW (n, C) uint16 that
 * the gather side of the next convolution reads instead: surf_bn_relu_apply16 / surf_bn_relu_backward16 = the fp32 entry points
 * below with the shadow of their output written in the same pass (out16 / dx16 may be NULL), surf_rows_to_bf16 for rows that come
 * from elsewhere, surf_spconv_rows16 = surf_spconv (no BN epilogue) for the (16, 8) channel pair - the one where halving the row
 * bytes pays - gathering from the shadow; fp32 weights, accumulation and output.  Never used by inference. */
int surf_spconv_rows16(const uint16_t* in16, int cin, const int32_t* in_table, int D_in, const int32_t* out_coords, int64_t n_out,
                       int mode, const float* weight, int cout, float* out, void* stream);
int surf_rows_to_bf16(const float* x, int64_t n_floats, uint16_t* out16, void* stream);
int surf_bn_relu_apply16(const float* x, int64_t n, int channels, const float* scale, const float* shift, const float* skip,
                         float* out, uint16_t* out16, void* stream);
int surf_bn_relu_backward16(const float* x, const float* dy, int64_t n, int channels, const float* scale, const float* shift,
                            const float* mean, const float* invstd, int train, void* workspace, float* dgamma, float* dbeta,
                            float* dx, uint16_t* dx16, void* stream);
/*
 * The same convolution on the matrix cores for the wide layers (C_in, C_out in {16, 32, 64}): per-offset gather-GEMM,
 * fp32 operands split exactly into three bf16 pieces (fp32-equivalent results).  surf_spconv_packed_bytes returns 0 for a
 * channel pair without such a kernel (use surf_spconv); surf_spconv_pack_weights (a device kernel on `stream`) turns the
 * (27, C_in, C_out) kernel into the split operand image `packed` once per model.
 */
int64_t surf_spconv_packed_bytes(int cin, int cout);
int surf_spconv_pack_weights(const float* weight, int cin, int cout, void* packed, void* stream);
/* bf16_operands != 0 (training under train_precision = bf16, BASELINE configs[3]): both operands rounded to bf16 - the first
 * piece of the same image, the first piece of the gathered rows - one product per k-step with fp32 accumulation instead of
 * the six of the exact split. */
int surf_spconv_mfma(const float* in, int cin, const int32_t* in_table, int D_in, const int32_t* out_coords, int64_t n_out,
                     int mode, const void* packed, int cout, const float* bn_scale, const float* bn_shift,
                     const float* skip, float* out, int bf16_operands, void* stream);

/*
 * BatchNorm with BATCH statistics over the voxel rows x (n, C) (train mode of spnn.BatchNorm, reg_network.py:14-15,28-29;
 * C in {8, 16, 32, 64}): scale = gamma / sqrt(var + eps), shift = beta - mean scale (device, C floats each), fp64 reduction;
 * running_mean / running_var (may both be NULL) are updated in place as torch.nn.BatchNorm1d does (momentum, unbiased
 * variance); batch_stats (may be NULL): mean | 1/sqrt(var + eps) (2C floats), what the backward needs.  surf_bn_relu_apply: out = relu(x scale + shift) (+ skip).  workspace: surf_bn_workspace_bytes(C) device bytes.
 */
int64_t surf_bn_workspace_bytes(int channels);
int surf_bn_train_affine(const float* x, int64_t n, int channels, const float* gamma, const float* beta, float eps,
                         float momentum, float* running_mean, float* running_var, float* scale, float* shift, float* batch_stats,
                         void* workspace, void* stream);
int surf_bn_relu_apply(const float* x, int64_t n, int channels, const float* scale, const float* shift, const float* skip,
                       float* out, void* stream);
/* Backward of y = relu(x scale + shift) (+ skip; the skip's gradient is dy itself) with scale = gamma invstd, shift = beta -
 * mean scale: x the raw convolution output, scale / shift the forward's affine, mean / invstd (C each) the statistics behind it.
 * train != 0: batch statistics (dx = scale (zbar - mean(zbar) - xhat mean(zbar xhat))); 0: running statistics (dx = scale zbar).
 * Outputs dgamma, dbeta (C), dx (n, C).  workspace: surf_bn_workspace_bytes(C). */
int surf_bn_relu_backward(const float* x, const float* dy, int64_t n, int channels, const float* scale, const float* shift,
                          const float* mean, const float* invstd, int train, void* workspace, float* dgamma, float* dbeta,
                          float* dx, void* stream);
/* InstanceNorm + ReLU (+ skip) backward of an FPN layer (models/modules/feature_network.py: nn.InstanceNorm2d, no affine
 * parameters) for all N views in three launches: x, dy, dx (N, hw, C) NHWC; stats (N, C, 2) = mean | rstd as surf_inorm_relu
 * wrote them; workspace: surf_inorm_backward_workspace_bytes(N, C) device bytes.  (= surf_bn_relu_backward per view with
 * scale = rstd, shift = -mean rstd, which is what the FPN backward called N times per layer until round 5.) */
int64_t surf_inorm_backward_workspace_bytes(int N, int channels);
int surf_inorm_relu_backward(const float* x, const float* dy, int N, int64_t hw, int channels, const float* stats, void* workspace,
                             float* dx, void* stream);

/* ---- backward of the volume build (train mode; the autograd of surf.py:80-131 under loss.backward(), runner.py:163) ----
 * surf_matching_depth_backward: same geometry arguments as surf_matching_depth; g_full (nv,H,W) = d loss / d depth maps
 * of the two views that carry gradient, view0 and view1 (the reference view and src_idx, matching_field.py:129-133; view1 < 0
 * or == view0: one view); the other views' maps are ignored; g_lr (nv,h,w) scratch; dmvol (D,D,D) ACCUMULATED.
 * The bands are constants (pre_depths are detached, matching_field.py:104).  stats: what surf_matching_depth wrote for the
 * same arguments (same jitter), or NULL - then the softmax statistics are recomputed by a first walk over the samples.
 * surf_densify_backward: g_rows[i * row_stride] += g_dense[coords[i]]; g_prev (D/2)^3 (may be NULL) accumulates the
 * background's share through the transposed x2 trilinear upsample (sites of the index table pass nothing).
 * surf_scatter_rows_add: backward of surf_gather_rows (float rows, atomics).
 * surf_costvol_backward: for the kept voxels `coords` (n,3) of a stage and g (n,8) = [d mean | d var], accumulates the
 * gradients of the summed feature levels into h_gfeats[l] (texel4 maps like h_feats[l], l >= stage) and of agg_mlp
 * (w1 | b1 | w2 | b2, 49 floats, device) into g_agg.  workspace: surf_costvol_backward_workspace_floats(n, nv, H, W) device
 * floats, (H, W) = the finest feature level (the kernel sorts its (view, voxel) adds by image tile: 9 floats per pair - 1.1 to
 * 1.5 GB of transient memory at the finest DTU stage, ~6 M voxels x 5 views); an upper bound for every pyramid.
 * surf_costvol_backward_workspace_floats_for(n, nv, h_hw): the exact need for the pyramid h_hw (4 x (H, W), coarse to fine, as
 * passed to surf_costvol_backward): 4,096 floats when the tile-sorted form will not run (irregular pyramid such as the
 * Tanks&Temples shape, more than 16,384 (view, tile) buckets) and the direct scatter is used instead. */
int surf_matching_depth_backward(const float* mvol, int D, int nv, const float* h_kinv, const float* h_c2w, const float* h_rinv,
                                 const float* h_near_fars, int H, int W, int h, int w, const float* lin_x, const float* lin_y,
                                 const float* lin_n, int n, const float* pre_depths, float ratio_cur, float ratio_prev,
                                 const float* jitter, const float* g_full, int view0, int view1, float* g_lr, float* dmvol,
                                 const float* stats, void* stream);
int surf_densify_backward(const int32_t* coords, int64_t n, int D, const int32_t* table, const float* g_dense, int row_stride,
                          float* g_rows, float* g_prev, void* stream);
int surf_scatter_rows_add(const float* g_dst, const int32_t* idx, int64_t n, int row_words, int idx_shift, int dst_stride_words,
                          int dst_offset_words, float* g_src, void* stream);
int64_t surf_costvol_backward_workspace_floats(int64_t n, int nv, int H, int W);
int64_t surf_costvol_backward_workspace_floats_for(int64_t n, int nv, const int* h_hw);
int surf_costvol_backward(const int32_t* coords, const float* g, int64_t n, int D, const float* const* h_feats,
                          float* const* h_gfeats, const int* h_hw, int stage, int nv, const float* h_intrs, const float* h_w2c,
                          const float* h_agg, float* workspace, float* g_agg, void* stream);

/* Weight gradient of the FPN's 3x3 convolutions (train mode; the autograd of feature_network.py:6-25,57-75):
 * out[ky][kx][cb][cs] = sum_(n,ys,xs) big[n][ys S + ky - 1][xs S + kx - 1][cb] small[n][ys][xs][cs], zero padding, NHWC maps,
 * big (N, Hs S, Ws S, cb), small (N, Hs, Ws, cs).  Conv2d: big = layer input, small = d output -> [ky][kx][ci][co];
 * ConvTranspose2d (S = 2): big = d output, small = layer input -> [ky][kx][co][ci].  Input gradients run on the forward
 * kernels: stride 1 -> surf_conv3x3 with the flipped, transposed kernel; stride 2 -> surf_deconv3x3_s2; deconv -> stride-2
 * surf_conv3x3.  InstanceNorm + ReLU backward = surf_bn_relu_backward per view (scale = rstd, shift = -mean rstd). */
int64_t surf_conv3x3_wgrad_workspace_floats(int N, int Hs, int Ws, int cb, int cs);
int surf_conv3x3_wgrad(const float* big, const float* small, int N, int Hs, int Ws, int cb, int cs, int stride,
                       float* workspace, float* out, void* stream);
/* With a precision argument (as surf_conv3x3_p): pairs with max(cb, cs) >= 16 run on the matrix cores - a GEMM over the pixel
 * index with both operands read transposed, per-row-group partial sums in the workspace, summed in a fixed order. */
int surf_conv3x3_wgrad_p(const float* big, const float* small, int N, int Hs, int Ws, int cb, int cs, int stride,
                         float* workspace, float* out, int precision, void* stream);

/* Backward of surf_ptloss_terms w.r.t. the depth map (train mode; the autograd of losses/photometric_loss.py:54-125):
 * warp = the forward's warped images; coef (device, 4 floats) = upstream / (M_t + 1e-8) for the l1, gx, gy and ssim terms;
 * g_warp (ns,H,W,4) scratch (zeroed here); g_depth (H,W) written.  Only the topk sources selected by the forward receive
 * gradient (torch.topk's backward); validity masks and the SSIM mask pool are constants. */
int surf_ptloss_backward(const float* imgs_t4, int nv, int H, int W, const float* depth, const float* mask, int ref_idx, int topk,
                         const float* h_intrs, const float* h_c2w, const float* h_w2c, const float* warp, const float* coef,
                         float* g_warp, float* g_depth, void* stream);




/*
 * Weight gradient of surf_spconv (no BN): dW (27, C_in, C_out) += sum_i x[neighbour_k(i)] (x) dy[i]  (float atomics; the caller
 * zero-fills).  The INPUT gradient is surf_spconv itself on the swapped lattices: submanifold with mirrored offsets
 * (slice 26 - k), stride-2 down <-> transposed up, kernel slices transposed (surf_amd.ops.spconv_backward).
 */
int surf_spconv_wgrad(const float* x, int cin, const int32_t* in_table, int D_in, const int32_t* out_coords, int64_t n_out,
                      int mode, const float* dy, int cout, float* dW, void* stream);
/* The same weight gradient on the matrix cores (per offset the GEMM X_k^T dY over the sites: 64-site tiles, the 27 offsets dealt
 * over the four wavefronts, dy split once per tile; exact three-way bf16 split, six products: fp32-equivalent) for the channel
 * pairs where it beats the per-voxel kernels - surf_spconv_wgrad_mfma_supported returns 1 for those ((8,16), (16,32), (32,16),
 * (32,32)), 0 otherwise. */
int surf_spconv_wgrad_mfma_supported(int cin, int cout);
int surf_spconv_wgrad_mfma(const float* x, int cin, const int32_t* in_table, int D_in, const int32_t* out_coords, int64_t n_out,
                           int mode, const float* dy, int cout, float* dW, void* stream);

/* bbox (device int32[6]) = [min x, min y, min z, max x, max y, max z] of coords (n,3) */
int surf_coords_bbox(const int32_t* coords, int64_t n, int32_t* bbox, void* stream);
/* Output sites of a k3/s2 conv, marks[(D/2+1)^3] |= 1 (marks zeroed by the caller).  torchsparse 2.1 is absent, which of its
 * down-sampling rules the authors' checkpoint was trained with is unknown, so both are offered:
 *   SURF_DOWN_DILATE  q is an output site when 2q lies in the 3^3 window of an input voxel and inside the inputs'
 *                     bounding box (device int32[6] from surf_coords_bbox: no host round trip)        [default]
 *   SURF_DOWN_FLOOR   q = floor(c / 2) for every input voxel c (bbox unused, may be NULL) */
#define SURF_DOWN_DILATE 0
#define SURF_DOWN_FLOOR 1
int surf_mark_down_sites(const int32_t* coords, int64_t n, int D, const int32_t* bbox, uint8_t* marks, int rule, void* stream);
/* keys (ascending lattice site numbers, e.g. from surf_compact) -> coords (n,3) and table[key] = rank */
int surf_sites_from_keys(const int32_t* keys, int64_t n, int D, int32_t* coords, int32_t* table, void* stream);
/* table[coords[i]] = i (table pre-filled with -1 by the caller) */
int surf_table_from_coords(const int32_t* coords, int64_t n, int D, int32_t* table, void* stream);
/* out = in @ W^T for rows of 8 floats (out_lin, reg_network.py:67,86) */
int surf_row_linear8(const float* in, const float* weight, int64_t n, float* out, void* stream);

/* =====================================================================================================
 * FPN (feature_network.py:126-178).  Activations NHWC fp32, channels a multiple of 4 (the RGB input is the
 * texel4 image); weights repacked by the host to [ky][kx][c_in][c_out].
 * ===================================================================================================== */

/* 3x3 convolution, padding 1, stride 1 or 2, no bias (feature_network.py:15,151).  out (N, H/stride, W/stride, cout). */
int surf_conv3x3(const float* in, const float* weight, int N, int H, int W, int cin, int cout, int stride, float* out,
                 void* stream);
/* ConvTranspose2d(3x3, stride 2, padding 1, output_padding 1), no bias (feature_network.py:66,155).  out (N,2H,2W,cout). */
int surf_deconv3x3_s2(const float* in, const float* weight, int N, int H, int W, int cin, int cout, float* out, void* stream);
/* The same two with a precision argument.  Layers with cin >= 16 run on the matrix cores (csrc/fpn_mfma.hip, round 6):
 * precision 0 = fp32-equivalent (both operands split exactly into three bf16 pieces, six products, fp32 accumulate: what
 * surf_conv3x3 / surf_deconv3x3_s2 call), 1 = operands rounded to bf16, one product (the bf16 training policy, conf key
 * train_precision; never used by inference).  cin < 16: the fp32 VALU kernels whatever the precision.  weight: fp32
 * [ky][kx][cin][cout] as above - split on the fly, there is no packed weight image. */
int surf_conv3x3_p(const float* in, const float* weight, int N, int H, int W, int cin, int cout, int stride, float* out,
                   int precision, void* stream);
int surf_deconv3x3_s2_p(const float* in, const float* weight, int N, int H, int W, int cin, int cout, float* out, int precision,
                        void* stream);
/* In place: x = relu(InstanceNorm2d(x)) (+ skip)  (feature_network.py:16-17,21-24; skip add :170).
 * workspace: surf_inorm_workspace_doubles(N,H,W,C) doubles; stats (N,C,2) receives mean and 1/sqrt(var+1e-5).
 * C must be a multiple of 4 that divides 256 (4, 8, 16, 32, 64: the channel counts surf_conv3x3 / surf_deconv3x3_s2 are
 * instantiated for; since round 5 the statistics pass assigns 256 / C pixels to a workgroup row); any other C is SURF_E_ARG.
 * surf_inorm_relu_backward: C in {8, 16, 32, 64}. */
int64_t surf_inorm_workspace_doubles(int N, int H, int W, int C);
int surf_inorm_relu(float* x, int N, int H, int W, int C, const float* skip, double* workspace, float* stats, void* stream);
/* The same into `out` (N,H,W,C), x left as it is: a recording (train-mode) forward keeps the raw convolution output for
 * surf_inorm_relu_backward and needs no copy of it. */
int surf_inorm_relu_out(const float* x, int N, int H, int W, int C, const float* skip, double* workspace, float* stats,
                        float* out, void* stream);

/*
 * Surface patches for the LNCC loss (training outputs ref_gray_val / sampled_gray_val).  Replaces render_core's tail
 * (implicit_surface.py:217-245), surface_patch_warp2 (projector.py:560-627) and patch_homography (:630-645).
 *   surf_upsample_bilinear_t4  F.interpolate(bilinear, align_corners=False) of a texel4 map (n,h,w,4) -> (n,H,W,4)
 *   surf_surface_points        pts = rays_o + rays_d * z with z = z_sdf0 zeroed outside [0, max(z_vals)] (:217-220);
 *                              workspace: one device uint32
 *   surf_patch_warp            maps: the three finest FPN levels as texel4 (nv,H,W,4) at FULL resolution (levels 1, 2
 *                              upsampled); h_intrs / h_c2w (nv,4,4) HOST; h_kinv_ref = inverse(intrs)[0][:3,:3] (9 floats);
 *                              grads = raw SDF gradients at pts; ref_out (1,R,p*p,12), src_out (nv-1,R,p*p,12)
 */
int surf_upsample_bilinear_t4(const float* src, int n, int h, int w, int H, int W, float* dst, void* stream);
int surf_surface_points(const float* rays_o, const float* rays_d, const float* z_sdf0, int n_rays, const float* z_vals,
                        int64_t n_z, unsigned* workspace, float* pts, void* stream);
int surf_patch_warp(const float* pts, const float* grads, int n_rays, const float* const* h_maps, int nv, int H, int W,
                    const float* h_intrs, const float* h_kinv_ref, const float* h_c2w, int patch_size, float* ref_out,
                    float* src_out, void* stream);

/* =====================================================================================================
 * Small fused helpers of the training step (csrc/train_small.hip): chains of tiny torch ops as single launches.
 * ===================================================================================================== */

/* lookup_volume(pts, mask_volumes, 'nearest').any(-1) (models/modules/implicit_surface.py:175; the mask volume of a level is
 * 1 exactly where its index table is >= 0, volume.py:112-130): out[i] = 1 when the voxel nearest to pts[i] (grid_sample
 * 'nearest', align_corners=False: round-half-even of ((p + 1) D - 1) / 2, zeros outside) is occupied on ANY level.
 * pts (n,3) device; h_tables / h_dims: HOST arrays of `levels` device index tables (D^3 int32) and their D; out (n) bytes. */
int surf_occupied_any(const float* pts, int64_t n, const int32_t* const* h_tables, const int* h_dims, int levels, uint8_t* out,
                      void* stream);

/* sum(|pred - target| mask) / (sum(mask) + 1e-8) over n elements: the masked L1 of the depth terms of models/losses/loss.py
 * (:71-93).  mask_kind 0: mask = n floats; 1: n bytes (torch.bool); 2: mask = target > 0 (`mask` unused).  Sums in fp64, in a
 * fixed order.  workspace: surf_masked_l1_workspace_bytes() device bytes; counter: one device uint32 that is 0 on entry (the
 * kernel leaves it 0: allocate and clear it once per stream); out2[0] = the loss, out2[1] = 1 / (sum(mask) + 1e-8).
 * The backward: g_pred[i] = upstream[0] out2[1] sgn(pred - target) mask  (upstream: a device scalar). */
int64_t surf_masked_l1_workspace_bytes(void);
int surf_masked_l1(const float* pred, const float* target, const void* mask, int mask_kind, int64_t n, void* workspace,
                   unsigned* counter, float* out2, void* stream);
int surf_masked_l1_backward(const float* pred, const float* target, const void* mask, int mask_kind, int64_t n, const float* out2,
                            const float* upstream, float* g_pred, void* stream);

/* Backward of the weight-norm re-parameterisation W = g v / |v|_row (models/modules/sdf_network.py:88-89, nn.utils.weight_norm)
 * of `layers` (<= 8) linear layers in one launch: dg = <dW, v>_row / |v|, dv = (g / |v|) (dW - (dg / |v|) v).
 * HOST arrays of device pointers: v, dW, dv (rows[l] x cols[l]), g, dg (rows[l]). */
int surf_weight_norm_backward(int layers, const float* const* h_v, const float* const* h_g, const float* const* h_dW,
                              const int* h_rows, const int* h_cols, float* const* h_dv, float* const* h_dg, void* stream);

/*
 * Marching cubes on a (nx, ny, nz) fp32 lattice u[x][y][z] (z fastest).  Replaces mcubes.marching_cubes(u, isovalue)
 * (PyMCubes 0.1.4, called at models/modules/implicit_surface.py:353): classic 256-case table, `u <= isovalue` inside
 * test, one vertex per sign-changing lattice edge by linear interpolation in double precision, lattice-index units.
 * Call sequence (the host sizes the outputs between the steps):
 *   surf_mc_classify  -> flags (nx*ny*nz bytes): bits 0-2 sign change along x/y/z from the point, bits 3-5 triangles of
 *                        the cell whose origin it is
 *   surf_compact(flags) -> active (ascending lattice indices of the non-zero flags)
 *   surf_mc_count     -> workspace (surf_mc_workspace_ints(n_active) int32), totals[0] = vertices, totals[1] = triangles
 *   surf_mc_emit      -> vertices (totals[0], 3) double, triangles (totals[1], 3) int32 vertex ids; vbase: scratch of
 *                        nx*ny*nz int32 (first vertex id of every active point)
 * Vertex order: (owner lattice point, axis); triangle order: cell (x outermost) then table order.
 */
int surf_mc_classify(const float* u, int nx, int ny, int nz, double isovalue, uint8_t* flags, void* stream);
int64_t surf_mc_workspace_ints(int64_t n_active);
int surf_mc_count(const uint8_t* flags, const int32_t* active, int64_t n_active, int32_t* workspace, int32_t* totals, void* stream);
int surf_mc_emit(const float* u, int nx, int ny, int nz, double isovalue, const uint8_t* flags, const int32_t* active,
                 int64_t n_active, const int32_t* workspace, int32_t* vbase, double* vertices, int32_t* triangles, void* stream);

/*
 * First-hit face ids of a triangle mesh from one pinhole view, by z-buffer rasterisation (mesh cleaning of the
 * evaluation: the faces some mask-pixel ray hits first; utils/clean_mesh.py:37-108 uses trimesh + pyembree for it).
 *   vertices (nv,3) fp32, faces (nf,3) int32 on the device; h_K (9) and h_w2c (12) HOST row-major; (h, w) the image the
 *   intrinsics refer to, (Hup, Wup) the sample lattice torch.linspace(0, w-1, Wup) x torch.linspace(0, h-1, Hup)
 *   zbuf (Hup*Wup) uint64 pre-filled with ~0: on return (depth bits << 32 | face id) of the nearest covering face
 */
int surf_raster_first_hit(const float* vertices, const int32_t* faces, int64_t n_faces, const float* h_K, const float* h_w2c,
                          int h, int w, int Hup, int Wup, unsigned long long* zbuf, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SURF_HIP_H */
