/* sgrender.h -- C ABI of libsgrender.so: the MI355X (gfx950) implementation of the
 * spherical-Gaussian x microfacet render path of lzqsd/InverseRenderingOfIndoorScene.
 *
 * The reference has no native layer: this path is ~105 eager aten launches made from
 * Python.  The boundary a maintainer would bind is therefore the
 * reference's own Python call boundary, and each entry point below names the reference
 * call it replaces (file:line relative to the reference checkout).  INTEGRATION.md shows
 * the ctypes stub that goes on the reference side.
 *
 * Conventions
 *   - all tensors are fp32, contiguous, device (HBM) pointers; the layouts are the
 *     reference's NCHW-style layouts, spelled out per argument;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); every call
 *     only enqueues work on that stream: no allocation, no host synchronisation;
 *   - return value: 0 on success, otherwise a hipError_t (> 0) or one of the
 *     SGR_ERR_* codes (< 0); sgr_last_error() returns a thread-local description;
 *   - pointers marked "nullable" may be NULL to skip that output / input;
 *   - `premap`: 1 = `lamb` / `weight` are the light decoders' raw outputs and the entry point applies
 *     tan(pi/2 * 0.999 x) first (output2env.output2env, models.py:396-400); 0 = they are post-tan already
 *     (output2env.fromSGtoIm); 2 (backward entry points) = they are the post-tan values a forward call returned
 *     (`lamb_tan` / `weight_tan`), and the gradients are still those w.r.t. the RAW inputs -- the backward then
 *     applies the chain rule d tan = 0.999 pi/2 (1 + y^2) from y alone instead of re-evaluating 4K tangents per cell.
 *     3 (sgr_fused_fwd, sgr_fused_bwd_sg, sgr_fused_fwd_recon(_tan), sgr_fused_bwd_recon; where sgr_heads_prologue_supported
 *     says so) = `axis` / `lamb` / `weight` are the light decoders' LAST-CONVOLUTION outputs: the output activations of
 *     models.py:336-346 (1.01 tanh -> unit axis / clamp(0.5 (. + 1), 0, 1)) run as the
 *     kernels' prologue, then the tan pre-map; the backward entry points return the gradients w.r.t. those raw outputs
 *     (the heads' chain rule is their epilogue).  Same shapes; no intermediate tensors.
 *
 * Shapes:  bn images; K = SGNum lobes per cell (<= SGR_MAX_LOBES); env grid R x C
 *   (envRow x envCol == the renderingLayer ctor's imHeight x imWidth); J = eh*ew
 *   quadrature directions per cell; BRDF maps are imH x imW with imH/R == imW/C in {1, 2}
 *   (other ratios: pool first -- the host layer uses torch's adaptive_avg_pool2d, the reference's own op -- and pass R x C maps).
 */
#ifndef SGRENDER_H_
#define SGRENDER_H_

#ifdef __cplusplus
extern "C" {
#endif

#define SGR_ABI_VERSION 5
#define SGR_MAX_LOBES 32

#define SGR_OK 0
#define SGR_ERR_BAD_ARG (-1)        /* NULL required pointer, non-positive size */
#define SGR_ERR_UNSUPPORTED (-2)    /* K > SGR_MAX_LOBES, pooling ratio not in {1,2} */

int sgr_abi_version(void);
const char* sgr_last_error(void);

/* Direction table of output2env.__init__ (models.py:353-363) and renderingLayer.__init__
 * (models.py:437-452), device layout used by every kernel below (`dirs` argument):
 *   dirs[j*4 + {0,1,2,3}] = (l_x, l_y, l_z, omega_j),  j = e*ew + a,  padded with zero rows
 *   up to Jpad = sgr_dirs_padded(eh*ew) directions (a multiple of 32); followed by the same
 *   table in separable form (l_j = (s_e ca_a, s_e sa_a, c_e)):
 *   rows[e*8 + ..] = (s_e, c_e, omega_e, s_e^2, 2 s_e c_e, c_e^2, 0, 0)  for e < eh rounded up to even,
 *   cols (8*ew floats): (ca_a, sa_a) for a < ew/2, then at float offset ew (ca_a^2, 2 ca_a sa_a, sa_a^2, 0)
 *   for a < ew/2 (the second half row is the exact negation of the first), then at float offset 4*ew
 *   (ca_a, ca_a+1, sa_a, sa_a+1) per azimuth pair a = 0, 2, .. < ew/2 (operands of the packed-fp32 kernels).
 * Host-side helper: fills `out_host` (sgr_dirs_floats(eh, ew) floats of host memory). */
int sgr_dirs_padded(int J);
int sgr_dirs_floats(int eh, int ew);
int sgr_fill_direction_table(float* out_host, int eh, int ew);

/* View vectors of renderingLayer.__init__ (models.py:415-430): out_host[3*R*C]. */
int sgr_fill_view_vectors(float* out_host, int R, int C, float fov_deg, const float* camera_pos3);

/* output2env.output2env (models.py:391-404) when premap != 0, output2env.fromSGtoIm
 * (models.py:371-389) when premap == 0.
 *   axis   [bn,K,3,R,C]   lamb [bn,K,R,C]   weight [bn,3K,R,C] (channel k*3+rgb)
 *   env    [bn,3,R,C,eh,ew]  (out)
 *   lamb_tan [bn,K,R,C], weight_tan [bn,3K,R,C]  (out, nullable; post-tan values the
 *   reference returns alongside the env image) */
int sgr_sg_to_env_fwd(const float* axis, const float* lamb, const float* weight, const float* dirs,
                      float* env, float* lamb_tan, float* weight_tan,
                      int bn, int K, int R, int C, int eh, int ew, int premap, void* stream);

/* renderingLayer.forwardEnv (models.py:461-522).
 *   albedo, normal [bn,3,imH,imW]   rough [bn,1,imH,imW]   env [bn,3,R,C,eh,ew]
 *   view [3,R,C] (sgr_fill_view_vectors)   diffuse, spec [bn,3,R,C] (out) */
int sgr_render_env_fwd(const float* albedo, const float* normal, const float* rough, const float* env,
                       const float* dirs, const float* view, float* diffuse, float* spec,
                       int bn, int R, int C, int eh, int ew, int imH, int imW, float F0, void* stream);

/* Fused output2env.output2env + renderingLayer.forwardEnv (wrapperBRDFLight.py:177,194
 * back to back): one pass, the env image is written only if `env` is non-NULL. */
int sgr_fused_fwd(const float* albedo, const float* normal, const float* rough,
                  const float* axis, const float* lamb, const float* weight,
                  const float* dirs, const float* view,
                  float* env /* nullable */, float* diffuse, float* spec,
                  int bn, int K, int R, int C, int eh, int ew, int imH, int imW, float F0,
                  int premap, void* stream);

/* sgr_fused_fwd that also returns the post-tan sharpness / intensity (nullable; the values output2env.output2env
 * returns next to the env image, models.py:396-404) for the backward entry points' premap = 2 mode. */
int sgr_fused_fwd_tan(const float* albedo, const float* normal, const float* rough,
                      const float* axis, const float* lamb, const float* weight,
                      const float* dirs, const float* view,
                      float* env /* nullable */, float* lamb_tan /* nullable */, float* weight_tan /* nullable */,
                      float* diffuse, float* spec,
                      int bn, int K, int R, int C, int eh, int ew, int imH, int imW, float F0,
                      int premap, void* stream);

/* Backward of output2env.output2env / fromSGtoIm (torch.autograd of models.py:371-404).
 *   g_env [bn,3,R,C,eh,ew] in;  g_axis [bn,K,3,R,C], g_lamb [bn,K,R,C], g_weight [bn,3K,R,C] out.
 *   With premap == 1 the inputs are the raw decoder outputs and the gradients are w.r.t.
 *   those (chain rule through tan applied); premap == 2: the same gradients from post-tan inputs;
 *   with premap == 0 they are w.r.t. the post-tan lamb / weight (fromSGtoIm). */
int sgr_sg_to_env_bwd(const float* g_env, const float* axis, const float* lamb, const float* weight,
                      const float* dirs, float* g_axis, float* g_lamb, float* g_weight,
                      int bn, int K, int R, int C, int eh, int ew, int premap, void* stream);

/* Backward of the fused pass w.r.t. the SG parameters (what trainLight.py needs:
 * wrapperBRDFLight.py:194 detaches albedo, the BRDF nets are frozen, trainLight.py:121-144).
 *   g_env (nullable) is the cotangent of the env image coming from other consumers (the
 *   reconstruction loss, wrapperBRDFLight.py:179-188); the quadrature's own contribution
 *   to dL/dEnv is recomputed in-kernel from g_diffuse / g_spec and never written to HBM. */
int sgr_fused_bwd_sg(const float* g_env /* nullable */, const float* g_diffuse, const float* g_spec,
                     const float* albedo, const float* normal, const float* rough,
                     const float* axis, const float* lamb, const float* weight,
                     const float* dirs, const float* view,
                     float* g_axis, float* g_lamb, float* g_weight,
                     int bn, int K, int R, int C, int eh, int ew, int imH, int imW, float F0,
                     int premap, void* stream);

/* dL/dEnv of renderingLayer.forwardEnv alone (autograd of models.py:511-520):
 *   g_env[b,c,r,cc,j] = omega_j ndl_j (g_diffuse_c A_c/pi + g_spec_c spec_j)   (out, dense) */
int sgr_render_env_bwd_env(const float* g_diffuse, const float* g_spec,
                           const float* albedo, const float* normal, const float* rough,
                           const float* dirs, const float* view, float* g_env,
                           int bn, int R, int C, int eh, int ew, int imH, int imW, float F0, void* stream);

/* d/d{albedo, normal, rough} of renderingLayer.forwardEnv (autograd of models.py:461-522,
 * including the 2x2 pooling, the normal renormalisation and the local-frame construction).
 * The env image is either given (`env` non-NULL: the un-fused API) or re-evaluated from the
 * SG parameters (`env` NULL: the fused API; premap as in sgr_fused_fwd).
 *   g_albedo, g_normal [bn,3,imH,imW], g_rough [bn,1,imH,imW]  (out, fully written) */
int sgr_render_bwd_brdf(const float* g_diffuse, const float* g_spec,
                        const float* albedo, const float* normal, const float* rough,
                        const float* env /* nullable */,
                        const float* axis, const float* lamb, const float* weight /* nullable trio */,
                        const float* dirs, const float* view,
                        float* g_albedo, float* g_normal, float* g_rough,
                        int bn, int K, int R, int C, int eh, int ew, int imH, int imW, float F0,
                        int premap, void* stream);

/* ---- scale-invariant regressions and the render loss ------------------------------------- */

/* Floats of scratch the loss entry points need for a batch of bn images. */
int sgr_loss_workspace_floats(int bn);

/* Render loss of wrapperBRDFLight.py:170-171,192,197-207 for this rank's shard:
 *   im_small = avgpool(im), seg_small = avgpool(seg)              (:170-171)
 *   (kd, ks) = LSregressDiffSpec coefficients of (diffuse, spec, im_small)   (models.py:23-84)
 *   rendered = clamp(kd diffuse + ks spec, 0, 1)                  (:203)
 *   parts[0] = sum (rendered - im_small)^2 seg_small,  parts[1] = sum seg_small   (:192,205-207)
 * The caller forms renderErr = parts[0] / max(parts[1], 1e-5) / 3 (after summing parts over
 * ranks when the batch is sharded).  No host synchronisation.
 *   diffuse, spec [bn,3,R,C]   im [bn,3,imH,imW]   seg [bn,1,imH,imW]   (imH/R == imW/C in {1,2})
 *   out: im_small, rendered [bn,3,R,C]; seg_small [bn,1,R,C]; coef [bn,2]; parts [2] */
int sgr_render_loss_fwd(const float* diffuse, const float* spec, const float* im, const float* seg,
                        float* im_small, float* seg_small, float* rendered, float* coef, float* parts,
                        float* workspace, int bn, int R, int C, int imH, int imW, void* stream);

/* The same three passes, with the loss value formed by the third one when the batch is not sharded (no separate
 * sgr_loss_finalize launch):  *loss = parts[0] / max(parts[1], 1e-5) / divisor,  *scale = d loss / d parts[0]
 * (wrapperBRDFLight.py:192,205-207: divisor 3).  loss and scale may both be NULL (= sgr_render_loss_fwd: all-reduce
 * parts over the ranks, then sgr_loss_finalize).  Three launches; the batch totals are folded by the last workgroup of
 * the third pass to arrive, in a fixed order (bit-reproducible). */
int sgr_render_loss_fwd_total(const float* diffuse, const float* spec, const float* im, const float* seg,
                              float* im_small, float* seg_small, float* rendered, float* coef, float* parts,
                              float* loss /* [1], nullable */, float* scale /* [1], nullable */, float divisor,
                              float* workspace, int bn, int R, int C, int imH, int imW, void* stream);

/* ABI 5.  sgr_render_loss_fwd_total on one rank (loss / scale required) that also writes
 *   g_diffuse, g_spec [bn,3,R,C] = weight * d loss / d{diffuse, spec}
 * from its third pass -- bit-identical to sgr_render_loss_bwd_scaled(NULL, weight, scale, ...) without that fourth launch.  For callers
 * that form their gradients ahead of the backward call (the fused light objective: wrapperBRDFLight.py:192-207 + trainLight.py:237). */
int sgr_render_loss_fwd_total_grads(const float* diffuse, const float* spec, const float* im, const float* seg,
                                    float* im_small, float* seg_small, float* rendered, float* coef, float* parts,
                                    float* loss /* [1] */, float* scale /* [1] */, float divisor, float weight,
                                    float* g_diffuse, float* g_spec,
                                    float* workspace, int bn, int R, int C, int imH, int imW, void* stream);

/* d parts[0] / d{diffuse, spec} times *g_num (a device scalar); coefficients are constants as in
 * the reference (detached, models.py:54,76; wrapperBRDFLight.py:197-201). */
int sgr_render_loss_bwd(const float* g_num, const float* diffuse, const float* spec,
                        const float* im_small, const float* seg_small, const float* coef,
                        float* g_diffuse, float* g_spec, int bn, int R, int C, void* stream);

/* The two steps around the all-reduce of `parts` under batch sharding, kept on the device:
 *   sgr_loss_finalize:  *loss = parts[0] / max(parts[1], 1e-5) / divisor  (renderErr, wrapperBRDFLight.py:192,205-207:
 *                       divisor 3; reconstErr, :179-188: divisor 3 * envHeight * envWidth),  *scale = d loss / d parts[0]
 *                       (two separate device scalars: one is returned to the caller, the other kept for the backward pass);
 *   sgr_render_loss_bwd_scaled:  sgr_render_loss_bwd with *g_num = *g_loss * weight * *g_scale (g_scale = scale; a NULL device
 *                       scalar = 1; `weight` is a host float: the render weight of trainLight.py:237 travels as a kernel argument,
 *                       not through a cached device tensor -- ABI 4). */
int sgr_loss_finalize(const float* parts /* [2], rank-summed */, float* loss /* [1] */, float* scale /* [1] */, float divisor, void* stream);

int sgr_render_loss_bwd_scaled(const float* g_loss /* nullable */, float weight, const float* g_scale /* nullable */, const float* diffuse, const float* spec,
                               const float* im_small, const float* seg_small, const float* coef,
                               float* g_diffuse, float* g_spec, int bn, int R, int C, void* stream);

/* models.LSregress (models.py:7-21): coef[b] = clamp(<pred_b,gt_b> / max(<pred_b,pred_b>,1e-5), 1e-3, 1e3);
 * n elements per image. */
int sgr_lsregress_coef(const float* pred, const float* gt, float* coef, float* workspace,
                       int bn, long long n, void* stream);

/* models.LSregressDiffSpec (models.py:23-84): coef[b] = (c_im c_d, c_im c_s); n elements per image. */
int sgr_lsregress_diffspec_coef(const float* diffuse, const float* spec, const float* im, float* coef,
                                float* workspace, int bn, int n, void* stream);

/* utils.predToShading (utils.py:156-195): per-cell cosine-weighted irradiance of the
 * SG mixture, shading[b,c,r,cc] = max(sum_j env_c(l_j) cos(El_j) sin(El_j), 0) over an eh x ew hemisphere grid
 * (the reference uses 16 x 32).  Needs ew in {16, 32}, K <= 24 (SGR_ERR_UNSUPPORTED otherwise).
 *   axis [bn,K,3,R,C]  lamb [bn,K,R,C]  weight [bn,3K,R,C]  ->  shading [bn,3,R,C] */
int sgr_sg_shading(const float* axis, const float* lamb, const float* weight, const float* dirs, float* shading,
                   int bn, int K, int R, int C, int eh, int ew, int premap, void* stream);

/* ---- reconstruction loss ------------------------------------------------ */

int sgr_recon_workspace_floats(int bn, int R, int C);

/* Log-L2 env reconstruction loss of wrapperBRDFLight.py:172-188 for this rank's shard (two HBM streaming
 * passes over the predicted and the ground-truth env images):
 *   mask[b,p] = seg_small[b,p] * env_ind[b] * [mean_{c,j} env_gt > 0.001]                         (:172-174)
 *   coef[b]   = clamp(<env mask, env_gt mask> / max(<env mask, env mask>, 1e-5), 1e-3, 1e3)       (models.LSregress)
 *   parts[0]  = sum mask (log(coef env + offset) - log(env_gt + offset))^2,  parts[1] = sum mask  (:179,183-187)
 * The caller forms reconstErr = parts[0] / max(parts[1], 1e-5) / 3 / (eh*ew) (after summing parts over ranks).
 *   env, env_gt [bn,3,R,C,eh,ew]   seg_small [bn,1,R,C]   env_ind [bn]
 *   out: mask [bn,R*C], coef [bn], parts [2] */
int sgr_recon_loss_fwd(const float* env, const float* env_gt, const float* seg_small, const float* env_ind,
                       float* mask, float* coef, float* parts, float* workspace,
                       int bn, int R, int C, int eh, int ew, float offset, void* stream);

/* g_env = *g_num * d parts[0] / d env  (coef is a constant, models.py:13);  g_env [bn,3,R,C,eh,ew] out. */
int sgr_recon_loss_bwd(const float* g_num, const float* env, const float* env_gt, const float* mask,
                       const float* coef, float* g_env,
                       int bn, int R, int C, int eh, int ew, float offset, void* stream);

/* Whether premap = 3 (decoder heads as a prologue, see Conventions) is available for a configuration: envWidth 16 or 32,
 * 6 < SGNum <= 24, the default kernels (no SGR_*_MODE override). */
int sgr_heads_prologue_supported(int K, int R, int C, int eh, int ew);

/* ---- trainLight objective without the env image (fully fused) --------------
 * wrapperBRDFLight.py:172-207 in two heavy passes; neither the predicted env image (:177) nor its
 * cotangent is ever written.  Needs ew == 16 or 32 and K <= 24 (sgr_fused_recon_supported; SGR_ERR_UNSUPPORTED
 * otherwise -- use sgr_fused_fwd + sgr_recon_loss_* there).  Workspace is shared by the two calls. */
int sgr_fused_recon_supported(int K, int R, int C, int eh, int ew);
int sgr_fused_recon_workspace_floats(int bn, int R, int C);

/* Forward: sgr_fused_fwd without the env output, plus the statistics of sgr_recon_loss_fwd stage 0 taken on
 * the fly against env_gt:  diffuse, spec [bn,3,R,C];  mask [bn,R*C];  coef [bn] (LSregress scale, models.py:7-21);
 * parts (nullable) = (0, sum mask) for this rank's shard: what a sharded caller all-reduces before the backward pass. */
int sgr_fused_fwd_recon(const float* albedo, const float* normal, const float* rough, const float* axis,
                        const float* lamb, const float* weight, const float* dirs, const float* view,
                        const float* env_gt, const float* seg_small, const float* env_ind,
                        float* diffuse, float* spec, float* mask, float* coef, float* parts, float* workspace,
                        int bn, int K, int R, int C, int eh, int ew, int imH, int imW, float F0, int premap,
                        void* stream);

/* sgr_fused_fwd_recon that also returns the post-tan sharpness / intensity for sgr_fused_bwd_recon(premap = 2). */
int sgr_fused_fwd_recon_tan(const float* albedo, const float* normal, const float* rough, const float* axis,
                            const float* lamb, const float* weight, const float* dirs, const float* view,
                            const float* env_gt, const float* seg_small, const float* env_ind,
                            float* lamb_tan /* nullable */, float* weight_tan /* nullable */,
                            float* diffuse, float* spec, float* mask, float* coef, float* parts, float* workspace,
                            int bn, int K, int R, int C, int eh, int ew, int imH, int imW, float F0, int premap,
                            void* stream);

/* sgr_fused_fwd_recon_tan with the object mask at its own resolution: seg [bn,1,segH,segW], (segH, segW) = (R, C) or (2R, 2C);
 * the latter is pooled 2x2 by the kernel (wrapperBRDFLight.py:171: adaptive_avg_pool2d's sum order), no separate pooling pass. */
int sgr_fused_fwd_recon_seg(const float* albedo, const float* normal, const float* rough, const float* axis,
                            const float* lamb, const float* weight, const float* dirs, const float* view,
                            const float* env_gt, const float* seg, int segH, int segW, const float* env_ind,
                            float* lamb_tan /* nullable */, float* weight_tan /* nullable */,
                            float* diffuse, float* spec, float* mask, float* coef, float* parts /* nullable */, float* workspace,
                            int bn, int K, int R, int C, int eh, int ew, int imH, int imW, float F0, int premap,
                            void* stream);

/* ABI 5.  The forward half of the light objective on ONE rank (wrapperBRDFLight.py:167-207 up to the two loss values) in four launches:
 * sgr_fused_fwd_recon_seg's statistics kernel, then sgr_render_loss_fwd_total's three passes with (i) the per-image fold of the env
 * statistics (coef_env, the mask sums in recon_workspace) as an extra workgroup per image of the first pass and (ii), when g_diffuse /
 * g_spec are given, ren_weight * d renderErr / d{diffuse, spec} written by the third -- what sgr_fused_fwd_recon_seg +
 * sgr_render_loss_fwd_total + sgr_render_loss_bwd_scaled did in six.  im [bn,3,imH,imW], seg [bn,1,imH,imW] with (imH, imW) = (R, C) or
 * (2R, 2C); BRDF maps brdfH x brdfW as in sgr_fused_fwd; the outputs are those of the calls it replaces (render_err = *loss, scale_r =
 * *scale with divisor 3).  Follow with sgr_fused_bwd_recon_total on the same recon_workspace. */
int sgr_light_objective_fwd(const float* albedo, const float* normal, const float* rough, const float* axis, const float* lamb,
                            const float* weight, const float* dirs, const float* view, const float* env_gt, const float* im,
                            const float* seg, const float* env_ind, float* lamb_tan /* nullable */, float* weight_tan /* nullable */,
                            float* diffuse, float* spec, float* mask, float* coef_env, float* im_small, float* seg_small,
                            float* rendered, float* coef_ds, float* parts_r, float* render_err, float* scale_r, float ren_weight,
                            float* g_diffuse /* nullable pair */, float* g_spec, float* recon_workspace, float* loss_workspace,
                            int bn, int K, int R, int C, int eh, int ew, int imH, int imW, int brdfH, int brdfW, float F0, int premap,
                            void* stream);

/* Backward of  objective = (render terms, through g_diffuse / g_spec) + rec_weight * reconstErr,
 *   reconstErr = num / max(den, 1e-5) / 3 / (eh*ew),  num = sum mask (log(coef env + offset) - log(env_gt + offset))^2,
 * w.r.t. the SG parameters, with env recomputed in registers.  den = *den_global when given (the mask sum
 * all-reduced over ranks) else this shard's own.  Also returns parts = (num, local sum mask): the loss value
 * comes out of the same pass.  g_axis [bn,K,3,R,C]  g_lamb [bn,K,R,C]  g_weight [bn,3K,R,C].
 * ABI 4: the three gradient outputs may be NULL together (g_diffuse / g_spec are then ignored and may be NULL too): the pass
 * skips its gradient half -- no shading frame, no cotangents, no accumulators -- and only returns parts: the loss value for
 * forward-only callers of the objective (evaluation loops under torch.no_grad(), testLight.py). */
int sgr_fused_bwd_recon(const float* albedo, const float* normal, const float* rough, const float* axis,
                        const float* lamb, const float* weight, const float* dirs, const float* view,
                        const float* env_gt, const float* mask, const float* coef, const float* den_global,
                        const float* g_diffuse /* nullable with the gradients */, const float* g_spec,
                        float* g_axis /* nullable trio */, float* g_lamb, float* g_weight, float* parts, float* workspace,
                        int bn, int K, int R, int C, int eh, int ew, int imH, int imW, float F0, int premap,
                        float offset, float rec_weight, void* stream);

/* sgr_fused_bwd_recon for an unsharded batch (den = this shard's own mask sum), with the objective's scalar tail taken in the same
 * fold (no sgr_objective_finalize launch):  *recon_err = num / max(den, 1e-5) / (3 eh ew),
 * *objective = ren_weight * *render_err + rec_weight * *recon_err  (trainLight.py:237).  applied2 (nullable): slot 0 of
 * sgr_rescale_inplace_flip's pair is set to 1 (the gradients are those of the objective itself). */
int sgr_fused_bwd_recon_total(const float* albedo, const float* normal, const float* rough, const float* axis,
                              const float* lamb, const float* weight, const float* dirs, const float* view,
                              const float* env_gt, const float* mask, const float* coef,
                              const float* g_diffuse, const float* g_spec,
                              float* g_axis, float* g_lamb, float* g_weight, float* parts, float* workspace,
                              int bn, int K, int R, int C, int eh, int ew, int imH, int imW, float F0, int premap,
                              float offset, float rec_weight, const float* render_err, float ren_weight,
                              float* objective, float* recon_err, float* applied2, void* stream);

/* Tail of the light objective on the device: *recon_err = parts_e[0] / max(parts_e[1], 1e-5) / divisor_e (divisor 3 * eh * ew,
 * wrapperBRDFLight.py:179-188), *objective = ren_w * *render_err + rec_w * *recon_err (trainLight.py:237).  parts_e = (numerator,
 * env-mask sum), rank-summed by the caller when the batch is sharded. */
int sgr_objective_finalize(const float* render_err, const float* parts_e, float ren_w, float rec_w, float divisor_e,
                           float* objective, float* recon_err, void* stream);

/* Cotangent scaling for gradients produced ahead of the backward call: x[i][0..n[i]) *= *scale / *applied in
 * place (i < count <= 4; x, n are HOST arrays of device pointers / lengths), then *applied = *scale.  Skipped on the
 * device when the two are equal (the cotangent of a scalar objective is normally 1).  No host sync. */
int sgr_rescale_inplace(float* const* x, const long long* n, int count, const float* scale, float* applied, void* stream);

/* The same in ONE launch (1 <= count <= 4): `applied2` holds two slots, the factor currently applied in applied2[parity]; on
 * return the new one (*scale) is in applied2[1 - parity] -- the caller flips its parity after every call. */
int sgr_rescale_inplace_flip(float* const* x, const long long* n, int count, const float* scale, float* applied2, int parity,
                             void* stream);

/* ---- light-decoder output heads --------------------------------------------
 * The activations at the end of models.decoderLight.forward (models.py:336-346) for the three decoders
 * and, if `packed` is given, the cascade hand-off tensor envmapsPred of wrapperBRDFLight.py:167-168:
 *   axis   [bn,K,3,R,C] = a / max(|a|, 1e-6),  a = 1.01 tanh(x_axis)        (mode 0)
 *   lamb   [bn,K,R,C]   = clamp(0.5 (1.01 tanh(x_lamb) + 1), 0, 1)          (mode 1)
 *   weight [bn,3K,R,C]  = clamp(0.5 (1.01 tanh(x_weight) + 1), 0, 1)        (mode 2)
 *   packed [bn,7K,R,C]  = axis (3K channels) | lamb (K) | weight (3K)        (nullable)
 * x_* are the dconvFinal outputs ([bn,3K,R,C], [bn,K,R,C], [bn,3K,R,C]). */
int sgr_light_heads_fwd(const float* x_axis, const float* x_lamb, const float* x_weight,
                        float* axis, float* lamb, float* weight, float* packed,
                        int bn, int K, int R, int C, void* stream);

/* Cotangents may come through the separate outputs, the packed tensor, or both (each nullable; they add).
 * Clamp passes the cotangent on the closed interval, like torch; the norm clamp blocks it below 1e-6. */
int sgr_light_heads_bwd(const float* x_axis, const float* x_lamb, const float* x_weight,
                        const float* g_axis, const float* g_lamb, const float* g_weight, const float* g_packed,
                        float* gx_axis, float* gx_lamb, float* gx_weight,
                        int bn, int K, int R, int C, void* stream);

/* ---- glue either side of the path ---------------------------------------------- */

/* Floats of workspace for the two calls below. */
int sgr_glue_workspace_floats(int bn);

/* testReal.py:421-432: the global light / albedo scale after LSregressDiffSpec, kept on the device
 * (the reference reads four sums back with .item()):
 *   cDiff = sum(diffuse_scaled) / sum(diffuse),  cSpec = sum(spec_scaled) / sum(spec)      (sums over all n elements)
 *   cSpec < 1e-3 ?  cAlbedo = 1 / max(albedo), cLight = cDiff / cAlbedo
 *                :  cLight = cSpec, cAlbedo = clip(cDiff / cLight, 1e-3, 1 / max(albedo)), cLight = cDiff / cAlbedo
 * out4 = (cLight, cAlbedo, cDiff, cSpec).  n = elements of the four render images, n_albedo = elements of albedo. */
int sgr_light_albedo_scale(const float* diffuse_scaled, const float* diffuse, const float* spec_scaled, const float* spec,
                           const float* albedo, float* out4, float* workspace, long long n, long long n_albedo, void* stream);

/* wrapperBRDFLight.py:138-156: input of the light encoder from the BRDF predictions,
 *   out [bn,11,H,W] = cat( resize(im), resize(albedo / max(mean_b(albedo), 1e-10) / 3), 0.5 (resize(normal) + 1),
 *                          0.5 (resize(rough) + 1), resize(depth / max(mean_b(depth), 1e-10) / 3) )
 * with resize = F.interpolate(., [H,W], mode='bilinear') (align_corners=False), the means per image over all channels;
 * also writes the normalised albedo_norm [bn,3,h,w] and depth_norm [bn,1,h,w] the wrapper returns (:139-147).
 * im, albedo, normal [bn,3,h,w]; rough, depth [bn,1,h,w].  The reference uses H x W = 480 x 640. */
int sgr_light_input_fwd(const float* im, const float* albedo, const float* normal, const float* rough, const float* depth,
                        float* out, float* albedo_norm, float* depth_norm, float* workspace, int bn, int h, int w, int H, int W,
                        void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SGRENDER_H_ */
