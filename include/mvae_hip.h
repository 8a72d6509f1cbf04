/*
 * mvae_hip.h -- C ABI of libmvae_hip.so: the MI355X (gfx950) implementation of the per-batch hot path of
 * oskopek/mvae (ModelVAE.train_step and the Riemannian latent-space operators under it).
 *
 * The reference has no FFI: its "operator API" is a set of Python classes.  Each entry point
 * below therefore cites the reference Python interface it stands behind (file:line under /root/reference); the
 * Python host layer in mvae_amd/ binds these with ctypes and re-exposes them under the reference's class/method
 * names (see INTEGRATION.md for the binding a maintainer of the reference would add).
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer to fp32 unless stated otherwise;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); nothing synchronises, nothing allocates:
 *     every call is legal inside a HIP graph capture;
 *   - tensors are row-major, coordinates in the last dim, leading dims flattened to `rows`;
 *   - return value: 0 on success, MVAE_E_* (<0) for argument errors, a positive hipError_t for runtime errors;
 *     mvae_last_error() returns a message for the calling thread's last failure;
 *   - manifold `kind`: the latent component types of the reference's model-string grammar
 *     (mt/mvae/utils.py:30-38): e, h, s, p, d, u;
 *   - arithmetic: float32, the reference's formulas, guards and custom backward rules (LeakyClamp, Acosh, ...).  One
 *     deliberate difference: the derivative of the sphere's acos (spherical.py:104-116) at EXACTLY +-1 is capped at
 *     1/sqrt(1e-9) instead of infinite, so the backward pass stays finite where the reference's is NaN; no finite
 *     reference value is changed.
 */
#ifndef MVAE_HIP_H
#define MVAE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MVAE_ABI_VERSION 12

/* Manifold kinds = the letters of the model-string grammar (utils.py:30-38): e, h, s, p, d, u.
 * MVAE_PROJ_SPHERE: StereographicallyProjectedSphere (ops/spherical_projected.py).
 * MVAE_UNIVERSAL: Universal (ops/universal.py): its parameter is the CURVATURE K (component.py:225-242), passed
 * wherever the other kinds take their raw radius parameter; K < -1e-6 -> Poincare ball, K > 1e-6 -> projected sphere,
 * else Euclidean (universal.py:67-74), each at radius relu(1/sqrt|K|) (universal.py:30-32). */
enum {
  MVAE_EUCLIDEAN = 0,
  MVAE_HYPERBOLOID = 1,
  MVAE_SPHERE = 2,
  MVAE_POINCARE = 3,
  MVAE_PROJ_SPHERE = 4,
  MVAE_UNIVERSAL = 5
};

enum {
  MVAE_OK = 0,
  MVAE_E_BADARG = -1,     /* null pointer, negative size, unknown kind */
  MVAE_E_UNSUPPORTED = -2,/* shape outside what the kernels were built for (e.g. true_dim > MVAE_MAX_TRUE_DIM) */
  MVAE_E_ALIGN = -3,      /* a matrix whose row stride is not a multiple of 4 floats / base not 16-byte aligned */
  MVAE_E_SYSTEM = -4      /* an operating-system call failed (shared-memory flag page of the peer exchange) */
};

#define MVAE_MAX_TRUE_DIM 64
#define MVAE_MAX_COMPONENTS 64

int mvae_abi_version(void);
const char* mvae_last_error(void);

/* ------------------------------------------------------------------------------------------------------------------
 * Manifold primitives (forward).  Manifold interface: mt/mvae/ops/manifold.py:22-75.
 * `radius_param` points at ONE float: the raw nn.Parameter (_nradius / _pradius, component.py:123,142,160,179; for
 * MVAE_UNIVERSAL the `_curvature` parameter, component.py:232); the kernels apply radius = clamp(relu(p), 1e-8, 1e8)
 * themselves (manifold.py:73-75).  Ignored for MVAE_EUCLIDEAN.
 * `d` is the TRUE dimension; ambient A = d+1 for h/s, d for e/p/d/u.
 * Projected sphere: spherical_projected.py:31-88 (class), :140-196 (functions).
 * ------------------------------------------------------------------------------------------------------------------ */

/* Manifold.exp_map_mu0: x[rows,d] -> out[rows,A].   hyperbolics.py:28-29,114-121 | spherical.py:28-29,94-101 |
 * euclidean.py:78-79 | poincare.py:132-137 */
int mvae_exp_map_mu0(int kind, const float* x, float* out, int64_t rows, int d, const float* radius_param,
                     void* stream);

/* Manifold.inverse_exp_map_mu0: x[rows,A] -> out[rows,A].   hyperbolics.py:131-135 | spherical.py:112-116 |
 * euclidean.py:86-87 | poincare.py:148-149 */
int mvae_inverse_exp_map_mu0(int kind, const float* x, float* out, int64_t rows, int d, const float* radius_param,
                             void* stream);

/* Manifold.parallel_transport_mu0(x, dst): x[rows,A], dst[rows,A] -> out[rows,A].
 * hyperbolics.py:87-93 | spherical.py:74-77 | euclidean.py:66-67 | poincare.py:116-117 */
int mvae_parallel_transport_mu0(int kind, const float* x, const float* dst, float* out, int64_t rows, int d,
                                const float* radius_param, void* stream);

/* Manifold.inverse_parallel_transport_mu0(x, src).   hyperbolics.py:96-103 | spherical.py:80-83 | euclidean.py:70-71 |
 * poincare.py:120-121 */
int mvae_inverse_parallel_transport_mu0(int kind, const float* x, const float* src, float* out, int64_t rows, int d,
                                        const float* radius_param, void* stream);

/* Manifold.sample_projection_mu0(v, at_point) -> (z, (u, v)): v[rows,d]; at[at_rows,A] is broadcast over leading
 * sample dims (row r uses at[r % at_rows]); z[rows,A], u[rows,A].
 * hyperbolics.py:138-142 | spherical.py:119-123 | euclidean.py:90-93 | poincare.py:152-157 */
int mvae_sample_projection_mu0(int kind, const float* v, const float* at, float* z, float* u, int64_t rows,
                               int64_t at_rows, int d, const float* radius_param, void* stream);

/* Manifold.inverse_sample_projection_mu0(z, at_point) -> (u, v).   hyperbolics.py:145-148 | spherical.py:126-129 |
 * euclidean.py:96-99 | poincare.py:160-164 */
int mvae_inverse_sample_projection_mu0(int kind, const float* z, const float* at, float* u, float* v, int64_t rows,
                                       int64_t at_rows, int d, const float* radius_param, void* stream);

/* Manifold.logdet(mu, std, z, data): h/s take u = data[0] ([rows,A]); p/d/u take (mu, z) (poincare.py:55-89,
 * spherical_projected.py:56-88); e writes zeros.  out[rows].   hyperbolics.py:49-51,58-65 | spherical.py:49-51,58-67 */
int mvae_logdet(int kind, const float* u, const float* mu, const float* z, float* out, int64_t rows, int64_t at_rows,
                int d, const float* radius_param, void* stream);

/* Manifold exp / log maps at a GENERAL base point (the module-level functions the reference's op tests call directly,
 * tests/mvae/ops/test_hyperbolics.py:168-217): x[rows,A] a tangent vector at at[at_rows,A] (row r uses at[r % at_rows])
 * -> out[rows,A] on the manifold, and back.
 * exp_map: hyperbolics.py:106-111 | spherical.py:86-91 | euclidean.py:74-75 | poincare.py:124-129 |
 * spherical_projected.py:148-154.   inverse_exp_map: hyperbolics.py:124-128 | spherical.py:104-109 |
 * euclidean.py:82-83 | poincare.py:140-145 | spherical_projected.py:164-169. */
int mvae_exp_map(int kind, const float* x, const float* at, float* out, int64_t rows, int64_t at_rows, int d,
                 const float* radius_param, void* stream);
int mvae_inverse_exp_map(int kind, const float* x, const float* at, float* out, int64_t rows, int64_t at_rows, int d,
                         const float* radius_param, void* stream);

/* Geodesic distance between points x[rows,A] and y[y_rows,A] -> out[rows].
 * h: R*acosh(-<x,y>_L/R^2) with the guarded Acosh (tests/mvae/ops/test_hyperbolics.py:46-47; ops/common.py:76-94)
 * s: R*acos(clamp(<x,y>/R^2,-1,1))                 (tests/mvae/ops/test_spherical.py:45-48)
 * e: 2*|x-y|                                        (tests/mvae/ops/test_euclidean.py:41-42)
 * p: poincare_distance                              (ops/poincare.py:92-105)
 * d: spherical_projected_distance                   (ops/spherical_projected.py:90-97); with MVAE_DIST_GYRO
 *    spherical_projected_gyro_distance              (ops/spherical_projected.py:100-105)
 * u: by the sign of the curvature, as everywhere. */
enum { MVAE_DIST_GEODESIC = 0, MVAE_DIST_GYRO = 1 };
int mvae_geodesic_distance(int kind, int variant, const float* x, const float* y, float* out, int64_t rows,
                           int64_t y_rows, int d, const float* radius_param, void* stream);

/* Backward of any primitive above (what torch.autograd does for the reference, ops/manifold.py:22-60 being plain
 * differentiable torch code + the custom Functions of ops/common.py:28-94).  `op` = MVAE_OP_*; a, b, c3 are the
 * primitive's inputs in the order of its forward call (unused ones NULL):
 *   EXP0 a=x | LOG0 a=x | PT0/IPT0 a=x b=dst/src | SAMPLE a=v b=at | ISAMPLE a=z b=at | LOGDET a=u (h,s) or b=mu c3=z
 *   (p,d,u) | EXP/LOG a=x b=at | DIST/DIST_GYRO a=x b=y
 * g1 (and g2 for the two-output primitives SAMPLE: (z,u), ISAMPLE: (u,v); may be NULL) are the upstream gradients.
 * Outputs (any may be NULL): ga[rows,.], gb[rows,.] (PER ROW even when b was broadcast over at_rows < rows: the caller
 * sums the sample dim), gc[rows,.], gr[rows] = per-row terms of d/d(radius_param) (caller sums them in index order, so
 * the result is deterministic; for MVAE_UNIVERSAL this is d/dK). */
enum {
  MVAE_OP_EXP0 = 0, MVAE_OP_LOG0 = 1, MVAE_OP_PT0 = 2, MVAE_OP_IPT0 = 3, MVAE_OP_SAMPLE = 4, MVAE_OP_ISAMPLE = 5,
  MVAE_OP_LOGDET = 6, MVAE_OP_EXP = 7, MVAE_OP_LOG = 8, MVAE_OP_DIST = 9, MVAE_OP_DIST_GYRO = 10,
  /* auxiliary functions of mvae_manifold_aux below (a = x, b = y) */
  MVAE_OP_LPROD = 11, MVAE_OP_LNORM = 12, MVAE_OP_TO_BALL = 13, MVAE_OP_TO_AMBIENT = 14, MVAE_OP_LAMBDA = 15,
  MVAE_OP_MOBADD = 16,
  /* diagonal-normal pieces of mvae_normal_op below (a = value / eps / loc, b = loc, c3 = scale; for these BOTH b and c3
   * are broadcast over at_rows and gb, gc are per-row gradients) */
  MVAE_OP_NORMAL_LOGPROB = 17, MVAE_OP_NORMAL_RSAMPLE = 18, MVAE_OP_NORMAL_KL = 19
};
int mvae_primitive_backward(int op, int kind, const float* a, const float* b, const float* c3, const float* g1,
                            const float* g2, float* ga, float* gb, float* gc, float* gr, int64_t rows,
                            int64_t at_rows, int d, const float* radius_param, void* stream);

/* Small public helpers of the reference's ops modules, x[rows,.] (and y[rows,.]) -> out[rows,.]:
 *   MVAE_OP_LPROD      <x,y>_L (h: hyperbolics.py:72-78; other kinds: the plain dot product)         -> [rows,1]
 *   MVAE_OP_LNORM      sqrt(<x,x>_L) with the guarded sqrt (h: hyperbolics.py:81-84; else |x|_2)     -> [rows,1]
 *   MVAE_OP_TO_BALL    lorentz_to_poincare (h, hyperbolics.py:151-152) | spherical_to_projected (s,
 *                      spherical.py:132-133): [rows,d+1] -> [rows,d]
 *   MVAE_OP_TO_AMBIENT poincare_to_lorentz (p, poincare.py:167-170) | projected_to_spherical (d,
 *                      spherical_projected.py:191-196): [rows,d] -> [rows,d+1]   (d + 1 <= MVAE_MAX_TRUE_DIM)
 *   MVAE_OP_LAMBDA     conformal factor lambda_x (p: geoopt lambda_x; d: spherical_projected.py:124-129) -> [rows,1]
 *   MVAE_OP_MOBADD     Moebius addition x (+) y with c = 1/R^2 (p) or K = 1/R^2 (d, spherical_projected.py:107-113)
 * Differentiable through mvae_primitive_backward like the primitives above. */
int mvae_manifold_aux(int op, int kind, const float* x, const float* y, float* out, int64_t rows, int d,
                      const float* radius_param, void* stream);

/* The diagonal normal underneath WrappedNormal / EuclideanNormal (distributions/wrapped_normal.py:60,
 * wrapped_distributions.py:39-42; torch.distributions.Normal formulas), loc / scale [param_rows, d] broadcast over the
 * leading sample dims of a[rows, d]:
 *   MVAE_OP_NORMAL_LOGPROB  out[rows]    = sum_i log N(a_i; loc_i, scale_i)            (EuclideanNormal.log_prob)
 *   MVAE_OP_NORMAL_RSAMPLE  out[rows, d] = loc + a * scale                             (Normal.rsample with eps = a)
 *   MVAE_OP_NORMAL_KL       out[rows]    = KL(N(a, scale) || N(0, 1)) summed over d    (sampling_procedures.py:153-155;
 *                                          loc unused, param_rows == rows)
 * Differentiable through mvae_primitive_backward. */
int mvae_normal_op(int op, const float* a, const float* loc, const float* scale, float* out, int64_t rows,
                   int64_t param_rows, int d, void* stream);

/* The reference's guarded scalar functions and their custom derivative rules (ops/common.py:28-147: LeakyClamp, Atanh,
 * Acosh, cosh, sinh, sqrt, logsinh, logcosh) plus the short float32 elementary functions the kernels substitute for
 * libm (csrc/mvae_fastmath.hpp), evaluated by the DEVICE code of the manifold kernels: y[i] = f(x[i]) through the float
 * path, dy[i] = f'(x[i]) through the dual-number rule (dy may be NULL).  lo/hi: bounds of MVAE_FN_CLAMP.
 * The *_PAIR ids return the two values the manifolds compute together: y = cosh|cos, dy = sinh|sin. */
enum {
  MVAE_FN_CLAMP = 0, MVAE_FN_ATANH, MVAE_FN_ACOSH, MVAE_FN_COSH, MVAE_FN_SINH, MVAE_FN_SQRT, MVAE_FN_LOGSINH,
  MVAE_FN_LOGCOSH, MVAE_FN_COSH_SINH_PAIR, MVAE_FN_COS_SIN_PAIR, MVAE_FN_SOFTPLUS, MVAE_FN_ACOS, MVAE_FN_TAN,
  MVAE_FN_LOG1P_POS, MVAE_FN_EXP, MVAE_FN_LOG, MVAE_FN_STD /* softplus(x) + 1e-5, component.py:72 */, MVAE_FN_COUNT
};
int mvae_scalar_fn(int fn, const float* x, float* y, float* dy, int64_t n, float lo, float hi, void* stream);
/* out[i] = a[i] * b[i]: the chain-rule product g * f'(x) of the scalar functions' backward (out may alias a or b). */
int mvae_mul(const float* a, const float* b, float* out, int64_t n, void* stream);
/* out[r][j] = g[r][j] * s[r]: the backward of a per-row loss (sum_j BCE) given its upstream gradient s[rows]. */
int mvae_scale_rows(const float* g, const float* s, float* out, int64_t rows, int D, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * One latent component end to end:  Component.encode (component.py:63-75, given the two Linear-head outputs) ->
 * SamplingProcedure.reparametrize (sampling_procedures.py:93-99 | 147-151) -> q_z.rsample_with_parts
 * (wrapped_normal.py:70-78) -> SamplingProcedure.kl_loss (sampling_procedures.py:101-116 | 153-155).
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct mvae_component_desc {
  int32_t kind;       /* MVAE_* */
  int32_t true_dim;   /* d */
  int32_t mean_col;   /* first column of this component's fc_mean output inside a `heads` row   */
  int32_t logvar_col; /* first column of its fc_logvar output inside a `heads` row               */
  int32_t logvar_dim; /* d, or 1 under --scalar_parametrization (component.py:54-57)             */
  int32_t eps_col;    /* first column inside an `eps` row (eps rows are [sum true_dim])          */
  int32_t z_col;      /* first column inside a `concat_z` row (rows are [sum ambient dim])       */
  int32_t radius_idx; /* index into `radii` (one raw radius parameter per component; unused: e)  */
} mvae_component_desc;

/* heads[head_rows, heads_ld], eps[rows, eps_ld] (rows = n_samples * head_rows; row r uses heads[r % head_rows]),
 * radii[ncomp] raw parameters.  Outputs: z[rows, z_ld]; kl[ncomp, rows] (may be NULL); log_q / log_p [ncomp, rows]
 * (may be NULL: the importance-sampling path of ModelVAE.log_likelihood, vae.py:82-123, via rsample_log_probs
 * sampling_procedures.py:46-50,106-110); mu[head_rows, z_ld] and std[head_rows, eps_ld] (may be NULL; q_z.loc and
 * q_z.scale as the reference exposes them). */
int mvae_component_forward(const mvae_component_desc* comps, int ncomp, const float* heads, int heads_ld,
                           const float* eps, int eps_ld, const float* radii, float* z, int z_ld, float* kl,
                           float* log_q, float* log_p, float* mu, float* std, int64_t rows, int64_t head_rows,
                           void* stream);

/* Backward of the above for the training path (rows == head_rows): given dz[rows, z_ld] and dkl[ncomp, rows]
 * (NULL = the scalar `dkl_scalar` for every entry), writes dheads[rows, heads_ld] and dradii[ncomp] (may be NULL).
 * dradii is a fixed-order sum of per-row terms staged in `workspace`
 * (mvae_component_backward_workspace_floats(ncomp, rows) floats): no atomics, bit-reproducible.
 * Gradient rules include the reference's non-standard ones (common.py:28-94). */
int64_t mvae_component_backward_workspace_floats(int ncomp, int64_t rows);
int mvae_component_backward(const mvae_component_desc* comps, int ncomp, const float* heads, int heads_ld,
                            const float* eps, int eps_ld, const float* radii, const float* dz, int z_ld,
                            const float* dkl, float dkl_scalar, float* dheads, float* dradii, float* workspace,
                            int64_t rows, void* stream);
/* The two operators above with every intermediate of the latent chain in FLOAT64 (the reference CLI's default numerics,
 * run.py:77,98-101) between float32 tensors: heads / eps / radii in, z / kl / log-probabilities / gradients out are float32,
 * softplus, exp map, transport, log map, log-det, log-probabilities and their derivatives are evaluated in double (libm).  Same
 * arguments and error behaviour; true dimensions <= 8 (MVAE_E_UNSUPPORTED above).  In float32 the sphere's <mu, z> / R^2 and
 * the hyperboloid's Lorentz product lose 3-4 digits at the warm-up radii and acos' needs a cap at |x| = 1; here neither. */
int mvae_component_forward_f64(const mvae_component_desc* comps, int ncomp, const float* heads, int heads_ld, const float* eps,
                               int eps_ld, const float* radii, float* z, int z_ld, float* kl, float* log_q, float* log_p,
                               float* mu, float* sd, int64_t rows, int64_t head_rows, void* stream);
int mvae_component_backward_f64(const mvae_component_desc* comps, int ncomp, const float* heads, int heads_ld, const float* eps,
                                int eps_ld, const float* radii, const float* dz, int z_ld, const float* dkl, float dkl_scalar,
                                float* dheads, float* dradii, float* workspace, int64_t rows, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Dense layers (torch.nn.Linear semantics: y = x W^T + b, W is [N, K]).  FeedForwardVAE.encode / decode,
 * ffnn_vae.py:42-60; Component heads, component.py:52-57.
 * ------------------------------------------------------------------------------------------------------------------ */
int mvae_linear_forward(const float* x, const float* W, const float* b, float* y, int64_t M, int N, int K, int relu,
                        void* stream);
/* y = (x W^T) zeroed where mask[M, N] <= 0: the backward-data contraction of a layer whose input came out of a ReLU
 * (mask = that ReLU's output), the mask applied in the contraction's epilogue.  16-byte aligned operands, K, N % 4 == 0. */
int mvae_linear_forward_masked(const float* x, const float* W, const float* mask, float* y, int64_t M, int N, int K,
                               uint16_t* y_planes, int64_t y_ps, void* stream);
/* mvae_linear_forward (M >= 512 rows, 16-byte aligned operands, K and N multiples of 4) with the result's bf16 planes written
 * by the epilogue next to y (the mvae_p3 section below says what planes are). */
int mvae_linear_forward_planes(const float* x, const float* W, const float* b, float* y, uint16_t* y_planes, int64_t y_ps,
                               int64_t M, int N, int K, int relu, void* stream);
/* dW[N,K] = dy^T x ; db[N] = colsum(dy) ; dx[M,K] = dy W  (dx may be NULL).  With relu_in != 0, x is the output of a
 * ReLU and dx is additionally masked by x > 0 (folds the previous activation's backward into this call). */
int mvae_linear_backward(const float* x, const float* W, const float* dy, int relu_in, float* dW, float* db, float* dx,
                         int64_t M, int N, int K, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Building blocks of the convolutional architecture.  ConvolutionalVAE, conv_vae.py:28-79: every Conv2d /
 * ConvTranspose2d there has kernel 4, stride 2, padding 1; both run on the MFMA contractions through a patch matrix.
 * (b, c, y, x) element strides are explicit, so NCHW and channel-last activations share the kernels.
 * ------------------------------------------------------------------------------------------------------------------ */
/* col[(b,oy,ox), (c,ky,kx)] = src[b, c, 2oy-1+ky, 2ox-1+kx] (0 outside; with mask != NULL also 0 where mask <= 0).
 * col is [B*IH/2*IW/2, C*16].  Forward of Conv2d (conv_vae.py:47-49,60-62) and backward of ConvTranspose2d.
 * taps_major != 0: the patch axis is ordered (ky,kx,c) instead of the weight layout's (c,ky,kx) -- 16-byte coalesced
 * moves for channel-last tensors (needs sc == 1, C % 4 == 0, 16-byte alignment); the caller contracts with weights
 * permuted to the same order (mvae_permute_rc on the [OC, C, 16] view). */
int mvae_im2col_k4s2p1(const float* src, const float* mask, float* col, int B, int C, int IH, int IW, int64_t sb,
                       int64_t sc, int64_t sy, int64_t sx, int taps_major, void* stream);
/* dst[b,c,y,x] = act(bias[c] + sum of the (<= 4) entries col[(b,py,px),(c,ky,kx)] with y = 2py-1+ky, x = 2px-1+kx);
 * col is [B*H/2*W/2, C*16].  Forward of ConvTranspose2d (conv_vae.py:52-55,72-74) and backward-data of Conv2d.
 * relu != 0: act = ReLU; mask != NULL: the result is zeroed where mask[b,c,y,x] <= 0 (backward through a ReLU).
 * taps_major: as above, col's second axis is (ky,kx,c). */
int mvae_col2im_k4s2p1(const float* col, const float* bias, const float* mask, float* dst, int B, int C, int H, int W,
                       int64_t sb, int64_t sc, int64_t sy, int64_t sx, int relu, int taps_major, uint16_t* dst_planes,
                       int64_t dst_ps, void* stream);  /* dst_planes (NULL: none): bf16 planes of dst, taps_major only */
/* The channel-last layers WITHOUT a patch matrix in memory (implicit contraction; the gather happens in the operand fetch
 * of the LDS-tiled MFMA kernel).  src[B, IH, IW, C] channel-last, C % 32 == 0, IH and IW powers of two;
 * Wt[OC, 16 C] with the patch axis taps-major (ky, kx, c); rows of y / dy are (b, oy, ox), OH = IH/2, OW = IW/2.
 *   y[(b,oy,ox), oc] = act(bias[oc] + sum_{ky,kx,c} src[b, 2oy-1+ky, 2ox-1+kx, c] Wt[oc, (ky,kx,c)]), zeroed where
 *   mask[(b,oy,ox), oc] <= 0 (mask may be NULL): Conv2d forward (conv_vae.py:60-62) and -- with src = the incoming
 *   gradient, Wt = the ConvTranspose2d weight [IC, (ky,kx,oc)], mask = the previous ReLU's output -- the backward-data of
 *   a ConvTranspose2d (conv_vae.py:72-74).
 *   dWt[oc, (ky,kx,c)] = sum_{b,oy,ox} dy[(b,oy,ox), oc] src[b, 2oy-1+ky, 2ox-1+kx, c]: the weight gradient of either
 *   (rows are cut into slices added in index order; workspace = mvae_conv_k4s2p1_nhwc_wgrad_workspace_floats floats). */
/* The TRANSPOSED convolution of the same family without its [B IH IW, 16 OC] product and without col2im: the output
 * pixels of one parity class (oy % 2, ox % 2) receive exactly 4 taps each, so the layer is four implicit contractions over
 * K = (4 taps, C).  src[B, IH, IW, C] channel-last -> y[B, 2 IH, 2 IW, OC] channel-last;
 *   y[b, oy, ox, oc] = act(bias[oc] + sum_{ky,kx,c : oy = 2 iy - 1 + ky, ox = 2 ix - 1 + kx} src[b, iy, ix, c] Wt[c, (ky,kx,oc)]),
 *   zeroed where mask[b, oy, ox, oc] <= 0.  Wt[C, 16 OC] = the ConvTranspose2d weight [C, OC, 4, 4] with its columns
 *   taps-major: ConvTranspose2d forward (conv_vae.py:52-55,72-74); with src = the incoming gradient [.., OC'] of a
 *   Conv2d, Wt = that layer's weight stored [OC', (ky,kx,c)], mask = the previous ReLU's output: its backward-data.
 *   C % 32 == 0, OC % 4 == 0, IH and IW powers of two. */
int mvae_conv_transpose_k4s2p1_nhwc(const float* src, const float* Wt, const float* bias, const float* mask, float* y, int B, int C,
                           int IH, int IW, int OC, int relu, int pass, uint16_t* y_planes, int64_t y_ps, void* stream);
/* workspace (may be NULL): mvae_conv_k4s2p1_nhwc_workspace_floats(...) floats; when given, a layer with fewer than 256
 * output tiles and a patch axis >= 2048 splits the contraction into <= 4 slices added in index order (mask == NULL only). */
int64_t mvae_conv_k4s2p1_nhwc_workspace_floats(int B, int C, int IH, int IW, int OC, int has_mask);
/* y_planes (may be NULL; "planes": see the mvae_p3 section below): the bf16 planes of y, written by the epilogue next to it
 * (plane stride y_ps elements; the split-K form is not taken then). */
int mvae_conv_k4s2p1_nhwc(const float* src, const float* Wt, const float* bias, const float* mask, float* y, int B, int C,
                          int IH, int IW, int OC, int relu, float* workspace, int pass, uint16_t* y_planes, int64_t y_ps,
                          void* stream);
int64_t mvae_conv_k4s2p1_nhwc_wgrad_workspace_floats(int B, int C, int IH, int IW, int OC);
int mvae_conv_k4s2p1_nhwc_wgrad(const float* dy, const float* src, float* dWt, int B, int C, int IH, int IW, int OC,
                                float* workspace, void* stream);
/* mvae_linear_forward for few rows and a long contraction (the heads of the conv architecture, component.py:52-57 on
 * the 8192-wide flatten): K is split into slices whose partial products are added in index order.  `workspace`:
 * mvae_linear_forward_splitk_workspace_floats(M, N, K) floats (0 = not needed). */
int64_t mvae_linear_forward_splitk_workspace_floats(int64_t M, int N, int K);
int mvae_linear_forward_splitk(const float* x, const float* W, const float* b, float* y, int64_t M, int N, int K,
                               int relu, float* workspace, void* stream);
/* out[b][c][r] = in[b][r][c]: the `.view(bs, -1)` / `.view(-1, 128, 4, 4)` re-flattenings of conv_vae.py:65,71. */
int mvae_permute_rc(const float* in, float* out, int64_t B, int R, int Cc, void* stream);
/* out[NP, NQ] = P[M, NP]^T Q[M, NQ]  (weight gradients).  For M > 256 the rows are processed in slices whose partial
 * products are added in index order: `workspace` must then hold mvae_gemm_tn_workspace_floats(M, NP, NQ) floats. */
int64_t mvae_gemm_tn_workspace_floats(int64_t M, int NP, int NQ);
int mvae_gemm_tn(const float* P, const float* Q, float* out, int64_t M, int NP, int NQ, float* workspace,
                 void* stream);
/* dy[i] = 0 where y[i] <= 0  (backward through a ReLU whose output is y). */
int mvae_relu_mask(float* dy, const float* y, int64_t n, void* stream);
/* out[M, N] = G[M, K] W[K, N], optionally zeroed where mask[M, N] <= 0.  pass: MVAE_PASS_FORWARD (the product of a
 * ConvTranspose2d forward) | MVAE_PASS_BACKWARD (a backward-data product), see mvae_set_contraction_mode. */
int mvae_gemm_nn(const float* G, const float* W, const float* mask, float* out, int64_t M, int K, int N, int pass,
                 void* stream);
/* out[N] = column sums of G[M, N]  (bias gradients); for M > 512 `workspace` must hold
 * mvae_colsum_workspace_floats(M, N) floats (row slices are summed separately, then added in index order). */
int64_t mvae_colsum_workspace_floats(int64_t M, int N);
int mvae_colsum(const float* G, float* out, int64_t M, int N, float* workspace, void* stream);
/* bce[r] = sum_j BCE-with-logits(logits[r][j], x[r][j]) and g = d(sum bce)/d(logits) = sigmoid(logits) - x
 * (image_reconstruction.py:142-143 with soft targets; vae.py:131). */
int mvae_bce_forward_backward(const float* logits, const float* x, float* bce, float* g, int64_t rows, int D,
                              void* stream);
/* BatchStats (stats.py:144-212): adds sum_b bce, sum_b kl_i, sum_b(-bce - beta*sum_i kl_i) to the statistics record
 * (same layout as mvae_model_desc.stats). */
int mvae_batch_stats(const float* bce, const float* kl, float* stats, float beta, int B, int ncomp, void* stream);
/* nn.ConvTranspose2d(64, 3, 4, 2, 1) (conv_vae.py:54) direct, from the channel-last activation src [B*IH*IW, 64] to NCHW
 * logits: y[b,c,Y,X] = bias[c] + sum src[(b,y,x)][f] W[f][c*16+ky*4+kx] over Y = 2y-1+ky, X = 2x-1+kx.  Geometry fixed to 64
 * features -> 3 x 32 x 32 (MVAE_E_UNSUPPORTED otherwise; the generic form is mvae_gemm_nn + mvae_col2im_k4s2p1). */
int mvae_convt_to3_k4s2p1_forward(const float* src, const float* W, const float* bias, float* y, int B, int F, int IH,
                                  int IW, int C, void* stream);
/* How the LDS-tiled contractions of the conv architecture multiply (process-wide, read once per call; returns the previous
 * mode; a negative argument only queries).
 * 0: f32-input MFMA (v_mfma_f32_16x16x4_f32), the f32 vector rate, everywhere.
 * 1: (contractions with > 64 output columns) every float split EXACTLY into three bf16 pieces, the six largest piece
 *    products on the bf16 MFMA (16x the rate), f32 accumulation: |error| <= 2^-23 |a b| per product on top of f32
 *    accumulation, i.e. the float32 class of nn.Linear / nn.Conv2d on any BLAS (conv_vae.py:47-55).
 * 2: (DEFAULT) as 1 for the contractions of the BACKWARD pass only (backward-data, weight gradients: autograd of
 *    conv_vae.py:57-79); every forward contraction -- whose output decides a ReLU mask or is the logits -- is exactly
 *    mode 0's, so forward values and masks are bit-identical to mode 0.
 * The entry points that serve both passes take `pass`; mvae_conv_k4s2p1_nhwc_wgrad, mvae_gemm_tn and
 * mvae_linear_forward_masked are backward by nature, mvae_linear_forward / _splitk forward. */
/* Which kernel the exact-f32 contractions of whole-tile shapes (M % 128 == 0, N % 64 == 0, K % 32 == 0) take: 1 (default) the
 * ping-pong LDS-DMA kernel of csrc/mvae_f32pp.hip, 0 the register-staged k_gemm_tiled; the results are BIT-IDENTICAL (same
 * order of MFMA steps per output element).  Returns the previous value; a negative argument only queries. */
int mvae_set_forward_kernel(int pingpong);
#define MVAE_PASS_FORWARD 0
#define MVAE_PASS_BACKWARD 1
int mvae_set_contraction_mode(int mode);
/* ---- Contractions on PRE-SPLIT operands ("planes"), csrc/mvae_p3.hip: the backward pass of the conv architecture (autograd
 * of conv_vae.py:57-79) in contraction mode 2.  A float splits EXACTLY into three bf16 pieces (hi, mid, lo: 8 + 8 + 8
 * significant bits); a tensor's PLANES are three bf16 arrays of its shape, plane q at planes + q * plane_stride (elements).
 * Whoever produces a tensor writes its planes once (the `*_planes` outputs below, mvae_split3_planes); the contractions
 * stage them by LDS-DMA and multiply the six largest piece products on the bf16 MFMA with f32 accumulation -- the
 * arithmetic of mvae_set_contraction_mode(1), error vs float64 no larger than the f32-input MFMA's.  Whole tiles only:
 * mvae_p3_supported(form, M, N, K, C) says whether a shape qualifies (form 0: mvae_conv_k4s2p1_nhwc_p3, M = B OH OW, N = OC,
 * K = 16 C; 1: mvae_gemm_nn_p3; 2: mvae_conv_transpose_k4s2p1_nhwc_p3, M = B IH IW, N = OC, K = 4 C; 3:
 * mvae_conv_k4s2p1_nhwc_wgrad_p3, M = B OH OW, N = OC, K = 16 C); callers fall back to the f32-operand entry points. */
int mvae_p3_supported(int form, int64_t M, int N, int K, int C);
/* planes[j] (3 x n[j] bf16, plane stride n[j]) of src[j] (n[j] floats, a multiple of 4), up to 12 tensors in ONE launch:
 * the conv weights after the optimizer step, activations whose producer does not write planes. */
int mvae_split3_planes(int njobs, const float* const* src, uint16_t* const* planes, const int64_t* n, void* stream);
/* The same planes, QUEUED: nothing is launched; the jobs (at most 4 ride, more are performed by the next flush) travel as extra
 * workgroups of the next mvae_conv_latent_forward on this thread -- a latency-bound launch with memory bandwidth to spare --
 * and any plane contraction issued before that performs them first, as does mvae_split3_planes_flush / the next
 * mvae_split3_planes[_queue] call.  The sources and the planes must stay alive until then.  Host-side state per calling thread
 * (graph-capturable).  MVAE_SPLIT_RIDE=0: the latent forward does not take them (they run at the first consumer). */
int mvae_split3_planes_queue(int njobs, const float* const* src, uint16_t* const* planes, const int64_t* n, void* stream);
int mvae_split3_planes_flush(void* stream);
/* Grouped launches: between mvae_p3_group(1, stream) and mvae_p3_group(0, stream) up to two plane contractions are queued
 * instead of launched; group(0) launches them -- a weight gradient (mvae_conv_k4s2p1_nhwc_wgrad_p3) and the backward-data of the
 * same layer (the other three entry points) as ONE kernel whose workgroups of the second start as those of the first finish --
 * followed by the slice sums that were waiting for them.  The two must be independent (neither reads the other's result). */
int mvae_p3_group(int on, void* stream);
/* mvae_conv_k4s2p1_nhwc (a Conv2d forward, conv_vae.py:47-50,57-63, or the backward-data of a ConvTranspose2d,
 * conv_vae.py:52-55,72-74) on the planes of src [B IH IW, C] and of Wt [OC, 16 C]; y = mask(relu(sum + bias)) (bias NULL: none;
 * relu 0: none; mask NULL: none) in f32, its planes too when y_planes != NULL (not together with a split-K workspace, which
 * only a call without bias / relu / mask uses).  y = NULL with a workspace that holds K slices: they are left there un-added (slice
 * count = workspace floats / (B OH OW OC)) for a consumer that adds them itself, mvae_conv_latent_backward.  colsum_out [OC] +
 * colsum_part [(B OH OW / 128) OC] (both or neither; unsliced calls): the column sums of the result -- the bias gradient of the
 * layer whose backward-data this is -- leave the epilogue as per-row-tile partial sums and are added in index order by the
 * (deferrable) slice sum; with them and y_planes, y may be NULL: the f32 result is then not written at all. */
int64_t mvae_conv_k4s2p1_nhwc_p3_workspace_floats(int B, int C, int IH, int IW, int OC, int has_mask);
int mvae_conv_k4s2p1_nhwc_p3(const uint16_t* src_planes, int64_t src_ps, const uint16_t* Wt_planes, int64_t w_ps,
                             const float* mask, const float* bias, int relu, float* y, uint16_t* y_planes, int64_t y_ps,
                             float* colsum_out, float* colsum_part, int B, int C, int IH, int IW, int OC, float* workspace,
                             void* stream);
/* mvae_gemm_nn on the planes of G [M, K] and W [K, N] (the product a Conv2d backward-data folds with mvae_col2im_k4s2p1). */
int mvae_gemm_nn_p3(const uint16_t* G_planes, int64_t g_ps, const uint16_t* W_planes, int64_t w_ps, float* out, int64_t M,
                    int K, int N, void* stream);
/* mvae_conv_transpose_k4s2p1_nhwc (a ConvTranspose2d forward, conv_vae.py:52-55,72-74, or a Conv2d's backward-data,
 * conv_vae.py:47-50,57-63) on the planes of src [B IH IW, C] and of Wt [C, 16 OC]; epilogue as mvae_conv_k4s2p1_nhwc_p3. */
int mvae_conv_transpose_k4s2p1_nhwc_p3(const uint16_t* src_planes, int64_t src_ps, const uint16_t* Wt_planes, int64_t w_ps,
                                       const float* mask, const float* bias, int relu, float* y, uint16_t* y_planes,
                                       int64_t y_ps, float* colsum_out, float* colsum_ws, int B, int C, int IH, int IW, int OC,
                                       void* stream);
/* floats of colsum_ws above (per-tile column sums of the four parity classes + the column sum's slice partials); colsum_out [OC]
 * + colsum_ws: as in mvae_conv_k4s2p1_nhwc_p3 (y may then be NULL when y_planes is given). */
int64_t mvae_conv_transpose_k4s2p1_nhwc_p3_colsum_floats(int B, int IH, int IW, int OC);
/* mvae_conv_k4s2p1_nhwc_wgrad on the planes of dy [B OH OW, OC] and of src [B IH IW, C]. */
int64_t mvae_conv_k4s2p1_nhwc_wgrad_p3_workspace_floats(int B, int C, int IH, int IW, int OC);
int mvae_conv_k4s2p1_nhwc_wgrad_p3(const uint16_t* dy_planes, int64_t dy_ps, const uint16_t* src_planes, int64_t src_ps,
                                   float* dWt, int B, int C, int IH, int IW, int OC, float* workspace, void* stream);
/* The 3-channel layers at the image boundary without a patch matrix (csrc/mvae_edge.hip; fixed geometry C = 3, IH = IW = 32,
 * F = 64, anything else: MVAE_E_UNSUPPORTED and the caller takes mvae_im2col_k4s2p1 + a contraction).
 * mvae_conv3_k4s2p1_nchw: y[(b,oy,ox), f] = mask(relu(bias[f] + sum_{c,ky,kx} img[b,c,2oy-1+ky,2ox-1+kx] W[f, c*16+ky*4+kx])) --
 * the forward pass of Conv2d(3, 64, 4, 2, 1) on the NCHW input (conv_vae.py:47,57; bias, relu = 1) and the backward-data of
 * ConvTranspose2d(64, 3, 4, 2, 1) w.r.t. its channel-last input (conv_vae.py:54,74; img = the gradient of the NCHW logits,
 * mask = the layer input whose ReLU the gradient passes).  y [B*256, 64] f32, + its bf16 planes when y_planes != NULL.  The
 * same bits as mvae_im2col_k4s2p1 + mvae_linear_forward(_masked).
 * mvae_conv3_k4s2p1_nchw_wgrad: dW[f, c*16+ky*4+kx] = sum_{b,oy,ox} act[(b,oy,ox), f] img[b,c,2oy-1+ky,2ox-1+kx] -- the weight
 * gradient of both layers (act = dL/d(e0 output) and img = x; act = the d3 input and img = the gradient of the logits).
 * Per-workgroup partial sums in `workspace` (mvae_conv3_k4s2p1_nchw_wgrad_workspace_floats floats), added in index order by
 * the slice sum (deferrable: mvae_slice_sums_defer). */
int mvae_conv3_k4s2p1_nchw(const float* img, const float* W, const float* bias, const float* mask, int relu, float* y,
                           uint16_t* y_planes, int64_t y_ps, int B, int C, int IH, int IW, int F, void* stream);
int64_t mvae_conv3_k4s2p1_nchw_wgrad_workspace_floats(int B, int C, int IH, int IW, int F);
int mvae_conv3_k4s2p1_nchw_wgrad(const float* act, const float* img, float* dW, int B, int C, int IH, int IW, int F,
                                 float* workspace, void* stream);
/* The backward pass of ConvTranspose2d(64, 3, 4, 2, 1) in ONE launch: mvae_conv3_k4s2p1_nchw_wgrad(act, img, dW) and
 * mvae_conv3_k4s2p1_nchw(img, W, NULL, mask = act, 0, y, y_planes) with img = the gradient of the NCHW logits and act = the
 * layer's channel-last input (same workspace as the weight gradient alone).  colsum_out [64] + colsum_ws
 * (mvae_conv3_k4s2p1_nchw_backward_colsum_floats floats; both or neither): the column sums of y -- the bias gradient of the layer
 * below -- from per-workgroup partial sums added by the (deferrable) column sum; with them and y_planes, y may be NULL. */
int mvae_conv3_k4s2p1_nchw_backward(const float* act, const float* img, const float* W, float* dW, float* y, uint16_t* y_planes,
                                    int64_t y_ps, float* colsum_out, float* colsum_ws, int B, int C, int IH, int IW, int F,
                                    float* workspace, void* stream);
int64_t mvae_conv3_k4s2p1_nchw_backward_colsum_floats(int B);
/* The loss end of the conv step in one launch: mvae_bce_forward_backward + mvae_batch_stats (vae.py:125-147) + the bias
 * gradient of the last ConvTranspose2d, dbias[c] = sum_{b,y,x} g[b,c,y,x] (conv_vae.py:54; logits are NCHW rows of
 * D = C x HW, C <= 8, HW a multiple of 1024).  chan_part: [B, C] scratch; counter: 17 int32 that are 0 before the first
 * call and are left 0 by every call (arrival counters of the workgroups; the last one to finish performs the two
 * batch-wide sums in a fixed order, so results do not depend on the arrival order).  Between mvae_slice_sums_defer(1) and
 * mvae_slice_sums_flush() those two batch-wide sums -- `dbias` and the update of `stats` -- are QUEUED like the slice sums and
 * performed by the flush launch (same additions, same order; inside this launch they are five dependent memory round trips
 * of one workgroup after all the others have finished); bce, kl and chan_part must then stay valid until the flush, and the
 * counters are not touched.  MVAE_LOSS_TAIL_DEFER=0: always inside this launch. */
int mvae_conv_bce_stats(const float* logits, const float* x, float* bce, float* g, const float* kl, float* stats,
                        float beta, int64_t B, int D, int HW, int ncomp, float* chan_part, float* dbias,
                        int32_t* counter, void* stream);
/* mvae_convt_to3_k4s2p1_forward + mvae_conv_bce_stats in ONE launch (the training step's loss end: conv_vae.py:54,74, then
 * vae.py:125-147): the logits of image b = ConvTranspose2d(64, 3, 4, 2, 1)(src [B IH IW, 64] channel-last) + bias on the matrix
 * cores, written to `logits` [B, 3, 2 IH, 2 IW]; then everything mvae_conv_bce_stats does with them (same arguments, same
 * counter protocol).  Fixed geometry F = 64, IH = IW = 16, C = 3. */
int mvae_convt_to3_bce_stats(const float* src, const float* W, const float* bias, const float* x, float* logits, float* bce,
                             float* g, const float* kl, float* stats, float beta, int64_t B, int F, int IH, int IW, int C,
                             int ncomp, float* chan_part, float* dbias, int32_t* counter, void* stream);
/* The latent section of the conv architecture, conv_vae.py:65-71: the encoder's flatten -> fc_mean / fc_logvar of every
 * component (component.py:52-57) -> rsample + KL (component.py:59-78) -> decoder fc + ReLU -> view(-1, 128, 4, 4), in two
 * launches, and its backward in two.  a2 [B, 16, 512] and t0 [B, 16, 128] are the CHANNEL-LAST activations next to the
 * convolutions; W_heads [heads_dim, 8192] (fc_mean rows of all components, then fc_logvar rows) and W_d0 [2048, z_dim]
 * keep the reference's orders (column / row c * 16 + p).  Supported: heads_dim <= 16, z_dim <= 16, true dimensions <= 8
 * (mvae_conv_latent_supported; other models use the generic operators above).  workspace:
 * mvae_conv_latent_workspace_floats(B, ncomp) floats, 16-byte aligned, scratch of one call.  Forward writes heads [B, heads_dim], z [B, z_dim],
 * kl [ncomp, B], t0.  Backward (loss = <dt0, t0> + beta * sum kl) writes dW_heads, db_heads, da2 (already masked by the
 * encoder's last ReLU), dW_d0, db_d0, dradii [ncomp] (0 for Euclidean components; fixed-order sums) and dheads [B, heads_dim]. 
 * dt0_slices > 1: dt0 points at that many partial results dt0_slice_stride floats apart (the K slices mvae_conv_k4s2p1_nhwc_p3
 * leaves in its workspace when called with y = NULL), added in mvae's slice-sum order while they are read; <= 1: one tensor.
 * da2_chansum [512] + da2_chansum_ws [8192] (both or neither): sum of da2 over rows and pixels per channel (the bias gradient of
 * the last encoder convolution) from the same launch; with them and da2_planes, da2 may be NULL.
 * t0_planes / da2_planes (NULL: none): the bf16 planes (mvae_split3_planes layout, plane stride *_ps elements) of t0 and da2,
 * written by the same launches for the plane contractions that consume them. */
int mvae_conv_latent_supported(const mvae_component_desc* comps, int ncomp);
int64_t mvae_conv_latent_workspace_floats(int64_t B, int ncomp);
int mvae_conv_latent_forward(const mvae_component_desc* comps, int ncomp, const float* a2, const float* W_heads,
                             const float* b_heads, const float* eps, int eps_ld, const float* radii, const float* W_d0,
                             const float* b_d0, float* heads, float* z, float* kl, float* t0, uint16_t* t0_planes,
                             int64_t t0_ps, float* workspace, int64_t B, void* stream);
int mvae_conv_latent_backward(const mvae_component_desc* comps, int ncomp, const float* a2, const float* W_heads,
                              const float* heads, const float* eps, int eps_ld, const float* radii, const float* z,
                              const float* W_d0, const float* t0, const float* dt0, int dt0_slices,
                              int64_t dt0_slice_stride, float beta, float* dW_heads,
                              float* db_heads, float* da2, uint16_t* da2_planes, int64_t da2_ps, float* da2_chansum,
                              float* da2_chansum_ws, float* dW_d0, float* db_d0, float* dradii, float* dheads,
                              float* workspace, int64_t B, void* stream);
/* torch-Adam over a flat buffer laid out like mvae_model_desc's (first 64 floats = raw radii, SGD on the trainable
 * ones iff do_curvature_step); counters as mvae_model_desc.step_count.  CurvatureOptimizer.step, utils.py:174-180.
 * radius_trainable[i]: 0 fixed, 1 trainable radius, 3 trainable universal curvature -- the entries marked 3 form the
 * clip_grad_norm_(max_norm=1) group of vae.py:161-163 and are clipped IN PLACE in `grads` before the SGD step. */
/* advance_cursor != 0: the launch also advances counters[8], the batch cursor of mvae_prepare_batch (the conv step, whose
 * first launch does not do it the way the fused MLP step's does). */
int mvae_optimizer_step_flat(float* params, float* grads, float* adam_m, float* adam_v, int64_t n_params,
                             int32_t* counters, int ncomp, const uint8_t* radius_trainable, double lr,
                             double curvature_lr, int do_curvature_step, int advance_cursor, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Importance-sampled log-likelihood pieces.  ModelVAE.log_likelihood, vae.py:82-123 ("next" row f-1 of the scope table).
 * ------------------------------------------------------------------------------------------------------------------ */
/* out[r] = sum_j binary_cross_entropy_with_logits(logits[r][j], x[r % x_rows][j])   (vae.py:108-109 without
 * materialising x.repeat((n,1,1))).  logits[rows, D], x[x_rows, D]. */
int mvae_bce_rows(const float* logits, const float* x, float* out, int64_t rows, int64_t x_rows, int D, void* stream);
/* The MLP decoder and the per-row BCE in one launch (vae.py:98-109 with ffnn_vae.py:52-60 as `decode`):
 *   out[r] = sum_j binary_cross_entropy_with_logits((relu(z[r] W_d0^T + b_d0) W_l^T + b_l)[j], x[r % x_rows][j])
 * z[rows, Z] (the n * B sampled latents), W_d0[H, Z], W_l[D, H], x[x_rows, D].  Neither the hidden layer nor the logits
 * are written to memory.  H in {16, 64, 128, 256, 400, 512}, D % 16 == 0, Z <= 64, W_l 16-byte aligned, targets below 4 GB
 * (x_rows * D * 4 < 2^32: the kernel addresses them with 32-bit byte offsets); anything else -- and a device that refuses the
 * kernel's dynamic LDS size -- returns MVAE_E_UNSUPPORTED without touching `out` (no error message: the caller composes mvae_linear_forward x 2 + mvae_bce_rows). */
int mvae_decode_bce_rows(const float* z, int64_t rows, int Z, const float* Wd0, const float* bd0, const float* Wl,
                         const float* bl, const float* x, int64_t x_rows, int H, int D, float* out, void* stream);
/* log_px[b] = logsumexp_n(-bce + log_p - log_q) - log n ; mi[b] = logsumexp_n(log_q - log_p) - log n   (vae.py:113-117)
 * bce, log_p, log_q: [n, B]. */
int mvae_loglik_reduce(const float* bce, const float* log_p, const float* log_q, float* log_px, float* mi, int n,
                       int B, void* stream);
/* The same with the per-component terms added up inside -- log_p / log_q [ncomp][n][B] as mvae_component_forward writes
 * them -- and, from the same pass, zmean[b][j] = mean_n z[n][b][j] (z [n][B][Z], Z <= 64; zmean NULL: skipped). */
int mvae_loglik_reduce_comps(const float* bce, const float* log_p, const float* log_q, int ncomp, const float* z, int Z,
                             float* log_px, float* mi, float* zmean, int n, int B, void* stream);
/* out[0] = || (x - mean_b x)^T (zmean - mean_b zmean) ||_F : the covariance norm of vae.py:119-121 with the mean over the
 * samples taken first (zmean from mvae_loglik_reduce_comps).  x [B][D], zmean [B][Z].  workspace: mvae_cov_norm_workspace_floats(D)
 * floats, ZEROED ONCE by the caller (the last float is an arrival counter the launch re-arms).  Z > 64 or B (16 + Z) floats
 * beyond 48 KB: MVAE_E_UNSUPPORTED without an error message. */
int64_t mvae_cov_norm_workspace_floats(int D);
int mvae_cov_norm(const float* x, const float* zmean, int B, int D, int Z, float* workspace, float* out, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * The whole step.  ModelVAE.train_step, vae.py:149-166; BatchStats, stats.py:144-212; CurvatureOptimizer.step,
 * mt/mvae/utils.py:174-180 with the routing of Trainer.build_optimizer, train.py:327-360.
 * All buffers are allocated by the host layer (torch) and only referenced here.
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct mvae_model_desc {
  int32_t abi_version;  /* MVAE_ABI_VERSION */
  int32_t arch;         /* 0 = feed-forward MLP (FeedForwardVAE, ffnn_vae.py:27-60) */
  int32_t batch;        /* B: rows per step */
  int32_t in_dim;       /* D */
  int32_t h_dim;        /* H */
  int32_t ncomp;
  int32_t heads_dim;    /* NH = sum true_dim + sum logvar_dim: rows of the fused head matrix */
  int32_t z_dim;        /* Z  = sum ambient dims */
  int32_t eps_dim;      /* sum true_dim */
  int32_t n_params;     /* P: floats in each of params/grads/adam_m/adam_v (flat layout below) */
  const mvae_component_desc* comps; /* HOST pointer, copied */
  /* offsets (in floats) into the flat buffers; every matrix row-major, torch.nn.Linear layout [out, in] */
  int64_t off_radii;    /* MUST be 0: the first 64 floats hold the raw radius parameters, entry i = component i
                           (components.{i}._nradius/_pradius, or _curvature for MVAE_UNIVERSAL);
                           comps[i].radius_idx == i                                                             */
  int64_t off_w_heads;  /* [NH, H]            fc_mean rows of every component, then fc_logvar rows             */
  int64_t off_b_heads;  /* [NH]                                                                                */
  int64_t off_w_e0;     /* [H, D]             fc_e0                                                            */
  int64_t off_b_e0;     /* [H]                                                                                 */
  int64_t off_w_d0;     /* [H, Z]             fc_d0                                                            */
  int64_t off_b_d0;     /* [H]                                                                                 */
  int64_t off_w_logits; /* [D, H]             fc_logits                                                        */
  int64_t off_b_logits; /* [D]                                                                                 */
  float* params;        /* [P] */
  float* grads;         /* [P]  dense gradient of (-ELBO) after mvae_step_forward_backward                     */
  float* adam_m;        /* [P] */
  float* adam_v;        /* [P] */
  int32_t* step_count;  /* [32] device-side so that a captured graph advances it; zeroed by the host at creation:
                           [0] Adam step counter, [1] arrival scratch, [2..3] this step's {-lr/bc1, sqrt(bc2)} as
                           float bits, [8] batch cursor of mvae_prepare_batch, [16..31] arrival scratch          */
  float* workspace;     /* [mvae_workspace_floats(desc)] activations, partial sums and the per-direction dual records
                           {d kl, d z} of the latent components (written by the forward, contracted by the backward) */
  float* stats;         /* [3 * (4 + ncomp)]: {bce, kl, elbo, n_steps, kl_0..} batch sums accumulated over steps, then the
                           same record for the LAST step only (stats.py:120-127 without the per-step .item() syncs:
                           the host reads it when it wants to, e.g. once per epoch), then the Kahan compensation
                           terms of the running sums (the reference accumulates Python doubles); zeroed by the host */
  uint8_t* radius_trainable; /* [ncomp] host pointer, copied: 0 = fixed curvature (requires_grad False)        */
  double lr;            /* Adam learning rate (run.py:33); betas (0.9, 0.999), eps 1e-8 = torch defaults        */
  double curvature_lr;  /* SGD lr on radii, 1e-4 (train.py:346,351)                                            */
} mvae_model_desc;

typedef struct mvae_ctx mvae_ctx;

int64_t mvae_workspace_floats(const mvae_model_desc* desc);
int mvae_create(const mvae_model_desc* desc, mvae_ctx** out);
void mvae_destroy(mvae_ctx* ctx);
/* Change which radius / curvature parameters are trained (Parameter.requires_grad toggles of the --universal schedule,
 * mt/examples/run.py:153-165).  trainable[ncomp], host pointer, copied; takes effect from the next step call. */
int mvae_set_radius_trainable(mvae_ctx* ctx, const uint8_t* trainable);
/* Padding rows.  The fused kernels want a batch that is a multiple of 16; the reference CLI's default is 100
 * (mt/examples/run.py:32).  A caller rounds `batch` up to a multiple of 16 in mvae_create, keeps rows [valid_rows, batch) of
 * every x / eps buffer finite (zeros) and declares them here: they then contribute no reconstruction term, no KL term, no
 * gradient and no statistics (ModelVAE.train_step sums over the rows of the batch, vae.py:125-147), and
 * mvae_set_next_batch_feed / the step's in-launch input pipeline prepare valid_rows rows per batch.  Only the four-launch
 * step (mvae_step_kernel_path() == MVAE_PATH_FUSED, batch <= 256) and the fragment-order block kernels (MVAE_PATH_BLOCK,
 * z_dim 17 .. 64) mask: for any other model / shape the call returns
 * MVAE_E_UNSUPPORTED (no message) and the caller creates a context for exactly valid_rows rows instead; a later step whose
 * buffers' alignment takes it off the four-launch kernels fails with MVAE_E_UNSUPPORTED rather than sum padding rows. */
int mvae_set_valid_rows(mvae_ctx* ctx, int valid_rows);

/* forward -> ELBO -> backward: fills `grads` (all P entries are written, nothing accumulates) and adds this step's
 * bce / kl / elbo sums to `stats`.  x[B, D] (binarised or soft targets), eps[B, eps_dim] ~ N(0,1).
 * With want_outputs != 0 also writes logits[B,D], concat_z[B,Z], bce[B], kl[ncomp,B] (any may be NULL). */
int mvae_step_forward_backward(mvae_ctx* ctx, const float* x, const float* eps, float beta, int want_outputs,
                               float* logits, float* concat_z, float* bce, float* kl, void* stream);

/* optimizer: fused Adam over the flat buffer (radii excluded) + SGD(lr=curvature_lr) on trainable radii iff
 * do_curvature_step (the reference's `not fixed_curvature and epoch >= 10`, train.py:357-358); the gradients of
 * universal curvatures are first clipped to joint L2 norm 1, in place (vae.py:161-163).  In data-parallel runs the
 * host all-reduces `grads` (SUM) between the two calls. */
int mvae_step_optimizer(mvae_ctx* ctx, int do_curvature_step, void* stream);

/* Both of the above back to back (single-GPU ModelVAE.train_step). */
int mvae_train_step(mvae_ctx* ctx, const float* x, const float* eps, float beta, int do_curvature_step, void* stream);

/* Device-side input pipeline (scope row f-2; reference: DataLoader workers + ImageDynamicBinarization,
 * mt/data/image_reconstruction.py:44-53,70-74, and the Normal.rsample draw inside the step).  Gathers batch number
 * (counters[8] % batches_per_epoch) of `images` (uint8 [n_images, D], HBM-resident) through `perm` (device int32
 * permutation, may be NULL = identity), writes x[B, D] = (pixel/255 > U(0,1)) (train == 1: ImageDynamicBinarization),
 * (pixel/255 > 0.5) (train == 0) or pixel/255 itself (train == 2: the CIFAR pipeline, ToTensor only,
 * image_reconstruction.py:123-127), and eps[B, E] ~ N(0,1); both from Philox4x32-10 keyed by (seed, counters[8]).  counters = mvae_model_desc.step_count;
 * its entry 8 (the batch cursor) is advanced by the first launch of every step, so [prepare, step] pairs can be
 * captured into a HIP graph and replayed for a whole epoch without host work. */
int mvae_prepare_batch(const uint8_t* images, const int32_t* perm, int n_images, int D, int B, int E, uint64_t seed,
                       const int32_t* counters, int batches_per_epoch, int train, float* x, float* eps, void* stream);
/* out[0 .. n) ~ N(0, 1) in ONE launch, never exactly 0 (both Box-Muller uniforms on the open interval; an all-zero eps row is
 * 0 / 0 on a sphere component, spherical.py:87-88): the eps draw of an eager step and of log_likelihood's n x B samples
 * (vae.py:90: rsample of n samples; here a Philox4x32-10 stream keyed by (seed, offset) -- item i of four values at counter
 * (i, offset) -- so equal (seed, offset) give equal bits; the host layer takes both from the torch generator and advances it). */
int mvae_randn(float* out, int64_t n, uint64_t seed, uint64_t offset, void* stream);

/* The same preparation WITHOUT a launch of its own ("fused into the step", scope row f-2): arms `ctx` so that the NEXT step
 * launched on it (mvae_train_step, mvae_step_forward_backward[_parts HEAD], mvae_step_profile) also prepares, on spare
 * workgroups of its launch 4, the batch its successor will consume -- batch number counters[8] AFTER that step's launch 1
 * has advanced the cursor -- into x_next[B, D] / eps_next[B, E]: bit for bit what mvae_prepare_batch would write if it were
 * called after that step with the same arguments.  x_next / eps_next must not be the buffers that step itself reads (callers
 * alternate between two pairs); the first batch of a sequence comes from mvae_prepare_batch.  One-shot: the step
 * consumes the arming; images == NULL disarms.  Host-side state only (no launch, no stream): under graph capture every
 * captured step carries its own arming.  Why the next batch rather than "in the first layer's operand load": the encoder
 * layer's 16-row operand panel is loaded by every one of its H / 16 column-tile workgroups, each of which would repeat the
 * ten Philox rounds per four pixels (quarter-rate 32-bit multiplies) on its critical path; on launch 4's spare workgroups
 * the same work runs once, beside the tiles. */
int mvae_set_next_batch_feed(mvae_ctx* ctx, const uint8_t* images, const int32_t* perm, int n_images, uint64_t seed,
                             int batches_per_epoch, int train, float* x_next, float* eps_next);

/* Measurement aid (never captured into a graph, synchronises): runs `iters` full steps on `stream` with a HIP event
 * between consecutive launches and writes the average duration of each of the MVAE_STEP_KERNELS launches, in
 * milliseconds, to the HOST array ms_out[MVAE_STEP_KERNELS] (order: enc_fwd, latent_fwd, dec1_fwd, dec1_bwd,
 * latent_bwd, enc_bwd -- the six launches of mvae_train_step, whose gradient epilogues carry the optimizer).
 * Parameters and optimizer state advance exactly as in mvae_train_step. */
#define MVAE_STEP_KERNELS 6
int mvae_step_profile(mvae_ctx* ctx, const float* x, const float* eps, float beta, int do_curvature_step, int iters,
                      float* ms_out, void* stream);

/* Deferred slice sums (conv step): mvae_gemm_tn, mvae_conv_k4s2p1_nhwc_wgrad and mvae_colsum split tall contractions
 * into row slices and finish with "add the slices in index order".  Between mvae_slice_sums_defer(1) and
 * mvae_slice_sums_flush(stream) they only write their slices and QUEUE that final sum; the flush performs all queued sums
 * (same order of additions, hence the same bits) in ONE launch -- in the conv backward pass that replaces ~13 launches
 * whose outputs nobody reads before the optimizer; the tall column sums (mvae_colsum with M > 512 rows, 16-byte aligned,
 * N % 4 == 0) are queued as a whole and performed by one more launch at the flush.  The caller keeps every workspace --
 * and the INPUT of every queued mvae_colsum -- alive until the flush.  State is
 * per calling thread and host-side only (graph-capturable); a full queue (24 entries) flushes itself.
 * on = 1: defer (a queue left over from an aborted pass is dropped when deferral is switched on); on = 2: suspend --
 * sums requested now run immediately, the queue is kept (for an intermediate result read before the flush; switch back
 * with 1); on = 0: off, anything still queued is DROPPED. */
int mvae_slice_sums_defer(int on);
int mvae_slice_sums_flush(void* stream);

/* ---- Peer-read gradient exchange (data-parallel training on ONE node; new functionality -- the reference is
 * single-device).  The intra-node alternative to "mvae_step_forward_backward -> RCCL all-reduce ->
 * mvae_step_optimizer": every rank publishes its flat gradient buffer in its own HBM and the optimizer launch of every
 * rank reads all ranks' buffers directly (hipIpc mappings: xGMI on a node) and adds them in RANK ORDER, so the sums --
 * and therefore the parameters -- are bit-identical on every rank.  No host work per step; graph-capturable.
 *
 *   mvae_peer_create(n_floats = n_params, world, rank, shm_name, timeout_seconds, &peer)
 *       allocates the rank's slot pair and opens the flag page `shm_name` (POSIX shared memory; the SAME fresh name on
 *       every rank of the job, e.g. "/mvae-<random token broadcast by rank 0>").  world <= MVAE_PEER_MAX_WORLD.
 *   mvae_peer_export(peer, handle)            the hipIpc handle of the rank's slots (MVAE_IPC_HANDLE_BYTES bytes) -- to be
 *   mvae_peer_import(peer, r, handle_of_r)    all-gathered by the host layer and imported for every r != rank
 *   per step, on `stream`:  mvae_step_forward_backward(...)  ->  mvae_peer_publish(peer, grads, stream)
 *                           ->  mvae_step_optimizer_peer(ctx, peer, do_curvature_step, stream)
 *       publish = copy into the slot of the next sequence number, then one signal kernel raises the rank's flag and waits
 *       for every peer's flag (bounded: after timeout_seconds a rank stops waiting, COUNTS the event and goes on with
 *       whatever is in the slot -- mvae_peer_timeouts() > 0 means the run is no longer synchronous);
 *       mvae_step_optimizer_peer = mvae_step_optimizer with g := sum over ranks of their slots (also written to `grads`).
 *   Every rank must call publish the same number of times.  mvae_peer_destroy synchronises the device. */
typedef struct mvae_peer mvae_peer;
#define MVAE_PEER_MAX_WORLD 16
#define MVAE_IPC_HANDLE_BYTES 64
int mvae_peer_create(int64_t n_floats, int world, int rank, const char* shm_name, double timeout_seconds,
                     mvae_peer** out);
void mvae_peer_destroy(mvae_peer* peer);
int mvae_peer_export(mvae_peer* peer, uint8_t handle[MVAE_IPC_HANDLE_BYTES]);
int mvae_peer_import(mvae_peer* peer, int peer_rank, const uint8_t handle[MVAE_IPC_HANDLE_BYTES]);
int mvae_peer_publish(mvae_peer* peer, const float* grads, void* stream);
/* The form of the exchange (before the first publish; the same on every rank; anything else: MVAE_E_BADARG).
 * on = 0: one-shot (above).
 * on = 1: two-shot.  publish then also adds the rank's OWN 1/world slice of every slot (rank order) in place and raises a
 *   second flag; the optimizer launch reads slice j from rank j.  Same sums, same bits as the one-shot form; per xGMI link
 *   and step 2 n / world floats instead of n.
 * on = 2: SHARDED OPTIMIZER on the same two rounds.  mvae_step_optimizer_peer becomes three launches: rank r adds slice r of
 *   every slot (rank order), applies Adam to that slice with ITS slice of m and v (rank 0: clip + SGD of the radii, which
 *   live in slice 0) and writes the new parameters to `params` and into its slot; second flag round; every rank copies the
 *   other slices of the PARAMETERS out of their owners' slots.  Same parameters, bit for bit, as the other forms; the
 *   optimizer's reads and writes per rank shrink with the world size.  `grads` then holds the summed gradient on the owner's
 *   slice only, and adam_m / adam_v are valid on the owner only (slice r = floats [4 r s4, 4 (r + 1) s4),
 *   s4 = max(16, ceil(n_params / 4 / world))). */
int mvae_peer_set_two_shot(mvae_peer* peer, int on);
int mvae_step_optimizer_peer(mvae_ctx* ctx, mvae_peer* peer, int do_curvature_step, void* stream);
int mvae_peer_timeouts(mvae_peer* peer);
/* The sharded optimizer for exchanges that leave the SUMMED gradient of the rank's slice in `grads` itself (an in-place
 * reduce-scatter -- mvae_flat_reduce_scatter -- or a whole all-reduce): mvae_step_optimizer restricted to slice `rank` of
 * `world` (float4 [rank s4, (rank + 1) s4), s4 = max(16, ceil(n_params / 4 / world)); the radii -- clip + SGD -- on rank 0,
 * whose slice holds them).  The caller all-gathers `params` afterwards (mvae_flat_allgather); adam_m / adam_v are valid on
 * the owner only.  Advances the step counter like mvae_step_optimizer.  New functionality. */
int mvae_step_optimizer_slice(mvae_ctx* ctx, int rank, int world, int do_curvature_step, void* stream);

/* ---- Flat all-reduce on librccl directly (data-parallel training; new functionality -- the reference is single-device,
 * SURVEY.md section 8b names this export).  One process per GPU; the SUM of the flat gradient buffer is the ONE exchange
 * of a data-parallel step (the loss is a batch sum: mt/mvae/stats.py:200-202, vae.py:158).
 *   rank 0: mvae_rccl_unique_id(id) -> the host layer hands the 128 bytes to every rank (any side channel: the c10d store)
 *   all:    mvae_rccl_create(id, rank, world, &comm)            (collective: ncclCommInitRank)
 *   per step, on `stream`: mvae_step_forward_backward(...) -> mvae_flat_allreduce(comm, grads, n_params, stream)
 *                          -> mvae_step_optimizer(...)
 * ncclAllReduce is enqueued on the caller's stream: ordered with the step's launches and captured into HIP graphs like
 * them -- no torch ProcessGroupNCCL (and no watchdog thread polling events from another thread) exists on this route.
 * librccl is dlopen'ed on first use (mvae_rccl_load(path): an explicit path, else the copy already mapped into the
 * process, else the system's); single-GPU use never touches it. */
#define MVAE_RCCL_ID_BYTES 128
typedef struct mvae_rccl mvae_rccl;
int mvae_rccl_load(const char* path);
int mvae_rccl_unique_id(uint8_t id[MVAE_RCCL_ID_BYTES]);
int mvae_rccl_create(const uint8_t id[MVAE_RCCL_ID_BYTES], int rank, int world, mvae_rccl** out);
void mvae_rccl_destroy(mvae_rccl* comm);
int mvae_flat_allreduce(mvae_rccl* comm, float* buf, int64_t n, void* stream);
/* The two halves of the all-reduce around mvae_step_optimizer_slice, both IN PLACE on the flat buffer: after the
 * reduce-scatter rank r holds the sums of floats [r n / world, (r + 1) n / world) in that range of `buf` (the rest is
 * unspecified); the all-gather hands every rank's range to everybody.  n % world != 0: MVAE_E_UNSUPPORTED (keep the
 * replicated optimizer). */
int mvae_flat_reduce_scatter(mvae_rccl* comm, float* buf, int64_t n, void* stream);
int mvae_flat_allgather(mvae_rccl* comm, float* buf, int64_t n, void* stream);
int mvae_flat_broadcast(mvae_rccl* comm, void* buf, int64_t n_words, int root, void* stream);
int mvae_rccl_group(int begin);

/* Which kernels the latent part of the step takes for this context's shapes (a measurement / test aid; the result
 * of the step does not depend on it beyond float32 summation order):
 *   MVAE_PATH_ROW    one batch row per workgroup (k_latent_fwd / k_latent_bwd): the general case
 *   MVAE_PATH_FUSED  launches 2 + 3 fused (k_fwd23): heads_dim <= 16, z_dim in {2, 4, 6, 8} -- BASELINE configs [0], [1], [2]
 *   MVAE_PATH_BLOCK  16-row MFMA blocks (k_heads_comp, k_fwd3m with the dual-record workgroups, k_latent_bwd_blk): many
 *                    small components --
 *                    BASELINE config [3] (`6h2,6s2,6e2`)
 * Assumes 16-byte aligned x (true for every torch allocation). */
#define MVAE_PATH_ROW 0
#define MVAE_PATH_FUSED 1
#define MVAE_PATH_BLOCK 2
int mvae_step_kernel_path(const mvae_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* MVAE_HIP_H */
