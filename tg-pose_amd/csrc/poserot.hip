// The rotation R_DCD canonicalises with (losses/TDA_loss_sym_recon.py:327-333: get_vertical_rot_vec_in_batch :370-395 twice, the
// symmetric / non-symmetric choice, get_rot_mat_y_first :351-360) as ONE launch with its gradient.  Under torch autograd the same
// arithmetic on (B, 3) tensors is ~100 element-wise launches forward and ~200 backward per training step, each a few microseconds of
// a captured graph (profiles/r03_b_*: ATen glue 3.1 ms of a 20.8 ms step).  One thread per object evaluates the formulas on
// forward-mode dual numbers that carry the derivatives with respect to the eight differentiable inputs (predicted green axis,
// its confidence, predicted red axis, its confidence): the forward launch writes R (B, 3, 3) and its Jacobian (B, 9, 8), the
// backward launch is d in[b] = J[b]^T d R[b].  The formulas are written once, on the number type.
#include "tgp_common.h"

#define PR_NIN 8

struct PrDual {
    float v;
    float d[PR_NIN];
};
__device__ __forceinline__ PrDual pr_const(float v)
{
    PrDual r;
    r.v = v;
#pragma unroll
    for (int i = 0; i < PR_NIN; ++i) r.d[i] = 0.f;
    return r;
}
__device__ __forceinline__ PrDual pr_var(float v, int slot)
{
    PrDual r = pr_const(v);
    r.d[slot] = 1.f;
    return r;
}
#define PR_EACH for (int i = 0; i < PR_NIN; ++i)
__device__ __forceinline__ PrDual operator+(const PrDual &a, const PrDual &b)
{
    PrDual r;
    r.v = a.v + b.v;
#pragma unroll
    PR_EACH r.d[i] = a.d[i] + b.d[i];
    return r;
}
__device__ __forceinline__ PrDual operator-(const PrDual &a, const PrDual &b)
{
    PrDual r;
    r.v = a.v - b.v;
#pragma unroll
    PR_EACH r.d[i] = a.d[i] - b.d[i];
    return r;
}
__device__ __forceinline__ PrDual operator-(const PrDual &a)
{
    PrDual r;
    r.v = -a.v;
#pragma unroll
    PR_EACH r.d[i] = -a.d[i];
    return r;
}
__device__ __forceinline__ PrDual operator*(const PrDual &a, const PrDual &b)
{
    PrDual r;
    r.v = a.v * b.v;
#pragma unroll
    PR_EACH r.d[i] = a.d[i] * b.v + a.v * b.d[i];
    return r;
}
__device__ __forceinline__ PrDual operator/(const PrDual &a, const PrDual &b)
{
    PrDual r;
    r.v = a.v / b.v;
#pragma unroll
    PR_EACH r.d[i] = (a.d[i] - r.v * b.d[i]) / b.v;
    return r;
}
__device__ __forceinline__ PrDual pr_scale(const PrDual &a, float s)
{
    PrDual r;
    r.v = a.v * s;
#pragma unroll
    PR_EACH r.d[i] = a.d[i] * s;
    return r;
}
__device__ __forceinline__ PrDual pr_chain(const PrDual &a, float value, float slope)   // f(a) with f'(a.v) = slope
{
    PrDual r;
    r.v = value;
#pragma unroll
    PR_EACH r.d[i] = a.d[i] * slope;
    return r;
}
__device__ __forceinline__ PrDual pr_sqrt(const PrDual &a)
{
    const float s = sqrtf(a.v);
    return pr_chain(a, s, 0.5f / s);                      // (torch.norm's backward: the same quotient, infinite at 0)
}
__device__ __forceinline__ PrDual pr_acos(const PrDual &a) { return pr_chain(a, acosf(a.v), -1.f / sqrtf(1.f - a.v * a.v)); }
__device__ __forceinline__ PrDual pr_cos(const PrDual &a) { return pr_chain(a, cosf(a.v), -sinf(a.v)); }
__device__ __forceinline__ PrDual pr_sin(const PrDual &a) { return pr_chain(a, sinf(a.v), cosf(a.v)); }
// torch.clamp(x, lo, hi): gradient 1 inside [lo, hi] (bounds included), 0 outside
__device__ __forceinline__ PrDual pr_clamp(const PrDual &a, float lo, float hi)
{
    if (a.v < lo) return pr_const(lo);
    if (a.v > hi) return pr_const(hi);
    return a;
}

struct PrVec {
    PrDual x, y, z;
};
__device__ __forceinline__ PrVec pr_cross(const PrVec &a, const PrVec &b)
{
    PrVec r;
    r.x = a.y * b.z - a.z * b.y;
    r.y = a.z * b.x - a.x * b.z;
    r.z = a.x * b.y - a.y * b.x;
    return r;
}
__device__ __forceinline__ PrDual pr_dot(const PrVec &a, const PrVec &b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ PrVec pr_mul(const PrVec &a, const PrDual &s)
{
    PrVec r;
    r.x = a.x * s, r.y = a.y * s, r.z = a.z * s;
    return r;
}
__device__ __forceinline__ PrVec pr_add(const PrVec &a, const PrVec &b)
{
    PrVec r;
    r.x = a.x + b.x, r.y = a.y + b.y, r.z = a.z + b.z;
    return r;
}
__device__ __forceinline__ PrVec pr_div(const PrVec &a, const PrDual &s)
{
    PrVec r;
    r.x = a.x / s, r.y = a.y / s, r.z = a.z / s;
    return r;
}
// v rotated about the unit axis k by `angle` (to_rot_matrix_in_batch applied to v):  v c + (k x v) s + k (k . v)(1 - c)
__device__ __forceinline__ PrVec pr_rodrigues(const PrVec &k, const PrDual &angle, const PrVec &v)
{
    const PrDual c = pr_cos(angle), s = pr_sin(angle);
    const PrDual one_c = pr_const(1.f) - c;
    return pr_add(pr_add(pr_mul(v, c), pr_mul(pr_cross(k, v), s)), pr_mul(pr_mul(k, pr_dot(k, v)), one_c));
}
// get_vertical_rot_vec_in_batch (:370-395): rotate y and z about their common normal until they are perpendicular, sharing the
// correction in proportion to the OTHER axis' confidence
__device__ __forceinline__ void pr_vertical(const PrDual &c1, const PrDual &c2, const PrVec &y, const PrVec &z, PrVec &ny, PrVec &nz)
{
    PrVec k = pr_cross(y, z);
    const PrDual nrm = pr_sqrt(pr_dot(k, k)) + pr_const(1e-8f);
    k = pr_div(k, nrm);
    const PrDual theta = pr_acos(pr_clamp(pr_dot(y, z), -1.f + 1e-6f, 1.f - 1e-6f)) - pr_const(1.57079632679489661923f);
    const PrDual sum = c1 + c2;
    ny = pr_rodrigues(k, (c2 / sum) * theta, y);
    nz = pr_rodrigues(k, -((c1 / sum) * theta), z);
}
// F.normalize(v, dim=-1): v / max(|v|, 1e-12)
__device__ __forceinline__ PrVec pr_normalize(const PrVec &v)
{
    PrDual n = pr_sqrt(pr_dot(v, v));
    if (n.v < 1e-12f) n = pr_const(1e-12f);               // clamp_min: no gradient through the norm below the floor
    return pr_div(v, n);
}

// gR0: column 0 of the ground-truth rotation, g_R[b][:, 0] (B, 3); sym0: sym[b][0].  R (B, 3, 3) row-major, J (B, 9, 8)
__global__ void pose_rotation_kernel(const float *__restrict__ gR0, const float *__restrict__ p_g, const float *__restrict__ f_g,
                                     const float *__restrict__ p_r, const float *__restrict__ f_r, const float *__restrict__ sym0, int B,
                                     float *__restrict__ R, float *__restrict__ J)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    PrVec pg, pr, g0;
    pg.x = pr_var(p_g[3 * b], 0), pg.y = pr_var(p_g[3 * b + 1], 1), pg.z = pr_var(p_g[3 * b + 2], 2);
    const PrDual fg = pr_var(f_g[b], 3);
    pr.x = pr_var(p_r[3 * b], 4), pr.y = pr_var(p_r[3 * b + 1], 5), pr.z = pr_var(p_r[3 * b + 2], 6);
    const PrDual fr = pr_var(f_r[b], 7);
    g0.x = pr_const(gR0[3 * b]), g0.y = pr_const(gR0[3 * b + 1]), g0.z = pr_const(gR0[3 * b + 2]);
    PrVec y, x;
    if (sym0[b] == 1.f) pr_vertical(fg, pr_const(1e-5f), pg, g0, y, x);       // symmetric about y: the red axis is the ground truth's
    else pr_vertical(fg, fr, pg, pr, y, x);
    y = pr_normalize(y);
    const PrVec z = pr_normalize(pr_cross(x, y));
    const PrVec c0 = pr_cross(y, z);
    // torch.stack((cross(y, z), y, z), dim=-1): R[:, :, 0] = cross(y, z), R[:, :, 1] = y, R[:, :, 2] = z
    const PrDual *out[9] = {&c0.x, &y.x, &z.x, &c0.y, &y.y, &z.y, &c0.z, &y.z, &z.z};
#pragma unroll
    for (int e = 0; e < 9; ++e) {
        R[9 * b + e] = out[e]->v;
#pragma unroll
        for (int i = 0; i < PR_NIN; ++i) J[(9 * b + e) * PR_NIN + i] = out[e]->d[i];
    }
}

extern "C" int tgp_pose_rotation_fwd(const float *gR0, const float *p_g, const float *f_g, const float *p_r, const float *f_r,
                                     const float *sym0, int B, float *R, float *J, tgp_stream_t stream)
{
    TGP_REQUIRE(gR0 && p_g && f_g && p_r && f_r && sym0 && R && J && B > 0);
    hipLaunchKernelGGL(pose_rotation_kernel, dim3(tgp_cdiv(B, 64)), dim3(64), 0, tgp_hs(stream), gR0, p_g, f_g, p_r, f_r, sym0, B, R, J);
    return TGP_LAUNCH_RESULT();
}

// d in[b][i] = sum_e dR[b][e] J[b][e][i]:  in = (p_g x y z, f_g, p_r x y z, f_r)
__global__ void pose_rotation_bwd_kernel(const float *__restrict__ dR, const float *__restrict__ J, int B, float *__restrict__ din)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * PR_NIN) return;
    const int b = t / PR_NIN, i = t % PR_NIN;
    float acc = 0.f;
#pragma unroll
    for (int e = 0; e < 9; ++e) acc += dR[9 * b + e] * J[(9 * b + e) * PR_NIN + i];
    din[t] = acc;
}

extern "C" int tgp_pose_rotation_bwd(const float *dR, const float *J, int B, float *din, tgp_stream_t stream)
{
    TGP_REQUIRE(dR && J && din && B > 0);
    hipLaunchKernelGGL(pose_rotation_bwd_kernel, dim3(tgp_cdiv(B * PR_NIN, 64)), dim3(64), 0, tgp_hs(stream), dR, J, B, din);
    return TGP_LAUNCH_RESULT();
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Backward of tgp_head_post (PoseNet9D.py:57-66: axis / (|axis| + 1e-6), sigmoid of the confidences, T = ts[:3] + mean, s = ts[3:]):
// d green (B, 4), d red (B, 4), d ts (B, 6) from the gradients of the six outputs (a NULL gradient counts as zero).  Under autograd
// the same lines are ~10 launches forward and ~25 backward.
__device__ __forceinline__ void hp_axis_bwd(const float *v4, const float *gp, float gf, float *dv4)
{
    // v4 = (confidence logit, x, y, z);  p = v / (n + eps), n = |v|:  dv = g / (n + eps) - v (v . g) / (n (n + eps)^2)
    const float x = v4[1], y = v4[2], z = v4[3];
    const float n = sqrtf((x * x + y * y) + z * z), ne = n + 1e-6f;
    float gx = 0.f, gy = 0.f, gz = 0.f;
    if (gp) gx = gp[0], gy = gp[1], gz = gp[2];
    const float vg = (x * gx + y * gy) + z * gz;
    const float k = n > 0.f ? vg / (n * ne * ne) : 0.f;      // (torch.norm's backward at the origin: 0)
    dv4[1] = gx / ne - x * k, dv4[2] = gy / ne - y * k, dv4[3] = gz / ne - z * k;
    const float f = 1.0f / (1.0f + expf(-v4[0]));
    dv4[0] = gf * f * (1.f - f);
}

__global__ void head_post_bwd_kernel(const float *__restrict__ green, const float *__restrict__ red, int ldg, int ldr, int B,
                                     const float *__restrict__ g_pg, const float *__restrict__ g_pr, const float *__restrict__ g_fg,
                                     const float *__restrict__ g_fr, const float *__restrict__ g_T, const float *__restrict__ g_s,
                                     float *__restrict__ dgreen, float *__restrict__ dred, float *__restrict__ dts)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    hp_axis_bwd(green + b * ldg, g_pg ? g_pg + 3 * b : nullptr, g_fg ? g_fg[b] : 0.f, dgreen + 4 * b);
    hp_axis_bwd(red + b * ldr, g_pr ? g_pr + 3 * b : nullptr, g_fr ? g_fr[b] : 0.f, dred + 4 * b);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        dts[6 * b + c] = g_T ? g_T[3 * b + c] : 0.f;
        dts[6 * b + 3 + c] = g_s ? g_s[3 * b + c] : 0.f;
    }
}

extern "C" int tgp_head_post_bwd(const float *green, const float *red, int ldg, int ldr, int B, const float *g_pg, const float *g_pr,
                                 const float *g_fg, const float *g_fr, const float *g_T, const float *g_s, float *dgreen, float *dred,
                                 float *dts, tgp_stream_t stream)
{
    TGP_REQUIRE(green && red && dgreen && dred && dts && B > 0 && ldg >= 4 && ldr >= 4);
    hipLaunchKernelGGL(head_post_bwd_kernel, dim3(tgp_cdiv(B, 64)), dim3(64), 0, tgp_hs(stream), green, red, ldg, ldr, B, g_pg, g_pr, g_fg,
                       g_fr, g_T, g_s, dgreen, dred, dts);
    return TGP_LAUNCH_RESULT();
}
