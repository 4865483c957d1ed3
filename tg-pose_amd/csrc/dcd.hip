// Density-aware Chamfer loss pieces of TG-Pose on gfx950 (losses/TDA_loss_sym_recon.py):
//   calc_dcd :411-450  -- after the Chamfer search: exp(-alpha d), per-sample bincount of the nearest indices,
//                         weights count^(-lambda), (1 - e w).mean() per direction, loss1 + 0.5 loss2.
//                         The reference loops over the batch in Python with torch.bincount per sample; here one
//                         workgroup per object builds both histograms in LDS and reduces in a fixed order.
//   R_DCD    :326-342  -- rotation from the two predicted axes and their confidences
//                         (get_vertical_rot_vec_in_batch :370-395, get_rot_mat_y_first :351-360), then
//                         R^T (points - t) * s for every point.
#include "tgp_common.h"

#define DCD_MAX_PTS 4096

__device__ __forceinline__ float block_sum_256(float v, float *scratch)
{
    // fixed-order tree: lanes by xor butterfly, then the 4 waves in order
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    return ((scratch[0] + scratch[1]) + scratch[2]) + scratch[3];
}

__global__ __launch_bounds__(256) void dcd_fwd_kernel(const float *__restrict__ dist1, const float *__restrict__ dist2,
                                                      const int32_t *__restrict__ idx1, const int32_t *__restrict__ idx2,
                                                      int n, int m, float alpha, float lambda, float frac_21, float frac_12,
                                                      float *__restrict__ loss, float *__restrict__ w1_out,
                                                      float *__restrict__ w2_out)
{
    __shared__ int cnt1[DCD_MAX_PTS]; // hits per point of cloud 2 (indexed by idx1 values)
    __shared__ int cnt2[DCD_MAX_PTS];
    __shared__ float scratch[4];
    const int b = blockIdx.x;
    for (int t = threadIdx.x; t < m; t += 256) cnt1[t] = 0;
    for (int t = threadIdx.x; t < n; t += 256) cnt2[t] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 256) atomicAdd(&cnt1[idx1[(size_t)b * n + i]], 1);
    for (int j = threadIdx.x; j < m; j += 256) atomicAdd(&cnt2[idx2[(size_t)b * m + j]], 1);
    __syncthreads();
    float s1 = 0.f, s2 = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float e = expf(-dist1[(size_t)b * n + i] * alpha);
        float w = powf((float)cnt1[idx1[(size_t)b * n + i]], lambda);
        w = (1.0f / (w + 1e-6f)) * frac_21;
        if (w1_out) w1_out[(size_t)b * n + i] = w;
        s1 += (-e * w + 1.0f);
    }
    for (int j = threadIdx.x; j < m; j += 256) {
        const float e = expf(-dist2[(size_t)b * m + j] * alpha);
        float w = powf((float)cnt2[idx2[(size_t)b * m + j]], lambda);
        w = (1.0f / (w + 1e-6f)) * frac_12;
        if (w2_out) w2_out[(size_t)b * m + j] = w;
        s2 += (-e * w + 1.0f);
    }
    const float t1 = block_sum_256(s1, scratch);
    const float t2 = block_sum_256(s2, scratch);
    if (threadIdx.x == 0) loss[b] = t1 / (float)n + 0.5f * (t2 / (float)m);
}

extern "C" int tgp_dcd_fwd(const float *dist1, const float *dist2, const int32_t *idx1, const int32_t *idx2, int B, int n,
                           int m, float alpha, float n_lambda, int non_reg, float *loss, float *w1, float *w2,
                           tgp_stream_t stream)
{
    TGP_REQUIRE(dist1 && dist2 && idx1 && idx2 && loss && B > 0 && n > 0 && m > 0);
    if (n > DCD_MAX_PTS || m > DCD_MAX_PTS) return TGP_EUNSUPPORTED;
    // calc_dcd :420-425: frac_12 = n_pred / n_gt, frac_21 = n_gt / n_pred (clamped to >= 1 when non_reg)
    float frac_12 = (float)((double)n / (double)m), frac_21 = (float)((double)m / (double)n);
    if (non_reg) frac_12 = frac_12 < 1.f ? 1.f : frac_12, frac_21 = frac_21 < 1.f ? 1.f : frac_21;
    hipLaunchKernelGGL(dcd_fwd_kernel, dim3(B), dim3(256), 0, tgp_hs(stream), dist1, dist2, idx1, idx2, n, m, alpha, n_lambda,
                       frac_21, frac_12, loss, w1, w2);
    return TGP_LAUNCH_RESULT();
}

// d loss[b] / d dist1[b,i] = alpha * exp(-alpha d) * w1 / n (weights are detached in the reference, :433,438)
__global__ void dcd_bwd_kernel(const float *__restrict__ dist1, const float *__restrict__ dist2, const float *__restrict__ w1,
                               const float *__restrict__ w2, const float *__restrict__ gloss, int B, int n, int m, float alpha,
                               float *__restrict__ gd1, float *__restrict__ gd2)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n1 = (int64_t)B * n, n2 = (int64_t)B * m;
    if (t < n1) {
        const int b = (int)(t / n);
        gd1[t] = gloss[b] * (alpha * expf(-dist1[t] * alpha) * w1[t]) / (float)n;
    } else if (t < n1 + n2) {
        const int64_t u = t - n1;
        const int b = (int)(u / m);
        gd2[u] = gloss[b] * 0.5f * (alpha * expf(-dist2[u] * alpha) * w2[u]) / (float)m;
    }
}

extern "C" int tgp_dcd_bwd(const float *dist1, const float *dist2, const float *w1, const float *w2, const float *gloss,
                           int B, int n, int m, float alpha, float *gd1, float *gd2, tgp_stream_t stream)
{
    TGP_REQUIRE(dist1 && dist2 && w1 && w2 && gloss && gd1 && gd2 && B > 0 && n > 0 && m > 0);
    const int64_t total = (int64_t)B * (n + m);
    hipLaunchKernelGGL(dcd_bwd_kernel, dim3(tgp_cdiv(total, 256)), dim3(256), 0, tgp_hs(stream), dist1, dist2, w1, w2, gloss, B,
                       n, m, alpha, gd1, gd2);
    return TGP_LAUNCH_RESULT();
}

// ---------------------------------------------------------------------------------------------------
struct V3 {
    float x, y, z;
};
__device__ __forceinline__ V3 cross3(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ float dot3(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ float norm3(V3 a) { return sqrtf(dot3(a, a)); }

// Rodrigues matrix of to_rot_matrix_in_batch (:398-408) applied to v
__device__ __forceinline__ V3 rot_apply(V3 k, float s, float c, V3 v)
{
    const float oc = 1.0f - c;
    V3 o;
    o.x = (k.x * k.x * oc + c) * v.x + (k.x * k.y * oc - k.z * s) * v.y + (k.x * k.z * oc + k.y * s) * v.z;
    o.y = (k.y * k.x * oc + k.z * s) * v.x + (k.y * k.y * oc + c) * v.y + (k.y * k.z * oc - k.x * s) * v.z;
    o.z = (k.x * k.z * oc - k.y * s) * v.x + (k.z * k.y * oc + k.x * s) * v.y + (k.z * k.z * oc + c) * v.z;
    return o;
}

// get_vertical_rot_vec_in_batch: pull the two axes apart/together to 90 degrees, split by the confidences
__device__ __forceinline__ void vertical_axes(float c1, float c2, V3 y, V3 z, V3 &ny, V3 &nz)
{
    V3 k = cross3(y, z);
    const float kn = norm3(k) + 1e-8f;
    k.x /= kn, k.y /= kn, k.z /= kn;
    float cs = dot3(y, z);
    cs = fminf(fmaxf(cs, -1.0f + 1e-6f), 1.0f - 1e-6f);
    const float theta = acosf(cs);
    const float half_pi = 1.57079632679489661923f;
    const float th2 = c1 / (c1 + c2) * (theta - half_pi);
    const float th1 = c2 / (c1 + c2) * (theta - half_pi);
    ny = rot_apply(k, sinf(th1), cosf(th1), y);
    nz = rot_apply(k, sinf(-th2), cosf(-th2), z);
}

// out = (R^T (points - t)) * s with R from the predicted axes (R_DCD :326-339).  R (B,3,3) is also stored.
__global__ __launch_bounds__(256) void canonicalize_kernel(const float *__restrict__ points, const float *__restrict__ gR,
                                                           const float *__restrict__ p_g, const float *__restrict__ f_g,
                                                           const float *__restrict__ p_r, const float *__restrict__ f_r,
                                                           const float *__restrict__ p_t, const float *__restrict__ p_s,
                                                           const float *__restrict__ sym, int sym_ld, int n,
                                                           float *__restrict__ out, float *__restrict__ R_out)
{
    const int b = blockIdx.y;
    const V3 yg = {p_g[b * 3], p_g[b * 3 + 1], p_g[b * 3 + 2]};
    V3 ny, nx;
    if (sym[(size_t)b * sym_ld] == 1.0f) { // rotation about y is free: pair the green axis with the true x axis
        const V3 gx = {gR[b * 9 + 0], gR[b * 9 + 3], gR[b * 9 + 6]};
        vertical_axes(f_g[b], 1e-5f, yg, gx, ny, nx);
    } else {
        const V3 xr = {p_r[b * 3], p_r[b * 3 + 1], p_r[b * 3 + 2]};
        vertical_axes(f_g[b], f_r[b], yg, xr, ny, nx);
    }
    // get_rot_mat_y_first: y = normalize(y); z = normalize(x cross y); x = y cross z; columns (x, y, z)
    float yn = fmaxf(norm3(ny), 1e-12f);
    V3 y = {ny.x / yn, ny.y / yn, ny.z / yn};
    V3 z = cross3(nx, y);
    const float zn = fmaxf(norm3(z), 1e-12f);
    z.x /= zn, z.y /= zn, z.z /= zn;
    const V3 x = cross3(y, z);
    if (R_out && blockIdx.x == 0 && threadIdx.x == 0) {
        float *Ro = R_out + b * 9;
        Ro[0] = x.x, Ro[1] = y.x, Ro[2] = z.x;
        Ro[3] = x.y, Ro[4] = y.y, Ro[5] = z.y;
        Ro[6] = x.z, Ro[7] = y.z, Ro[8] = z.z;
    }
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *pt = points + ((size_t)b * n + i) * 3;
    const V3 d = {pt[0] - p_t[b * 3], pt[1] - p_t[b * 3 + 1], pt[2] - p_t[b * 3 + 2]};
    float *o = out + ((size_t)b * n + i) * 3;
    o[0] = dot3(x, d) * p_s[b * 3];       // R^T d: row k of R^T is column k of R
    o[1] = dot3(y, d) * p_s[b * 3 + 1];
    o[2] = dot3(z, d) * p_s[b * 3 + 2];
}

extern "C" int tgp_canonicalize(const float *points, const float *gR, const float *p_g, const float *f_g, const float *p_r,
                                const float *f_r, const float *p_t, const float *p_s, const float *sym, int sym_ld, int B,
                                int n, float *out, float *R_out, tgp_stream_t stream)
{
    TGP_REQUIRE(points && gR && p_g && f_g && p_r && f_r && p_t && p_s && sym && out && B > 0 && n > 0 && sym_ld > 0);
    hipLaunchKernelGGL(canonicalize_kernel, dim3(tgp_cdiv(n, 256), B), dim3(256), 0, tgp_hs(stream), points, gR, p_g, f_g, p_r,
                       f_r, p_t, p_s, sym, sym_ld, n, out, R_out);
    return TGP_LAUNCH_RESULT();
}

// generate_RT(mode='vec') of the evaluater (evaluater/RT_TDA_Evaluater.py:94; tools/geom_utils.generate_RT, whose
// source is missing from the reference tree: semantics from SURVEY.md 8c): the red confidence is zeroed for
// objects symmetric about y, R = to_R_matrices(f_green, f_red, p_green, p_red) (tools/rot_utils.py:95-98),
// RT = [[R, T], [0 0 0 1]].  One thread per object.
__global__ void generate_rt_kernel(const float *__restrict__ p_g, const float *__restrict__ p_r, const float *__restrict__ f_g,
                                   const float *__restrict__ f_r, const float *__restrict__ T, const float *__restrict__ sym,
                                   int sym_ld, int B, float *__restrict__ out)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const V3 yg = {p_g[b * 3], p_g[b * 3 + 1], p_g[b * 3 + 2]};
    const V3 xr = {p_r[b * 3], p_r[b * 3 + 1], p_r[b * 3 + 2]};
    const float fr = (sym && sym[(size_t)b * sym_ld] == 1.0f) ? 0.f : f_r[b];
    V3 ny, nx;
    vertical_axes(f_g[b], fr, yg, xr, ny, nx);
    const float yn = fmaxf(norm3(ny), 1e-12f);
    const V3 y = {ny.x / yn, ny.y / yn, ny.z / yn};
    V3 z = cross3(nx, y);
    const float zn = fmaxf(norm3(z), 1e-12f);
    z.x /= zn, z.y /= zn, z.z /= zn;
    const V3 x = cross3(y, z);
    float *o = out + (size_t)b * 16;
    o[0] = x.x, o[1] = y.x, o[2] = z.x, o[3] = T[b * 3];
    o[4] = x.y, o[5] = y.y, o[6] = z.y, o[7] = T[b * 3 + 1];
    o[8] = x.z, o[9] = y.z, o[10] = z.z, o[11] = T[b * 3 + 2];
    o[12] = 0.f, o[13] = 0.f, o[14] = 0.f, o[15] = 1.f;
}

extern "C" int tgp_generate_rt(const float *p_green, const float *p_red, const float *f_green, const float *f_red,
                               const float *T, const float *sym, int sym_ld, int B, float *rt, tgp_stream_t stream)
{
    TGP_REQUIRE(p_green && p_red && f_green && f_red && T && rt && B > 0 && (!sym || sym_ld > 0));
    hipLaunchKernelGGL(generate_rt_kernel, dim3(tgp_cdiv(B, 64)), dim3(64), 0, tgp_hs(stream), p_green, p_red, f_green, f_red, T,
                       sym, sym_ld, B, rt);
    return TGP_LAUNCH_RESULT();
}

// ---------------------------------------------------------------------------------------------------
// Differentiable half of R_DCD's canonicalisation (TDA_loss_sym_recon.py:334-337): out = (R^T (p - t)) * s per object,
// with R (B,3,3), t (B,3), s (B,3) given (the (B,3)-sized axis arithmetic that builds R stays with the caller).
__global__ __launch_bounds__(256) void pose_transform_kernel(const float *__restrict__ points, const float *__restrict__ R,
                                                             const float *__restrict__ t, const float *__restrict__ s, int n,
                                                             float *__restrict__ out)
{
    const int b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *Rb = R + b * 9;
    const float *pt = points + ((size_t)b * n + i) * 3;
    const float d0 = pt[0] - t[b * 3], d1 = pt[1] - t[b * 3 + 1], d2 = pt[2] - t[b * 3 + 2];
    float *o = out + ((size_t)b * n + i) * 3;
#pragma unroll
    for (int k = 0; k < 3; ++k) o[k] = ((Rb[k] * d0 + Rb[3 + k] * d1) + Rb[6 + k] * d2) * s[b * 3 + k];
}

extern "C" int tgp_pose_transform_fwd(const float *points, const float *R, const float *t, const float *s, int B, int n, float *out,
                                      tgp_stream_t stream)
{
    TGP_REQUIRE(points && R && t && s && out && B > 0 && n > 0);
    hipLaunchKernelGGL(pose_transform_kernel, dim3(tgp_cdiv(n, 256), B), dim3(256), 0, tgp_hs(stream), points, R, t, s, n, out);
    return TGP_LAUNCH_RESULT();
}

// backward: g = dout * s;  dpoints = R g;  dt = -sum_i R g_i;  dR[j][k] = sum_i d_j g_k;  ds_k = sum_i (R^T d)_k dout_k.
// One workgroup per object; the 15 sums are reduced in LDS in a fixed order (deterministic).
__global__ __launch_bounds__(256) void pose_transform_bwd_kernel(const float *__restrict__ points, const float *__restrict__ R,
                                                                 const float *__restrict__ t, const float *__restrict__ s,
                                                                 const float *__restrict__ dout, int n, float *__restrict__ dpoints,
                                                                 float *__restrict__ dR, float *__restrict__ dt, float *__restrict__ ds)
{
    __shared__ float red[15][256];
    const int b = blockIdx.x;
    const float *Rb = R + b * 9;
    float acc[15];
#pragma unroll
    for (int q = 0; q < 15; ++q) acc[q] = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        const size_t o = ((size_t)b * n + i) * 3;
        const float d[3] = {points[o] - t[b * 3], points[o + 1] - t[b * 3 + 1], points[o + 2] - t[b * 3 + 2]};
        float g[3], rd[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            g[k] = dout[o + k] * s[b * 3 + k];
            rd[k] = (Rb[k] * d[0] + Rb[3 + k] * d[1]) + Rb[6 + k] * d[2];
            acc[12 + k] += rd[k] * dout[o + k];
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float dp = (Rb[3 * j] * g[0] + Rb[3 * j + 1] * g[1]) + Rb[3 * j + 2] * g[2];
            if (dpoints) dpoints[o + j] = dp;
            acc[9 + j] -= dp;
#pragma unroll
            for (int k = 0; k < 3; ++k) acc[3 * j + k] += d[j] * g[k];
        }
    }
#pragma unroll
    for (int q = 0; q < 15; ++q) red[q][threadIdx.x] = acc[q];
    __syncthreads();
    if (threadIdx.x < 15) {
        float v = 0.f;
        for (int l = 0; l < 256; ++l) v += red[threadIdx.x][l];
        if (threadIdx.x < 9) dR[b * 9 + threadIdx.x] = v;
        else if (threadIdx.x < 12) dt[b * 3 + threadIdx.x - 9] = v;
        else ds[b * 3 + threadIdx.x - 12] = v;
    }
}

extern "C" int tgp_pose_transform_bwd(const float *points, const float *R, const float *t, const float *s, const float *dout, int B,
                                      int n, float *dpoints, float *dR, float *dt, float *ds, tgp_stream_t stream)
{
    TGP_REQUIRE(points && R && t && s && dout && dR && dt && ds && B > 0 && n > 0);
    hipLaunchKernelGGL(pose_transform_bwd_kernel, dim3(B), dim3(256), 0, tgp_hs(stream), points, R, t, s, dout, n, dpoints, dR, dt, ds);
    return TGP_LAUNCH_RESULT();
}
