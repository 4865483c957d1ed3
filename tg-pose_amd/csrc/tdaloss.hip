// The pose-regression loss bundle either side of Chamfer (losses/TDA_loss_sym_recon.py:39-311, losses/consistency_loss.py).
// The reference walks the batch in Python, one `if sym[i, 0] == 1` host synchronisation and a handful of tiny launches per
// object and per term; here every term of a batch comes out of one launch with the symmetry tests made on the device, so the
// whole loss can sit inside the captured training graph.  All of this is latency-sized work (B x 3 values, or B x N x 3
// for the symmetric reconstruction term); there is no roofline to chase, only launches and host round trips to remove.
#include "tgp_common.h"

#define TL_THREADS 256

// elementwise penalty of nn.L1Loss (kind 0) / nn.SmoothL1Loss(beta) (kind 1) and its derivative
__device__ __forceinline__ float tl_rho(float x, int kind, float beta)
{
    const float a = fabsf(x);
    return (kind == 0 || a >= beta) ? (kind == 0 ? a : a - 0.5f * beta) : 0.5f * x * x / beta;
}
__device__ __forceinline__ float tl_sign(float x) { return (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f); }
__device__ __forceinline__ float tl_drho(float x, int kind, float beta)
{
    return (kind == 0 || fabsf(x) >= beta) ? tl_sign(x) : x / beta;
}

// deterministic block sum of NV values per thread (fixed tree, result valid in thread 0)
template <int NV>
__device__ __forceinline__ void tl_block_sum(float (&v)[NV], float *lds)
{
    const int tid = threadIdx.x;
#pragma unroll
    for (int k = 0; k < NV; ++k) lds[k * TL_THREADS + tid] = v[k];
    __syncthreads();
    for (int s = TL_THREADS / 2; s > 0; s >>= 1) {
        if (tid < s)
#pragma unroll
            for (int k = 0; k < NV; ++k) lds[k * TL_THREADS + tid] += lds[k * TL_THREADS + tid + s];
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = lds[k * TL_THREADS];
    __syncthreads();
}

struct tl_pose_in {
    const float *rot1, *rot2, *f1, *f2, *tran, *size;          // predictions: (B,3) (B,3) (B) (B) (B,3) (B,3)
    const float *g_rot1, *g_rot2, *g_tran, *g_size;            // targets
    const int *sym;                                            // (B, sym_ld), column 0 = symmetric about y
    int sym_ld, B, kind;
    float beta;
};

// out[0..7] = Rot1, Rot1_cos, Rot2, Rot2_cos, Rot_regular, Tran, Size, R_con (unweighted); out[8] = objects with sym0 != 1
__global__ __launch_bounds__(TL_THREADS) void pose_terms_fwd_kernel(tl_pose_in p, float *__restrict__ out)
{
    __shared__ float lds[10 * TL_THREADS];
    float acc[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) acc[k] = 0.f;
    for (int b = threadIdx.x; b < p.B; b += TL_THREADS) {
        const int s0 = p.sym[(size_t)b * p.sym_ld];
        float l1 = 0, l2 = 0, lt = 0, ls = 0, d1 = 0, d2 = 0, d12 = 0, ng = 0, nr = 0;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float a1 = p.rot1[b * 3 + c], a2 = p.rot2[b * 3 + c], g1 = p.g_rot1[b * 3 + c], g2 = p.g_rot2[b * 3 + c];
            l1 += tl_rho(a1 - g1, p.kind, p.beta), l2 += tl_rho(a2 - g2, p.kind, p.beta);
            lt += tl_rho(p.tran[b * 3 + c] - p.g_tran[b * 3 + c], p.kind, p.beta);
            ls += tl_rho(p.size[b * 3 + c] - p.g_size[b * 3 + c], p.kind, p.beta);
            d1 += a1 * g1, d2 += a2 * g2, d12 += a1 * a2;
            ng += (a1 - g1) * (a1 - g1), nr += (a2 - g2) * (a2 - g2);
        }
        const float valid = (s0 != 1) ? 1.f : 0.f;                        // cal_loss_Rot2 / cosine_dis_sym / rot_regular skip sym0 == 1
        acc[0] += l1;                                                     // Rot1: mean over B*3
        acc[1] += (1.f - d1) * 2.f;                                       // Rot1_cos: mean over B
        acc[2] += valid * (l2 / 3.f);                                     // Rot2: per-object mean, averaged over the valid ones
        acc[3] += valid * ((1.f - d2) * 2.f);
        acc[4] += valid * fabsf(d12);
        acc[5] += lt, acc[6] += ls;
        // cal_loss_R_con (:205-221): target confidence exp(-13.7 |dv|^2) against the predicted one; the red-axis part only
        // where sym0 == 0, but divided by the whole batch
        const float sg = sqrtf(ng), sr = sqrtf(nr);
        acc[7] += tl_rho(expf(-13.7f * sg * sg) - p.f1[b], p.kind, p.beta);
        acc[8] += (s0 == 0) ? tl_rho(expf(-13.7f * sr * sr) - p.f2[b], p.kind, p.beta) : 0.f;
        acc[9] += valid;
    }
    tl_block_sum<10>(acc, lds);
    if (threadIdx.x == 0) {
        const float B = (float)p.B, nv = acc[9], dv = nv > 0.f ? nv : 1.f;
        out[0] = acc[0] / (3.f * B), out[1] = acc[1] / B;
        out[2] = acc[2] / dv, out[3] = acc[3] / dv, out[4] = acc[4] / dv;
        out[5] = acc[5] / (3.f * B), out[6] = acc[6] / (3.f * B);
        out[7] = acc[7] / B + acc[8] / B;
        out[8] = nv;
    }
}

// gradients of sum_k gw[k] * out[k] w.r.t. the six predictions; one thread per object
__global__ __launch_bounds__(TL_THREADS) void pose_terms_bwd_kernel(tl_pose_in p, const float *__restrict__ fwd_out,
                                                                    const float *__restrict__ gw, float *__restrict__ d_rot1,
                                                                    float *__restrict__ d_rot2, float *__restrict__ d_f1,
                                                                    float *__restrict__ d_f2, float *__restrict__ d_tran,
                                                                    float *__restrict__ d_size)
{
    const int b = blockIdx.x * TL_THREADS + threadIdx.x;
    if (b >= p.B) return;
    const float B = (float)p.B, nv = fwd_out[8], dv = nv > 0.f ? nv : 1.f;
    const int s0 = p.sym[(size_t)b * p.sym_ld];
    const float valid = (s0 != 1) ? 1.f : 0.f;
    float a1[3], a2[3], g1[3], g2[3];
    float d12 = 0, ng = 0, nr = 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        a1[c] = p.rot1[b * 3 + c], a2[c] = p.rot2[b * 3 + c], g1[c] = p.g_rot1[b * 3 + c], g2[c] = p.g_rot2[b * 3 + c];
        d12 += a1[c] * a2[c];
        ng += (a1[c] - g1[c]) * (a1[c] - g1[c]), nr += (a2[c] - g2[c]) * (a2[c] - g2[c]);
    }
    const float sg = sqrtf(ng), sr = sqrtf(nr);
    const float eg = expf(-13.7f * sg * sg), er = expf(-13.7f * sr * sr);
    const float cg = gw[7] * tl_drho(eg - p.f1[b], p.kind, p.beta) / B;
    const float cr = (s0 == 0) ? gw[7] * tl_drho(er - p.f2[b], p.kind, p.beta) / B : 0.f;
    d_f1[b] = -cg, d_f2[b] = -cr;
    const float sreg = gw[4] * valid * tl_sign(d12) / dv;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        d_rot1[b * 3 + c] = gw[0] * tl_drho(a1[c] - g1[c], p.kind, p.beta) / (3.f * B) - gw[1] * 2.f * g1[c] / B + sreg * a2[c] +
                            cg * eg * -27.4f * (a1[c] - g1[c]);
        d_rot2[b * 3 + c] = valid * (gw[2] * tl_drho(a2[c] - g2[c], p.kind, p.beta) / (3.f * dv) - gw[3] * 2.f * g2[c] / dv) +
                            sreg * a1[c] + cr * er * -27.4f * (a2[c] - g2[c]);
        d_tran[b * 3 + c] = gw[5] * tl_drho(p.tran[b * 3 + c] - p.g_tran[b * 3 + c], p.kind, p.beta) / (3.f * B);
        d_size[b * 3 + c] = gw[6] * tl_drho(p.size[b * 3 + c] - p.g_size[b * 3 + c], p.kind, p.beta) / (3.f * B);
    }
}

static bool tl_pose_ok(const tl_pose_in &p)
{
    return p.rot1 && p.rot2 && p.f1 && p.f2 && p.tran && p.size && p.g_rot1 && p.g_rot2 && p.g_tran && p.g_size && p.sym && p.B > 0 &&
           p.sym_ld >= 1 && (p.kind == 0 || (p.kind == 1 && p.beta > 0.f));
}

extern "C" int tgp_pose_terms_fwd(const float *rot1, const float *rot2, const float *f1, const float *f2, const float *tran,
                                  const float *size, const float *g_rot1, const float *g_rot2, const float *g_tran,
                                  const float *g_size, const int32_t *sym, int sym_ld, int B, int kind, float beta, float *out,
                                  tgp_stream_t stream)
{
    const tl_pose_in p{rot1, rot2, f1, f2, tran, size, g_rot1, g_rot2, g_tran, g_size, sym, sym_ld, B, kind, beta};
    TGP_REQUIRE(tl_pose_ok(p) && out);
    hipLaunchKernelGGL(pose_terms_fwd_kernel, dim3(1), dim3(TL_THREADS), 0, tgp_hs(stream), p, out);
    return TGP_LAUNCH_RESULT();
}

extern "C" int tgp_pose_terms_bwd(const float *rot1, const float *rot2, const float *f1, const float *f2, const float *tran,
                                  const float *size, const float *g_rot1, const float *g_rot2, const float *g_tran,
                                  const float *g_size, const int32_t *sym, int sym_ld, int B, int kind, float beta,
                                  const float *fwd_out, const float *gw, float *d_rot1, float *d_rot2, float *d_f1, float *d_f2,
                                  float *d_tran, float *d_size, tgp_stream_t stream)
{
    const tl_pose_in p{rot1, rot2, f1, f2, tran, size, g_rot1, g_rot2, g_tran, g_size, sym, sym_ld, B, kind, beta};
    TGP_REQUIRE(tl_pose_ok(p) && fwd_out && gw && d_rot1 && d_rot2 && d_f1 && d_f2 && d_tran && d_size);
    hipLaunchKernelGGL(pose_terms_bwd_kernel, dim3(tgp_cdiv(B, TL_THREADS)), dim3(TL_THREADS), 0, tgp_hs(stream), p, fwd_out, gw, d_rot1,
                       d_rot2, d_f1, d_f2, d_tran, d_size);
    return TGP_LAUNCH_RESULT();
}

// ---- prop_sym_matching_loss (losses/consistency_loss.py:19-81, TDA_loss_sym_recon.py:120-203) -------------------------------
// The target cloud is the input cloud mapped by the object's own symmetry: a half turn about y for bottle / bowl / can,
// the mirror z -> -z for laptop / mug, identity otherwise, all in the ground-truth object frame; L1 against the
// reconstruction, mean over B*N*3.  mode per object: 0 identity, 1 half turn (diag(-1,1,-1)), 2 mirror (diag(1,1,-1)),
// 3 neither flag set and the reconstruction kept (sym0 outside {0,1}), 4 both sides zero (sym0 == 1, no other flag).
__device__ __forceinline__ int tl_sym_mode(const int *sym, int sym_ld, int ncol)
{
    const int s0 = sym[0];
    int rest = 0;
    for (int c = 1; c < ncol; ++c) rest += sym[c];
    if (s0 == 1) return rest > 0 ? 1 : (rest == 0 ? 4 : 3);
    if (s0 == 0) return (ncol > 1 && sym[1] == 1) ? 2 : 0;
    return 3;
}

// target = R D R^T (p - t) + t, evaluated in the reference's order (canonicalise, flip, pose)
__device__ __forceinline__ void tl_sym_target(const float *R, const float *t, const float (&pt)[3], int mode, float (&tg)[3])
{
    if (mode == 0) {
        tg[0] = pt[0], tg[1] = pt[1], tg[2] = pt[2];
        return;
    }
    if (mode >= 3) {
        tg[0] = tg[1] = tg[2] = 0.f;
        return;
    }
    const float q[3] = {pt[0] - t[0], pt[1] - t[1], pt[2] - t[2]};
    float cn[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) cn[j] = (R[0 * 3 + j] * q[0] + R[1 * 3 + j] * q[1]) + R[2 * 3 + j] * q[2];
    cn[2] = -cn[2];
    if (mode == 1) cn[0] = -cn[0];
#pragma unroll
    for (int i = 0; i < 3; ++i) tg[i] = ((R[i * 3 + 0] * cn[0] + R[i * 3 + 1] * cn[1]) + R[i * 3 + 2] * cn[2]) + t[i];
}

__global__ __launch_bounds__(TL_THREADS) void sym_recon_fwd_kernel(const float *__restrict__ PC, const float *__restrict__ PC_re,
                                                                   const float *__restrict__ gR, const float *__restrict__ gt,
                                                                   const int *__restrict__ sym, int sym_ld, int sym_cols, int B, int N,
                                                                   float *__restrict__ partial)
{
    __shared__ float lds[TL_THREADS];
    const int b = blockIdx.y;
    const int mode = tl_sym_mode(sym + (size_t)b * sym_ld, sym_ld, sym_cols);
    float acc[1] = {0.f};
    for (int i = blockIdx.x * TL_THREADS + threadIdx.x; i < N; i += gridDim.x * TL_THREADS) {
        const size_t o = ((size_t)b * N + i) * 3;
        const float pt[3] = {PC[o], PC[o + 1], PC[o + 2]};
        float tg[3];
        tl_sym_target(gR + b * 9, gt + b * 3, pt, mode, tg);
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[0] += fabsf(tg[c] - (mode == 4 ? 0.f : PC_re[o + c]));
    }
    tl_block_sum<1>(acc, lds);
    if (threadIdx.x == 0) partial[blockIdx.y * gridDim.x + blockIdx.x] = acc[0];
}

// out[0] = scale * sum(partial[0..n))   (one workgroup, fixed order)
__global__ __launch_bounds__(TL_THREADS) void tl_final_sum_kernel(const float *__restrict__ partial, int n, float scale, float *__restrict__ out)
{
    __shared__ float lds[TL_THREADS];
    float acc[1] = {0.f};
    for (int i = threadIdx.x; i < n; i += TL_THREADS) acc[0] += partial[i];
    tl_block_sum<1>(acc, lds);
    if (threadIdx.x == 0) out[0] = acc[0] * scale;
}

__global__ __launch_bounds__(TL_THREADS) void sym_recon_bwd_kernel(const float *__restrict__ PC, const float *__restrict__ PC_re,
                                                                   const float *__restrict__ gR, const float *__restrict__ gt,
                                                                   const int *__restrict__ sym, int sym_ld, int sym_cols, int B, int N,
                                                                   const float *__restrict__ gout, float *__restrict__ dPC,
                                                                   float *__restrict__ dPC_re)
{
    const int b = blockIdx.y, i = blockIdx.x * TL_THREADS + threadIdx.x;
    if (i >= N) return;
    const int mode = tl_sym_mode(sym + (size_t)b * sym_ld, sym_ld, sym_cols);
    const float g = gout[0] / (3.f * (float)B * (float)N);
    const size_t o = ((size_t)b * N + i) * 3;
    const float pt[3] = {PC[o], PC[o + 1], PC[o + 2]};
    float tg[3], sg[3];
    const float *R = gR + b * 9;
    tl_sym_target(R, gt + b * 3, pt, mode, tg);
#pragma unroll
    for (int c = 0; c < 3; ++c) sg[c] = g * tl_sign(tg[c] - (mode == 4 ? 0.f : PC_re[o + c]));
    if (dPC_re)
#pragma unroll
        for (int c = 0; c < 3; ++c) dPC_re[o + c] = (mode == 4) ? 0.f : -sg[c];
    if (dPC) {
        if (mode == 0) {
            dPC[o] = sg[0], dPC[o + 1] = sg[1], dPC[o + 2] = sg[2];
        } else if (mode >= 3) {
            dPC[o] = dPC[o + 1] = dPC[o + 2] = 0.f;
        } else {                                            // (R D R^T)^T sg = R D R^T sg
            float cn[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) cn[j] = (R[0 * 3 + j] * sg[0] + R[1 * 3 + j] * sg[1]) + R[2 * 3 + j] * sg[2];
            cn[2] = -cn[2];
            if (mode == 1) cn[0] = -cn[0];
#pragma unroll
            for (int c = 0; c < 3; ++c) dPC[o + c] = (R[c * 3 + 0] * cn[0] + R[c * 3 + 1] * cn[1]) + R[c * 3 + 2] * cn[2];
        }
    }
}

static int tl_sym_blocks(int N) { return N >= 4 * TL_THREADS ? 4 : tgp_cdiv(N, TL_THREADS); }

extern "C" int64_t tgp_sym_recon_workspace_floats(int B, int N) { return (int64_t)B * tl_sym_blocks(N); }

extern "C" int tgp_sym_recon_fwd(const float *PC, const float *PC_re, const float *gt_R, const float *gt_t, const int32_t *sym, int sym_ld,
                                 int sym_cols, int B, int N, float *workspace, float *loss, tgp_stream_t stream)
{
    TGP_REQUIRE(PC && PC_re && gt_R && gt_t && sym && workspace && loss && B > 0 && B <= 65535 && N > 0 && sym_cols >= 1 &&
                sym_ld >= sym_cols);
    const int nb = tl_sym_blocks(N);
    hipLaunchKernelGGL(sym_recon_fwd_kernel, dim3(nb, B), dim3(TL_THREADS), 0, tgp_hs(stream), PC, PC_re, gt_R, gt_t, sym, sym_ld, sym_cols,
                       B, N, workspace);
    hipLaunchKernelGGL(tl_final_sum_kernel, dim3(1), dim3(TL_THREADS), 0, tgp_hs(stream), workspace, nb * B,
                       1.f / (3.f * (float)B * (float)N), loss);
    return TGP_LAUNCH_RESULT();
}

extern "C" int tgp_sym_recon_bwd(const float *PC, const float *PC_re, const float *gt_R, const float *gt_t, const int32_t *sym, int sym_ld,
                                 int sym_cols, int B, int N, const float *gloss, float *dPC, float *dPC_re, tgp_stream_t stream)
{
    TGP_REQUIRE(PC && PC_re && gt_R && gt_t && sym && gloss && (dPC || dPC_re) && B > 0 && B <= 65535 && N > 0 && sym_cols >= 1 &&
                sym_ld >= sym_cols);
    hipLaunchKernelGGL(sym_recon_bwd_kernel, dim3(tgp_cdiv(N, TL_THREADS), B), dim3(TL_THREADS), 0, tgp_hs(stream), PC, PC_re, gt_R, gt_t, sym,
                       sym_ld, sym_cols, B, N, gloss, dPC, dPC_re);
    return TGP_LAUNCH_RESULT();
}

// ---- ph_loss_fn / omega (TDA_loss_sym_recon.py:292-322): mean(|a - b| * w) with w = 1 on rows whose sum(wsrc) > 0 ------------
// The reference tests both operands for NaN / Inf on the host and then computes |a - a| instead; on the device that is:
// NaN when a holds a NaN / Inf (|x - x| is NaN there), 0 when only b does.  rows[b] = {sum |a-b|, weight, bad(a), bad(b)}.
__global__ __launch_bounds__(TL_THREADS) void rowl1_rows_kernel(const float *__restrict__ a, const float *__restrict__ bsrc,
                                                                const float *__restrict__ wsrc, int D, float *__restrict__ rows)
{
    __shared__ float lds[4 * TL_THREADS];
    const size_t o = (size_t)blockIdx.x * D;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = threadIdx.x; i < D; i += TL_THREADS) {
        const float x = a[o + i], y = bsrc[o + i];
        const bool bx = !(fabsf(x) <= 3.402823466e38f), by = !(fabsf(y) <= 3.402823466e38f);
        acc[0] += (bx || by) ? 0.f : fabsf(x - y);
        acc[1] += wsrc[o + i];
        acc[2] += bx ? 1.f : 0.f, acc[3] += by ? 1.f : 0.f;
    }
    tl_block_sum<4>(acc, lds);
    if (threadIdx.x == 0) {
        float *r = rows + (size_t)blockIdx.x * 4;
        r[0] = acc[0], r[1] = (acc[1] > 0.f) ? 1.f : 0.f, r[2] = acc[2], r[3] = acc[3];
    }
}

// out[0] = loss, out[1] = 1 when the loss is the plain weighted mean (the backward is live), else 0
__global__ __launch_bounds__(TL_THREADS) void rowl1_final_kernel(const float *__restrict__ rows, int B, int D, float *__restrict__ out)
{
    __shared__ float lds[3 * TL_THREADS];
    float acc[3] = {0.f, 0.f, 0.f};
    for (int b = threadIdx.x; b < B; b += TL_THREADS) {
        acc[0] += rows[b * 4] * rows[b * 4 + 1];
        acc[1] += rows[b * 4 + 2], acc[2] += rows[b * 4 + 3];
    }
    tl_block_sum<3>(acc, lds);
    if (threadIdx.x == 0) {
        const bool bad_a = acc[1] > 0.f, bad_b = acc[2] > 0.f;
        out[0] = bad_a ? __uint_as_float(0x7fc00000u) : (bad_b ? 0.f : acc[0] / ((float)B * (float)D));
        out[1] = (bad_a || bad_b) ? 0.f : 1.f;
    }
}

__global__ __launch_bounds__(TL_THREADS) void rowl1_bwd_kernel(const float *__restrict__ a, const float *__restrict__ bsrc,
                                                               const float *__restrict__ rows, const float *__restrict__ fwd_out,
                                                               const float *__restrict__ gout, int B, int D, float *__restrict__ da)
{
    const size_t o = (size_t)blockIdx.x * D;
    const float g = fwd_out[1] * rows[(size_t)blockIdx.x * 4 + 1] * gout[0] / ((float)B * (float)D);
    for (int i = threadIdx.x; i < D; i += TL_THREADS) da[o + i] = g * tl_sign(a[o + i] - bsrc[o + i]);
}

extern "C" int tgp_rowl1_fwd(const float *a, const float *b, const float *wsrc, int B, int D, float *rows, float *out, tgp_stream_t stream)
{
    TGP_REQUIRE(a && b && wsrc && rows && out && B > 0 && D > 0);
    hipLaunchKernelGGL(rowl1_rows_kernel, dim3(B), dim3(TL_THREADS), 0, tgp_hs(stream), a, b, wsrc, D, rows);
    hipLaunchKernelGGL(rowl1_final_kernel, dim3(1), dim3(TL_THREADS), 0, tgp_hs(stream), rows, B, D, out);
    return TGP_LAUNCH_RESULT();
}

extern "C" int tgp_rowl1_bwd(const float *a, const float *b, const float *rows, const float *fwd_out, const float *gout, int B, int D,
                             float *da, tgp_stream_t stream)
{
    TGP_REQUIRE(a && b && rows && fwd_out && gout && da && B > 0 && D > 0);
    hipLaunchKernelGGL(rowl1_bwd_kernel, dim3(B), dim3(TL_THREADS), 0, tgp_hs(stream), a, b, rows, fwd_out, gout, B, D, da);
    return TGP_LAUNCH_RESULT();
}

// ---- feat_consistency_loss (losses/consistency_loss.py:11-16): 2 - 2 sum_b <x1/|x1|, x2/|x2|> / B -------------------------
// rows[b] = {<x1,x2>, max(|x1|, eps), max(|x2|, eps)} (F.normalize's eps = 1e-12), kept for the backward
__global__ __launch_bounds__(TL_THREADS) void cosrows_kernel(const float *__restrict__ x1, const float *__restrict__ x2, int C,
                                                             float *__restrict__ rows)
{
    __shared__ float lds[3 * TL_THREADS];
    const size_t o = (size_t)blockIdx.x * C;
    float acc[3] = {0.f, 0.f, 0.f};
    for (int i = threadIdx.x; i < C; i += TL_THREADS) {
        const float u = x1[o + i], v = x2[o + i];
        acc[0] += u * v, acc[1] += u * u, acc[2] += v * v;
    }
    tl_block_sum<3>(acc, lds);
    if (threadIdx.x == 0) {
        float *r = rows + (size_t)blockIdx.x * 3;
        r[0] = acc[0], r[1] = fmaxf(sqrtf(acc[1]), 1e-12f), r[2] = fmaxf(sqrtf(acc[2]), 1e-12f);
    }
}

__global__ __launch_bounds__(TL_THREADS) void cos_final_kernel(const float *__restrict__ rows, int B, float *__restrict__ out)
{
    __shared__ float lds[TL_THREADS];
    float acc[1] = {0.f};
    for (int b = threadIdx.x; b < B; b += TL_THREADS) acc[0] += rows[b * 3] / (rows[b * 3 + 1] * rows[b * 3 + 2]);
    tl_block_sum<1>(acc, lds);
    if (threadIdx.x == 0) out[0] = 2.f - 2.f * acc[0] / (float)B;
}

// d/dx1 of <x1,x2>/(n1 n2) = x2/(n1 n2) - <x1,x2> x1/(n1^3 n2) where n1 is not clamped, x2/(n1 n2) where it is (likewise x2)
__global__ __launch_bounds__(TL_THREADS) void cos_bwd_kernel(const float *__restrict__ x1, const float *__restrict__ x2,
                                                             const float *__restrict__ rows, const float *__restrict__ gout, int B, int C,
                                                             float *__restrict__ d1, float *__restrict__ d2)
{
    const size_t o = (size_t)blockIdx.x * C;
    const float *r = rows + (size_t)blockIdx.x * 3;
    const float dot = r[0], n1 = r[1], n2 = r[2], g = -2.f * gout[0] / (float)B, inv = 1.f / (n1 * n2);
    const float k1 = (n1 > 1e-12f) ? dot * inv / (n1 * n1) : 0.f, k2 = (n2 > 1e-12f) ? dot * inv / (n2 * n2) : 0.f;
    for (int i = threadIdx.x; i < C; i += TL_THREADS) {
        const float u = x1[o + i], v = x2[o + i];
        if (d1) d1[o + i] = g * (v * inv - k1 * u);
        if (d2) d2[o + i] = g * (u * inv - k2 * v);
    }
}

extern "C" int tgp_feat_consistency_fwd(const float *x1, const float *x2, int B, int C, float *rows, float *loss, tgp_stream_t stream)
{
    TGP_REQUIRE(x1 && x2 && rows && loss && B > 0 && C > 0);
    hipLaunchKernelGGL(cosrows_kernel, dim3(B), dim3(TL_THREADS), 0, tgp_hs(stream), x1, x2, C, rows);
    hipLaunchKernelGGL(cos_final_kernel, dim3(1), dim3(TL_THREADS), 0, tgp_hs(stream), rows, B, loss);
    return TGP_LAUNCH_RESULT();
}

extern "C" int tgp_feat_consistency_bwd(const float *x1, const float *x2, const float *rows, const float *gloss, int B, int C, float *d1,
                                        float *d2, tgp_stream_t stream)
{
    TGP_REQUIRE(x1 && x2 && rows && gloss && (d1 || d2) && B > 0 && C > 0);
    hipLaunchKernelGGL(cos_bwd_kernel, dim3(B), dim3(TL_THREADS), 0, tgp_hs(stream), x1, x2, rows, gloss, B, C, d1, d2);
    return TGP_LAUNCH_RESULT();
}
