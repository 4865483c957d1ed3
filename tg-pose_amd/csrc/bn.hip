// Training-mode BatchNorm1d over channel-last rows (nn.BatchNorm1d in train mode: FaceRecon.py:28-30,95-109,125-129,
// PoseR.py:22-24, PoseTs.py:24-26), gfx950.
//
//   mean_c = E[x_c],  var_c = E[(x_c - mean_c)^2]   (biased, two passes: no cancellation)
//   y = (x - mean) / sqrt(var + eps) * gamma + beta, then the layer's (Leaky)ReLU, optionally the max over each
//   object's points (order-preserving atomicMax keys, as in the GEMM epilogue).
// Statistics are reduced deterministically: a workgroup owns 64 columns x one chunk of rows, the chunk partials
// are summed in chunk order by a second tiny kernel.  x may be a column slice of a wider buffer (row stride ld).
#include "tgp_common.h"

#define BN_CHUNK 1024 // rows per workgroup

// partial[chunk][c] = sum over the chunk's rows of (x - center[c]) or (x - center[c])^2
template <bool SQUARE>
__global__ __launch_bounds__(256) void bn_partial_kernel(const float *__restrict__ x, int ld, int64_t rows, int C,
                                                         const float *__restrict__ center, float *__restrict__ partial)
{
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int slice = threadIdx.x >> 6;
    const int64_t r0 = (int64_t)blockIdx.y * BN_CHUNK;
    const int64_t r1 = r0 + BN_CHUNK < rows ? r0 + BN_CHUNK : rows;
    float acc = 0.f;
    if (c < C) {
        const float mu = center ? center[c] : 0.f;
#pragma unroll 8
    for (int64_t r = r0 + slice; r < r1; r += 4) {
            const float d = x[r * ld + c] - mu;
            acc += SQUARE ? d * d : d;
        }
    }
    red[slice][threadIdx.x & 63] = acc;
    __syncthreads();
    if (slice == 0 && c < C)
        partial[(int64_t)blockIdx.y * C + c] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}

__global__ void bn_finish_kernel(const float *__restrict__ partial, int chunks, int C, float inv_rows, float *__restrict__ out)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float s = 0.f;
    for (int k = 0; k < chunks; ++k) s += partial[(int64_t)k * C + c];
    out[c] = s * inv_rows;
}

extern "C" int64_t tgp_bn_workspace_floats(int64_t rows, int C)
{
    if (rows <= 0 || C <= 0) return 0;
    return (int64_t)tgp_cdiv(rows, BN_CHUNK) * C;
}

extern "C" int tgp_bn_stats(const float *x, int ld, int64_t rows, int C, float *mean, float *var, float *workspace,
                            tgp_stream_t stream)
{
    TGP_REQUIRE(x && mean && var && workspace && rows > 0 && C > 0 && ld >= C);
    const int chunks = tgp_cdiv(rows, BN_CHUNK);
    const dim3 grid(tgp_cdiv(C, 64), chunks), block(256);
    const float inv = (float)(1.0 / (double)rows);
    hipLaunchKernelGGL(bn_partial_kernel<false>, grid, block, 0, tgp_hs(stream), x, ld, rows, C, (const float *)nullptr, workspace);
    hipLaunchKernelGGL(bn_finish_kernel, dim3(tgp_cdiv(C, 256)), dim3(256), 0, tgp_hs(stream), workspace, chunks, C, inv, mean);
    hipLaunchKernelGGL(bn_partial_kernel<true>, grid, block, 0, tgp_hs(stream), x, ld, rows, C, (const float *)mean, workspace);
    hipLaunchKernelGGL(bn_finish_kernel, dim3(tgp_cdiv(C, 256)), dim3(256), 0, tgp_hs(stream), workspace, chunks, C, inv, var);
    return TGP_LAUNCH_RESULT();
}

__global__ __launch_bounds__(256) void bn_apply_kernel(const float *__restrict__ x, int ld, int64_t rows, int C,
                                                       const float *__restrict__ mean, const float *__restrict__ var,
                                                       const float *__restrict__ gamma, const float *__restrict__ beta, float eps,
                                                       int act, float slope, const float *__restrict__ slope_vec,
                                                       float *__restrict__ out, int ldo, uint32_t *__restrict__ cm, int ldcm,
                                                       int cm_cols, int rows_per_obj)
{
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int slice = threadIdx.x >> 6;
    if (c >= C) return;
    const float mu = mean[c];
    const float a = gamma[c] / sqrtf(var[c] + eps);
    const float b = beta[c];
    const float sl = slope_vec ? slope_vec[c] : slope;
    const int64_t r0 = (int64_t)blockIdx.y * BN_CHUNK;
    const int64_t r1 = r0 + BN_CHUNK < rows ? r0 + BN_CHUNK : rows;
    const bool do_cm = cm && c < cm_cols;
    uint32_t run_key = 0;
    int64_t run_obj = -1;
#pragma unroll 8
    for (int64_t r = r0 + slice; r < r1; r += 4) {
        float v = (x[r * ld + c] - mu) * a + b;
        if (act == 1) v = v > 0.f ? v : v * sl;
        if (out) out[r * ldo + c] = v;
        if (do_cm) {
            const int64_t obj = r / rows_per_obj;
            const uint32_t key = tgp_float_key(v);
            if (obj != run_obj) {
                if (run_obj >= 0) atomicMax(cm + run_obj * ldcm + c, run_key);
                run_obj = obj, run_key = key;
            } else {
                run_key = key > run_key ? key : run_key;
            }
        }
    }
    if (do_cm && run_obj >= 0) atomicMax(cm + run_obj * ldcm + c, run_key);
}

extern "C" int tgp_bn_apply(const float *x, int ld, int64_t rows, int C, const float *mean, const float *var,
                            const float *gamma, const float *beta, float eps, int act, float slope, const float *slope_vec,
                            float *out, int ldo, uint32_t *colmax_keys, int ldcm, int cm_cols, int rows_per_obj,
                            tgp_stream_t stream)
{
    TGP_REQUIRE(x && mean && var && gamma && beta && rows > 0 && C > 0 && ld >= C && (out || colmax_keys));
    TGP_REQUIRE(!out || ldo >= C);
    TGP_REQUIRE(!colmax_keys || (rows_per_obj > 0 && ldcm > 0));
    hipLaunchKernelGGL(bn_apply_kernel, dim3(tgp_cdiv(C, 64), tgp_cdiv(rows, BN_CHUNK)), dim3(256), 0, tgp_hs(stream), x, ld,
                       rows, C, mean, var, gamma, beta, eps, act, slope, slope_vec, out, ldo, colmax_keys, ldcm,
                       cm_cols > 0 ? cm_cols : C, rows_per_obj > 0 ? rows_per_obj : 1);
    return TGP_LAUNCH_RESULT();
}

// nn.Dropout in train mode on a small per-object tensor: y = x * keep / (1 - p), keep in {0,1} (uint8, drawn by the host
// framework's generator)
__global__ void dropout_apply_kernel(const float *__restrict__ x, const uint8_t *__restrict__ keep, float scale, int64_t count,
                                     float *__restrict__ y)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < count) y[t] = keep[t] ? x[t] * scale : 0.f;
}

extern "C" int tgp_dropout_apply(const float *x, const uint8_t *keep, float p, int64_t count, float *y, tgp_stream_t stream)
{
    TGP_REQUIRE(x && keep && y && count > 0 && p >= 0.f && p < 1.f);
    hipLaunchKernelGGL(dropout_apply_kernel, dim3(tgp_cdiv(count, 256)), dim3(256), 0, tgp_hs(stream), x, keep,
                       1.0f / (1.0f - p), count, y);
    return TGP_LAUNCH_RESULT();
}
