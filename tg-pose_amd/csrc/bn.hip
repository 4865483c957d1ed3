// Training-mode BatchNorm1d over channel-last rows (nn.BatchNorm1d in train mode: FaceRecon.py:28-30,95-109,125-129,
// PoseR.py:22-24, PoseTs.py:24-26), gfx950.
//
//   mean_c = E[x_c],  var_c = E[(x_c - mean_c)^2]   (biased; one pass of shifted sums when the rows are 16-byte addressable --
//   "round 3" below -- otherwise two passes)
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

// ---- round 3: the statistics in ONE pass of 16-byte loads --------------------------------------------------------------------
// Round 2 read the activation twice for the statistics (mean, then the centred squares) with 4-byte loads per lane: the three
// BatchNorm passes of the trainer's step ran at ~1.8 TB/s.  Here a workgroup owns 256 columns x one chunk of BNV_CHUNK rows,
// a thread a column quad (float4 loads: 1 KB per wave instruction) and every fourth row; it accumulates the sums of d and d^2 with
// d = x - K, K = the chunk's first row -- shifted sums: the cancellation in S2 - S1^2 / n is relative to (mean - K)^2 ~ var,
// not to mean^2, so a single pass stays well conditioned however far the column's mean is from zero.  The chunks' (n, mean, M2)
// are merged IN CHUNK ORDER by one thread per column with the parallel-variance update (Chan et al.): deterministic,
// bit-repeatable, no float atomics; agrees with the two-pass values to ~1e-7 relative (test_bn_train_kernels_vs_torch).
#define BNV_CHUNK 256      // rows per workgroup: enough workgroups (C / 256 x rows / 256) to hide HBM latency -- with 1024-row chunks the
                           // 16-byte kernels were SLOWER than the 4-byte ones (a quarter of the workgroups: 8 waves per CU)

__global__ __launch_bounds__(256) void bn_stats_partial_v4_kernel(const float *__restrict__ x, int ld, int64_t rows, int C,
                                                                  float *__restrict__ partial /* [chunks][3][C]: S1, S2, K */)
{
    __shared__ float4 red[2][4][64];
    const int quad = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const int c = blockIdx.x * 256 + quad * 4;
    const int64_t r0 = (int64_t)blockIdx.y * BNV_CHUNK;
    const int64_t r1 = r0 + BNV_CHUNK < rows ? r0 + BNV_CHUNK : rows;
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1, K = s1;
    if (c < C) {
        K = *reinterpret_cast<const float4 *>(x + r0 * ld + c);
#pragma unroll 4
        for (int64_t r = r0 + slice; r < r1; r += 4) {
            const float4 v = *reinterpret_cast<const float4 *>(x + r * ld + c);
            const float dx = v.x - K.x, dy = v.y - K.y, dz = v.z - K.z, dw = v.w - K.w;
            s1.x += dx, s1.y += dy, s1.z += dz, s1.w += dw;
            s2.x += dx * dx, s2.y += dy * dy, s2.z += dz * dz, s2.w += dw * dw;
        }
    }
    red[0][slice][quad] = s1, red[1][slice][quad] = s2;
    __syncthreads();
    if (slice == 0 && c < C) {
        float4 a = red[0][0][quad], b = red[1][0][quad];
#pragma unroll
        for (int k = 1; k < 4; ++k) {                      // fixed order
            const float4 u = red[0][k][quad], w = red[1][k][quad];
            a.x += u.x, a.y += u.y, a.z += u.z, a.w += u.w;
            b.x += w.x, b.y += w.y, b.z += w.z, b.w += w.w;
        }
        float *p = partial + (int64_t)blockIdx.y * 3 * C + c;
        *reinterpret_cast<float4 *>(p) = a;
        *reinterpret_cast<float4 *>(p + C) = b;
        *reinterpret_cast<float4 *>(p + 2 * C) = K;
    }
}

// 64 columns x 4 chunk groups per workgroup: group g merges the chunks [g * per, (g + 1) * per) in order, the four group results
// are merged in group order (a fixed tree: deterministic)
// run_mean / run_var / batches (may be NULL): nn.BatchNorm1d's buffers, moved here instead of by four tiny torch launches per
// module and step -- running <- (1 - m) running + m batch, the variance with the unbiased factor rows / (rows - 1);
// num_batches_tracked += 1.  (Two roundings per buffer, as `running.mul_(1 - m).add_(batch, alpha = m)` has, but the factor is
// folded as (m * unbias) * var where torch forms the unbiased variance first: the running variance can differ from torch's sequence
// in the last bit -- tests hold the buffers to 1e-6.)
__global__ __launch_bounds__(256) void bn_stats_finish_v4_kernel(const float *__restrict__ partial, int chunks, int C, int64_t rows,
                                                                 float *__restrict__ mean, float *__restrict__ var,
                                                                 float *__restrict__ run_mean, float *__restrict__ run_var, float momentum,
                                                                 float unbias, int64_t *__restrict__ batches)
{
    __shared__ float s_n[4][64], s_mean[4][64], s_m2[4][64];
    const int l = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + l;
    const int per = (chunks + 3) / 4;
    float n_a = 0.f, mean_a = 0.f, m2_a = 0.f;
    if (c < C) {
        const int k1 = (g + 1) * per < chunks ? (g + 1) * per : chunks;
#pragma unroll 4
        for (int k = g * per; k < k1; ++k) {
            const float *p = partial + (int64_t)k * 3 * C + c;
            const int64_t left = rows - (int64_t)k * BNV_CHUNK;
            const float n_b = (float)(left < BNV_CHUNK ? left : BNV_CHUNK);
            const float s1 = p[0], s2 = p[C], K = p[2 * C];
            const float mean_b = K + s1 / n_b, m2_b = s2 - s1 * (s1 / n_b);
            const float n = n_a + n_b, delta = mean_b - mean_a;
            mean_a = mean_a + delta * (n_b / n);
            m2_a = (m2_a + m2_b) + delta * delta * (n_a * (n_b / n));
            n_a = n;
        }
    }
    s_n[g][l] = n_a, s_mean[g][l] = mean_a, s_m2[g][l] = m2_a;
    __syncthreads();
    if (g == 0 && c < C) {
#pragma unroll
        for (int q = 1; q < 4; ++q) {
            const float n_b = s_n[q][l];
            if (n_b > 0.f) {
                const float mean_b = s_mean[q][l], m2_b = s_m2[q][l];
                const float n = n_a + n_b, delta = mean_b - mean_a;
                mean_a = mean_a + delta * (n_b / n);
                m2_a = (m2_a + m2_b) + delta * delta * (n_a * (n_b / n));
                n_a = n;
            }
        }
        mean[c] = mean_a;
        float v = m2_a / n_a;
        v = v < 0.f ? 0.f : v;                               // (a NaN stays a NaN)
        var[c] = v;
        if (run_mean) {
            const float keep = 1.0f - momentum;
            run_mean[c] = run_mean[c] * keep + momentum * mean_a;
            run_var[c] = run_var[c] * keep + (momentum * unbias) * v;
        }
        if (batches && c == 0) batches[0] += 1;
    }
}

// the same update after the two-pass statistics (rows that are not 16-byte addressable)
__global__ void bn_running_update_kernel(const float *__restrict__ mean, const float *__restrict__ var, int C, float *__restrict__ run_mean,
                                         float *__restrict__ run_var, float momentum, float unbias, int64_t *__restrict__ batches)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float keep = 1.0f - momentum;
    run_mean[c] = run_mean[c] * keep + momentum * mean[c];
    run_var[c] = run_var[c] * keep + (momentum * unbias) * var[c];
    if (batches && c == 0) batches[0] += 1;
}

static bool bn_vec_ok(const void *x, int ld, int C)
{
#ifdef TGP_BN_SCALAR        // measurement builds only (scripts/ab_bench.py): the round-2 scalar two-pass kernels
    return false;
#endif
    return (C & 3) == 0 && (ld & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
}

extern "C" int64_t tgp_bn_workspace_floats(int64_t rows, int C)
{
    if (rows <= 0 || C <= 0) return 0;
    const int64_t two_pass = (int64_t)tgp_cdiv(rows, BN_CHUNK) * C, one_pass = (int64_t)tgp_cdiv(rows, BNV_CHUNK) * 3 * C;
    return two_pass > one_pass ? two_pass : one_pass;
}

extern "C" int tgp_bn_stats_running(const float *x, int ld, int64_t rows, int C, float *mean, float *var, float *workspace,
                                    float *run_mean, float *run_var, float momentum, int64_t *batches, tgp_stream_t stream)
{
    TGP_REQUIRE(x && mean && var && workspace && rows > 0 && C > 0 && ld >= C);
    TGP_REQUIRE((run_mean == nullptr) == (run_var == nullptr) && momentum >= 0.f && momentum <= 1.f);
    TGP_REQUIRE(!(run_mean && rows < 2));      // nn.BatchNorm1d refuses one value per channel in training: no unbiased variance
    const float unbias = (float)((double)rows / (double)(rows > 1 ? rows - 1 : 1));
    if (bn_vec_ok(x, ld, C) && (reinterpret_cast<uintptr_t>(workspace) & 15) == 0) {
        const int vchunks = tgp_cdiv(rows, BNV_CHUNK);
        hipLaunchKernelGGL(bn_stats_partial_v4_kernel, dim3(tgp_cdiv(C, 256), vchunks), dim3(256), 0, tgp_hs(stream), x, ld, rows, C, workspace);
        hipLaunchKernelGGL(bn_stats_finish_v4_kernel, dim3(tgp_cdiv(C, 64)), dim3(256), 0, tgp_hs(stream), workspace, vchunks, C, rows, mean, var,
                           run_mean, run_var, momentum, unbias, batches);
        return TGP_LAUNCH_RESULT();
    }
    const int chunks = tgp_cdiv(rows, BN_CHUNK);
    const dim3 grid(tgp_cdiv(C, 64), chunks), block(256);
    const float inv = (float)(1.0 / (double)rows);
    hipLaunchKernelGGL(bn_partial_kernel<false>, grid, block, 0, tgp_hs(stream), x, ld, rows, C, (const float *)nullptr, workspace);
    hipLaunchKernelGGL(bn_finish_kernel, dim3(tgp_cdiv(C, 256)), dim3(256), 0, tgp_hs(stream), workspace, chunks, C, inv, mean);
    hipLaunchKernelGGL(bn_partial_kernel<true>, grid, block, 0, tgp_hs(stream), x, ld, rows, C, (const float *)mean, workspace);
    hipLaunchKernelGGL(bn_finish_kernel, dim3(tgp_cdiv(C, 256)), dim3(256), 0, tgp_hs(stream), workspace, chunks, C, inv, var);
    if (run_mean)
        hipLaunchKernelGGL(bn_running_update_kernel, dim3(tgp_cdiv(C, 256)), dim3(256), 0, tgp_hs(stream), mean, var, C, run_mean, run_var,
                           momentum, unbias, batches);
    return TGP_LAUNCH_RESULT();
}

extern "C" int tgp_bn_stats(const float *x, int ld, int64_t rows, int C, float *mean, float *var, float *workspace,
                            tgp_stream_t stream)
{
    return tgp_bn_stats_running(x, ld, rows, C, mean, var, workspace, nullptr, nullptr, 0.f, nullptr, stream);
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

// the same with a column quad per thread (16-byte loads and stores); per-element arithmetic and key handling unchanged
__global__ __launch_bounds__(256) void bn_apply_v4_kernel(const float *__restrict__ x, int ld, int64_t rows, int C,
                                                          const float *__restrict__ mean, const float *__restrict__ var,
                                                          const float *__restrict__ gamma, const float *__restrict__ beta, float eps,
                                                          int act, float slope, const float *__restrict__ slope_vec,
                                                          float *__restrict__ out, int ldo, uint32_t *__restrict__ cm, int ldcm,
                                                          int cm_cols, int rows_per_obj)
{
    const int c = blockIdx.x * 256 + (threadIdx.x & 63) * 4;
    const int slice = threadIdx.x >> 6;
    if (c >= C) return;
    float mu[4], a[4], b[4], sl[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        mu[q] = mean[c + q];
        a[q] = gamma[c + q] / sqrtf(var[c + q] + eps);
        b[q] = beta[c + q];
        sl[q] = slope_vec ? slope_vec[c + q] : slope;
    }
    const int64_t r0 = (int64_t)blockIdx.y * BNV_CHUNK;
    const int64_t r1 = r0 + BNV_CHUNK < rows ? r0 + BNV_CHUNK : rows;
    const bool do_cm = cm && c < cm_cols;                    // cm_cols % 4 == 0 on this path (host-checked)
    uint32_t run_key[4] = {0, 0, 0, 0};
    int64_t run_obj = -1;
#pragma unroll 4
    for (int64_t r = r0 + slice; r < r1; r += 4) {
        const float4 xv = *reinterpret_cast<const float4 *>(x + r * ld + c);
        float v[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            v[q] = (v[q] - mu[q]) * a[q] + b[q];
            if (act == 1) v[q] = v[q] > 0.f ? v[q] : v[q] * sl[q];
        }
        if (out) *reinterpret_cast<float4 *>(out + r * ldo + c) = make_float4(v[0], v[1], v[2], v[3]);
        if (do_cm) {
            const int64_t obj = r / rows_per_obj;
            if (obj != run_obj) {
                if (run_obj >= 0) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) atomicMax(cm + run_obj * ldcm + c + q, run_key[q]);
                }
                run_obj = obj;
#pragma unroll
                for (int q = 0; q < 4; ++q) run_key[q] = tgp_float_key(v[q]);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint32_t key = tgp_float_key(v[q]);
                    run_key[q] = key > run_key[q] ? key : run_key[q];
                }
            }
        }
    }
    if (do_cm && run_obj >= 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) atomicMax(cm + run_obj * ldcm + c + q, run_key[q]);
    }
}

extern "C" int tgp_bn_apply(const float *x, int ld, int64_t rows, int C, const float *mean, const float *var,
                            const float *gamma, const float *beta, float eps, int act, float slope, const float *slope_vec,
                            float *out, int ldo, uint32_t *colmax_keys, int ldcm, int cm_cols, int rows_per_obj,
                            tgp_stream_t stream)
{
    TGP_REQUIRE(x && mean && var && gamma && beta && rows > 0 && C > 0 && ld >= C && (out || colmax_keys));
    TGP_REQUIRE(!out || ldo >= C);
    TGP_REQUIRE(!colmax_keys || (rows_per_obj > 0 && ldcm > 0));
    if (bn_vec_ok(x, ld, C) && (!out || ((ldo & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0)) && ((cm_cols > 0 ? cm_cols : C) & 3) == 0) {
        hipLaunchKernelGGL(bn_apply_v4_kernel, dim3(tgp_cdiv(C, 256), tgp_cdiv(rows, BNV_CHUNK)), dim3(256), 0, tgp_hs(stream), x, ld,
                           rows, C, mean, var, gamma, beta, eps, act, slope, slope_vec, out, ldo, colmax_keys, ldcm,
                           cm_cols > 0 ? cm_cols : C, rows_per_obj > 0 ? rows_per_obj : 1);
        return TGP_LAUNCH_RESULT();
    }
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
