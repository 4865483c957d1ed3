// Graph convolution, ORL pooling, kNN max-pool and row gathers of the 3D-GCN encoder on gfx950.
//
// The reference materialises (B,n,k,7*C) tensors three times per layer (theta, gathered support,
// their product: gcn3d.py:166-177) and reduces them afterwards; here one wavefront owns one point
// and keeps the 7 x 4-channel running maxima in registers while it streams the k gathered support
// rows (float4 per lane, whole 512 B..1 KiB row segments per wave-instruction) -- nothing of size
// k*7*C ever exists.  All workgroups of one object are placed on one XCD (tgp_xcd_object_tile) so
// the object's projection table (n x 8C floats, 3.7 MB at n=1028, C=128) is served from that L2.
#include <stdlib.h>

#include "tgp_common.h"

#define GC_S 7          // support directions per kernel (config/config.py:44 gcn_sup_num)
#define GC_PTS 16       // points per workgroup (4 waves x 4)
#define GC_MAXK 64

// F.normalize(directions, dim=0): column c of a (3, SC) matrix scaled to unit length (eps 1e-12)
__global__ void normalize_dirs_kernel(const float *__restrict__ d, int SC, float *__restrict__ out)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= SC) return;
    const float x = d[c], y = d[SC + c], z = d[2 * SC + c];
    const float nrm = fmaxf(sqrtf((x * x + y * y) + z * z), 1e-12f);
    out[c] = x / nrm;
    out[SC + c] = y / nrm;
    out[2 * SC + c] = z / nrm;
}

extern "C" int tgp_normalize_dirs(const float *directions, int SC, float *out, tgp_stream_t stream)
{
    TGP_REQUIRE(directions && out && SC > 0);
    hipLaunchKernelGGL(normalize_dirs_kernel, dim3(tgp_cdiv(SC, 256)), dim3(256), 0, tgp_hs(stream), directions, SC, out);
    return TGP_LAUNCH_RESULT();
}

// backward of normalize_dirs: n = d / max(|d|, eps) per column -> dd = (g - n (n . g)) / |d|   (|d| > eps; below it n = d / eps, dd = g / eps)
__global__ void normalize_dirs_bwd_kernel(const float *__restrict__ d, const float *__restrict__ g, int SC, float *__restrict__ dd)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= SC) return;
    const float x = d[c], y = d[SC + c], z = d[2 * SC + c];
    const float gx = g[c], gy = g[SC + c], gz = g[2 * SC + c];
    const float nrm = sqrtf((x * x + y * y) + z * z);
    if (nrm > 1e-12f) {
        const float nx = x / nrm, ny = y / nrm, nz = z / nrm;
        const float dot = (nx * gx + ny * gy) + nz * gz;
        dd[c] = (gx - nx * dot) / nrm, dd[SC + c] = (gy - ny * dot) / nrm, dd[2 * SC + c] = (gz - nz * dot) / nrm;
    } else {
        dd[c] = gx / 1e-12f, dd[SC + c] = gy / 1e-12f, dd[2 * SC + c] = gz / 1e-12f;
    }
}

extern "C" int tgp_normalize_dirs_bwd(const float *directions, const float *grad, int SC, float *out, tgp_stream_t stream)
{
    TGP_REQUIRE(directions && grad && out && SC > 0);
    hipLaunchKernelGGL(normalize_dirs_bwd_kernel, dim3(tgp_cdiv(SC, 256)), dim3(256), 0, tgp_hs(stream), directions, grad, SC, out);
    return TGP_LAUNCH_RESULT();
}

// Two fp32 lanes per instruction (v_pk_mul_f32 / v_pk_fma_f32 run at the full VALU rate on gfx950, so the <direction, support
// direction> products cost half the issue slots); each component is the same IEEE operation as the scalar fmaf / multiply.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk_fma(const f32x2 a, const f32x2 b, const f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 pk_max(const f32x2 a, const f32x2 b) { return __builtin_elementwise_max(a, b); }

// Lane layout shared by the gather kernels: a row of C floats is covered by C/4 lanes (float4 each).
//   C = 128: 32 lanes per row, the two wave halves take alternate neighbours (SPLIT = 2)
//   C = 256: 64 lanes per row
//   C = 512: two 256-column chunks, one workgroup per chunk (CHUNKS = 2)
template <int C>
struct RowLanes {
    static constexpr int LPR = (C / 4) < 64 ? (C / 4) : 64;
    static constexpr int SPLIT = 64 / LPR;
    static constexpr int CHUNKS = (C / 4 + 63) / 64;
};

// (Measured and dropped: conv_1's 3.7 MB per-object support table does not quite fit an XCD's 4 MB L2 next to the streams passing
// through -- FETCH_SIZE shows every row fetched ~6 times, 682 MB per launch -- but splitting the row into two 64-column chunks
// processed one after the other, so that 1.8 MB stays resident, was slower: 188 vs 134 us.  Half rows mean 256-byte gathers,
// twice the per-point index / direction work and a padded neighbour slot.)
template <int C, bool SURFACE>
__global__ __launch_bounds__(256) void gconv_kernel(const float *__restrict__ xyz, const int32_t *__restrict__ idx,
                                                    const float *__restrict__ proj, int ldp,
                                                    const float *__restrict__ sdn, int B, int n, int k,
                                                    float *__restrict__ out, int ldo, int tiles_per_obj, int xyz_pad)
{
    using RL = RowLanes<C>;
    int b, tile;
    if (!tgp_xcd_object_tile(blockIdx.x, B, tiles_per_obj, b, tile)) return;
    const int chunk = tile % RL::CHUNKS;
    const int ptile = tile / RL::CHUNKS;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int half = lane / RL::LPR;
    const int cb = chunk * 256 + 4 * (lane % RL::LPR);
    constexpr int SC = GC_S * C;

    // unit support directions of this lane's 4 channels: 7 supports x 3 components
    float4 sd[GC_S][3];
#pragma unroll
    for (int s = 0; s < GC_S; ++s)
#pragma unroll
        for (int c = 0; c < 3; ++c) sd[s][c] = *reinterpret_cast<const float4 *>(sdn + c * SC + s * C + cb);

    for (int pp = wave; pp < GC_PTS; pp += 4) {
        const int i = ptile * GC_PTS + pp;
        if (i >= n) break;
        const int64_t rowi = (int64_t)b * n + i;
        // lanes 0..k-1: neighbour id and unit direction (gcn3d.py:48-58)
        int nj = 0;
        float dx = 0.f, dy = 0.f, dz = 0.f;
        if (lane < k) {
            nj = idx[rowi * k + lane];
            const float *pn = xyz + ((int64_t)b * n + nj) * 3;
            const float *pc = xyz + rowi * 3;
            dx = pn[0] - pc[0];
            dy = pn[1] - pc[1];
            dz = pn[2] - pc[2];
            const float nrm = fmaxf(sqrtf((dx * dx + dy * dy) + dz * dz), 1e-12f);
            dx = dx / nrm;
            dy = dy / nrm;
            dz = dz / nrm;
        }
        float4 m[GC_S];
        const float init = SURFACE ? 0.f : -INFINITY;
#pragma unroll
        for (int s = 0; s < GC_S; ++s) m[s] = make_float4(init, init, init, init);

        // two neighbours per lane group and iteration, so that the running maxima take both candidates in one v_max3_f32; a slot
        // past k repeats neighbour 0 (a duplicate candidate cannot change a maximum): no branches, no selects
        const f32x2 zero2 = {0.f, 0.f};
        for (int jj = 0; jj * RL::SPLIT < k; jj += 2) {
            const int j0 = jj * RL::SPLIT + half, j1 = j0 + RL::SPLIT;
            const int src0 = j0 < k ? j0 : 0, src1 = j1 < k ? j1 : 0;
            const float ux0 = __shfl(dx, src0, 64), uy0 = __shfl(dy, src0, 64), uz0 = __shfl(dz, src0, 64);
            const float ux1 = __shfl(dx, src1, 64), uy1 = __shfl(dy, src1, 64), uz1 = __shfl(dz, src1, 64);
            float4 sup0[GC_S], sup1[GC_S];
            if (!SURFACE) {
                const float *prow0 = proj + ((int64_t)b * n + __shfl(nj, src0, 64)) * ldp + C + cb;
                const float *prow1 = proj + ((int64_t)b * n + __shfl(nj, src1, 64)) * ldp + C + cb;
#pragma unroll
                for (int s = 0; s < GC_S; ++s) sup0[s] = *reinterpret_cast<const float4 *>(prow0 + s * C);
#pragma unroll
                for (int s = 0; s < GC_S; ++s) sup1[s] = *reinterpret_cast<const float4 *>(prow1 + s * C);
            }
            const f32x2 vx0 = {ux0, ux0}, vy0 = {uy0, uy0}, vz0 = {uz0, uz0}, vx1 = {ux1, ux1}, vy1 = {uy1, uy1}, vz1 = {uz1, uz1};
#pragma unroll
            for (int s = 0; s < GC_S; ++s) {
                // theta = relu(ux * sx + uy * sy + uz * sz) in the reference's order, channels (x, y) and (z, w) in pairs
                const f32x2 sxa = {sd[s][0].x, sd[s][0].y}, sya = {sd[s][1].x, sd[s][1].y}, sza = {sd[s][2].x, sd[s][2].y};
                const f32x2 sxb = {sd[s][0].z, sd[s][0].w}, syb = {sd[s][1].z, sd[s][1].w}, szb = {sd[s][2].z, sd[s][2].w};
                f32x2 ta0 = pk_fma(vz0, sza, pk_fma(vy0, sya, vx0 * sxa)), tb0 = pk_fma(vz0, szb, pk_fma(vy0, syb, vx0 * sxb));
                f32x2 ta1 = pk_fma(vz1, sza, pk_fma(vy1, sya, vx1 * sxa)), tb1 = pk_fma(vz1, szb, pk_fma(vy1, syb, vx1 * sxb));
                if (!SURFACE) {
                    ta0 = pk_max(ta0, zero2) * f32x2{sup0[s].x, sup0[s].y}, tb0 = pk_max(tb0, zero2) * f32x2{sup0[s].z, sup0[s].w};
                    ta1 = pk_max(ta1, zero2) * f32x2{sup1[s].x, sup1[s].y}, tb1 = pk_max(tb1, zero2) * f32x2{sup1[s].z, sup1[s].w};
                }   // (SURFACE: the maxima start at 0, so max(m, relu(t)) = max(m, t))
                m[s].x = fmaxf(m[s].x, fmaxf(ta0.x, ta1.x)), m[s].y = fmaxf(m[s].y, fmaxf(ta0.y, ta1.y));
                m[s].z = fmaxf(m[s].z, fmaxf(tb0.x, tb1.x)), m[s].w = fmaxf(m[s].w, fmaxf(tb0.y, tb1.y));
            }
        }
#pragma unroll
        for (int off = 32; off >= RL::LPR; off >>= 1) {           // combine the lane groups that took alternate neighbours
#pragma unroll
            for (int s = 0; s < GC_S; ++s) {
                m[s].x = fmaxf(m[s].x, __shfl_xor(m[s].x, off, 64));
                m[s].y = fmaxf(m[s].y, __shfl_xor(m[s].y, off, 64));
                m[s].z = fmaxf(m[s].z, __shfl_xor(m[s].z, off, 64));
                m[s].w = fmaxf(m[s].w, __shfl_xor(m[s].w, off, 64));
            }
        }
        if (half == 0) {
            float4 acc = m[0]; // torch.mean over the 7 supports: sequential sum, then / 7
#pragma unroll
            for (int s = 1; s < GC_S; ++s) acc.x += m[s].x, acc.y += m[s].y, acc.z += m[s].z, acc.w += m[s].w;
            acc.x = acc.x / 7.0f, acc.y = acc.y / 7.0f, acc.z = acc.z / 7.0f, acc.w = acc.w / 7.0f;
            if (!SURFACE) {
                const float4 ctr = *reinterpret_cast<const float4 *>(proj + rowi * ldp + cb);
                acc.x = ctr.x + acc.x, acc.y = ctr.y + acc.y, acc.z = ctr.z + acc.z, acc.w = ctr.w + acc.w;
            }
            *reinterpret_cast<float4 *>(out + rowi * ldo + cb) = acc;
            // xyz_pad: the point itself as columns C .. C+3 = (x, y, z, 0) of the output row, so that the layer's last GEMM takes
            // the STE convolution (Conv1d 3 -> C on xyz, gcn3d.py:79,87) as four more K columns instead of a GEMM of its own
            if (SURFACE && xyz_pad && chunk == 0 && lane == 0) {
                const float *pc = xyz + rowi * 3;
                *reinterpret_cast<float4 *>(out + rowi * ldo + C) = make_float4(pc[0], pc[1], pc[2], 0.f);
            }
        }
    }
}

// HS_layer.graph_conv with the object's support table staged in LDS ("LDS-staged neighbour tiles").
// The wave-per-point kernel above gathers k rows of 7C floats per point from L2: every support row is fetched ~k times
// (conv_1: 2.4 GB of L2 -> CU traffic per forward for a 118 MB table), and the L2's bandwidth bounds it.  Here a workgroup
// owns (object, CH channels): it loads that channel slice of the object's table -- [n][7][CH] floats, each element of the
// table read from global memory exactly once per forward -- into LDS, then every thread (point, channel pair) walks its k
// neighbours with ds_read_b64.  Same per-element arithmetic and order as gconv_kernel (fmaf chain, max over neighbours,
// sequential mean over the 7 supports): results are bit-identical.
// The unit neighbour directions (gcn3d.py:48-58 get_neighbor_direction_norm: a square root and three IEEE divisions per
// neighbour) do not depend on the channel: computed per thread they were half of this kernel's vector instructions, repeated by
// the C / 2 threads that share a point.  nbr_dirs_kernel writes them once per (point, neighbour) as float4 into the caller's
// scratch; the four threads of a point then read the same 16 bytes from L2.
__global__ void nbr_dirs_kernel(const float *__restrict__ xyz, const int32_t *__restrict__ idx, int B, int n, int k, float4 *__restrict__ dirs)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)B * n * k) return;
    const int64_t rowi = t / k;
    const int b = (int)(rowi / n);
    const float *pc = xyz + rowi * 3;
    const float *pn = xyz + ((int64_t)b * n + idx[t]) * 3;
    float dx = pn[0] - pc[0], dy = pn[1] - pc[1], dz = pn[2] - pc[2];
    const float nrm = fmaxf(sqrtf((dx * dx + dy * dy) + dz * dz), 1e-12f);
    dirs[t] = make_float4(dx / nrm, dy / nrm, dz / nrm, 0.f);
}

template <int CH>
__global__ __launch_bounds__(1024) void gconv_lds_kernel(const float4 *__restrict__ dirs, const int32_t *__restrict__ idx,
                                                         const float *__restrict__ proj, int ldp, const float *__restrict__ sdn,
                                                         int B, int n, int k, int C, float *__restrict__ out, int ldo)
{
    extern __shared__ __attribute__((aligned(16))) float gl_smem[];
    constexpr int QC = CH / 4;                 // float4 chunks per table row segment (fill)
    constexpr int PC = CH / 2;                 // channel pairs per workgroup: a thread owns (point, pair) -- 42 direction
    constexpr int ROW = GC_S * CH;             // registers instead of 84, which keeps 4 waves per SIMD
    // (Round 4, three measurements on the 60 us launches of conv_2 / conv_3, none of which moved them: 25 % fewer vector instructions
    // (the ReLU as an output clamp, see pk_max above): 61 us; the next four neighbours' ids and directions requested before this
    // trip's arithmetic: 64 us; table rows padded against bank conflicts (below): 62 us.)
    // (Measured and dropped, round 4: a row stride of GC_S * CH + 2 floats.  A wave's 16 points read 16 neighbour rows at the same
    // (support, pair) offset, and with 56 / 112-float rows those start in 4 / 2 bank groups -- 50 % / 32 % of the LDS-busy cycles are
    // conflicts in the SQ counters -- but spreading them over 16 changed nothing: 62 vs 60 us, 19 vs 19.  The kernel waits on the
    // index -> row -> value chains, not on LDS bandwidth.)
    float *s_tab = gl_smem;
    int b, chunk;
    if (!tgp_xcd_object_tile(blockIdx.x, B, C / CH, b, chunk)) return;
    const int c0 = chunk * CH;
    const int SC = GC_S * C;
    const int tid = threadIdx.x, nthr = blockDim.x;
    for (int e = tid; e < n * GC_S * QC; e += nthr) {
        const int row = e / (GC_S * QC), rem = e - row * (GC_S * QC);
        const int sidx = rem / QC, q = rem - sidx * QC;
        const float4 v = *reinterpret_cast<const float4 *>(proj + ((int64_t)b * n + row) * ldp + C + sidx * C + c0 + q * 4);
        *reinterpret_cast<float4 *>(s_tab + row * ROW + sidx * CH + q * 4) = v;
    }
    const int q = tid % PC, pl = tid / PC, pstep = nthr / PC;
    const int cb = c0 + q * 2;
    float2 sd[GC_S][3];
#pragma unroll
    for (int sidx = 0; sidx < GC_S; ++sidx)
#pragma unroll
        for (int c = 0; c < 3; ++c) sd[sidx][c] = *reinterpret_cast<const float2 *>(sdn + c * SC + sidx * C + cb);
    __syncthreads();
    for (int i = pl; i < n; i += pstep) {
        const int64_t rowi = (int64_t)b * n + i;
        float2 m[GC_S];
#pragma unroll
        for (int sidx = 0; sidx < GC_S; ++sidx) m[sidx] = make_float2(-INFINITY, -INFINITY);
        const int32_t *nbrs = idx + rowi * k;
        const float4 *udir = dirs + rowi * k;
#pragma unroll 4
        for (int j = 0; j < k; ++j) {
            const int nb = nbrs[j];
            const float4 u = udir[j];
            const float ux = u.x, uy = u.y, uz = u.z;
            const float *trow = s_tab + nb * ROW + q * 2;
#pragma unroll
            for (int sidx = 0; sidx < GC_S; ++sidx) {
                const float2 sup = *reinterpret_cast<const float2 *>(trow + sidx * CH);
                f32x2 t = pk_fma(f32x2{uz, uz}, f32x2{sd[sidx][2].x, sd[sidx][2].y},
                                 pk_fma(f32x2{uy, uy}, f32x2{sd[sidx][1].x, sd[sidx][1].y}, f32x2{ux, ux} * f32x2{sd[sidx][0].x, sd[sidx][0].y}));
                t = pk_max(t, f32x2{0.f, 0.f}) * f32x2{sup.x, sup.y};
                m[sidx].x = fmaxf(m[sidx].x, t.x), m[sidx].y = fmaxf(m[sidx].y, t.y);
            }
        }
        float2 acc = m[0];   // torch.mean over the 7 supports: sequential sum, then / 7
#pragma unroll
        for (int sidx = 1; sidx < GC_S; ++sidx) acc.x += m[sidx].x, acc.y += m[sidx].y;
        acc.x = acc.x / 7.0f, acc.y = acc.y / 7.0f;
        const float2 ctr = *reinterpret_cast<const float2 *>(proj + rowi * ldp + cb);
        acc.x = ctr.x + acc.x, acc.y = ctr.y + acc.y;
        *reinterpret_cast<float2 *>(out + rowi * ldo + cb) = acc;
    }
}

// picks the widest channel slice whose table fits with two workgroups per CU (<= 72 KB): measured 137 -> 88 us (n = 257,
// C = 256), 37 -> 25 us (n = 64, C = 512); larger clouds keep the L2 gather
template <int CH>
static int gconv_lds_go(const float4 *dirs, const int32_t *idx, const float *proj, int ldp, const float *sdn, int B, int n, int k,
                        int C, float *out, int ldo, size_t lds, hipStream_t stream)
{
    auto fn = gconv_lds_kernel<CH>;
    static TgpLdsAttr attr;
    if (lds > 64 * 1024)
        if (const int e = tgp_lds_attr(attr, reinterpret_cast<const void *>(fn), 150 * 1024)) return e;
    const int threads = lds > 72 * 1024 ? 1024 : 512;
    hipLaunchKernelGGL(fn, dim3(tgp_xcd_grid(B, C / CH)), dim3(threads), lds, stream, dirs, idx, proj, ldp, sdn, B, n, k, C, out, ldo);
    return TGP_LAUNCH_RESULT();
}

#ifdef TGP_DEV   // development builds only (scripts/gconv_ab.py): 0 = always the gather-from-L2 kernel
int tgp_gconv_lds_mode = 1;
extern "C" void tgp_debug_set_gconv_lds(int v) { tgp_gconv_lds_mode = v; }
#else
static constexpr int tgp_gconv_lds_mode = 1;
#endif

static int gconv_lds_launch(const float *xyz, const int32_t *idx, const float *proj, int ldp, const float *sdn, int B, int n, int k,
                            int C, float *out, int ldo, float *dirs_ws, hipStream_t stream, bool &done, bool have_dirs = false)
{
    done = false;
    if (!tgp_gconv_lds_mode || !dirs_ws) return 0;          // without the directions' scratch: the L2-gather kernel
    auto bytes = [&](int ch) { return (size_t)n * GC_S * ch * sizeof(float); };
    const int ch = (bytes(16) <= 72 * 1024 && C % 16 == 0) ? 16 : (bytes(8) <= 72 * 1024 && C % 8 == 0) ? 8
                   // a 4-channel slice (n = 1028: 115 KB, one workgroup per CU) measured slower than the L2 gather: 208 vs 168 us.
                   // Round 3 rebuilt that slice with a thread slot per (point, quarter of the neighbours, all four channels) --
                   // float4 LDS reads, one index / direction load per 112 bytes of LDS, quad shuffles for the maxima, 512 threads
                   // at 152 VGPRs -- bit-identical and slower still: 216-224 us at (32, 1028, 128) against 152-155 for the L2
                   // gather in the same harness (random graphs; 132 in the forward), and 123 vs 74 us at (32, 257, 256) against
                   // the 8-channel form.  A four-channel row is 112 bytes: 64 lanes gathering 16 bytes each from random rows
                   // collide in the LDS banks ~3 deep, and one workgroup of 8 waves per CU does not hide the index -> row ->
                   // value chains.  Removed again (profiles/r03_gconv_lds4_ab.txt).
                   : (tgp_gconv_lds_mode == 2 && bytes(4) <= 150 * 1024) ? 4 : 0;
    if (!ch) return 0;
    done = true;
    float4 *dirs = reinterpret_cast<float4 *>(dirs_ws);
    const int64_t nd = (int64_t)B * n * k;
    if (!have_dirs) hipLaunchKernelGGL(nbr_dirs_kernel, dim3(tgp_cdiv(nd, 256)), dim3(256), 0, stream, xyz, idx, B, n, k, dirs);
    if (ch == 16) return gconv_lds_go<16>(dirs, idx, proj, ldp, sdn, B, n, k, C, out, ldo, bytes(16), stream);
    if (ch == 8) return gconv_lds_go<8>(dirs, idx, proj, ldp, sdn, B, n, k, C, out, ldo, bytes(8), stream);
    return gconv_lds_go<4>(dirs, idx, proj, ldp, sdn, B, n, k, C, out, ldo, bytes(4), stream);
}

static int gconv_check(const void *xyz, const void *idx, const void *sdn, const void *out, int B, int n, int k, int S,
                       int C, int ldo)
{
    if (!xyz || !idx || !sdn || !out || B <= 0 || n <= 0 || k <= 0) return TGP_EINVAL;
    if (S != GC_S || k > GC_MAXK || !(C == 128 || C == 256 || C == 512)) return TGP_EUNSUPPORTED;
    if (ldo < C || (ldo & 3) || (reinterpret_cast<uintptr_t>(out) & 15) || (reinterpret_cast<uintptr_t>(sdn) & 15))
        return TGP_EINVAL;
    return 0;
}

template <bool SURFACE>
static int gconv_launch(const float *xyz, const int32_t *idx, const float *proj, int ldp, const float *sdn, int B, int n,
                        int k, int C, float *out, int ldo, hipStream_t stream, int xyz_pad = 0)
{
    const int ptiles = tgp_cdiv(n, GC_PTS);
    // (round 4, measured: capping the workgroups per CU at 4 / 3 / 2 with unused dynamic LDS -- to keep one object's 3.7 MB table in
    // the XCD's L2 instead of two -- leaves conv_1's launch at 138 us in every case: the kernel is bound by vector-instruction issue,
    // 4.6e7 wave instructions of which three quarters are theta, not by its L2 misses)
#define GC_GO(CC)                                                                                                     \
    {                                                                                                                 \
        const int tiles = ptiles * RowLanes<CC>::CHUNKS;                                                              \
        hipLaunchKernelGGL((gconv_kernel<CC, SURFACE>), dim3(tgp_xcd_grid(B, tiles)), dim3(256), 0, stream, xyz, idx, \
                           proj, ldp, sdn, B, n, k, out, ldo, tiles, xyz_pad);                                        \
    }
    if (C == 128) GC_GO(128) else if (C == 256) GC_GO(256) else GC_GO(512)
#undef GC_GO
    return TGP_LAUNCH_RESULT();
}

extern "C" int tgp_gconv_surface_fwd(const float *xyz, const int32_t *idx, const float *sdn, int B, int n, int k, int S,
                                     int C, float *out, int ldo, int xyz_pad, tgp_stream_t stream)
{
    const int chk = gconv_check(xyz, idx, sdn, out, B, n, k, S, C, ldo);
    if (chk) return chk;
    TGP_REQUIRE(!xyz_pad || ldo >= C + 4);
    return gconv_launch<true>(xyz, idx, nullptr, 0, sdn, B, n, k, C, out, ldo, tgp_hs(stream), xyz_pad);
}

extern "C" int tgp_gconv_hs_fwd(const float *xyz, const int32_t *idx, const float *proj, int ldp, const float *sdn, int B,
                                int n, int k, int S, int C, float *out, int ldo, float *dirs_ws, tgp_stream_t stream)
{
    const int chk = gconv_check(xyz, idx, sdn, out, B, n, k, S, C, ldo);
    if (chk) return chk;
    TGP_REQUIRE(proj && ldp >= (S + 1) * C && (ldp & 3) == 0 && (reinterpret_cast<uintptr_t>(proj) & 15) == 0);
    TGP_REQUIRE((reinterpret_cast<uintptr_t>(dirs_ws) & 15) == 0);
    bool done = false;
    const int rc = gconv_lds_launch(xyz, idx, proj, ldp, sdn, B, n, k, C, out, ldo, dirs_ws, tgp_hs(stream), done);
    if (done || rc) return rc;
    return gconv_launch<false>(xyz, idx, proj, ldp, sdn, B, n, k, C, out, ldo, tgp_hs(stream));
}

// tgp_gconv_hs_fwd with the unit neighbour directions already in `dirs` (B, n, k) float4 -- tgp_knn_feat_dirs wrote them beside
// the list -- : the LDS-staged kernel alone.  TGP_EUNSUPPORTED (nothing launched) where that kernel does not serve the shape.
extern "C" int tgp_gconv_hs_fwd_dirs(const float *xyz, const int32_t *idx, const float *proj, int ldp, const float *sdn, int B,
                                     int n, int k, int S, int C, float *out, int ldo, const float *dirs, tgp_stream_t stream)
{
    const int chk = gconv_check(xyz, idx, sdn, out, B, n, k, S, C, ldo);
    if (chk) return chk;
    TGP_REQUIRE(proj && ldp >= (S + 1) * C && (ldp & 3) == 0 && (reinterpret_cast<uintptr_t>(proj) & 15) == 0);
    TGP_REQUIRE(dirs && (reinterpret_cast<uintptr_t>(dirs) & 15) == 0);
    bool done = false;
    const int rc = gconv_lds_launch(xyz, idx, proj, ldp, sdn, B, n, k, C, out, ldo, const_cast<float *>(dirs), tgp_hs(stream), done, true);
    if (rc) return rc;
    return done ? 0 : TGP_EUNSUPPORTED;
}

// ---------------------------------------------------------------------------------------------------
// ORL: g[b,c] = mean_i max_j feat[b, idx[b,i,j], c]   (gcn3d.py:210-217)
// stage 1: each workgroup sums the neighbour-max rows of ORL_PTS points -> partial[b, tile, :]
// stage 2: partials summed in tile order and divided by n (deterministic, no float atomics)
#define ORL_PTS 64

template <int C>
__global__ __launch_bounds__(256) void orl_partial_kernel(const float *__restrict__ feat, int ldf,
                                                          const int32_t *__restrict__ idx, int B, int n, int k,
                                                          float *__restrict__ partial, int ptiles, int tiles_per_obj)
{
    using RL = RowLanes<C>;
    __shared__ float4 red[4][64];
    int b, tile;
    if (!tgp_xcd_object_tile(blockIdx.x, B, tiles_per_obj, b, tile)) return;
    const int chunk = tile % RL::CHUNKS;
    const int ptile = tile / RL::CHUNKS;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int half = lane / RL::LPR;
    const int cb = chunk * 256 + 4 * (lane % RL::LPR);
    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int pp = wave; pp < ORL_PTS; pp += 4) {
        const int i = ptile * ORL_PTS + pp;
        if (i >= n) break;
        const int nj = (lane < k) ? idx[((int64_t)b * n + i) * k + lane] : 0;
        float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        for (int jj = 0; jj * RL::SPLIT < k; ++jj) {
            const int j = jj * RL::SPLIT + half;
            const bool valid = j < k;
            const int nb = __shfl(nj, valid ? j : 0, 64);
            const float4 v = *reinterpret_cast<const float4 *>(feat + ((int64_t)b * n + nb) * ldf + cb);
            if (valid) m.x = fmaxf(m.x, v.x), m.y = fmaxf(m.y, v.y), m.z = fmaxf(m.z, v.z), m.w = fmaxf(m.w, v.w);
        }
        if (RL::SPLIT == 2) {
            m.x = fmaxf(m.x, __shfl_xor(m.x, 32, 64));
            m.y = fmaxf(m.y, __shfl_xor(m.y, 32, 64));
            m.z = fmaxf(m.z, __shfl_xor(m.z, 32, 64));
            m.w = fmaxf(m.w, __shfl_xor(m.w, 32, 64));
        }
        sum.x += m.x, sum.y += m.y, sum.z += m.z, sum.w += m.w;
    }
    red[wave][lane] = sum;
    __syncthreads();
    if (wave == 0 && half == 0) {
        float4 t = red[0][lane];
#pragma unroll
        for (int w = 1; w < 4; ++w) t.x += red[w][lane].x, t.y += red[w][lane].y, t.z += red[w][lane].z, t.w += red[w][lane].w;
        *reinterpret_cast<float4 *>(partial + ((int64_t)b * ptiles + ptile) * C + cb) = t;
    }
}

// The same neighbour max / per-tile point sums with the object's feature table staged in LDS in 16-channel slices
// (n = 1028: 66 KB), every row read from global memory once instead of ~k times.  A thread owns (point tile, wave slot,
// channel pair) and reproduces the summation order of orl_partial_kernel exactly (slot w sums points w, w+4, ... of the
// tile, then ((s0 + s1) + s2) + s3), so `partial` is bit-identical.
// (Round 4, measured and dropped: the point's neighbour list as 16-byte loads issued together ahead of the walk, here and in
// gconv_lds_kernel, instead of one index load per neighbour inside the 4-times unrolled loop: 25.6 -> 30.7 us (n = 1028), 56 -> 67 us for
// the LDS graph convolution at n = 257 -- the unrolled loop already keeps four index loads in flight, and the eight predicated
// 16-byte loads per point serialise ahead of the first table read.)
#define ORL_CH 16
// (round 4) A 16-channel slice of the table is one K-tile of the layer's last GEMM, whose A operand the table is: while the slice sits
// in LDS the workgroup also writes it as fp16 hi / lo planes in the blocked layout of the pre-split GEMM (include/tgpose.h,
// tgp_gemm_args.A_planes) -- every element of the table is then read once for both purposes, the pieces leave as 512-byte runs --
// with the per-row-block magnitudes of that kernel's range guard.  xyz_tile (conv_0): the workgroup of slice 0 also writes K-tile
// C / 16 = (x, y, z, 0, 0 ...), the STE convolution's four extra K columns (engine.surface_layer).
typedef _Float16 ol_f16x4 __attribute__((ext_vector_type(4)));
typedef float ol_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void ol_split(const float4 v, uint2 &hi, uint2 &lo)
{
    const ol_f32x4 x = {v.x, v.y, v.z, v.w};
    const ol_f16x4 h = __builtin_convertvector(x, ol_f16x4);
    const ol_f32x4 rest = x - __builtin_convertvector(h, ol_f32x4);
    const ol_f16x4 l = __builtin_convertvector(rest, ol_f16x4);
    hi = __builtin_bit_cast(uint2, h);
    lo = __builtin_bit_cast(uint2, l);
}

__global__ __launch_bounds__(1024) void orl_lds_kernel(const float *__restrict__ feat, int ldf, const int32_t *__restrict__ idx, int B,
                                                       int n, int k, int C, float *__restrict__ partial, int ptiles,
                                                       char *__restrict__ planes, int kts, uint32_t *__restrict__ amax,
                                                       const float *__restrict__ xyz_tile, const float *__restrict__ w2t,
                                                       float *__restrict__ g_out, float *__restrict__ rb, float *contrib,
                                                       int *tickets, int lds_idx)
{
    extern __shared__ __attribute__((aligned(16))) float ol_smem[];
    float *s_tab = ol_smem;                          // [n][ORL_CH]
    float *s_red = ol_smem + (size_t)n * ORL_CH;     // [ptiles][4][ORL_CH]
    uint32_t *s_amax = reinterpret_cast<uint32_t *>(s_red + (size_t)ptiles * 4 * ORL_CH);   // [n / 32 + 4] (planes only)
    // (round 5) the object's neighbour lists as 16-bit ids, [n][k], behind the magnitudes (lds_idx: the launch made room for them):
    // the walk below was a chain  global index load (an L2 round trip) -> LDS row read  per neighbour, 320 of them per thread
    // (84 % of the wave cycles parked in the SQ counters); with the lists in LDS both links are LDS reads
    uint16_t *s_idx = lds_idx ? reinterpret_cast<uint16_t *>(s_amax + (planes ? n / 32 + 4 : 0)) : nullptr;
    int b, chunk;
    if (!tgp_xcd_object_tile(blockIdx.x, B, C / ORL_CH, b, chunk)) return;
    const int c0 = chunk * ORL_CH;
    const int tid = threadIdx.x, nthr = blockDim.x;
    for (int e = tid; e < n * (ORL_CH / 4); e += nthr) {
        const int row = e / (ORL_CH / 4), q = e % (ORL_CH / 4);
        *reinterpret_cast<float4 *>(s_tab + row * ORL_CH + q * 4) =
            *reinterpret_cast<const float4 *>(feat + ((int64_t)b * n + row) * ldf + c0 + q * 4);
    }
    const int rb0 = (int)(((int64_t)b * n) >> 5), nrb = (int)((((int64_t)b * n + n - 1) >> 5) - rb0 + 1);
    if (planes)
        for (int e = tid; e < nrb; e += nthr) s_amax[e] = 0u;
    if (s_idx) {
        const int32_t *src = idx + (int64_t)b * n * k;
        const int total = n * k;                                  // (n * k % 4 == 0 and the lists 16-byte aligned: checked by the launch)
        for (int e = tid * 4; e < total; e += nthr * 4) {
            const int4 v = *reinterpret_cast<const int4 *>(src + e);
            *reinterpret_cast<uint2 *>(s_idx + e) = make_uint2((uint32_t)v.x | ((uint32_t)v.y << 16), (uint32_t)v.z | ((uint32_t)v.w << 16));
        }
    }
    __syncthreads();
    if (planes) {
        // piece (row, h) = 8 channels of one row: lanes 0-31 of a wave take 32 consecutive rows' h = 0, lanes 32-63 their h = 1
        const int tiles = (xyz_tile && chunk == 0) ? 2 : 1;
        for (int tl = 0; tl < tiles; ++tl)
            for (int e = tid; e < ((n + 31) & ~31) * 2; e += nthr) {
                const int row = (e >> 6) * 32 + (e & 31), h = (e >> 5) & 1;
                if (row >= n) continue;
                const int64_t gr = (int64_t)b * n + row;
                float4 v0, v1;
                if (tl == 0) {
                    v0 = *reinterpret_cast<const float4 *>(s_tab + row * ORL_CH + h * 8);
                    v1 = *reinterpret_cast<const float4 *>(s_tab + row * ORL_CH + h * 8 + 4);
                } else {
                    const float *pc = xyz_tile + gr * 3;
                    v0 = h == 0 ? make_float4(pc[0], pc[1], pc[2], 0.f) : make_float4(0.f, 0.f, 0.f, 0.f);
                    v1 = make_float4(0.f, 0.f, 0.f, 0.f);
                }
                uint2 h0, l0, h1, l1;
                ol_split(v0, h0, l0);
                ol_split(v1, h1, l1);
                const int kt = tl == 0 ? chunk : C / ORL_CH;
                char *o = planes + ((gr >> 5) * kts + kt) * 2048 + h * 512 + (int)(gr & 31) * 16;
                *reinterpret_cast<uint4 *>(o) = make_uint4(h0.x, h0.y, h1.x, h1.y);
                *reinterpret_cast<uint4 *>(o + 1024) = make_uint4(l0.x, l0.y, l1.x, l1.y);
                const uint32_t bt[8] = {__float_as_uint(v0.x), __float_as_uint(v0.y), __float_as_uint(v0.z), __float_as_uint(v0.w),
                                        __float_as_uint(v1.x), __float_as_uint(v1.y), __float_as_uint(v1.z), __float_as_uint(v1.w)};
                uint32_t m = 0u;
#pragma unroll
                for (int q = 0; q < 8; ++q) m = (bt[q] & 0x7fffffffu) > m ? (bt[q] & 0x7fffffffu) : m;
                if (m) atomicMax(s_amax + (int)((gr >> 5) - rb0), m);
            }
        __syncthreads();
        if (amax)
            for (int e = tid; e < nrb; e += nthr)
                if (s_amax[e]) atomicMax(amax + rb0 + e, s_amax[e]);
    }
    constexpr int PC = ORL_CH / 2;
    for (int t = tid; t < ptiles * 4 * PC; t += nthr) {
        const int pair = t % PC, w = (t / PC) & 3, pt = t / (4 * PC);
        float2 sum = make_float2(0.f, 0.f);
        for (int pp = w; pp < ORL_PTS; pp += 4) {
            const int i = pt * ORL_PTS + pp;
            if (i >= n) break;
            const int32_t *nb = idx + ((int64_t)b * n + i) * k;
            float2 m = make_float2(-INFINITY, -INFINITY);
            if (s_idx) {
                const uint16_t *nl = s_idx + i * k;
#pragma unroll 4
                for (int j = 0; j < k; ++j) {
                    const float2 v = *reinterpret_cast<const float2 *>(s_tab + (int)nl[j] * ORL_CH + pair * 2);
                    m.x = fmaxf(m.x, v.x), m.y = fmaxf(m.y, v.y);
                }
            } else {
#pragma unroll 4
                for (int j = 0; j < k; ++j) {
                    const float2 v = *reinterpret_cast<const float2 *>(s_tab + nb[j] * ORL_CH + pair * 2);
                    m.x = fmaxf(m.x, v.x), m.y = fmaxf(m.y, v.y);
                }
            }
            sum.x += m.x, sum.y += m.y;
        }
        *reinterpret_cast<float2 *>(s_red + (pt * 4 + w) * ORL_CH + pair * 2) = sum;
    }
    __syncthreads();
    if (!tickets) {
        for (int t = tid; t < ptiles * ORL_CH; t += nthr) {
            const int pt = t / ORL_CH, c = t % ORL_CH;
            const float *r = s_red + pt * 4 * ORL_CH + c;
            partial[((int64_t)b * ptiles + pt) * C + c0 + c] = ((r[0] + r[ORL_CH]) + r[2 * ORL_CH]) + r[3 * ORL_CH];
        }
        return;
    }
    // ---- fused finish (round 4; tgp_orl_rowbias_fused): this workgroup holds every point tile of its 16 channels, so their mean
    // over points g (orl_finish_kernel's sum, term by term) needs no other workgroup; its share of the projection
    // rb[b, :] = g[b, :] @ W2^T is the 16-row slice  contrib[b, chunk, o] = sum_c g[c0 + c] w2t[c0 + c, o]  (one fmaf chain), and
    // the LAST of the object's C / 16 workgroups to arrive (a ticket per object, taken after the slice's stores are acknowledged:
    // s_waitcnt vmcnt(0) by hand) adds the
    // slices in chunk order -- a fixed order whichever workgroup that is -- and hands the ticket back as 0.
    float *s_g = s_tab;                              // the table is no longer read
    int *s_last = reinterpret_cast<int *>(s_tab + ORL_CH);
    if (tid < ORL_CH) {
        float sum = 0.f;
        for (int pt = 0; pt < ptiles; ++pt) {
            const float *r = s_red + pt * 4 * ORL_CH + tid;
            sum += ((r[0] + r[ORL_CH]) + r[2 * ORL_CH]) + r[3 * ORL_CH];
        }
        sum = sum / (float)n;
        s_g[tid] = sum;
        if (g_out) g_out[(int64_t)b * C + c0 + tid] = sum;
    }
    __syncthreads();
    const int nch = C / ORL_CH;
    for (int o = tid; o < C; o += nthr) {
        float acc = 0.f;
        const float *w = w2t + (int64_t)c0 * C + o;
#pragma unroll
        for (int c = 0; c < ORL_CH; ++c) acc = fmaf(s_g[c], w[(int64_t)c * C], acc);
        // (agent-scope atomic store and load: they travel to the point where every XCD sees them, sc1, and stay out of the
        // fences.  A __threadfence() here is a write-back of the XCD's whole L2 per wave -- with the planes this kernel has just
        // written sitting in it the launch took 60-100 us instead of 13-29.)
        __hip_atomic_store(contrib + ((int64_t)b * nch + chunk) * C + o, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // The slice's stores must be ACKNOWLEDGED before the ticket is taken.  __syncthreads() does not do that: a workgroup-scope
    // release on gfx950 (all waves of a workgroup share one L1) carries no vmcnt wait for global stores -- the first build relied on
    // it, and the object's last workgroup could read a slice still in flight (one captured-vs-eager trainer comparison in ~6
    // failed).  Written by hand.  This hand-off lies outside the language's memory model: it is correct at ISA level on gfx9-family
    // targets because (a) agent-scope relaxed atomic stores compile to `global_store ... sc1`, which writes through to the point every
    // XCD reads from, (b) `s_waitcnt vmcnt(0)` (SIMM16 0x0f70 in the gfx9 encoding: vmcnt = 0, expcnt / lgkmcnt untouched) returns
    // only when those stores are acknowledged there, and (c) the last workgroup reads with `sc1` loads behind a barrier (the
    // micro-architecture guide's valid form "sc1 payload -> vmcnt(0) -> counter add by one lane behind a workgroup barrier").
    // Another target or a compiler that reorders relaxed atomics around the inline wait would break it silently: the gate is
    // test_orl_rowbias_one_launch_form's alternating-input repeats.
#if !defined(__gfx950__) && !defined(__gfx942__) && !defined(__gfx90a__) && defined(__HIP_DEVICE_COMPILE__)
#error "orl_lds_kernel's ticket hand-off hard-codes the gfx9 s_waitcnt encoding and sc1 write-through semantics: re-derive it for this target"
#endif
    __builtin_amdgcn_s_waitcnt(0x0f70);              // vmcnt(0)
    __syncthreads();
    if (tid == 0) *s_last = atomicAdd(tickets + b, 1) == nch - 1;
    __syncthreads();
    if (!*s_last) return;
    for (int o = tid; o < C; o += nthr) {
        float sum = 0.f;
        for (int ch0 = 0; ch0 < nch; ch0 += 8) {          // nch = 8, 16 or 32; eight loads in flight, added in chunk order
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                v[u] = __hip_atomic_load(contrib + ((int64_t)b * nch + ch0 + u) * C + o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int u = 0; u < 8; ++u) sum += v[u];
        }
        rb[(int64_t)b * C + o] = sum;
    }
    if (tid == 0) tickets[b] = 0;
}

#ifdef TGP_DEV   // development builds only: 0 = always the gather-from-L2 kernel; neighbour lists staged in LDS or read from memory
int tgp_orl_lds_mode = 1, tgp_orl_idx_mode = 1;
extern "C" void tgp_debug_set_orl_lds(int v) { tgp_orl_lds_mode = v; }
extern "C" void tgp_debug_set_orl_idx(int v) { tgp_orl_idx_mode = v; }
#else
static constexpr int tgp_orl_lds_mode = 1, tgp_orl_idx_mode = 1;
#endif

// returns true when the LDS form was launched
static bool orl_lds_launch(const float *feat, int ldf, const int32_t *idx, int B, int n, int k, int C, float *partial, int ptiles,
                           hipStream_t stream, int &rc, char *planes = nullptr, int kts = 0, uint32_t *amax = nullptr,
                           const float *xyz_tile = nullptr, const float *w2t = nullptr, float *g_out = nullptr, float *rb = nullptr,
                           float *contrib = nullptr, int *tickets = nullptr)
{
    rc = 0;
    size_t lds = ((size_t)n * ORL_CH + (size_t)ptiles * 4 * ORL_CH + (planes ? n / 32 + 4 : 0)) * sizeof(float);
    if (!tgp_orl_lds_mode || lds > 72 * 1024 || C % ORL_CH) return false;
    // the neighbour lists as 16-bit ids beside the table (n = 1028, k = 20: 41 KB) when they fit and can be read 16 bytes at a time
    const size_t idx_bytes = (size_t)n * k * 2;
    const int lds_idx = n <= 65535 && ((n * k) & 3) == 0 && (reinterpret_cast<uintptr_t>(idx) & 15) == 0 && lds + idx_bytes <= 150 * 1024 &&
                        tgp_orl_idx_mode;
    if (lds_idx) lds += idx_bytes;
    static TgpLdsAttr attr;
    if (lds > 64 * 1024) {
        rc = tgp_lds_attr(attr, reinterpret_cast<const void *>(orl_lds_kernel), 150 * 1024);
        if (rc) return true;
    }
    const int slots = ptiles * 4 * (ORL_CH / 2);
    const int threads = slots >= 768 ? 1024 : (slots >= 384 ? 512 : 256);
    hipLaunchKernelGGL(orl_lds_kernel, dim3(tgp_xcd_grid(B, C / ORL_CH)), dim3(threads), lds, stream, feat, ldf, idx, B, n, k, C, partial,
                       ptiles, planes, kts, amax, xyz_tile, w2t, g_out, rb, contrib, tickets, lds_idx);
    rc = TGP_LAUNCH_RESULT();
    return true;
}

__global__ void orl_finish_kernel(const float *__restrict__ partial, int B, int n, int C, int ptiles,
                                  float *__restrict__ out)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * C) return;
    const int b = t / C, c = t - b * C;
    float s = 0.f;
    for (int p = 0; p < ptiles; ++p) s += partial[((int64_t)b * ptiles + p) * C + c];
    out[t] = s / (float)n;
}

// Same as orl_finish_kernel, then rb[b, :] = g[b, :] @ W2^T: the "global" half of the layer's conv2
// (gcn3d.py:108-112: conv2(cat[f, g]) = W1 f + W2 g).  Workgroup = (object, 64 output channels); its 4 waves
// split the input channels and are combined through LDS in fixed order.  w2t is W2 transposed (C_in, C_out) so
// consecutive lanes read consecutive floats.
__global__ __launch_bounds__(256) void orl_finish_project_kernel(const float *__restrict__ partial, int n, int C, int ptiles,
                                                                 const float *__restrict__ w2t, float *__restrict__ g_out,
                                                                 float *__restrict__ rb)
{
    __shared__ float g[512];
    __shared__ float part[4][64];
    const int b = blockIdx.x, ob = blockIdx.y * 64;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float s = 0.f;
        for (int p = 0; p < ptiles; ++p) s += partial[((int64_t)b * ptiles + p) * C + c];
        s = s / (float)n;
        g[c] = s;
        if (g_out && blockIdx.y == 0) g_out[(int64_t)b * C + c] = s;
    }
    __syncthreads();
    const int o = threadIdx.x & 63, ks = threadIdx.x >> 6;
    const int per = C / 4;
    float acc = 0.f;
    const float *w = w2t + (int64_t)(ks * per) * C + ob + o;
#pragma unroll 8
    for (int c = 0; c < per; ++c) acc = fmaf(g[ks * per + c], w[(int64_t)c * C], acc);
    part[ks][o] = acc;
    __syncthreads();
    if (ks == 0) rb[(int64_t)b * C + ob + o] = ((part[0][o] + part[1][o]) + part[2][o]) + part[3][o];
}

extern "C" int64_t tgp_orl_partial_floats(int B, int n, int C)
{
    if (B <= 0 || n <= 0 || C <= 0) return 0;
    return (int64_t)B * tgp_cdiv(n, ORL_PTS) * C;
}

extern "C" int tgp_orl_global(const float *feat, int ldf, const int32_t *idx, int B, int n, int k, int C, float *partial,
                              float *out, tgp_stream_t stream)
{
    TGP_REQUIRE(feat && idx && partial && out && B > 0 && n > 0 && k > 0);
    if (k > GC_MAXK || !(C == 128 || C == 256 || C == 512)) return TGP_EUNSUPPORTED;
    TGP_REQUIRE(ldf >= C && (ldf & 3) == 0 && (reinterpret_cast<uintptr_t>(feat) & 15) == 0 &&
                (reinterpret_cast<uintptr_t>(partial) & 15) == 0);
    const int ptiles = tgp_cdiv(n, ORL_PTS);
#define ORL_GO(CC)                                                                                                   \
    {                                                                                                                \
        const int tiles = ptiles * RowLanes<CC>::CHUNKS;                                                             \
        hipLaunchKernelGGL(orl_partial_kernel<CC>, dim3(tgp_xcd_grid(B, tiles)), dim3(256), 0, tgp_hs(stream), feat,  \
                           ldf, idx, B, n, k, partial, ptiles, tiles);                                               \
    }
    int lrc = 0;
    if (orl_lds_launch(feat, ldf, idx, B, n, k, C, partial, ptiles, tgp_hs(stream), lrc)) {
        if (lrc) return lrc;
    } else if (C == 128) ORL_GO(128) else if (C == 256) ORL_GO(256) else ORL_GO(512)
#undef ORL_GO
    hipLaunchKernelGGL(orl_finish_kernel, dim3(tgp_cdiv(B * C, 256)), dim3(256), 0, tgp_hs(stream), partial, B, n, C,
                       ptiles, out);
    return TGP_LAUNCH_RESULT();
}

extern "C" int tgp_orl_rowbias(const float *feat, int ldf, const int32_t *idx, int B, int n, int k, int C, float *partial,
                               const float *w2t, float *g_out, float *rb, tgp_stream_t stream)
{
    TGP_REQUIRE(feat && idx && partial && w2t && rb && B > 0 && n > 0 && k > 0);
    if (k > GC_MAXK || !(C == 128 || C == 256 || C == 512)) return TGP_EUNSUPPORTED;
    TGP_REQUIRE(ldf >= C && (ldf & 3) == 0 && (reinterpret_cast<uintptr_t>(feat) & 15) == 0 &&
                (reinterpret_cast<uintptr_t>(partial) & 15) == 0);
    const int ptiles = tgp_cdiv(n, ORL_PTS);
#define ORL_GO(CC)                                                                                                   \
    {                                                                                                                \
        const int tiles = ptiles * RowLanes<CC>::CHUNKS;                                                             \
        hipLaunchKernelGGL(orl_partial_kernel<CC>, dim3(tgp_xcd_grid(B, tiles)), dim3(256), 0, tgp_hs(stream), feat,  \
                           ldf, idx, B, n, k, partial, ptiles, tiles);                                               \
    }
    int lrc = 0;
    if (orl_lds_launch(feat, ldf, idx, B, n, k, C, partial, ptiles, tgp_hs(stream), lrc)) {
        if (lrc) return lrc;
    } else if (C == 128) ORL_GO(128) else if (C == 256) ORL_GO(256) else ORL_GO(512)
#undef ORL_GO
    hipLaunchKernelGGL(orl_finish_project_kernel, dim3(B, C / 64), dim3(256), 0, tgp_hs(stream), partial, n, C, ptiles, w2t, g_out,
                       rb);
    return TGP_LAUNCH_RESULT();
}

// tgp_orl_rowbias that also leaves feat -- the A operand of the layer's last GEMM -- as blocked fp16 planes (see orl_lds_kernel);
// only where the LDS form serves the shape: TGP_EUNSUPPORTED otherwise (the caller then uses tgp_orl_rowbias and an fp32 operand)
extern "C" int tgp_orl_rowbias_planes(const float *feat, int ldf, const int32_t *idx, int B, int n, int k, int C, float *partial,
                                      const float *w2t, float *g_out, float *rb, void *planes, int kts, uint32_t *amax,
                                      const float *xyz_tile, tgp_stream_t stream)
{
    TGP_REQUIRE(feat && idx && partial && w2t && rb && planes && B > 0 && n > 0 && k > 0);
    if (k > GC_MAXK || !(C == 128 || C == 256 || C == 512)) return TGP_EUNSUPPORTED;
    TGP_REQUIRE(ldf >= C && (ldf & 3) == 0 && (reinterpret_cast<uintptr_t>(feat) & 15) == 0 &&
                (reinterpret_cast<uintptr_t>(partial) & 15) == 0 && (reinterpret_cast<uintptr_t>(planes) & 15) == 0);
    TGP_REQUIRE(kts >= C / 16 + (xyz_tile ? 1 : 0));
    const int ptiles = tgp_cdiv(n, ORL_PTS);
    const size_t lds = ((size_t)n * ORL_CH + (size_t)ptiles * 4 * ORL_CH + n / 32 + 4) * sizeof(float);
    if (!tgp_orl_lds_mode || lds > 72 * 1024) return TGP_EUNSUPPORTED;
    int lrc = 0;
    if (!orl_lds_launch(feat, ldf, idx, B, n, k, C, partial, ptiles, tgp_hs(stream), lrc, reinterpret_cast<char *>(planes), kts, amax,
                        xyz_tile))
        return TGP_EUNSUPPORTED;
    if (lrc) return lrc;
    hipLaunchKernelGGL(orl_finish_project_kernel, dim3(B, C / 64), dim3(256), 0, tgp_hs(stream), partial, n, C, ptiles, w2t, g_out,
                       rb);
    return TGP_LAUNCH_RESULT();
}

// tgp_orl_rowbias[_planes] as ONE launch: the mean over points and the projection are finished inside the pooling kernel (see
// orl_lds_kernel).  contrib: B * (C / 16) * C floats of scratch; tickets: B ints, zero on entry and zero again on return (one
// buffer serves every call that is ordered after the previous one).  planes may be NULL.  rb sums its C / 16 slices in another
// order than orl_finish_project_kernel: the two forms agree to rounding, not bit for bit.  TGP_EUNSUPPORTED where the LDS form
// does not serve the shape (nothing launched).
extern "C" int tgp_orl_rowbias_fused(const float *feat, int ldf, const int32_t *idx, int B, int n, int k, int C, const float *w2t,
                                     float *g_out, float *rb, void *planes, int kts, uint32_t *amax, const float *xyz_tile,
                                     float *contrib, int32_t *tickets, tgp_stream_t stream)
{
    TGP_REQUIRE(feat && idx && w2t && rb && contrib && tickets && B > 0 && n > 0 && k > 0);
    if (k > GC_MAXK || !(C == 128 || C == 256 || C == 512)) return TGP_EUNSUPPORTED;
    TGP_REQUIRE(ldf >= C && (ldf & 3) == 0 && (reinterpret_cast<uintptr_t>(feat) & 15) == 0 &&
                (reinterpret_cast<uintptr_t>(planes) & 15) == 0);
    TGP_REQUIRE(!planes || kts >= C / 16 + (xyz_tile ? 1 : 0));
    const int ptiles = tgp_cdiv(n, ORL_PTS);
    const size_t lds = ((size_t)n * ORL_CH + (size_t)ptiles * 4 * ORL_CH + n / 32 + 4) * sizeof(float);
    if (!tgp_orl_lds_mode || lds > 72 * 1024 || n < 2) return TGP_EUNSUPPORTED;
    int lrc = 0;
    if (!orl_lds_launch(feat, ldf, idx, B, n, k, C, nullptr, ptiles, tgp_hs(stream), lrc, reinterpret_cast<char *>(planes), kts, amax,
                        xyz_tile, w2t, g_out, rb, contrib, tickets))
        return TGP_EUNSUPPORTED;
    return lrc;
}

// ---------------------------------------------------------------------------------------------------
// Pool_layer: neighbour max at the sampled rows only (the reference computes all n rows and keeps n/4)
// planes != NULL (round 4): the pooled features -- the A operand of the next layer's projection GEMM -- are also written as blocked
// fp16 planes (include/tgpose.h, tgp_gemm_args.A_planes) with their per-row-block magnitude words; C4 a divisor of 64, so a wave
// holds whole rows and at most two row blocks
__global__ void pool_kernel(const float *__restrict__ xyz, const float *__restrict__ feat, int ldf,
                            const int32_t *__restrict__ idx, int ldi, const int32_t *__restrict__ sample, int B, int n,
                            int n_out, int kpool, int C4, float *__restrict__ out_xyz, float *__restrict__ out_f, int ldo,
                            char *__restrict__ planes, int kts, uint32_t *__restrict__ amax)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = t < (int64_t)B * n_out * C4;
    const int64_t tc = live ? t : 0;
    const int c4 = (int)(tc % C4);
    const int64_t pm = tc / C4;
    const int mrow = (int)(pm % n_out), b = (int)(pm / n_out);
    const int s = sample[mrow];
    const int32_t *nb = idx + ((int64_t)b * n + s) * ldi;
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    for (int j = 0; j < kpool; ++j) {
        const float4 v = *reinterpret_cast<const float4 *>(feat + ((int64_t)b * n + nb[j]) * ldf + 4 * c4);
        m.x = fmaxf(m.x, v.x), m.y = fmaxf(m.y, v.y), m.z = fmaxf(m.z, v.z), m.w = fmaxf(m.w, v.w);
    }
    if (live) {
        *reinterpret_cast<float4 *>(out_f + pm * ldo + 4 * c4) = m;
        if (c4 == 0) {
            const float *p = xyz + ((int64_t)b * n + s) * 3;
            float *o = out_xyz + pm * 3;
            o[0] = p[0], o[1] = p[1], o[2] = p[2];
        }
    }
    if (planes) {                                                    // (kernel-uniform)
        uint32_t mb = 0u;
        if (live) {
            uint2 ph, pl;
            ol_split(m, ph, pl);
            const int pc = 4 * c4;
            char *dst = planes + ((pm >> 5) * kts + (pc >> 4)) * 2048 + ((pc >> 3) & 1) * 512 + (int)(pm & 31) * 16 + (pc & 4) * 2;
            *reinterpret_cast<uint2 *>(dst) = ph;
            *reinterpret_cast<uint2 *>(dst + 1024) = pl;
            const uint32_t b0 = __float_as_uint(m.x) & 0x7fffffffu, b1 = __float_as_uint(m.y) & 0x7fffffffu;
            const uint32_t b2 = __float_as_uint(m.z) & 0x7fffffffu, b3 = __float_as_uint(m.w) & 0x7fffffffu;
            mb = max(max(b0, b1), max(b2, b3));
        }
        if (amax) {
            // the wave's rows lie in at most two row blocks: the first lane's and the next
            const int64_t rb0 = __shfl(pm >> 5, 0, 64);
            uint32_t m0 = (live && (pm >> 5) == rb0) ? mb : 0u, m1 = (live && (pm >> 5) != rb0) ? mb : 0u;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                m0 = max(m0, (uint32_t)__shfl_xor((int)m0, o, 64));
                m1 = max(m1, (uint32_t)__shfl_xor((int)m1, o, 64));
            }
            if ((threadIdx.x & 63) == 0) {
                if (m0) atomicMax(amax + rb0, m0);
                if (m1) atomicMax(amax + rb0 + 1, m1);
            }
        }
    }
}

extern "C" int tgp_pool_fwd(const float *xyz, const float *feat, int ldf, const int32_t *idx, int ldi,
                            const int32_t *sample, int B, int n, int n_out, int kpool, int C, float *out_xyz, float *out_f,
                            int ldo, tgp_stream_t stream)
{
    TGP_REQUIRE(xyz && feat && idx && sample && out_xyz && out_f);
    TGP_REQUIRE(B > 0 && n > 0 && n_out > 0 && n_out <= n && kpool > 0 && ldi >= kpool && C > 0 && (C & 3) == 0);
    TGP_REQUIRE(ldf >= C && ldo >= C && (ldf & 3) == 0 && (ldo & 3) == 0);
    TGP_REQUIRE((reinterpret_cast<uintptr_t>(feat) & 15) == 0 && (reinterpret_cast<uintptr_t>(out_f) & 15) == 0);
    const int64_t total = (int64_t)B * n_out * (C / 4);
    hipLaunchKernelGGL(pool_kernel, dim3(tgp_cdiv(total, 256)), dim3(256), 0, tgp_hs(stream), xyz, feat, ldf, idx, ldi,
                       sample, B, n, n_out, kpool, C / 4, out_xyz, out_f, ldo, nullptr, 0, nullptr);
    return TGP_LAUNCH_RESULT();
}

extern "C" int tgp_pool_fwd_planes(const float *xyz, const float *feat, int ldf, const int32_t *idx, int ldi,
                                   const int32_t *sample, int B, int n, int n_out, int kpool, int C, float *out_xyz, float *out_f,
                                   int ldo, void *planes, int kts, uint32_t *amax, tgp_stream_t stream)
{
    TGP_REQUIRE(xyz && feat && idx && sample && out_xyz && out_f && planes);
    TGP_REQUIRE(B > 0 && n > 0 && n_out > 0 && n_out <= n && kpool > 0 && ldi >= kpool && C > 0 && (C & 15) == 0);
    TGP_REQUIRE(ldf >= C && ldo >= C && (ldf & 3) == 0 && (ldo & 3) == 0 && kts >= C / 16);
    TGP_REQUIRE((reinterpret_cast<uintptr_t>(feat) & 15) == 0 && (reinterpret_cast<uintptr_t>(out_f) & 15) == 0 &&
                (reinterpret_cast<uintptr_t>(planes) & 15) == 0);
    if (64 % (C / 4)) return TGP_EUNSUPPORTED;                          // a wave holds whole rows: C = 16 .. 256 in powers of two
    const int64_t total = (int64_t)B * n_out * (C / 4);
    hipLaunchKernelGGL(pool_kernel, dim3(tgp_cdiv(total, 256)), dim3(256), 0, tgp_hs(stream), xyz, feat, ldf, idx, ldi,
                       sample, B, n, n_out, kpool, C / 4, out_xyz, out_f, ldo, reinterpret_cast<char *>(planes), kts, amax);
    return TGP_LAUNCH_RESULT();
}

__global__ void gather_rows_kernel(const float *__restrict__ src, int lds, const int32_t *__restrict__ idx, int B,
                                   int n_src, int n_out, int C4, float *__restrict__ dst, int ldd)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)B * n_out * C4) return;
    const int c4 = (int)(t % C4);
    const int64_t row = t / C4;
    const int b = (int)(row / n_out);
    const int s = idx[row];
    *reinterpret_cast<float4 *>(dst + row * ldd + 4 * c4) =
        *reinterpret_cast<const float4 *>(src + ((int64_t)b * n_src + s) * lds + 4 * c4);
}

extern "C" int tgp_gather_rows(const float *src, int lds, const int32_t *idx, int B, int n_src, int n_out, int C,
                               float *dst, int ldd, tgp_stream_t stream)
{
    TGP_REQUIRE(src && idx && dst && B > 0 && n_src > 0 && n_out > 0 && C > 0 && (C & 3) == 0);
    TGP_REQUIRE(lds >= C && ldd >= C && (lds & 3) == 0 && (ldd & 3) == 0);
    TGP_REQUIRE((reinterpret_cast<uintptr_t>(src) & 15) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0);
    const int64_t total = (int64_t)B * n_out * (C / 4);
    hipLaunchKernelGGL(gather_rows_kernel, dim3(tgp_cdiv(total, 256)), dim3(256), 0, tgp_hs(stream), src, lds, idx, B,
                       n_src, n_out, C / 4, dst, ldd);
    return TGP_LAUNCH_RESULT();
}

// tail columns of the concat buffer: one-hot category | centred xyz | zero padding
__global__ void fill_tail_kernel(const float *__restrict__ obj_id, const float *__restrict__ xyz_c, int n, int64_t rows,
                                 int n_cls, float *__restrict__ feat, int ld, int col0)
{
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= rows) return;
    const int b = (int)(row / n);
    const int cls = (int)obj_id[b]; // obj_idh.long()  (FaceRecon.py:54)
    float *f = feat + row * ld + col0;
    for (int c = 0; c < n_cls; ++c) f[c] = (c == cls) ? 1.f : 0.f;
    f[n_cls + 0] = xyz_c[row * 3 + 0];
    f[n_cls + 1] = xyz_c[row * 3 + 1];
    f[n_cls + 2] = xyz_c[row * 3 + 2];
    for (int c = col0 + n_cls + 3; c < ld; ++c) feat[row * ld + c] = 0.f;
}

extern "C" int tgp_fill_tail(const float *obj_id, const float *xyz_c, int B, int n, int n_cls, float *feat, int ld,
                             int col0, tgp_stream_t stream)
{
    TGP_REQUIRE(obj_id && xyz_c && feat && B > 0 && n > 0 && n_cls > 0 && col0 >= 0 && ld >= col0 + n_cls + 3);
    const int64_t rows = (int64_t)B * n;
    hipLaunchKernelGGL(fill_tail_kernel, dim3(tgp_cdiv(rows, 256)), dim3(256), 0, tgp_hs(stream), obj_id, xyz_c, n, rows,
                       n_cls, feat, ld, col0);
    return TGP_LAUNCH_RESULT();
}
