// Backward of the graph layers (gcn3d.py:91-106 HSlayer_surface.graph_conv, :157-180 HS_layer.graph_conv, :210-217
// get_ORL_global, :225-245 Pool_layer, FaceRecon.py:66-72 nearest upsampling), gfx950.  First version: one thread per
// (point, channel) recomputes the forward's arg-max over the neighbours and scatters with hardware float atomics (the
// reference's own CUDA autograd scatters with atomics too: index_add / max backward); the support-direction gradients are
// reduced deterministically through per-workgroup partials.
#include "tgp_common.h"
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

#define GB_S 7
#define GB_PTS 16   // points per workgroup and channel lane
#define GB_MAXK 64
#define GB_GROUP 32  // partials summed per workgroup in the first reduction level

// thread -> (channel c, point stream): blockDim 256; C <= 256: 256 / C point streams per workgroup
struct GbMap {
    int c, stream, streams, chunk;
};
__device__ __forceinline__ GbMap gb_map(int C, int ychunk)
{
    GbMap m;
    const int cw = C < 256 ? C : 256;           // channels per workgroup
    m.streams = 256 / cw;
    m.stream = threadIdx.x / cw;
    m.c = ychunk * cw + threadIdx.x % cw;       // ychunk: channel chunk
    m.chunk = cw;
    return m;
}

// out = centre + mean_s max_j relu(dir_j . D_s) * support[nbr_j][s]      (SURFACE: without centre / support)
//   dproj[p][c]                 += dg[p][c]                                  (centre)
//   dproj[nbr*][C + s C + c]    += dg[p][c] / 7 * theta*                      (support, atomics)
//   dsdn[:, s C + c]            += dg[p][c] / 7 * support* * dir_j*  if theta* > 0   (partials per workgroup stream)
template <bool SURFACE>
__global__ __launch_bounds__(256) void gconv_bwd_kernel(const float *__restrict__ xyz, const int32_t *__restrict__ idx,
                                                        const float *__restrict__ proj, int ldp, const float *__restrict__ sdn,
                                                        const float *__restrict__ dg, int ldg, int B, int n, int k, int C,
                                                        float *__restrict__ dproj, int lddp, float *__restrict__ dsdn_partial)
{
    // (round 3) an object's workgroups share an XCD, as in the forward kernel: its projection table (n x 8C floats) is then served
    // from ONE L2 instead of being pulled through all eight
    const int ychunks = C < 256 ? 1 : C / 256;
    const int SC = GB_S * C;
    const int streams_ = 256 / (C < 256 ? C : 256);
    const int tiles_per_obj = (n + GB_PTS * streams_ - 1) / (GB_PTS * streams_);
    int b, t_;
    if (!tgp_xcd_object_tile(blockIdx.x, B, tiles_per_obj * ychunks, b, t_)) return;
    const int tile = t_ / ychunks;
    const GbMap mp = gb_map(C, t_ % ychunks);
    const int bx = b * tiles_per_obj + tile;      // the workgroup's slot in the partial buffer
    const int c = mp.c;
    float D[GB_S][3], dD[GB_S][3];
#pragma unroll
    for (int s = 0; s < GB_S; ++s)
#pragma unroll
        for (int a = 0; a < 3; ++a) D[s][a] = c < C ? sdn[a * SC + s * C + c] : 0.f, dD[s][a] = 0.f;

    // neighbour ids and unit directions of the workgroup's points, once, into LDS: the per-channel loops below then depend
    // on global memory only for the support rows (independent loads the compiler can keep in flight)
    __shared__ int s_nb[2 * GB_PTS][GB_MAXK];
    __shared__ float s_dir[2 * GB_PTS][GB_MAXK][3];
    {
        const int pts = GB_PTS * mp.streams;
        const int base = tile * pts;
        for (int e = threadIdx.x; e < pts * k; e += 256) {
            const int lp = e / k, j = e - lp * k;
            const int i = base + lp;
            int nb = 0;
            float ux = 0.f, uy = 0.f, uz = 0.f;
            if (i < n) {
                const int64_t rowi = (int64_t)b * n + i;
                nb = idx[rowi * k + j];
                const int64_t rown = (int64_t)b * n + nb;
                ux = xyz[rown * 3] - xyz[rowi * 3], uy = xyz[rown * 3 + 1] - xyz[rowi * 3 + 1], uz = xyz[rown * 3 + 2] - xyz[rowi * 3 + 2];
                const float nrm = fmaxf(sqrtf((ux * ux + uy * uy) + uz * uz), 1e-12f);
                ux = ux / nrm, uy = uy / nrm, uz = uz / nrm;
            }
            s_nb[lp][j] = nb;
            s_dir[lp][j][0] = ux, s_dir[lp][j][1] = uy, s_dir[lp][j][2] = uz;
        }
        __syncthreads();
    }
    const int p0 = (tile * mp.streams + mp.stream) * GB_PTS;
    for (int pp = 0; pp < GB_PTS; ++pp) {
        const int i = p0 + pp;
        if (i >= n || c >= C) break;
        const int lp = mp.stream * GB_PTS + pp;
        const int64_t rowi = (int64_t)b * n + i;
        const float g = dg[rowi * ldg + c] / 7.0f;
        float best[GB_S], bth[GB_S], bsup[GB_S];
        int barg[GB_S], bj[GB_S];
#pragma unroll
        for (int s = 0; s < GB_S; ++s) best[s] = SURFACE ? 0.f : -INFINITY, barg[s] = -1, bj[s] = 0, bth[s] = 0.f, bsup[s] = 0.f;
#pragma unroll 4
        for (int j = 0; j < k; ++j) {
            const int nb = s_nb[lp][j];
            const int64_t rown = (int64_t)b * n + nb;
            const float ux = s_dir[lp][j][0], uy = s_dir[lp][j][1], uz = s_dir[lp][j][2];
            float sup[GB_S];
#pragma unroll
            for (int s = 0; s < GB_S; ++s) sup[s] = SURFACE ? 1.f : proj[rown * ldp + C + s * C + c];
#pragma unroll
            for (int s = 0; s < GB_S; ++s) {
                float th = fmaf(uz, D[s][2], fmaf(uy, D[s][1], ux * D[s][0]));   // same expression as the forward kernel
                th = fmaxf(th, 0.f);
                const float v = SURFACE ? th : th * sup[s];
                if (v > best[s]) best[s] = v, barg[s] = nb, bj[s] = j, bth[s] = th, bsup[s] = sup[s];   // first maximum wins
            }
        }
        if (!SURFACE) unsafeAtomicAdd(dproj + rowi * lddp + c, dg[rowi * ldg + c]);
#pragma unroll
        for (int s = 0; s < GB_S; ++s) {
            if (barg[s] < 0) continue;     // SURFACE: every theta was 0 -> max = 0 from the relu floor, no gradient
            if (!SURFACE && bth[s] != 0.f) unsafeAtomicAdd(dproj + ((int64_t)b * n + barg[s]) * lddp + C + s * C + c, g * bth[s]);
            if (bth[s] > 0.f) {
                const float dth = SURFACE ? g : g * bsup[s];
                dD[s][0] += dth * s_dir[lp][bj[s]][0], dD[s][1] += dth * s_dir[lp][bj[s]][1], dD[s][2] += dth * s_dir[lp][bj[s]][2];
            }
        }
    }
    if (c < C) {
        float *o = dsdn_partial + ((int64_t)bx * mp.streams + mp.stream) * 3 * SC;
#pragma unroll
        for (int s = 0; s < GB_S; ++s)
#pragma unroll
            for (int a = 0; a < 3; ++a) o[a * SC + s * C + c] = dD[s][a];
    }
}

// out[group][t] = sum of up to `group_size` consecutive parts (fixed order): run twice for a two-level deterministic sum
__global__ void partial_sum_kernel(const float *__restrict__ partial, int64_t parts, int64_t width, int64_t group_size,
                                   float *__restrict__ out)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= width) return;
    const int64_t q0 = (int64_t)blockIdx.y * group_size;
    const int64_t q1 = q0 + group_size < parts ? q0 + group_size : parts;
    float s = 0.f;
    for (int64_t q = q0; q < q1; ++q) s += partial[q * width + t];
    out[(int64_t)blockIdx.y * width + t] = s;
}

static int64_t gb_parts(int B, int n, int C)
{
    const int cw = C < 256 ? C : 256;
    const int streams = 256 / cw;
    return (int64_t)B * tgp_cdiv(n, GB_PTS * streams) * streams;
}

extern "C" int64_t tgp_gconv_bwd_workspace_floats(int B, int n, int C)
{
    if (B <= 0 || n <= 0 || C <= 0) return 0;
    const int64_t parts = gb_parts(B, n, C);
    return (parts + tgp_cdiv(parts, (int64_t)GB_GROUP)) * 3 * GB_S * C;
}

static int gconv_bwd_launch(bool surface, const float *xyz, const int32_t *idx, const float *proj, int ldp, const float *sdn,
                            const float *dg, int ldg, int B, int n, int k, int S, int C, float *dproj, int lddp, float *dsdn,
                            float *workspace, hipStream_t stream)
{
    if (!(xyz && idx && sdn && dg && dsdn && workspace) || B <= 0 || n <= 0 || k <= 0 || k > GB_MAXK || S != GB_S || C <= 0 ||
        ldg < C || (C > 256 && C % 256 != 0) || (C < 256 && 256 % C != 0))
        return TGP_EINVAL;
    const int cw = C < 256 ? C : 256;
    const int streams = 256 / cw;
    const dim3 grid(tgp_xcd_grid(B, tgp_cdiv(n, GB_PTS * streams) * tgp_cdiv(C, cw)));
    // every (workgroup, stream) writes its full 3 x S x C-chunk slice, so the partial buffer needs no clearing -- but the
    // chunks of different blockIdx.y share a slice: they write disjoint channels of it
    if (surface)
        hipLaunchKernelGGL(gconv_bwd_kernel<true>, grid, dim3(256), 0, stream, xyz, idx, proj, ldp, sdn, dg, ldg, B, n, k, C, dproj,
                           lddp, workspace);
    else
        hipLaunchKernelGGL(gconv_bwd_kernel<false>, grid, dim3(256), 0, stream, xyz, idx, proj, ldp, sdn, dg, ldg, B, n, k, C, dproj,
                           lddp, workspace);
    const int64_t width = 3 * (int64_t)GB_S * C, parts = gb_parts(B, n, C);
    const int groups = tgp_cdiv(parts, (int64_t)GB_GROUP);
    float *level2 = workspace + parts * width;
    hipLaunchKernelGGL(partial_sum_kernel, dim3(tgp_cdiv(width, (int64_t)256), groups), dim3(256), 0, stream, workspace, parts,
                       width, (int64_t)GB_GROUP, level2);
    hipLaunchKernelGGL(partial_sum_kernel, dim3(tgp_cdiv(width, (int64_t)256), 1), dim3(256), 0, stream, level2, (int64_t)groups,
                       width, (int64_t)groups, dsdn);
    return TGP_LAUNCH_RESULT();
}

extern "C" int tgp_gconv_surface_bwd(const float *xyz, const int32_t *idx, const float *sdn, const float *dg, int ldg, int B, int n,
                                     int k, int S, int C, float *dsdn, float *workspace, tgp_stream_t stream)
{
    return gconv_bwd_launch(true, xyz, idx, nullptr, 0, sdn, dg, ldg, B, n, k, S, C, nullptr, 0, dsdn, workspace, tgp_hs(stream));
}

extern "C" int tgp_gconv_hs_bwd(const float *xyz, const int32_t *idx, const float *proj, int ldp, const float *sdn, const float *dg,
                                int ldg, int B, int n, int k, int S, int C, float *dproj, int lddp, float *dsdn, float *workspace,
                                tgp_stream_t stream)
{
    TGP_REQUIRE(proj && dproj && ldp >= 8 * C && lddp >= 8 * C);
    return gconv_bwd_launch(false, xyz, idx, proj, ldp, sdn, dg, ldg, B, n, k, S, C, dproj, lddp, dsdn, workspace, tgp_hs(stream));
}

// ---------------------------------------------------------------------------------------------------
// y[p][c] = max_j src[idx[p][j]][c]  (ORL pooling, Pool_layer): dsrc[idx[p][j*]][c] += dy
//   per_object != 0: dy is (B, C) and every point of the object receives dy[b][c] * scale (the mean over points of
//   get_ORL_global); otherwise dy is (B * n_rows, C).
__global__ void nbrmax_bwd_kernel(const float *__restrict__ src, int lds_, const int32_t *__restrict__ idx, int B, int n_src,
                                  int n_rows, int k, int C, const float *__restrict__ dy, int lddy, int per_object, float scale,
                                  float *__restrict__ dsrc, int ldds)
{
    const int c = blockIdx.y * blockDim.x + threadIdx.x;
    const int64_t row = blockIdx.x;            // b * n_rows + p
    if (c >= C) return;
    const int b = (int)(row / n_rows);
    const int32_t *nb = idx + row * k;
    float best = 0.f;
    int arg = -1;
    for (int j = 0; j < k; ++j) {
        const int q = nb[j];
        const float v = src[((int64_t)b * n_src + q) * lds_ + c];
        if (arg < 0 || v > best) best = v, arg = q;
    }
    const float g = per_object ? dy[(int64_t)b * lddy + c] * scale : dy[row * lddy + c];
    unsafeAtomicAdd(dsrc + ((int64_t)b * n_src + arg) * ldds + c, g);
}

extern "C" int tgp_nbrmax_bwd(const float *src, int ld_src, const int32_t *idx, int B, int n_src, int n_rows, int k, int C,
                              const float *dy, int lddy, int per_object, float scale, float *dsrc, int ld_dsrc, tgp_stream_t stream)
{
    TGP_REQUIRE(src && idx && dy && dsrc && B > 0 && n_src > 0 && n_rows > 0 && k > 0 && C > 0);
    TGP_REQUIRE(ld_src >= C && lddy >= C && ld_dsrc >= C);
    const int bx = C < 256 ? ((C + 63) / 64) * 64 : 256;
    hipLaunchKernelGGL(nbrmax_bwd_kernel, dim3((unsigned)((int64_t)B * n_rows), tgp_cdiv(C, bx)), dim3(bx), 0, tgp_hs(stream), src,
                       ld_src, idx, B, n_src, n_rows, k, C, dy, lddy, per_object, scale, dsrc, ld_dsrc);
    return TGP_LAUNCH_RESULT();
}

// y[p] = src[idx[p]] (nearest upsampling): dsrc[idx[p]] += dy[p]
__global__ void gather_rows_bwd_kernel(const float *__restrict__ dy, int lddy, const int32_t *__restrict__ idx, int n_src, int n_out,
                                       int C, float *__restrict__ dsrc, int ldds)
{
    const int c = blockIdx.y * blockDim.x + threadIdx.x;
    const int64_t row = blockIdx.x;
    if (c >= C) return;
    const int b = (int)(row / n_out);
    unsafeAtomicAdd(dsrc + ((int64_t)b * n_src + idx[row]) * ldds + c, dy[row * lddy + c]);
}

extern "C" int tgp_gather_rows_bwd(const float *dy, int lddy, const int32_t *idx, int B, int n_src, int n_out, int C, float *dsrc,
                                   int ld_dsrc, tgp_stream_t stream)
{
    TGP_REQUIRE(dy && idx && dsrc && B > 0 && n_src > 0 && n_out > 0 && C > 0 && lddy >= C && ld_dsrc >= C);
    const int bx = C < 256 ? ((C + 63) / 64) * 64 : 256;
    hipLaunchKernelGGL(gather_rows_bwd_kernel, dim3((unsigned)((int64_t)B * n_out), tgp_cdiv(C, bx)), dim3(bx), 0, tgp_hs(stream), dy,
                       lddy, idx, n_src, n_out, C, dsrc, ld_dsrc);
    return TGP_LAUNCH_RESULT();
}
