// Scatter-free backward of the graph layers (round 3).  The first version (csrc/gconv_bwd.hip) sends every gradient element to
// its arg-max neighbour with a float atomic, as the reference's CUDA autograd does (index_add / max backward).  On this part a
// device-scope float atomic is resolved beyond the XCDs' L2s: every such kernel ran at 30-35 G atomics/s whatever its mapping
// (conv_1's graph convolution: 33.7 M atomics, 664 us against 141 us for its forward; profiles/r03_b_*), in an order that changes
// from run to run.  Here the neighbour lists are inverted once per graph (tgp_reverse_graph: for every source row q the (point,
// slot) pairs that list q, sorted) and the backward runs as two dense passes:
//   pass 1, per point:   recompute the forward's arg-max, store its SLOT (one byte) -- and for the graph convolution the value
//                        that travels, g / 7 * theta* -- ; the direction gradients go through per-workgroup partials as before;
//   pass 2, per source:  walk the reverse list and sum the entries whose stored slot names this source.
// Dense coalesced reads and writes, no atomics, no zero fill, bit-repeatable.
#include "tgp_common.h"

#define RG_THREADS 1024
#define RG_MAX_SRC 8192
#define RG_LDS_MAX (150 * 1024)

// One workgroup per object (x a few that repeat the cheap histogram and fill and rank a share of the entries each).  idx (B, n_rows, k): ids in [0, n_src).  rptr[b * n_src + q] = first slot of q's list in rent
// (global offset b * n_rows * k + ...), rent[...] = (p << 6) | j for every (p, j) with idx[b][p][j] == q, ascending.
__global__ __launch_bounds__(RG_THREADS) void reverse_graph_kernel(const int32_t *__restrict__ idx, int B, int n_rows, int k, int n_src,
                                                                   int32_t *__restrict__ rptr, int32_t *__restrict__ rent)
{
    extern __shared__ int s_dyn[];
    __shared__ int s_part[RG_THREADS];
    const int E = n_rows * k;
    int *s_ent = s_dyn, *s_cnt = s_dyn + E, *s_start = s_cnt + n_src;
    const int b = blockIdx.x, tid = threadIdx.x;              // blockIdx.y: which share of the entries this workgroup ranks and writes
    const int32_t *ib = idx + (size_t)b * E;
    for (int q = tid; q < n_src; q += RG_THREADS) s_cnt[q] = 0;
    __syncthreads();
    for (int e = tid; e < E; e += RG_THREADS) {
        int q = ib[e];
        q = q < 0 ? 0 : (q >= n_src ? n_src - 1 : q);     // (an id outside [0, n_src) would be the caller's bug: clamped, never a fault)
        atomicAdd(&s_cnt[q], 1);                          // integer counts: order-free
    }
    __syncthreads();
    const int per = (n_src + RG_THREADS - 1) / RG_THREADS;
    const int q0 = tid * per;
    int local = 0;
    for (int q = q0; q < q0 + per && q < n_src; ++q) local += s_cnt[q];
    s_part[tid] = local;
    __syncthreads();
    for (int d = 1; d < RG_THREADS; d <<= 1) {
        const int v = tid >= d ? s_part[tid - d] : 0;
        __syncthreads();
        s_part[tid] += v;
        __syncthreads();
    }
    int run = s_part[tid] - local;
    for (int q = q0; q < q0 + per && q < n_src; ++q) {
        s_start[q] = run;
        if (blockIdx.y == 0) rptr[(size_t)b * n_src + q] = b * E + run;
        run += s_cnt[q];
        s_cnt[q] = 0;                                     // becomes the fill counter
    }
    if (b == B - 1 && tid == 0 && blockIdx.y == 0) rptr[(size_t)B * n_src] = B * E;
    __syncthreads();
    for (int e = tid; e < E; e += RG_THREADS) {
        int q = ib[e];
        q = q < 0 ? 0 : (q >= n_src ? n_src - 1 : q);
        const int slot = atomicAdd(&s_cnt[q], 1);         // arrival order: undone by the ranking below
        s_ent[s_start[q] + slot] = ((e / k) << 6) | (e % k);
    }
    __syncthreads();
    const int share = (E + gridDim.y - 1) / gridDim.y;
    const int ea = blockIdx.y * share, eb = ea + share < E ? ea + share : E;
    for (int e = ea + tid; e < eb; e += RG_THREADS) {
        int q = ib[e];
        q = q < 0 ? 0 : (q >= n_src ? n_src - 1 : q);
        const int v = ((e / k) << 6) | (e % k);
        const int s0 = s_start[q], L = s_cnt[q];
        int rank = 0;
        for (int t = 0; t < L; ++t) rank += s_ent[s0 + t] < v;   // entries are distinct: the rank is the sorted position
        rent[(size_t)b * E + s0 + rank] = v;
    }
}

extern "C" int tgp_reverse_graph(const int32_t *idx, int B, int n_rows, int k, int n_src, int32_t *rptr, int32_t *rent, tgp_stream_t stream)
{
    TGP_REQUIRE(idx && rptr && rent && B > 0 && n_rows > 0 && k > 0 && n_src > 0);
    const size_t lds = ((size_t)n_rows * k + 2 * (size_t)n_src) * sizeof(int);
    if (k > 64 || n_src > RG_MAX_SRC || lds > RG_LDS_MAX || (int64_t)B * n_rows * k >= 0x7fffffff || n_rows >= (1 << 25))
        return TGP_EUNSUPPORTED;
    static TgpLdsAttr attr;
    if (const int e = tgp_lds_attr(attr, reinterpret_cast<const void *>(reverse_graph_kernel), RG_LDS_MAX)) return e;
    hipLaunchKernelGGL(reverse_graph_kernel, dim3(B, B < 64 ? 4 : 1), dim3(RG_THREADS), lds, tgp_hs(stream), idx, B, n_rows, k, n_src, rptr, rent);
    return TGP_LAUNCH_RESULT();
}

// ---------------------------------------------------------------------------------------------------------------------------------
// y[p][c] = max_j src[idx[p][j]][c]  (ORL pooling gcn3d.py:210-217, Pool_layer :225-245)
// pass 1: the winning slot of every (row, channel), four channels per thread, neighbour rows fetched four at a time
__global__ __launch_bounds__(256) void nbrmax_arg_kernel(const float *__restrict__ src, int lds_, const int32_t *__restrict__ idx,
                                                         int B, int n_src, int n_rows, int k, int C, uint8_t *__restrict__ arg,
                                                         int tiles_per_obj)
{
    const int lanes = C >> 2;
    int b, tile;
    if (!tgp_xcd_object_tile(blockIdx.x, B, tiles_per_obj, b, tile)) return;        // an object's workgroups share an XCD: its table stays in that L2
    const int p = tile * (256 / lanes) + threadIdx.x / lanes;
    if (p >= n_rows) return;
    const int64_t row = (int64_t)b * n_rows + p;
    const int c = (threadIdx.x % lanes) * 4;
    const int32_t *nb = idx + row * k;
    const float *sb = src + (int64_t)b * n_src * lds_ + c;
    float4 best = *reinterpret_cast<const float4 *>(sb + (int64_t)nb[0] * lds_);
    int ax = 0, ay = 0, az = 0, aw = 0;
#define NB_TAKE(v, jj)                         \
    if (v.x > best.x) best.x = v.x, ax = jj;   \
    if (v.y > best.y) best.y = v.y, ay = jj;   \
    if (v.z > best.z) best.z = v.z, az = jj;   \
    if (v.w > best.w) best.w = v.w, aw = jj;
    int j = 1;
    for (; j + 3 < k; j += 4) {
        const int q0 = nb[j], q1 = nb[j + 1], q2 = nb[j + 2], q3 = nb[j + 3];
        const float4 v0 = *reinterpret_cast<const float4 *>(sb + (int64_t)q0 * lds_);
        const float4 v1 = *reinterpret_cast<const float4 *>(sb + (int64_t)q1 * lds_);
        const float4 v2 = *reinterpret_cast<const float4 *>(sb + (int64_t)q2 * lds_);
        const float4 v3 = *reinterpret_cast<const float4 *>(sb + (int64_t)q3 * lds_);
        NB_TAKE(v0, j) NB_TAKE(v1, j + 1) NB_TAKE(v2, j + 2) NB_TAKE(v3, j + 3)
    }
    for (; j < k; ++j) {
        const float4 v = *reinterpret_cast<const float4 *>(sb + (int64_t)nb[j] * lds_);
        NB_TAKE(v, j)
    }
#undef NB_TAKE
    *reinterpret_cast<uchar4 *>(arg + row * C + c) = make_uchar4((uint8_t)ax, (uint8_t)ay, (uint8_t)az, (uint8_t)aw);
}

// pass 2: dsrc[q][c] = sum over (p, j) in q's reverse list with arg[p][c] == j of dy[p][c]  (per_object: dy[b][c] * scale)
__global__ __launch_bounds__(256) void nbrmax_gather_kernel(const uint8_t *__restrict__ arg, const int32_t *__restrict__ rptr,
                                                            const int32_t *__restrict__ rent, int B, int n_src, int n_rows, int C,
                                                            const float *__restrict__ dy, int lddy, int per_object, float scale,
                                                            float *__restrict__ dsrc, int ldds, int tiles_per_obj)
{
    const int lanes = C >> 2;
    int b, tile;
    if (!tgp_xcd_object_tile(blockIdx.x, B, tiles_per_obj, b, tile)) return;
    const int qs = tile * (256 / lanes) + threadIdx.x / lanes;
    if (qs >= n_src) return;
    const int64_t q = (int64_t)b * n_src + qs;                                       // b * n_src + source row
    const int c = (threadIdx.x % lanes) * 4;
    const int e0 = rptr[q], e1 = rptr[q + 1];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    if (per_object) {
        g = *reinterpret_cast<const float4 *>(dy + (int64_t)b * lddy + c);
        g.x *= scale, g.y *= scale, g.z *= scale, g.w *= scale;
    }
#pragma unroll 4
    for (int e = e0; e < e1; ++e) {
        const int v = rent[e];
        const int64_t row = (int64_t)b * n_rows + (v >> 6);
        const int j = v & 63;
        const uchar4 a = *reinterpret_cast<const uchar4 *>(arg + row * C + c);
        if (!per_object && (a.x == j || a.y == j || a.z == j || a.w == j)) g = *reinterpret_cast<const float4 *>(dy + row * lddy + c);
        if (a.x == j) acc.x += g.x;
        if (a.y == j) acc.y += g.y;
        if (a.z == j) acc.z += g.z;
        if (a.w == j) acc.w += g.w;
    }
    *reinterpret_cast<float4 *>(dsrc + q * ldds + c) = acc;
}

extern "C" int tgp_nbrmax_bwd_gather(const float *src, int ld_src, const int32_t *idx, const int32_t *rptr, const int32_t *rent, int B,
                                     int n_src, int n_rows, int k, int C, const float *dy, int lddy, int per_object, float scale,
                                     uint8_t *arg_ws, float *dsrc, int ld_dsrc, tgp_stream_t stream)
{
    TGP_REQUIRE(src && idx && rptr && rent && dy && arg_ws && dsrc && B > 0 && n_src > 0 && n_rows > 0 && k > 0 && k <= 64 && C > 0);
    TGP_REQUIRE(ld_src >= C && lddy >= C && ld_dsrc >= C);
    const int lanes = C >> 2;
    if ((C & 3) || lanes > 256 || 256 % lanes || ((ld_src | lddy | ld_dsrc) & 3) ||
        ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dsrc) |
          reinterpret_cast<uintptr_t>(arg_ws)) & 15))
        return TGP_EUNSUPPORTED;
    const int rpw = 256 / lanes;
    const int t1 = tgp_cdiv(n_rows, rpw), t2 = tgp_cdiv(n_src, rpw);
    hipLaunchKernelGGL(nbrmax_arg_kernel, dim3(tgp_xcd_grid(B, t1)), dim3(256), 0, tgp_hs(stream), src, ld_src, idx, B, n_src, n_rows, k, C,
                       arg_ws, t1);
    hipLaunchKernelGGL(nbrmax_gather_kernel, dim3(tgp_xcd_grid(B, t2)), dim3(256), 0, tgp_hs(stream), arg_ws, rptr, rent, B, n_src, n_rows,
                       C, dy, lddy, per_object, scale, dsrc, ld_dsrc, t2);
    return TGP_LAUNCH_RESULT();
}

// ---------------------------------------------------------------------------------------------------------------------------------
// HS_layer.graph_conv (gcn3d.py:157-180):  out = centre + mean_s max_j relu(dir_j . D_s) * support[nbr_j][s]
#define GG_S 7
#define GG_PTS 16
#define GG_MAXK 64
#define GG_THREADS 896          // 7 x 128: one thread per (direction, channel) of a 128-channel chunk of the support block

// pass 1: the forward kernel's structure (csrc/gconv.hip gconv_kernel: one wave per point, four channels per lane, the k gathered
// support rows streamed as 16-byte loads, an object's workgroups on one XCD) with the running maxima carrying their slot.  Writes
// only the winning slot of every (point, direction, channel): one byte.
template <int C>
struct GgLanes {
    static constexpr int LPR = (C / 4) < 64 ? (C / 4) : 64;     // lanes per row
    static constexpr int SPLIT = 64 / LPR;                       // lane groups taking alternate neighbours
    static constexpr int CHUNKS = (C / 4 + 63) / 64;             // 256-column chunks, one workgroup each
};

template <int C>
__global__ __launch_bounds__(256) void gconv_bwd_slot_kernel(const float *__restrict__ xyz, const int32_t *__restrict__ idx,
                                                             const float *__restrict__ proj, int ldp, const float *__restrict__ sdn,
                                                             int B, int n, int k, uint8_t *__restrict__ arg, int tiles_per_obj,
                                                             float *__restrict__ out, int ldo)
{
    using RL = GgLanes<C>;
    int b, tile;
    if (!tgp_xcd_object_tile(blockIdx.x, B, tiles_per_obj, b, tile)) return;
    const int chunk = tile % RL::CHUNKS;
    const int ptile = tile / RL::CHUNKS;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int half = lane / RL::LPR;
    const int cb = chunk * 256 + 4 * (lane % RL::LPR);
    constexpr int SC = GG_S * C;
    float4 sd[GG_S][3];
#pragma unroll
    for (int s = 0; s < GG_S; ++s)
#pragma unroll
        for (int c = 0; c < 3; ++c) sd[s][c] = *reinterpret_cast<const float4 *>(sdn + c * SC + s * C + cb);

    for (int pp = wave; pp < GG_PTS; pp += 4) {
        const int i = ptile * GG_PTS + pp;
        if (i >= n) break;
        const int64_t rowi = (int64_t)b * n + i;
        int nj = 0;
        float dx = 0.f, dy = 0.f, dz = 0.f;
        if (lane < k) {
            nj = idx[rowi * k + lane];
            const float *pn = xyz + ((int64_t)b * n + nj) * 3;
            const float *pc = xyz + rowi * 3;
            dx = pn[0] - pc[0], dy = pn[1] - pc[1], dz = pn[2] - pc[2];
            const float nrm = fmaxf(sqrtf((dx * dx + dy * dy) + dz * dz), 1e-12f);
            dx = dx / nrm, dy = dy / nrm, dz = dz / nrm;
        }
        float4 m[GG_S];
        int ax[GG_S], ay[GG_S], az[GG_S], aw[GG_S];
#pragma unroll
        for (int s = 0; s < GG_S; ++s) m[s] = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY), ax[s] = ay[s] = az[s] = aw[s] = 255;
        // theta as in the forward kernel (fmaf chain, relu), value theta * support, strict '>' in slot order: the first maximum wins
        for (int j = half; j < k; j += RL::SPLIT) {
            const float ux = __shfl(dx, j, 64), uy = __shfl(dy, j, 64), uz = __shfl(dz, j, 64);
            const float *prow = proj + ((int64_t)b * n + __shfl(nj, j, 64)) * ldp + C + cb;
            float4 sup[GG_S];
#pragma unroll
            for (int s = 0; s < GG_S; ++s) sup[s] = *reinterpret_cast<const float4 *>(prow + s * C);
#pragma unroll
            for (int s = 0; s < GG_S; ++s) {
                float tx = fmaf(uz, sd[s][2].x, fmaf(uy, sd[s][1].x, ux * sd[s][0].x));
                float ty = fmaf(uz, sd[s][2].y, fmaf(uy, sd[s][1].y, ux * sd[s][0].y));
                float tz = fmaf(uz, sd[s][2].z, fmaf(uy, sd[s][1].z, ux * sd[s][0].z));
                float tw = fmaf(uz, sd[s][2].w, fmaf(uy, sd[s][1].w, ux * sd[s][0].w));
                tx = fmaxf(tx, 0.f) * sup[s].x, ty = fmaxf(ty, 0.f) * sup[s].y, tz = fmaxf(tz, 0.f) * sup[s].z, tw = fmaxf(tw, 0.f) * sup[s].w;
                if (tx > m[s].x) m[s].x = tx, ax[s] = j;
                if (ty > m[s].y) m[s].y = ty, ay[s] = j;
                if (tz > m[s].z) m[s].z = tz, az[s] = j;
                if (tw > m[s].w) m[s].w = tw, aw[s] = j;
            }
        }
        if (RL::SPLIT == 2) {       // the two lane groups took alternate slots: the larger value wins, on a tie the smaller slot
#pragma unroll
            for (int s = 0; s < GG_S; ++s) {
#define GG_MERGE(mm, aa)                                                          \
    {                                                                             \
        const float mo = __shfl_xor(mm, 32, 64);                                  \
        const int ao = __shfl_xor(aa, 32, 64);                                    \
        if (mo > mm || (mo == mm && ao < aa)) mm = mo, aa = ao;                   \
    }
                GG_MERGE(m[s].x, ax[s]) GG_MERGE(m[s].y, ay[s]) GG_MERGE(m[s].z, az[s]) GG_MERGE(m[s].w, aw[s])
#undef GG_MERGE
            }
        }
        if (half == 0) {
#pragma unroll
            for (int s = 0; s < GG_S; ++s)
                *reinterpret_cast<uchar4 *>(arg + rowi * SC + s * C + cb) = make_uchar4((uint8_t)ax[s], (uint8_t)ay[s], (uint8_t)az[s], (uint8_t)aw[s]);
            if (out) {             // the layer's output as well (training forward: this kernel then replaces gconv_kernel) -- the same
                float4 acc = m[0]; // maxima, the same sequential mean over the 7 supports and the same centre add, bit for bit
#pragma unroll
                for (int s = 1; s < GG_S; ++s) acc.x += m[s].x, acc.y += m[s].y, acc.z += m[s].z, acc.w += m[s].w;
                acc.x = acc.x / 7.0f, acc.y = acc.y / 7.0f, acc.z = acc.z / 7.0f, acc.w = acc.w / 7.0f;
                const float4 ctr = *reinterpret_cast<const float4 *>(proj + rowi * ldp + cb);
                acc.x = ctr.x + acc.x, acc.y = ctr.y + acc.y, acc.z = ctr.z + acc.z, acc.w = ctr.w + acc.w;
                *reinterpret_cast<float4 *>(out + rowi * ldo + cb) = acc;
            }
        }
    }
}

// pass 1b: one thread per element e = s * C + c (896 threads = one 128-channel chunk of the support block), GG_PTS points per
// workgroup.  From the stored slot: theta* again (same expression), the winner's support value (ONE gathered float instead of k),
// the value that travels to the winner's support row, g / 7 * theta* -> contrib (slot rewritten to 255 where nothing travels), and
// the direction gradients d D += g / 7 * support* * dir* (theta* > 0), one partial per workgroup.
__global__ __launch_bounds__(GG_THREADS) void gconv_bwd_value_kernel(const float *__restrict__ xyz, const int32_t *__restrict__ idx,
                                                                     const float *__restrict__ proj, int ldp, const float *__restrict__ sdn,
                                                                     const float *__restrict__ dg, int ldg, int B, int n, int k, int C,
                                                                     uint8_t *__restrict__ arg, float *__restrict__ contrib,
                                                                     float *__restrict__ dsdn_partial)
{
    const int SC = GG_S * C;
    const int tiles_per_obj = (n + GG_PTS - 1) / GG_PTS;
    const int chunks = SC / GG_THREADS;
    int b, t;
    if (!tgp_xcd_object_tile(blockIdx.x, B, tiles_per_obj * chunks, b, t)) return;   // an object's workgroups share an XCD (its proj table: one L2)
    const int tile = t / chunks;
    const int part = b * tiles_per_obj + tile;
    const int e = (t % chunks) * GG_THREADS + threadIdx.x;   // s * C + c
    const int c = e % C;
    const float D0 = sdn[e], D1 = sdn[SC + e], D2 = sdn[2 * SC + e];
    float dD0 = 0.f, dD1 = 0.f, dD2 = 0.f;
    __shared__ float4 s_nd[GG_PTS][GG_MAXK];                 // (unit direction, neighbour id)
    const int base = tile * GG_PTS;
    for (int u = threadIdx.x; u < GG_PTS * k; u += GG_THREADS) {
        const int lp = u / k, j = u - lp * k;
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
        s_nd[lp][j] = make_float4(ux, uy, uz, __int_as_float(nb));
    }
    __syncthreads();
    const float *pb = proj + (int64_t)b * n * ldp + C + e;
    // in rounds of eight points, each round three batches of independent loads (slots | gradient, then the winners' support values):
    // point by point the chain slot -> LDS -> gathered support value -> store ran sixteen times in a row per thread
    for (int lp0 = 0; lp0 < GG_PTS; lp0 += 8) {
        int bj[8];
        float g[8], sup[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = base + lp0 + u;
            const int64_t rowi = (int64_t)b * n + (i < n ? i : n - 1);
            bj[u] = i < n ? arg[rowi * SC + e] : 255;
            g[u] = dg[rowi * ldg + c] / 7.0f;
        }
        float4 nd[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            nd[u] = s_nd[lp0 + u][bj[u] != 255 ? bj[u] : 0];
            sup[u] = pb[(int64_t)__float_as_int(nd[u].w) * ldp];     // (slot 0 of an inactive point: a valid row, value unused)
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = base + lp0 + u;
            if (i >= n) continue;
            const int64_t rowi = (int64_t)b * n + i;
            float send = 0.f;
            bool travels = false;
            if (bj[u] != 255) {
                float th = fmaf(nd[u].z, D2, fmaf(nd[u].y, D1, nd[u].x * D0));   // same expression as the forward kernel
                th = fmaxf(th, 0.f);
                send = g[u] * th;
                travels = th != 0.f;
                if (th > 0.f) {
                    const float dth = g[u] * sup[u];
                    dD0 += dth * nd[u].x, dD1 += dth * nd[u].y, dD2 += dth * nd[u].z;
                }
            }
            if (!travels && bj[u] != 255) arg[rowi * SC + e] = 255;
            contrib[rowi * SC + e] = send;
        }
    }
    float *o = dsdn_partial + (int64_t)part * 3 * SC;
    o[e] = dD0, o[SC + e] = dD1, o[2 * SC + e] = dD2;
}

// pass 2: dproj[q][C + e] = sum over (p, j) in q's reverse list with arg[p][e] == j of contrib[p][e];  dproj[q][c] = dg[q][c].
// Four consecutive elements per thread (one 4-byte load of slots per entry, a 16-byte load of values on a match): 224 threads
// cover a 896-element chunk.  (Measured and dropped: a workgroup per EIGHT consecutive source rows, their lists walked as one loop with
// eight slot loads in flight across row boundaries -- 200 us per launch against 150: an eighth of the workgroups, each a long
// serial walk; the one-row form's short chains overlap across the 8 workgroups a CU holds.  Also measured: one WAVE per row with 16
// elements per lane (32 rows per CU in flight: 150 us, no change) and four independent waves per workgroup walking four rows each
// (250 us).  Neither the walks in flight nor the workgroup count is what bounds this pass.)
#define GG_GTHREADS (GG_THREADS / 4)
__global__ __launch_bounds__(GG_GTHREADS) void gconv_bwd_gather_kernel(const uint8_t *__restrict__ arg, const float *__restrict__ contrib,
                                                                       const int32_t *__restrict__ rptr, const int32_t *__restrict__ rent,
                                                                       const float *__restrict__ dg, int ldg, int B, int n, int C,
                                                                       float *__restrict__ dproj, int lddp)
{
    const int SC = GG_S * C;
    const int chunks = SC / GG_THREADS;
    int b, t;
    if (!tgp_xcd_object_tile(blockIdx.x, B, n * chunks, b, t)) return;
    const int64_t q = (int64_t)b * n + t / chunks;           // b * n + source row
    const int ychunk = t % chunks;
    const int e = ychunk * GG_THREADS + threadIdx.x * 4;
    const int e0 = rptr[q], e1 = rptr[q + 1];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    // the list goes through LDS in slabs (one load per lane instead of a scalar load in front of every entry's slot load)
    __shared__ int s_ent[GG_GTHREADS];
    for (int l0 = e0; l0 < e1; l0 += GG_GTHREADS) {
    const int ln = e1 - l0 < GG_GTHREADS ? e1 - l0 : GG_GTHREADS;
    __syncthreads();
    if ((int)threadIdx.x < ln) s_ent[threadIdx.x] = rent[l0 + threadIdx.x];
    __syncthreads();
#pragma unroll 8
    for (int t2 = 0; t2 < ln; ++t2) {
        const int v = s_ent[t2];
        const int64_t off = ((int64_t)b * n + (v >> 6)) * SC + e;
        const uint8_t j = (uint8_t)(v & 63);
        const uchar4 a = *reinterpret_cast<const uchar4 *>(arg + off);
        if (a.x == j || a.y == j || a.z == j || a.w == j) {
            const float4 cv = *reinterpret_cast<const float4 *>(contrib + off);
            if (a.x == j) acc.x += cv.x;
            if (a.y == j) acc.y += cv.y;
            if (a.z == j) acc.z += cv.z;
            if (a.w == j) acc.w += cv.w;
        }
    }
    }
    *reinterpret_cast<float4 *>(dproj + q * lddp + C + e) = acc;
    if (ychunk == 0)
        for (int c = threadIdx.x * 4; c < C; c += GG_GTHREADS * 4)
            *reinterpret_cast<float4 *>(dproj + q * lddp + c) = *reinterpret_cast<const float4 *>(dg + q * ldg + c);
}

__global__ void gg_partial_sum_kernel(const float *__restrict__ partial, int64_t parts, int64_t width, int64_t group_size,
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

static int gg_supported(const void *proj, const void *sdn, int ldp, int k, int C)
{
    return (C == 128 || C == 256 || C == 512) && k <= GG_MAXK - 1 && (ldp & 3) == 0 && (reinterpret_cast<uintptr_t>(proj) & 15) == 0 &&
           (reinterpret_cast<uintptr_t>(sdn) & 15) == 0;
}

static void gg_launch_slots(const float *xyz, const int32_t *idx, const float *proj, int ldp, const float *sdn, int B, int n, int k, int C,
                            uint8_t *arg, float *out, int ldo, hipStream_t stream)
{
    const int ptiles = tgp_cdiv(n, GG_PTS);
#define GG_GO(CC)                                                                                                                       \
    {                                                                                                                                   \
        const int tiles = ptiles * GgLanes<CC>::CHUNKS;                                                                                 \
        hipLaunchKernelGGL((gconv_bwd_slot_kernel<CC>), dim3(tgp_xcd_grid(B, tiles)), dim3(256), 0, stream, xyz, idx, proj, ldp, sdn, B, \
                           n, k, arg, tiles, out, ldo);                                                                                 \
    }
    if (C == 128) GG_GO(128) else if (C == 256) GG_GO(256) else GG_GO(512)
#undef GG_GO
}

// HS_layer.graph_conv forward that also records the winning slot of every (point, direction, channel) for the scatter-free
// backward (slots: B*n*7C bytes): same output as tgp_gconv_hs_fwd, bit for bit.
extern "C" int tgp_gconv_hs_fwd_slots(const float *xyz, const int32_t *idx, const float *proj, int ldp, const float *sdn, int B, int n,
                                      int k, int S, int C, float *out, int ldo, uint8_t *slots, tgp_stream_t stream)
{
    TGP_REQUIRE(xyz && idx && proj && sdn && out && slots && B > 0 && n > 0 && k > 0 && S == GG_S && ldp >= 8 * C && ldo >= C);
    if (!gg_supported(proj, sdn, ldp, k, C) || (ldo & 3) || ((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(slots)) & 15))
        return TGP_EUNSUPPORTED;
    gg_launch_slots(xyz, idx, proj, ldp, sdn, B, n, k, C, slots, out, ldo, tgp_hs(stream));
    return TGP_LAUNCH_RESULT();
}

// workspace (floats): partials of d D, as tgp_gconv_bwd_workspace_floats; arg_ws: B*n*7C bytes; contrib_ws: B*n*7C floats.
// have_slots != 0: arg_ws already holds the slots recorded by tgp_gconv_hs_fwd_slots (the slot pass is skipped).
extern "C" int tgp_gconv_hs_bwd_gather(const float *xyz, const int32_t *idx, const int32_t *rptr, const int32_t *rent, const float *proj,
                                       int ldp, const float *sdn, const float *dg, int ldg, int B, int n, int k, int S, int C,
                                       float *dproj, int lddp, float *dsdn, float *workspace, uint8_t *arg_ws, float *contrib_ws,
                                       int have_slots, tgp_stream_t stream)
{
    TGP_REQUIRE(xyz && idx && rptr && rent && proj && sdn && dg && dproj && dsdn && workspace && arg_ws && contrib_ws);
    TGP_REQUIRE(B > 0 && n > 0 && k > 0 && S == GG_S && C > 0 && ldg >= C && ldp >= 8 * C && lddp >= 8 * C);
    if (!gg_supported(proj, sdn, ldp, k, C) || ((ldg | lddp) & 3) ||
        ((reinterpret_cast<uintptr_t>(dg) | reinterpret_cast<uintptr_t>(dproj) | reinterpret_cast<uintptr_t>(arg_ws) |
          reinterpret_cast<uintptr_t>(contrib_ws)) & 15))
        return TGP_EUNSUPPORTED;
    const int64_t parts = (int64_t)B * tgp_cdiv(n, GG_PTS);
    const int chunks = GG_S * C / GG_THREADS;
    const int ptiles = tgp_cdiv(n, GG_PTS);
    if (!have_slots) gg_launch_slots(xyz, idx, proj, ldp, sdn, B, n, k, C, arg_ws, nullptr, 0, tgp_hs(stream));
    const dim3 grid(tgp_xcd_grid(B, ptiles * chunks));
    hipLaunchKernelGGL(gconv_bwd_value_kernel, grid, dim3(GG_THREADS), 0, tgp_hs(stream), xyz, idx, proj, ldp, sdn, dg, ldg, B, n, k, C,
                       arg_ws, contrib_ws, workspace);
    const int64_t width = 3 * (int64_t)GG_S * C;
    const int groups = tgp_cdiv(parts, (int64_t)32);
    float *level2 = workspace + parts * width;
    hipLaunchKernelGGL(gg_partial_sum_kernel, dim3(tgp_cdiv(width, (int64_t)256), groups), dim3(256), 0, tgp_hs(stream), workspace, parts,
                       width, (int64_t)32, level2);
    hipLaunchKernelGGL(gg_partial_sum_kernel, dim3(tgp_cdiv(width, (int64_t)256), 1), dim3(256), 0, tgp_hs(stream), level2, (int64_t)groups,
                       width, (int64_t)groups, dsdn);
    hipLaunchKernelGGL(gconv_bwd_gather_kernel, dim3(tgp_xcd_grid(B, n * chunks)), dim3(GG_GTHREADS), 0, tgp_hs(stream), arg_ws, contrib_ws,
                       rptr, rent, dg, ldg, B, n, C, dproj, lddp);
    return TGP_LAUNCH_RESULT();
}
