// Input side of the evaluation loader (evaluation/load_data_eval.py:302-357, 404-417, 451-462): depth image + detection
// mask + box window -> the camera-frame cloud the network is fed.  Byte / integer work: per detection 65536 ROI pixels x
// (2 B depth + 1 B mask) in, 12 KB out.  One 1024-thread workgroup per detection walks the ROI in row-major rounds of 4096
// pixels and compacts in order (wave ballots + a 4 x 16-entry LDS table, one barrier per round), because the reference's
// boolean indexing keeps row-major order and its outlier cut is anchored on point number 25 of that order.
//
// What lives in HBM between the passes is a 4-byte RECORD per valid pixel, (ROI pixel index << 16) | depth, not the 12-byte
// point: a point is a pure function of its record and the detection's window / intrinsics, so pass 2 and the resampling
// kernels recompute it (25 FMAs) instead of moving it.  The first version wrote and re-read points (319 KB fetched +
// 580 KB written per detection against 209 KB algorithmic, profiles/r01_k_input_side_sq_counters.txt); records cut the
// scratch to a third (262 KB per detection, L2-resident) and only the n_pts sampled points are ever materialised.
//
// The ROI resampling is cv2.warpAffine(..., INTER_NEAREST) of a pure scale + shift (tools/dataset_utils.py:80-136 with
// rot = 0): OpenCV inverts the matrix in double and walks it in 10-bit fixed point,
//     X = (cvRound((M1*y + M2)*1024) + 512 + cvRound(M0*x*1024)) >> 10.
// For a roi_size that divides 1024 every product is an integer in exact arithmetic (M0 = s/roi_size, M2 = cx - s/2 with
// cx a multiple of 1/2), far from a rounding tie, so the double-precision noise of OpenCV's inversion cannot change the
// result and the source pixel is the integer expression in src_coord() below.
#include <stdlib.h>

#include "tgp_common.h"

#define ROI_THREADS 1024
#define ROI_WAVES (ROI_THREADS / TGP_WAVE)

__device__ __forceinline__ int src_coord(int sum_lo_hi, int s, int x, int step)
{
    // 1024*centre - 512*s + 512 + x*s*(1024/roi_size), arithmetic shift = floor
    return (512 * sum_lo_hi - 512 * s + 512 + x * s * step) >> 10;
}

// ROI_SUB = sub-rounds per barrier: a round covers ROI_SUB * 1024 consecutive pixels, element (k, tid) = k*1024 + tid.
// Ordered compaction of one round: slot[k] = base + rank of element (k, tid) among the round's keepers in element order;
// adds the round's total to base.  ONE barrier per round (tot is double buffered by the caller's round parity).
template <int ROI_SUB>
__device__ __forceinline__ void ordered_slots(const bool (&keep)[ROI_SUB], int (*tot)[ROI_SUB][ROI_WAVES], int parity, int &base,
                                              int (&slot)[ROI_SUB])
{
    const int lane = threadIdx.x & (TGP_WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / TGP_WAVE);
    unsigned long long bal[ROI_SUB];
#pragma unroll
    for (int k = 0; k < ROI_SUB; ++k) {
        bal[k] = __ballot(keep[k]);
        if (lane == 0) tot[parity][k][wave] = __popcll(bal[k]);
    }
    __syncthreads();
    // every wave scans the round's ROI_SUB * 16 wave totals across its own lanes (lane l holds entry l in element order):
    // a dozen instructions per round instead of a 16-step loop per pixel
    const int own = lane < ROI_SUB * ROI_WAVES ? (&tot[parity][0][0])[lane] : 0;
    int incl = own;
#pragma unroll
    for (int o = 1; o < ROI_SUB * ROI_WAVES; o <<= 1) {
        const int up = __shfl_up(incl, o);
        incl += lane >= o ? up : 0;
    }
    const int excl = incl - own;
#pragma unroll
    for (int k = 0; k < ROI_SUB; ++k)
        slot[k] = base + __builtin_amdgcn_readlane(excl, k * ROI_WAVES + wave) + __popcll(bal[k] & ((1ull << lane) - 1ull));
    base += __builtin_amdgcn_readlane(incl, ROI_SUB * ROI_WAVES - 1);
}

__device__ __forceinline__ float wave_min(float v)
{
#pragma unroll
    for (int o = TGP_WAVE / 2; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int o = TGP_WAVE / 2; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// x / b for a divisor that is uniform over the launch: the refined reciprocal of the compiler's IEEE division sequence
// (v_rcp_f32 + one Newton step) is computed once, and each quotient is that sequence's remaining five operations --
// q = a*r; q += fma(-b,q,a)*r; q += fma(-b,q,a)*r -- i.e. exactly what `a / b` compiles to when v_div_scale / v_div_fixup
// are no-ops (operands far from overflow / denormals: pixel coordinates x millimetres over focal lengths and 1000).
// Correctly rounded there; the parity tests compare bit patterns with numpy's division over ~10^6 quotients.
struct UniformDiv {
    float b, r;
    __device__ __forceinline__ explicit UniformDiv(float divisor) : b(divisor)
    {
        const float r0 = __builtin_amdgcn_rcpf(divisor);
        r = fmaf(fmaf(-divisor, r0, 1.0f), r0, r0);
    }
    __device__ __forceinline__ float operator()(float a) const
    {
        float q = a * r;
        q = fmaf(fmaf(-b, q, a), r, q);
        return fmaf(fmaf(-b, q, a), r, q);
    }
};

// A detection's geometry: everything a record needs to become a point (load_data_eval.py:451-462 then /1000.0 at :338,
// float32 step by step, correctly rounded quotients).
//
// tab (training loader, datasets/load_data.py:235-250): the window comes from the box augmentation aug_bbox_DZI
// (tools/dataset_utils.py:24-61) -- a real-valued centre and scale -- so OpenCV's fixed-point walk has no integer closed form.
// The host evaluates it once per detection in double, as OpenCV does, into tab[0 .. roi) = source column of ROI column x and
// tab[roi .. 2 roi) = source row of ROI row y (rot = 0: the map is separable); the kernels then look the source pixel up.
struct RoiGeom {
    int sumc, sumr, s, step, roi_log2;
    float cx, cy;
    UniformDiv div_fx, div_fy, div_k;
    const int *tab;
    __device__ __forceinline__ RoiGeom(const int *__restrict__ window, const float *__restrict__ camk, int j, int img, int lg,
                                       const int *__restrict__ tables = nullptr)
        : sumc(tables ? 0 : window[j * 3]), sumr(tables ? 0 : window[j * 3 + 1]), s(tables ? 0 : window[j * 3 + 2]), step(1024 >> lg),
          roi_log2(lg), cx(camk[img * 4 + 2]), cy(camk[img * 4 + 3]), div_fx(camk[img * 4]), div_fy(camk[img * 4 + 1]), div_k(1000.0f),
          tab(tables ? tables + ((size_t)j << (lg + 1)) : nullptr)
    {
    }
    __device__ __forceinline__ int src_x(int x) const { return tab ? tab[x] : src_coord(sumc, s, x, step); }
    __device__ __forceinline__ int src_y(int y) const { return tab ? tab[(1 << roi_log2) + y] : src_coord(sumr, s, y, step); }
    __device__ __forceinline__ void point(uint32_t rec, float &px, float &py, float &pz) const
    {
        const int p = (int)(rec >> 16), x = p & ((1 << roi_log2) - 1), y = p >> roi_log2;
        const float dep = (float)(rec & 0xffffu);
        px = div_k(div_fx(((float)src_x(x) - cx) * dep));
        py = div_k(div_fy(((float)src_y(y) - cy) * dep));
        pz = div_k(dep);
    }
};

template <int ROI_SUB>
__global__ __launch_bounds__(ROI_THREADS) void roi_cloud_kernel(const uint16_t *__restrict__ depth, const uint8_t *__restrict__ masks,
                                                                const int64_t *__restrict__ mask_off, const int *__restrict__ mask_stride,
                                                                const int *__restrict__ det_img, const int *__restrict__ window,
                                                                const float *__restrict__ camk, int H, int W, int roi_log2,
                                                                uint32_t *recs, int *__restrict__ counts, const int *__restrict__ tables,
                                                                const int *__restrict__ mask_val, float cut_frac)
{
    __shared__ int tot[2][ROI_SUB][ROI_WAVES];
    __shared__ float red[6][ROI_WAVES];
    __shared__ int n_depth_waves[ROI_WAVES];
    const int j = blockIdx.x, tid = threadIdx.x;
    const int roi = 1 << roi_log2;
    const int cap = roi * roi;
    const int img = det_img[j];
    const RoiGeom g(window, camk, j, img, roi_log2, tables);
    const int mval = mask_val ? mask_val[j] : 0;       // 0: any non-zero mask byte (a detection's own channel); v > 0: the byte equals v
    const uint16_t *dimg = depth + (size_t)img * H * W;
    const uint8_t *mimg = masks + mask_off[j];
    const int mstride = mask_stride[j];
    uint32_t *rec = recs + (size_t)j * cap;

    // ---- pass 1: ROI pixel -> source pixel -> (depth > 0) & mask -> record, kept in ROI order.
    // The extent of the cloud (:341-344) is tracked on the dividends: p = div_k(div_f(t)) is monotone in t = (coord - c) * depth
    // (rounding is monotone), so min / max over the points = the map of min / max over t -- no division per pixel here.
    float lo0 = INFINITY, lo1 = INFINITY, lo2 = INFINITY, hi0 = -INFINITY, hi1 = -INFINITY, hi2 = -INFINITY;
    int base = 0, n_depth = 0;
    const int rounds = cap / (ROI_THREADS * ROI_SUB);
    // 1024 threads are a whole number of ROI rows (or a fraction of one): a thread keeps its column for the whole walk and a
    // wave sits inside one row, so everything that depends on x alone is hoisted and the row terms are wave-uniform
    const int x = tid & (roi - 1);
    const int sx = g.src_x(x);
    const bool inbx = sx >= 0 && sx < W;
    const int colq = min(max(sx, 0), W - 1);
    const float xm = (float)sx - g.cx;
    const int row0 = __builtin_amdgcn_readfirstlane(tid >> roi_log2);      // this wave's row within a 1024-pixel sub-round
    // The walk is a chain of dependent rounds (loads -> ballot -> barrier -> scan -> stores) on one CU, so what bounds it is
    // latency, not bandwidth or issue (SQ counters: waves wait 69 % of their cycles).  Round c+1's loads are therefore issued
    // before round c's barrier and stores: they do not depend on round c, and being older than the stores they can be waited
    // for without waiting for the stores (vmcnt retires in order).
    int d_next[ROI_SUB], m_next[ROI_SUB];
    float ym_next[ROI_SUB];
    auto fetch = [&](int c) {
#pragma unroll
        for (int k = 0; k < ROI_SUB; ++k) {
            const int y = ((c * ROI_SUB + k) * ROI_THREADS >> roi_log2) + row0;        // = (round element index) / roi
            const int sy = g.src_y(y);
            const bool inb = inbx && sy >= 0 && sy < H;
            // border pixels read a clamped address and are zeroed afterwards: no branch round the loads, all 2*ROI_SUB in flight
            const int q = min(max(sy, 0), H - 1) * W + colq;                            // < 2^24 (host-checked)
            ym_next[k] = (float)sy - g.cy;
            d_next[k] = inb ? (int)dimg[q] : 0;
            m_next[k] = inb ? (int)mimg[q * mstride] : 0;
        }
    };
    fetch(0);
    for (int c = 0; c < rounds; ++c) {
        int d[ROI_SUB], m[ROI_SUB];
        float ym[ROI_SUB];
        bool keep[ROI_SUB];
        int slot[ROI_SUB];
#pragma unroll
        for (int k = 0; k < ROI_SUB; ++k) d[k] = d_next[k], m[k] = m_next[k], ym[k] = ym_next[k];
        if (c + 1 < rounds) fetch(c + 1);
#pragma unroll
        for (int k = 0; k < ROI_SUB; ++k) {
            n_depth += d[k] > 0;
            keep[k] = d[k] > 0 && (mval ? m[k] == mval : m[k] != 0);
        }
        ordered_slots(keep, tot, c & 1, base, slot);
#pragma unroll
        for (int k = 0; k < ROI_SUB; ++k)
            if (keep[k]) {
                const int p = (c * ROI_SUB + k) * ROI_THREADS + tid;
                rec[slot[k]] = ((uint32_t)p << 16) | (uint32_t)d[k];
                const float dep = (float)d[k], tx = xm * dep, ty = ym[k] * dep;
                lo0 = fminf(lo0, tx), lo1 = fminf(lo1, ty), lo2 = fminf(lo2, dep);
                hi0 = fmaxf(hi0, tx), hi1 = fmaxf(hi1, ty), hi2 = fmaxf(hi2, dep);
            }
    }
    const int n_valid = base;

    // ---- extent of the cloud and the count of depth-valid ROI pixels
    {
        const int lane = tid & (TGP_WAVE - 1), wave = tid / TGP_WAVE;
        lo0 = wave_min(lo0), lo1 = wave_min(lo1), lo2 = wave_min(lo2);
        hi0 = wave_max(hi0), hi1 = wave_max(hi1), hi2 = wave_max(hi2);
        int nd = n_depth;
#pragma unroll
        for (int o = TGP_WAVE / 2; o > 0; o >>= 1) nd += __shfl_xor(nd, o);
        if (lane == 0) {
            red[0][wave] = lo0, red[1][wave] = lo1, red[2][wave] = lo2;
            red[3][wave] = hi0, red[4][wave] = hi1, red[5][wave] = hi2;
            n_depth_waves[wave] = nd;
        }
    }
    __syncthreads();        // also orders pass 1's global stores before pass 2's loads (same workgroup, same CU)
    int nd_all = 0;
#pragma unroll
    for (int w = 0; w < ROI_WAVES; ++w) {
        lo0 = fminf(lo0, red[0][w]), lo1 = fminf(lo1, red[1][w]), lo2 = fminf(lo2, red[2][w]);
        hi0 = fmaxf(hi0, red[3][w]), hi1 = fmaxf(hi1, red[4][w]), hi2 = fmaxf(hi2, red[5][w]);
        nd_all += n_depth_waves[w];
    }
    if (tid == 0) counts[j * 3] = nd_all, counts[j * 3 + 1] = n_valid;
    if (n_valid < 26) {     // the reference indexes point 25 (:350) and raises; the host mirror raises for -1
        if (tid == 0) counts[j * 3 + 2] = -1;
        return;
    }

    // ---- pass 2: drop the points within a quarter of the extent's diagonal of point 25 (:345-355), compacting the records in place
    float r0, r1, r2;
    {
        // the quotient maps are monotone, increasing or decreasing with the sign of the focal length: take both ends
        const float a0 = g.div_k(g.div_fx(lo0)), b0 = g.div_k(g.div_fx(hi0)), a1 = g.div_k(g.div_fy(lo1)), b1 = g.div_k(g.div_fy(hi1));
        r0 = fmaxf(a0, b0) - fminf(a0, b0), r1 = fmaxf(a1, b1) - fminf(a1, b1), r2 = g.div_k(hi2) - g.div_k(lo2);
    }
    const float thr = __fsqrt_rn((r0 * r0 + r1 * r1) + r2 * r2) * cut_frac;     // float32 product, as numpy's (0.25 eval, 0.15 training)
    // sqrt_rn is monotone, so sqrt_rn(v) > thr  <=>  v > v_max, v_max = the largest float whose rounded root is <= thr:
    // found once per detection from thr*thr by stepping ulps; the per-point test is then a compare of the squared distance
    float v_max = thr * thr;
    for (int it = 0; it < 8 && __fsqrt_rn(v_max) > thr; ++it) v_max = __uint_as_float(__float_as_uint(v_max) - 1u);
    for (int it = 0; it < 8; ++it) {
        const float up = __uint_as_float(__float_as_uint(v_max) + 1u);
        if (!(__fsqrt_rn(up) <= thr)) break;
        v_max = up;
    }
    // (thr = 0, a cloud of one repeated point: v_max stays 0 and v > 0 is the same test; thr is never negative or NaN here)
    float c0, c1, c2;
    g.point(rec[25], c0, c1, c2);
    __syncthreads();        // everyone holds point 25 before round 0 may overwrite it
    base = 0;
    const int rounds2 = (n_valid + ROI_THREADS * ROI_SUB - 1) / (ROI_THREADS * ROI_SUB);
    // same pipelining: round c+1 reads records >= (c+1)*4096, round c writes records < (c+1)*4096
    uint32_t nrec[ROI_SUB];
    auto fetch2 = [&](int c) {
#pragma unroll
        for (int k = 0; k < ROI_SUB; ++k) nrec[k] = rec[min((c * ROI_SUB + k) * ROI_THREADS + tid, n_valid - 1)];
    };
    fetch2(0);
    for (int c = 0; c < rounds2; ++c) {
        uint32_t cur[ROI_SUB];
        bool keep[ROI_SUB];
        int slot[ROI_SUB];
#pragma unroll
        for (int k = 0; k < ROI_SUB; ++k) cur[k] = nrec[k];
        if (c + 1 < rounds2) fetch2(c + 1);
#pragma unroll
        for (int k = 0; k < ROI_SUB; ++k) {
            const int i = (c * ROI_SUB + k) * ROI_THREADS + tid;
            float px, py, pz;
            g.point(cur[k], px, py, pz);
            const float d0 = px - c0, d1 = py - c1, d2 = pz - c2;
            // numpy: sqrt(add.reduce(x*x)) > thr with the squares summed left to right
            keep[k] = i < n_valid && (d0 * d0 + d1 * d1) + d2 * d2 > v_max;
        }
        // slot <= i, and every load of records < (c+1)*4096 precedes the barrier inside ordered_slots: compacting in place is safe
        ordered_slots(keep, tot, c & 1, base, slot);
#pragma unroll
        for (int k = 0; k < ROI_SUB; ++k)
            if (keep[k]) rec[slot[k]] = cur[k];
    }
    if (tid == 0) counts[j * 3 + 2] = base;
}

static int roi_log2_of(int roi_size)
{
    int lg = 0;
    while ((1 << lg) < roi_size) ++lg;
    // whole rounds of 4096 pixels; a pixel index must fit the record's 16 bits
    return ((1 << lg) == roi_size && roi_size >= 64 && roi_size <= 256) ? lg : -1;
}

extern "C" int tgp_roi_cloud_ex(const uint16_t *depth, const uint8_t *masks, const int64_t *mask_off, const int *mask_stride,
                                const int *det_img, const int *window, const float *camk, int D, int H, int W, int roi_size, uint32_t *recs,
                                int *counts, const int *tables, const int *mask_val, float cut_frac, tgp_stream_t stream)
{
    TGP_REQUIRE(depth && masks && mask_off && mask_stride && det_img && (window || tables) && camk && recs && counts);
    TGP_REQUIRE(cut_frac >= 0.f && cut_frac <= 1.f);
    TGP_REQUIRE(D > 0 && H > 0 && W > 0 && H < 32768 && W < 32768 && (int64_t)H * W < (1ll << 24));   // x mask stride < 128: 31-bit offsets
    const int lg = roi_log2_of(roi_size);
    if (lg < 0) return TGP_EUNSUPPORTED;
    static const int sub = [] { const char *e = getenv("TGP_ROI_SUB"); return e ? atoi(e) : 4; }();     // development A/B
    auto kern = sub == 1 ? roi_cloud_kernel<1> : sub == 2 ? roi_cloud_kernel<2> : roi_cloud_kernel<4>;
    hipLaunchKernelGGL(kern, dim3(D), dim3(ROI_THREADS), 0, tgp_hs(stream), depth, masks, mask_off, mask_stride, det_img, window, camk, H, W,
                       lg, recs, counts, tables, mask_val, cut_frac);
    return TGP_LAUNCH_RESULT();
}

extern "C" int tgp_roi_cloud(const uint16_t *depth, const uint8_t *masks, const int64_t *mask_off, const int *mask_stride,
                             const int *det_img, const int *window, const float *camk, int D, int H, int W, int roi_size, uint32_t *recs,
                             int *counts, tgp_stream_t stream)
{
    TGP_REQUIRE(window);
    return tgp_roi_cloud_ex(depth, masks, mask_off, mask_stride, det_img, window, camk, D, H, W, roi_size, recs, counts, nullptr, nullptr,
                            0.25f, stream);
}

// _sample_points (:404-417) as a gather that materialises the selected points: the host supplies the selection (tiled indices,
// or the prefix of the permutation it drew from np.random in the reference's order), out[j, i] = point(recs[j, sel[j, i]]).
// An index outside [0, roi_size^2) yields NaNs.
__global__ void cloud_select_kernel(const uint32_t *__restrict__ recs, const int *__restrict__ sel, const int *__restrict__ det_img,
                                    const int *__restrict__ window, const float *__restrict__ camk, int64_t total, int roi_log2, int n_pts,
                                    float *__restrict__ out, const int *__restrict__ tables)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int j = (int)(t / n_pts), cap = 1 << (2 * roi_log2);
    const int k = sel[t];
    float a = NAN, b = NAN, c = NAN;
    if (k >= 0 && k < cap) RoiGeom(window, camk, j, det_img[j], roi_log2, tables).point(recs[(size_t)j * cap + k], a, b, c);
    out[t * 3] = a, out[t * 3 + 1] = b, out[t * 3 + 2] = c;
}

extern "C" int tgp_cloud_select_ex(const uint32_t *recs, const int32_t *sel, const int *det_img, const int *window, const float *camk, int D,
                                   int roi_size, int n_pts, float *out, const int *tables, tgp_stream_t stream)
{
    TGP_REQUIRE(recs && sel && det_img && (window || tables) && camk && out && D > 0 && n_pts > 0);
    const int lg = roi_log2_of(roi_size);
    if (lg < 0) return TGP_EUNSUPPORTED;
    const int64_t total = (int64_t)D * n_pts;
    hipLaunchKernelGGL(cloud_select_kernel, dim3(tgp_cdiv(total, 256)), dim3(256), 0, tgp_hs(stream), recs, sel, det_img, window, camk, total,
                       lg, n_pts, out, tables);
    return TGP_LAUNCH_RESULT();
}

extern "C" int tgp_cloud_select(const uint32_t *recs, const int32_t *sel, const int *det_img, const int *window, const float *camk, int D,
                                int roi_size, int n_pts, float *out, tgp_stream_t stream)
{
    TGP_REQUIRE(window);
    return tgp_cloud_select_ex(recs, sel, det_img, window, camk, D, roi_size, n_pts, out, nullptr, stream);
}

// The same resampling without the host in the loop: a keyed bijection of [0, 2^b) (four Feistel rounds on b/2-bit
// halves), cycle-walked into [0, total), gives element i of a pseudo-random permutation of the cloud statelessly; the
// first n_pts elements are the sample.  Same distribution family as np.random.permutation(total)[:n_pts] (a uniformly
// chosen ordered subset), not the same draw; a cloud shorter than n_pts is tiled exactly as the reference tiles it.
__device__ __forceinline__ uint32_t mix32(uint32_t x)
{
    x ^= x >> 16, x *= 0x7feb352du, x ^= x >> 15, x *= 0x846ca68bu, x ^= x >> 16;
    return x;
}
__device__ __forceinline__ uint32_t feistel(uint32_t v, int half_bits, uint32_t key)
{
    const uint32_t mask = (1u << half_bits) - 1u;
    uint32_t l = v >> half_bits, r = v & mask;
#pragma unroll
    for (int round = 0; round < 4; ++round) {
        const uint32_t f = mix32(r ^ (key + 0x9e3779b9u * (round + 1))) & mask;
        const uint32_t nl = r;
        r = l ^ f;
        l = nl;
    }
    return (l << half_bits) | r;
}

__global__ void cloud_sample_kernel(const uint32_t *__restrict__ recs, const int *__restrict__ counts, const int *__restrict__ det_img,
                                    const int *__restrict__ window, const float *__restrict__ camk, int64_t total_out, int roi_log2, int n_pts,
                                    uint64_t seed, float *__restrict__ out)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total_out) return;
    const int j = (int)(t / n_pts), i = (int)(t % n_pts), cap = 1 << (2 * roi_log2);
    const int total = min(counts[j * 3 + 2], cap);
    float a = NAN, b = NAN, c = NAN;
    if (total > 0) {
        uint32_t k;
        if (total <= n_pts) {
            k = (uint32_t)(i % total);
        } else {
            int half_bits = 1;
            while ((1u << (2 * half_bits)) < (uint32_t)total) ++half_bits;
            const uint32_t key = mix32((uint32_t)seed ^ mix32((uint32_t)(seed >> 32) + 0x632be5abu * (uint32_t)(j + 1)));
            k = feistel((uint32_t)i, half_bits, key);
            while (k >= (uint32_t)total) k = feistel(k, half_bits, key);      // a bijection of [0, 2^2h): the walk returns
        }
        RoiGeom(window, camk, j, det_img[j], roi_log2).point(recs[(size_t)j * cap + k], a, b, c);
    }
    out[t * 3] = a, out[t * 3 + 1] = b, out[t * 3 + 2] = c;
}

extern "C" int tgp_cloud_sample(const uint32_t *recs, const int *counts, const int *det_img, const int *window, const float *camk, int D,
                                int roi_size, int n_pts, uint64_t seed, float *out, tgp_stream_t stream)
{
    TGP_REQUIRE(recs && counts && det_img && window && camk && out && D > 0 && n_pts > 0);
    const int lg = roi_log2_of(roi_size);
    if (lg < 0) return TGP_EUNSUPPORTED;
    const int64_t total = (int64_t)D * n_pts;
    hipLaunchKernelGGL(cloud_sample_kernel, dim3(tgp_cdiv(total, 256)), dim3(256), 0, tgp_hs(stream), recs, counts, det_img, window, camk, total,
                       lg, n_pts, seed, out);
    return TGP_LAUNCH_RESULT();
}
