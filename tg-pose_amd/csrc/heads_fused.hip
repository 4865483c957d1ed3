// The three pose heads' conv1 -> BatchNorm -> ReLU -> conv2 -> BatchNorm -> ReLU -> max over points as ONE kernel
// (network/fs_net_repo/PoseR.py:26-36, PoseTs.py:31-42 in eval mode, on the factored form of conv1: DESIGN.md section 3).
//
// As two launches the path writes and re-reads the (B*N, 3 x 1024) activation (404 MB at B = 32) and spends most of its time in
// the first launch's gather epilogue.  Here the activation never leaves the CU.  A wave owns 32 points of one head:
//   phase 1, per block of 32 conv1 channels:  acc1[channel][point] = W_fine[channel][:] . fine[point][:]   (K = 268, 17 steps of
//            v_mfma_f32_32x32x16_f16 x 3 split terms; the weights' fp16 planes come from LDS, the points' from registers);
//   epilogue 1 in registers: + bias + P1[near1(point)] + P2[near2(point)] (the coarse products of the factored layer), BatchNorm
//            fold, ReLU -- in the two-launch form's order -- then split into fp16 hi / lo;
//   phase 2: the MFMA accumulator layout gives lane (point r, half h) channels {4h + (e & 3) + 8 (e >> 2)} of the block: exactly a
//            32x32x16 A-operand fragment (8 k-values per lane) if conv2's K order is permuted accordingly.  The permutation is
//            applied to W2 when it is packed (heads_pack_w2_kernel), so the 16 values feed two MFMA steps of
//            acc2[point][out] += H[point][channels] . W2[out][channels] without touching LDS;
//   after the 32 channel blocks: + bias2, BatchNorm fold, ReLU, max over the object's points as order-preserving keys (atomicMax).
// One wave per SIMD (the points' fine features as B fragments: 136 registers, conv2's accumulators: 128), four waves per
// workgroup sharing the LDS-staged weight blocks (double buffered: 2 x (34.5 + 36 KB)).
#include "tgp_common.h"
#include "../../include/tgpose.h"

typedef _Float16 hf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 hf16x4 __attribute__((ext_vector_type(4)));
typedef float hf32x4 __attribute__((ext_vector_type(4)));
typedef float hf32x16 __attribute__((ext_vector_type(16)));

#define HF_STEPS 17                 // K steps of conv1: 272 / 16
#define HF_C1 1024                  // conv1 channels per head
#define HF_C2 256                   // conv2 channels per head
#define HF_NCB (HF_C1 / 32)
#define HF_AROW (HF_STEPS * 64 + 16)      // LDS row of the conv1 weight block: 17 steps x 2 planes x 32 B, + 16 B (conflict-free b128)
#define HF_WROW (2 * 64 + 16)             // LDS row of the conv2 weight block: 2 steps x 2 planes x 32 B, + 16 B
// every region a whole number of kilobytes: one LDS-DMA wave-instruction (64 lanes x 16 B) then lies inside one region
#define HF_ABYTES (35 * 1024)             // 32 rows x 1104 B = 35328, padded
#define HF_PBYTES 1024                    // bias | scale | shift of the block's 32 channels (384 B used)
#define HF_WBYTES (HF_C2 * HF_WROW)       // 36864 = 36 KB
#define HF_BUF (HF_ABYTES + HF_PBYTES + HF_WBYTES)
#define HF_NDMA (HF_BUF / 1024 / 4)       // 18 wave-instructions per wave and block

struct HeadsParams {
    const float *fine; int ldf, K;
    const uint16_t *wa_s;                 // [heads * 1024][17][2][16] fp16
    const float *p1; int ldp1; const int32_t *idx1;
    const float *p2; int ldp2; const int32_t *idx2;
    const float *bias1, *scale1, *shift1; // heads * 1024
    const uint16_t *w2p;                  // [heads][32][256][2][2][16] fp16
    const float *bias2, *scale2, *shift2; // heads * 256
    uint32_t *keys;                       // (heads, B, 256)
    int *overflow;
    int M, rows_per_obj, B, heads, tiles;
    // (round 4) the points' features as blocked fp16 planes (tgp_gemm_args.A_planes' layout): fragments load as 1 KB runs, no split
    const char *fine_pl; int fine_kt; const uint32_t *fine_amax;
};

__device__ __forceinline__ void hf_split(const float4 v, uint2 &hi, uint2 &lo)
{
    const hf32x4 x = {v.x, v.y, v.z, v.w};
    const hf16x4 h = __builtin_convertvector(x, hf16x4);
    const hf32x4 rest = x - __builtin_convertvector(h, hf32x4);
    const hf16x4 l = __builtin_convertvector(rest, hf16x4);
    hi = __builtin_bit_cast(uint2, h);
    lo = __builtin_bit_cast(uint2, l);
}

// PERSISTENT (round 4): the grid is one workgroup per CU and every workgroup walks the tiles t = blockIdx.x, + gridDim.x, ... (head-major,
// so the workgroups running at any moment share a head's weights in L2).  As one workgroup per tile, a tile was prologue (34 strided
// loads per lane + their split, the first weight block's DMA from cold) -> 32 channel blocks -> epilogue, and a quarter of a
// workgroup's ~125 us lay outside the channel-block loop with nothing to overlap it (one wave per SIMD, 144 KB of LDS: nothing else
// fits on the CU).  Here the next tile's operands arrive during the current tile's LAST channel block: its first weight block by
// the same double-buffered DMA that feeds every block (the chain of blocks simply continues across tiles), its points' fragments by
// loads issued right after the last use of the current ones (conv1 of block 31) and in flight under conv2 of block 31 and the
// tile's epilogue.
template <bool PLANES>
__global__ __launch_bounds__(256, 1) void heads_fused_kernel(HeadsParams p)
{
    extern __shared__ __attribute__((aligned(16))) char hf_smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int total = p.heads * p.tiles;
    int t = blockIdx.x;
    if (t >= total) return;

    uint4 bh[HF_STEPS], bl[HF_STEPS];     // the wave's points as B fragments: fp16 hi / lo planes of fine[row][16 s + 8 h .. + 7]
    // range guard state of the tile being loaded: the largest magnitude this lane splits into fp16 (NaN-poisoned once it met a NaN
    // or an infinity); with planes, the producer's per-block magnitude word says the same
    float amax_ld = 0.f, poison_ld = 0.f;
    const int nblk = (p.M + 31) >> 5;
    auto load_b = [&](const int ptile) {
        const int m0 = ptile * 128 + wave * 32;
        amax_ld = 0.f, poison_ld = 0.f;
        if constexpr (PLANES) {
            const int rb = min(m0 >> 5, nblk - 1);
            const char *src = p.fine_pl + (int64_t)rb * p.fine_kt * 2048 + lane * 16;
#pragma unroll
            for (int s = 0; s < HF_STEPS; ++s) {
                bh[s] = *reinterpret_cast<const uint4 *>(src + s * 2048);
                bl[s] = *reinterpret_cast<const uint4 *>(src + s * 2048 + 1024);
            }
            if (m0 + 32 > p.M) {                                    // (wave-uniform) rows past the end were never written: zero them
                const bool dead = m0 + r >= p.M;
#pragma unroll
                for (int s = 0; s < HF_STEPS; ++s)
                    if (dead) bh[s] = make_uint4(0u, 0u, 0u, 0u), bl[s] = make_uint4(0u, 0u, 0u, 0u);
            }
            const uint32_t am = p.fine_amax ? p.fine_amax[rb] : 0x3f800000u;
            amax_ld = __uint_as_float(am < 0x7f800000u ? am : 0x7f800000u);   // inf for an inf / NaN block: !(amax < 65504) below
        } else {
            const int row = min(m0 + r, p.M - 1);
            const float *fr = p.fine + (int64_t)row * p.ldf + 8 * h;
#pragma unroll
            for (int s = 0; s < HF_STEPS; ++s) {
                float4 v0 = *reinterpret_cast<const float4 *>(fr + 16 * s), v1 = *reinterpret_cast<const float4 *>(fr + 16 * s + 4);
                if (s == HF_STEPS - 1) {                            // the K tail: columns >= K are not the caller's to define
                    const int k0 = 16 * s + 8 * h;
                    v0.x = k0 + 0 < p.K ? v0.x : 0.f, v0.y = k0 + 1 < p.K ? v0.y : 0.f, v0.z = k0 + 2 < p.K ? v0.z : 0.f, v0.w = k0 + 3 < p.K ? v0.w : 0.f;
                    v1.x = k0 + 4 < p.K ? v1.x : 0.f, v1.y = k0 + 5 < p.K ? v1.y : 0.f, v1.z = k0 + 6 < p.K ? v1.z : 0.f, v1.w = k0 + 7 < p.K ? v1.w : 0.f;
                }
                amax_ld = fmaxf(amax_ld, fmaxf(fmaxf(fabsf(v0.x), fabsf(v0.y)), fmaxf(fabsf(v0.z), fabsf(v0.w))));
                amax_ld = fmaxf(amax_ld, fmaxf(fmaxf(fabsf(v1.x), fabsf(v1.y)), fmaxf(fabsf(v1.z), fabsf(v1.w))));
                poison_ld += 0.f * (((v0.x + v0.y) + (v0.z + v0.w)) + ((v1.x + v1.y) + (v1.z + v1.w)));
                uint2 h0, l0, h1, l1;
                hf_split(v0, h0, l0), hf_split(v1, h1, l1);
                bh[s] = make_uint4(h0.x, h0.y, h1.x, h1.y), bl[s] = make_uint4(l0.x, l0.y, l1.x, l1.y);
            }
        }
    };

    // ---- staging of a channel block: conv1 weight rows, their epilogue vectors, the permuted conv2 weight block, by LDS-DMA
    // (global_load_lds_dwordx4: no register destination).  A wave-instruction fills 1 KB of LDS linearly (wave-uniform base + 16 x
    // lane), so the padded image is written as it lies: kilobyte j of the buffer is wave (j % 4)'s instruction j / 4; a lane that
    // lands on a row's padding piece re-reads the row's last piece.  Issued one at a time between the MFMA groups.
    static_assert(HF_NDMA == 18 && HF_ABYTES / 1024 == 35, "the region tests below");
    int dma_off[HF_NDMA];                                       // the lane's source offset of its j0-th instruction (block-invariant)
#pragma unroll
    for (int j0 = 0; j0 < HF_NDMA; ++j0) {
        const int j = j0 * 4 + wave;
        if (j < 35) {                                            // conv1 weight rows: 69 pieces per LDS row, 68 in memory
            const int c = j * 64 + lane, rw = c / 69, pc = c % 69;
            dma_off[j0] = rw < 32 ? (rw * 68 + (pc < 68 ? pc : 67)) * 16 : 0;
        } else if (j == 35) {                                    // bias | scale | shift: 8 pieces each
            dma_off[j0] = lane < 24 ? (lane & 7) * 16 : 0;
        } else {                                                 // conv2 weight rows: 9 pieces per LDS row, 8 in memory
            const int c = (j - 36) * 64 + lane, rw = c / 9, pc = c % 9;
            dma_off[j0] = (rw * 8 + (pc < 8 ? pc : 7)) * 16;
        }
    }
    auto dma = [&](const int hd, const int cb, const int buf, const int j0) {
        const int j = j0 * 4 + wave;                             // wave-uniform
        const char *src;
        if (j < 35) src = reinterpret_cast<const char *>(p.wa_s) + ((int64_t)hd * HF_C1 + cb * 32) * (HF_STEPS * 64);
        else if (j == 35) {
            const float *v = lane < 8 ? p.bias1 : lane < 16 ? p.scale1 : p.shift1;
            src = reinterpret_cast<const char *>(v + hd * HF_C1 + cb * 32);
        } else src = reinterpret_cast<const char *>(p.w2p) + ((int64_t)hd * HF_NCB + cb) * (HF_C2 * 128);
        // inline assembly: opaque to the compiler's counters (no vmcnt(0) before the LDS reads that follow); vmcnt(0) is written by
        // hand before the barrier that ends a block.  The LDS base travels in m0 as a register-constrained input.
        const uint32_t lds = __builtin_amdgcn_readfirstlane(
            (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)(hf_smem + buf * HF_BUF + j * 1024));
        asm volatile("s_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src + dma_off[j0]), "{m0}"(lds) : "memory");
    };

    int hd = t / p.tiles, ptile = t - hd * p.tiles;
    load_b(ptile);
#pragma unroll
    for (int j0 = 0; j0 < HF_NDMA; ++j0) dma(hd, 0, 0, j0);
    __builtin_amdgcn_s_waitcnt(0x0f70);                         // vmcnt(0): this wave's DMA has landed
    __syncthreads();

    int par = 0;                                                // LDS buffer of the current channel block (blocks alternate across tiles too)
    for (;;) {
        const int tn = t + (int)gridDim.x;
        const bool more = tn < total;
        const int hd_n = more ? tn / p.tiles : hd, ptile_n = more ? tn - hd_n * p.tiles : ptile;
        const int m0 = ptile * 128 + wave * 32;                 // the wave's first point (may lie past M: then the wave only helps staging)
        const int row = min(m0 + r, p.M - 1);
        float amax = amax_ld, poison = poison_ld;
        const float amax_in = amax_ld;
        const int i1 = p.idx1[row], i2 = p.idx2[row];
        const float *g1p = p.p1 + (int64_t)i1 * p.ldp1 + hd * HF_C1 + 4 * h;       // + cb * 32 + 8 m: four channels of the lane
        const float *g2p = p.p2 + (int64_t)i2 * p.ldp2 + hd * HF_C1 + 4 * h;

        hf32x16 acc2[HF_C2 / 32];
#pragma unroll
        for (int ob = 0; ob < HF_C2 / 32; ++ob)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc2[ob][e] = 0.f;

        for (int cb = 0; cb < HF_NCB; ++cb, par ^= 1) {
            const char *base = hf_smem + par * HF_BUF;
            const bool last = cb + 1 == HF_NCB;
            const bool stage = !last || more;                    // a block follows: this tile's next one, or the next tile's first
            const int hd_s = last ? hd_n : hd, cb_s = last ? 0 : cb + 1;
            // the lane's gathered coarse products for this block: consumed after phase 1
            float4 g1[4], g2[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                g1[m] = *reinterpret_cast<const float4 *>(g1p + cb * 32 + 8 * m);
                g2[m] = *reinterpret_cast<const float4 *>(g2p + cb * 32 + 8 * m);
            }
            // ---- phase 1: conv1, 32 channels x 32 points
            hf32x16 acc1;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc1[e] = 0.f;
            const char *arow = base + r * HF_AROW + h * 16;
            {
                // fragments two steps ahead of their MFMAs: with one wave per SIMD nothing else hides the LDS latency
                uint4 fh0 = *reinterpret_cast<const uint4 *>(arow), fl0 = *reinterpret_cast<const uint4 *>(arow + 32);
                uint4 fh1 = *reinterpret_cast<const uint4 *>(arow + 64), fl1 = *reinterpret_cast<const uint4 *>(arow + 64 + 32);
#pragma unroll
                for (int s = 0; s < HF_STEPS; ++s) {
                    uint4 fh2 = fh1, fl2 = fl1;
                    if (s + 2 < HF_STEPS) {
                        fh2 = *reinterpret_cast<const uint4 *>(arow + (s + 2) * 64);
                        fl2 = *reinterpret_cast<const uint4 *>(arow + (s + 2) * 64 + 32);
                    }
                    // smallest terms first, as in the tile kernel: lo x hi, hi x lo, hi x hi
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(hf16x8, fl0), __builtin_bit_cast(hf16x8, bh[s]), acc1, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(hf16x8, fh0), __builtin_bit_cast(hf16x8, bl[s]), acc1, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(hf16x8, fh0), __builtin_bit_cast(hf16x8, bh[s]), acc1, 0, 0, 0);
                    // the next block's operands, 8 of the 18 pieces here (the rest in phase 2): they target the other buffer, whose
                    // last readers passed the barrier
                    if (s >= 1 && s <= 8 && stage) dma(hd_s, cb_s, par ^ 1, s - 1);
                    __builtin_amdgcn_sched_barrier(0);                // keep the lookahead: do not sink the reads to their uses
                    fh0 = fh1, fl0 = fl1, fh1 = fh2, fl1 = fl2;
                }
            }
            // the current points' fragments have had their last use: the next tile's start to arrive under conv2 and the epilogue
            if (last && more) load_b(ptile_n);
            // ---- epilogue 1: element e of the lane is channel 4 h + (e & 3) + 8 (e >> 2) of the block, point r
            uint4 a2h[2], a2l[2];
            {
                const float *pv = reinterpret_cast<const float *>(base + HF_ABYTES) + 4 * h;
                uint2 hh[4], ll[4];
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const float4 b = *reinterpret_cast<const float4 *>(pv + 8 * m), sc = *reinterpret_cast<const float4 *>(pv + 32 + 8 * m),
                                 sh = *reinterpret_cast<const float4 *>(pv + 64 + 8 * m);
                    float4 v = make_float4(acc1[4 * m], acc1[4 * m + 1], acc1[4 * m + 2], acc1[4 * m + 3]);
                    v.x += b.x, v.y += b.y, v.z += b.z, v.w += b.w;
                    v.x += g1[m].x, v.y += g1[m].y, v.z += g1[m].z, v.w += g1[m].w;
                    v.x += g2[m].x, v.y += g2[m].y, v.z += g2[m].z, v.w += g2[m].w;
                    v.x = v.x * sc.x + sh.x, v.y = v.y * sc.y + sh.y, v.z = v.z * sc.z + sh.z, v.w = v.w * sc.w + sh.w;
                    v.x = v.x > 0.f ? v.x : v.x * 0.f, v.y = v.y > 0.f ? v.y : v.y * 0.f;
                    v.z = v.z > 0.f ? v.z : v.z * 0.f, v.w = v.w > 0.f ? v.w : v.w * 0.f;
                    poison += 0.f * ((v.x + v.y) + (v.z + v.w));
                    amax = fmaxf(amax, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));     // (v >= 0 after the ReLU)
                    hf_split(v, hh[m], ll[m]);
                }
                a2h[0] = make_uint4(hh[0].x, hh[0].y, hh[1].x, hh[1].y), a2h[1] = make_uint4(hh[2].x, hh[2].y, hh[3].x, hh[3].y);
                a2l[0] = make_uint4(ll[0].x, ll[0].y, ll[1].x, ll[1].y), a2l[1] = make_uint4(ll[2].x, ll[2].y, ll[3].x, ll[3].y);
            }
            // ---- phase 2: conv2 partial sums over this block's 32 channels, 32 points x 256 outputs
            const char *wrow = base + HF_ABYTES + HF_PBYTES + r * HF_WROW + h * 16;
            {
                constexpr int NOB = HF_C2 / 32;
                auto wfrag = [&](int q, int plane) {                  // q = s2 * NOB + ob
                    return *reinterpret_cast<const uint4 *>(wrow + (q % NOB) * 32 * HF_WROW + (q / NOB) * 64 + plane * 32);
                };
                uint4 wh0 = wfrag(0, 0), wl0 = wfrag(0, 1), wh1 = wfrag(1, 0), wl1 = wfrag(1, 1);
#pragma unroll
                for (int q = 0; q < 2 * NOB; ++q) {
                    uint4 wh2 = wh1, wl2 = wl1;
                    if (q + 2 < 2 * NOB) wh2 = wfrag(q + 2, 0), wl2 = wfrag(q + 2, 1);
                    const int s2 = q / NOB, ob = q % NOB;
                    acc2[ob] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(hf16x8, a2h[s2]), __builtin_bit_cast(hf16x8, wl0), acc2[ob], 0, 0, 0);
                    acc2[ob] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(hf16x8, a2l[s2]), __builtin_bit_cast(hf16x8, wh0), acc2[ob], 0, 0, 0);
                    acc2[ob] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(hf16x8, a2h[s2]), __builtin_bit_cast(hf16x8, wh0), acc2[ob], 0, 0, 0);
                    if (q < HF_NDMA - 8 && stage) dma(hd_s, cb_s, par ^ 1, q + 8);   // the other 10, one per MFMA group
                    __builtin_amdgcn_sched_barrier(0);
                    wh0 = wh1, wl0 = wl1, wh1 = wh2, wl1 = wl2;
                }
            }
            __builtin_amdgcn_s_waitcnt(0x0f70);                      // vmcnt(0): this wave's share of the next block has landed
            __syncthreads();                                         // ... everybody's has, and this buffer's readers are done
        }

        // ---- epilogue 2: lane (out column r of block ob, half h) holds points (e & 3) + 8 (e >> 2) + 4 h of the wave's 32.
        // fp16 range guard: a lane that split a magnitude >= 65504 (or met a NaN) spoils its wave's sums.  With a flag to raise, the wave
        // writes no keys and the caller's predicated two-launch form (guarded arithmetic) supplies them; without one, NaN keys are loud.
        // And the small side: a wave whose input features are ALL below 2^-4 (and not all zero) would lose relative precision in every
        // product of conv1 (gemm.hip, small side of the range guard): same treatment.
        if (m0 < p.M) {
            const bool tiny_in = __ballot(amax_in >= 0.0625f) == 0ull && __ballot(amax_in > 0.f) != 0ull;
            if (p.overflow && (tiny_in || __ballot(!(amax < 65504.f) || poison != poison) != 0ull)) {
                if (lane == 0) atomicOr(p.overflow, 1);
            } else {
                const int obj0 = m0 / p.rows_per_obj, bound = (obj0 + 1) * p.rows_per_obj;
#pragma unroll
                for (int ob = 0; ob < HF_C2 / 32; ++ob) {
                    const int o = hd * HF_C2 + ob * 32 + r;
                    const float b2 = p.bias2[o], sc2 = p.scale2[o], sh2 = p.shift2[o];
                    uint32_t k0 = 0, k1 = 0;
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int prow = m0 + (e & 3) + 8 * (e >> 2) + 4 * h;
                        float v = acc2[ob][e] + b2;
                        v = v * sc2 + sh2;
                        v = v > 0.f ? v : v * 0.f;
                        const uint32_t key = prow < p.M ? tgp_float_key(v) : 0u;
                        if (prow >= bound) k1 = key > k1 ? key : k1;
                        else k0 = key > k0 ? key : k0;
                    }
                    const uint32_t o0 = (uint32_t)__shfl_xor((int)k0, 32, 64), o1 = (uint32_t)__shfl_xor((int)k1, 32, 64);
                    k0 = o0 > k0 ? o0 : k0, k1 = o1 > k1 ? o1 : k1;
                    if (h == 0) {
                        uint32_t *kp = p.keys + ((int64_t)hd * p.B + obj0) * HF_C2 + ob * 32 + r;
                        if (k0) atomicMax(kp, k0);
                        if (k1) atomicMax(kp + HF_C2, k1);
                    }
                }
            }
        }
        if (!more) break;
        t = tn, hd = hd_n, ptile = ptile_n;
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// conv -> BatchNorm -> LeakyReLU -> max over points of a factored layer whose activation only feeds the max (conv_5 of Face_Enc,
// FaceRecon.py:76-77 `conv_5` + `feat.max(1)`): the fused heads kernel's phase 1 with the operands' roles swapped -- the points'
// fragments are the A operand, the weight rows the B operand -- so that a lane holds ONE channel and 16 points: the epilogue's
// vectors are scalars per lane, the gathered coarse products are 4-byte loads coalesced over the channels, and the max over the
// points is 15 in-lane maxima + one cross-half shuffle.  No conv2 accumulators: ~190 registers, two waves per SIMD; a workgroup
// takes 128 points and CBW of the channel blocks (70 KB of LDS: two workgroups per CU).  Same products in the same order as the
// tile kernel (activation hi x weight lo, lo x hi, hi x hi; K ascending), same epilogue order.
#define CM_BUF (36 * 1024)                // conv weight rows (35 KB incl. padding) + 1 KB of epilogue vectors
#define CM_NDMA 9                         // wave-instructions per wave and block

struct ConvMaxParams {
    const float *fine; int ldf, K;
    const uint16_t *wa_s;
    const float *p1; int ldp1; const int32_t *idx1;
    const float *p2; int ldp2; const int32_t *idx2;
    const float *bias, *scale, *shift;
    float slope;
    uint32_t *keys; int ldk;              // (B, C) keys, row stride ldk
    int *overflow;
    int M, rows_per_obj, C, tiles, chunks;
    int main_tiles;                       // tiles [0, main_tiles) are cut into `chunks` channel chunks, the others into one per block
    const char *fine_pl; int fine_kt; const uint32_t *fine_amax;      // (round 4) the points' features as blocked fp16 planes
};

template <bool PLANES>
__global__ __launch_bounds__(256, 2) void conv_max_fused_kernel(ConvMaxParams p)
{
    extern __shared__ __attribute__((aligned(16))) char hf_smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    // The benchmark's 32896 points are 257 tiles of 128: in two chunks each that is 514 workgroups for 512 resident slots, and two of
    // them would run alone for a whole second round.  The tiles past the last full round are cut into single channel blocks
    // instead: many short workgroups that drain in a fraction of a round.
    const int ncb = p.C / 32;
    int ptile, cb0, cb1;
    if ((int)blockIdx.x < p.main_tiles * p.chunks) {
        ptile = blockIdx.x / p.chunks;
        const int cbw = (ncb + p.chunks - 1) / p.chunks;
        cb0 = (blockIdx.x % p.chunks) * cbw, cb1 = min(cb0 + cbw, ncb);
    } else {
        const int rest = blockIdx.x - p.main_tiles * p.chunks;
        ptile = p.main_tiles + rest / ncb;
        cb0 = rest % ncb, cb1 = cb0 + 1;
    }
    const int m0 = ptile * 128 + wave * 32;
    const int row = min(m0 + r, p.M - 1);

    uint4 ah[HF_STEPS], al[HF_STEPS];     // the wave's points: A fragments, fp16 hi / lo planes of fine[row][16 s + 8 h .. + 7]
    float amax = 0.f, poison = 0.f;
    if constexpr (PLANES) {
        // the fragments as they lie in the fine buffer's planes (1 KB runs per K-tile and plane), the block's magnitude word for the guard
        const int rb = min(m0 >> 5, ((p.M + 31) >> 5) - 1);
        const char *src = p.fine_pl + (int64_t)rb * p.fine_kt * 2048 + lane * 16;
#pragma unroll
        for (int s = 0; s < HF_STEPS; ++s) {
            ah[s] = *reinterpret_cast<const uint4 *>(src + s * 2048);
            al[s] = *reinterpret_cast<const uint4 *>(src + s * 2048 + 1024);
        }
        if (m0 + 32 > p.M) {                                        // rows past the end were never written
            const bool dead = m0 + r >= p.M;
#pragma unroll
            for (int s = 0; s < HF_STEPS; ++s)
                if (dead) ah[s] = make_uint4(0u, 0u, 0u, 0u), al[s] = make_uint4(0u, 0u, 0u, 0u);
        }
        const uint32_t am = p.fine_amax ? p.fine_amax[rb] : 0x3f800000u;
        amax = __uint_as_float(am < 0x7f800000u ? am : 0x7f800000u);
    } else {
        const float *fr = p.fine + (int64_t)row * p.ldf + 8 * h;
#pragma unroll
        for (int s = 0; s < HF_STEPS; ++s) {
            float4 v0 = *reinterpret_cast<const float4 *>(fr + 16 * s), v1 = *reinterpret_cast<const float4 *>(fr + 16 * s + 4);
            if (s == HF_STEPS - 1) {
                const int k0 = 16 * s + 8 * h;
                v0.x = k0 + 0 < p.K ? v0.x : 0.f, v0.y = k0 + 1 < p.K ? v0.y : 0.f, v0.z = k0 + 2 < p.K ? v0.z : 0.f, v0.w = k0 + 3 < p.K ? v0.w : 0.f;
                v1.x = k0 + 4 < p.K ? v1.x : 0.f, v1.y = k0 + 5 < p.K ? v1.y : 0.f, v1.z = k0 + 6 < p.K ? v1.z : 0.f, v1.w = k0 + 7 < p.K ? v1.w : 0.f;
            }
            amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v0.x), fabsf(v0.y)), fmaxf(fabsf(v0.z), fabsf(v0.w))));
            amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v1.x), fabsf(v1.y)), fmaxf(fabsf(v1.z), fabsf(v1.w))));
            poison += 0.f * (((v0.x + v0.y) + (v0.z + v0.w)) + ((v1.x + v1.y) + (v1.z + v1.w)));
            uint2 h0, l0, h1, l1;
            hf_split(v0, h0, l0), hf_split(v1, h1, l1);
            ah[s] = make_uint4(h0.x, h0.y, h1.x, h1.y), al[s] = make_uint4(l0.x, l0.y, l1.x, l1.y);
        }
    }
    // the lane's 16 points (accumulator rows (e & 3) + 8 (e >> 2) + 4 h): element offsets of their coarse products' rows.  They
    // depend on (h, e) only, so they live in 256 bytes of LDS per wave instead of 32 registers per lane.
    __shared__ __attribute__((aligned(16))) int s_off[4][2][2][16];
    {
        const int lvl = lane >> 5, hh = (lane >> 4) & 1, e = lane & 15;
        const int pr = min(m0 + (e & 3) + 8 * (e >> 2) + 4 * hh, p.M - 1);
        s_off[wave][lvl][hh][e] = lvl ? p.idx2[pr] * p.ldp2 : p.idx1[pr] * p.ldp1;
    }
    const bool live = m0 < p.M;
    const bool bad = __ballot(!(amax < 65504.f) || poison != poison) != 0ull ||      // fp16 range guard, as in the heads kernel
                     (__ballot(amax >= 0.0625f) == 0ull && __ballot(amax > 0.f) != 0ull);   // ... and its small side (all inputs < 2^-4)
    if (live && bad && p.overflow && lane == 0) atomicOr(p.overflow, 1);
    const int obj0 = m0 / p.rows_per_obj, bound = (obj0 + 1) * p.rows_per_obj;

    int dma_off[CM_NDMA];
#pragma unroll
    for (int j0 = 0; j0 < CM_NDMA; ++j0) {
        const int j = j0 * 4 + wave;
        if (j < 35) {
            const int c = j * 64 + lane, rw = c / 69, pc = c % 69;
            dma_off[j0] = rw < 32 ? (rw * 68 + (pc < 68 ? pc : 67)) * 16 : 0;
        } else dma_off[j0] = lane < 24 ? (lane & 7) * 16 : 0;
    }
    auto dma = [&](int cb, int buf, int j0) {
        const int j = j0 * 4 + wave;
        const char *src;
        if (j < 35) src = reinterpret_cast<const char *>(p.wa_s) + (int64_t)cb * 32 * (HF_STEPS * 64);
        else {
            const float *v = lane < 8 ? p.bias : lane < 16 ? p.scale : p.shift;
            src = reinterpret_cast<const char *>(v + cb * 32);
        }
        const uint32_t lds = __builtin_amdgcn_readfirstlane(
            (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)(hf_smem + buf * CM_BUF + j * 1024));
        asm volatile("s_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src + dma_off[j0]), "{m0}"(lds) : "memory");
    };
#pragma unroll
    for (int j0 = 0; j0 < CM_NDMA; ++j0) dma(cb0, 0, j0);
    __builtin_amdgcn_s_waitcnt(0x0f70);
    __syncthreads();

    for (int cb = cb0; cb < cb1; ++cb) {
        const int buf = (cb - cb0) & 1;
        const char *base = hf_smem + buf * CM_BUF;
        float g1[16], g2[16];
        {
            const int4 *o1 = reinterpret_cast<const int4 *>(s_off[wave][0][h]), *o2 = reinterpret_cast<const int4 *>(s_off[wave][1][h]);
            const float *q1 = p.p1 + cb * 32 + r, *q2 = p.p2 + cb * 32 + r;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int4 a = o1[m], c = o2[m];
                g1[4 * m] = q1[a.x], g1[4 * m + 1] = q1[a.y], g1[4 * m + 2] = q1[a.z], g1[4 * m + 3] = q1[a.w];
                g2[4 * m] = q2[c.x], g2[4 * m + 1] = q2[c.y], g2[4 * m + 2] = q2[c.z], g2[4 * m + 3] = q2[c.w];
            }
        }
        hf32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        const char *wrow = base + r * HF_AROW + h * 16;
        {
            uint4 fh0 = *reinterpret_cast<const uint4 *>(wrow), fl0 = *reinterpret_cast<const uint4 *>(wrow + 32);
            uint4 fh1 = *reinterpret_cast<const uint4 *>(wrow + 64), fl1 = *reinterpret_cast<const uint4 *>(wrow + 64 + 32);
#pragma unroll
            for (int s = 0; s < HF_STEPS; ++s) {
                uint4 fh2 = fh1, fl2 = fl1;
                if (s + 2 < HF_STEPS) {
                    fh2 = *reinterpret_cast<const uint4 *>(wrow + (s + 2) * 64);
                    fl2 = *reinterpret_cast<const uint4 *>(wrow + (s + 2) * 64 + 32);
                }
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(hf16x8, ah[s]), __builtin_bit_cast(hf16x8, fl0), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(hf16x8, al[s]), __builtin_bit_cast(hf16x8, fh0), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(hf16x8, ah[s]), __builtin_bit_cast(hf16x8, fh0), acc, 0, 0, 0);
                if (s >= 1 && s <= CM_NDMA && cb + 1 < cb1) dma(cb + 1, buf ^ 1, s - 1);
                __builtin_amdgcn_sched_barrier(0);
                fh0 = fh1, fl0 = fl1, fh1 = fh2, fl1 = fl2;
            }
        }
        // epilogue: the lane's channel is 32 cb + r, its 16 accumulator elements are 16 points
        const float *pv = reinterpret_cast<const float *>(base + 35 * 1024);
        const float b = pv[r], sc = pv[32 + r], sh = pv[64 + r];
        uint32_t k0 = 0, k1 = 0;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int prow = m0 + (e & 3) + 8 * (e >> 2) + 4 * h;
            float v = acc[e] + b;
            v += g1[e];
            v += g2[e];
            v = v * sc + sh;
            v = v > 0.f ? v : v * p.slope;
            const uint32_t key = prow < p.M ? tgp_float_key(v) : 0u;
            if (prow >= bound) k1 = key > k1 ? key : k1;
            else k0 = key > k0 ? key : k0;
        }
        const uint32_t o0 = (uint32_t)__shfl_xor((int)k0, 32, 64), o1 = (uint32_t)__shfl_xor((int)k1, 32, 64);
        k0 = o0 > k0 ? o0 : k0, k1 = o1 > k1 ? o1 : k1;
        if (live && h == 0 && !(bad && p.overflow)) {
            uint32_t *kp = p.keys + (int64_t)obj0 * p.ldk + cb * 32 + r;
            if (k0) atomicMax(kp, k0);
            if (k1) atomicMax(kp + p.ldk, k1);
        }
        __builtin_amdgcn_s_waitcnt(0x0f70);
        __syncthreads();
    }
}

extern "C" int tgp_conv_max_fused(const tgp_conv_max_fused_args *a, tgp_stream_t stream)
{
    TGP_REQUIRE(a && a->fine && a->wa_s && a->p1 && a->p2 && a->idx1 && a->idx2 && a->bias && a->scale && a->shift && a->keys);
    TGP_REQUIRE(a->M > 0 && a->C > 0 && (a->C & 31) == 0 && a->rows_per_obj >= 32 && a->M % a->rows_per_obj == 0 && a->ldk >= a->C);
    TGP_REQUIRE(a->K > 0 && a->K <= 16 * HF_STEPS && a->K > 16 * (HF_STEPS - 1) && a->ldf >= 16 * HF_STEPS && (a->ldf & 3) == 0);
    TGP_REQUIRE(a->ldp1 >= a->C && a->ldp2 >= a->C);
    // the coarse products' rows are addressed with 32-bit element offsets
    TGP_REQUIRE((int64_t)a->p1_rows * a->ldp1 < (1ll << 31) && (int64_t)a->p2_rows * a->ldp2 < (1ll << 31));
    auto al16 = [](const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    TGP_REQUIRE(al16(a->fine) && al16(a->wa_s) && al16(a->bias) && al16(a->scale) && al16(a->shift));
    ConvMaxParams p;
    p.fine = a->fine, p.ldf = a->ldf, p.K = a->K;
    p.wa_s = reinterpret_cast<const uint16_t *>(a->wa_s);
    p.p1 = a->p1, p.ldp1 = a->ldp1, p.idx1 = a->idx1, p.p2 = a->p2, p.ldp2 = a->ldp2, p.idx2 = a->idx2;
    p.bias = a->bias, p.scale = a->scale, p.shift = a->shift, p.slope = a->slope;
    p.keys = a->keys, p.ldk = a->ldk, p.overflow = a->overflow;
    p.M = a->M, p.rows_per_obj = a->rows_per_obj, p.C = a->C, p.tiles = tgp_cdiv(a->M, 128);
    p.fine_pl = reinterpret_cast<const char *>(a->fine_planes), p.fine_kt = a->fine_kt, p.fine_amax = a->fine_amax;
    TGP_REQUIRE(!p.fine_pl || (p.fine_kt >= HF_STEPS && (reinterpret_cast<uintptr_t>(p.fine_pl) & 15) == 0));
    // channel chunks: enough workgroups for two per CU; a few tiles past a whole number of rounds go in single channel blocks
    p.chunks = (p.C / 32) >= 2 && p.tiles < 512 ? 2 : 1;
    const int round_tiles = 512 / p.chunks, over = p.tiles % round_tiles;
    p.main_tiles = (p.tiles > round_tiles && over > 0 && over <= 8) ? p.tiles - over : p.tiles;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(conv_max_fused_kernel<true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 2 * CM_BUF);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(conv_max_fused_kernel<false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 2 * CM_BUF);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const int grid = p.main_tiles * p.chunks + (p.tiles - p.main_tiles) * (p.C / 32);
    if (p.fine_pl) hipLaunchKernelGGL(conv_max_fused_kernel<true>, dim3(grid), dim3(256), 2 * CM_BUF, tgp_hs(stream), p);
    else hipLaunchKernelGGL(conv_max_fused_kernel<false>, dim3(grid), dim3(256), 2 * CM_BUF, tgp_hs(stream), p);
    return TGP_LAUNCH_RESULT();
}

// W2 (heads, 256, 1024) fp32 -> [head][channel block][out][step][plane][16] fp16 with conv2's K order permuted to the layout the
// conv1 accumulators leave the channels in: slot 8 h + t of step s2 is channel 32 cb + 16 s2 + 8 (t >> 2) + 4 h + (t & 3)
__global__ void heads_pack_w2_kernel(const float *__restrict__ w2, int heads, uint16_t *__restrict__ out)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)heads * HF_NCB * HF_C2 * 32;
    if (t >= total) return;
    const int slot = (int)(t & 15), s2 = (int)((t >> 4) & 1), o = (int)((t >> 5) % HF_C2);
    const int cb = (int)((t >> 5) / HF_C2 % HF_NCB), hd = (int)((t >> 5) / HF_C2 / HF_NCB);
    const int hh = slot >> 3, tt = slot & 7;
    const int ch = 32 * cb + 16 * s2 + 8 * (tt >> 2) + 4 * hh + (tt & 3);
    const float v = w2[((int64_t)hd * HF_C2 + o) * HF_C1 + ch];
    const _Float16 hi = (_Float16)v;
    const _Float16 lo = (_Float16)(v - (float)hi);
    uint16_t *dst = out + ((((int64_t)hd * HF_NCB + cb) * HF_C2 + o) * 2 + s2) * 32;
    dst[slot] = __builtin_bit_cast(uint16_t, hi);
    dst[16 + slot] = __builtin_bit_cast(uint16_t, lo);
}

extern "C" int tgp_heads_pack_w2(const float *w2, int heads, void *out, tgp_stream_t stream)
{
    TGP_REQUIRE(w2 && out && heads > 0);
    const int64_t total = (int64_t)heads * HF_NCB * HF_C2 * 32;
    hipLaunchKernelGGL(heads_pack_w2_kernel, dim3(tgp_cdiv(total, 256)), dim3(256), 0, tgp_hs(stream), w2, heads,
                       reinterpret_cast<uint16_t *>(out));
    return TGP_LAUNCH_RESULT();
}

extern "C" int tgp_heads_fused(const tgp_heads_fused_args *a, tgp_stream_t stream)
{
    TGP_REQUIRE(a && a->fine && a->wa_s && a->p1 && a->p2 && a->idx1 && a->idx2 && a->bias1 && a->scale1 && a->shift1 && a->w2p &&
                a->bias2 && a->scale2 && a->shift2 && a->keys);
    TGP_REQUIRE(a->M > 0 && a->B > 0 && a->heads > 0 && a->rows_per_obj >= 32 && (int64_t)a->B * a->rows_per_obj == a->M);
    TGP_REQUIRE(a->K > 0 && a->K <= 16 * HF_STEPS && a->K > 16 * (HF_STEPS - 1) && a->ldf >= 16 * HF_STEPS && (a->ldf & 3) == 0);
    TGP_REQUIRE((a->ldp1 & 3) == 0 && (a->ldp2 & 3) == 0 && a->ldp1 >= a->heads * HF_C1 && a->ldp2 >= a->heads * HF_C1);
    auto al16 = [](const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    TGP_REQUIRE(al16(a->fine) && al16(a->wa_s) && al16(a->p1) && al16(a->p2) && al16(a->bias1) && al16(a->scale1) && al16(a->shift1) &&
                al16(a->w2p));
    HeadsParams p;
    p.fine = a->fine, p.ldf = a->ldf, p.K = a->K;
    p.wa_s = reinterpret_cast<const uint16_t *>(a->wa_s);
    p.p1 = a->p1, p.ldp1 = a->ldp1, p.idx1 = a->idx1, p.p2 = a->p2, p.ldp2 = a->ldp2, p.idx2 = a->idx2;
    p.bias1 = a->bias1, p.scale1 = a->scale1, p.shift1 = a->shift1;
    p.w2p = reinterpret_cast<const uint16_t *>(a->w2p);
    p.bias2 = a->bias2, p.scale2 = a->scale2, p.shift2 = a->shift2;
    p.keys = a->keys;
    p.overflow = a->overflow;
    p.M = a->M, p.rows_per_obj = a->rows_per_obj, p.B = a->B, p.heads = a->heads, p.tiles = tgp_cdiv(a->M, 128);
    if (a->rows > 0) {          // the first a->rows rows only (the caller covers the rest through the tile kernels)
        TGP_REQUIRE(a->rows <= a->M && a->rows % 128 == 0);
        p.M = a->rows, p.tiles = a->rows / 128;
    }
    p.fine_pl = reinterpret_cast<const char *>(a->fine_planes), p.fine_kt = a->fine_kt, p.fine_amax = a->fine_amax;
    TGP_REQUIRE(!p.fine_pl || (p.fine_kt >= HF_STEPS && (reinterpret_cast<uintptr_t>(p.fine_pl) & 15) == 0));
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(heads_fused_kernel<true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 2 * HF_BUF);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(heads_fused_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    2 * HF_BUF);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    // persistent: one workgroup per CU (144 KB of LDS, 512 registers per lane: nothing else fits beside it), each walking its tiles
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
            cus = 256;
    }
    const int total = p.heads * p.tiles;
    TGP_REQUIRE(a->workgroups >= 0);
    const int want = a->workgroups > 0 ? a->workgroups : cus;
    const dim3 grid(total < want ? total : want);
    if (p.fine_pl) hipLaunchKernelGGL(heads_fused_kernel<true>, grid, dim3(256), 2 * HF_BUF, tgp_hs(stream), p);
    else hipLaunchKernelGGL(heads_fused_kernel<false>, grid, dim3(256), 2 * HF_BUF, tgp_hs(stream), p);
    return TGP_LAUNCH_RESULT();
}
