// An HS layer's LAST GEMM and the NEXT layer's projection GEMM as ONE kernel (network/fs_net_repo/gcn3d.py:87-89 / 108-112 / 147-155:
// fm = act(bn(W1 [g | ...] + conv2's global half (a per-object bias) + g + f_STE)) followed, one layer on, by gcn3d.py:170
// `feature_map @ weights + bias`), for the two places of Face_Enc (FaceRecon.py:61-66) where no pooling sits between them:
// conv_0 -> conv_1 (1028 points, 132 -> 128 -> 1152) and conv_2 -> conv_3 (257 points, 256 -> 256 -> 2304).
//
// As two launches of the tile kernel (csrc/gemm_pp.hip) the pair is a latency chain of 514 small workgroups followed by a launch
// that re-reads its 128-column operand for each of 9 / 18 column tiles and is bound by its prologue + 8-16 steps + 64 KB store
// burst per tile (conv_1's projection: 168 MB in 66 us).  Here a wave owns 32 points for both layers, in the
// style of heads_fused.hip / dec_fused.hip (one wave per SIMD, every non-matrix instruction in the gaps behind the MFMAs):
//   layer 1: acc1[out block][point] += W1 . A, the points' operand fragments loaded once from the planes their producer wrote
//            (16 bytes per lane, K-tile and plane), the weights from LDS (linear LDS-DMA of a fragment-blocked image, tgp_hs_chain_pack);
//   epilogue 1 in registers, in the tile kernel's order (+ per-object bias, + residual(s), BatchNorm fold, ReLU); the result is
//            stored (fp32 and as the blocked fp16 planes other consumers stage) and becomes the next layer's operand WITHOUT leaving
//            the registers: the accumulator layout gives lane (point r, half h) channels {4 h + (e & 3) + 8 (e >> 2)} of a block, and one
//            v_permlane32_swap per register pair between the wave's halves turns two channel quads into the eight consecutive
//            k-values of a 32x32x16 operand fragment.  K order natural (ascending), fp16 hi / lo split of the fp32 value: the
//            projection's sums are the tile kernel's, bit for bit (same products, same order) -- unlike dec_fused.hip, which permutes K;
//   layer 2: four output blocks at a time (four independent accumulator chains), weights by LDS-DMA in 64 KB units, + bias, stored
//            from the accumulators as 16-byte pieces that complete 32-byte sectors per row.
// Weights stay in L2 / LDS, the operand is read once, and the intermediate activation is never re-read: FLOP per staged byte is that
// of a 128-row x N tile instead of a 128 x 128 one.
//
// fp16 range: a wave whose operand block (the producer's magnitude word) or whose intermediate block holds a magnitude >= 65504 / a NaN,
// or is wholly below 2^-4, raises the device flag; the caller's two tile-kernel launches follow predicated on it (tgp_gemm_args.pred)
// and rewrite both results with their own per-tile guards.  The rule is the tile kernels' applied per 32-row block (a superset of their
// per-tile rule), so whenever the flag stays 0 both paths compute the same bits.
#include "tgp_common.h"
#include "../../include/tgpose.h"
#include <type_traits>
#include <cstdlib>

typedef _Float16 hc16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 hc16x4 __attribute__((ext_vector_type(4)));
typedef float hc32x4 __attribute__((ext_vector_type(4)));
typedef float hc32x16 __attribute__((ext_vector_type(16)));

#define HC_UNIT (64 * 1024)                // one staging unit in memory: 32 (K-step, output block) pairs x 2 planes of 1 KB
#define HC_BUF (72 * 1024)                 // an LDS buffer: room for 36 pairs (layer 1 of the 132 -> 128 shape is ONE unit of 9 K-steps x 4 blocks)
#define HC_SB() __builtin_amdgcn_sched_barrier(0)
#define HR_SLOT (32 * 1024)                // hs_proj_kernel: one slot of its ring of four half units
__device__ __forceinline__ constexpr int hc_vmcnt(int n) { return 0x0f70 | (n & 15) | ((n >> 4) << 14); }

struct HcParams {
    const char *a_pl; int a_kt; const uint32_t *a_amax;          // layer 1's operand (M rows) as blocked fp16 planes + its magnitude words
    const char *units;                                           // tgp_hs_chain_pack's image
    const float *rowbias; int ldrb, rows_per_obj;                // (B, N1) per-object bias of layer 1
    const float *res1; int ldr1;                                 // residuals of layer 1 (res2 may be NULL)
    const float *res2; int ldr2;
    const float *scale1, *shift1;                                // BatchNorm fold of layer 1 (both NULL: none)
    int relu;
    float *c1; int ldc1;                                         // layer 1's result, fp32
    char *c1_pl; int c1_kt, c1_kt0; uint32_t *c1_amax;           // ... and as planes (first K-tile c1_kt0), magnitude words (may be NULL)
    const float *bias2;
    float *c2; int ldc2;                                         // layer 2's result (M, N2)
    int *flag;
    int M, main_tiles, tiles, nsplit;
    int knob;                                                    // (development build) timing-only variants: 1 = no stores of layer 2, 2 = no stores of layer 1, 4 = no layer-2 MFMAs
    // hs_proj_kernel only: the fp32 operands of the exact path, the layer's true K, its groups of four output blocks
    const float *a_f32; int lda; const float *w_f32; int ldw; int K2, ngt;
};

__device__ __forceinline__ float hc_mix_lo(uint32_t hpair, float v)     // v - (float)(low half of hpair), one rounding
{
    float d;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(hpair), "v"(v));
    return d;
}
__device__ __forceinline__ float hc_mix_hi(uint32_t hpair, float v)     // v - (float)(high half of hpair)
{
    float d;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(hpair), "v"(v));
    return d;
}

// ---- the staging-unit body shared by the kernels of this file (expects: hc_smem, lane, dma(buf, j0), u_src)
// One staging unit = up to 32 (K-step s, output block j) pairs, q = s * NB + j, three MFMAs each (smallest terms first, as in the
// tile kernel: W lo x A hi, W hi x A lo, W hi x A hi); gap 1: the weight fragments of pair q + 2 (+ FILL_A), gap 2: a DMA piece of
// the next unit (NDN of them; + FILL_B), gap 3: FILL_C.  BH / BL: the points' hi / lo fragments of K-step s.
#define HC_BODY(BUF, NQ, ACC, NB, BH, BL, NDN, WAITCNT, FILL_A, FILL_B, FILL_C)                                                       \
    {                                                                                                                        \
        const char *wrow = hc_smem + (BUF) * HC_BUF + lane * 16;                                                             \
        auto wfrag = [&](int q, int plane) { return *reinterpret_cast<const uint4 *>(wrow + (q * 2 + plane) * 1024); };      \
        uint4 wh0 = wfrag(0, 0), wl0 = wfrag(0, 1), wh1 = wfrag((NQ) > 1 ? 1 : 0, 0), wl1 = wfrag((NQ) > 1 ? 1 : 0, 1);      \
        HC_SB();                                                                                                             \
        _Pragma("unroll") for (int q = 0; q < (NQ); ++q) {                                                                   \
            const int s = q / (NB), j = q % (NB);                                                                            \
            uint4 wh2 = wh1, wl2 = wl1;                                                                                      \
            const hc16x8 bh = BH, bl = BL;                                                                                   \
            ACC[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(hc16x8, wl0), bh, ACC[j], 0, 0, 0);           \
            HC_SB();                                                                                                         \
            if (q + 2 < (NQ)) wh2 = wfrag(q + 2, 0), wl2 = wfrag(q + 2, 1);                                                  \
            FILL_A;                                                                                                          \
            HC_SB();                                                                                                         \
            ACC[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(hc16x8, wh0), bl, ACC[j], 0, 0, 0);           \
            HC_SB();                                                                                                         \
            if (q < (NDN)) dma((BUF) ^ 1, q);                                                                                \
            FILL_B;                                                                                                          \
            HC_SB();                                                                                                         \
            ACC[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(hc16x8, wh0), bh, ACC[j], 0, 0, 0);           \
            HC_SB();                                                                                                         \
            FILL_C;                                                                                                          \
            HC_SB();                                                                                                         \
            wh0 = wh1, wl0 = wl1, wh1 = wh2, wl1 = wl2;                                                                      \
        }                                                                                                                    \
        _Pragma("unroll") for (int q = (NQ); q < 16; ++q)       /* (a short unit has fewer gaps than the next unit has pieces) */ \
            if (q < (NDN)) dma((BUF) ^ 1, q);                                                                                \
        __builtin_amdgcn_s_waitcnt(WAITCNT);    /* this wave's share of the next unit has landed (what was issued after it may fly) */ \
        __syncthreads();                        /* ... everybody's has, and this buffer's readers are done */                \
    }

#define HC_ZERO(ACC)                                                  \
    _Pragma("unroll") for (int j = 0; j < 4; ++j)                     \
        _Pragma("unroll") for (int e = 0; e < 16; ++e) ACC[j][e] = 0.f;
#define HC_GROUP(GI, ACC, PREV, GU0)                                                                                              \
    {                                                                                                                        \
        HC_ZERO(ACC)                                                                                                         \
        const bool have_prev = (GI) > 0;                                                                                     \
        _Pragma("unroll") for (int w = 0; w < U2; ++w) {                                                                     \
            const int gu = (GU0) + (GI) * U2 + w;                          /* the unit's place in this workgroup's stream */    \
            const int ndn = ((GI) + 1 < ng || w + 1 < U2) ? 16 : 0;                                                          \
            u_src = u2_base + (int64_t)((GI) * U2 + w + 1) * HC_UNIT;                                                        \
            /* the previous group's 16 stores go out in the unit's second half, behind its DMA pieces: the wait at the unit's end */  \
            /* leaves exactly them in flight (vmcnt(16)) instead of draining the store path once per unit */                         \
            if (have_prev && w == 0 && m0 < p.M) {      /* (a wave past the last row issues no stores: it takes the plain wait) */ \
                HC_BODY(gu & 1, 32, ACC, 4, __builtin_bit_cast(hc16x8, a2h[8 * w + s]), __builtin_bit_cast(hc16x8, a2l[8 * w + s]), \
                        ndn, 0x4f70, if (q >= 16) store2(PREV, g0 + (GI) - 1, q - 16), , )                                   \
            } else {                                                                                                         \
                HC_BODY(gu & 1, 32, ACC, 4, __builtin_bit_cast(hc16x8, a2h[8 * w + s]), __builtin_bit_cast(hc16x8, a2l[8 * w + s]), \
                        ndn, 0x0f70, , , )                                                                                   \
            }                                                                                                                \
        }                                                                                                                    \
    }
// layer 2 of a workgroup: groups g0 .. g0 + ng - 1 (expects a2h / a2l [K2T], U2, u2_base, g0, ng, rowc, live, h, p; the first unit staged
// in buffer GU0_ & 1; BIASP_: the groups' bias in LDS)
#define HC_LAYER2(GU0_, BIASP_) \
    /* ================================================================= layer 2: per group of four output blocks U2 units of eight K-steps */ \
    hc32x16 accA[4], accB[4]; \
    float *c2p = p.c2 + (int64_t)rowc * p.ldc2 + 4 * h; \
    /* a finished group leaves in the gaps of the next one: block jj, quad g of group `grp` from `acc`, + bias */ \
    auto store2 = [&](const hc32x16 (&acc)[4], const int grp, const int idx) {      /* idx = 0 .. 15: (block, quad) */ \
        const int jj = idx >> 2, g = idx & 3; \
        const int c = 128 * grp + 32 * jj + 8 * g; \
        const float4 b = *reinterpret_cast<const float4 *>((BIASP_) + (c - 128 * g0) + 4 * h); \
        const float4 v = make_float4(acc[jj][4 * g] + b.x, acc[jj][4 * g + 1] + b.y, acc[jj][4 * g + 2] + b.z, acc[jj][4 * g + 3] + b.w); \
        if (live && !(p.knob & 1)) *reinterpret_cast<float4 *>(c2p + c) = v; \
    }; \
    /* group gi of this workgroup into ACC, the previous group's stores (from PREV) in its first unit's gaps */ \
    int gi = 0; \
    if (p.knob & 4) return; \
    _Pragma("unroll 1") \
    for (; gi + 1 < ng; gi += 2) { \
        HC_GROUP(gi, accA, accB, GU0_) \
        HC_GROUP(gi + 1, accB, accA, GU0_) \
    } \
    if (gi < ng) {                                                /* an odd group count: the last one */ \
        HC_GROUP(gi, accA, accB, GU0_) \
    _Pragma("unroll") \
        for (int i = 0; i < 16; ++i) store2(accA, g0 + gi, i); \
    } else { \
    _Pragma("unroll") \
        for (int i = 0; i < 16; ++i) store2(accB, g0 + ng - 1, i); \
    }

// K1T: K-tiles of layer 1's operand; NB1 = N1 / 32; NG: groups of four output blocks of layer 2 (N2 = 128 NG); a main workgroup takes
// NG / nsplit of them, the workgroups of the tiles past the last full round one each.
template <int K1T, int NB1, int NG>
__global__ __launch_bounds__(256, 1) void hs_chain_kernel(HcParams p)
{
    constexpr int Q1 = K1T * NB1, QCAP = Q1 <= 36 ? Q1 : 32;                  // layer 1: (K-step, block) pairs; pairs per unit
    constexpr int U1 = (Q1 + QCAP - 1) / QCAP, SPU1 = QCAP / NB1;             // ... its units, K-steps per unit
    constexpr int U1M = (Q1 + 31) / 32;                                       // ... and the 64 KB units its image takes in memory
    constexpr int K2T = 2 * NB1, U2 = K2T / 8;                                // layer 2: K-steps, units per group (8 K-steps x 4 blocks each)
    static_assert(QCAP % NB1 == 0 && K2T % 8 == 0 && Q1 % QCAP == 0, "unit geometry");
    extern __shared__ __attribute__((aligned(16))) char hc_smem[];            // 2 x HC_BUF, then scale1 | shift1 (N1 each) | bias2 (this workgroup's groups)
    float *s_vec = reinterpret_cast<float *>(hc_smem + 2 * HC_BUF);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    int tile, g0, ng;
    {
        const int gpw = NG / p.nsplit, mainwg = p.main_tiles * p.nsplit;
        if ((int)blockIdx.x < mainwg) tile = blockIdx.x / p.nsplit, g0 = (blockIdx.x % p.nsplit) * gpw, ng = gpw;
        else tile = p.main_tiles + ((int)blockIdx.x - mainwg) / NG, g0 = ((int)blockIdx.x - mainwg) % NG, ng = 1;
    }
    const bool writer = g0 == 0;                                  // layer 1's result is stored by the workgroup that holds the tile's first group
    const int m0 = tile * 128 + wave * 32;                        // the wave's first point (may lie past M: then the wave only helps staging)
    const int nblk = (p.M + 31) >> 5;
    const int rb = min(m0 >> 5, nblk - 1);
    const int row = m0 + r;
    const bool live = row < p.M;
    const int rowc = min(row, p.M - 1);

    // ---- staging: unit u is 64 KB at units + 64 KB u; piece j = 4 j0 + wave is 1 KB at offset 1024 j of the unit and of the buffer
    const uint32_t voff0 = lane * 16 + wave * 1024;
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)hc_smem) + wave * 1024;
    const char *u_src = p.units;                                  // scalar base of the unit being staged
    auto dma = [&](const int buf, const int j0) {
        const uint32_t lds = lds0 + buf * HC_BUF + j0 * 4096;
        const uint32_t vo = voff0 + j0 * 4096;
        // inline assembly: opaque to the compiler's counters; vmcnt(0) is written by hand before the barrier that ends a unit
        asm volatile("s_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(vo), "s"(u_src), "{m0}"(lds) : "memory");
    };
    constexpr int ND_FIRST = (QCAP * 2 + 3) / 4;
#pragma unroll
    for (int j0 = 0; j0 < ND_FIRST; ++j0) dma(0, j0);

    // the epilogues' per-channel vectors into LDS: a global load between the stores would make every wait for it drain the stores
    // issued before it (one in-order counter for loads and stores on gfx9)
    constexpr int N1 = 32 * NB1;
    for (int i = tid; i < 2 * N1 + 128 * ng; i += 256)
        s_vec[i] = i < N1 ? (p.scale1 ? p.scale1[i] : 1.f) : i < 2 * N1 ? (p.shift1 ? p.shift1[i - N1] : 0.f) : p.bias2[128 * g0 + (i - 2 * N1)];
    // fp16 range guard of the operand, from what its producer recorded (bits of max |a| per 32-row block)
    if (m0 < p.M && p.a_amax) {
        const uint32_t am = p.a_amax[rb];
        if ((am >= 0x477fe000u || (am != 0u && am < 0x3d800000u)) && lane == 0) atomicOr(p.flag, 1);
    }
    // the wave's points as B fragments of layer 1: K-tile kt, plane q at a_pl + ((rb * a_kt + kt) * 2 + q) * 1024 + 16 lane
    uint4 a1h[K1T], a1l[K1T];
    {
        const char *src = p.a_pl + (int64_t)rb * p.a_kt * 2048 + lane * 16;
#pragma unroll
        for (int s = 0; s < K1T; ++s) {
            a1h[s] = *reinterpret_cast<const uint4 *>(src + s * 2048);
            a1l[s] = *reinterpret_cast<const uint4 *>(src + s * 2048 + 1024);
        }
    }
    // epilogue 1's operands (per-object bias, residuals) of the first block: requested before layer 1, they arrive under it
    const int obj = rowc / p.rows_per_obj;
    const float *rbp = p.rowbias ? p.rowbias + (int64_t)obj * p.ldrb + 4 * h : nullptr;
    const float *r1p = p.res1 ? p.res1 + (int64_t)rowc * p.ldr1 + 4 * h : nullptr;
    const float *r2p = p.res2 ? p.res2 + (int64_t)rowc * p.ldr2 + 4 * h : nullptr;
    // ... a ring of HC_RING blocks' operands: block j's are requested three blocks ahead and BEFORE the previous block's stores go out
    // (the wait for them then leaves those stores in flight; loads and stores share one in-order counter on gfx9)
    constexpr int HC_RING = 4;
    float4 ring[HC_RING][3][4];
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    auto fetch = [&](const int j, float4 (&d)[3][4]) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c = 32 * j + 8 * g;
            d[0][g] = rbp ? *reinterpret_cast<const float4 *>(rbp + c) : zero4;
            d[1][g] = r1p ? *reinterpret_cast<const float4 *>(r1p + c) : zero4;
            d[2][g] = r2p ? *reinterpret_cast<const float4 *>(r2p + c) : zero4;
        }
    };
#pragma unroll
    for (int j = 0; j < HC_RING - 1 && j < NB1; ++j) fetch(j, ring[j]);
    hc32x16 acc1[NB1];
#pragma unroll
    for (int j = 0; j < NB1; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc1[j][e] = 0.f;

    __builtin_amdgcn_s_waitcnt(0x0f70);                           // vmcnt(0): this wave's DMA (and fragments) have landed
    __syncthreads();

    // ================================================================= layer 1: U1 units of SPU1 K-steps x NB1 output blocks (the last may be short)
    const char *u2_base = p.units + (int64_t)(U1M + g0 * U2) * HC_UNIT;         // this workgroup's first unit of layer 2
    auto unit1 = [&](auto uc) {
        constexpr int u = decltype(uc)::value;
        constexpr int nq = QCAP;
        constexpr int nq_next = u + 1 < U1 ? QCAP : 32;          // (then layer 2's first unit)
        u_src = u + 1 < U1 ? p.units + (int64_t)(u + 1) * QCAP * 2048 : u2_base;
        HC_BODY(u & 1, nq, acc1, NB1, __builtin_bit_cast(hc16x8, a1h[u * SPU1 + s]), __builtin_bit_cast(hc16x8, a1l[u * SPU1 + s]),
                (nq_next * 2 + 3) / 4, 0x0f70, , , )
    };
    unit1(std::integral_constant<int, 0>{});
    if constexpr (U1 > 1) unit1(std::integral_constant<int, 1>{});
    if constexpr (U1 > 2) unit1(std::integral_constant<int, 2>{});
    if constexpr (U1 > 3) unit1(std::integral_constant<int, 3>{});
    static_assert(U1 <= 4, "layer 1: at most four units");

    // ================================================================= epilogue 1: the tile kernel's order, element by element
    // lane (point r, half h) holds channels 32 j + 8 g + 4 h .. + 3 of block j in elements 4 g .. 4 g + 3
    uint4 a2h[K2T], a2l[K2T];                                     // layer 2's operand fragments: K-step 2 j + s2 = channels 32 j + 16 s2 .. + 15
    uint32_t amid = 0u;
    {
        float *c1p = p.c1 + (int64_t)rowc * p.ldc1 + 4 * h;
        char *plp = p.c1_pl ? p.c1_pl + ((int64_t)rb * p.c1_kt + p.c1_kt0) * 2048 + lane * 16 : nullptr;
        const bool st1 = writer && live && !(p.knob & 2);
#pragma unroll
        for (int j = 0; j < NB1; ++j) {
            if (j + HC_RING - 1 < NB1) fetch(j + HC_RING - 1, ring[(j + HC_RING - 1) % HC_RING]);
            HC_SB();
            const float4 (&lrb)[4] = ring[j % HC_RING][0], (&lr1)[4] = ring[j % HC_RING][1], (&lr2)[4] = ring[j % HC_RING][2];
            float4 v[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = 32 * j + 8 * g;
                v[g] = make_float4(acc1[j][4 * g], acc1[j][4 * g + 1], acc1[j][4 * g + 2], acc1[j][4 * g + 3]);
                if (rbp) v[g].x += lrb[g].x, v[g].y += lrb[g].y, v[g].z += lrb[g].z, v[g].w += lrb[g].w;
                if (r1p) v[g].x += lr1[g].x, v[g].y += lr1[g].y, v[g].z += lr1[g].z, v[g].w += lr1[g].w;
                if (r2p) v[g].x += lr2[g].x, v[g].y += lr2[g].y, v[g].z += lr2[g].z, v[g].w += lr2[g].w;
                if (p.scale1) {
                    const float4 sc = *reinterpret_cast<const float4 *>(s_vec + c + 4 * h), sh = *reinterpret_cast<const float4 *>(s_vec + N1 + c + 4 * h);
                    v[g].x = v[g].x * sc.x + sh.x, v[g].y = v[g].y * sc.y + sh.y, v[g].z = v[g].z * sc.z + sh.z, v[g].w = v[g].w * sc.w + sh.w;
                }
                if (p.relu) {
                    v[g].x = v[g].x > 0.f ? v[g].x : v[g].x * 0.f, v[g].y = v[g].y > 0.f ? v[g].y : v[g].y * 0.f;
                    v[g].z = v[g].z > 0.f ? v[g].z : v[g].z * 0.f, v[g].w = v[g].w > 0.f ? v[g].w : v[g].w * 0.f;
                }
                const uint32_t b0 = __float_as_uint(v[g].x) & 0x7fffffffu, b1 = __float_as_uint(v[g].y) & 0x7fffffffu;
                const uint32_t b2 = __float_as_uint(v[g].z) & 0x7fffffffu, b3 = __float_as_uint(v[g].w) & 0x7fffffffu;
                const uint32_t m01 = b0 > b1 ? b0 : b1, m23 = b2 > b3 ? b2 : b3, m = m01 > m23 ? m01 : m23;
                if (live) amid = m > amid ? m : amid;
            }
            HC_SB();
#pragma unroll
            for (int g = 0; g < 4; ++g)
                if (st1) *reinterpret_cast<float4 *>(c1p + 32 * j + 8 * g) = v[g];
            // two channel quads of each half -> the eight consecutive k-values of a fragment: the upper half's quad g = 2 s2 and the
            // lower half's quad 2 s2 + 1 change places
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                float4 x = v[2 * s2], y = v[2 * s2 + 1];
                {
                    auto sw = [](float &a, float &b) {
                        const auto t = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
                        a = __uint_as_float(t[0]), b = __uint_as_float(t[1]);
                    };
                    sw(x.x, y.x), sw(x.y, y.y), sw(x.z, y.z), sw(x.w, y.w);
                }
                const hc32x4 fx = {x.x, x.y, x.z, x.w}, fy = {y.x, y.y, y.z, y.w};
                const uint2 hx = __builtin_bit_cast(uint2, __builtin_convertvector(fx, hc16x4));
                const uint2 hy = __builtin_bit_cast(uint2, __builtin_convertvector(fy, hc16x4));
                const hc32x4 lx = {hc_mix_lo(hx.x, x.x), hc_mix_hi(hx.x, x.y), hc_mix_lo(hx.y, x.z), hc_mix_hi(hx.y, x.w)};
                const hc32x4 ly = {hc_mix_lo(hy.x, y.x), hc_mix_hi(hy.x, y.y), hc_mix_lo(hy.y, y.z), hc_mix_hi(hy.y, y.w)};
                const uint2 qx = __builtin_bit_cast(uint2, __builtin_convertvector(lx, hc16x4));
                const uint2 qy = __builtin_bit_cast(uint2, __builtin_convertvector(ly, hc16x4));
                a2h[2 * j + s2] = make_uint4(hx.x, hx.y, hy.x, hy.y);
                a2l[2 * j + s2] = make_uint4(qx.x, qx.y, qy.x, qy.y);
                if (st1 && plp) {                                // (rows past M stay unwritten, as the tile kernel leaves them)
                    *reinterpret_cast<uint4 *>(plp + (2 * j + s2) * 2048) = a2h[2 * j + s2];
                    *reinterpret_cast<uint4 *>(plp + (2 * j + s2) * 2048 + 1024) = a2l[2 * j + s2];
                }
            }
        }
    }
    // the intermediate's magnitude word (the consumers' guard) and this kernel's own guard of it
    {
        uint32_t m = amid;
#pragma unroll
        for (int s = 1; s < 64; s <<= 1) {
            const uint32_t o = (uint32_t)__shfl_xor((int)m, s);
            m = o > m ? o : m;
        }
        if (lane == 0 && m0 < p.M) {
            if (writer && p.c1_amax && m) atomicMax(p.c1_amax + rb, m);
            if (m >= 0x477fe000u || (m != 0u && m < 0x3d800000u)) atomicOr(p.flag, 1);
        }
    }

    HC_LAYER2(U1, s_vec + 2 * N1)
}

// ------------------------------------------------------------------------------------------------------------------------
// A projection GEMM alone -- C = A W^T (+ bias) with K = 128, 256 or 512 and N a multiple of 128 (gcn3d.py:170 `feature_map @ weights +
// bias` of conv_1 .. conv_4: N = 9 x the layer's width; the two coarse products of the factored wide layers, K = 512, N = 4608) -- as the layer-2 half of the kernel above: a wave keeps its 32 rows' operand
// fragments (from the planes their producer wrote) for ALL output columns, the weights stream through LDS once per 128 rows instead of
// once per 128 x 128 tile, and the result leaves at the rate the memory takes it (stores behind the DMA pieces, counted waits).  The
// tile kernel on this shape is bound by staging both operands again for every column tile and by its store bursts (conv_1's projection:
// 168 MB in 66 us).  Same products in the same order, same fp16 range rule per 128-row tile: the tile kernel's bits, except for a tile
// the rule sends to the exact path, which is computed here by fp32 fma chains in ascending k (the tile kernel uses the fp32 MFMA).
template <int K2T>
__global__ __launch_bounds__(256, 1) void hs_proj_kernel(HcParams p)
{
    constexpr int U2 = K2T / 8;
    extern __shared__ __attribute__((aligned(16))) char hc_smem[];            // 4 x HR_SLOT, then the bias of this workgroup's groups
    float *s_vec = reinterpret_cast<float *>(hc_smem + 4 * HR_SLOT);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    int tile, g0, ng;
    {
        // a main tile's groups go to nsplit workgroups in contiguous, nearly equal shares; the tiles behind them one group per workgroup
        const int mainwg = p.main_tiles * p.nsplit;
        if ((int)blockIdx.x < mainwg) {
            const int sp = blockIdx.x % p.nsplit;
            tile = blockIdx.x / p.nsplit, g0 = sp * p.ngt / p.nsplit, ng = (sp + 1) * p.ngt / p.nsplit - g0;
        } else tile = p.main_tiles + ((int)blockIdx.x - mainwg) / p.ngt, g0 = ((int)blockIdx.x - mainwg) % p.ngt, ng = 1;
    }
    const int m0 = tile * 128 + wave * 32;
    const int nblk = (p.M + 31) >> 5;
    const int rb = min(m0 >> 5, nblk - 1);
    const int row = m0 + r;
    const bool live = row < p.M;
    const int rowc = min(row, p.M - 1);
    (void)r;
    // fp16 range rule of the tile kernel, per 128-row tile: a magnitude >= 65504 (or a NaN), or nothing at or above 2^-4 -> exact path
    bool exact = false;
    if (p.a_amax) {
        uint32_t am = 0u;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int b = tile * 4 + i;
            const uint32_t v = b < nblk ? p.a_amax[b] : 0u;
            am = v > am ? v : am;
        }
        exact = am >= 0x477fe000u || (am != 0u && am < 0x3d800000u);
    }
    if (exact) {                                                  // (workgroup-uniform; nothing staged yet)
        for (int n = 128 * g0 + lane; n < 128 * (g0 + ng); n += 64) {
            const float *wr = p.w_f32 + (int64_t)n * p.ldw;
            const float bn = p.bias2 ? p.bias2[n] : 0.f;
            for (int rr = 0; rr < 32 && m0 + rr < p.M; ++rr) {
                const float *ar = p.a_f32 + (int64_t)(m0 + rr) * p.lda;
                float acc = 0.f;
                for (int k = 0; k < p.K2; ++k) acc = fmaf(ar[k], wr[k], acc);
                p.c2[(int64_t)(m0 + rr) * p.ldc2 + n] = acc + bn;
            }
        }
        return;
    }
    // ---- staging: the workgroup's weights are a stream of HALF units (32 KB = 16 (K-step, block) pairs = 48 MFMAs per wave) through a
    // ring of four LDS slots: half unit k + 3 is requested while k is multiplied, so a piece has two half units (~3000 cycles) to
    // arrive.  With two 64 KB buffers the next unit was requested one unit ahead and waited for at the end of the current one: on the
    // K = 512 products that wait was a third of the kernel (138 us where the MFMAs are 81).
    const uint32_t voff0 = lane * 16 + wave * 1024;
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)hc_smem) + wave * 1024;
    const char *u2_base = p.units + (int64_t)g0 * U2 * HC_UNIT;
    const char *u_src = u2_base;
    const int nhu = ng * U2 * 2;                                  // half units of this workgroup
    auto dmah = [&](const int slot, const int j0) {               // piece j = 4 j0 + wave of the half unit at u_src
        const uint32_t lds = lds0 + slot * HR_SLOT + j0 * 4096;
        const uint32_t vo = voff0 + j0 * 4096;
        asm volatile("s_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(vo), "s"(u_src), "{m0}"(lds) : "memory");
    };
    // the wave's operand fragments and the bias first, then the first three half units: the matrix work starts when the fragments and
    // half unit 0 are in (the rounds of half units 1 and 2 -- the 16 youngest requests -- still fly)
    uint4 a2h[K2T], a2l[K2T];
    {
        const char *src = p.a_pl + (int64_t)rb * p.a_kt * 2048 + lane * 16;
#pragma unroll
        for (int s = 0; s < K2T; ++s) {
            a2h[s] = *reinterpret_cast<const uint4 *>(src + s * 2048);
            a2l[s] = *reinterpret_cast<const uint4 *>(src + s * 2048 + 1024);
        }
    }
    for (int i = tid; i < 128 * ng; i += 256) s_vec[i] = p.bias2 ? p.bias2[128 * g0 + i] : 0.f;
    HC_SB();
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (k < nhu) {
            u_src = u2_base + (int64_t)k * HR_SLOT;
#pragma unroll
            for (int j0 = 0; j0 < 8; ++j0) dmah(k, j0);
        }
    }
    if (nhu >= 3) __builtin_amdgcn_s_waitcnt(0x0070 | (16 & 15) | ((16 >> 4) << 14));      // vmcnt(16) lgkmcnt(0)
    else __builtin_amdgcn_s_waitcnt(0x0070);                      // vmcnt(0) lgkmcnt(0)
    __builtin_amdgcn_s_barrier();

    // (a wave past the last row issues no stores: its waits count none, i.e. wait for more; a partly live wave issues all of them,
    // masked)
    const bool full = m0 < p.M;
    hc32x16 accA[4], accB[4];
    float *c2p = p.c2 + (int64_t)rowc * p.ldc2 + 4 * h;
    // a finished group leaves in the gaps of the next one: block jj, quad g of group `grp` from `acc`, + bias.  (Store shapes measured with
    // the development build's knobs: this direct form touches 32 rows x 32 bytes per instruction; 16 rows x 64 bytes after one
    // v_permlane16_swap per register pair was slower -- the swaps are vector instructions --, and the same bytes as 8 rows x 128 bytes
    // through an LDS tile saturated the LDS, which the fragment reads already load to 60 %.)
    auto store2 = [&](const hc32x16 (&acc)[4], const int grp, const int idx) {      // idx = 0 .. 15: (block, quad)
        const int jj = idx >> 2, g = idx & 3;
        const int c = 128 * grp + 32 * jj + 8 * g;
        const float4 b = *reinterpret_cast<const float4 *>(s_vec + (c - 128 * g0) + 4 * h);
        const float4 v = make_float4(acc[jj][4 * g] + b.x, acc[jj][4 * g + 1] + b.y, acc[jj][4 * g + 2] + b.z, acc[jj][4 * g + 3] + b.w);
        if (live && !(p.knob & 1)) *reinterpret_cast<float4 *>(c2p + c) = v;
    };
    // half unit K_ (slot K_ & 3): pairs q = 0 .. 15 = K-steps S0_ .. S0_ + 3 x four output blocks into ACC; gap 1: the fragments of pair
    // q + 2 and FILL_A, gap 2: a piece round of half unit K_ + 3.  The wait at its end leaves in flight what was requested after half
    // unit K_ + 1: the rounds of K_ + 2 and K_ + 3 (8 each, where they exist) and this half unit's 16 stores (NST: it has them).
#define HR_HALF(K_, ACC, S0_, NST, FILL_A)                                                                                   \
    {                                                                                                                        \
        const int k_ = (K_);                                                                                                 \
        const char *wrow = hc_smem + (k_ & 3) * HR_SLOT + lane * 16;                                                         \
        const bool stage = k_ + 3 < nhu;                                                                                     \
        u_src = u2_base + (int64_t)(k_ + 3) * HR_SLOT;                                                                       \
        auto wfrag = [&](int q, int plane) { return *reinterpret_cast<const uint4 *>(wrow + (q * 2 + plane) * 1024); };      \
        uint4 wh0 = wfrag(0, 0), wl0 = wfrag(0, 1), wh1 = wfrag(1, 0), wl1 = wfrag(1, 1);                                    \
        HC_SB();                                                                                                             \
        _Pragma("unroll") for (int q = 0; q < 16; ++q) {                                                                     \
            const int s = (S0_) + q / 4, j = q % 4;                                                                          \
            uint4 wh2 = wh1, wl2 = wl1;                                                                                      \
            const hc16x8 bh = __builtin_bit_cast(hc16x8, a2h[s]), bl = __builtin_bit_cast(hc16x8, a2l[s]);                   \
            ACC[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(hc16x8, wl0), bh, ACC[j], 0, 0, 0);           \
            HC_SB();                                                                                                         \
            if (q + 2 < 16) wh2 = wfrag(q + 2, 0), wl2 = wfrag(q + 2, 1);                                                    \
            FILL_A;                                                                                                          \
            HC_SB();                                                                                                         \
            ACC[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(hc16x8, wh0), bl, ACC[j], 0, 0, 0);           \
            HC_SB();                                                                                                         \
            if (q < 8 && stage) dmah((k_ + 3) & 3, q);                                                                       \
            HC_SB();                                                                                                         \
            ACC[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(hc16x8, wh0), bh, ACC[j], 0, 0, 0);           \
            HC_SB();                                                                                                         \
            wh0 = wh1, wl0 = wl1, wh1 = wh2, wl1 = wl2;                                                                      \
        }                                                                                                                    \
        {                                                                                                                    \
            const int nd = (k_ + 2 < nhu ? 1 : 0) + (stage ? 1 : 0);                                                         \
            if ((NST) && full) {                                                                                             \
                if (nd == 2) __builtin_amdgcn_s_waitcnt(hc_vmcnt(32));                                                       \
                else if (nd == 1) __builtin_amdgcn_s_waitcnt(hc_vmcnt(24));                                                  \
                else __builtin_amdgcn_s_waitcnt(hc_vmcnt(16));                                                               \
            } else {                                                                                                         \
                if (nd == 2) __builtin_amdgcn_s_waitcnt(hc_vmcnt(16));                                                       \
                else if (nd == 1) __builtin_amdgcn_s_waitcnt(hc_vmcnt(8));                                                   \
                else __builtin_amdgcn_s_waitcnt(hc_vmcnt(0));                                                                \
            }                                                                                                                \
        }                                                                                                                    \
        __builtin_amdgcn_s_barrier();       /* the next half unit is in for everybody, and this slot's readers are done */     \
    }
    // group GI of this workgroup into ACC; the previous group's 16 stores (from PREV) ride in its second half unit (NST: are there any)
#define HR_GROUP(GI, ACC, PREV)                                                                                              \
    {                                                                                                                        \
        HC_ZERO(ACC)                                                                                                         \
        const bool have_prev = (GI) > 0;                                                                                     \
        _Pragma("unroll") for (int w = 0; w < U2; ++w) {                                                                     \
            HR_HALF(((GI) * U2 + w) * 2, ACC, 8 * w, false, )                                                                \
            HR_HALF(((GI) * U2 + w) * 2 + 1, ACC, 8 * w + 4, (w == 0 && have_prev), if (w == 0 && have_prev) store2(PREV, g0 + (GI) - 1, q)) \
        }                                                                                                                    \
    }
    if (p.knob & 4) return;
#pragma unroll 1
    for (int gi = 0; gi < ng; gi += 2) {
        HR_GROUP(gi, accA, accB)
        if (gi + 1 < ng) HR_GROUP(gi + 1, accB, accA)
    }
    if (ng & 1) {
#pragma unroll
        for (int i = 0; i < 16; ++i) store2(accA, g0 + ng - 1, i);
    } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) store2(accB, g0 + ng - 1, i);
    }
#undef HR_GROUP
#undef HR_HALF
}

// ---- weights -> staging units.  Layer 1: W1 (N1, K1) -> U1 units, pair q = s * NB1 + j of unit u = K-step u * (32 / NB1) + s, output
// block j; layer 2: W2 (N2, N1) -> per group of four output blocks (N2 / 128 groups) N1 / 128 units, pair q = s * 4 + j of unit w =
// K-step 8 w + s, output block 4 group + j.  A pair is two pieces of 1 KB, [lane = 32 h + r][8 fp16] = W[32 block + r][16 step + 8 h + t]
// as its fp16 hi (first piece) / lo part; columns >= K are zero.
__global__ void hs_chain_pack_kernel(const float *__restrict__ W, int ld, int N, int K, int nb, int steps, int units, int layer2,
                                     uint16_t *__restrict__ out)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)units * 32 * 512;
    if (t >= total) return;
    const int tt = (int)(t & 7), rr = (int)((t >> 3) & 31), hh = (int)((t >> 8) & 1);
    const int q = (int)((t >> 9) & 31);
    const int unit = (int)(t >> 14);
    int step, blk;
    if (layer2) {
        const int upg = steps / 8;                               // units per group
        step = (unit % upg) * 8 + q / 4, blk = (unit / upg) * 4 + q % 4;
    } else {
        const int spu = 32 / nb;
        step = unit * spu + q / nb, blk = q % nb;
    }
    const int col = 16 * step + 8 * hh + tt, rowi = 32 * blk + rr;
    const float v = (step < steps && col < K && rowi < N) ? W[(int64_t)rowi * ld + col] : 0.f;
    const _Float16 hi = (_Float16)v;
    const _Float16 lo = (_Float16)(v - (float)hi);
    uint16_t *dst = out + (int64_t)unit * (HC_UNIT / 2) + (int64_t)(q * 2) * 512 + (hh * 32 + rr) * 8 + tt;
    dst[0] = __builtin_bit_cast(uint16_t, hi);
    dst[512] = __builtin_bit_cast(uint16_t, lo);
}

static bool hc_shape(int K1, int N1, int N2, int &k1t, int &u1, int &ng, int &u2)
{
    if (!((N1 == 128 && N2 == 1152 && K1 > 128 && K1 <= 144) || (N1 == 256 && N2 == 2304 && K1 > 240 && K1 <= 256))) return false;
    k1t = (K1 + 15) / 16;
    const int nb1 = N1 / 32;
    u1 = (k1t * nb1 + 31) / 32, ng = N2 / 128, u2 = (2 * nb1) / 8;
    return true;
}

extern "C" int64_t tgp_hs_chain_pack_bytes(int K1, int N1, int N2)
{
    int k1t, u1, ng, u2;
    if (!hc_shape(K1, N1, N2, k1t, u1, ng, u2)) return -1;
    return (int64_t)(u1 + ng * u2) * HC_UNIT;
}

extern "C" int tgp_hs_chain_pack(const float *w1, int ld1, int K1, int N1, const float *w2, int ld2, int N2, void *out, tgp_stream_t stream)
{
    int k1t, u1, ng, u2;
    TGP_REQUIRE(w1 && w2 && out && (reinterpret_cast<uintptr_t>(out) & 15) == 0 && ld1 >= K1 && ld2 >= N1);
    TGP_REQUIRE(hc_shape(K1, N1, N2, k1t, u1, ng, u2));
    uint16_t *o = reinterpret_cast<uint16_t *>(out);
    hipLaunchKernelGGL(hs_chain_pack_kernel, dim3(u1 * 64), dim3(256), 0, tgp_hs(stream), w1, ld1, N1, K1, N1 / 32, k1t, u1, 0, o);
    hipLaunchKernelGGL(hs_chain_pack_kernel, dim3(ng * u2 * 64), dim3(256), 0, tgp_hs(stream), w2, ld2, N2, N1, 4, N1 / 16, ng * u2, 1,
                       o + (int64_t)u1 * (HC_UNIT / 2));
    return TGP_LAUNCH_RESULT();
}

#ifdef TGP_DEV
static int tgp_hs_chain_knobs = 0;
extern "C" int tgp_debug_set_hs_chain_knobs(int v) { tgp_hs_chain_knobs = v; return 0; }
#endif

extern "C" int tgp_hs_chain(const tgp_hs_chain_args *a, tgp_stream_t stream)
{
    TGP_REQUIRE(a && a->a_planes && a->units && a->c1 && a->bias2 && a->c2 && a->flag && a->M > 0);
    int k1t, u1, ng, u2;
    TGP_REQUIRE(hc_shape(a->K1, a->N1, a->N2, k1t, u1, ng, u2) && a->a_kt >= k1t);
    TGP_REQUIRE((a->scale1 == nullptr) == (a->shift1 == nullptr));
    TGP_REQUIRE(!a->rowbias || (a->rows_per_obj > 0 && a->M % a->rows_per_obj == 0 && (a->ldrb & 3) == 0));
    TGP_REQUIRE((a->ldc1 & 3) == 0 && (a->ldc2 & 3) == 0 && a->ldc1 >= a->N1 && a->ldc2 >= a->N2);
    TGP_REQUIRE(!a->res1 || ((a->ldr1 & 3) == 0)) ;
    TGP_REQUIRE(!a->res2 || ((a->ldr2 & 3) == 0));
    auto al16 = [](const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    TGP_REQUIRE(al16(a->a_planes) && al16(a->units) && al16(a->c1) && al16(a->c2) && al16(a->bias2) && al16(a->rowbias) && al16(a->res1) &&
                al16(a->res2) && al16(a->scale1) && al16(a->shift1) && al16(a->c1_planes));
    TGP_REQUIRE(!a->c1_planes || (a->c1_kt >= a->c1_kt0 + a->N1 / 16 && a->c1_kt0 >= 0));
    HcParams p;
    p.a_pl = reinterpret_cast<const char *>(a->a_planes), p.a_kt = a->a_kt, p.a_amax = a->a_amax;
    p.units = reinterpret_cast<const char *>(a->units);
    p.rowbias = a->rowbias, p.ldrb = a->ldrb, p.rows_per_obj = a->rowbias ? a->rows_per_obj : a->M;
    p.res1 = a->res1, p.ldr1 = a->ldr1, p.res2 = a->res2, p.ldr2 = a->ldr2;
    p.scale1 = a->scale1, p.shift1 = a->shift1, p.relu = a->relu;
    p.c1 = a->c1, p.ldc1 = a->ldc1;
    p.c1_pl = reinterpret_cast<char *>(a->c1_planes), p.c1_kt = a->c1_kt, p.c1_kt0 = a->c1_kt0, p.c1_amax = a->c1_amax;
    p.bias2 = a->bias2, p.c2 = a->c2, p.ldc2 = a->ldc2, p.flag = a->flag;
    p.M = a->M, p.tiles = tgp_cdiv(a->M, 128);
    p.knob = 0;
#ifdef TGP_DEV
    p.knob = tgp_hs_chain_knobs;
#endif
    // conv_0 -> conv_1 (N1 = 128): a workgroup takes all nine groups of its 128 points (the launch is bound by the 168 MB it writes);
    // conv_2 -> conv_3 (N1 = 256, 65 tiles): three workgroups per tile, six groups each.  Tiles past the last full round of 256
    // workgroups are cut into one workgroup per group.
    p.nsplit = a->N1 == 128 ? 1 : 3;
    const int round_tiles = 256 / p.nsplit, over = p.tiles % round_tiles;
    p.main_tiles = (p.tiles > round_tiles && over > 0 && over <= 8) ? p.tiles - over : p.tiles;
    const int grid = p.main_tiles * p.nsplit + (p.tiles - p.main_tiles) * ng;
    static TgpLdsAttr attr_a, attr_b;
    const int lds = 2 * HC_BUF + (2 * a->N1 + a->N2) * 4;
    if (a->N1 == 128) {
        if (const int e = tgp_lds_attr(attr_a, reinterpret_cast<const void *>(hs_chain_kernel<9, 4, 9>), lds)) return e;
        hipLaunchKernelGGL((hs_chain_kernel<9, 4, 9>), dim3(grid), dim3(256), lds, tgp_hs(stream), p);
    } else {
        if (const int e = tgp_lds_attr(attr_b, reinterpret_cast<const void *>(hs_chain_kernel<16, 8, 18>), lds)) return e;
        hipLaunchKernelGGL((hs_chain_kernel<16, 8, 18>), dim3(grid), dim3(256), lds, tgp_hs(stream), p);
    }
    return TGP_LAUNCH_RESULT();
}

extern "C" int64_t tgp_proj_pack_bytes(int K, int N)
{
    if (!((K == 128 || K == 256 || K == 512) && N > 0 && N % 128 == 0)) return -1;
    return (int64_t)(N / 128) * (K / 128) * HC_UNIT;
}

extern "C" int tgp_proj_pack(const float *w, int ld, int K, int N, void *out, tgp_stream_t stream)
{
    TGP_REQUIRE(w && out && (reinterpret_cast<uintptr_t>(out) & 15) == 0 && ld >= K && tgp_proj_pack_bytes(K, N) > 0);
    const int units = (N / 128) * (K / 128);
    hipLaunchKernelGGL(hs_chain_pack_kernel, dim3(units * 64), dim3(256), 0, tgp_hs(stream), w, ld, N, K, 4, K / 16, units, 1,
                       reinterpret_cast<uint16_t *>(out));
    return TGP_LAUNCH_RESULT();
}

extern "C" int tgp_proj_planes(const tgp_proj_planes_args *a, tgp_stream_t stream)
{
    TGP_REQUIRE(a && a->a_planes && a->units && a->c && a->M > 0 && tgp_proj_pack_bytes(a->K, a->N) > 0);
    TGP_REQUIRE(a->a_kt >= a->K / 16 && (a->ldc & 3) == 0 && a->ldc >= a->N);
    TGP_REQUIRE(!a->a_amax || (a->a && a->w && a->lda >= a->K && a->ldw >= a->K));      // the exact path reads the fp32 operands
    auto al16 = [](const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    TGP_REQUIRE(al16(a->a_planes) && al16(a->units) && al16(a->c) && al16(a->bias));
    HcParams p = {};
    p.a_pl = reinterpret_cast<const char *>(a->a_planes), p.a_kt = a->a_kt, p.a_amax = a->a_amax;
    p.units = reinterpret_cast<const char *>(a->units);
    p.bias2 = a->bias, p.c2 = a->c, p.ldc2 = a->ldc;
    p.a_f32 = a->a, p.lda = a->lda, p.w_f32 = a->w, p.ldw = a->ldw, p.K2 = a->K, p.ngt = a->N / 128;
    p.M = a->M, p.tiles = tgp_cdiv(a->M, 128);
    p.knob = 0;
#ifdef TGP_DEV
    p.knob = tgp_hs_chain_knobs;
#endif
    // one workgroup per CU and round.  A tile's groups are cut over as many workgroups as still fit one round of 256: in equal shares
    // (a divisor of the group count) over all tiles -- or, for the K = 512 products with a partial last tile (8224 rows = 64 full tiles
    // + 32 rows), over the FULL tiles in nearly equal shares (4 workgroups x 9 of the 36 groups = 256) with the partial tile's groups one
    // per workgroup behind the round: 124-133 against 138-140 us on the level-1 coarse product, same box; the K <= 256 projections lose
    // 1-3 us that way (their groups are short) and keep the first rule.  With more than 256 tiles the few past the last full round go
    // one group per workgroup.
    const int full_tiles = a->M / 128;
    p.nsplit = 1;
    p.main_tiles = p.tiles;
    if (a->K == 512 && full_tiles >= 1 && full_tiles < p.tiles && full_tiles <= 256) {
        p.nsplit = 256 / full_tiles < p.ngt ? 256 / full_tiles : p.ngt;
        p.main_tiles = full_tiles;
    } else {
        if (p.tiles <= 256)
            for (int d = 1; d <= p.ngt; ++d)
                if (p.ngt % d == 0 && p.tiles * d <= 256) p.nsplit = d;
        const int rt = 256 / p.nsplit, ov = p.tiles % rt;
        if (p.tiles > rt && ov > 0 && ov <= 8) p.main_tiles = p.tiles - ov;
    }
    while ((p.ngt + p.nsplit - 1) / p.nsplit > 24) ++p.nsplit;      // (a workgroup's bias slice lives in LDS behind the ring: <= 12 KB)
    const int grid = p.main_tiles * p.nsplit + (p.tiles - p.main_tiles) * p.ngt;
    const int lds = 4 * HR_SLOT + 128 * ((p.ngt + p.nsplit - 1) / p.nsplit) * 4;      // the ring + the bias of one workgroup's groups
    const int lds_max = 4 * HR_SLOT + 128 * 24 * 4;               // (the attribute is set once per device: the largest a launch can ask for)
    static TgpLdsAttr attr8, attr16, attr32;
    if (a->K == 512) {
        if (const int e = tgp_lds_attr(attr32, reinterpret_cast<const void *>(hs_proj_kernel<32>), lds_max)) return e;
        hipLaunchKernelGGL((hs_proj_kernel<32>), dim3(grid), dim3(256), lds, tgp_hs(stream), p);
    } else if (a->K == 128) {
        if (const int e = tgp_lds_attr(attr8, reinterpret_cast<const void *>(hs_proj_kernel<8>), lds_max)) return e;
        hipLaunchKernelGGL((hs_proj_kernel<8>), dim3(grid), dim3(256), lds, tgp_hs(stream), p);
    } else {
        if (const int e = tgp_lds_attr(attr16, reinterpret_cast<const void *>(hs_proj_kernel<16>), lds_max)) return e;
        hipLaunchKernelGGL((hs_proj_kernel<16>), dim3(grid), dim3(256), lds, tgp_hs(stream), p);
    }
    return TGP_LAUNCH_RESULT();
}
