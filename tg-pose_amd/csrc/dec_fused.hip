// The decoder behind its first conv -- conv 512 -> 512, conv 512 -> 256, conv 256 -> 128 (each + BatchNorm(eval) + ReLU), conv 128 -> 3
// and the un-sort of the rows (network/fs_net_repo/FaceRecon.py:105-117 Face_Dec.conv1d_block[3:] / recon_head, in eval mode) -- as ONE
// kernel in the style of heads_fused.hip: a wave owns 32 points for the whole chain and no activation between the layers leaves its
// registers.
//
//   layer 2 (512 -> 512):  acc2[out][point] += W2[out][k] . H1[point][k], K = 512 in 32 steps of 16; the points' operand fragments come
//            from the first conv's result as blocked fp16 planes (tgp_gemm_args.C_planes of that launch: 1 KB runs in lane order,
//            one 16-byte load per lane, step and plane), the weights' from LDS.  Products and order as gemm_pp_tile's (K ascending;
//            A_hi W_lo, A_lo W_hi, A_hi W_hi), operands' roles swapped: the sums are that kernel's, bit for bit;
//   epilogue in registers (+ bias, BatchNorm fold, ReLU, fp16 hi / lo split), block by block, IN PLACE: the accumulator layout gives
//            lane (point r, half h) the channels {4 h + (e & 3) + 8 (e >> 2)} of a 32-channel block -- exactly a 32x32x16 operand
//            fragment (8 k-values per lane) if the next layer's K order is permuted accordingly (heads_fused.hip's trick; the
//            permutation is applied to W3 / W4 when they are packed), so 16 accumulator registers become the 16 registers of two
//            K-steps' hi / lo fragments;
//   layer 3 (512 -> 256), layer 4 (256 -> 128) the same way; the last conv (128 -> 3) is 64 fmaf per lane and output + one exchange
//            between the two halves of the wave, written to the row the sort took the point from (tgp_rows_out's `map`).
//
// Every non-matrix instruction sits BETWEEN the dependent MFMAs (heads_fused.hip, round 5): with one wave per SIMD whatever follows a
// run of MFMAs into the same accumulators hides behind its last one only.  The weights arrive by LDS-DMA in units of 64 KB = 96 MFMAs
// per wave (layer 2: two K-steps x 16 output blocks; layer 3: four x 8; layer 4: eight x 4), fragment-blocked in memory so that a
// unit is one linear copy (tgp_dec_pack); double buffered.
//
// Against the four launches it replaces (three gemm_pp launches + tgp_rows_out) layer 2's sums are identical; layers 3 / 4 add the
// sixteen products of a K-step in another order (the permutation) and the last conv adds its 128 products per half wave first: results
// agree to rounding (tests/test_gpu_parity.py::test_decoder_chain_on_planes_only states the bar), not bit for bit.
#include "tgp_common.h"
#include "../../include/tgpose.h"

typedef _Float16 df16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 df16x4 __attribute__((ext_vector_type(4)));
typedef float df32x4 __attribute__((ext_vector_type(4)));
typedef float df32x16 __attribute__((ext_vector_type(16)));

#define DF_C1 512                          // channels of the operand (the first conv's output) and of layer 2
#define DF_C3 256
#define DF_C4 128
#define DF_UNIT (64 * 1024)                // one staging unit: 64 pieces of 1 KB = 32 (K-step, output block) pairs x 2 planes
#define DF_NDMA 16                         // LDS-DMA wave-instructions per wave and unit
// The 512 channels of layer 2 are produced and consumed in two halves (its sums would fill 256 registers per lane, with layer 3's 128
// beside them): half a = channels 0-255 (8 units of four K-steps x 8 output blocks), then the part of layer 3's sums that reads them
// (its K-steps 0-15: 4 units), half b the same, then layer 4 (2 units of eight K-steps x 4 output blocks).  Layer 3's K order is
// unchanged (ascending), so its sums are what one pass over K = 512 gives.
#define DF_U2H 8                           // units of one half of layer 2
#define DF_U3H 4                           // units of one half of layer 3's K range
#define DF_U4 2
#define DF_UNITS (2 * DF_U2H + 2 * DF_U3H + DF_U4)
#define DF_NVEC (3 * DF_C1 + 3 * DF_C3 + 3 * DF_C4 + 3 * DF_C4 + 4)      // bias | scale | shift per layer, W5 (3 x 128), b5 (3, padded)
#define DF_SB() __builtin_amdgcn_sched_barrier(0)

struct DecParams {
    const char *h1_pl; int h1_kt; const uint32_t *h1_amax;       // the operand: (M, 512) as blocked fp16 planes + its magnitude words
    const char *units;                                           // tgp_dec_pack's image: DF_UNITS x 64 KB
    const float *vec[3][3];                                      // [layer 2 / 3 / 4][bias | scale | shift]
    const float *w5, *b5;                                        // (3, 128), (3)
    const int64_t *map; int rows_per_obj;                        // out row of point i of object b: b * rows_per_obj + map[b * rows_per_obj + i]
    float *out;                                                  // (M, 3)
    int *flag;                                                   // fp16 range flag of the chain (raised, never cleared)
    const int *pred;                                             // (may be NULL) run only while *pred == 0 ... unused
    int M, tiles;
};

__device__ __forceinline__ float df_mix_lo(uint32_t hpair, float v)     // v - (float)(low half of hpair), one rounding
{
    float d;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(hpair), "v"(v));
    return d;
}
__device__ __forceinline__ float df_mix_hi(uint32_t hpair, float v)     // v - (float)(high half of hpair)
{
    float d;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(hpair), "v"(v));
    return d;
}

__global__ __launch_bounds__(256, 1) void dec_fused_kernel(DecParams p)
{
    extern __shared__ __attribute__((aligned(16))) char df_smem[];       // 2 x DF_UNIT, then DF_NVEC floats
    float *s_vec = reinterpret_cast<float *>(df_smem + 2 * DF_UNIT);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int tile = blockIdx.x;
    if (tile >= p.tiles) return;
    const int m0 = tile * 128 + wave * 32;                       // the wave's first point (may lie past M: then the wave only helps staging)
    const int nblk = (p.M + 31) >> 5;
    const int rb = min(m0 >> 5, nblk - 1);

    // ---- staging: unit u is 64 KB at units + 64 KB u; piece j = 4 j0 + wave is 1 KB at offset 1024 j of the unit and of the buffer
    const uint32_t voff0 = lane * 16 + wave * 1024;
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)df_smem) + wave * 1024;
    const char *u_src = p.units;                                 // scalar base of the unit being staged
    auto dma = [&](const int buf, const int j0) {
        const uint32_t lds = lds0 + buf * DF_UNIT + j0 * 4096;
        const uint32_t vo = voff0 + j0 * 4096;
        // inline assembly: opaque to the compiler's counters; vmcnt(0) is written by hand before the barrier that ends a unit
        asm volatile("s_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(vo), "s"(u_src), "{m0}"(lds) : "memory");
    };
#pragma unroll
    for (int j0 = 0; j0 < DF_NDMA; ++j0) dma(0, j0);
    // the layers' vectors and the last conv's weight into LDS
    for (int i = tid; i < DF_NVEC; i += 256) {
        float v;
        if (i < 3 * DF_C1) v = p.vec[0][i / DF_C1][i % DF_C1];
        else if (i < 3 * DF_C1 + 3 * DF_C3) v = p.vec[1][(i - 3 * DF_C1) / DF_C3][(i - 3 * DF_C1) % DF_C3];
        else if (i < 3 * DF_C1 + 3 * DF_C3 + 3 * DF_C4) v = p.vec[2][(i - 3 * DF_C1 - 3 * DF_C3) / DF_C4][(i - 3 * DF_C1 - 3 * DF_C3) % DF_C4];
        else if (i < DF_NVEC - 4) v = p.w5[i - (3 * DF_C1 + 3 * DF_C3 + 3 * DF_C4)];
        else v = i - (DF_NVEC - 4) < 3 ? p.b5[i - (DF_NVEC - 4)] : 0.f;
        s_vec[i] = v;
    }
    // fp16 range guard of the operand, from what its producer recorded (bits of max |a| per 32-row block): a block holding a magnitude
    // >= 65504 (or a NaN), or nothing at or above 2^-4, cannot be recomputed here (no fp32 operand): the chain's flag tells the caller's
    // predicated fp32 chain to redo the layers (as gemm_pp_tile does for a planes-only operand)
    if (p.h1_amax && m0 < p.M) {
        const uint32_t am = p.h1_amax[rb];
        if ((am >= 0x477fe000u || (am != 0u && am < 0x3d800000u)) && lane == 0) atomicOr(p.flag, 1);
    }
    // the wave's points as B fragments of layer 2: K-step kt, plane q at h1_pl + ((rb * kt_all + kt) * 2 + q) * 1024 + 16 lane
    const char *h1 = p.h1_pl + (int64_t)rb * p.h1_kt * 2048 + lane * 16;
    uint4 bcur[4][2], bnxt[4][2];                                // [K-step of the unit][plane]
    auto load_h1 = [&](const int unit, uint4 (&dst)[4][2], const int i) {      // i = 0 .. 7: (K-step, plane) of the unit
        dst[i >> 1][i & 1] = *reinterpret_cast<const uint4 *>(h1 + ((4 * unit + (i >> 1)) * 2 + (i & 1)) * 1024);
    };
#pragma unroll
    for (int i = 0; i < 8; ++i) load_h1(0, bcur, i);

    df32x16 acc2[8];                                             // one half of layer 2's sums, then (in place) its activations as fp16 hi / lo fragments
    df32x16 acc3[DF_C3 / 32];
    df32x16 acc4[DF_C4 / 32];
#pragma unroll
    for (int j = 0; j < DF_C3 / 32; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc3[j][e] = 0.f;

    // ---- a block's epilogue in slices of four vector instructions (one slice per MFMA gap): + bias, BatchNorm fold, ReLU, then hi = fp16(v),
    // lo = fp16(v - hi), written back INTO the accumulator's registers: elements 0-3 / 4-7 = hi fragments of the block's two K-steps,
    // 8-11 / 12-15 = lo.  Group g (four channels 4 h + 8 g .. + 3 of the block) takes slices 8 g .. 8 g + 7; a block is 32 slices.
    float4 e_b, e_sc, e_sh, e_v;
    uint32_t e_hi[8], e_lo[8];                                   // the block's packed halves: [2 g] = channels 0, 1 of group g, [2 g + 1] = 2, 3
    float e_l4[4];
    auto epi = [&](df32x16 &acc, const float *vec, const int C, const int blk, const int sl) {
        const int g = sl >> 3, ph = sl & 7;
        const float *pv = vec + 32 * blk + 4 * h + 8 * g;
        if (ph == 0) {                                            // (the vectors two gaps ahead of their first use)
            e_b = *reinterpret_cast<const float4 *>(pv), e_sc = *reinterpret_cast<const float4 *>(pv + C);
            e_sh = *reinterpret_cast<const float4 *>(pv + 2 * C);
        } else if (ph == 1) {
            e_v = make_float4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
        } else if (ph == 2) {
            e_v.x += e_b.x, e_v.y += e_b.y, e_v.z += e_b.z, e_v.w += e_b.w;
        } else if (ph == 3) {
            e_v.x *= e_sc.x, e_v.y *= e_sc.y, e_v.z *= e_sc.z, e_v.w *= e_sc.w;
        } else if (ph == 4) {
            e_v.x += e_sh.x, e_v.y += e_sh.y, e_v.z += e_sh.z, e_v.w += e_sh.w;
        } else if (ph == 5) {
            // ReLU as max(v, -0): what `v > 0 ? v : v * 0` gives for every number (a NaN becomes -0; non-finite sums are caught at the end)
            e_v.x = fmaxf(e_v.x, -0.f), e_v.y = fmaxf(e_v.y, -0.f), e_v.z = fmaxf(e_v.z, -0.f), e_v.w = fmaxf(e_v.w, -0.f);
        } else if (ph == 6) {
            const df32x4 x = {e_v.x, e_v.y, e_v.z, e_v.w};
            const uint2 hh = __builtin_bit_cast(uint2, __builtin_convertvector(x, df16x4));
            e_hi[2 * g] = hh.x, e_hi[2 * g + 1] = hh.y;
            e_l4[0] = df_mix_lo(hh.x, e_v.x), e_l4[1] = df_mix_hi(hh.x, e_v.y);
        } else {
            e_l4[2] = df_mix_lo(e_hi[2 * g + 1], e_v.z), e_l4[3] = df_mix_hi(e_hi[2 * g + 1], e_v.w);
            const df32x4 rest = {e_l4[0], e_l4[1], e_l4[2], e_l4[3]};
            const uint2 ll = __builtin_bit_cast(uint2, __builtin_convertvector(rest, df16x4));
            e_lo[2 * g] = ll.x, e_lo[2 * g + 1] = ll.y;
            if (g == 3) {                                        // the block's last slice: the packed halves go back into the accumulator
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __uint_as_float(e_hi[i]), acc[8 + i] = __uint_as_float(e_lo[i]);
            }
        }
    };
    auto frag = [&](const df32x16 &a, const int half, const int plane) {       // K-step `half` of a converted block, plane 0 = hi
        const int o = plane * 8 + half * 4;
        return __builtin_bit_cast(df16x8, make_uint4(__float_as_uint(a[o]), __float_as_uint(a[o + 1]), __float_as_uint(a[o + 2]), __float_as_uint(a[o + 3])));
    };
    const float *v2 = s_vec, *v3 = s_vec + 3 * DF_C1, *v4 = v3 + 3 * DF_C3, *w5 = v4 + 3 * DF_C4, *b5 = w5 + 3 * DF_C4;

    __builtin_amdgcn_s_waitcnt(0x0f70);                          // vmcnt(0): this wave's DMA has landed
    __syncthreads();

    // One staging unit = 32 (K-step s, output block j) pairs, q = s * NB + j, three MFMAs each (smallest terms first, as in the tile kernel:
    // W lo x H hi, W hi x H lo, W hi x H hi); gap 1: the weight fragments of pair q + 2 (+ FILL_A), gap 2: a DMA piece of the next unit
    // (+ FILL_B), gap 3: FILL_C.  BH / BL: the points' hi / lo fragments of K-step s.
#define DF_BODY(GU, ACC, NB, BH, BL, STAGE, FILL_A, FILL_B, FILL_C)                                                          \
    {                                                                                                                        \
        const char *wrow = df_smem + ((GU) & 1) * DF_UNIT + lane * 16;                                                       \
        u_src = p.units + (int64_t)((GU) + 1) * DF_UNIT;                                                                     \
        auto wfrag = [&](int q, int plane) { return *reinterpret_cast<const uint4 *>(wrow + (q * 2 + plane) * 1024); };      \
        uint4 wh0 = wfrag(0, 0), wl0 = wfrag(0, 1), wh1 = wfrag(1, 0), wl1 = wfrag(1, 1);                                    \
        DF_SB();                                                                                                             \
        _Pragma("unroll") for (int q = 0; q < 32; ++q) {                                                                     \
            const int s = q / (NB), j = q % (NB);                                                                            \
            uint4 wh2 = wh1, wl2 = wl1;                                                                                      \
            const df16x8 bh = BH, bl = BL;                                                                                   \
            ACC[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(df16x8, wl0), bh, ACC[j], 0, 0, 0);           \
            DF_SB();                                                                                                         \
            if (q + 2 < 32) wh2 = wfrag(q + 2, 0), wl2 = wfrag(q + 2, 1);                                                    \
            FILL_A;                                                                                                          \
            DF_SB();                                                                                                         \
            ACC[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(df16x8, wh0), bl, ACC[j], 0, 0, 0);           \
            DF_SB();                                                                                                         \
            if ((STAGE) && q < DF_NDMA) dma(((GU) & 1) ^ 1, q);                                                              \
            FILL_B;                                                                                                          \
            DF_SB();                                                                                                         \
            ACC[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(df16x8, wh0), bh, ACC[j], 0, 0, 0);           \
            DF_SB();                                                                                                         \
            FILL_C;                                                                                                          \
            DF_SB();                                                                                                         \
            wh0 = wh1, wl0 = wl1, wh1 = wh2, wl1 = wl2;                                                                      \
        }                                                                                                                    \
        __builtin_amdgcn_s_waitcnt(0x0f70);     /* vmcnt(0): this wave's share of the next unit has landed (and its fragments) */ \
        __syncthreads();                        /* ... everybody's has, and this buffer's readers are done */                \
    }
    // layer 3, unit U of a half (HALF = 0 / 1): K-steps 4 U .. 4 U + 3 of the half = converted blocks 2 U, 2 U + 1 of acc2; in its gaps the
    // epilogue of blocks 2 U + 2, 2 U + 3 (two slices per pair); its last unit loads the second half's first operand fragments
#define DF_L3(HALF, U)                                                                                                       \
    DF_BODY(DF_U2H + (HALF) * (DF_U2H + DF_U3H) + (U), acc3, 8, frag(acc2[2 * (U) + (s >> 1)], s & 1, 0),                   \
            frag(acc2[2 * (U) + (s >> 1)], s & 1, 1), true,                                                                  \
            if ((U) + 1 < DF_U3H) epi(acc2[2 * (U) + 2 + (q >> 4)], v2 + 256 * (HALF), DF_C1, 2 * (U) + 2 + (q >> 4), 2 * (q & 15)), \
            if ((U) + 1 < DF_U3H) epi(acc2[2 * (U) + 2 + (q >> 4)], v2 + 256 * (HALF), DF_C1, 2 * (U) + 2 + (q >> 4), 2 * (q & 15) + 1), \
            if ((HALF) == 0 && (U) + 1 == DF_U3H && q < 8) load_h1(0, bcur, q))

#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
        // ============================================================= layer 2, one half: 8 units of four K-steps x 8 output blocks
#pragma unroll
        for (int ob = 0; ob < 8; ++ob)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc2[ob][e] = 0.f;
        const int g0 = half * (DF_U2H + DF_U3H);
#pragma unroll 1
        for (int u = 0; u < DF_U2H; ++u) {
            DF_BODY(g0 + u, acc2, 8, __builtin_bit_cast(df16x8, bcur[s][0]), __builtin_bit_cast(df16x8, bcur[s][1]), true, ,
                    , if (q >= 16 && q < 24 && u + 1 < DF_U2H) load_h1(u + 1, bnxt, q - 16))
#pragma unroll
            for (int i = 0; i < 8; ++i) bcur[i >> 1][i & 1] = bnxt[i >> 1][i & 1];
        }
        // its first two blocks become fragments before layer 3 reads them; the others inside layer 3's units
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int sl = 0; sl < 32; ++sl) epi(acc2[blk], v2 + 256 * half, DF_C1, blk, sl);
        // ============================================================= layer 3, K-steps of this half: 4 units of four K-steps x 8 output blocks
        if (half == 0) {
            DF_L3(0, 0) DF_L3(0, 1) DF_L3(0, 2) DF_L3(0, 3)
        } else {
            DF_L3(1, 0) DF_L3(1, 1) DF_L3(1, 2) DF_L3(1, 3)
        }
    }
#undef DF_L3
    // layer 3's first four blocks become fragments before layer 4 starts (its first unit reads them); the other four inside that unit
#pragma unroll
    for (int blk = 0; blk < 4; ++blk)
#pragma unroll
        for (int sl = 0; sl < 32; ++sl) epi(acc3[blk], v3, DF_C3, blk, sl);
#pragma unroll
    for (int j = 0; j < DF_C4 / 32; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc4[j][e] = 0.f;
    // ================================================================= layer 4: 2 units of eight K-steps (= four blocks of layer 3) x 4 output blocks
    DF_BODY(2 * DF_U2H + 2 * DF_U3H, acc4, 4, frag(acc3[s >> 1], s & 1, 0), frag(acc3[s >> 1], s & 1, 1), true,
            epi(acc3[4 + (q >> 3)], v3, DF_C3, 4 + (q >> 3), 4 * (q & 7)),
            epi(acc3[4 + (q >> 3)], v3, DF_C3, 4 + (q >> 3), 4 * (q & 7) + 1),
            { epi(acc3[4 + (q >> 3)], v3, DF_C3, 4 + (q >> 3), 4 * (q & 7) + 2); epi(acc3[4 + (q >> 3)], v3, DF_C3, 4 + (q >> 3), 4 * (q & 7) + 3); })
    DF_BODY(2 * DF_U2H + 2 * DF_U3H + 1, acc4, 4, frag(acc3[4 + (s >> 1)], s & 1, 0), frag(acc3[4 + (s >> 1)], s & 1, 1), false, , , )
#undef DF_BODY

    // ================================================================= layer 4's epilogue and the last conv (128 -> 3)
    // lane (point r, half h) holds channels 32 j + 4 h + (e & 3) + 8 (e >> 2) of block j
    if (m0 >= p.M) return;
    float o3[3] = {0.f, 0.f, 0.f};
    bool finite = true;
#pragma unroll
    for (int j = 0; j < DF_C4 / 32; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c = 32 * j + 4 * h + 8 * g;
            const float4 b = *reinterpret_cast<const float4 *>(v4 + c), sc = *reinterpret_cast<const float4 *>(v4 + DF_C4 + c);
            const float4 sh = *reinterpret_cast<const float4 *>(v4 + 2 * DF_C4 + c);
            float x[4] = {acc4[j][4 * g], acc4[j][4 * g + 1], acc4[j][4 * g + 2], acc4[j][4 * g + 3]};
            const float bb[4] = {b.x, b.y, b.z, b.w}, ss[4] = {sc.x, sc.y, sc.z, sc.w}, hh[4] = {sh.x, sh.y, sh.z, sh.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float v = x[i] + bb[i];
                v = v * ss[i] + hh[i];
                v = v > 0.f ? v : v * 0.f;
                finite &= m0 + r >= p.M || (v == v && v < __builtin_inff());       // (rows past the end were never written)
                x[i] = v;
            }
#pragma unroll
            for (int o = 0; o < 3; ++o) {
                const float4 w = *reinterpret_cast<const float4 *>(w5 + o * DF_C4 + c);
                o3[o] = fmaf(x[3], w.w, fmaf(x[2], w.z, fmaf(x[1], w.y, fmaf(x[0], w.x, o3[o]))));
            }
        }
    // a magnitude beyond fp16's range (or a NaN) anywhere in the chain makes the point's sums non-finite: the predicated fp32 chain redoes it
    if (__ballot(!finite) != 0ull && lane == 0) atomicOr(p.flag, 1);
#pragma unroll
    for (int o = 0; o < 3; ++o) o3[o] += __shfl_xor(o3[o], 32, 64);
    const int row = m0 + r;
    if (h == 0 && row < p.M) {
        const int64_t dst = p.map ? (int64_t)(row / p.rows_per_obj) * p.rows_per_obj + p.map[row] : row;
        p.out[dst * 3 + 0] = o3[0] + b5[0], p.out[dst * 3 + 1] = o3[1] + b5[1], p.out[dst * 3 + 2] = o3[2] + b5[2];
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// The decoder's FIRST conv (FaceRecon.py:112 conv1d_block[0..2] on the factored form of this repo's engine:
// conv(feat)[m] = W_fine . fine[m] + P1[idx1[m]] + P2[idx2[m]] + per-object bias, then BatchNorm(eval) + ReLU) as the conv1 half of
// heads_fused.hip's kernel: a wave owns 32 points, their fine features stay in registers as B fragments, the weights' 32-channel
// blocks arrive by linear LDS-DMA, and block i's 51 MFMAs carry block i - 1's epilogue, this block's gathers and block i - 2's
// stores in their gaps.  The activation leaves as the operand dec_fused_kernel loads: fp16 hi / lo fragments in the order the
// accumulators hold the channels (slot 8 h + t of K-tile 2 b + s2 = channel 32 b + 16 s2 + 8 (t >> 2) + 4 h + (t & 3)): a lane stores the
// 16 bytes it will later load, so the tile GEMM's LDS transposition (and its 4 x 2056 workgroups) is not needed; W2's K order is
// permuted to match when it is packed.  Per 32-row block the largest activation goes to the magnitude words (the consumer's guard).
#define DL_NCB (DF_C1 / 32)
#define DL_STEPS 17                        // K steps over the fine buffer: 272 / 16
#define DL_APIECES (2 * DL_STEPS)
#define DL_ABUF (DL_APIECES * 1024)
#define DL_NDMA 9                          // piece j = 4 j0 + wave < 34

struct DecL1Params {
    const char *fine_pl; int fine_kt; const uint32_t *fine_amax;
    const char *wa_pl;                     // (512, 272) as blocked planes, 17 K-tiles per 32-channel block
    const float *p1; int ldp1; const int32_t *idx1;
    const float *p2; int ldp2; const int32_t *idx2;
    const float *bias, *scale, *shift;     // 512
    const float *rowbias; int ldrb, rows_per_obj;      // (may be NULL) per-object bias (B, 512)
    char *h1_pl; int h1_kt; uint32_t *h1_amax;          // out: the activation as planes in accumulator order, its magnitude words
    int *flag;
    int knob;                              // (development build) timing-only variants: 1 = no stores, 2 = no gathers
    int M, tiles;
    int main_tiles;                        // tiles [0, main_tiles) are one workgroup each; the others one workgroup per 32-channel block
};

struct DlG {                               // a block's gathered terms of the lane's four channel groups
    float4 g1[4], g2[4], rb[4];
};

__global__ __launch_bounds__(256, 1) void dec_l1_kernel(DecL1Params p)
{
    extern __shared__ __attribute__((aligned(16))) char dl_smem[];       // 2 x DL_ABUF, then 3 x 512 floats
    float *s_vec = reinterpret_cast<float *>(dl_smem + 2 * DL_ABUF);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    // The benchmark's 32896 points are 257 tiles of 128 on 256 CUs: the 257th would run alone for a whole second round.  Tiles past the
    // last full round are cut into their 16 channel blocks instead -- each block's result is independent of the others, so the same
    // bits -- many short workgroups that drain in a fraction of a round (conv_max_fused_kernel's rule).
    int tile = blockIdx.x, single = -1;
    if (tile >= p.main_tiles) {
        const int rest = tile - p.main_tiles;
        tile = p.main_tiles + rest / DL_NCB, single = rest % DL_NCB;
    }
    if (tile >= p.tiles) return;
    const int m0 = tile * 128 + wave * 32;
    const int row = min(m0 + r, p.M - 1);
    const int nblk = (p.M + 31) >> 5;
    const int rb = min(m0 >> 5, nblk - 1);

    const uint32_t voff0 = lane * 16 + wave * 1024;
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)dl_smem) + wave * 1024;
    const char *u_src = p.wa_pl + (int64_t)(single < 0 ? 0 : single) * DL_ABUF;
    auto dma = [&](const int buf, const int j0) {
        if (j0 * 4 + 3 >= DL_APIECES && wave >= (DL_APIECES & 3)) return;      // pieces 34, 35 do not exist
        const uint32_t lds = lds0 + buf * DL_ABUF + j0 * 4096;
        const uint32_t vo = voff0 + j0 * 4096;
        asm volatile("s_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(vo), "s"(u_src), "{m0}"(lds) : "memory");
    };
#pragma unroll
    for (int j0 = 0; j0 < DL_NDMA; ++j0) dma(0, j0);
    for (int i = tid; i < 3 * DF_C1; i += 256) s_vec[i] = (i < DF_C1 ? p.bias : i < 2 * DF_C1 ? p.scale : p.shift)[i % DF_C1];
    // the wave's points as B fragments (as heads_fused_kernel)
    uint4 bh[DL_STEPS], bl[DL_STEPS];
    {
        const char *src = p.fine_pl + (int64_t)rb * p.fine_kt * 2048 + lane * 16;
#pragma unroll
        for (int s = 0; s < DL_STEPS; ++s) {
            bh[s] = *reinterpret_cast<const uint4 *>(src + s * 2048);
            bl[s] = *reinterpret_cast<const uint4 *>(src + s * 2048 + 1024);
        }
        if (m0 + 32 > p.M) {
            const bool dead = m0 + r >= p.M;
#pragma unroll
            for (int s = 0; s < DL_STEPS; ++s)
                if (dead) bh[s] = make_uint4(0u, 0u, 0u, 0u), bl[s] = make_uint4(0u, 0u, 0u, 0u);
        }
        // the input side of the fp16 range guard, as in the tile kernels: a block that cannot be split faithfully raises the chain's flag
        if (p.fine_amax && m0 < p.M) {
            const uint32_t am = p.fine_amax[rb];
            if ((am >= 0x477fe000u || (am != 0u && am < 0x3d800000u)) && lane == 0) atomicOr(p.flag, 1);
        }
    }
    const int i1 = p.idx1[row], i2 = p.idx2[row];
    const float *g1p = p.p1 + (int64_t)i1 * p.ldp1 + 4 * h;               // + cb * 32 + 8 m: four channels of the lane
    const float *g2p = p.p2 + (int64_t)i2 * p.ldp2 + 4 * h;
    const float *rbp = p.rowbias ? p.rowbias + (int64_t)(row / p.rows_per_obj) * p.ldrb + 4 * h : nullptr;
    char *h1 = p.h1_pl + (int64_t)rb * p.h1_kt * 2048 + lane * 16;          // + ((2 cb + s2) * 2 + plane) * 1024

    df32x16 acc[2];
    DlG g[2];
    uint32_t a2[2][16];                    // a block's activations as packed fp16: [0..7] hi (K-tile 2 b: 0..3, 2 b + 1: 4..7), [8..15] lo
    float amax = 0.f;
    auto gather1 = [&](DlG &d, const int cb, const int i) {              // i = 0 .. 11: one 16-byte load
        const int m = i & 3, which = i >> 2;
        if (p.knob & 2) return;
        if (which == 0) d.g1[m] = *reinterpret_cast<const float4 *>(g1p + cb * 32 + 8 * m);
        else if (which == 1) d.g2[m] = *reinterpret_cast<const float4 *>(g2p + cb * 32 + 8 * m);
        else d.rb[m] = rbp ? *reinterpret_cast<const float4 *>(rbp + cb * 32 + 8 * m) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto store1 = [&](const uint32_t (&a)[16], const int cb, const int i) {   // i = 0 .. 3: (K-tile of the block, plane)
        const int s2 = i >> 1, plane = i & 1;
        if (m0 >= p.M || (p.knob & 1)) return;
        *reinterpret_cast<uint4 *>(h1 + ((2 * cb + s2) * 2 + plane) * 1024) =
            make_uint4(a[plane * 8 + s2 * 4], a[plane * 8 + s2 * 4 + 1], a[plane * 8 + s2 * 4 + 2], a[plane * 8 + s2 * 4 + 3]);
    };
    // epilogue of a block in 32 slices of a few vector instructions each; group m = four channels 4 h + 8 m .. + 3 takes slices 8 m .. 8 m + 7
    // (its vectors are read from LDS four slices earlier); in the tile kernel's order: + bias, + P1 row, + P2 row, + per-object bias,
    // BatchNorm fold, ReLU
    float4 e_b, e_sc, e_sh, e_bn, e_scn, e_shn, e_v;
    float e_l4[4];
    auto epi_read = [&](const int cb, const int m) {
        const float *pv = s_vec + 32 * cb + 4 * h + 8 * m;
        e_bn = *reinterpret_cast<const float4 *>(pv), e_scn = *reinterpret_cast<const float4 *>(pv + DF_C1);
        e_shn = *reinterpret_cast<const float4 *>(pv + 2 * DF_C1);
    };
    auto epi = [&](const df32x16 &ac, const DlG &d, uint32_t (&a)[16], const int cb, const int sl) {
        const int m = sl >> 3, ph = sl & 7;
        if (ph == 0) {
            e_b = e_bn, e_sc = e_scn, e_sh = e_shn;
            e_v = make_float4(ac[4 * m], ac[4 * m + 1], ac[4 * m + 2], ac[4 * m + 3]);
        } else if (ph == 1) {
            e_v.x += e_b.x, e_v.y += e_b.y, e_v.z += e_b.z, e_v.w += e_b.w;
        } else if (ph == 2) {
            e_v.x += d.g1[m].x, e_v.y += d.g1[m].y, e_v.z += d.g1[m].z, e_v.w += d.g1[m].w;
        } else if (ph == 3) {
            e_v.x += d.g2[m].x, e_v.y += d.g2[m].y, e_v.z += d.g2[m].z, e_v.w += d.g2[m].w;
            if (m + 1 < 4) epi_read(cb, m + 1);
        } else if (ph == 4) {
            e_v.x += d.rb[m].x, e_v.y += d.rb[m].y, e_v.z += d.rb[m].z, e_v.w += d.rb[m].w;
        } else if (ph == 5) {
            e_v.x *= e_sc.x, e_v.y *= e_sc.y, e_v.z *= e_sc.z, e_v.w *= e_sc.w;
            e_v.x += e_sh.x, e_v.y += e_sh.y;
        } else if (ph == 6) {
            e_v.z += e_sh.z, e_v.w += e_sh.w;
            e_v.x = fmaxf(e_v.x, -0.f), e_v.y = fmaxf(e_v.y, -0.f), e_v.z = fmaxf(e_v.z, -0.f), e_v.w = fmaxf(e_v.w, -0.f);   // ReLU as max(v, -0)
            amax = fmaxf(amax, fmaxf(fmaxf(e_v.x, e_v.y), fmaxf(e_v.z, e_v.w)));
        } else {
            const df32x4 x = {e_v.x, e_v.y, e_v.z, e_v.w};
            const uint2 hh = __builtin_bit_cast(uint2, __builtin_convertvector(x, df16x4));
            e_l4[0] = df_mix_lo(hh.x, e_v.x), e_l4[1] = df_mix_hi(hh.x, e_v.y), e_l4[2] = df_mix_lo(hh.y, e_v.z), e_l4[3] = df_mix_hi(hh.y, e_v.w);
            const df32x4 rest = {e_l4[0], e_l4[1], e_l4[2], e_l4[3]};
            const uint2 ll = __builtin_bit_cast(uint2, __builtin_convertvector(rest, df16x4));
            a[2 * m] = hh.x, a[2 * m + 1] = hh.y, a[8 + 2 * m] = ll.x, a[8 + 2 * m + 1] = ll.y;
        }
    };

    // the weights of block 0 and the vectors (LDS) are in when at most the 34 fragment loads issued after them (and what followed) fly:
    // block 0's conv1 starts on the fragments that have arrived (heads_fused_kernel's prologue)
    __builtin_amdgcn_s_waitcnt(0x0070 | (34 & 15) | ((34 >> 4) << 14));      // vmcnt(34) lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
    // iteration cb (0 .. 17): conv1 of block cb (cb < 16) into acc[cb & 1]; in its gaps the weights of block cb + 1, the epilogue of
    // block cb - 1 (from acc[(cb - 1) & 1], g[(cb - 1) & 1] into a2[(cb - 1) & 1]), the gathers of block cb (into g[cb & 1]) and the
    // stores of block cb - 2 (from a2[cb & 1]).  MM / E1 / ST: which of the three this instance contains (compile-time).
#define DL_ITER(CB, P, MM, E1, ST, DM)                                                                                       \
    {                                                                                                                        \
        const int cb = (CB);                                                                                                 \
        const char *arow = dl_smem + (P) * DL_ABUF + lane * 16;                                                              \
        u_src = p.wa_pl + (int64_t)(cb + 1 < DL_NCB ? cb + 1 : DL_NCB - 1) * DL_ABUF;                                        \
        df32x16 &ax = acc[(P)];                                                                                              \
        if (MM) { _Pragma("unroll") for (int e = 0; e < 16; ++e) ax[e] = 0.f; }                                              \
        uint4 fh0 = *reinterpret_cast<const uint4 *>(arow), fl0 = *reinterpret_cast<const uint4 *>(arow + 1024);             \
        uint4 fh1 = *reinterpret_cast<const uint4 *>(arow + 2048), fl1 = *reinterpret_cast<const uint4 *>(arow + 3072);      \
        if (E1) epi_read(cb - 1, 0);                                                                                         \
        DF_SB();                                                                                                             \
        _Pragma("unroll") for (int s = 0; s < DL_STEPS; ++s) {                                                               \
            uint4 fh2 = fh1, fl2 = fl1;                                                                                      \
            if (MM) ax = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(df16x8, fl0), __builtin_bit_cast(df16x8, bh[s]), ax, 0, 0, 0); \
            DF_SB();                                                                                                         \
            if (s + 2 < DL_STEPS && (MM)) {                                                                                  \
                fh2 = *reinterpret_cast<const uint4 *>(arow + (s + 2) * 2048);                                               \
                fl2 = *reinterpret_cast<const uint4 *>(arow + (s + 2) * 2048 + 1024);                                        \
            }                                                                                                                \
            if ((MM) && (DM) && s < DL_NDMA) dma((P) ^ 1, s);                                                                \
            if ((MM) && s >= DL_NDMA) {                                                                                      \
                _Pragma("unroll") for (int t2 = 0; t2 < 2; ++t2)                                                             \
                    if (2 * (s - DL_NDMA) + t2 < 12) gather1(g[(P)], cb, 2 * (s - DL_NDMA) + t2);                            \
            }                                                                                                                \
            DF_SB();                                                                                                         \
            if (MM) ax = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(df16x8, fh0), __builtin_bit_cast(df16x8, bl[s]), ax, 0, 0, 0); \
            DF_SB();                                                                                                         \
            if ((E1) && 2 * s < 32) epi(acc[(P) ^ 1], g[(P) ^ 1], a2[(P) ^ 1], cb - 1, 2 * s);                               \
            if ((ST) && s >= 2 && s < 6) store1(a2[(P)], cb - 2, s - 2);                                                     \
            DF_SB();                                                                                                         \
            if (MM) ax = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(df16x8, fh0), __builtin_bit_cast(df16x8, bh[s]), ax, 0, 0, 0); \
            DF_SB();                                                                                                         \
            if ((E1) && 2 * s + 1 < 32) epi(acc[(P) ^ 1], g[(P) ^ 1], a2[(P) ^ 1], cb - 1, 2 * s + 1);                       \
            DF_SB();                                                                                                         \
            fh0 = fh1, fl0 = fl1, fh1 = fh2, fl1 = fl2;                                                                      \
        }                                                                                                                    \
        /* the next block's weights have landed; this block's twelve gathers -- issued after the last DMA piece, so the twelve  \
           youngest operations in flight -- may still be on their way (their first use, in the next iteration, waits for them) */ \
        if (MM) __builtin_amdgcn_s_waitcnt(0x0f70 | 12); else __builtin_amdgcn_s_waitcnt(0x0f70);                             \
        __syncthreads();                                                                                                     \
    }
    if (single >= 0) {                                            // one channel block of a tile behind the last full round
        DL_ITER(single, 0, true, false, false, false)
        DL_ITER(single + 1, 1, false, true, false, false)
        DL_ITER(single + 2, 0, false, false, true, false)
    } else {
        DL_ITER(0, 0, true, false, false, true)
        DL_ITER(1, 1, true, true, false, true)
#pragma unroll 1
        for (int c2 = 2; c2 < DL_NCB; c2 += 2) {
            DL_ITER(c2, 0, true, true, true, true)
            DL_ITER(c2 + 1, 1, true, true, true, true)
        }
        DL_ITER(DL_NCB, 0, false, true, true, true)
        DL_ITER(DL_NCB + 1, 1, false, false, true, true)
    }
#undef DL_ITER
    // the block's largest activation for the consumer's range guard (bits of a non-negative float order as the floats do)
    if (m0 < p.M && p.h1_amax) {
        float m = amax;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        if (lane == 0 && m > 0.f) atomicMax(p.h1_amax + rb, __float_as_uint(m));
    }
}

#ifdef TGP_DEV
static int tgp_dec_l1_knobs = 0;
extern "C" int tgp_debug_set_dec_l1_knobs(int v) { tgp_dec_l1_knobs = v; return 0; }
#endif

extern "C" int tgp_dec_l1(const tgp_dec_l1_args *a, tgp_stream_t stream)
{
    TGP_REQUIRE(a && a->fine_planes && a->wa_planes && a->p1 && a->p2 && a->idx1 && a->idx2 && a->bias && a->scale && a->shift &&
                a->h1_planes && a->flag && a->M > 0);
    TGP_REQUIRE(a->fine_kt >= DL_STEPS && a->h1_kt >= DF_C1 / 16 && (a->ldp1 & 3) == 0 && (a->ldp2 & 3) == 0 && a->ldp1 >= DF_C1 && a->ldp2 >= DF_C1);
    TGP_REQUIRE(!a->rowbias || (a->rows_per_obj > 0 && a->M % a->rows_per_obj == 0 && (a->ldrb & 3) == 0 && a->ldrb >= DF_C1));
    auto al16 = [](const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    TGP_REQUIRE(al16(a->fine_planes) && al16(a->wa_planes) && al16(a->p1) && al16(a->p2) && al16(a->h1_planes) && al16(a->rowbias));
    DecL1Params p;
    p.fine_pl = reinterpret_cast<const char *>(a->fine_planes), p.fine_kt = a->fine_kt, p.fine_amax = a->fine_amax;
    p.wa_pl = reinterpret_cast<const char *>(a->wa_planes);
    p.p1 = a->p1, p.ldp1 = a->ldp1, p.idx1 = a->idx1, p.p2 = a->p2, p.ldp2 = a->ldp2, p.idx2 = a->idx2;
    p.bias = a->bias, p.scale = a->scale, p.shift = a->shift;
    p.rowbias = a->rowbias, p.ldrb = a->ldrb, p.rows_per_obj = a->rows_per_obj > 0 ? a->rows_per_obj : a->M;
    p.h1_pl = reinterpret_cast<char *>(a->h1_planes), p.h1_kt = a->h1_kt, p.h1_amax = a->h1_amax;
    p.flag = a->flag, p.M = a->M, p.tiles = tgp_cdiv(a->M, 128);
    p.knob = 0;
#ifdef TGP_DEV
    p.knob = tgp_dec_l1_knobs;
#endif
    // a few tiles past a whole number of rounds (one workgroup per CU) go in single channel blocks
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
            cus = 256;
    }
    const int over = p.tiles % cus;
    p.main_tiles = (p.tiles > cus && over > 0 && over <= 8) ? p.tiles - over : p.tiles;
    const int lds = 2 * DL_ABUF + 3 * DF_C1 * 4;
    static TgpLdsAttr attr;
    if (const int e = tgp_lds_attr(attr, reinterpret_cast<const void *>(dec_l1_kernel), lds)) return e;
    hipLaunchKernelGGL(dec_l1_kernel, dim3(p.main_tiles + (p.tiles - p.main_tiles) * DL_NCB), dim3(256), lds, tgp_hs(stream), p);
    return TGP_LAUNCH_RESULT();
}

// ---- weights -> staging units.  W (N, K) fp32 row-major; a unit is 64 pieces of 1 KB: piece (q * 2 + plane), q = s * (N / 32) + j over the
// unit's K-steps s and the N / 32 output blocks j, holds [lane = 32 h + r][8 fp16] = W[32 j + r][k(step, 8 h + t)], t = 0 .. 7, as its
// fp16 hi (plane 0) / lo (plane 1) part.  permuted = 0: k = 16 step + slot (layer 2: the operand planes' natural order); 1: slot
// 8 h + t of step 2 b + s2 is channel 32 b + 16 s2 + 8 (t >> 2) + 4 h + (t & 3) -- the order in which the previous layer's accumulators hold
// a block's channels.
__global__ void dec_pack_kernel(const float *__restrict__ W, int ld, int N, int K, int permuted, uint16_t *__restrict__ out)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)N * K;
    if (t >= total) return;
    const int nb = N / 32, spu = 32 / nb;                         // K-steps per unit
    const int tt = (int)(t & 7), rr = (int)((t >> 3) & 31), hh = (int)((t >> 8) & 1);
    const int q = (int)((t >> 9) & 31);
    const int unit = (int)(t >> 14);
    const int s = q / nb, j = q % nb;
    const int step = unit * spu + s;
    const int col = permuted ? 32 * (step >> 1) + 16 * (step & 1) + 8 * (tt >> 2) + 4 * hh + (tt & 3) : 16 * step + 8 * hh + tt;
    const float v = W[(int64_t)(32 * j + rr) * ld + col];
    const _Float16 hi = (_Float16)v;
    const _Float16 lo = (_Float16)(v - (float)hi);
    uint16_t *dst = out + (int64_t)unit * (DF_UNIT / 2) + (int64_t)(q * 2) * 512 + (hh * 32 + rr) * 8 + tt;
    dst[0] = __builtin_bit_cast(uint16_t, hi);
    dst[512] = __builtin_bit_cast(uint16_t, lo);
}

extern "C" int64_t tgp_dec_pack_bytes(void) { return (int64_t)DF_UNITS * DF_UNIT; }

extern "C" int tgp_dec_pack(const float *w2, const float *w3, const float *w4, int h1_permuted, void *out, tgp_stream_t stream)
{
    TGP_REQUIRE(w2 && w3 && w4 && out && (reinterpret_cast<uintptr_t>(out) & 15) == 0);
    uint16_t *o = reinterpret_cast<uint16_t *>(out);
    const int64_t U = DF_UNIT / 2;
    // staging order: layer 2 rows 0-255 | layer 3 columns 0-255 | layer 2 rows 256-511 | layer 3 columns 256-511 | layer 4
    for (int half = 0; half < 2; ++half) {
        uint16_t *base = o + (int64_t)half * (DF_U2H + DF_U3H) * U;
        hipLaunchKernelGGL(dec_pack_kernel, dim3(256 * DF_C1 / 256), dim3(256), 0, tgp_hs(stream), w2 + (int64_t)half * 256 * DF_C1, DF_C1, 256,
                           DF_C1, h1_permuted ? 1 : 0, base);
        hipLaunchKernelGGL(dec_pack_kernel, dim3(DF_C3 * 256 / 256), dim3(256), 0, tgp_hs(stream), w3 + half * 256, DF_C1, DF_C3, 256, 1,
                           base + (int64_t)DF_U2H * U);
    }
    hipLaunchKernelGGL(dec_pack_kernel, dim3(DF_C4 * DF_C3 / 256), dim3(256), 0, tgp_hs(stream), w4, DF_C3, DF_C4, DF_C3, 1,
                       o + (int64_t)(2 * DF_U2H + 2 * DF_U3H) * U);
    return TGP_LAUNCH_RESULT();
}

extern "C" int tgp_dec_fused(const tgp_dec_fused_args *a, tgp_stream_t stream)
{
    TGP_REQUIRE(a && a->h1_planes && a->units && a->w5 && a->b5 && a->out && a->flag && a->M > 0 && a->h1_kt >= DF_C1 / 16);
    for (int l = 0; l < 3; ++l)
        for (int v = 0; v < 3; ++v) TGP_REQUIRE(a->vec[l][v] != nullptr);
    TGP_REQUIRE((reinterpret_cast<uintptr_t>(a->h1_planes) & 15) == 0 && (reinterpret_cast<uintptr_t>(a->units) & 15) == 0);
    TGP_REQUIRE(!a->map || (a->rows_per_obj > 0 && a->M % a->rows_per_obj == 0));
    DecParams p;
    p.h1_pl = reinterpret_cast<const char *>(a->h1_planes), p.h1_kt = a->h1_kt, p.h1_amax = a->h1_amax;
    p.units = reinterpret_cast<const char *>(a->units);
    for (int l = 0; l < 3; ++l)
        for (int v = 0; v < 3; ++v) p.vec[l][v] = a->vec[l][v];
    p.w5 = a->w5, p.b5 = a->b5, p.map = a->map, p.rows_per_obj = a->rows_per_obj, p.out = a->out, p.flag = a->flag, p.pred = nullptr;
    p.M = a->M, p.tiles = tgp_cdiv(a->M, 128);
    const int lds = 2 * DF_UNIT + DF_NVEC * 4;
    static TgpLdsAttr attr;
    if (const int e = tgp_lds_attr(attr, reinterpret_cast<const void *>(dec_fused_kernel), lds)) return e;
    hipLaunchKernelGGL(dec_fused_kernel, dim3(p.tiles), dim3(256), lds, tgp_hs(stream), p);
    return TGP_LAUNCH_RESULT();
}
