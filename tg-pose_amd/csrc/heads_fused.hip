// The three pose heads' conv1 -> BatchNorm -> ReLU -> conv2 -> BatchNorm -> ReLU -> max over points as ONE kernel
// (network/fs_net_repo/PoseR.py:26-36, PoseTs.py:31-42 in eval mode, on the factored form of conv1: DESIGN.md section 3).
//
// As two launches the path writes and re-reads the (B*N, 3 x 1024) activation (404 MB at B = 32) and spends most of its time in
// the first launch's gather epilogue.  Here the activation never leaves the CU.  A wave owns 32 points of one head:
//   conv1, per block of 32 channels:  acc1[channel][point] = W_fine[channel][:] . fine[point][:]   (K = 268, 17 steps of
//            v_mfma_f32_32x32x16_f16 x 3 split terms; the weights' fp16 planes come from LDS, the points' from registers);
//   epilogue 1 in registers: + bias + P1[near1(point)] + P2[near2(point)] (the coarse products of the factored layer), BatchNorm
//            fold, ReLU -- in the two-launch form's order -- then split into fp16 hi / lo;
//   conv2:   the MFMA accumulator layout gives lane (point r, half h) channels {4h + (e & 3) + 8 (e >> 2)} of the block: exactly a
//            32x32x16 A-operand fragment (8 k-values per lane) if conv2's K order is permuted accordingly.  The permutation is
//            applied to W2 when it is packed (heads_pack_w2_kernel), so the 16 values feed two MFMA steps of
//            acc2[point][out] += H[point][channels] . W2[out][channels] without touching LDS;
//   after the 32 channel blocks: + bias2, BatchNorm fold, ReLU, max over the object's points as order-preserving keys (atomicMax).
// One wave per SIMD (the points' fine features as B fragments: 136 registers, conv2's accumulators: 128), four waves per
// workgroup sharing the LDS-staged weight blocks (double buffered: 2 x 67 KB).
#include "tgp_common.h"
#include "../../include/tgpose.h"
#include <type_traits>

typedef _Float16 hf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 hf16x4 __attribute__((ext_vector_type(4)));
typedef float hf32x4 __attribute__((ext_vector_type(4)));
typedef float hf32x2 __attribute__((ext_vector_type(2)));
typedef float hf32x16 __attribute__((ext_vector_type(16)));

#define HF_STEPS 17                 // K steps of conv1: 272 / 16
#define HF_C1 1024                  // conv1 channels per head
#define HF_C2 256                   // conv2 channels per head
#define HF_NCB (HF_C1 / 32)
#define HF_AROW (HF_STEPS * 64 + 16)      // (conv_max_fused_kernel) LDS row of a conv weight block: 17 steps x 2 planes x 32 B, + 16 B (conflict-free b128)

struct HeadsParams {
    const float *fine; int ldf, K;
    const float *p1; int ldp1; const int32_t *idx1;
    const float *p2; int ldp2; const int32_t *idx2;
    const float *bias2, *scale2, *shift2; // heads * 256
    uint32_t *keys;                       // (heads, B, 256)
    int *overflow;
    int M, rows_per_obj, B, heads, tiles;
    // the points' features as blocked fp16 planes (tgp_gemm_args.A_planes' layout): fragments load as 1 KB runs, no split
    const char *fine_pl; int fine_kt; const uint32_t *fine_amax;
    const char *wa_pl;                     // conv1 weights of the heads as blocked planes: [head * 32 + block][17][2][h][r][8] fp16
    const char *w2b;                       // [head][block][33 pieces]: conv2 weights [2 steps][8 out blocks][2 planes][h][r][8] fp16, K permuted; vectors
    unsigned long long *stamps;            // (development build) per-wave cycle stamps
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

// ------------------------------------------------------------------------------------------------------------------------
// The heads kernel (round 5 form): every non-matrix instruction sits BETWEEN the matrix instructions.
//
// What the stamps and timing knobs of the round-2..4 kernel said (scripts/heads_time.py; per workgroup 227 k shader cycles at 1.8 GHz, 101 k of
// them the 3168 MFMAs): with ONE wave per SIMD and in-order issue, a run of MFMAs that accumulate into the same registers blocks
// the instruction stream -- each waits for its predecessor -- so whatever follows the run can hide behind its LAST MFMA only.
// The old loop issues three dependent MFMAs, then the step's LDS reads, its DMA piece and (after conv1) 184 vector instructions
// of epilogue 1 (that loop: git history, csrc/heads_fused.hip before round 5): their issue time ADDS to the matrix time (removing the DMA, the gathers, the fragment reads or epilogue 1 each
// shortened the kernel by nearly that part's full issue time).  Here:
//   * every MFMA is followed by its own slice of the other work, fenced by sched_barrier so that the compiler keeps it there: a gap
//     holds <= ~24 cycles of issue (two fragment reads, or one DMA piece, or one gather, or 3-4 vector instructions);
//   * iteration i runs conv1 of block i with epilogue 1 of block i - 1 in its gaps (stage A), then conv2 of block i - 1 with the
//     gathers of block i, the DMA and the accumulator read-out in its gaps (stage B);
//   * the weight images are FRAGMENT-BLOCKED in memory (conv1: the blocked planes of tgp_gemm_args.W_planes, one 32-channel block
//     = 34 contiguous KB; conv2 + the block's bias | scale | shift: 33 contiguous 1 KB pieces per block in lane order), so an
//     LDS-DMA piece is a linear 1 KB copy -- scalar base + one vector offset, no branches, no per-lane offset table, no row
//     padding, 67 pieces per block instead of 72 -- and every fragment read is ds_read_b128 at base + 16 x lane + immediate;
//   * epilogue 1 is 88 vector instructions per block instead of 184 (ReLU as max(v, -0), the low plane by one mixed-precision
//     fma per element, no per-block range tracking: a magnitude beyond fp16's range makes every conv2 sum of its point
//     non-finite, which epilogue 2 sees);
//   * epilogue 2 is branch-free, its 24 per-lane vectors loaded under the last block's conv2.
// Same products in the same order as that kernel: bit-identical keys (checked against it on the benchmark's shape and on odd ones
// before it was removed).
#define HP_APIECES (2 * HF_STEPS)          // conv1 weight block: 17 K-tiles x 2 planes
#define HP_WPIECES 33                      // conv2 weight block (2 steps x 8 out blocks x 2 planes) + 1 piece of epilogue vectors
#define HP_PPIECE (HP_APIECES + 32)        // bias | scale | shift of the block (384 B used)
#define HP_PIECES (HP_APIECES + HP_WPIECES)   // 67
#define HP_BUF (HP_PIECES * 1024)
#define HP_NDMA ((HP_PIECES + 3) / 4)      // 17 wave-instructions per wave and unit (piece j = 4 j0 + wave; piece 67 does not exist)
#define HP_DMA_A 10                        // DMA pieces of a unit issued in stage A (one per K-step from step 0), the other 7 in stage B
#define HP_SB() __builtin_amdgcn_sched_barrier(0)

template <bool PLANES, bool STAMPS, int KNOB = 0>
__global__ __launch_bounds__(256, 1) void heads_fused_kernel(HeadsParams p)
{
    extern __shared__ __attribute__((aligned(16))) char hf_smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int t = blockIdx.x;
    if (t >= p.heads * p.tiles) return;
    const int hd = t / p.tiles, ptile = t - hd * p.tiles;
    const int m0 = ptile * 128 + wave * 32;                     // the wave's first point (may lie past M: then the wave only helps staging)
    const int row = min(m0 + r, p.M - 1);
#define HP_T() (STAMPS ? (unsigned long long)__builtin_readcyclecounter() : 0ull)
    const unsigned long long t_entry = HP_T(), w_entry = STAMPS ? wall_clock64() : 0ull;
    unsigned long long t_a = 0, t_b = 0, t_bar = 0;

    // ---- staging of a unit u = { conv1 weights of block u + 1 | conv2 weights and epilogue vectors of block u } into buffer buf: piece
    // j = 4 j0 + wave is 1 KB at offset 1024 j of the buffer and of the unit's image in memory (pieces 0 .. 33 from the conv1
    // planes, 34 .. 66 from the conv2 image), so a piece is { LDS base + 4096 j0, scalar source base, vector offset + 4096 j0 }
    const uint32_t voff0 = lane * 16 + wave * 1024;
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)hf_smem) + wave * 1024;
    const char *a_src, *w_src, *mid_src;                         // scalar bases of the unit being staged (set per iteration)
    auto dma_unit = [&](const int u) {
        const int ua = u + 1 < HF_NCB ? u + 1 : HF_NCB - 1;     // (the last unit has no next conv1 block: it re-reads the last one)
        a_src = p.wa_pl + (int64_t)(hd * HF_NCB + ua) * (HP_APIECES * 1024);
        w_src = p.w2b + (int64_t)(hd * HF_NCB + (u < 0 ? 0 : u)) * (HP_WPIECES * 1024) - HP_APIECES * 1024;
        mid_src = wave < 2 ? a_src : w_src;                      // pieces 32 .. 35: two of each image
    };
    auto dma = [&](const int buf, const int j0) {
        if (KNOB & 1) return;
        const char *src = j0 * 4 + 3 < HP_APIECES ? a_src : j0 * 4 >= HP_APIECES ? w_src : mid_src;
        if (j0 == HP_NDMA - 1 && (HP_PIECES & 3) != 0) {         // the last round of pieces is short: wave 3 repeats its previous piece
            const uint32_t lds = lds0 + buf * HP_BUF + (wave < (HP_PIECES & 3) ? j0 : j0 - 1) * 4096;
            const uint32_t vo = voff0 + (wave < (HP_PIECES & 3) ? j0 : j0 - 1) * 4096;
            asm volatile("s_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(vo), "s"(src), "{m0}"(lds) : "memory");
            return;
        }
        const uint32_t lds = lds0 + buf * HP_BUF + j0 * 4096;
        const uint32_t vo = voff0 + j0 * 4096;
        // inline assembly: opaque to the compiler's counters (no vmcnt(0) before the LDS reads that follow); vmcnt(0) is written by
        // hand before the barrier that ends an iteration.  The LDS base travels in m0 as a register-constrained input.
        asm volatile("s_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(vo), "s"(src), "{m0}"(lds) : "memory");
    };

    // the first unit (its conv1 half: weights of block 0), then the wave's points as B fragments: fp16 hi / lo planes of fine[row][16 s + 8 h .. + 7]
    dma_unit(-1);
    {
        const char *keep = w_src;
        w_src = a_src, mid_src = a_src;                          // (conv2 pieces of "unit -1": any valid source, never read)
#pragma unroll
        for (int j0 = 0; j0 <= (HP_APIECES - 1) / 4; ++j0) {
            const uint32_t lds = lds0 + j0 * 4096;
            const uint32_t vo = voff0 + j0 * 4096;
            if (j0 * 4 + 3 < HP_APIECES || wave < (HP_APIECES & 3))
                asm volatile("s_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(vo), "s"(a_src), "{m0}"(lds) : "memory");
        }
        w_src = keep;
    }
    uint4 bh[HF_STEPS], bl[HF_STEPS];
    float amax_in = 0.f, poison_in = 0.f;
    if constexpr (PLANES) {
        const int nblk = (p.M + 31) >> 5;
        const int rb = min(m0 >> 5, nblk - 1);
        const char *src = p.fine_pl + (int64_t)rb * p.fine_kt * 2048 + lane * 16;
#pragma unroll
        for (int s = 0; s < HF_STEPS; ++s) {
            bh[s] = *reinterpret_cast<const uint4 *>(src + s * 2048);
            bl[s] = *reinterpret_cast<const uint4 *>(src + s * 2048 + 1024);
        }
        if (m0 + 32 > p.M) {                                        // (wave-uniform) rows past the end were never written: zero them
            const bool dead = m0 + r >= p.M;
#pragma unroll
            for (int s = 0; s < HF_STEPS; ++s)
                if (dead) bh[s] = make_uint4(0u, 0u, 0u, 0u), bl[s] = make_uint4(0u, 0u, 0u, 0u);
        }
        const uint32_t am = p.fine_amax ? p.fine_amax[rb] : 0x3f800000u;
        amax_in = __uint_as_float(am < 0x7f800000u ? am : 0x7f800000u);   // inf for an inf / NaN block: !(amax < 65504) below
    } else {
        const float *fr = p.fine + (int64_t)row * p.ldf + 8 * h;
#pragma unroll
        for (int s = 0; s < HF_STEPS; ++s) {
            float4 v0 = *reinterpret_cast<const float4 *>(fr + 16 * s), v1 = *reinterpret_cast<const float4 *>(fr + 16 * s + 4);
            if (s == HF_STEPS - 1) {                                // the K tail: columns >= K are not the caller's to define
                const int k0 = 16 * s + 8 * h;
                v0.x = k0 + 0 < p.K ? v0.x : 0.f, v0.y = k0 + 1 < p.K ? v0.y : 0.f, v0.z = k0 + 2 < p.K ? v0.z : 0.f, v0.w = k0 + 3 < p.K ? v0.w : 0.f;
                v1.x = k0 + 4 < p.K ? v1.x : 0.f, v1.y = k0 + 5 < p.K ? v1.y : 0.f, v1.z = k0 + 6 < p.K ? v1.z : 0.f, v1.w = k0 + 7 < p.K ? v1.w : 0.f;
            }
            amax_in = fmaxf(amax_in, fmaxf(fmaxf(fabsf(v0.x), fabsf(v0.y)), fmaxf(fabsf(v0.z), fabsf(v0.w))));
            amax_in = fmaxf(amax_in, fmaxf(fmaxf(fabsf(v1.x), fabsf(v1.y)), fmaxf(fabsf(v1.z), fabsf(v1.w))));
            poison_in += 0.f * (((v0.x + v0.y) + (v0.z + v0.w)) + ((v1.x + v1.y) + (v1.z + v1.w)));
            uint2 h0, l0, h1, l1;
            hf_split(v0, h0, l0), hf_split(v1, h1, l1);
            bh[s] = make_uint4(h0.x, h0.y, h1.x, h1.y), bl[s] = make_uint4(l0.x, l0.y, l1.x, l1.y);
        }
    }
    const int i1 = p.idx1[row], i2 = p.idx2[row];
    const float *g1p = p.p1 + (int64_t)i1 * p.ldp1 + hd * HF_C1 + 4 * h;           // + cb * 32 + 8 m: four channels of the lane
    const float *g2p = p.p2 + (int64_t)i2 * p.ldp2 + hd * HF_C1 + 4 * h;

    hf32x16 acc2[HF_C2 / 32];
#pragma unroll
    for (int ob = 0; ob < HF_C2 / 32; ++ob)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc2[ob][e] = 0.f;

    float4 g1[4], g2[4];                   // the lane's gathered coarse products of the block whose epilogue 1 comes next
    float pre[16];                         // that block's conv1 sums (element e: channel 4 h + (e & 3) + 8 (e >> 2), point r)
    uint4 a2h[2], a2l[2];                  // ... and its activations as conv2's A fragments
    auto gather1 = [&](const int cb, const int k) {                // k = 0 .. 7: one 16-byte load
        const int m = k >> 1;
        if (KNOB & 2) { (k & 1 ? g2[m] : g1[m]) = make_float4(0.f, 0.f, 0.f, 0.f); return; }
        if (k & 1) g2[m] = *reinterpret_cast<const float4 *>(g2p + cb * 32 + 8 * m);
        else g1[m] = *reinterpret_cast<const float4 *>(g1p + cb * 32 + 8 * m);
    };

    // ---- epilogue 1 of a block in slices, one per MFMA gap of the next block's conv1 (k = 0 .. 33): group m (four channels of the lane)
    // computes at k = 8 m + 2 .. 8 m + 7 on vectors read three gaps earlier; element by element in the two-launch form's order:
    // + bias, + P1 row, + P2 row, BatchNorm fold, ReLU, fp16 hi / lo
    float4 e_b, e_sc, e_sh, e_v;
    uint2 hh[4], ll[4];
    float e_lo[4];
    auto epi1_read = [&](const char *base, const int m) {
        const float *pv = reinterpret_cast<const float *>(base + HP_PPIECE * 1024) + 4 * h + 8 * m;
        e_b = *reinterpret_cast<const float4 *>(pv), e_sc = *reinterpret_cast<const float4 *>(pv + 32);
        e_sh = *reinterpret_cast<const float4 *>(pv + 64);
    };
    auto epi1 = [&](const char *base, const int k) {
        if (k >= 7 && (k - 7) % 8 == 0 && (k - 7) / 8 + 1 < 4) epi1_read(base, (k - 7) / 8 + 1);     // k = 7, 15, 23: the next group's vectors
        if (k < 2) return;
        const int m = (k - 2) >> 3, ph = (k - 2) & 7;             // four vector instructions per gap
        if (ph == 0) {
            e_v = make_float4(pre[4 * m], pre[4 * m + 1], pre[4 * m + 2], pre[4 * m + 3]);
            e_v.x += e_b.x, e_v.y += e_b.y, e_v.z += e_b.z, e_v.w += e_b.w;
        } else if (ph == 1) {
            e_v.x += g1[m].x, e_v.y += g1[m].y, e_v.z += g1[m].z, e_v.w += g1[m].w;
        } else if (ph == 2) {
            e_v.x += g2[m].x, e_v.y += g2[m].y, e_v.z += g2[m].z, e_v.w += g2[m].w;
        } else if (ph == 3) {
            e_v.x *= e_sc.x, e_v.y *= e_sc.y, e_v.z *= e_sc.z, e_v.w *= e_sc.w;
        } else if (ph == 4) {
            e_v.x += e_sh.x, e_v.y += e_sh.y, e_v.z += e_sh.z, e_v.w += e_sh.w;
        } else if (ph == 5) {
            // ReLU as max(v, -0): v for v > 0, +0 for +0, -0 for v < 0 and for -0 -- what `v > 0 ? v : v * 0` gives for every number.  A
            // NaN would become -0 here; an infinity stays: either makes the point's conv2 sums non-finite through the planes of
            // the block that produced it (an infinity) or came from non-finite operands (the input guard): epilogue 2 flags both
            e_v.x = fmaxf(e_v.x, -0.f), e_v.y = fmaxf(e_v.y, -0.f), e_v.z = fmaxf(e_v.z, -0.f), e_v.w = fmaxf(e_v.w, -0.f);
        } else if (ph == 6) {
            // hi = fp16(v); lo = fp16(v - hi), the difference by one mixed-precision fma per element (hi -> fp32 is exact, one rounding:
            // the value hf_split's convert-and-subtract gives)
            const hf32x4 x = {e_v.x, e_v.y, e_v.z, e_v.w};
            hh[m] = __builtin_bit_cast(uint2, __builtin_convertvector(x, hf16x4));
            asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(e_lo[0]) : "v"(hh[m].x), "v"(e_v.x));
            asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(e_lo[1]) : "v"(hh[m].x), "v"(e_v.y));
        } else {
            asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(e_lo[2]) : "v"(hh[m].y), "v"(e_v.z));
            asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(e_lo[3]) : "v"(hh[m].y), "v"(e_v.w));
            const hf32x4 rest = {e_lo[0], e_lo[1], e_lo[2], e_lo[3]};
            ll[m] = __builtin_bit_cast(uint2, __builtin_convertvector(rest, hf16x4));
        }
    };
    auto epi1_pack = [&]() {
        a2h[0] = make_uint4(hh[0].x, hh[0].y, hh[1].x, hh[1].y), a2h[1] = make_uint4(hh[2].x, hh[2].y, hh[3].x, hh[3].y);
        a2l[0] = make_uint4(ll[0].x, ll[0].y, ll[1].x, ll[1].y), a2l[1] = make_uint4(ll[2].x, ll[2].y, ll[3].x, ll[3].y);
    };

    // ---- stage A: conv1 of block i (32 channels x 32 points, 17 steps x 3 split terms).  Gap 3 s: the fragments of step s + 2 and
    // a DMA piece; gaps 3 s + 1, 3 s + 2: slices 2 s, 2 s + 1 of the previous block's epilogue 1
    hf32x16 acc1;
    auto stage_a = [&](const char *base, const bool with_e1, const int wbuf, const bool all_dma) {
#pragma unroll
        for (int e = 0; e < 16; ++e) acc1[e] = 0.f;
        const char *arow = base + lane * 16;
        // fragments two steps ahead of their MFMAs: with one wave per SIMD nothing else hides the LDS latency
        uint4 fh0 = *reinterpret_cast<const uint4 *>(arow), fl0 = *reinterpret_cast<const uint4 *>(arow + 1024);
        uint4 fh1 = *reinterpret_cast<const uint4 *>(arow + 2048), fl1 = *reinterpret_cast<const uint4 *>(arow + 3072);
        if (with_e1) epi1_read(base, 0);
        HP_SB();
#pragma unroll
        for (int s = 0; s < HF_STEPS; ++s) {
            uint4 fh2 = fh1, fl2 = fl1;
            // smallest terms first, as in the tile kernel: lo x hi, hi x lo, hi x hi
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(hf16x8, fl0), __builtin_bit_cast(hf16x8, bh[s]), acc1, 0, 0, 0);
            HP_SB();
            if (s + 2 < HF_STEPS && !(KNOB & 4)) {
                fh2 = *reinterpret_cast<const uint4 *>(arow + (s + 2) * 2048);
                fl2 = *reinterpret_cast<const uint4 *>(arow + (s + 2) * 2048 + 1024);
            }
            // the next unit's pieces target the other buffer, whose last readers passed the barrier
            if (all_dma) dma(wbuf, s);
            else if (s < HP_DMA_A) dma(wbuf, s);
            HP_SB();
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(hf16x8, fh0), __builtin_bit_cast(hf16x8, bl[s]), acc1, 0, 0, 0);
            HP_SB();
            if (with_e1) epi1(base, 2 * s);
            HP_SB();
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(hf16x8, fh0), __builtin_bit_cast(hf16x8, bh[s]), acc1, 0, 0, 0);
            HP_SB();
            if (with_e1) epi1(base, 2 * s + 1);
            HP_SB();
            fh0 = fh1, fl0 = fl1, fh1 = fh2, fl1 = fl2;
        }
        if (with_e1) epi1_pack();
    };
    // ---- stage B: conv2 partial sums over the PREVIOUS block's 32 channels (32 points x 256 outputs, 2 steps x 8 out blocks x 3 terms).
    // Gap 3 q: the fragments of group q + 2; gap 3 q + 1: a DMA piece (q <= 8); gap 3 q + 2: one gather of this block (q <= 7), then
    // its conv1 sums out of the accumulators, two per gap
    auto stage_b = [&](const char *base, const int gcb, const bool copy_pre, const bool stage, const int wbuf) {
        const char *wrow = base + HP_APIECES * 1024 + lane * 16;
        constexpr int NQ = 2 * (HF_C2 / 32);
        auto wfrag = [&](int q, int plane) { return *reinterpret_cast<const uint4 *>(wrow + (q * 2 + plane) * 1024); };
        uint4 wh0 = wfrag(0, 0), wl0 = wfrag(0, 1), wh1 = wfrag(1, 0), wl1 = wfrag(1, 1);
        HP_SB();
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            uint4 wh2 = wh1, wl2 = wl1;
            const int s2 = q / (HF_C2 / 32), ob = q % (HF_C2 / 32);
            acc2[ob] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(hf16x8, a2h[s2]), __builtin_bit_cast(hf16x8, wl0), acc2[ob], 0, 0, 0);
            HP_SB();
            if (q + 2 < NQ && !(KNOB & 4)) wh2 = wfrag(q + 2, 0), wl2 = wfrag(q + 2, 1);
            HP_SB();
            acc2[ob] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(hf16x8, a2l[s2]), __builtin_bit_cast(hf16x8, wh0), acc2[ob], 0, 0, 0);
            HP_SB();
            if (stage && q + HP_DMA_A < HP_NDMA) dma(wbuf, q + HP_DMA_A);
            HP_SB();
            acc2[ob] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(hf16x8, a2h[s2]), __builtin_bit_cast(hf16x8, wh0), acc2[ob], 0, 0, 0);
            HP_SB();
            if (gcb >= 0 && q < 8) gather1(gcb, q);
            if (copy_pre && q >= 8) pre[2 * (q - 8)] = acc1[2 * (q - 8)], pre[2 * (q - 8) + 1] = acc1[2 * (q - 8) + 1];
            HP_SB();
            wh0 = wh1, wl0 = wl1, wh1 = wh2, wl1 = wl2;
        }
    };

    // this wave's DMA pieces have landed when at most the 34 fragment loads issued after them (and what followed) are in flight: block 0's
    // conv1 starts on the fragments that have arrived (the compiler's own waits cover each first use) instead of behind the last of them
    if constexpr (PLANES) __builtin_amdgcn_s_waitcnt(0x0f70 | (34 & 15) | ((34 >> 4) << 14));
    else __builtin_amdgcn_s_waitcnt(0x0f70);
    __builtin_amdgcn_s_barrier();
    const unsigned long long t_pro = HP_T();

    // iteration 0: conv1 of block 0, gathers of block 0, the whole unit 0 (no conv2 to spread it through)
#pragma unroll
    for (int k = 0; k < 8; ++k) gather1(0, k);
    dma_unit(0);
    stage_a(hf_smem, false, 1, true);
#pragma unroll
    for (int e = 0; e < 16; ++e) pre[e] = acc1[e];
    __builtin_amdgcn_s_waitcnt(0x0f70);
    __syncthreads();
    // iterations 1 .. 31: buffer i & 1 holds { conv1 weights of block i | conv2 weights and vectors of block i - 1 }
#pragma unroll 1
    for (int i = 1; i < HF_NCB; ++i) {
        const char *base = hf_smem + (i & 1) * HP_BUF;
        const unsigned long long t0 = HP_T();
        dma_unit(i);
        stage_a(base, true, (i & 1) ^ 1, false);
        if (STAMPS) HP_SB();
        const unsigned long long t1 = HP_T();
        if (STAMPS) HP_SB();
        stage_b(base, i, true, true, (i & 1) ^ 1);
        if (STAMPS) HP_SB();
        const unsigned long long t2 = HP_T();
        __builtin_amdgcn_s_waitcnt(0x0f70);                      // vmcnt(0): this wave's share of the next unit has landed (and its gathers)
        __syncthreads();                                         // ... everybody's has, and this buffer's readers are done
        const unsigned long long t3 = HP_T();
        t_a += t1 - t0, t_b += t2 - t1, t_bar += t3 - t2;
    }
    const unsigned long long t_loop = HP_T();
    // iteration 32: epilogue 1 and conv2 of block 31; the vectors of epilogue 2 arrive meanwhile
    float b2v[HF_C2 / 32], sc2v[HF_C2 / 32], sh2v[HF_C2 / 32];
#pragma unroll
    for (int ob = 0; ob < HF_C2 / 32; ++ob) {
        const int o = hd * HF_C2 + ob * 32 + r;
        b2v[ob] = p.bias2[o], sc2v[ob] = p.scale2[o], sh2v[ob] = p.shift2[o];
    }
    {
        const char *base = hf_smem + (HF_NCB & 1) * HP_BUF;
        epi1_read(base, 0);
#pragma unroll
        for (int k = 0; k < 2 * HF_STEPS; ++k) epi1(base, k);
        epi1_pack();
        stage_b(base, -1, false, false, 0);
    }
    const unsigned long long t_last = HP_T();

    // ---- epilogue 2: lane (out column r of block ob, half h) holds points (e & 3) + 8 (e >> 2) + 4 h of the wave's 32.
    // fp16 range guard: a wave whose input features or conv1 activations left fp16's range (or held a NaN) has non-finite sums.  With a
    // flag to raise, the wave writes no keys and the caller's predicated two-launch form (guarded arithmetic) supplies them; without
    // one, NaN keys are loud.  And the small side: a wave whose input features are ALL below 2^-4 (and not all zero) would lose
    // relative precision in every product of conv1 (gemm.hip, small side of the range guard): same treatment.
    if (m0 < p.M) {
        const int obj0 = m0 / p.rows_per_obj, bound = (obj0 + 1) * p.rows_per_obj;
        uint32_t *kp = p.keys + ((int64_t)hd * p.B + obj0) * HF_C2 + r;
        const bool plain = m0 + 32 <= bound && m0 + 32 <= p.M;     // (wave-uniform) the usual wave: 32 live points of one object
        uint32_t k0[HF_C2 / 32], k1[HF_C2 / 32];
        bool finite = true;
        if (plain) {
            // the maximum is taken on the floats and keyed once per column: the key is order-preserving (and max(-0, +0) = +0 as
            // key(-0) < key(+0)), so key(max v) = max key(v) for every value that is not a NaN; NaNs and infinities show in the sum of
            // the pre-activation values (a finite sum of sixteen floats that overflows flags too: the repair then recomputes a tile
            // that was fine, nothing else).  ReLU as max(v, -0): what `v > 0 ? v : v * 0` gives for every number.
#pragma unroll
            for (int ob = 0; ob < HF_C2 / 32; ++ob) {
                float mx = -0.f, sum = 0.f;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    float v = acc2[ob][e] + b2v[ob];
                    v = v * sc2v[ob] + sh2v[ob];
                    sum += v;
                    mx = fmaxf(mx, v);
                }
                finite &= sum - sum == 0.f;                          // (inf - inf and NaN - NaN are NaN)
                k0[ob] = tgp_float_key(mx), k1[ob] = 0;
            }
        } else {                                                   // a wave that straddles two objects, or the batch's end
#pragma unroll
            for (int ob = 0; ob < HF_C2 / 32; ++ob) {
                uint32_t ka = 0, kb = 0;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int prow = m0 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    float v = acc2[ob][e] + b2v[ob];
                    v = v * sc2v[ob] + sh2v[ob];
                    v = v > 0.f ? v : v * 0.f;
                    const bool live = prow < p.M;
                    finite &= !live || (v == v && v < __builtin_inff());
                    const uint32_t key = live ? tgp_float_key(v) : 0u;
                    const uint32_t xa = prow >= bound ? 0u : key, xb = prow >= bound ? key : 0u;
                    ka = xa > ka ? xa : ka, kb = xb > kb ? xb : kb;
                }
                k0[ob] = ka, k1[ob] = kb;
            }
        }
        const bool tiny_in = __ballot(amax_in >= 0.0625f) == 0ull && __ballot(amax_in > 0.f) != 0ull;
        const bool bad = tiny_in || __ballot(!(amax_in < 65504.f) || poison_in != poison_in || !finite) != 0ull;
        if (p.overflow && bad) {
            if (lane == 0) atomicOr(p.overflow, 1);
        } else {
#pragma unroll
            for (int ob = 0; ob < HF_C2 / 32; ++ob) {
                uint32_t ka = k0[ob], kb = k1[ob];
                if (bad) ka = 0xffc00000u;                             // (no flag to raise: the key of a NaN -- loud)
                const uint32_t o0 = (uint32_t)__shfl_xor((int)ka, 32, 64), o1 = (uint32_t)__shfl_xor((int)kb, 32, 64);
                ka = o0 > ka ? o0 : ka, kb = o1 > kb ? o1 : kb;
                if (h == 0) {
                    if (ka) atomicMax(kp + ob * 32, ka);
                    if (kb) atomicMax(kp + ob * 32 + HF_C2, kb);
                }
            }
        }
    }
    if (STAMPS && p.stamps && lane == 0) {
        unsigned long long *o = p.stamps + ((size_t)blockIdx.x * 4 + wave) * 12;
        const unsigned long long t_end = HP_T();
        o[0] = t_pro - t_entry, o[1] = t_loop - t_pro, o[2] = t_end - t_last, o[3] = t_last - t_loop, o[4] = t_a, o[5] = 0, o[6] = t_b, o[7] = t_bar;
        o[8] = t_end - t_entry, o[9] = wall_clock64() - w_entry, o[10] = w_entry, o[11] = 0;
    }
#undef HP_T
}


// ------------------------------------------------------------------------------------------------------------------------
// conv -> BatchNorm -> LeakyReLU -> max over points of a factored layer whose activation only feeds the max (conv_5 of Face_Enc,
// FaceRecon.py:76-77 `conv_5` + `feat.max(1)`): the fused heads kernel's conv1 with the operands' roles swapped -- the points'
// fragments are the A operand, the weight rows the B operand -- so that a lane holds ONE channel and 16 points: the epilogue's
// vectors are scalars per lane, the gathered coarse products are 4-byte loads coalesced over the channels, and the max over the
// points is 15 in-lane maxima + one cross-half shuffle.  No conv2 accumulators: two waves per SIMD; a workgroup takes 128 points and
// CBW of the channel blocks (68 KB of LDS: two workgroups per CU).  Same products in the same order as the tile kernel (activation
// hi x weight lo, lo x hi, hi x hi; K ascending), same epilogue order.
//
// Round 5: software-pipelined over the channel blocks, as heads_fused_kernel's stage A.  The first form ran a block as gathers ->
// 51 MFMAs -> ~220 vector instructions of epilogue -> atomics -> vmcnt(0) -> barrier, one phase after the other, and the two
// workgroups of a CU -- started together -- met in the same phases: matrix cores busy 28 % of the launch.  Now iteration i issues
// block i's MFMAs into one of two accumulators and places, in the gaps behind them, the fragment reads of the next step, the
// LDS-DMA of block i + 1's weights, block i - 1's epilogue element by element, and -- as soon as an element has consumed its
// gathered values -- the gathers of block i into the same registers (a whole iteration ahead of their use).  The iteration's wait
// is counted (the DMA pieces are older than the last sixteen gathers), so neither gathers nor atomics are waited for at the
// barrier.  The weights come as blocked fp16 planes (tgp_gemm_args.W_planes' layout, 17 K-tiles: one 32-channel block = 34
// contiguous 1 KB pieces in lane order): a DMA piece is a linear copy, a fragment read is base + 16 x lane + immediate.
#define CM_PIECES (2 * HF_STEPS)          // 34 pieces of 1 KB per channel block
#define CM_BUF (CM_PIECES * 1024)
#define CM_NDMA ((CM_PIECES + 3) / 4)     // 9 wave-instructions per wave and block (the last round is short: waves 2, 3 repeat a piece)

struct ConvMaxParams {
    const float *fine; int ldf, K;
    const char *wa_pl;                    // the layer's weights as blocked planes: [block of 32 channels][17][2][h][r][8] fp16
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

__device__ __forceinline__ constexpr int cm_vmcnt(int n) { return 0x0f70 | (n & 15) | ((n >> 4) << 14); }

template <bool PLANES, int KNOB = 0>
__global__ __launch_bounds__(256, 2) void conv_max_fused_kernel(ConvMaxParams p)
{
    extern __shared__ __attribute__((aligned(16))) char hf_smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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

    // the first block's weights travel while the points' fragments are fetched
    const uint32_t voff0 = lane * 16 + wave * 1024;
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)hf_smem) + wave * 1024;
    auto dma = [&](const int cb, const int buf, const int j0) {
        if ((KNOB & 1) && cb != cb0) return;
        const char *src = p.wa_pl + (int64_t)cb * CM_BUF;
        const int jj = (j0 == CM_NDMA - 1 && wave >= (CM_PIECES & 3)) ? j0 - 1 : j0;    // pieces 34, 35 do not exist
        const uint32_t lds = lds0 + buf * CM_BUF + jj * 4096;
        const uint32_t vo = voff0 + jj * 4096;
        // inline assembly: opaque to the compiler's counters; the waits are written by hand before the barrier that ends an iteration
        asm volatile("s_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(vo), "s"(src), "{m0}"(lds) : "memory");
    };
#pragma unroll
    for (int j0 = 0; j0 < CM_NDMA; ++j0) dma(cb0, 0, j0);

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
    // the lane's 16 points (accumulator rows (e & 3) + 8 (e >> 2) + 4 h): BYTE offsets of their coarse products' rows.  They
    // depend on (h, e) only, so they live in 256 bytes of LDS per wave instead of 32 registers per lane.
    __shared__ __attribute__((aligned(16))) uint32_t s_off[4][2][2][16];
    {
        const int lvl = lane >> 5, hh = (lane >> 4) & 1, e = lane & 15;
        const int pr = min(m0 + (e & 3) + 8 * (e >> 2) + 4 * hh, p.M - 1);
        s_off[wave][lvl][hh][e] = 4u * (uint32_t)(lvl ? p.idx2[pr] * p.ldp2 : p.idx1[pr] * p.ldp1);
    }
    const bool live = m0 < p.M;
    const bool bad = __ballot(!(amax < 65504.f) || poison != poison) != 0ull ||      // fp16 range guard, as in the heads kernel
                     (__ballot(amax >= 0.0625f) == 0ull && __ballot(amax > 0.f) != 0ull);   // ... and its small side (all inputs < 2^-4)
    if (live && bad && p.overflow && lane == 0) atomicOr(p.overflow, 1);
    const int obj0 = m0 / p.rows_per_obj, bound = (obj0 + 1) * p.rows_per_obj;
    // accumulator element e is the point m0 + (e & 3) + 8 (e >> 2) + 4 h: past the object's last row when that offset reaches lim_b,
    // past the batch's when it reaches lim_m.  A wave whose 32 points are one object's and exist takes the plain epilogue.
    const int lim_b = bound - m0 - 4 * h, lim_m = p.M - m0 - 4 * h;
    const bool plain_wave = m0 + 32 <= bound && m0 + 32 <= p.M;
    const bool store = live && h == 0 && !(bad && p.overflow);
    const uint32_t *o1 = s_off[wave][0][h], *o2 = s_off[wave][1][h];
    const uint32_t r4 = 4u * r;
    const float ninf = -__builtin_inff();

    float g1[16], g2[16];                 // the gathered coarse products of the block whose epilogue comes next
    float bP, scP, shP, bN, scN, shN;     // bias / scale / shift of the lane's channel: previous block, this block
    float mx0 = ninf, mx1 = ninf, chk = 0.f, ev = 0.f;
    uint32_t ea = 0, ec = 0;
    // (scalar base + 32-bit lane offset: one address add per gather)
    auto at = [&](const float *base, const uint32_t off) { return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(base) + off); };
    // block cb's epilogue, slot t of 34: the lane's channel is 32 cb + r, its 16 accumulator elements are 16 points.  Slots 2 e, 2 e + 1:
    // element e, in the tile kernel's order (+ bias, + P1, + P2, BatchNorm fold, LeakyReLU); its gathered values are replaced by block
    // cb + 1's as soon as they are consumed (next).  Written for the fewest vector instructions: their issue time ADDS to the matrix
    // time in this kernel (timing knobs of the development build: without the epilogue 41 us, with its arithmetic alone 60, all of it
    // 64; packed fp32 pairs are split again by the compiler inside the MFMAs' shadow and only cost registers: 114 us).  The maximum is
    // taken on the floats -- LeakyReLU with a slope in [0, 1] is max(v, slope v), to the bit -- and keyed once per block; a NaN or an
    // infinity among the values makes the key a NaN's (chk), as the per-element keys did.  Slots 32, 33: the other half's maxima, the
    // atomics.
    auto epi = [&](auto plain, const hf32x16 &acc, const int cb, const int t, const bool next_) {
        const bool next = next_ && !(KNOB & 16);
        if (t < 32) {
            const int e = t >> 1, off = (e & 3) + 8 * (e >> 2);
            if ((t & 1) == 0) {
                if (t == 0) mx0 = ninf, mx1 = ninf, chk = 0.f;
                ev = acc[e] + bP;
                ev += g1[e];
                ev += g2[e];
                ev = ev * scP + shP;
                if (next) ea = o1[e] + r4, ec = o2[e] + r4;
            } else {
                const float v = fmaxf(ev, ev * p.slope);
                chk = __builtin_fmaf(v, 0.f, chk);
                if constexpr (decltype(plain)::value) mx0 = fmaxf(mx0, v);
                else {
                    const float w = off < lim_m ? v : ninf;
                    const bool own = off < lim_b;
                    mx0 = fmaxf(mx0, own ? w : ninf), mx1 = fmaxf(mx1, own ? ninf : w);
                }
                if (next && !(KNOB & 2)) g1[e] = at(p.p1 + (cb + 1) * 32, ea), g2[e] = at(p.p2 + (cb + 1) * 32, ec);
            }
        } else if (t == 32) {
            mx0 = fmaxf(mx0, __shfl_xor(mx0, 32, 64)), chk += __shfl_xor(chk, 32, 64);
            if constexpr (!decltype(plain)::value) mx1 = fmaxf(mx1, __shfl_xor(mx1, 32, 64));
        } else {
            if (store) {
                uint32_t *kp = p.keys + (int64_t)obj0 * p.ldk + cb * 32 + r;
                const bool loud = chk != 0.f;                    // (chk is 0 or a NaN)
                if (mx0 > ninf || loud) atomicMax(kp, loud ? 0xffc00000u : tgp_float_key(mx0));
                if constexpr (!decltype(plain)::value)
                    if (mx1 > ninf || loud) atomicMax(kp + p.ldk, loud ? 0xffc00000u : tgp_float_key(mx1));
            }
        }
    };
    // iteration: block cb's 51 MFMAs into accN; in their gaps the next step's fragments, the DMA of block cb + 1 (with_dma), block
    // cb - 1's epilogue from accP (with_epi) and block cb's vectors
    auto iter = [&](auto plain, hf32x16 &accN, const hf32x16 &accP, const int cb, const int buf, const bool with_epi, const bool with_dma) {
#pragma unroll
        for (int e = 0; e < 16; ++e) accN[e] = 0.f;
        const char *wrow = hf_smem + buf * CM_BUF + lane * 16;
        // (fragments one step ahead: the SIMD's other wave covers the LDS latency, and the registers are needed)
        uint4 fh0 = *reinterpret_cast<const uint4 *>(wrow), fl0 = *reinterpret_cast<const uint4 *>(wrow + 1024);
        HP_SB();
#pragma unroll
        for (int s = 0; s < HF_STEPS; ++s) {
            uint4 fh1 = fh0, fl1 = fl0;
            accN = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(hf16x8, ah[s]), __builtin_bit_cast(hf16x8, fl0), accN, 0, 0, 0);
            HP_SB();
            if (s + 1 < HF_STEPS && !(KNOB & 4)) {
                fh1 = *reinterpret_cast<const uint4 *>(wrow + (s + 1) * 2048);
                fl1 = *reinterpret_cast<const uint4 *>(wrow + (s + 1) * 2048 + 1024);
            }
            if (with_dma && s < CM_NDMA) dma(cb + 1, buf ^ 1, s);
            HP_SB();
            accN = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(hf16x8, al[s]), __builtin_bit_cast(hf16x8, fh0), accN, 0, 0, 0);
            HP_SB();
            if (with_epi && !(KNOB & 8)) epi(plain, accP, cb - 1, 2 * s, true);
            HP_SB();
            accN = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(hf16x8, ah[s]), __builtin_bit_cast(hf16x8, fh0), accN, 0, 0, 0);
            HP_SB();
            if (with_epi && !(KNOB & 8)) epi(plain, accP, cb - 1, 2 * s + 1, true);
            if (s == HF_STEPS - 1) bN = p.bias[cb * 32 + r], scN = p.scale[cb * 32 + r], shN = p.shift[cb * 32 + r];
            HP_SB();
            fh0 = fh1, fl0 = fl1;
        }
        bP = bN, scP = scN, shP = shN;
        if (with_dma) {
            // this wave's DMA pieces have landed when at most the sixteen gathers issued after the last piece (and whatever followed
            // them) are outstanding; without an epilogue nothing was issued behind them
            if (with_epi && !(KNOB & 10)) __builtin_amdgcn_s_waitcnt(cm_vmcnt(16));
            else __builtin_amdgcn_s_waitcnt(cm_vmcnt(0));
            __builtin_amdgcn_s_barrier();
        }
    };
    auto run = [&](auto plain) {
        hf32x16 acc0, acc1;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc1[e] = 0.f;
        iter(plain, acc0, acc1, cb0, 0, false, cb0 + 1 < cb1);
        int cb = cb0 + 1;
        for (; cb + 1 < cb1; cb += 2) {
            iter(plain, acc1, acc0, cb, 1, true, true);
            iter(plain, acc0, acc1, cb + 1, 0, true, cb + 2 < cb1);
        }
        if (cb < cb1) {
            iter(plain, acc1, acc0, cb, 1, true, false);
#pragma unroll
            for (int t = 0; t < 34; ++t) epi(plain, acc1, cb, t, false);
        } else {
#pragma unroll
            for (int t = 0; t < 34; ++t) epi(plain, acc0, cb - 1, t, false);
        }
    };

#pragma unroll
    for (int e = 0; e < 16; ++e) g1[e] = at(p.p1 + cb0 * 32, o1[e] + r4), g2[e] = at(p.p2 + cb0 * 32, o2[e] + r4);
    __builtin_amdgcn_s_waitcnt(cm_vmcnt(0));
    __syncthreads();
    if (plain_wave) run(std::true_type{});
    else run(std::false_type{});
}

#ifdef TGP_DEV   // development build: per-wave cycle stamps and timing-only knobs (scripts/heads_time.py)
static unsigned long long *tgp_heads_stamps = nullptr;
static int tgp_heads_knobs = 0;
extern "C" int tgp_debug_set_heads_stamps(void *buf) { tgp_heads_stamps = reinterpret_cast<unsigned long long *>(buf); return 0; }
extern "C" int tgp_debug_set_heads_knobs(int v) { tgp_heads_knobs = v; return 0; }
#endif

extern "C" int tgp_conv_max_fused(const tgp_conv_max_fused_args *a, tgp_stream_t stream)
{
    TGP_REQUIRE(a && a->fine && a->wa_planes && a->p1 && a->p2 && a->idx1 && a->idx2 && a->bias && a->scale && a->shift && a->keys);
    TGP_REQUIRE(a->M > 0 && a->C > 0 && (a->C & 31) == 0 && a->rows_per_obj >= 32 && a->M % a->rows_per_obj == 0 && a->ldk >= a->C);
    TGP_REQUIRE(a->K > 0 && a->K <= 16 * HF_STEPS && a->K > 16 * (HF_STEPS - 1) && a->ldf >= 16 * HF_STEPS && (a->ldf & 3) == 0);
    TGP_REQUIRE(a->ldp1 >= a->C && a->ldp2 >= a->C);
    // the coarse products' rows are addressed with 32-bit byte offsets
    TGP_REQUIRE((int64_t)a->p1_rows * a->ldp1 < (1ll << 30) && (int64_t)a->p2_rows * a->ldp2 < (1ll << 30));
    TGP_REQUIRE(a->slope >= 0.f && a->slope <= 1.f);             // LeakyReLU as max(v, slope v)
    auto al16 = [](const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    TGP_REQUIRE(al16(a->fine) && al16(a->wa_planes) && al16(a->bias) && al16(a->scale) && al16(a->shift));
    ConvMaxParams p;
    p.fine = a->fine, p.ldf = a->ldf, p.K = a->K;
    p.wa_pl = reinterpret_cast<const char *>(a->wa_planes);
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
    static TgpLdsAttr attr_t, attr_f;
    if (const int e = tgp_lds_attr(attr_t, reinterpret_cast<const void *>(conv_max_fused_kernel<true>), 2 * CM_BUF)) return e;
    if (const int e = tgp_lds_attr(attr_f, reinterpret_cast<const void *>(conv_max_fused_kernel<false>), 2 * CM_BUF)) return e;
    const int grid = p.main_tiles * p.chunks + (p.tiles - p.main_tiles) * (p.C / 32);
#ifdef TGP_DEV
    static TgpLdsAttr attr_k[5];
#define CM_KNOB(I, N)                                                                                        \
    if (p.fine_pl && tgp_heads_knobs == N) {                                                                 \
        if (const int e = tgp_lds_attr(attr_k[I], reinterpret_cast<const void *>(conv_max_fused_kernel<true, N>), 2 * CM_BUF)) return e; \
        hipLaunchKernelGGL((conv_max_fused_kernel<true, N>), dim3(grid), dim3(256), 2 * CM_BUF, tgp_hs(stream), p); \
        return TGP_LAUNCH_RESULT();                                                                          \
    }
    CM_KNOB(0, 1) CM_KNOB(1, 2) CM_KNOB(2, 4) CM_KNOB(3, 8) CM_KNOB(4, 16)
#undef CM_KNOB
#endif
    if (p.fine_pl) hipLaunchKernelGGL(conv_max_fused_kernel<true>, dim3(grid), dim3(256), 2 * CM_BUF, tgp_hs(stream), p);
    else hipLaunchKernelGGL(conv_max_fused_kernel<false>, dim3(grid), dim3(256), 2 * CM_BUF, tgp_hs(stream), p);
    return TGP_LAUNCH_RESULT();
}

// W2 (heads, 256, 1024) fp32 + the heads' conv1 epilogue vectors -> per (head, channel block) 33 pieces of 1 KB:
// pieces (s2 * 8 + ob) * 2 + plane: [lane = 32 h + r][8] fp16 in the lane order of the B operand (lane (r, h): out column 32 ob + r,
// k slots 8 h .. 8 h + 7 of step s2), conv2's K order permuted to the layout the conv1 accumulators leave the channels in (slot
// 8 h + t of step s2 is channel 32 cb + 16 s2 + 8 (t >> 2) + 4 h + (t & 3)); piece 32: bias | scale | shift of the block's 32
// channels as 3 x 32 floats (the rest of the piece is zero)
__global__ void heads_pack_w2_kernel(const float *__restrict__ w2, const float *__restrict__ bias1, const float *__restrict__ scale1,
                                      const float *__restrict__ shift1, int heads, uint16_t *__restrict__ out)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)heads * HF_NCB * HF_C2 * 32;
    if (t >= total) return;
    const int tt = (int)(t & 7), rr = (int)((t >> 3) & 31), hh = (int)((t >> 8) & 1), ob = (int)((t >> 9) & 7), s2 = (int)((t >> 12) & 1);
    const int cb = (int)((t >> 13) % HF_NCB), hd = (int)((t >> 13) / HF_NCB);
    const int ch = 32 * cb + 16 * s2 + 8 * (tt >> 2) + 4 * hh + (tt & 3);
    uint16_t *blk = out + ((int64_t)hd * HF_NCB + cb) * (HP_WPIECES * 512);
    if (w2) {                                                    // (NULL: only the vectors are refreshed -- the BatchNorm fold moved)
        const float v = w2[((int64_t)hd * HF_C2 + ob * 32 + rr) * HF_C1 + ch];
        const _Float16 hi = (_Float16)v;
        const _Float16 lo = (_Float16)(v - (float)hi);
        uint16_t *dst = blk + (int64_t)((s2 * 8 + ob) * 2) * 512 + (hh * 32 + rr) * 8 + tt;
        dst[0] = __builtin_bit_cast(uint16_t, hi);
        dst[512] = __builtin_bit_cast(uint16_t, lo);
    }
    const int q = (int)(t & 8191);                               // the block's thread index: the first 256 fill the vector piece
    if (q < 256) {
        float *pv = reinterpret_cast<float *>(blk + 32 * 512);
        const int c = hd * HF_C1 + cb * 32 + (q & 31);
        pv[q] = q < 32 ? bias1[c] : q < 64 ? scale1[c] : q < 96 ? shift1[c] : 0.f;
    }
}

extern "C" int64_t tgp_heads_w2_bytes(int heads) { return heads > 0 ? (int64_t)heads * HF_NCB * HP_WPIECES * 1024 : 0; }

extern "C" int tgp_heads_pack_w2(const float *w2, const float *bias1, const float *scale1, const float *shift1, int heads, void *out,
                                 tgp_stream_t stream)
{
    TGP_REQUIRE(bias1 && scale1 && shift1 && out && heads > 0);
    const int64_t total = (int64_t)heads * HF_NCB * HF_C2 * 32;
    hipLaunchKernelGGL(heads_pack_w2_kernel, dim3(tgp_cdiv(total, 256)), dim3(256), 0, tgp_hs(stream), w2, bias1, scale1, shift1, heads,
                       reinterpret_cast<uint16_t *>(out));
    return TGP_LAUNCH_RESULT();
}

extern "C" int tgp_heads_fused(const tgp_heads_fused_args *a, tgp_stream_t stream)
{
    TGP_REQUIRE(a && a->fine && a->wa_planes && a->p1 && a->p2 && a->idx1 && a->idx2 && a->w2p && a->bias2 && a->scale2 && a->shift2 &&
                a->keys);
    TGP_REQUIRE(a->M > 0 && a->B > 0 && a->heads > 0 && a->rows_per_obj >= 32 && (int64_t)a->B * a->rows_per_obj == a->M);
    TGP_REQUIRE(a->K > 0 && a->K <= 16 * HF_STEPS && a->K > 16 * (HF_STEPS - 1) && a->ldf >= 16 * HF_STEPS && (a->ldf & 3) == 0);
    TGP_REQUIRE((a->ldp1 & 3) == 0 && (a->ldp2 & 3) == 0 && a->ldp1 >= a->heads * HF_C1 && a->ldp2 >= a->heads * HF_C1);
    auto al16 = [](const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    TGP_REQUIRE(al16(a->fine) && al16(a->wa_planes) && al16(a->p1) && al16(a->p2) && al16(a->w2p));
    HeadsParams p;
    p.fine = a->fine, p.ldf = a->ldf, p.K = a->K;
    p.wa_pl = reinterpret_cast<const char *>(a->wa_planes);
    p.p1 = a->p1, p.ldp1 = a->ldp1, p.idx1 = a->idx1, p.p2 = a->p2, p.ldp2 = a->ldp2, p.idx2 = a->idx2;
    p.w2b = reinterpret_cast<const char *>(a->w2p);
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
    p.stamps = nullptr;
    // one workgroup per 128-point tile and head; a CU holds one (134 KB of LDS, one wave per SIMD)
    const dim3 grid(p.heads * p.tiles);
    static TgpLdsAttr attr;
#define HF_LAUNCH(...)                                                                                       \
    do {                                                                                                     \
        const int e_ = tgp_lds_attr(attr, reinterpret_cast<const void *>(heads_fused_kernel<__VA_ARGS__>), 2 * HP_BUF);   \
        if (e_) return e_;                                                                                   \
        hipLaunchKernelGGL((heads_fused_kernel<__VA_ARGS__>), grid, dim3(256), 2 * HP_BUF, tgp_hs(stream), p); \
        return TGP_LAUNCH_RESULT();                                                                          \
    } while (0)
#ifdef TGP_DEV
    static TgpLdsAttr attr_s, attr_k[4];
    if (p.fine_pl && tgp_heads_stamps) {
        p.stamps = tgp_heads_stamps;
        const int e_ = tgp_lds_attr(attr_s, reinterpret_cast<const void *>(heads_fused_kernel<true, true, 0>), 2 * HP_BUF);
        if (e_) return e_;
        hipLaunchKernelGGL((heads_fused_kernel<true, true, 0>), grid, dim3(256), 2 * HP_BUF, tgp_hs(stream), p);
        return TGP_LAUNCH_RESULT();
    }
#define HF_KNOB(I, N)                                                                                        \
    if (p.fine_pl && tgp_heads_knobs == N) {                                                                 \
        const int e_ = tgp_lds_attr(attr_k[I], reinterpret_cast<const void *>(heads_fused_kernel<true, false, N>), 2 * HP_BUF); \
        if (e_) return e_;                                                                                   \
        hipLaunchKernelGGL((heads_fused_kernel<true, false, N>), grid, dim3(256), 2 * HP_BUF, tgp_hs(stream), p); \
        return TGP_LAUNCH_RESULT();                                                                          \
    }
    HF_KNOB(0, 1) HF_KNOB(1, 2) HF_KNOB(2, 4) HF_KNOB(3, 7)
#undef HF_KNOB
#endif
    static TgpLdsAttr attr_f;
    if (p.fine_pl) HF_LAUNCH(true, false, 0);
    {
        const int e_ = tgp_lds_attr(attr_f, reinterpret_cast<const void *>(heads_fused_kernel<false, false, 0>), 2 * HP_BUF);
        if (e_) return e_;
        hipLaunchKernelGGL((heads_fused_kernel<false, false, 0>), grid, dim3(256), 2 * HP_BUF, tgp_hs(stream), p);
        return TGP_LAUNCH_RESULT();
    }
#undef HF_LAUNCH
}
