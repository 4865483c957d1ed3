// Shared device/host helpers for libtgpose_hip.so (gfx950 only).
// The library is compiled with -ffp-contract=off: every fused multiply-add in the kernels is an
// explicit fmaf()/MFMA, every other product/sum rounds on its own, as the CPU path does.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/tgpose.h"

#define TGP_WAVE 64

#define TGP_REQUIRE(cond)            \
    do {                             \
        if (!(cond)) return TGP_EINVAL; \
    } while (0)

#define TGP_LAUNCH_RESULT() ((int)hipGetLastError())

static inline hipStream_t tgp_hs(tgp_stream_t s) { return (hipStream_t)s; }

static inline int tgp_cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE attribute of a kernel: a process that runs on a second GPU after the
// first needs it set there too (a launch asking for more than 64 KB of LDS fails without it).  One TgpLdsAttr per kernel
// instance remembers the devices it has been set on; returns 0 or the hipError_t.
struct TgpLdsAttr {
    bool done[64] = {};
};
static inline int tgp_lds_attr(TgpLdsAttr &st, const void *fn, int bytes)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    if (dev >= 0 && dev < 64 && st.done[dev]) return 0;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return (int)e;
    if (dev >= 0 && dev < 64) st.done[dev] = true;
    return 0;
}

// Objects are dealt to XCDs so that all workgroups of one object run on one XCD (its gathered
// tables then stay in that XCD's 4 MiB L2).  Workgroups are dispatched round-robin over the 8
// XCDs, so linear block id L sits on XCD-group L % 8; this is a speed hint only, correctness
// never depends on it.  Returns false for padding blocks (object index beyond B).
__device__ __forceinline__ bool tgp_xcd_object_tile(int L, int B, int tiles_per_obj, int &obj, int &tile)
{
    const int xcd = L & 7;
    const int seq = L >> 3;
    obj = (seq / tiles_per_obj) * 8 + xcd;
    tile = seq % tiles_per_obj;
    return obj < B;
}
static inline int tgp_xcd_grid(int B, int tiles_per_obj) { return ((B + 7) / 8) * 8 * tiles_per_obj; }

// order-preserving float -> uint32 key (larger float <=> larger key); key 0 is below every float
__device__ __forceinline__ uint32_t tgp_float_key(float f)
{
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float tgp_key_float(uint32_t k)
{
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}
