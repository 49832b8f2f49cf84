// k_common.h — device-side helpers shared by the gfx950 kernels.
//
// Parity rules: the reference is Rust — every f32 expression is evaluated in
// f32 with one rounding per operation and NO fused multiply-add.  The bit-exact kernels are therefore
// compiled with -ffp-contract=off and use '/' and sqrtf, which hipcc lowers to the IEEE-correct
// sequences (-fhip-fp32-correctly-rounded-divide-sqrt, f32 denormals on).  Where an FMA is used it is
// written explicitly (__builtin_fmaf) in a place where it is *proved* to round like the reference's
// expression (div255).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <algorithm>

#define PFX_DEV __device__ __forceinline__

namespace pfxk {

// Pins a value in a VGPR at this point.  Used where both sides of a per-pixel choice are computed on purpose and a select
// picks one: without it LLVM sinks the unchosen side back behind a divergent branch (exec-mask save / restore / s_cbranch per
// choice, which costs more issue slots than the few VALU ops it skips — the lanes of a wave take both sides anyway).
PFX_DEV float pin(float v)
{
    asm volatile("" : "+v"(v));
    return v;
}

// x / 255.0f for integer-valued x in [0,255], correctly rounded, in 2 VALU ops instead of an IEEE divide.
// 1/255 = C_HI + C_LO (C_HI = RN(1/255), C_LO = RN(1/255 - C_HI)); RN(x*C_HI + RN(x*C_LO)) == RN(x/255)
// for all 256 inputs (exhaustively checked on the host at context creation, pfx_ctx.cpp:selfcheck_div255,
// and in tests/test_host_logic.py).
PFX_DEV float div255(float x)
{
    const float C_HI = __builtin_bit_cast(float, 998277249u);   // 0x3B808081
    const float C_LO = __builtin_bit_cast(float, 2944335615u);  // 0xAF7EFEFF
    return __builtin_fmaf(x, C_HI, x * C_LO);
}

// ---- correctly rounded f32 division, reciprocal part hoistable ------------------------------------------------
// y = v_rcp_f32(d) refined by one Newton step (2 FMA), then per numerator  q0 = n*y;  r0 = fma(-d, q0, n);
// q = fma(r0, y, q0): 3 instructions per quotient, the 3-instruction reciprocal shared by every division with the same
// denominator.  hipcc's own IEEE sequence (v_div_scale x2, rcp, 2 FMA, mul, 4 FMA, v_div_fmas, v_div_fixup) runs one more
// residual/correction round; on gfx950 that round never changes the result: tools/lab/div_exhaust.hip compares this
// sequence with `/` for ALL 2^46 pairs of f32 significands (profiles/r02_div_exhaust.json: 0 mismatches of 7.04e13; the
// same sweep with an unrefined v_rcp finds 47045, which is why the Newton step stays).  Scaling n or d by a power of two
// scales every intermediate exactly while nothing leaves the normal range and the residual stays representable
// (exponent(n) >= -100 suffices), which holds for every use in this library: numerators 0 or in [2^-100, 2^20],
// denominators in [2^-48, 2^20], checked where used.  pfx_selftest_division() re-checks 2.7e8 random pairs of the
// kernels' operand range against `/` on the device the context runs on.
struct rdiv { float d, y; };
PFX_DEV rdiv rdiv_prepare(float d)
{
    const float y0 = __builtin_amdgcn_rcpf(d);
    const float e = __builtin_fmaf(-d, y0, 1.0f);
    return {d, __builtin_fmaf(e, y0, y0)};
}
PFX_DEV float rdiv_apply(const rdiv k, float n)
{
    const float q0 = n * k.y;
    const float r0 = __builtin_fmaf(-k.d, q0, n);
    return __builtin_fmaf(r0, k.y, q0);
}
PFX_DEV float fdiv_fast(float n, float d) { return rdiv_apply(rdiv_prepare(d), n); }

// x clamped to [0, 1] for an x that is never NaN where the result is used: folds into the `clamp` output modifier of the
// instruction producing x (no instruction of its own).  Only used where one side of the clamp is provably inactive, i.e.
// in place of a one-sided fminf(x, 1) with x >= 0 or fmaxf(x, 0) with x <= 1.
PFX_DEV float clamp01(float x) { return __builtin_fminf(__builtin_fmaxf(x, 0.0f), 1.0f); }

// Rust `v.clamp(0.0, 255.0) as u8` kept as an integer-valued float (0..255), NaN -> 0.
PFX_DEV float quant255(float v)
{
    // fmaxf(NaN, 0) = 0 on AMDGPU (v_max_f32 returns the non-NaN operand), matching `NaN as u8 == 0`.
    return __builtin_fabsf(__builtin_truncf(__builtin_fminf(__builtin_fmaxf(v, 0.0f), 255.0f)));
}
// Rust `v as u8` (saturating truncation) without a prior clamp
PFX_DEV float trunc_u8f(float v) { return quant255(v); }
// Rust `v.round().clamp(0.0, 255.0) as u8` (round half away from zero)
PFX_DEV float round_u8f(float v)
{
    // roundf(x) = trunc(x + copysign(0.5, x)) is NOT exact for |x| just below .5 boundaries; use the
    // two-step form: t = trunc(x); if (|x - t| >= 0.5) t += copysign(1, x).
    float t = __builtin_truncf(v);
    float d = v - t; // exact (Sterbenz / same binade)
    if (__builtin_fabsf(d) >= 0.5f) t += __builtin_copysignf(1.0f, v);
    return __builtin_fabsf(__builtin_fminf(__builtin_fmaxf(t, 0.0f), 255.0f));
}

PFX_DEV float ubyte0(uint32_t p) { return (float)(p & 0xffu); }
PFX_DEV float ubyte1(uint32_t p) { return (float)((p >> 8) & 0xffu); }
PFX_DEV float ubyte2(uint32_t p) { return (float)((p >> 16) & 0xffu); }
PFX_DEV float ubyte3(uint32_t p) { return (float)(p >> 24); }

PFX_DEV uint32_t pack_rgba(float r, float g, float b, float a)
{
    return (uint32_t)r | ((uint32_t)g << 8) | ((uint32_t)b << 16) | ((uint32_t)a << 24);
}

// Four channels of Rust's `v.round().clamp(0.0, 255.0) as u8` packed into one pixel in 12 instructions (round_u8f + pack_rgba: ~40).
// v_cvt_pk_u8_f32 rounds to nearest-EVEN and saturates to [0, 255]; setting the lowest significand bit first turns every exact tie
// k + 0.5 (whose low bit is 0) into the next float above it and moves no other value across a tie, so the conversion then rounds half
// away from zero.  v_med3_f32(v, -1, 300) in front keeps +inf from becoming a NaN under the OR and sends NaN to -1 (-> 0, like
// `NaN as u8`).  Checked against the step-by-step formula for all 2^32 bit patterns on gfx950 (tools/lab/round_exhaust.hip,
// profiles/r02_round_exhaust.json): identical for every pattern except signalling NaNs, which no arithmetic instruction produces.
PFX_DEV float round_tie_prep(float v)
{
    return __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, __builtin_amdgcn_fmed3f(v, -1.0f, 300.0f)) | 1u);
}
PFX_DEV uint32_t pack_round_rgba(float r, float g, float b, float a)
{
    uint32_t p = __builtin_amdgcn_cvt_pk_u8_f32(round_tie_prep(r), 0, 0u);
    p = __builtin_amdgcn_cvt_pk_u8_f32(round_tie_prep(g), 1, p);
    p = __builtin_amdgcn_cvt_pk_u8_f32(round_tie_prep(b), 2, p);
    return __builtin_amdgcn_cvt_pk_u8_f32(round_tie_prep(a), 3, p);
}
// the same for values known to be finite (no med3: 8 instructions)
PFX_DEV uint32_t pack_round_rgba_finite(float r, float g, float b, float a)
{
    auto tie = [](float v) { return __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, v) | 1u); };
    uint32_t p = __builtin_amdgcn_cvt_pk_u8_f32(tie(r), 0, 0u);
    p = __builtin_amdgcn_cvt_pk_u8_f32(tie(g), 1, p);
    p = __builtin_amdgcn_cvt_pk_u8_f32(tie(b), 2, p);
    return __builtin_amdgcn_cvt_pk_u8_f32(tie(a), 3, p);
}

// Rust f32::clamp
PFX_DEV float rs_clamp(float x, float lo, float hi)
{
    if (x < lo) x = lo;
    if (x > hi) x = hi;
    return x;
}

// XCD-aware block remap: blocks are dealt round-robin to the 8 XCDs (block b -> XCD b % 8), so give each
// XCD a contiguous range of tiles: neighbouring tiles (shared halo rows/columns) then hit the same L2.
PFX_DEV uint32_t xcd_swizzle(uint32_t bid, uint32_t nblocks)
{
    const uint32_t NXCD = 8;
    uint32_t per = nblocks / NXCD;
    uint32_t main = per * NXCD;
    if (bid >= main) return bid; // ragged tail keeps its position
    return (bid % NXCD) * per + (bid / NXCD);
}


// hipFuncAttributeMaxDynamicSharedMemorySize belongs to a (kernel, device) pair: a launcher keeps one `lds_grant` per kernel instantiation (a function-local
// static of the templated launch lambda) and asks the runtime only when the pair needs more than it was granted before — not on every launch.
struct lds_grant { std::atomic<uint32_t> bytes[32]; };
inline hipError_t grant_lds(lds_grant& g, const void* kernel, size_t lds)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e) return e;
    const bool slot = dev >= 0 && dev < 32;
    if (slot && g.bytes[dev].load(std::memory_order_relaxed) >= lds) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e) return e;
    if (slot) g.bytes[dev].store((uint32_t)lds, std::memory_order_relaxed);
    return hipSuccess;
}

// the same keyed by the kernel's address, for launchers whose kernel is a run-time choice among instantiations of one signature (a function-local static would be
// shared by all of them): a small table under a spin lock — a few compares per launch instead of a runtime call
inline hipError_t grant_lds_for(const void* kernel, size_t lds)
{
    struct slot { const void* k; int dev; uint32_t bytes; };
    static slot table[256];
    static std::atomic<int> n{0};
    static std::atomic_flag busy = ATOMIC_FLAG_INIT;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e) return e;
    const int have = n.load(std::memory_order_acquire);
    for (int i = 0; i < have; ++i)
        if (table[i].k == kernel && table[i].dev == dev && table[i].bytes >= lds) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e) return e;
    while (busy.test_and_set(std::memory_order_acquire)) {}
    const int cnt = n.load(std::memory_order_relaxed);
    int at = -1;
    for (int i = 0; i < cnt; ++i) if (table[i].k == kernel && table[i].dev == dev) at = i;
    if (at >= 0) table[at].bytes = std::max(table[at].bytes, (uint32_t)lds);   // entries only grow: a reader that sees the old value just asks the runtime again
    else if (cnt < 256) { table[cnt] = {kernel, dev, (uint32_t)lds}; n.store(cnt + 1, std::memory_order_release); }
    busy.clear(std::memory_order_release);
    return hipSuccess;
}

} // namespace pfxk
