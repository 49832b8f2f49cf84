// k_script.hip — device side of the script front-end:
//   * vm_kernel: executes the register bytecode of a per-pixel Rhai closure (pfx_rhai.h, BcOp) — one lane per pixel, the
//     lane's registers in LDS ([reg][lane], conflict-free), per-lane program counter so data-dependent `if` / loops diverge
//     like any other SIMT code.  Semantics of the reference's bulk iterators (src/ops/scripting.rs:442-609): the closure
//     sees the pre-call image, an array result of >= 4 elements updates the pixel with clamp(0,255) per integer element,
//     anything else leaves the pixel; the first failing pixel in row-major order reports the error.
//   * the pixel-moving host functions of the transform / selection API: flips, rotations, canvas resize, selection masks,
//     fill / delete selected (src/ops/scripting.rs:640-819, 1356-1481).
// All integer work; 4 B read + 4 B written per pixel plus the bytecode's arithmetic.
#include "k_common.h"
#include "pfx_kernels.h"
#include "pfx_rhai.h"

using namespace pfxk;
using namespace rhai;

namespace {

PFX_DEV double as_f(uint64_t b) { return __builtin_bit_cast(double, b); }
PFX_DEV uint64_t f_bits(double d) { return __builtin_bit_cast(uint64_t, d); }

// LCODE: the program (12 bytes per instruction) is staged in LDS behind the register file — instruction fetch is then an LDS
// read instead of a dependent global load per bytecode instruction, which is what the interpreter's time went into.
// A workgroup walks over pixels with a grid stride so the staged program serves many pixels.
// HEAVY: the instantiation that carries the f64 libm routines (fmod, pow, sin, cos, tan, atan2, exp, log: 128 VGPRs and scratch);
// programs without them run the light instantiation at twice the occupancy.
template <bool LCODE, bool HEAVY>
__global__ void vm_kernel(const pfxk_vm_args A)
{
    extern __shared__ uint64_t regs[];
    const int lanes = blockDim.x, t = threadIdx.x;
    const BcIns* __restrict__ gcode = (const BcIns*)A.code;
    BcIns* lcode = reinterpret_cast<BcIns*>(regs + (size_t)lanes * A.n_regs);
    if constexpr (LCODE) {
        const int words = A.n_code * 3; // 12-byte instructions
        for (int i = t; i < words; i += lanes) reinterpret_cast<uint32_t*>(lcode)[i] = reinterpret_cast<const uint32_t*>(gcode)[i];
        __syncthreads();
    }
    auto fetch = [&](int at) -> BcIns {
        if constexpr (LCODE) return lcode[at];
        else return gcode[at];
    };
    const int rw = A.x1 - A.x0;
    const long long total = (long long)rw * (A.y1 - A.y0);
    const uint64_t* __restrict__ K = A.consts;
#define R(i) regs[(size_t)(i) * lanes + t]
  for (int i = 0; i < A.n_pre; ++i) { const BcIns I = fetch(i); R(I.dst) = K[I.a]; }   // the hoisted constants: once per lane, the registers are nothing else's
  // a region holds at most 256 M pixels (pfx_dims_ok): its row-major index fits 32 bits, and so do its division by the row length and the grid stride's sum
  for (uint32_t local = blockIdx.x * (uint32_t)lanes + (uint32_t)t; local < (uint32_t)total; local += gridDim.x * (uint32_t)lanes) {
    // a pixel behind an already recorded failure cannot become the first failing pixel: skip it (the result is discarded anyway)
    if ((unsigned long long)local > (__atomic_load_n(A.err, __ATOMIC_RELAXED) >> 24)) continue;
    const uint32_t row = local / (uint32_t)rw;
    const int x = A.x0 + (int)(local - row * (uint32_t)rw), y = A.y0 + (int)row;
    const size_t pi = (size_t)y * A.w + x;
    const uint32_t px = A.src[pi];
    {
        int p = 0;
        if (A.n_params == 6) { R(0) = (uint64_t)(int64_t)x; R(1) = (uint64_t)(int64_t)y; p = 2; }
        R(p) = px & 0xffu; R(p + 1) = (px >> 8) & 0xffu; R(p + 2) = (px >> 16) & 0xffu; R(p + 3) = px >> 24;
    }
    int pc = A.n_pre;
    uint32_t steps = 0;
    int err = 0, err_line = 0;
    uint32_t out = px;
    bool done = false;
    // One bytecode instruction.  Inlined twice below: with a wave-uniform instruction (fields in SGPRs: scalar fetch, scalar
    // jump table, uniform LDS offsets) and with a per-lane one (lanes of a wave at different program counters).
    auto step = [&](const BcIns I) {
        if (++steps > A.step_budget) { err = BCE_TOO_MANY_OPS; return; }
        const int64_t ia = (int64_t)R(I.a), ib = (int64_t)R(I.b);
        int64_t ir;
        switch (I.op) {
        case BC_LOADK: R(I.dst) = K[I.a]; break;
        case BC_MOV: R(I.dst) = R(I.a); break;
        case BC_IADD: if (__builtin_add_overflow(ia, ib, &ir)) { err = BCE_ADD_OVERFLOW; } R(I.dst) = (uint64_t)ir; break;
        case BC_ISUB: if (__builtin_sub_overflow(ia, ib, &ir)) { err = BCE_SUB_OVERFLOW; } R(I.dst) = (uint64_t)ir; break;
        case BC_IMUL: if (__builtin_mul_overflow(ia, ib, &ir)) { err = BCE_MUL_OVERFLOW; } R(I.dst) = (uint64_t)ir; break;
        // A 64-bit division is ~250 instructions on this chip and pixel arithmetic never needs it: operands that are both non-negative and below 2^31 (channel
        // values, coordinates, small constants) divide as 32-bit unsigned numbers — the same quotient and remainder, a tenth of the instructions.
        case BC_IDIV:
            if (ib == 0) err = BCE_DIV_ZERO;
            else if ((((uint64_t)ia | (uint64_t)ib) >> 31) == 0) R(I.dst) = (uint64_t)((uint32_t)ia / (uint32_t)ib);
            else if (ia == INT64_MIN && ib == -1) err = BCE_DIV_OVERFLOW;
            else R(I.dst) = (uint64_t)(ia / ib);
            break;
        case BC_IMOD:
            if (ib == 0) err = BCE_MOD_ZERO;
            else if ((((uint64_t)ia | (uint64_t)ib) >> 31) == 0) R(I.dst) = (uint64_t)((uint32_t)ia % (uint32_t)ib);
            else if (ia == INT64_MIN && ib == -1) err = BCE_DIV_OVERFLOW;
            else R(I.dst) = (uint64_t)(ia % ib);
            break;
        case BC_INEG: if (ia == INT64_MIN) err = BCE_NEG_OVERFLOW; else R(I.dst) = (uint64_t)(-ia); break;
        case BC_IPOW: {
            if (ib < 0) { err = BCE_POW_NEGATIVE; break; }
            int64_t r = 1, base = ia, e = ib;
            while (e > 0 && !err) {
                if (e & 1) { if (__builtin_mul_overflow(r, base, &r)) err = BCE_POW_OVERFLOW; }
                e >>= 1;
                if (e > 0 && !err && __builtin_mul_overflow(base, base, &base)) err = BCE_POW_OVERFLOW;
            }
            R(I.dst) = (uint64_t)r;
            break;
        }
        case BC_IAND: R(I.dst) = (uint64_t)(ia & ib); break;
        case BC_IOR: R(I.dst) = (uint64_t)(ia | ib); break;
        case BC_IXOR: R(I.dst) = (uint64_t)(ia ^ ib); break;
        case BC_ISHL:
        case BC_ISHR: {
            bool left = I.op == BC_ISHL;
            int64_t n = ib;
            if (n < 0) { left = !left; n = (n == INT64_MIN) ? 64 : -n; }
            R(I.dst) = left ? (n >= 64 ? 0ull : ((uint64_t)ia << n)) : (uint64_t)(n >= 64 ? (ia < 0 ? -1 : 0) : (ia >> n));
            break;
        }
        case BC_IABS: R(I.dst) = (uint64_t)(ia < 0 ? (int64_t)(0ull - (uint64_t)ia) : ia); break;
        case BC_ISIGN: R(I.dst) = (uint64_t)(int64_t)(ia > 0 ? 1 : (ia < 0 ? -1 : 0)); break;
        case BC_IMIN: R(I.dst) = (uint64_t)(ia < ib ? ia : ib); break;
        case BC_IMAX: R(I.dst) = (uint64_t)(ia > ib ? ia : ib); break;
        case BC_ICLAMP: { const int64_t hi = (int64_t)R(I.c); int64_t v = ia; if (v < ib) v = ib; if (v > hi) v = hi; R(I.dst) = (uint64_t)v; break; }
        case BC_IEQ: R(I.dst) = ia == ib; break;
        case BC_INE: R(I.dst) = ia != ib; break;
        case BC_ILT: R(I.dst) = ia < ib; break;
        case BC_ILE: R(I.dst) = ia <= ib; break;
        case BC_IGT: R(I.dst) = ia > ib; break;
        case BC_IGE: R(I.dst) = ia >= ib; break;
        case BC_FADD: R(I.dst) = f_bits(as_f(R(I.a)) + as_f(R(I.b))); break;
        case BC_FSUB: R(I.dst) = f_bits(as_f(R(I.a)) - as_f(R(I.b))); break;
        case BC_FMUL: R(I.dst) = f_bits(as_f(R(I.a)) * as_f(R(I.b))); break;
        case BC_FDIV: R(I.dst) = f_bits(as_f(R(I.a)) / as_f(R(I.b))); break;
        case BC_FMOD: if constexpr (HEAVY) { R(I.dst) = f_bits(fmod(as_f(R(I.a)), as_f(R(I.b)))); } else err = 255; break;
        case BC_FNEG: R(I.dst) = R(I.a) ^ 0x8000000000000000ull; break;
        case BC_FPOW: if constexpr (HEAVY) { R(I.dst) = f_bits(pow(as_f(R(I.a)), as_f(R(I.b)))); } else err = 255; break;
        case BC_FABS: R(I.dst) = R(I.a) & 0x7fffffffffffffffull; break;
        case BC_FMIN: R(I.dst) = f_bits(fmin(as_f(R(I.a)), as_f(R(I.b)))); break;
        case BC_FMAX: R(I.dst) = f_bits(fmax(as_f(R(I.a)), as_f(R(I.b)))); break;
        case BC_FCLAMP: { double v = as_f(R(I.a)); const double lo = as_f(R(I.b)), hi = as_f(R(I.c)); if (v < lo) v = lo; if (v > hi) v = hi; R(I.dst) = f_bits(v); break; }
        case BC_FFLOOR: R(I.dst) = f_bits(floor(as_f(R(I.a)))); break;
        case BC_FCEIL: R(I.dst) = f_bits(ceil(as_f(R(I.a)))); break;
        case BC_FROUND: R(I.dst) = f_bits(round(as_f(R(I.a)))); break;
        case BC_FSQRT: R(I.dst) = f_bits(sqrt(as_f(R(I.a)))); break;
        case BC_FSIN: if constexpr (HEAVY) { R(I.dst) = f_bits(sin(as_f(R(I.a)))); } else err = 255; break;
        case BC_FCOS: if constexpr (HEAVY) { R(I.dst) = f_bits(cos(as_f(R(I.a)))); } else err = 255; break;
        case BC_FTAN: if constexpr (HEAVY) { R(I.dst) = f_bits(tan(as_f(R(I.a)))); } else err = 255; break;
        case BC_FATAN2: if constexpr (HEAVY) { R(I.dst) = f_bits(atan2(as_f(R(I.a)), as_f(R(I.b)))); } else err = 255; break;
        case BC_FEXP: if constexpr (HEAVY) { R(I.dst) = f_bits(exp(as_f(R(I.a)))); } else err = 255; break;
        case BC_FLN: if constexpr (HEAVY) { R(I.dst) = f_bits(log(as_f(R(I.a)))); } else err = 255; break;
        case BC_FLERP: { const double a = as_f(R(I.a)), b = as_f(R(I.b)), tt = as_f(R(I.c)); R(I.dst) = f_bits(a + (b - a) * tt); break; }
        case BC_FDIST: { // scripting.rs:1266: ((x2-x1)^2 + (y2-y1)^2).sqrt(), operands in 4 consecutive registers
            const double x1 = as_f(R(I.a)), y1 = as_f(R(I.a + 1)), x2 = as_f(R(I.a + 2)), y2 = as_f(R(I.a + 3));
            R(I.dst) = f_bits(sqrt((x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1)));
            break;
        }
        case BC_FEQ: R(I.dst) = as_f(R(I.a)) == as_f(R(I.b)); break;
        case BC_FNE: R(I.dst) = as_f(R(I.a)) != as_f(R(I.b)); break;
        case BC_FLT: R(I.dst) = as_f(R(I.a)) < as_f(R(I.b)); break;
        case BC_FLE: R(I.dst) = as_f(R(I.a)) <= as_f(R(I.b)); break;
        case BC_FGT: R(I.dst) = as_f(R(I.a)) > as_f(R(I.b)); break;
        case BC_FGE: R(I.dst) = as_f(R(I.a)) >= as_f(R(I.b)); break;
        case BC_I2F: R(I.dst) = f_bits((double)ia); break;
        case BC_F2I: {
            const double f = as_f(R(I.a));
            if (!(f > -9223372036854775809.0 && f < 9223372036854775808.0)) err = BCE_F2I_RANGE;
            else R(I.dst) = (uint64_t)(int64_t)f;
            break;
        }
        case BC_NOT: R(I.dst) = R(I.a) == 0; break;
        case BC_JMP: pc = I.a; break;
        case BC_JZ: if (R(I.b) == 0) pc = I.a; break;
        case BC_JNZ: if (R(I.b) != 0) pc = I.a; break;
        case BC_GETCH: { // get_pixel / get_r..a: 0 outside the image (scripting.rs:360-383,408-419)
            uint64_t v = 0;
            if (ia >= 0 && ib >= 0 && ia < A.w && ib < A.h) v = (A.src[(size_t)ib * A.w + (size_t)ia] >> (8 * I.c)) & 0xffu;
            R(I.dst) = v;
            break;
        }
        case BC_ISSEL: { // scripting.rs:337-348
            uint64_t v = 0;
            if (ia >= 0 && ib >= 0 && ia < A.w && ib < A.h) v = A.mask ? (A.mask[(size_t)ib * A.w + (size_t)ia] > 0) : 1;
            R(I.dst) = v;
            break;
        }
        case BC_RET_ARR: { // `arr[k].as_int().unwrap_or(old).clamp(0, 255) as u8`
            uint32_t o = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                uint32_t c = (px >> (8 * k)) & 0xffu;
                if (I.b & (1 << k)) { int64_t v = (int64_t)R(I.a + k); v = v < 0 ? 0 : (v > 255 ? 255 : v); c = (uint32_t)v; }
                o |= c << (8 * k);
            }
            out = o;
            done = true;
            break;
        }
        case BC_RET_UNIT: done = true; break;
        case BC_ERR: err = I.a; break;
        default: err = 255; break;
        }
    };
    BcIns pre{};
    int pre_pc = -1;
    for (;;) {
        if (pc < 0 || pc >= A.n_code) break; // falling off the end = unit
        int upc = __builtin_amdgcn_readfirstlane(pc);
        if (__all(pc == upc)) {
            // Converged — the usual case: straight-line closures and branches every lane takes the same way.  The wave stays in this inner loop, whose
            // condition is scalar (no exec-mask bookkeeping per instruction: the interpreter is bound by the CU's one scalar unit), until a branch splits
            // the lanes, one of them fails or the closure returns; the instruction's fields sit in SGPRs and the next one is requested before this one runs.
            bool stay;
            do {
                const BcIns I = (pre_pc == upc) ? pre : fetch(upc);
                pre_pc = min(upc + 1, A.n_code - 1);
                pre = fetch(pre_pc);
                pc = upc + 1;
                step(I);
                if (err) err_line = I.line;
                bool together = true;
                if (I.op == BC_JMP || I.op == BC_JZ || I.op == BC_JNZ) {   // only a jump moves a lane's pc anywhere but to the next instruction (scalar test: I is in SGPRs)
                    upc = __builtin_amdgcn_readfirstlane(pc);
                    together = __all(pc == upc);
                } else ++upc;
                stay = together && !__any((err != 0) | done) && upc >= 0 && upc < A.n_code;
            } while (stay);
            if (err || done) break;
            continue;
        }
        const BcIns I = fetch(pc++);   // lanes of the wave at different program counters: per-lane fetch
        step(I);
        if (err) { err_line = I.line; break; }
        if (done) break;
    }
    if (err) {
        const unsigned long long local_row_major = (unsigned long long)(y - A.y0) * (unsigned long long)rw + (unsigned long long)(x - A.x0);
        atomicMin(A.err, (local_row_major << 24) | ((unsigned long long)(err & 0xff) << 16) | (unsigned long long)(err_line & 0xffff));
        continue;
    }
    A.dst[pi] = out;
  }
#undef R
}

// ---- transforms (imageops::flip_horizontal / flip_vertical / rotate180 / rotate90 / rotate270: pure permutations) ----
// mode: 0 flip_h, 1 flip_v, 2 rotate180, 3 rotate90 (cw), 4 rotate270 (ccw).  (w, h) = SOURCE size.
__global__ __launch_bounds__(256) void permute_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, int mode, int w, int h)
{
    // iterate over destination pixels so that stores are coalesced
    const int dw = mode >= 3 ? h : w, dh = mode >= 3 ? w : h;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= dw || y >= dh) return;
    int sx, sy;
    switch (mode) {
    case 0: sx = w - 1 - x; sy = y; break;
    case 1: sx = x; sy = h - 1 - y; break;
    case 2: sx = w - 1 - x; sy = h - 1 - y; break;
    case 3: sx = y; sy = h - 1 - x; break;  // rotate90: out(h-1-sy, sx) = in(sx, sy)
    default: sx = w - 1 - y; sy = x; break; // rotate270: out(sy, w-1-sx) = in(sx, sy)
    }
    dst[(size_t)y * dw + x] = src[(size_t)sy * w + sx];
}

// resize_canvas (scripting.rs:798-812): zeroed new image, old pixels copied at (offset_x, offset_y)
__global__ __launch_bounds__(256) void recanvas_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, int ow, int oh, int nw, int nh,
                                                       int off_x, int off_y, uint32_t fill)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= nw || y >= nh) return;
    const int sx = x - off_x, sy = y - off_y;
    dst[(size_t)y * nw + x] = (sx >= 0 && sy >= 0 && sx < ow && sy < oh) ? src[(size_t)sy * ow + sx] : fill;
}

// selection masks (scripting.rs:1359-1431).  op: 0 rect [x0,x1)x[y0,y1), 1 ellipse (f64 like the reference), 2 invert, 3 clear to 0
__global__ __launch_bounds__(256) void mask_kernel(uint8_t* __restrict__ mask, int op, int x0, int y0, int x1, int y1, double cx, double cy, double rx2,
                                                   double ry2, int w, int h)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const size_t i = (size_t)y * w + x;
    if (op == 0) mask[i] = (x >= x0 && x < x1 && y >= y0 && y < y1) ? 255 : 0;
    else if (op == 1) {
        const double dx = (double)x - cx, dy = (double)y - cy;
        mask[i] = ((dx * dx) / rx2 + (dy * dy) / ry2 <= 1.0) ? 255 : 0;
    } else if (op == 2) mask[i] = (uint8_t)(255 - mask[i]);
    else mask[i] = 0;
}

// fill_selected / delete_selected (scripting.rs:1435-1481)
__global__ __launch_bounds__(256) void fill_kernel(uint32_t* __restrict__ img, const uint8_t* __restrict__ mask, uint32_t rgba, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        if (!mask || mask[i] > 0) img[i] = rgba;
}

dim3 tile_grid(int w, int h) { return dim3((w + 63) / 64, (h + 3) / 4); }

} // namespace

extern "C" hipError_t pfxk_vm_run(hipStream_t s, const pfxk_vm_args* A)
{
    const long long n = (long long)(A->x1 - A->x0) * (A->y1 - A->y0);
    if (n <= 0) return hipSuccess;
    int lanes = (int)(65536 / ((size_t)A->n_regs * 8));
    lanes = lanes >= 256 ? 256 : (lanes / 64) * 64;
    if (lanes < 64) return hipErrorInvalidValue; // > 128 registers: rejected by the compiler before we get here
    const size_t lds_regs = (size_t)lanes * A->n_regs * 8, lds_code = ((size_t)A->n_code * 12 + 15) & ~(size_t)15;
    long long blocks = (n + lanes - 1) / lanes;
    if (blocks > 256 * 32) blocks = 256 * 32; // grid stride beyond that: the staged program is reused
    const bool lcode = lds_regs + lds_code <= 65536;
    const size_t lds = lcode ? lds_regs + lds_code : lds_regs;
    const uint32_t g = (uint32_t)blocks;
    if (A->heavy) { if (lcode) vm_kernel<true, true><<<g, lanes, lds, s>>>(*A); else vm_kernel<false, true><<<g, lanes, lds, s>>>(*A); }
    else          { if (lcode) vm_kernel<true, false><<<g, lanes, lds, s>>>(*A); else vm_kernel<false, false><<<g, lanes, lds, s>>>(*A); }
    return hipGetLastError();
}

extern "C" hipError_t pfxk_permute(hipStream_t s, const uint8_t* d_src, uint8_t* d_dst, int mode, uint32_t w, uint32_t h)
{
    if (w == 0 || h == 0) return hipSuccess;
    const int dw = mode >= 3 ? (int)h : (int)w, dh = mode >= 3 ? (int)w : (int)h;
    permute_kernel<<<tile_grid(dw, dh), 256, 0, s>>>((const uint32_t*)d_src, (uint32_t*)d_dst, mode, (int)w, (int)h);
    return hipGetLastError();
}

extern "C" hipError_t pfxk_recanvas(hipStream_t s, const uint8_t* d_src, uint8_t* d_dst, uint32_t ow, uint32_t oh, uint32_t nw, uint32_t nh, int off_x,
                                    int off_y, uint32_t fill_rgba)
{
    if (nw == 0 || nh == 0) return hipSuccess;
    recanvas_kernel<<<tile_grid((int)nw, (int)nh), 256, 0, s>>>((const uint32_t*)d_src, (uint32_t*)d_dst, (int)ow, (int)oh, (int)nw, (int)nh, off_x, off_y,
                                                                fill_rgba);
    return hipGetLastError();
}

extern "C" hipError_t pfxk_mask_op(hipStream_t s, uint8_t* d_mask, int op, int x0, int y0, int x1, int y1, double cx, double cy, double rx2, double ry2,
                                   uint32_t w, uint32_t h)
{
    if (w == 0 || h == 0) return hipSuccess;
    mask_kernel<<<tile_grid((int)w, (int)h), 256, 0, s>>>(d_mask, op, x0, y0, x1, y1, cx, cy, rx2, ry2, (int)w, (int)h);
    return hipGetLastError();
}

extern "C" hipError_t pfxk_fill_masked(hipStream_t s, uint8_t* d_img, const uint8_t* d_mask, uint32_t rgba, uint32_t w, uint32_t h)
{
    const size_t n = (size_t)w * h;
    if (n == 0) return hipSuccess;
    size_t blocks = (n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    fill_kernel<<<(uint32_t)blocks, 256, 0, s>>>((uint32_t*)d_img, d_mask, rgba, n);
    return hipGetLastError();
}
