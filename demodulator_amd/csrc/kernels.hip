// kernels.hip -- hand-written CDNA4 (gfx950) kernels of the FM-demodulation hot
// path.  Compiled with -ffp-contract=off: every fused multiply-add below is an
// explicit fmaf(), everything else rounds once per operation.
//
// demod_rows   the fused streaming kernel: uint8 IQ -> unpack -> [complex
//              look-ahead stencil] -> conjugate-multiply discriminator + atan2
//              -> real look-ahead stencil -> float32.  One wavefront streams a
//              segment of 1 KiB rows from its end to its start, so the
//              look-ahead values a row needs from the row after it have been
//              computed already; they and the next lane's values inside a row
//              travel through a per-wavefront exchange line in LDS (no
//              barriers).  16-byte loads and stores, 4 B of HBM traffic per IQ
//              sample.
//              The wavefront that owns a chunk's first row also replays the
//              reference's sectioned recurrences for the first few outputs of
//              the chunk (partial sections, overflow pollution; SURVEY.md
//              Appendix A.4), one output per lane (chunk_head).
// demod_heads  the same replay, one thread per chunk: -q1, the default designs
//              with chunks below 256 KiB (where a kernel of its own is cheaper)
//              and in a bounds-checked (ragged) launch.
// demod_heads_wave  chunk_head with room for 8 sections per stage and for -q2 in
//              front of -L, one wavefront per chunk: designs of degree 9-16.
// demod_tiny   literal whole-chunk path for chunks shorter than one row.
// correct_iq_offsets   -q1: the running DC offsets of correctIq, replayed piece by piece
//              after a run-in, into a table the other kernels subtract from the bytes.
//
// Reference semantics restated here (never copied):
//   unpack         /root/reference/src/matrix.c:82-98, :100-118, :120-140
//   discriminator  src/matrix.c:159-176
//   filters        src/filter.c:212-231 (real), :233-259 (complex)
//   chunk loop     src/matrix.c:236-273
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
#include <cstdint>
#include <type_traits>

#include "kernels.h"

namespace demod {
namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// Packed f32: two independent values in an aligned register pair, one v_pk_mul_f32 /
// v_pk_fma_f32 for both -- per element the very operations (and roundings) of the scalar
// forms.  Used where the data comes in pairs by nature: (re, im) of a sample, and two
// neighbouring outputs through the atan2 polynomial.  A quarter fewer VALU instructions
// per row; the SLP vectoriser stays off (it pairs values that are not adjacent and pays
// for it in moves).
typedef float pf2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ pf2 fma2(pf2 a, pf2 b, pf2 c) { return __builtin_elementwise_fma(a, b, c); }

#ifndef DEMOD_LOAD_POLICY
#define DEMOD_LOAD_POLICY " nt"   // cache policy suffix of the streaming loads: every byte is read once
#endif
// 4 independent wavefronts per workgroup: +0.5-0.8 % over 8 on two boxes after the packed-f32
// diet (2, 1: -3 %), profiles/r2_ab_workgroup.txt
#ifndef DEMOD_WAVES_PER_BLOCK
#define DEMOD_WAVES_PER_BLOCK 4
#endif
constexpr int kWavesPerBlock = DEMOD_WAVES_PER_BLOCK;  // independent wavefronts per workgroup of demod_rows
// Workgroups are handed to the 8 XCDs round-robin.  With 8, the workgroups of a group of
// eight take interleaved segments (XCD x streams segments x, x+8, x+16, ... of the group's
// window) instead of eight contiguous 32 KiB pieces: +2-4 % on every box tried
// (profiles/r1_ab_rows_per_wave.txt); 0 = plain order.
#ifndef DEMOD_XCD_INTERLEAVE
#define DEMOD_XCD_INTERLEAVE 8
#endif
#ifndef DEMOD_ROWS_PER_WAVE
#define DEMOD_ROWS_PER_WAVE 4
#endif
constexpr int kGroup = DEMOD_ROWS_PER_WAVE;  // rows (1 KiB each) of a segment: what one wavefront owns
// Rows a wavefront keeps in flight: a segment's first kFlight rows are asked for up front, row k + kFlight when row k
// has been computed and stored (into the registers row k's bytes arrived in).  kFlight == kGroup: everything up front.
#ifndef DEMOD_LOADS_IN_FLIGHT
#define DEMOD_LOADS_IN_FLIGHT (DEMOD_ROWS_PER_WAVE < 4 ? DEMOD_ROWS_PER_WAVE : 4)
#endif
constexpr int kFlight = DEMOD_LOADS_IN_FLIGHT;
static_assert(kFlight >= 1 && kFlight <= kGroup && kGroup <= 8, "segment shape");
// Vector-memory operations issued after row k's load and before row k is needed (loads and stores retire in issue
// order: waiting until only that many are outstanding is waiting for row k's bytes and for nothing younger).  The
// sequence is: look-ahead load, loads of rows 0 .. kFlight-1, then per row k: [wait] compute, store k, load k + kFlight
// (while there is one).
constexpr int younger_than_row(int k) {
    int n = 0;
    if (k < kFlight) {
        n = kFlight - 1 - k;
        for (int j = 0; j < k; ++j) n += 1 + (j + kFlight < kGroup ? 1 : 0);
    } else {
        for (int j = k - kFlight + 1; j < k; ++j) n += 1 + (j + kFlight < kGroup ? 1 : 0);
    }
    return n;
}
// 16 bytes per lane of a row: scalar base + this lane's offset (issued from inline asm, invisible to hipcc's own wait counting)
__device__ __forceinline__ void ask_row(u32x4 &q, uint32_t lane_off, const uint8_t *row) {
    asm volatile("global_load_dwordx4 %0, %1, %2" DEMOD_LOAD_POLICY : "=v"(q) : "v"(lane_off), "s"(row) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm(u32x4 &q) {
    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(q) : "i"(N) : "memory");
}


// ---------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------

// Lane i receives the value lane i + 1 holds; lane 63 receives `halo`.
// One v_mov_b32 with DPP wave_shl:1 (gfx9 family); `halo` is the DPP "old"
// operand, kept by the lane that has no source.
__device__ __forceinline__ float from_next_lane(float v, float halo) {
    const int r = __builtin_amdgcn_update_dpp(__float_as_int(halo), __float_as_int(v), 0x130, 0xF,
                                              0xF, false);
    return __int_as_float(r);
}

__device__ __forceinline__ float wave_first(float v) {
    return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}

// v_mul_legacy_f32 (0 * x = 0 for every x, infinities included) through its LLVM
// intrinsic, so that hipcc schedules it and pads its hazards like any other
// instruction (an inline-asm consumer right behind v_rcp_f32 read a stale value).
extern "C" __device__ float demod_llvm_fmul_legacy(float, float) __asm("llvm.amdgcn.fmul.legacy");
__device__ __forceinline__ float fmul_legacy(float a, float b) { return demod_llvm_fmul_legacy(a, b); }

// a ? b : c, bit by bit: one v_bitop3_b32 (full-rate on gfx950; v_bfi_b32,
// v_cndmask_b32, v_min/v_max and v_cmp issue at roughly half that rate --
// tools/microbench/valu_rate.hip).
__device__ __forceinline__ uint32_t bit_select(uint32_t a, uint32_t b, uint32_t c) {
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0xCA);
}

// atan2(y, x) for FINITE arguments (guaranteed by plan.cpp::plan_magnitude_bound: a design
// that could overflow is refused at context creation) with IEEE signed-zero behaviour:
//   atan2(+-0, x >= +0) = +-0, atan2(+-0, x <= -0) = +-pi  (SURVEY.md Appendix A.2).
// Octant reduction to q = min/max in [0, 1], odd minimax polynomial q*P(q^2)
// (8 coefficients, |error| < 7e-8 before rounding), one v_rcp_f32.  The octant
// logic is written with sign-bit masks instead of compares and selects so that
// everything except the reciprocal is a full-rate VALU operation.
constexpr float kAtanLead = -4.054567311e-03f;  // leading coefficient of the polynomial below
__device__ __forceinline__ float fast_atan2(float y, float x, float lead = kAtanLead) {
    const uint32_t xb = __float_as_uint(x), yb = __float_as_uint(y);
    const float t = fabsf(x) - fabsf(y);
    const uint32_t swap = (uint32_t) (__float_as_int(t) >> 31);  // all ones when |y| > |x|
    const float hi = fabsf(__uint_as_float(bit_select(swap, yb, xb)));
    const float lo = fabsf(__uint_as_float(bit_select(swap, xb, yb)));
    // v_mul_legacy_f32: 0 * anything = 0, so atan2(0, 0) (lo = 0, 1/hi = inf) gives q = 0
    const float q = fmul_legacy(lo, __builtin_amdgcn_rcpf(hi));
    const float s = q * q;
    // `lead` is kAtanLead; the streaming kernel hands it in from a register it keeps
    // for the whole segment (gfx9 encodings take one literal or scalar per instruction,
    // so "c7 * s + c6" would otherwise start with a v_mov of c7 for every output)
    float p = fmaf(lead, s, 2.186295763e-02f);
    p = fmaf(p, s, -5.591232702e-02f);
    p = fmaf(p, s, 9.642197192e-02f);
    p = fmaf(p, s, -1.390862912e-01f);
    p = fmaf(p, s, 1.994656622e-01f);
    p = fmaf(p, s, -3.332985938e-01f);
    p = fmaf(p, s, 9.999993443e-01f);
    // r = atan(lo / hi) in [0, pi/4];  |y| > |x|:  pi/2 - r  ==  |r - pi/2|, the
    // subtraction riding on the polynomial's last multiply
    float r = fmaf(p, q, __uint_as_float(swap & 0xBFC90FDBu));
    // x negative (sign bit, so -0 counts):  pi - r'  ==  |r' - pi|
    const uint32_t xneg = (uint32_t) ((int32_t) xb >> 31);
    r = fabsf(r) - __uint_as_float(xneg & 0x40490FDBu);
    // magnitude of r, sign of y
    return __uint_as_float(bit_select(0x7FFFFFFFu, __float_as_uint(r), yb));
}

// Two atan2 at once: the octant logic per element as in fast_atan2, the polynomial on the
// pair (v_pk_mul_f32 / v_pk_fma_f32: the same operations on the same values).
__device__ __forceinline__ void fast_atan2_pair(float y0, float x0, float y1, float x1, float lead, float &r0, float &r1) {
    const uint32_t xb0 = __float_as_uint(x0), yb0 = __float_as_uint(y0), xb1 = __float_as_uint(x1), yb1 = __float_as_uint(y1);
    const float t0 = fabsf(x0) - fabsf(y0), t1 = fabsf(x1) - fabsf(y1);
    const uint32_t sw0 = (uint32_t) (__float_as_int(t0) >> 31), sw1 = (uint32_t) (__float_as_int(t1) >> 31);
    const float lo0 = fabsf(__uint_as_float(bit_select(sw0, xb0, yb0)));
    const float lo1 = fabsf(__uint_as_float(bit_select(sw1, xb1, yb1)));
    const float hi0 = fabsf(__uint_as_float(bit_select(sw0, yb0, xb0)));
    const float hi1 = fabsf(__uint_as_float(bit_select(sw1, yb1, xb1)));
    const pf2 q = {fmul_legacy(lo0, __builtin_amdgcn_rcpf(hi0)), fmul_legacy(lo1, __builtin_amdgcn_rcpf(hi1))};
    const pf2 s = q * q;
    auto k = [](float c) { return pf2{c, c}; };
    pf2 p = fma2(k(lead), s, k(2.186295763e-02f));
    p = fma2(p, s, k(-5.591232702e-02f));
    p = fma2(p, s, k(9.642197192e-02f));
    p = fma2(p, s, k(-1.390862912e-01f));
    p = fma2(p, s, k(1.994656622e-01f));
    p = fma2(p, s, k(-3.332985938e-01f));
    p = fma2(p, s, k(9.999993443e-01f));
    const pf2 r = fma2(p, q, pf2{__uint_as_float(sw0 & 0xBFC90FDBu), __uint_as_float(sw1 & 0xBFC90FDBu)});
    const uint32_t xn0 = (uint32_t) ((int32_t) xb0 >> 31), xn1 = (uint32_t) ((int32_t) xb1 >> 31);
    const float a0 = fabsf(r.x) - __uint_as_float(xn0 & 0x40490FDBu), a1 = fabsf(r.y) - __uint_as_float(xn1 & 0x40490FDBu);
    r0 = __uint_as_float(bit_select(0x7FFFFFFFu, __float_as_uint(a0), yb0));
    r1 = __uint_as_float(bit_select(0x7FFFFFFFu, __float_as_uint(a1), yb1));
}

// The discriminator's argument for complex samples z0 = (a, b), z1 = (c, d), as the pair
// (zr, zj) = (b d + a c, b c + (-a) d): the two products of a first, then one fused
// multiply-add per lane -- operation for operation what discriminate() does.  The operand
// selectors pick the lanes (op_sel: low half source, op_sel_hi: high half source), neg_hi
// flips the sign of a d on its way into the sum [-(a d) == (-a) d bit for bit].
__device__ __forceinline__ pf2 discriminator_argument(pf2 z0, pf2 z1) {
    pf2 t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t) : "v"(z0), "v"(z1));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[0,0,1]" : "=v"(r) : "v"(z0), "v"(z1), "v"(t));
    return r;
}

// Quadrature discriminator of one pair of complex samples z0 = a + ib,
// z1 = c + id: arg(z0 * conj(z1)), src/matrix.c:170-173.
__device__ __forceinline__ float discriminate(float a, float b, float c, float d, float lead = kAtanLead) {
    const float zr = fmaf(b, d, a * c);
    const float zj = fmaf(b, c, (-a) * d);
    return fast_atan2(zj, zr, lead);
}

// The same with the reference's rounding (two products, one sum, src/matrix.c:170-171),
// for the literal-replay kernels: their inputs are bit-identical to the reference's, so
// zr and zj are too, and no +pi / -pi ambiguity can arise where the chunk-head weights
// make it hard to recognise.
__device__ __forceinline__ float discriminate_literal(float a, float b, float c, float d) {
    const float zr = a * c + b * d;
    const float zj = -a * d + b * c;
    const float r = fast_atan2(zj, zr);
    // src/matrix.c:174 maps a NaN result to 0.  build_plan proves every value of the path
    // finite from the design (plan_magnitude_bound), so this never fires; the literal
    // kernels keep the reference's statement, the streaming kernel relies on the proof.
    return r != r ? 0.f : r;
}

// Four bytes -> (byte + 1) & 255 each, all four at once (SWAR): adding 1 to the
// low 7 bits of every byte cannot carry into the next byte; the carry into bit 7
// is folded in with an xor.  (byte + 1) & 255 == (int8_t)(byte - 127) + 128.
__device__ __forceinline__ uint32_t rebias4(uint32_t w) {
    const uint32_t t = (w & 0x7F7F7F7Fu) + 0x01010101u;
    return t ^ (w & 0x80808080u);
}

// 128.0f that hipcc cannot see through: keeps "(float) byte - 128" a
// v_cvt_f32_ubyteN and a float subtract instead of integer SDWA arithmetic
// followed by v_cvt_f32_i32 (both exact; the former is cheaper on gfx950).
__device__ __forceinline__ float opaque_128() {
    float k = 128.0f;
    asm("" : "+v"(k));
    return k;
}

// FOLD: hand back (byte + 1) & 255 as a float, i.e. the sample plus 128, and let
// the complex stencil absorb the -128 in its constant term (see process_row).
template <int COND, bool FOLD = false>
__device__ __forceinline__ void unpack_dword(uint32_t w, float scale, float &i0, float &q0, float &i1,
                                             float &q1) {
    if constexpr (COND == kCorrectIq) {
        // the byte itself; its running offset is subtracted by the caller (src/matrix.c:131-135)
        i0 = (float) (w & 0xFFu);
        q0 = (float) ((w >> 8) & 0xFFu);
        i1 = (float) ((w >> 16) & 0xFFu);
        q1 = (float) (w >> 24);
    } else if constexpr (COND == kNormalize) {
        // byte * (2/255) - 1, product and difference rounded separately (src/matrix.c:106,112)
        i0 = (float) (w & 0xFFu) * scale - 1.0f;
        q0 = (float) ((w >> 8) & 0xFFu) * scale - 1.0f;
        i1 = (float) ((w >> 16) & 0xFFu) * scale - 1.0f;
        q1 = (float) (w >> 24) * scale - 1.0f;
    } else {
        // (int8_t)(byte - 127): 255 -> -128, 127 -> +0 (src/matrix.c:92-96), as
        // v_cvt_f32_ubyteN of the re-biased byte minus 128 (both exact).  The
        // byte-select conversions issue beside the main VALU pipe; the SDWA
        // sign-extending form costs more than twice a full-rate operation
        // (tools/microbench/valu_rate.hip).
        const uint32_t u = rebias4(w);
        if constexpr (FOLD) {
            i0 = (float) (u & 0xFFu);
            q0 = (float) ((u >> 8) & 0xFFu);
            i1 = (float) ((u >> 16) & 0xFFu);
            q1 = (float) (u >> 24);
        } else {
            const float k = opaque_128();
            i0 = (float) (u & 0xFFu) - k;
            q0 = (float) ((u >> 8) & 0xFFu) - k;
            i1 = (float) ((u >> 16) & 0xFFu) - k;
            q1 = (float) (u >> 24) - k;
        }
    }
}

template <int COND>
__device__ __forceinline__ float unpack_byte(uint8_t b, float scale) {
    if constexpr (COND == kNormalize) return (float) b * scale - 1.0f;
    if constexpr (COND == kCorrectIq) return (float) b;  // minus its running offset, see q1_offset()
    return (float) (int8_t) (uint8_t) (b - 127);
}

// -q1: the offset pair complex sample k of a chunk of K samples was corrected with.  correctIq
// walks a chunk from both ends at once: iteration t takes samples t and K-1-t with the same
// offsets (src/matrix.c:131-135), so sample k belongs to iteration min(k, K-1-k).
__device__ __forceinline__ float2 q1_offset(const float *chunk_table, int k, int K) {
    const int t = k < K - 1 - k ? k : K - 1 - k;
    return reinterpret_cast<const float2 *>(chunk_table)[t];
}
struct Q1Row {
    const float *table;  // the chunk's offsets
    int k0;              // chunk-local index of this lane's first complex sample
    int K;               // complex samples of the chunk
};

// Look-ahead values a row needs from the row after it (lane 0 of that row).
struct Halo {
    float xr[kMaxTaps - 1];  // re of its first complex samples
    float xi[kMaxTaps - 1];
    float d[kMaxTaps - 1];   // its first discriminator outputs
};

// How the lanes of a row get the values that follow their own 16 bytes: each lane
// needs the first few samples (and discriminator outputs) of the lane after it,
// lane 63 those of lane 0 of the row that follows.  Rows are processed from the
// end of a segment to its start, so that row has always been seen already.
//
// (1) Registers: v_mov_b32 with a DPP wave shift inside the row, the following
//     row's lane-0 values in scalar registers (v_readfirstlane).  Three VALU
//     operations per value and row; used by the bounds-checked build.
//
// `edge`: this row starts a chunk.  What its lane 0 hands to the row before it is then not its
// own values but what the reference finds past a chunk's end: 0.0f samples (`past_re`, `past_im`:
// 128 where the unpack bias is folded, (0, s) where the stencil is divided by its first tap), 0.0f
// discriminator outputs.  (A segment spans chunks when chunks
// are shorter than a segment.)
struct RegisterExchange {
    Halo &halo;
    bool edge = false;
    float past_re = 0.f, past_im = 0.f;
    template <int N>
    __device__ __forceinline__ void next_x(float *xr, float *xi) {
#pragma unroll
        for (int j = 0; j < N; ++j) {
            xr[8 + j] = from_next_lane(xr[j], halo.xr[j]);
            xi[8 + j] = from_next_lane(xi[j], halo.xi[j]);
            halo.xr[j] = edge ? past_re : wave_first(xr[j]);
            halo.xi[j] = edge ? past_im : wave_first(xi[j]);
        }
    }
    template <int N>
    __device__ __forceinline__ void next_d(float *d) {
#pragma unroll
        for (int j = 0; j < N; ++j) {
            d[4 + j] = from_next_lane(d[j], halo.d[j]);
            halo.d[j] = edge ? 0.f : wave_first(d[j]);
        }
    }
};

// (2) LDS: every lane writes its first values to its slot of the row's exchange
//     line and reads the slot after it; lane 63 reads slot 0 of the line the
//     following row wrote (two lines per wavefront, used alternately).  The LDS
//     queue takes these instead of the VALU, which is the unit this kernel is
//     short of.  One wavefront's LDS operations execute in order, so no barrier
//     is needed: a line is only ever touched by the wavefront that owns it.
typedef __attribute__((address_space(3))) float lds_float;
typedef float f32x2 __attribute__((ext_vector_type(2)));
struct LdsExchange {
    lds_float *wr_x;
    const lds_float *rd_x;
    lds_float *wr_d;
    const lds_float *rd_d;
    bool edge0 = false;  // lane 0 of a row that starts a chunk: publishes past-the-end values (see RegisterExchange)
    float past_re = 0.f, past_im = 0.f;
    template <int N>
    __device__ __forceinline__ void next_x(float *xr, float *xi) {
#pragma unroll
        for (int j = 0; j < N; ++j) {
            f32x2 v;
            v.x = edge0 ? past_re : xr[j];
            v.y = edge0 ? past_im : xi[j];
            *reinterpret_cast<__attribute__((address_space(3))) f32x2 *>(wr_x + 2 * j) = v;
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const f32x2 v = *reinterpret_cast<const __attribute__((address_space(3))) f32x2 *>(rd_x + 2 * j);
            xr[8 + j] = v.x;
            xi[8 + j] = v.y;
        }
    }
    template <int N>
    __device__ __forceinline__ void next_d(float *d) {
#pragma unroll
        for (int j = 0; j < N; ++j) wr_d[j] = edge0 ? 0.f : d[j];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < N; ++j) d[4 + j] = rd_d[j];
    }
};

// T-tap forward stencil  sum_j h[j] * v[j]  (+ c0 folded into the first product)
template <int T>
__device__ __forceinline__ float stencil(const float *h, const float *v, float c0) {
    float acc = fmaf(h[0], v[0], c0);
#pragma unroll
    for (int j = 1; j < T; ++j) acc = fmaf(h[j], v[j], acc);
    return acc;
}
template <int T>
__device__ __forceinline__ float stencil0(const float *h, const float *v) {
    float acc = h[0] * v[0];
#pragma unroll
    for (int j = 1; j < T; ++j) acc = fmaf(h[j], v[j], acc);
    return acc;
}

// With the complex input filter and the shiftOrigin unpack, the "- 128" of the
// unpack is folded into the stencil's constant term:
//   h0 (u0 - 128) + h1 (u1 - 128) [+ bias] = fma(h1, u1, fma(h0, u0, C)),  C = [bias] - 128 (h0 + h1)
// (4 fewer VALU operations per output; the different rounding moves the result
// by less than the reference's own FMA / non-FMA builds differ).
// Samples past a chunk end are 0.0f in the reference, i.e. 128 in this domain.
template <int TIN, int COND, bool RAGGED>
constexpr bool fold_unpack() {
    return TIN > 0 && COND == kShiftOrigin;
}
// The complex stencil divided by its first tap (plan.h, kShiftOriginNorm): z / h0 = X[k] + sum_j rho_j X[k+j]
// with X = (re, im + s) straight from the unpack's subtraction of (128, 128 - s).
template <int TIN, int COND>
constexpr bool norm_stencil() {
    return TIN > 0 && COND == kShiftOriginNorm;
}
template <int COND>
constexpr bool shift_origin() {
    return COND == kShiftOrigin || COND == kShiftOriginPlain || COND == kShiftOriginNorm;
}
template <int T>
__device__ __forceinline__ float stencil_norm(const float *rho, const float *v) {
    float acc = v[0];
#pragma unroll
    for (int j = 1; j < T; ++j) acc = fmaf(rho[j], v[j], acc);
    return acc;
}

// One row for one lane: 16 input bytes (8 complex samples) -> 4 outputs.
// `valid` = usable bytes of this lane's 16 (16 unless RAGGED); `ex` supplies the
// look-ahead values and takes this row's for the row before it; `lead`: see fast_atan2.
template <int TIN, int TOUT, int COND, bool RAGGED, class EX>
__device__ __forceinline__ void process_row(const uint4 w, const RowLaunch &a, EX &ex, int valid, float lead,
                                            float (&y)[4], const Q1Row q1 = Q1Row{nullptr, 0, 0}) {
    static_assert(COND != kCorrectIq || RAGGED, "-q1 runs in the bounds-checked build");
    constexpr int NX = 8 + (TIN > 0 ? TIN - 1 : 0);
    constexpr int ND = 4 + TOUT - 1;
    float xr[NX], xi[NX];
    constexpr bool FOLD = fold_unpack<TIN, COND, RAGGED>();
    constexpr bool NORM = norm_stencil<TIN, COND>();
    // the plain shiftOrigin unpack of the streaming builds takes its 128 off (re, im) pairs: eight
    // v_pk_add_f32 instead of sixteen subtractions (the same values); the normalised stencil's unpack
    // takes (128, 128 - s) off them
    constexpr bool PAIRS = !RAGGED && !FOLD && shift_origin<COND>();
    constexpr bool RAW = FOLD || PAIRS || NORM;  // unpack_dword hands back sample + 128
    unpack_dword<COND, RAW>(w.x, a.scale, xr[0], xi[0], xr[1], xi[1]);
    unpack_dword<COND, RAW>(w.y, a.scale, xr[2], xi[2], xr[3], xi[3]);
    unpack_dword<COND, RAW>(w.z, a.scale, xr[4], xi[4], xr[5], xi[5]);
    unpack_dword<COND, RAW>(w.w, a.scale, xr[6], xi[6], xr[7], xi[7]);
    if constexpr (PAIRS && NORM) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const pf2 p = fma2(pf2{xr[k], xi[k]}, pf2{a.alpha, a.alpha}, pf2{a.nk_re, a.nk_im});  // exact (plan.cpp)
            xr[k] = p.x;
            xi[k] = p.y;
        }
    } else if constexpr (PAIRS) {
        const float k128 = opaque_128();
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const pf2 p = pf2{xr[k], xi[k]} - pf2{k128, k128};
            xr[k] = p.x;
            xi[k] = p.y;
        }
    } else if constexpr (NORM) {
        // (bounds-checked build: the same operations, one by one)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            xr[k] = fmaf(xr[k], a.alpha, a.nk_re);
            xi[k] = fmaf(xi[k], a.alpha, a.nk_im);
        }
    }
    if constexpr (COND == kCorrectIq) {
        // Sample k was corrected in iteration min(k, K-1-k) (q1_offset): a lane's 8 samples lie
        // on one side of the chunk's middle whenever K is a multiple of 16, and their 8 offset
        // pairs are then 64 consecutive bytes of the table, ascending in the first half,
        // descending in the second: four 16-byte loads instead of eight 8-byte ones.
        const int K = q1.K, k0 = q1.k0;
        const bool first_half = k0 + 8 <= (K >> 1), second_half = k0 >= (K >> 1) && k0 + 8 <= K;
        if ((K & 15) == 0 && (first_half || second_half)) {
            const float4 *p = reinterpret_cast<const float4 *>(q1.table + 2 * (first_half ? k0 : K - 8 - k0));
            const float4 t0 = p[0], t1 = p[1], t2 = p[2], t3 = p[3];
            const float offr[8] = {t0.x, t0.z, t1.x, t1.z, t2.x, t2.z, t3.x, t3.z};
            const float offi[8] = {t0.y, t0.w, t1.y, t1.w, t2.y, t2.w, t3.y, t3.w};
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                xr[k] = xr[k] - (first_half ? offr[k] : offr[7 - k]);
                xi[k] = xi[k] - (first_half ? offi[k] : offi[7 - k]);
            }
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (k0 + k < K) {
                    const float2 off = q1_offset(q1.table, k0 + k, K);
                    xr[k] = xr[k] - off.x;
                    xi[k] = xi[k] - off.y;
                }
            }
        }
    }
    if constexpr (RAGGED) {
        // past the end of a short final chunk the reference's buffers hold 0.0f
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const bool ok = 2 * k < valid;
            xr[k] = ok ? xr[k] : (FOLD ? 128.f : 0.f);
            xi[k] = ok ? xi[k] : (FOLD ? 128.f : NORM ? a.past_im : 0.f);
        }
    }

    float d[ND];
    // complex samples 8.. come from the next lane
    if constexpr (TIN > 1) ex.template next_x<TIN - 1>(xr, xi);
    if constexpr (!RAGGED) {
        // the streaming build works on (re, im) pairs
        pf2 z[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if constexpr (TIN == 0) {
                z[k] = pf2{xr[k], xi[k]};
            } else if constexpr (NORM) {
                pf2 acc = pf2{xr[k], xi[k]};
#pragma unroll
                for (int j = 1; j < TIN; ++j) acc = fma2(pf2{a.rho[j], a.rho[j]}, pf2{xr[k + j], xi[k + j]}, acc);
                z[k] = acc;
            } else if constexpr (FOLD) {
                pf2 acc = fma2(pf2{a.hin[0], a.hin[0]}, pf2{xr[k], xi[k]}, pf2{a.fold_re, a.fold_im});
#pragma unroll
                for (int j = 1; j < TIN; ++j) acc = fma2(pf2{a.hin[j], a.hin[j]}, pf2{xr[k + j], xi[k + j]}, acc);
                z[k] = acc;
            } else {
                // (the real lane starts from +0.0f where the imaginary lane starts from its bias: a pair)
                pf2 acc = fma2(pf2{a.hin[0], a.hin[0]}, pf2{xr[k], xi[k]}, pf2{0.f, a.bias_im});
#pragma unroll
                for (int j = 1; j < TIN; ++j) acc = fma2(pf2{a.hin[j], a.hin[j]}, pf2{xr[k + j], xi[k + j]}, acc);
                z[k] = acc;
            }
        }
        pf2 arg[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) arg[j] = discriminator_argument(z[2 * j], z[2 * j + 1]);
        fast_atan2_pair(arg[0].y, arg[0].x, arg[1].y, arg[1].x, lead, d[0], d[1]);
        fast_atan2_pair(arg[2].y, arg[2].x, arg[3].y, arg[3].x, lead, d[2], d[3]);
    } else if constexpr (TIN > 0) {
        float zr[8], zi[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if constexpr (NORM) {
                zr[k] = stencil_norm<TIN>(a.rho, &xr[k]);
                zi[k] = stencil_norm<TIN>(a.rho, &xi[k]);
            } else if constexpr (FOLD) {
                zr[k] = stencil<TIN>(a.hin, &xr[k], a.fold_re);
                zi[k] = stencil<TIN>(a.hin, &xi[k], a.fold_im);
            } else {
                zr[k] = stencil<TIN>(a.hin, &xr[k], 0.f);   // from +0.0f, as the streaming build's pair does
                zi[k] = stencil<TIN>(a.hin, &xi[k], a.bias_im);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            d[j] = discriminate(zr[2 * j], zi[2 * j], zr[2 * j + 1], zi[2 * j + 1], lead);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            d[j] = discriminate(xr[2 * j], xi[2 * j], xr[2 * j + 1], xi[2 * j + 1], lead);
    }
    if constexpr (RAGGED) {
#pragma unroll
        for (int j = 0; j < 4; ++j) d[j] = 4 * j < valid ? d[j] : 0.f;
    }
    if constexpr (TOUT > 1) ex.template next_d<TOUT - 1>(d);
#pragma unroll
    for (int j = 0; j < 4; ++j) y[j] = stencil0<TOUT>(a.hout, &d[j]);
}

// Look-ahead a segment needs from the row after it: only that row's lane-0
// values (its first complex samples and first discriminator outputs), which
// depend on its first lanes alone.  For the common short stencils they depend on
// lane 0's own 16 bytes: every lane has read THOSE bytes (one broadcast request), so every
// lane computes the same values and nothing has to be fetched from lane 0; only the samples
// that are read are unpacked.  The long stencils fall back to a full row (its own look-ahead
// is irrelevant), lane 0's results broadcast.  Same operations on the same values as
// process_row performs for that row's lane 0.
template <int TIN, int TOUT, int COND>
__device__ __forceinline__ void next_row_lookahead(const uint4 w, const RowLaunch &a, Halo &halo) {
    constexpr int ND = TOUT - 1;                        // discriminator outputs to hand back
    constexpr int NX = 2 * ND + (TIN > 0 ? TIN - 1 : 0);  // complex samples they read
    if constexpr (NX <= 8 && (TIN > 0 ? TIN - 1 : 0) <= 8) {
        float xr[8], xi[8];
        constexpr bool FOLD = fold_unpack<TIN, COND, false>();
        constexpr bool NORM = norm_stencil<TIN, COND>();
        constexpr bool PAIRS = !FOLD && shift_origin<COND>();
        constexpr bool RAW = FOLD || PAIRS;
        unpack_dword<COND, RAW>(w.x, a.scale, xr[0], xi[0], xr[1], xi[1]);
        if constexpr (NX > 2) unpack_dword<COND, RAW>(w.y, a.scale, xr[2], xi[2], xr[3], xi[3]);
        if constexpr (NX > 4) unpack_dword<COND, RAW>(w.z, a.scale, xr[4], xi[4], xr[5], xi[5]);
        if constexpr (NX > 6) unpack_dword<COND, RAW>(w.w, a.scale, xr[6], xi[6], xr[7], xi[7]);
        pf2 x[NX > 0 ? NX : 1];
        if constexpr (PAIRS && NORM) {
#pragma unroll
            for (int k = 0; k < NX; ++k) x[k] = fma2(pf2{xr[k], xi[k]}, pf2{a.alpha, a.alpha}, pf2{a.nk_re, a.nk_im});
        } else if constexpr (PAIRS) {
            const float k128 = opaque_128();
#pragma unroll
            for (int k = 0; k < NX; ++k) x[k] = pf2{xr[k], xi[k]} - pf2{k128, k128};
        } else {
#pragma unroll
            for (int k = 0; k < NX; ++k) x[k] = pf2{xr[k], xi[k]};
        }
#pragma unroll
        for (int j = 0; j < ND; ++j) {
            pf2 z[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int k = 2 * j + e;
                if constexpr (TIN == 0) {
                    z[e] = x[k];
                } else if constexpr (NORM) {
                    pf2 acc = x[k];
#pragma unroll
                    for (int t = 1; t < TIN; ++t) acc = fma2(pf2{a.rho[t], a.rho[t]}, x[k + t], acc);
                    z[e] = acc;
                } else {
                    pf2 acc = fma2(pf2{a.hin[0], a.hin[0]}, x[k], FOLD ? pf2{a.fold_re, a.fold_im} : pf2{0.f, a.bias_im});
#pragma unroll
                    for (int t = 1; t < TIN; ++t) acc = fma2(pf2{a.hin[t], a.hin[t]}, x[k + t], acc);
                    z[e] = acc;
                }
            }
            const pf2 arg = discriminator_argument(z[0], z[1]);
            halo.d[j] = fast_atan2(arg.y, arg.x);
        }
        if constexpr (TIN > 1) {
#pragma unroll
            for (int j = 0; j < TIN - 1; ++j) {
                halo.xr[j] = x[j].x;
                halo.xi[j] = x[j].y;
            }
        }
    } else {
        float unused[4];
        RegisterExchange ex{halo};
        process_row<TIN, TOUT, COND, false>(w, a, ex, 16, kAtanLead, unused);
    }
}

// 1 KiB of bytes that unpack to 0.0f samples (shiftOrigin: byte 127): what a segment that ends a
// chunk "reads" as the row after it -- the reference's buffers hold zeros there.  Such a row's
// first discriminator outputs are +0.0f too (samples (0, c), (0, c) with the complex stage's
// constant c: zr = c c >= +0, zj = +0), so the look-ahead of a chunk end is computed like any
// other and no lane has to select.  (Not for the folded unpack, whose stencil of zeros is not
// exactly its constant, and not for normalizeInput, which has no byte for 0.0f.)
struct PastEndBytes {
    uint32_t w[kRowBytes / 4];
    constexpr PastEndBytes() : w() {
        for (uint32_t i = 0; i < kRowBytes / 4; ++i) w[i] = 0x7F7F7F7Fu;
    }
};
__device__ __attribute__((aligned(16))) const PastEndBytes kPastEndRow = PastEndBytes();

__device__ __forceinline__ uint4 load_row16(const uint8_t *row_base, uint32_t lane) {
    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(row_base) + lane);
    return make_uint4(v.x, v.y, v.z, v.w);
}

// Cache policy of the result stores.  hipcc's __builtin_nontemporal_store gives "nt"; the bare
// copy with this kernel's pattern (tools/microbench/policy_sweep.hip: 4 rows per wavefront
// requested up front, "nt" loads) runs at 0.801 of peak with "nt" stores and at 0.814 with
// "sc1 nt" (0.810 "sc1", 0.747 none).  The kernel follows where memory alone binds it: without a
// complex stage (BASELINE config 2) 0.797 -> 0.809 and 0.805 -> 0.814 on two boxes, three
// interleaved rounds each; with one (config 3, the target) +0.3 % on one box, -0.4 % on the
// other -- there the VALU side co-limits and the policy stays "nt" (profiles/r2_ab_workgroup.txt).
// There is no builtin for sc1: that store is spelled out, with the wait states a VALU write to
// the data registers of a wide store needs behind it (hipcc cannot see into the statement).
// DEMOD_STORE_POLICY_ID (experiments): 0 = by build as above, 1 = " nt" spelled out everywhere
// (control), 2 = " sc1 nt" everywhere, 3 = " sc0 sc1 nt", 4 = " sc1", 5 = the builtin everywhere.
#ifndef DEMOD_STORE_POLICY_ID
#define DEMOD_STORE_POLICY_ID 0
#endif
#if DEMOD_STORE_POLICY_ID == 1
#define DEMOD_STORE_POLICY " nt"
#elif DEMOD_STORE_POLICY_ID == 3
#define DEMOD_STORE_POLICY " sc0 sc1 nt"
#elif DEMOD_STORE_POLICY_ID == 4
#define DEMOD_STORE_POLICY " sc1"
#else
#define DEMOD_STORE_POLICY " sc1 nt"
#endif
// WIDE: this build's stores take the spelled-out policy
template <bool WIDE>
__device__ __forceinline__ void store_out16(float *dst, const float (&y)[4]) {
    f32x4 v;
    v.x = y[0]; v.y = y[1]; v.z = y[2]; v.w = y[3];
    if constexpr (DEMOD_STORE_POLICY_ID != 5 && (WIDE || DEMOD_STORE_POLICY_ID != 0))
        asm volatile("global_store_dwordx4 %0, %1, off" DEMOD_STORE_POLICY "\n\ts_nop 1" : : "v"(dst), "v"(v) : "memory");
    else
        __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(dst));
}

// int16 output (DEMOD_OUT_S16): one signed-normalised conversion per pair,
// v_cvt_pknorm_i16_f32: clamp(y * scale, -1, 1) * 32767, rounded to nearest.
typedef short s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t to_s16_pair(float lo, float hi, float scale) {
    const s16x2 p = __builtin_amdgcn_cvt_pknorm_i16(lo * scale, hi * scale);
    return (uint32_t) (uint16_t) p.x | ((uint32_t) (uint16_t) p.y << 16);
}
__device__ __forceinline__ int16_t to_s16(float v, float scale) {
    return (int16_t) (to_s16_pair(v, 0.f, scale) & 0xFFFFu);
}
__device__ __forceinline__ void store_out8_s16(void *dst, const float (&y)[4], float scale) {
    u32x2 v;
    v.x = to_s16_pair(y[0], y[1], scale);
    v.y = to_s16_pair(y[2], y[3], scale);
    __builtin_nontemporal_store(v, reinterpret_cast<u32x2 *>(dst));
}
// Four results of one lane to element `index` of the output, in the output's format.
template <bool S16, bool WIDE = false>
__device__ __forceinline__ void store_quad(void *dst, const float (&y)[4], float scale) {
    if constexpr (S16) store_out8_s16(dst, y, scale);
    else store_out16<WIDE>(static_cast<float *>(dst), y);
}
// The same for a row of the streaming kernel: `row` is wave-uniform (it travels in scalar
// registers), `lane_off` this lane's byte offset in the row.  Spelled out: hipcc turns
// uniform-plus-lane pointer arithmetic into a 64-bit VALU add per store.  The wait states are
// the ones a VALU write to the data registers of a wide store needs behind it.
template <bool S16, bool WIDE>
__device__ __forceinline__ void store_row(char *row, uint32_t lane_off, const float (&y)[4], float scale) {
    if constexpr (S16) {
        u32x2 v;
        v.x = to_s16_pair(y[0], y[1], scale);
        v.y = to_s16_pair(y[2], y[3], scale);
        asm volatile("global_store_dwordx2 %0, %1, %2 nt\n\ts_nop 1" : : "v"(lane_off), "v"(v), "s"(row) : "memory");
    } else {
        f32x4 v;
        v.x = y[0]; v.y = y[1]; v.z = y[2]; v.w = y[3];
        if constexpr (DEMOD_STORE_POLICY_ID != 5 && (WIDE || DEMOD_STORE_POLICY_ID != 0))
            asm volatile("global_store_dwordx4 %0, %1, %2" DEMOD_STORE_POLICY "\n\ts_nop 1" : : "v"(lane_off), "v"(v), "s"(row) : "memory");
        else
            asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 1" : : "v"(lane_off), "v"(v), "s"(row) : "memory");
    }
}
template <bool S16>
__device__ __forceinline__ char *out_address(void *base, uint64_t index) {
    return static_cast<char *>(base) + index * (S16 ? 2u : 4u);
}

// 16 bytes of a ragged row: only `valid` (multiple of 4) bytes exist.
__device__ __forceinline__ uint4 load_row16_ragged(const uint8_t *row_base, uint32_t lane, int valid) {
    const uint32_t *p = reinterpret_cast<const uint32_t *>(row_base) + 4 * lane;
    uint4 w = make_uint4(0, 0, 0, 0);
    if (valid >= 4) w.x = p[0];
    if (valid >= 8) w.y = p[1];
    if (valid >= 12) w.z = p[2];
    if (valid >= 16) w.w = p[3];
    return w;
}

// ---------------------------------------------------------------------------
// chunk head inside the streaming kernel: literal replay, one output per lane
// ---------------------------------------------------------------------------
// The first H = max(2 So - 1, 4 Si - 2) outputs of a chunk are not interior: cells near
// the chunk start see only some of the sections, and with a complex stage its last
// iterations write past its output into the cells the output stage starts from (SURVEY.md
// Appendix A.4).  They are reproduced by executing the reference's own operation sequence
// (src/filter.c:221-230, :242-258; same association, -ffp-contract=off).  That sequence is
// sequential in the reference, but cell n after section m,
//     P(n, m) = (b0 P(n, m+1) + b1 P(n+1, m+2) [+ 1]) - ((a0 + a1 x[n]) + a2 x[n+1])    if iteration n - S - m exists,
//             = P(n, m+1)                                                                   otherwise,
// only needs its right-hand neighbour two sections earlier, so lane n carries output n (and
// complex cells 2n, 2n+1), walks m = S-1 .. 0 and fetches the neighbour's previous value
// from lane n + 1: every lane runs the literal operations of its own cells, all at once.
// Bit-identical to demod_heads (same operations on the same values).
constexpr int kFusedSections = kFusedHeadSections;  // sections per stage chunk_head handles inside demod_rows (degrees <= 8)
constexpr int kWaveHeadSections = kMaxSections;     // ... and in a kernel of its own (demod_heads_wave: any degree)

__device__ __forceinline__ float lane_read(float v, uint32_t src_lane) {
    return __int_as_float(__builtin_amdgcn_ds_bpermute((int) (src_lane << 2), __float_as_int(v)));
}

template <bool PLUS_ONE>
__device__ __forceinline__ float section_update(const float (&row)[6], float cur, float next, float x0, float x1) {
    float t = row[0] * cur + row[1] * next;
    if constexpr (PLUS_ONE) t = t + 1.0f;
    return t - (row[3] + row[4] * x0 + row[5] * x1);
}

// PRE (demod_heads_wave only): -q2 in front of -L.  highpassDc runs first, the same recurrence with
// its own two sections from a buffer of its own; the -L stage then reads ITS cells instead of the
// samples, and finds highpassDc's overflow cells -- three constants of the design, its inputs
// behind its buffer being zeros -- where its own output starts (src/matrix.c:155-156, :260-264:
// filterRet + bufSize is both).
// `lane` counts from the head's own lane 0, which is lane `lane_base` of the wavefront: with chunks shorter than a
// segment the heads of a segment's chunks are replayed side by side, sixteen lanes each (`chunk`, `out_index` and
// `n_out` are then per-lane values).  The neighbour fetches shift the whole wavefront, so a head's lane 15 receives
// the next head's lane 0: harmless where output n only reaches lanes up to n + sec_in + sec_out + 1 < 16.
// `take` ? a : b for floats, `take` as a bit mask (all ones / zero): one v_bitop3_b32.  chunk_head's selects are all
// of this kind (does iteration n - S - m exist for this lane's cell; is the lane odd), their masks a shift of a lane
// expression -- where a compare and a v_cndmask_b32 on VCC cost several times a full-rate operation
// (profiles/r1_microbench_valu_rate2.txt).
__device__ __forceinline__ float pick(uint32_t take, float a, float b) {
    return __uint_as_float(bit_select(take, __float_as_uint(a), __float_as_uint(b)));
}
// all ones where v >= 0
__device__ __forceinline__ uint32_t nonneg(int v) { return ~(uint32_t) (v >> 31); }

template <int COND, bool S16, int MAXSEC = kFusedSections, class LAUNCH = RowLaunch, bool PRE = false>
__device__ __forceinline__ void chunk_head(const LAUNCH &a, const uint8_t *chunk, uint64_t out_index, uint32_t lane, uint32_t n_out,
                                           uint32_t lane_base = 0u) {
    // the segment's own stores (among them the interior-formula values of this row's first
    // lanes, which are overwritten below) are complete before anything here is stored
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int Si = a.sec_in, So = a.sec_out;
    const uint32_t *w = reinterpret_cast<const uint32_t *>(chunk) + lane;
    const uint32_t w0 = w[0], w1 = w[1];
    // complex samples 2n, 2n+1, 2n+2 of the chunk, n = lane
    const float x0r = unpack_byte<COND>((uint8_t) w0, a.scale), x0i = unpack_byte<COND>((uint8_t) (w0 >> 8), a.scale);
    const float x1r = unpack_byte<COND>((uint8_t) (w0 >> 16), a.scale), x1i = unpack_byte<COND>((uint8_t) (w0 >> 24), a.scale);
    const float x2r = unpack_byte<COND>((uint8_t) w1, a.scale), x2i = unpack_byte<COND>((uint8_t) (w1 >> 8), a.scale);
    float Ar = x0r, Ai = x0i, Br = x1r, Bi = x1i;  // what the discriminator reads: samples, or the complex stage's cells 2n, 2n+1
    float y0 = 0.f;                                // what the output stage finds in its cell n
    // what the complex stage reads (cells 2n, 2n+1, 2n+2) and what its own cells 2n, 2n+1 hold before it runs
    float u0r = x0r, u0i = x0i, u1r = x1r, u1i = x1i, u2r = x2r, u2i = x2i;
    float sAr = 0.f, sAi = 0.f, sBr = 0.f, sBi = 0.f;
    if constexpr (PRE) {
        if (a.sec_pre > 0) {
            const int Sp = a.sec_pre;
            float cAr = 0.f, cAi = 0.f, cBr = 0.f, cBi = 0.f, pAr = 0.f, pAi = 0.f, pBr = 0.f, pBi = 0.f;
            const int cellA = 2 * (int) lane;
#pragma unroll
            for (int m = 1; m >= 0; --m) {
                if (m < Sp) {
                    const float nBr = from_next_lane(pAr, 0.f), nBi = from_next_lane(pAi, 0.f);
                    const float vAr = section_update<true>(a.sos_pre[m], cAr, pBr, x0r, x1r);
                    const float vAi = section_update<false>(a.sos_pre[m], cAi, pBi, x0i, x1i);
                    const float vBr = section_update<true>(a.sos_pre[m], cBr, nBr, x1r, x2r);
                    const float vBi = section_update<false>(a.sos_pre[m], cBi, nBi, x1i, x2i);
                    const uint32_t exA = nonneg(cellA - Sp - m), exB = nonneg(cellA + 1 - Sp - m);
                    pAr = cAr; pAi = cAi; pBr = cBr; pBi = cBi;
                    cAr = pick(exA, vAr, cAr); cAi = pick(exA, vAi, cAi);
                    cBr = pick(exB, vBr, cBr); cBi = pick(exB, vBi, cBi);
                }
            }
            u0r = cAr; u0i = cAi; u1r = cBr; u1i = cBi;
            u2r = from_next_lane(cAr, 0.f); u2i = from_next_lane(cAi, 0.f);
            // highpassDc's overflow cells K + t, t < 2 Sp - 1 (the same for every chunk: inputs 0.0f)
            float hr[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, hi[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < 3; ++u) {
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const int t = u + 1 - Sp + m;
                    if (u < 2 * Sp - 1 && m < Sp && t >= 0) {
                        const float b0 = a.sos_pre[m][0], b1 = a.sos_pre[m][1];
                        const float a0 = a.sos_pre[m][3], a1 = a.sos_pre[m][4], a2 = a.sos_pre[m][5];
                        float v = b0 * hr[t] + b1 * hr[t + 1] + 1.0f;
                        hr[t] = v - (a0 + a1 * 0.f + a2 * 0.f);
                        v = b0 * hi[t] + b1 * hi[t + 1];
                        hi[t] = v - (a0 + a1 * 0.f + a2 * 0.f);
                    }
                }
            }
            const int nh = 2 * Sp - 1;  // cell 2n of the -L stage's output is overflow cell 2n, cell 2n+1 is 2n+1
            sAr = lane == 0u ? hr[0] : (lane == 1u && nh > 2 ? hr[2] : 0.f);
            sAi = lane == 0u ? hi[0] : (lane == 1u && nh > 2 ? hi[2] : 0.f);
            sBr = lane == 0u && nh > 1 ? hr[1] : 0.f;
            sBi = lane == 0u && nh > 1 ? hi[1] : 0.f;
        }
    }
    if (Si > 0) {
        // complex stage, cells A = 2n and B = 2n + 1; c* = P(cell, m+1), p* = P(cell, m+2)
        float cAr = sAr, cAi = sAi, cBr = sBr, cBi = sBi, pAr = sAr, pAi = sAi, pBr = sBr, pBi = sBi;
        const int cellA = 2 * (int) lane;
#pragma unroll
        for (int m = MAXSEC - 1; m >= 0; --m) {
            if (m < Si) {
                const float nBr = from_next_lane(pAr, 0.f), nBi = from_next_lane(pAi, 0.f);  // cell 2n + 2 lives in the next lane
                const float vAr = section_update<true>(a.sos_in[m], cAr, pBr, u0r, u1r);
                const float vAi = section_update<false>(a.sos_in[m], cAi, pBi, u0i, u1i);
                const float vBr = section_update<true>(a.sos_in[m], cBr, nBr, u1r, u2r);
                const float vBi = section_update<false>(a.sos_in[m], cBi, nBi, u1i, u2i);
                const uint32_t exA = nonneg(cellA - Si - m), exB = nonneg(cellA + 1 - Si - m);
                pAr = cAr; pAi = cAi; pBr = cBr; pBi = cBi;
                cAr = pick(exA, vAr, cAr); cAi = pick(exA, vAi, cAi);
                cBr = pick(exB, vBr, cBr); cBi = pick(exB, vBi, cBi);
            }
        }
        Ar = cAr; Ai = cAi; Br = cBr; Bi = cBi;
        // overflow cells K + t, t = lane: written by the stage's last 2 Si - 1 iterations into the
        // buffer the output stage starts from.  Their "input" is the head of the stage's own
        // output, x[K + t] == z[t] (the buffers are adjacent, src/matrix.c:260-264) -- or zeros
        // when the stage reads a buffer of its own (-q2).  Cell t lives in lane t / 2.
        const uint32_t s0 = lane >> 1, s1 = (lane + 1u) >> 1;
        const float rA0 = lane_read(Ar, lane_base + s0), rB0 = lane_read(Br, lane_base + s0), iA0 = lane_read(Ai, lane_base + s0), iB0 = lane_read(Bi, lane_base + s0);
        const float rA1 = lane_read(Ar, lane_base + s1), rB1 = lane_read(Br, lane_base + s1), iA1 = lane_read(Ai, lane_base + s1), iB1 = lane_read(Bi, lane_base + s1);
        const uint32_t own = a.in_separate == 0 ? ~0u : 0u, odd = 0u - (lane & 1u);
        const float ztr = __uint_as_float(own & __float_as_uint(pick(odd, rB0, rA0))), zti = __uint_as_float(own & __float_as_uint(pick(odd, iB0, iA0)));
        const float zur = __uint_as_float(own & __float_as_uint(pick(odd, rA1, rB1))), zui = __uint_as_float(own & __float_as_uint(pick(odd, iA1, iB1)));
        float cr = 0.f, ci = 0.f, pr = 0.f, pi = 0.f;
#pragma unroll
        for (int m = MAXSEC - 1; m >= 0; --m) {
            if (m < Si) {
                const float nr = from_next_lane(pr, 0.f), ni = from_next_lane(pi, 0.f);
                const float vr = section_update<true>(a.sos_in[m], cr, nr, ztr, zur);
                const float vi = section_update<false>(a.sos_in[m], ci, ni, zti, zui);
                const uint32_t ex = nonneg(m + Si - 1 - (int) lane);  // iteration K - 2 Si + 1 + (t - 1 + Si - m) is one of the last 2 Si - 1
                pr = cr; pi = ci;
                cr = pick(ex, vr, cr); ci = pick(ex, vi, ci);
            }
        }
        // output cell n starts from overflow cell n / 2: its real part (n even) or imaginary part (n odd)
        const float ovr = lane_read(cr, lane_base + (lane >> 1)), ovi = lane_read(ci, lane_base + (lane >> 1));
        y0 = pick(0u - (lane & 1u), ovi, ovr);
    }
    const float d0 = discriminate_literal(Ar, Ai, Br, Bi);
    const float d1 = from_next_lane(d0, 0.f);
    float cy = y0, py = y0;  // P(n, So) = P(n, So + 1) = what the cell held before
#pragma unroll
    for (int m = MAXSEC - 1; m >= 0; --m) {
        if (m < So) {
            const float ny = from_next_lane(py, 0.f);
            const float v = section_update<true>(a.sos_out[m], cy, ny, d0, d1);
            const uint32_t ex = nonneg((int) lane - So - m);
            py = cy;
            cy = pick(ex, v, cy);
        }
    }
    if (lane < n_out) {
        if constexpr (S16) __builtin_nontemporal_store(to_s16(cy, a.out_scale), reinterpret_cast<int16_t *>(a.out) + out_index + lane);
        else __builtin_nontemporal_store(cy, reinterpret_cast<float *>(a.out) + out_index + lane);
    }
}

// ---------------------------------------------------------------------------
// the fused streaming kernel
// ---------------------------------------------------------------------------
// Register budget: 8 wavefronts per SIMD (64 VGPRs).  The common stencils fit without
// spilling (56-62 VGPRs; 68 without the hint, i.e. 7 wavefronts) and one more resident
// wavefront is worth 1.5-2.5 % (profiles/r1_ab_rows_per_wave.txt); the 5-tap generic build
// does not fit and keeps what it needs (the hint does not force spills).
#ifndef DEMOD_MIN_WAVES_PER_SIMD
#define DEMOD_MIN_WAVES_PER_SIMD 8
#endif
// SPAN: chunks are shorter than a segment (1-2 KiB chunks), so a segment holds several chunks and
// the look-ahead restarts inside it (LdsExchange::edge0); a build of its own, because the extra
// selects do not fit the 64-register budget of the streaming build, and a spilled register of a
// load in flight would be read before it has landed (tools/check_asm_waits.py finds those).
// (The int16 store and SPAN need a few registers more: 7 wavefronts per SIMD rather than spills.)
template <int TIN, int TOUT, int COND, bool RAGGED, bool S16, bool SPAN = false>
__global__ __launch_bounds__(64 * kWavesPerBlock, (S16 || SPAN) ? DEMOD_MIN_WAVES_PER_SIMD - 1 : DEMOD_MIN_WAVES_PER_SIMD) void demod_rows(const RowLaunch a) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#if DEMOD_XCD_INTERLEAVE
    // Consecutive workgroups go to consecutive XCDs: give the kXcd workgroups of a group
    // interleaved segments, so that each XCD streams every kXcd-th segment of a compact window
    // instead of a contiguous kWavesPerBlock-segment piece of it.
    constexpr uint32_t kXcd = DEMOD_XCD_INTERLEAVE;
    const uint32_t seg = (blockIdx.x / kXcd) * (kXcd * (uint32_t) kWavesPerBlock) + wave * kXcd + blockIdx.x % kXcd;
#else
    const uint32_t seg = blockIdx.x * (uint32_t) kWavesPerBlock + wave;
#endif

    // rows are numbered from the start of the chunk that encloses buffer row 0
    const uint32_t g0 = a.phase_rows;
    const uint32_t g1 = g0 + a.n_rows;
    const uint32_t gs = ((g0 >> a.seg_log2) + seg) << a.seg_log2;  // where the segment starts, clipped or not
    uint32_t gb = gs;
    uint32_t ge = gb + (1u << a.seg_log2);
    gb = gb < g0 ? g0 : gb;
    ge = ge > g1 ? g1 : ge;
    if (gb >= ge) return;
    const uint32_t rb = gb - g0, re = ge - g0;  // buffer rows [rb, re), processed from re-1 down to rb
    const uint32_t rpc_mask = (1u << a.rpc_log2) - 1u;
    const uint32_t last_row = a.n_rows - 1u;

    auto lane_valid = [&](uint32_t r) -> int {
        if constexpr (!RAGGED) return 16;
        if (r != last_row || a.ragged_bytes == 0) return 16;
        const int v = (int) a.ragged_bytes - (int) (lane * 16u);
        return v < 0 ? 0 : (v > 16 ? 16 : v);
    };
    auto load = [&](uint32_t r) -> uint4 {
        const uint8_t *base = a.in + (uint64_t) r * kRowBytes;
        if constexpr (RAGGED) {
            if (r == last_row && a.ragged_bytes != 0) return load_row16_ragged(base, lane, lane_valid(r));
        }
        return load_row16(base, lane);
    };

    Halo halo;
#pragma unroll
    for (int j = 0; j < kMaxTaps - 1; ++j) halo.xr[j] = halo.xi[j] = halo.d[j] = 0.f;

    if constexpr (RAGGED) {
        // bounds-checked build (stream end only): one row at a time
        // a sample past the end (0.0f) in the domain the stencil works in
        const float past_re = fold_unpack<TIN, COND, true>() ? 128.f : 0.f;
        const float past_im = fold_unpack<TIN, COND, true>() ? 128.f : norm_stencil<TIN, COND>() ? a.past_im : 0.f;
#pragma unroll
        for (int j = 0; j < kMaxTaps - 1; ++j) {
            halo.xr[j] = past_re;
            halo.xi[j] = past_im;
        }
        RegisterExchange ex{halo, false, past_re, past_im};
        auto q1_row = [&](uint32_t r) -> Q1Row {
            if constexpr (COND != kCorrectIq) return Q1Row{nullptr, 0, 0};
            const uint32_t chunk = (g0 + r) >> a.rpc_log2;
            const int K = chunk == a.q1_last_chunk ? (int) a.q1_last_k : (int) (512u << a.rpc_log2);
            return Q1Row{a.q1_table + 2ull * chunk * a.q1_stride, (int) ((((g0 + r) & rpc_mask) << 9) + 8u * lane), K};
        };
        if ((ge & rpc_mask) != 0 && re < a.rows_readable) {
            float unused[4];
            process_row<TIN, TOUT, COND, true>(load(re), a, ex, lane_valid(re), kAtanLead, unused, q1_row(re));
        }
        for (uint32_t r = re; r-- > rb;) {
            float y[4];
            const int valid = lane_valid(r);
            ex.edge = ((g0 + r) & rpc_mask) == 0u;
            process_row<TIN, TOUT, COND, true>(load(r), a, ex, valid, kAtanLead, y, q1_row(r));
            const bool head = (((g0 + r) & rpc_mask) == 0u) && lane < a.head_quads;
            char *dst = out_address<S16>(a.out, (uint64_t) r * kRowOutputs + 4u * lane);
            if (!head) {
                if (valid >= 16) {
                    store_quad<S16>(dst, y, a.out_scale);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (4 * j < valid) {
                            if constexpr (S16) reinterpret_cast<int16_t *>(dst)[j] = to_s16(y[j], a.out_scale);
                            else reinterpret_cast<float *>(dst)[j] = y[j];
                        }
                }
            }
        }
    } else {
        // A wavefront owns at most kGroup rows (4 KiB of input by default) and asks for all
        // of them, plus the row that follows when the segment ends inside a chunk,
        // before it computes anything: short-lived wavefronts over a compact,
        // advancing window of memory are what reaches the HBM streaming rate
        // (tools/microbench/stream_copy.hip: 1-4 rows per wavefront 0.80-0.83 of
        // peak, 32 rows 0.74, 256 rows 0.65).
        // a sample past the chunk end: 0.0f, which is 128 where the unpack bias is folded and
        // (0, s) where the stencil is divided by its first tap
        const float past_re = fold_unpack<TIN, COND, false>() ? 128.f : 0.f;
        const float past_im = fold_unpack<TIN, COND, false>() ? 128.f : norm_stencil<TIN, COND>() ? a.past_im : 0.f;
        const bool need_next = (ge & rpc_mask) != 0u;             // segment ends inside a chunk
        const bool next_in_buffer = need_next && re < a.rows_readable;  // ... and the next row is ours to read
        const bool next_is_halo = need_next && !next_in_buffer && a.halo != nullptr;
        // The row after the segment is asked for first (it is needed first), from
        // wherever it lives: this buffer, the next shard's first DEMOD_HALO_BYTES
        // bytes, or -- at a chunk or stream end, where the reference's buffers hold
        // zeros -- a row of bytes that unpack to zeros (kPastEndRow; the builds that have no
        // such byte read row rb instead and select afterwards): the prologue has no branch
        // around a load.
        const bool have_next = next_in_buffer || next_is_halo;
        constexpr bool kZeroRow = shift_origin<COND>() && !fold_unpack<TIN, COND, false>();
        const uint8_t *nbase = next_in_buffer ? a.in + (uint64_t) re * kRowBytes
                               : next_is_halo ? a.halo
                               : kZeroRow     ? reinterpret_cast<const uint8_t *>(kPastEndRow.w)
                                              : a.in + (uint64_t) rb * kRowBytes;
        // Only lane 0 of that row matters when the look-ahead fits its 16 bytes (the
        // short stencils): then every lane reads lane 0's bytes -- one broadcast
        // request instead of a 64-lane row (1/kGroup less read traffic).
        constexpr bool lane0_only = (2 * (TOUT - 1) + (TIN > 0 ? TIN - 1 : 0)) <= 8;
        const uint32_t nlane = lane0_only ? 0u
                               : next_is_halo ? (lane < DEMOD_HALO_BYTES / 16u ? lane : 0u)
                                              : lane;
        // ---- exchange lines in LDS (see LdsExchange) ------------------------
        constexpr int kExX = 2 * (TIN > 1 ? TIN - 1 : 0);  // floats a lane publishes per row: re, im of its first samples
        constexpr int kExD = TOUT - 1;                     // ... and its first discriminator outputs
        __shared__ float lds_x[kWavesPerBlock][2][64 * (kExX > 0 ? kExX : 1)];
        __shared__ float lds_d[kWavesPerBlock][2][64 * (kExD > 0 ? kExD : 1)];
        lds_float *wr_x[2], *wr_d[2];
        const lds_float *rd_x[2], *rd_d[2];
        // A wavefront's two lines lie one behind the other, 128 slots in all: the slot after slot s of line p is slot
        // (64 p + s + 1) mod 128 -- lane 63 of line 0 lands on slot 0 of line 1 and the other way round by itself, no
        // select (v_cndmask_b32 with VCC is the slowest VALU operation this kernel had: 8.7 cycles,
        // profiles/r1_microbench_valu_rate2.txt; round 3)
        lds_float *const line_x = (lds_float *) &lds_x[wave][0][0];
        lds_float *const line_d = (lds_float *) &lds_d[wave][0][0];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const uint32_t own = 64u * (uint32_t) p + lane, next = (own + 1u) & 127u;
            wr_x[p] = line_x + own * kExX;
            wr_d[p] = line_d + own * kExD;
            rd_x[p] = line_x + next * kExX;
            rd_d[p] = line_d + next * kExD;
        }
        // ---- loads and waits spelled out ------------------------------------
        // hipcc keeps one vmcnt score for loads and stores together and, with
        // both kinds pending, drains it to 0 before any use.  Here every row's
        // result is stored as soon as it is computed (no result buffer, ~30
        // fewer VGPRs) while later rows' loads are still
        // in flight, so the loads are issued from inline asm (invisible to
        // hipcc) and waited for by count: gfx950 retires vector-memory
        // operations in issue order, and when row k is needed exactly
        // kFlight-1-k younger loads and k stores have been issued after its
        // load -- kFlight-1 operations for every k when the whole segment is asked for up front
        // (kFlight == kGroup), younger_than_row(k) in general.  Requirements kept below:
        // always the same loads (a clipped segment asks for row rb again in
        // place of the rows it lacks; those are never computed, only waited for
        // at the end), exactly one store per computed row, nothing else that
        // touches vector memory before the last row is stored, wait statements
        // that name the loaded registers.  tools/check_asm_waits.py replays the
        // generated assembly against these rules.
        constexpr int nx = 2 * (TOUT - 1) + (TIN > 0 ? TIN - 1 : 0);
        const uint8_t *np = nbase + 16u * nlane;
        // Rows are addressed as (wave-uniform base in scalar registers) + (this lane's 16 bytes
        // of a row, one VGPR for every load and every float32 store of the segment): no
        // per-row 64-bit address arithmetic on the VALU.
        const uint32_t lane_in = 16u * lane;
        const uint32_t lane_out = (S16 ? 8u : 16u) * lane;
        // The body exists twice: FULL (all kGroup rows: every segment but the first and last of
        // a launch, and every one when chunks are at least kGroup rows long) is straight-line
        // code, so that hipcc schedules across the rows; the other form serves a clipped
        // segment.  Each issues its own loads: registers of a load in flight must not cross a
        // branch (hipcc would be free to copy them on the way).
        auto segment = [&](auto full_tag) {
            constexpr bool FULL = decltype(full_tag)::value;
            u32x4 qn4 = {0, 0, 0, 0};
            u32x2 qn2 = {0, 0};
            uint32_t qn1 = 0;
            if constexpr (nx > 4) asm volatile("global_load_dwordx4 %0, %1, off" DEMOD_LOAD_POLICY : "=v"(qn4) : "v"(np) : "memory");
            else if constexpr (nx > 2) asm volatile("global_load_dwordx2 %0, %1, off" DEMOD_LOAD_POLICY : "=v"(qn2) : "v"(np) : "memory");
            else asm volatile("global_load_dword %0, %1, off" DEMOD_LOAD_POLICY : "=v"(qn1) : "v"(np) : "memory");
            u32x4 qa[kFlight];
            // row k of the segment (counted from its end); a clipped segment asks for row rb in place of the rows it lacks
            auto row_at = [&](int k) -> const uint8_t * {
                const uint32_t row = (FULL || re >= rb + 1u + (uint32_t) k) ? re - 1u - (uint32_t) k : rb;
                return a.in + (uint64_t) row * kRowBytes;
            };
#pragma unroll
            for (int k = 0; k < kFlight; ++k) ask_row(qa[k], lane_in, row_at(k));
            __builtin_amdgcn_sched_barrier(0);
            // the look-ahead row was asked for first: kFlight younger loads may stay in flight
            if constexpr (nx > 4) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(qn4) : "i"(kFlight) : "memory");
            else if constexpr (nx > 2) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(qn2) : "i"(kFlight) : "memory");
            else asm volatile("s_waitcnt vmcnt(%1)" : "+v"(qn1) : "i"(kFlight) : "memory");
            {
                uint4 qn = make_uint4(0, 0, 0, 0);
                if constexpr (nx > 4) qn = make_uint4(qn4.x, qn4.y, qn4.z, qn4.w);
                else if constexpr (nx > 2) { qn.x = qn2.x; qn.y = qn2.y; }
                else qn.x = qn1;
                Halo h = halo;
                next_row_lookahead<TIN, TOUT, COND>(qn, a, h);
                // the first row processed uses line 0 and, in lane 63, slot 0 of line 1
                // (kZeroRow: a segment that ends its chunk has read zeros and computed the
                // past-the-end values like any others)
#pragma unroll
                for (int j = 0; j < kExX / 2; ++j) {
                    wr_x[1][2 * j] = (kZeroRow || have_next) ? h.xr[j] : past_re;
                    wr_x[1][2 * j + 1] = (kZeroRow || have_next) ? h.xi[j] : past_im;
                }
#pragma unroll
                for (int j = 0; j < TOUT - 1; ++j) wr_d[1][j] = (kZeroRow || have_next) ? h.d[j] : 0.f;
            }
            float lead = kAtanLead;
            asm volatile("" : "+v"(lead));  // one register for the whole segment, see fast_atan2
#pragma unroll
            for (int k = 0; k < kGroup; ++k) {
                if constexpr (!FULL) {
                    // rows a clipped segment lacks are not computed (wave-uniform; the computed
                    // rows come first, so the counts above hold for each of them)
                    if (re < rb + 1u + (uint32_t) k) break;
                }
                const uint32_t r = re - 1u - (uint32_t) k;
                u32x4 &q = qa[k % kFlight];
                switch (k) {   // (k is a constant once the loop is unrolled: one wait statement survives)
                case 0: wait_vm<younger_than_row(0)>(q); break;
                case 1: wait_vm<younger_than_row(1)>(q); break;
                case 2: wait_vm<younger_than_row(2)>(q); break;
                case 3: wait_vm<younger_than_row(3)>(q); break;
                case 4: wait_vm<younger_than_row(4)>(q); break;
                case 5: wait_vm<younger_than_row(5)>(q); break;
                case 6: wait_vm<younger_than_row(6)>(q); break;
                default: wait_vm<younger_than_row(7)>(q); break;
                }
                float y[4];
                LdsExchange ex{wr_x[k & 1], rd_x[k & 1], wr_d[k & 1], rd_d[k & 1],
                               SPAN && lane == 0u && ((g0 + r) & rpc_mask) == 0u, past_re, past_im};
                process_row<TIN, TOUT, COND, false>(make_uint4(q.x, q.y, q.z, q.w), a, ex, 16, lead, y);
                // (the first lanes of a chunk's first row hold interior-formula values for
                // outputs that are not interior: chunk_head / demod_heads overwrites them)
                store_row<S16, TIN == 0>(out_address<S16>(a.out, (uint64_t) r * kRowOutputs), lane_out, y, a.out_scale);
                if (k + kFlight < kGroup) ask_row(q, lane_in, row_at(k + kFlight));   // into the registers this row's bytes arrived in
            }
            if constexpr (!FULL) {
                // the stand-in loads of the missing rows land in registers hipcc believes free
                // from here on: wait for them before anything else runs
#pragma unroll
                for (int k = 0; k < kFlight; ++k) asm volatile("s_waitcnt vmcnt(0)" : "+v"(qa[k]) : : "memory");
            }
        };
        if (re - rb == (uint32_t) kGroup) segment(std::true_type{});
        else segment(std::false_type{});
        // ---- the heads of the chunks that start in this segment ---------------
        // (one, in the segment's first row, unless chunks are shorter than a segment)
        if constexpr (SPAN) {
            if (a.fused_heads != 0u && a.heads_packed != 0u) {
                // every chunk that starts in the segment, sixteen lanes each, in ONE pass of chunk_head (round 3)
                const uint32_t rpc = 1u << a.rpc_log2;
                const uint32_t first = rb + ((rpc - ((g0 + rb) & rpc_mask)) & rpc_mask);   // the segment's first row that starts a chunk
                for (uint32_t pass = 0; first + 4u * pass * rpc < re; ++pass) {   // four heads per pass
                    const uint32_t r = first + (4u * pass + (lane >> 4)) * rpc;
                    const bool has = r < re;
                    const uint32_t rr = has ? r : rb;
                    chunk_head<COND, S16>(a, a.in + (uint64_t) rr * kRowBytes, (uint64_t) rr * kRowOutputs, lane & 15u,
                                          has ? 4u * a.head_quads : 0u, lane & ~15u);
                }
            } else if (a.fused_heads != 0u) {
                for (uint32_t r = rb; r < re; ++r)
                    if (((g0 + r) & rpc_mask) == 0u)
                        chunk_head<COND, S16>(a, a.in + (uint64_t) r * kRowBytes, (uint64_t) r * kRowOutputs, lane, 4u * a.head_quads);
            }
        } else {
            if (a.fused_heads != 0u && (gs & rpc_mask) == 0u && gs >= g0)
                chunk_head<COND, S16>(a, a.in + (uint64_t) rb * kRowBytes, (uint64_t) rb * kRowOutputs, lane, 4u * a.head_quads);
        }
    }
}

// ---------------------------------------------------------------------------
// chunk heads: literal replay over a short window (one thread per chunk)
// ---------------------------------------------------------------------------
constexpr int kHeadMaxOut = 32;                                  // ceil4(max(2*8-1, 4*8-2))
constexpr int kHeadMaxD = kHeadMaxOut + kMaxSections + 2;        // discriminator values needed
constexpr int kHeadMaxZ = 2 * kHeadMaxD + 2 * kMaxSections + 2;  // complex cells simulated

// SI / SO >= 0: section counts fixed at compile time -- every loop unrolls and
// every array lives in registers (the designs the CLI defaults produce);
// SI = SO = -1: any section count up to kMaxSections, arrays in scratch.
template <int COND, int SI, int SO>
__global__ __launch_bounds__(64) void demod_heads(const HeadLaunch a) {
    constexpr bool kFixed = SI >= 0;
    constexpr int kH = kFixed ? 4 * (((2 * SO - 1 > 4 * SI - 2 ? 2 * SO - 1 : 4 * SI - 2) + 3) / 4) : kHeadMaxOut;
    constexpr int kD = kFixed ? kH + SO + 2 : kHeadMaxD;                    // d[0..W], W = H + So + 1
    constexpr int kZ = kFixed ? 2 * (kH + SO + 1) + 2 * SI + 2 : kHeadMaxZ;  // complex cells simulated
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= a.n_heads) return;
    const uint8_t *bytes = a.in + (uint64_t) c * a.chunk_bytes;
    const uint64_t out0 = (uint64_t) c * (a.chunk_bytes >> 2);  // element index of this chunk's first output
    const uint32_t chunk = (c == a.n_heads - 1) ? a.last_chunk_bytes : (uint32_t) a.chunk_bytes;
    const int K = (int) (chunk >> 1);   // complex samples in this chunk
    const int Ld = (int) (chunk >> 2);  // outputs of this chunk
    const int Si = kFixed ? SI : a.sec_in, So = kFixed ? SO : a.sec_out;
    const int H = kFixed ? kH : (int) a.head_outputs;
    const int W = H + So + 1;           // discriminator values d[0..W) feed outputs [0, H)

    float d[kD];
    float yc[kD + 1];
#pragma unroll
    for (int n = 0; n <= W; ++n) yc[n] = 0.f;

    // The window's bytes: the fixed builds fetch them with a few 16-byte loads up front (every
    // index below is a compile-time constant there, so the unpack reads bytes out of
    // registers); one byte load per use, 64 cache lines per wave-instruction, was most of this
    // kernel's time.  A chunk handled here is at least one row long, so the window exists.
    constexpr int kRawVec = kFixed ? (2 * (kZ + 2) + 15) / 16 : 1;
    uint32_t raw[4 * kRawVec];
    if constexpr (kFixed) {
#pragma unroll
        for (int v = 0; v < kRawVec; ++v) {
            const uint4 q = reinterpret_cast<const uint4 *>(bytes)[v];
            raw[4 * v] = q.x; raw[4 * v + 1] = q.y; raw[4 * v + 2] = q.z; raw[4 * v + 3] = q.w;
        }
    }
    auto q1 = [&](int k, int lane01, float v) -> float {  // -q1: minus the running offset of the sample's iteration
        if constexpr (COND != kCorrectIq) return v;
        const float2 off = q1_offset(a.q1_table + 2ull * c * a.q1_stride, k, K);
        return v - (lane01 ? off.y : off.x);
    };
    auto sample = [&](int k, int lane01) -> float {  // unpacked input, 0.0f past the chunk
        const int i = 2 * k + lane01;
        if constexpr (kFixed) {
            const uint8_t b = (uint8_t) (raw[i >> 2] >> (8 * (i & 3)));
            return k < K ? q1(k, lane01, unpack_byte<COND>(b, a.scale)) : 0.f;
        } else {
            return k < K ? q1(k, lane01, unpack_byte<COND>(bytes[i], a.scale)) : 0.f;
        }
    };

    if (Si > 0) {
        // complex stage, src/filter.c:242-258, cells [0, NZ) brought to their final value
        float zr[kZ], zi[kZ];
        const int NZ = 2 * W;
#pragma unroll
        for (int k = 0; k < NZ + 2 * Si + 2; ++k) zr[k] = zi[k] = 0.f;

        // -q2 in front of -L (generic build only): highpassDc's cells come first, computed
        // from a buffer of its own (zeros behind the chunk); its three overflow cells are
        // constants of the design and land where the -L stage's output buffer starts
        // (src/matrix.c:155-156, :260-264: filterRet + bufSize is both)
        constexpr int kP = kFixed ? 1 : kHeadMaxZ + kMaxSections + 8;
        float pr[kP], pi[kP];
        bool pre = false;
        if constexpr (!kFixed) {
            pre = a.sec_pre > 0;
            if (pre) {
                const int Sp = a.sec_pre;
                const int NP = NZ + Si + 2;  // the -L stage reads cells up to NZ + Si
                for (int k = 0; k < NP + Sp + 2; ++k) pr[k] = pi[k] = 0.f;
                for (int i = 0; i + Sp < NP; ++i) {
                    for (int m = 0; m < Sp; ++m) {
                        const int n = i + Sp + m;
                        const float b0 = a.sos_pre[m][0], b1 = a.sos_pre[m][1];
                        const float a0 = a.sos_pre[m][3], a1 = a.sos_pre[m][4], a2 = a.sos_pre[m][5];
                        float t = b0 * pr[n] + b1 * pr[n + 1] + 1.0f;
                        pr[n] = t - (a0 + a1 * sample(n, 0) + a2 * sample(n + 1, 0));
                        t = b0 * pi[n] + b1 * pi[n + 1];
                        pi[n] = t - (a0 + a1 * sample(n, 1) + a2 * sample(n + 1, 1));
                    }
                }
                float hr[2 * 2 + 2] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, hi[2 * 2 + 2] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                for (int u = 0; u < 2 * Sp - 1; ++u) {
                    for (int m = 0; m < Sp; ++m) {
                        const int t = u + 1 - Sp + m;  // cell K + t, inputs 0.0f
                        if (t < 0) continue;
                        const float b0 = a.sos_pre[m][0], b1 = a.sos_pre[m][1];
                        const float a0 = a.sos_pre[m][3], a1 = a.sos_pre[m][4], a2 = a.sos_pre[m][5];
                        float v = b0 * hr[t] + b1 * hr[t + 1] + 1.0f;
                        hr[t] = v - (a0 + a1 * 0.f + a2 * 0.f);
                        v = b0 * hi[t] + b1 * hi[t + 1];
                        hi[t] = v - (a0 + a1 * 0.f + a2 * 0.f);
                    }
                }
                for (int t = 0; t < 2 * Sp - 1; ++t) {
                    zr[t] = hr[t];
                    zi[t] = hi[t];
                }
            }
        }
        auto xin = [&](int n, int lane01) -> float {  // what the -L stage reads as its input
            if constexpr (!kFixed) {
                if (pre) return lane01 ? pi[n] : pr[n];
            }
            return sample(n, lane01);
        };
#pragma unroll
        for (int i = 0; i + Si < NZ; ++i) {  // iteration i finalises cell i + Si
#pragma unroll
            for (int m = 0; m < Si; ++m) {
                const int n = i + Si + m;
                const float b0 = a.sos_in[m][0], b1 = a.sos_in[m][1];
                const float a0 = a.sos_in[m][3], a1 = a.sos_in[m][4], a2 = a.sos_in[m][5];
                float t = b0 * zr[n] + b1 * zr[n + 1] + 1.0f;
                zr[n] = t - (a0 + a1 * xin(n, 0) + a2 * xin(n + 1, 0));
                t = b0 * zi[n] + b1 * zi[n + 1];
                zi[n] = t - (a0 + a1 * xin(n, 1) + a2 * xin(n + 1, 1));
            }
        }
        // cells K .. K+2Si-2 overflow into the output stage's buffer.  Their
        // "input" is the aliased head of the complex stage's own output
        // (x[K + t] == z[t]) -- or zeros when the stage reads a buffer of its
        // own (-q2); the last 2Si-1 iterations (u = 0 .. 2Si-2) write them.
        const bool own = a.in_separate == 0;
        float ovr[2 * (kFixed ? SI : kMaxSections) + 1], ovi[2 * (kFixed ? SI : kMaxSections) + 1];
#pragma unroll
        for (int t = 0; t <= 2 * Si; ++t) ovr[t] = ovi[t] = 0.f;
#pragma unroll
        for (int u = 0; u < 2 * Si - 1; ++u) {  // iteration K - 2Si + 1 + u
#pragma unroll
            for (int m = 0; m < Si; ++m) {
                const int t = u + 1 - Si + m;  // cell K + t
                if (t < 0) continue;
                const float b0 = a.sos_in[m][0], b1 = a.sos_in[m][1];
                const float a0 = a.sos_in[m][3], a1 = a.sos_in[m][4], a2 = a.sos_in[m][5];
                float v = b0 * ovr[t] + b1 * ovr[t + 1] + 1.0f;
                ovr[t] = v - (a0 + a1 * (own ? zr[t] : 0.f) + a2 * (own ? zr[t + 1] : 0.f));
                v = b0 * ovi[t] + b1 * ovi[t + 1];
                ovi[t] = v - (a0 + a1 * (own ? zi[t] : 0.f) + a2 * (own ? zi[t + 1] : 0.f));
            }
        }
#pragma unroll
        for (int t = 0; t < 2 * Si - 1; ++t) {
            if (2 * t <= W) yc[2 * t] = ovr[t];
            if (2 * t + 1 <= W) yc[2 * t + 1] = ovi[t];
        }
#pragma unroll
        for (int n = 0; n < W; ++n)
            d[n] = n < Ld ? discriminate_literal(zr[2 * n], zi[2 * n], zr[2 * n + 1], zi[2 * n + 1]) : 0.f;
    } else {
#pragma unroll
        for (int n = 0; n < W; ++n)
            d[n] = n < Ld ? discriminate_literal(sample(2 * n, 0), sample(2 * n, 1), sample(2 * n + 1, 0),
                                         sample(2 * n + 1, 1))
                          : 0.f;
    }
    d[W] = 0.f;  // never reaches an output below H; keeps the window reads defined

    // real stage, src/filter.c:221-230: iteration i finalises cell i + So
#pragma unroll
    for (int i = 0; i + So < H; ++i) {
        if (i >= Ld) break;
#pragma unroll
        for (int m = 0; m < So; ++m) {
            const int n = i + So + m;
            const float b0 = a.sos_out[m][0], b1 = a.sos_out[m][1];
            const float a0 = a.sos_out[m][3], a1 = a.sos_out[m][4], a2 = a.sos_out[m][5];
            const float t = b0 * yc[n] + b1 * yc[n + 1] + 1.0f;
            yc[n] = t - (a0 + a1 * d[n] + a2 * d[n + 1]);
        }
    }
#pragma unroll
    for (int n = 0; n < H; n += 4)
    {
        const float q[4] = {yc[n], yc[n + 1], yc[n + 2], yc[n + 3]};
        if (a.out_s16) store_quad<true>(out_address<true>(a.out, out0 + n), q, a.out_scale);
        else store_quad<false>(out_address<false>(a.out, out0 + n), q, a.out_scale);
    }
}

// ---------------------------------------------------------------------------
// chunk heads of designs with more than four sections per stage: one wavefront per head
// ---------------------------------------------------------------------------
// demod_heads<COND, -1, -1> keeps its cells in scratch memory, one thread per chunk: with a
// few thousand chunks per dispatch that is one long latency chain (0.15 ms per 2^30 bytes of
// default chunks at -L -D 9 -d 9).  chunk_head's lane-parallel replay has no such chain; here it
// runs with room for kMaxSections sections, one wavefront per head.  Same operations on the same
// values as the thread-per-chunk kernel (tests compare the two bit for bit), highpassDc in front
// of -L included (PRE).  Not for -q1 (chunk_head has no offsets table); the chunk must hold the
// window the 64 lanes read (264 bytes: every chunk of a whole row does).
constexpr int kHeadWaves = 4;
template <int COND, bool S16>
__global__ __launch_bounds__(64 * kHeadWaves) void demod_heads_wave(const HeadLaunch a) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t head = blockIdx.x * (uint32_t) kHeadWaves + (threadIdx.x >> 6);
    if (head >= a.n_heads) return;
    chunk_head<COND, S16, kWaveHeadSections, HeadLaunch, true>(a, a.in + (uint64_t) head * a.chunk_bytes,
                                                               (uint64_t) head * (a.chunk_bytes >> 2), lane, a.head_outputs);
}

// ---------------------------------------------------------------------------
// chunks shorter than one row: literal whole-chunk replay (one thread per chunk)
// ---------------------------------------------------------------------------
constexpr uint32_t kTinyPad = 4 * kMaxSections + 8;

// A thread's scratch, element i at p[64 i]: the 64 threads of a wavefront walk their buffers in
// step, so element i of all of them is one 256-byte line (a region of its own per thread made
// every access 64 separate transactions).
struct TinyScratch {
    float *p;
    __device__ __forceinline__ float &operator[](uint64_t i) const { return p[i * 64u]; }
    __device__ __forceinline__ TinyScratch operator+(uint64_t k) const { return TinyScratch{p + k * 64u}; }
};

template <int COND>
__global__ __launch_bounds__(64) void demod_tiny(const TinyLaunch a) {
    const uint64_t tid = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t nthreads = (uint64_t) gridDim.x * blockDim.x;
    const uint64_t total = a.n_chunks + (a.tail_bytes ? 1 : 0);
    const TinyScratch F{a.work + (uint64_t) blockIdx.x * 64u * a.work_stride + threadIdx.x};
    const int Si = a.sec_in, So = a.sec_out;

    for (uint64_t c = tid; c < total; c += nthreads) {
        const uint32_t B = c < a.n_chunks ? (uint32_t) a.chunk_bytes : a.tail_bytes;
        const uint8_t *bytes = a.in + c * a.chunk_bytes;
        const uint64_t out0 = c * (a.chunk_bytes >> 2);
        const uint32_t Ld = B >> 2;
        // same buffer layout as the reference (src/matrix.c:184-187,228,234):
        // work = 2B floats (4B with the input filter) and a separate B-float
        // demod buffer; both padded so that the recurrences' reach past the
        // end (which the reference leaves to the heap) stays in zeroed scratch.
        const TinyScratch dem = F + 4 * a.chunk_bytes + kTinyPad;
        // -q2: the complex stage reads highpassDc's own buffer (zeros behind it)
        // and writes where the unpacked samples would otherwise be
        const bool pre = a.sec_pre > 0;  // -q2 in front of -L: highpassDc from its own buffer into F first
        const bool x_apart = a.in_separate || pre;
        const TinyScratch X = x_apart ? dem + a.chunk_bytes + kTinyPad : F;
        // What the reference's calloc/memset leave at zero, as far as anything below reads it
        // (the scratch is most of this kernel's memory traffic): the stages' outputs, which
        // accumulate, and `reach` floats behind every buffer -- a stage with S sections reads
        // 4 S - 1 floats past its input and output.  The samples' own B floats are overwritten.
        {
            const int smax = Si > So ? (Si > a.sec_pre ? Si : a.sec_pre) : (So > a.sec_pre ? So : a.sec_pre);
            const uint32_t reach = 4u * (uint32_t) smax + 8u;
            const uint32_t y2_at = (Si > 0 && !a.in_separate) ? 2u * B : B;  // where the output stage's cells start
            for (uint32_t i = x_apart ? 0u : B; i < y2_at + Ld + reach; ++i) F[i] = 0.f;
            for (uint32_t i = 0; i < Ld + reach; ++i) dem[i] = 0.f;
            if (x_apart)
                for (uint32_t i = B; i < B + reach; ++i) X[i] = 0.f;
        }
        auto sample = [&](uint32_t i, uint8_t byte) -> float {
            float v = unpack_byte<COND>(byte, a.scale);
            if constexpr (COND == kCorrectIq) {
                const float2 off = q1_offset(a.q1_table + 2ull * c * a.q1_stride, (int) (i >> 1), (int) (B >> 1));
                v = v - ((i & 1u) ? off.y : off.x);
            }
            return v;
        };
        // without a complex stage the samples go straight into the discriminator: no trip
        // through the scratch (the stages are what has to be literal, this is element by element)
        const bool direct = Si == 0 && !pre;
        const bool words = (((uintptr_t) bytes) & 3u) == 0;  // (chunks are whole words: four bytes per request)
        if (direct) {
            for (uint32_t n = 0; n < Ld; ++n) {
                float v[4];
                if (words) {
                    const uint32_t w = *reinterpret_cast<const uint32_t *>(bytes + 4 * n);
#pragma unroll
                    for (uint32_t j = 0; j < 4; ++j) v[j] = sample(4 * n + j, (uint8_t) (w >> (8 * j)));
                } else {
#pragma unroll
                    for (uint32_t j = 0; j < 4; ++j) v[j] = sample(4 * n + j, bytes[4 * n + j]);
                }
                dem[n] = discriminate_literal(v[0], v[1], v[2], v[3]);
            }
        } else if (words) {
            for (uint32_t i = 0; i + 4 <= B; i += 4) {
                const uint32_t w = *reinterpret_cast<const uint32_t *>(bytes + i);
#pragma unroll
                for (uint32_t j = 0; j < 4; ++j) X[i + j] = sample(i + j, (uint8_t) (w >> (8 * j)));
            }
        } else {
            for (uint32_t i = 0; i < B; ++i) X[i] = sample(i, bytes[i]);
        }
        TinyScratch src = F;
        TinyScratch y2 = F + B;
        if (pre) {
            for (uint32_t i = 0; i < B; i += 2) {
                for (int m = 0; m < a.sec_pre; ++m) {
                    const uint32_t n = i + 2 * a.sec_pre + 2 * m;
                    const float b0 = a.sos_pre[m][0], b1 = a.sos_pre[m][1];
                    const float a0 = a.sos_pre[m][3], a1 = a.sos_pre[m][4], a2 = a.sos_pre[m][5];
                    float t = b0 * F[n] + b1 * F[n + 2] + 1.0f;
                    F[n] = t - (a0 + a1 * X[n] + a2 * X[n + 2]);
                    t = b0 * F[n + 1] + b1 * F[n + 3];
                    F[n + 1] = t - (a0 + a1 * X[n + 1] + a2 * X[n + 3]);
                }
            }
        }
        if (Si > 0) {
            const TinyScratch x = pre ? F : X, y = a.in_separate ? F : F + B;  // -L: x[B + t] aliases y[t], exactly as in the reference
            for (uint32_t i = 0; i < B; i += 2) {
                for (int m = 0; m < Si; ++m) {
                    const uint32_t n = i + 2 * Si + 2 * m;
                    const float b0 = a.sos_in[m][0], b1 = a.sos_in[m][1];
                    const float a0 = a.sos_in[m][3], a1 = a.sos_in[m][4], a2 = a.sos_in[m][5];
                    float t = b0 * y[n] + b1 * y[n + 2] + 1.0f;
                    y[n] = t - (a0 + a1 * x[n] + a2 * x[n + 2]);
                    t = b0 * y[n + 1] + b1 * y[n + 3];
                    y[n + 1] = t - (a0 + a1 * x[n + 1] + a2 * x[n + 3]);
                }
            }
            src = y;
            y2 = y + B;
        }
        if (!direct)
            for (uint32_t n = 0; n < Ld; ++n)
                dem[n] = discriminate_literal(src[4 * n], src[4 * n + 1], src[4 * n + 2], src[4 * n + 3]);
        for (uint32_t i = 0; i < Ld; ++i) {
            for (int m = 0; m < So; ++m) {
                const uint32_t n = i + So + m;
                const float b0 = a.sos_out[m][0], b1 = a.sos_out[m][1];
                const float a0 = a.sos_out[m][3], a1 = a.sos_out[m][4], a2 = a.sos_out[m][5];
                const float t = b0 * y2[n] + b1 * y2[n + 1] + 1.0f;
                y2[n] = t - (a0 + a1 * dem[n] + a2 * dem[n + 1]);
            }
        }
        if (a.out_s16) {
            int16_t *out = static_cast<int16_t *>(a.out) + out0;
            for (uint32_t n = 0; n < Ld; ++n) out[n] = to_s16(y2[n], a.out_scale);
        } else {
            float *out = static_cast<float *>(a.out) + out0;
            for (uint32_t n = 0; n < Ld; ++n) out[n] = y2[n];
        }
    }
}

// ---------------------------------------------------------------------------
// -q1: the running offsets of correctIq
// ---------------------------------------------------------------------------
// correctIq (src/matrix.c:120-140) walks a chunk from both ends: iteration t corrects complex
// samples t and K-1-t with the offsets it finds and then moves them,
//     off += ((byte[t] - off) + (byte[K-1-t] - off)) * esr        (re and im lanes alike),
// and its static offsets carry on from chunk to chunk: a float recurrence over the whole stream,
// every step rounded, so it cannot be re-associated.  It contracts, though -- a start value is
// forgotten at the rate (1 - 2 esr) per iteration -- and the rounded step g is monotone in the
// offset wherever the offset is not tiny: g(x) = fl(x + p(x)) with p = fl(fl(fl(b - x) + fl(m - x)) esr)
// a non-increasing step function of x; for x <= y the exact sums satisfy
//     (y + p(y)) - (x + p(x)) >= (y - x)(1 - 2 esr) - J,   J = esr (2^-14 + 510 2^-22) <= esr 2^-12
// (what the roundings of b - x, m - x, their sum and the product can undo), so g(y) >= g(x) as soon
// as one ulp of the offsets exceeds J, i.e. for offsets above a.floor2 / 2 = esr 2^12 (1.6 at 125 kS/s:
// a float32 sweep finds no inversion above 0.8 in 3.6e8 pairs and 2e-6 of them below), and
// g(y) >= g(x) - J always.  So a thread replays the literal operations over
// a RUN-IN before the piece of the stream it writes, on a BRACKET: two copies of the state, started
// at 0 and at 255, between which the reference's own state lies (a running mean of bytes) and
// stays -- the first kQ1Careful iterations of the run-in push the lower copy down by a.widen >= 2 J
// per iteration (it starts at 0, below the monotone range), and from then on it has to stay above
// a.floor2 (checked every eight iterations: it cannot halve in between), else nothing is concluded.  When the
// two copies meet, bit for bit, the state is the reference's -- proven, not assumed -- and the
// thread goes on through its piece with it, writing the offsets every iteration starts from.  On bytes with some entropy the
// bracket closes after about ln(255 / ulp) / (2 esr) iterations plus a few time constants of luck
// (the run-in handed down by the host: a.warm).  On constant or near-constant bytes it never does:
// the rounded recurrence stalls in a dead band around its fixed point and WHERE depends on where it
// came from (round 2 started such run-ins from 127.5 and trusted them: offsets 1e-2 off the
// reference's on a constant stream).  A bracket still open at the piece's start doubles
// the run-in and starts over, until it reaches the call's first iteration, where the state is the
// one the previous call left (carry_in): exact for every input, at the price of a long walk for
// inputs that carry no information.
// -ffp-contract=off: the reference's generic build rounds the sum, the product and the update
// separately.  A piece is a part of a long chunk (a.parts threads share the chunk) or a.group
// whole chunks: the host deals the stream out to a fixed budget of threads (api.cpp
// run_offsets), the longest dependent chain is about the run-in + the piece.
constexpr int kQ1Span = 64;  // iterations per span: 8 pairs of 16-byte loads, one from each end of the chunk
constexpr int kQ1Careful = 4 * kQ1Span;  // iterations at the start of a run-in that widen the bracket (see above)

struct __attribute__((packed, aligned(1))) Unaligned16 { uint32_t x, y, z, w; };  // a chunk's far end need not be 16-byte aligned

// Where a thread is in the stream: chunk `cc` (at `chunk`, `len` bytes), iteration `t` of it.
struct Q1Pos {
    const uint8_t *chunk;
    uint64_t len, t, cc;
};

__global__ __launch_bounds__(64) void correct_iq_offsets(const OffsetLaunch a) {
    const uint64_t id = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t total = a.n_chunks + (a.tail_bytes ? 1 : 0);
    // what this thread writes: from iteration p0 of chunk c to iteration p1 of chunk c_last
    // (a piece of one chunk when chunks are shared, a.parts > 1; a.group whole chunks otherwise)
    const uint64_t c = a.parts > 1 ? id / a.parts : id * a.group;
    const uint32_t part = a.parts > 1 ? (uint32_t) (id - c * a.parts) : 0u;
    if (c >= total) return;
    const uint64_t c_last = a.parts > 1 ? c : (c + a.group < total ? c + a.group : total) - 1;
    const uint64_t T = a.chunk_bytes >> 2;  // iterations of a full chunk
    auto bytes_of = [&](uint64_t chunk) -> uint64_t { return chunk < a.n_chunks ? a.chunk_bytes : a.tail_bytes; };
    const uint64_t len = bytes_of(c);
    const uint64_t Tc = len >> 2;           // iterations of this chunk
    // Chunks of whole spans (256 bytes and more) take the span-wise path below: every thread of a
    // wavefront then runs the same instructions over its own bytes from the first iteration of its
    // run-in to its last one -- run-in and piece are cut in whole spans -- whichever chunk
    // boundaries it crosses on the way.  (Round 2's first form walked chunk by chunk: threads
    // that crossed at different times took turns, 124k iterations of wavefront time for 67k.)
    const bool spans = (T % kQ1Span) == 0 && (a.warm % kQ1Span) == 0 && (a.retry % kQ1Span) == 0 && (((uintptr_t) a.in) & 15u) == 0;
    const uint64_t grain = spans ? (uint64_t) kQ1Span : 8u;
    const uint64_t per = ((Tc + a.parts - 1) / a.parts + grain - 1) / grain * grain;  // iterations per part
    const uint64_t p0 = (uint64_t) part * per;
    if (p0 >= Tc) return;
    const uint64_t p1 = c_last > c ? bytes_of(c_last) >> 2 : (p0 + per < Tc ? p0 + per : Tc);
    // (lo, hi): the bracket during the run-in; `off` = lo is the state from the piece's start on.
    // Each is the (re, im) offset pair as one packed pair: v_pk_add_f32 / v_pk_mul_f32 round each
    // element exactly as the scalar operations do; a wavefront alone pays ~9 cycles per dependent
    // operation either way (tools/microbench/chain_latency.hip), and the bracket's second copy is an
    // independent chain that fills the first one's latency
    pf2 off = {0.f, 0.f}, hi = {0.f, 0.f};
    float lowest = 0.f;        // of the bracket's lower copy, sampled every eight iterations behind the widened ones (a running minimum: no branch in the loop)
    const pf2 esr2 = {a.esr, a.esr};
    auto update1 = [&](pf2 &o, uint32_t fb0, uint32_t fb1, uint32_t mb0, uint32_t mb1) {
        const pf2 f = pf2{(float) fb0, (float) fb1} - o;
        const pf2 m = pf2{(float) mb0, (float) mb1} - o;
        o = o + (f + m) * esr2;
    };
    // bracket tags: 0 = one state (the piece), 1 = the bracket, 2 = the bracket, widened every iteration (first span)
    typedef std::integral_constant<int, 0> OneState;
    typedef std::integral_constant<int, 1> Bracket;
    typedef std::integral_constant<int, 2> BracketWidened;
    // (only the lower copy: the upper one is in the monotone range whenever the lower one's floor holds.  A relative
    // step as well as an absolute one: fl(x - W) must really be below x - J wherever x is, and W < ulp(x) / 2 above 8)
    auto widen = [&]() { off = off * pf2{1.f - 0x1p-20f, 1.f - 0x1p-20f} - pf2{a.widen, a.widen}; };
    auto byte_of = [](const uint4 &v, int i) -> uint32_t {
        const uint32_t w = i < 4 ? v.x : i < 8 ? v.y : i < 12 ? v.z : v.w;
        return (w >> (8 * (i & 3))) & 0xFFu;
    };
    // eight iterations from one 16-byte block of each end; `row` (or null) receives the offsets
    // every iteration starts from; BRACKET: the second copy of the state moves along
    // (row is 16-byte aligned here: two iterations' offsets leave as one store -- the stores go to
    // 64 different cache lines per instruction, and there are as few of them as the data allows)
    auto block = [&](auto bracket_tag, const uint4 &f, const uint4 &m, float2 *row) {
        constexpr int BRACKET = decltype(bracket_tag)::value;
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
            const pf2 before = off;
            update1(off, byte_of(f, 2 * j), byte_of(f, 2 * j + 1), byte_of(m, 14 - 2 * j), byte_of(m, 15 - 2 * j));
            if constexpr (BRACKET != 0) update1(hi, byte_of(f, 2 * j), byte_of(f, 2 * j + 1), byte_of(m, 14 - 2 * j), byte_of(m, 15 - 2 * j));
            if constexpr (BRACKET == 2) widen();
            if (row) *reinterpret_cast<float4 *>(row + j) = make_float4(before.x, before.y, off.x, off.y);
            update1(off, byte_of(f, 2 * j + 2), byte_of(f, 2 * j + 3), byte_of(m, 12 - 2 * j), byte_of(m, 13 - 2 * j));
            if constexpr (BRACKET != 0) update1(hi, byte_of(f, 2 * j + 2), byte_of(f, 2 * j + 3), byte_of(m, 12 - 2 * j), byte_of(m, 13 - 2 * j));
            if constexpr (BRACKET == 2) widen();
        }
        if constexpr (BRACKET == 1) lowest = fminf(lowest, fminf(off.x, off.y));
    };
    // iterations [t0, t1) of a chunk, any alignment, any length: the path of chunks below 256
    // bytes and of what a stream's last chunk has beyond its whole spans
    auto run = [&](auto bracket_tag, const uint8_t *chunk, uint64_t len, uint64_t t0, uint64_t t1, float2 *row) {
        constexpr int BRACKET = decltype(bracket_tag)::value;
        uint64_t t = t0;
        auto single = [&]() {
            const uint8_t *f = chunk + 2 * t, *m = chunk + len - 2 * t - 2;
            if (row) row[t] = make_float2(off.x, off.y);
            update1(off, f[0], f[1], m[0], m[1]);
            if constexpr (BRACKET != 0) update1(hi, f[0], f[1], m[0], m[1]);
            if constexpr (BRACKET == 2) widen();
            if constexpr (BRACKET == 1) lowest = fminf(lowest, fminf(off.x, off.y));
            ++t;
        };
        const bool wide = (len & 15u) == 0 && (((uintptr_t) chunk) & 15u) == 0;
        if (wide) {
            while (t < t1 && (t & 7u)) single();
            if (t + 8 <= t1) {
                uint4 f = *reinterpret_cast<const uint4 *>(chunk + 2 * t);
                uint4 m = *reinterpret_cast<const uint4 *>(chunk + len - 2 * t - 16);
                for (; t + 8 <= t1; t += 8) {
                    const uint64_t tn = t + 16 <= t1 ? t + 8 : t;  // one block ahead
                    const uint4 fn = *reinterpret_cast<const uint4 *>(chunk + 2 * tn);
                    const uint4 mn = *reinterpret_cast<const uint4 *>(chunk + len - 2 * tn - 16);
                    block(bracket_tag, f, m, row ? row + t : nullptr);
                    f = fn;
                    m = mn;
                }
            }
        }
        while (t < t1) single();
    };
    float2 *const table = reinterpret_cast<float2 *>(a.table);
    const uint64_t first = c * T + p0;   // where the written piece starts, in iterations of the whole call
    // The thread is alone with a long dependent chain, so nothing but the arithmetic may be
    // on it: a span's 16 loads are requested while the span before it is worked through
    // (one pair of loads ahead left the chain waiting ~1 us for every pair), into register
    // sets that take turns, with no branch between a request and its use (hipcc would sink
    // the loads to the use).
    auto request = [&](const Q1Pos &p, uint4 (&f)[8], uint4 (&m)[8]) {
        const uint4 *fp = reinterpret_cast<const uint4 *>(p.chunk + 2 * p.t);
        const Unaligned16 *mp = reinterpret_cast<const Unaligned16 *>(p.chunk + p.len - 2 * p.t - 16);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            f[j] = fp[j];
            const Unaligned16 u = mp[-j];
            m[j] = make_uint4(u.x, u.y, u.z, u.w);
        }
    };
    // the span after p: the next one of the chunk, or the first one of the next chunk (only
    // chunks before this thread's last are ever finished here, and those are full ones)
    auto after = [&](const Q1Pos &p) -> Q1Pos {
        Q1Pos n = p;
        n.t += kQ1Span;
        if (n.t == T && p.cc < c_last) {
            n.chunk += a.chunk_bytes;
            n.cc = p.cc + 1;
            n.len = bytes_of(n.cc);
            n.t = 0;
        }
        return n;
    };
    Q1Pos cur{a.in, bytes_of(0), 0, 0};
    auto phase = [&](auto write_tag, uint64_t left) {
        constexpr bool kWrite = decltype(write_tag)::value;
        if (left == 0) return;
        auto work = [&](const uint4 (&f)[8], const uint4 (&m)[8], const Q1Pos &p) {
            float2 *row = kWrite ? table + p.cc * a.stride + p.t : nullptr;
#pragma unroll
            for (int j = 0; j < 8; ++j) block(std::integral_constant<int, kWrite ? 0 : 1>{}, f[j], m[j], kWrite ? row + 8 * j : nullptr);
        };
        uint4 fa[8], ma[8], fb[8], mb[8];
        request(cur, fa, ma);
        while (left >= 2) {
            const Q1Pos n1 = after(cur);
            request(n1, fb, mb);
            __builtin_amdgcn_sched_barrier(0);
            work(fa, ma, cur);
            const Q1Pos n2 = after(n1);
            request(left > 2 ? n2 : n1, fa, ma);  // nothing beyond the phase is asked for (n1 again, dropped)
            __builtin_amdgcn_sched_barrier(0);
            work(fb, mb, n1);
            cur = n2;
            left -= 2;
        }
        if (left == 1) {
            work(fa, ma, cur);
            cur = after(cur);
        }
    };
    // ---- the run-in: a bracket that has to close (see above) ---------------------------------
    for (uint64_t warm = a.warm, attempt = 0;; ++attempt) {
        const bool from_start = first <= warm;     // the run-in reaches the call's first iteration: the carried state, exact
        const uint64_t g = from_start ? 0 : first - warm;
        const uint64_t cc = g / T, t0 = g - cc * T;
        off = from_start ? pf2{a.carry_in[0], a.carry_in[1]} : pf2{0.f, 0.f};
        hi = from_start ? off : pf2{255.f, 255.f};
        lowest = 3.0e38f;
        // the first kQ1Careful iterations widen the bracket (its lower copy starts below the monotone range) ...
        uint64_t careful = from_start ? 0 : (first - g < (uint64_t) kQ1Careful ? first - g : (uint64_t) kQ1Careful);
        const uint64_t plain_spans = (first - g - careful) / kQ1Span;
        {
            uint64_t w = cc, t = t0;
            while (careful) {   // (a span lies in one chunk when chunks are whole spans; tiny chunks: piece by piece)
                const uint64_t Tw = bytes_of(w) >> 2;
                const uint64_t n = careful < Tw - t ? careful : Tw - t;
                run(BracketWidened{}, a.in + w * a.chunk_bytes, bytes_of(w), t, t + n, nullptr);
                careful -= n;
                t += n;
                if (t == Tw) {
                    ++w;
                    t = 0;
                }
            }
            // ... and the rest runs on the plain bracket, its lower copy checked against the floor
            if (!spans) {
                for (; w < c; ++w, t = 0) run(Bracket{}, a.in + w * a.chunk_bytes, a.chunk_bytes, t, T, nullptr);
                if (w == c && t < p0) run(Bracket{}, a.in + c * a.chunk_bytes, len, t, p0, nullptr);   // the run-in's last stretch lies in this chunk
            } else if (plain_spans) {
                cur = Q1Pos{a.in + w * a.chunk_bytes, bytes_of(w), t, w};
                phase(std::false_type{}, plain_spans);
            }
        }
        // closed, and the lower copy never left the range where the step is monotone: the state is the reference's
        if (from_start || (lowest >= a.floor2 && __float_as_uint(off.x) == __float_as_uint(hi.x) && __float_as_uint(off.y) == __float_as_uint(hi.y))) break;
        // still open: a little further back (a bracket that is an ulp wide closes at the rate of one time constant),
        // then twice as far each time (bytes that carry no information: back to the call's first iteration)
        warm = attempt == 0 ? warm + a.retry : warm > (~0ull >> 2) ? ~0ull : 2 * warm;
    }
    // ---- the piece ----------------------------------------------------------------------------
    if (!spans) {
        for (uint64_t w = c; w <= c_last; ++w)
            run(OneState{}, a.in + w * a.chunk_bytes, bytes_of(w), w == c ? p0 : 0, w == c_last ? p1 : T, table + w * a.stride);
    } else {
        cur = Q1Pos{a.in + c * a.chunk_bytes, len, p0, c};
        const uint64_t last = c_last * T + p1;  // in iterations of the whole stream
        const uint64_t whole = (last - first) / kQ1Span;
        phase(std::true_type{}, whole);
        // (what a stream's last chunk has beyond its whole spans)
        if (first + whole * kQ1Span < last)
            run(OneState{}, a.in + c_last * a.chunk_bytes, bytes_of(c_last), first + whole * kQ1Span - c_last * T, p1, table + c_last * a.stride);
    }
    if (c_last == total - 1 && p1 == (bytes_of(c_last) >> 2)) {
        a.carry_out[0] = off.x;
        a.carry_out[1] = off.y;
    }
}

// ---------------------------------------------------------------------------
// synthetic input and self-test probes
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint64_t splitmix64_word(uint64_t seed, uint64_t j) {
    uint64_t z = seed + (j + 1) * 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

__global__ __launch_bounds__(256) void synth_fill(uint64_t *dst, uint64_t first_word, uint64_t n_pairs,
                                                  uint64_t seed) {
    const uint64_t stride = (uint64_t) gridDim.x * blockDim.x;
    for (uint64_t p = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; p < n_pairs; p += stride) {
        const uint64_t j = first_word + 2 * p;
        ulonglong2 v;
        v.x = splitmix64_word(seed, j);
        v.y = splitmix64_word(seed, j + 1);
        reinterpret_cast<ulonglong2 *>(dst)[p] = v;
    }
}

__global__ void atan2_probe(const float *y, const float *x, float *out, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = fast_atan2(y[i], x[i]);
}

__global__ void lane_probe(const float *in, float *out, float halo) {
    out[threadIdx.x] = from_next_lane(in[threadIdx.x], halo);
}

inline int hip_ok(hipError_t e) { return e == hipSuccess ? 0 : -(int) e - 1000; }

template <int TIN, int TOUT, int COND>
int launch_rows_t(const RowLaunch &a, hipStream_t s) {
    const uint32_t g0 = a.phase_rows, g1 = a.phase_rows + a.n_rows;
    const uint32_t segs = ((g1 + (1u << a.seg_log2) - 1u) >> a.seg_log2) - (g0 >> a.seg_log2);
    uint32_t blocks = (segs + (uint32_t) kWavesPerBlock - 1u) / (uint32_t) kWavesPerBlock;
#if DEMOD_XCD_INTERLEAVE
    blocks = (blocks + DEMOD_XCD_INTERLEAVE - 1u) / DEMOD_XCD_INTERLEAVE * DEMOD_XCD_INTERLEAVE;
#endif
    const dim3 threads(64 * kWavesPerBlock);
    const bool span = a.rpc_log2 < a.seg_log2;  // chunks shorter than a segment
    if (a.out_s16) {
        if (a.ragged_bytes)
            hipLaunchKernelGGL((demod_rows<TIN, TOUT, COND, true, true>), dim3(blocks), threads, 0, s, a);
        else if (span)
            hipLaunchKernelGGL((demod_rows<TIN, TOUT, COND, false, true, true>), dim3(blocks), threads, 0, s, a);
        else
            hipLaunchKernelGGL((demod_rows<TIN, TOUT, COND, false, true>), dim3(blocks), threads, 0, s, a);
    } else {
        if (a.ragged_bytes)
            hipLaunchKernelGGL((demod_rows<TIN, TOUT, COND, true, false>), dim3(blocks), threads, 0, s, a);
        else if (span)
            hipLaunchKernelGGL((demod_rows<TIN, TOUT, COND, false, false, true>), dim3(blocks), threads, 0, s, a);
        else
            hipLaunchKernelGGL((demod_rows<TIN, TOUT, COND, false, false>), dim3(blocks), threads, 0, s, a);
    }
    return hip_ok(hipGetLastError());
}

template <int COND>
int launch_rows_c(const RowLaunch &a, int tin, int tout, hipStream_t s) {
#ifdef DEMOD_LAB
    // laboratory build (tools/isa_table.py --lab, A/B variants): only the BASELINE configurations' kernels
    if (tin == 2 && tout == 2) return launch_rows_t<2, 2, COND>(a, s);
    if constexpr (COND != kShiftOriginNorm)
        if (tin == 0 && tout == 2) return launch_rows_t<0, 2, COND>(a, s);
    return -1;
#else
    // specialised stencil lengths for the common designs, a 5-tap build for the rest
    if (tin <= 3 && tout <= 3) {
        switch (tin * 4 + tout) {
        case 1 * 4 + 1: return launch_rows_t<1, 1, COND>(a, s);
        case 1 * 4 + 2: return launch_rows_t<1, 2, COND>(a, s);
        case 1 * 4 + 3: return launch_rows_t<1, 3, COND>(a, s);
        case 2 * 4 + 1: return launch_rows_t<2, 1, COND>(a, s);
        case 2 * 4 + 2: return launch_rows_t<2, 2, COND>(a, s);
        case 2 * 4 + 3: return launch_rows_t<2, 3, COND>(a, s);
        case 3 * 4 + 1: return launch_rows_t<3, 1, COND>(a, s);
        case 3 * 4 + 2: return launch_rows_t<3, 2, COND>(a, s);
        case 3 * 4 + 3: return launch_rows_t<3, 3, COND>(a, s);
        default: break;
        }
        // (the stencil divided by its first tap exists with a complex stage only)
        if constexpr (COND != kShiftOriginNorm) {
            switch (tin * 4 + tout) {
            case 0 * 4 + 1: return launch_rows_t<0, 1, COND>(a, s);
            case 0 * 4 + 2: return launch_rows_t<0, 2, COND>(a, s);
            case 0 * 4 + 3: return launch_rows_t<0, 3, COND>(a, s);
            default: break;
            }
        }
    }
    // 4 taps either side: what -L gives degrees 10-13 (without -D the input filter takes the output filter's degree, src/matrix.c:190-192)
    if (tin == 4 && tout == 4) return launch_rows_t<4, 4, COND>(a, s);
    // zero-padded taps make the long stencils exact for any shorter design
    if constexpr (COND != kShiftOriginNorm) {
        if (tin == 0) return tout == 4 ? launch_rows_t<0, 4, COND>(a, s) : launch_rows_t<0, kMaxTaps, COND>(a, s);
    }
    return launch_rows_t<kMaxTaps, kMaxTaps, COND>(a, s);
#endif
}

}  // namespace

uint32_t max_segment_rows() { return (uint32_t) kGroup; }

namespace {
// -q1: the bounds-checked build (its unpack reads the offsets table), with the stencil lengths of
// the default designs or the zero-padded 5-tap stencils
template <int TIN, int TOUT>
int launch_rows_q1(const RowLaunch &a, hipStream_t s) {
    const uint32_t g0 = a.phase_rows, g1 = a.phase_rows + a.n_rows;
    const uint32_t segs = ((g1 + (1u << a.seg_log2) - 1u) >> a.seg_log2) - (g0 >> a.seg_log2);
    uint32_t blocks = (segs + (uint32_t) kWavesPerBlock - 1u) / (uint32_t) kWavesPerBlock;
#if DEMOD_XCD_INTERLEAVE
    blocks = (blocks + DEMOD_XCD_INTERLEAVE - 1u) / DEMOD_XCD_INTERLEAVE * DEMOD_XCD_INTERLEAVE;
#endif
    const dim3 threads(64 * kWavesPerBlock);
    if (a.out_s16) hipLaunchKernelGGL((demod_rows<TIN, TOUT, kCorrectIq, true, true>), dim3(blocks), threads, 0, s, a);
    else hipLaunchKernelGGL((demod_rows<TIN, TOUT, kCorrectIq, true, false>), dim3(blocks), threads, 0, s, a);
    return hip_ok(hipGetLastError());
}
}  // namespace

int launch_rows(const RowLaunch &a, int taps_in, int taps_out, int conditioning, void *stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (a.n_rows == 0) return 0;
    if (conditioning == kShiftOriginNorm && taps_in > 0) return launch_rows_c<kShiftOriginNorm>(a, taps_in, taps_out, s);
    if (conditioning == kShiftOriginPlain) return launch_rows_c<kShiftOriginPlain>(a, taps_in, taps_out, s);
#ifdef DEMOD_LAB
    return conditioning == kShiftOrigin ? launch_rows_c<kShiftOrigin>(a, taps_in, taps_out, s) : -1;
#else
    if (conditioning == kCorrectIq) {
        if (taps_in == 0 && taps_out == 2) return launch_rows_q1<0, 2>(a, s);
        if (taps_in == 2 && taps_out == 2) return launch_rows_q1<2, 2>(a, s);
        return taps_in == 0 ? launch_rows_q1<0, kMaxTaps>(a, s) : launch_rows_q1<kMaxTaps, kMaxTaps>(a, s);
    }
    if (conditioning == kNormalize) return launch_rows_c<kNormalize>(a, taps_in, taps_out, s);
    return launch_rows_c<kShiftOrigin>(a, taps_in, taps_out, s);
#endif
}

namespace {
template <int COND>
void launch_heads_c(const HeadLaunch &a, hipStream_t s) {
    const uint32_t blocks = (a.n_heads + 63u) / 64u;
    const int want_h = 4 * ((std::max(2 * a.sec_out - 1, 4 * a.sec_in - 2) + 3) / 4);
    const bool std_window = (int) a.head_outputs == want_h && a.sec_pre == 0;
#define DEMOD_HEADS_CASE(SI, SO)                                                                   \
    if (std_window && a.sec_in == SI && a.sec_out == SO) {                                         \
        hipLaunchKernelGGL((demod_heads<COND, SI, SO>), dim3(blocks), dim3(64), 0, s, a);          \
        return;                                                                                    \
    }
    DEMOD_HEADS_CASE(0, 1)
    DEMOD_HEADS_CASE(0, 2)
    DEMOD_HEADS_CASE(1, 1)
    DEMOD_HEADS_CASE(1, 2)
    DEMOD_HEADS_CASE(2, 1)
    DEMOD_HEADS_CASE(2, 2)
#undef DEMOD_HEADS_CASE
    // any other design: one wavefront per head where chunk_head covers the case (see
    // demod_heads_wave), the thread-per-chunk replay with its cells in scratch for the rest --
    // -q1, a last chunk shorter than the lanes' window
    if constexpr (COND != kCorrectIq) {
        if (!a.thread_per_head && a.sec_pre <= 2 && a.head_outputs <= 64u && a.chunk_bytes >= kRowBytes) {
            const uint32_t by_wave = a.last_chunk_bytes >= kRowBytes ? a.n_heads : a.n_heads - 1u;
            if (by_wave) {
                HeadLaunch w = a;
                w.n_heads = by_wave;
                const dim3 grid((by_wave + kHeadWaves - 1u) / kHeadWaves), threads(64 * kHeadWaves);
                if (a.out_s16) hipLaunchKernelGGL((demod_heads_wave<COND, true>), grid, threads, 0, s, w);
                else hipLaunchKernelGGL((demod_heads_wave<COND, false>), grid, threads, 0, s, w);
            }
            if (by_wave == a.n_heads) return;
            HeadLaunch t = a;  // the short last chunk's head
            t.in = a.in + (uint64_t) by_wave * a.chunk_bytes;
            t.out = static_cast<char *>(a.out) + (uint64_t) by_wave * (a.chunk_bytes >> 2) * (a.out_s16 ? 2u : 4u);
            t.n_heads = 1;
            hipLaunchKernelGGL((demod_heads<COND, -1, -1>), dim3(1), dim3(64), 0, s, t);
            return;
        }
    }
    hipLaunchKernelGGL((demod_heads<COND, -1, -1>), dim3(blocks), dim3(64), 0, s, a);
}
}  // namespace

int launch_heads(const HeadLaunch &a, int conditioning, void *stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (a.n_heads == 0 || a.head_outputs == 0) return 0;
    if (conditioning == kNormalize)
        launch_heads_c<kNormalize>(a, s);
    else if (conditioning == kCorrectIq)
        launch_heads_c<kCorrectIq>(a, s);
    else
        launch_heads_c<kShiftOrigin>(a, s);
    return hip_ok(hipGetLastError());
}

int launch_tiny(const TinyLaunch &a, int conditioning, uint32_t threads, void *stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    const uint32_t blocks = (threads + 63u) / 64u;
    if (blocks == 0) return 0;
    if (conditioning == kNormalize)
        hipLaunchKernelGGL(demod_tiny<kNormalize>, dim3(blocks), dim3(64), 0, s, a);
    else if (conditioning == kCorrectIq)
        hipLaunchKernelGGL(demod_tiny<kCorrectIq>, dim3(blocks), dim3(64), 0, s, a);
    else
        hipLaunchKernelGGL(demod_tiny<kShiftOrigin>, dim3(blocks), dim3(64), 0, s, a);
    return hip_ok(hipGetLastError());
}

int launch_offsets(const OffsetLaunch &a, void *stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    const uint64_t chunks = a.n_chunks + (a.tail_bytes ? 1 : 0);
    const uint64_t total = a.parts > 1 ? chunks * a.parts : (chunks + a.group - 1) / a.group;
    if (total == 0) return 0;
    hipLaunchKernelGGL(correct_iq_offsets, dim3((uint32_t) ((total + 63) / 64)), dim3(64), 0, s, a);
    return hip_ok(hipGetLastError());
}

int launch_synth_fill(void *dst, uint64_t first_byte, uint64_t n_bytes, uint64_t seed, void *stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    const uint64_t pairs = n_bytes / 16;
    if (pairs == 0) return 0;
    uint64_t blocks = (pairs + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(synth_fill, dim3((uint32_t) blocks), dim3(256), 0, s, static_cast<uint64_t *>(dst),
                       first_byte / 8, pairs, seed);
    return hip_ok(hipGetLastError());
}

int launch_atan2_probe(const float *y, const float *x, float *out, uint32_t n, void *stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (n == 0) return 0;
    hipLaunchKernelGGL(atan2_probe, dim3((n + 255u) / 256u), dim3(256), 0, s, y, x, out, n);
    return hip_ok(hipGetLastError());
}

int launch_lane_probe(const float *in, float *out, float halo, void *stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(lane_probe, dim3(1), dim3(64), 0, s, in, out, halo);
    return hip_ok(hipGetLastError());
}

}  // namespace demod
