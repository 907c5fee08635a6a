// plan.h -- host-side setup of the demodulation path: filter design, stencil
// taps, chunk-head extent.  Runs once per context; nothing here is on the
// per-sample path.
#pragma once

#include <cstddef>
#include <cstdint>
#include <string>

#include "demod_hip.h"

namespace demod {

constexpr int kMaxSections = (DEMOD_MAX_DEGREE + 1) / 2;  // 8
constexpr int kMaxTaps = 5;                               // look-ahead stencil length for degree <= 16

// Options with the reference's own precision (consumerArgs holds long double,
// include/matrix.h:50-53); demod_params (double) widens into this.
struct Options {
    long double sample_rate = 125000.L;
    long double lowpass_in = 0.L;
    long double lowpass_out = 12500.L;
    long double epsilon = .3;
    size_t in_degree = 0;
    size_t out_degree = 3;
    unsigned mode = 0x10;
    size_t buf_size = DEMOD_DEFAULT_BUF_SIZE;
};

Options options_from_params(const demod_params &p);

// kShiftOriginPlain: the shiftOrigin unpack without folding its -128 into the complex
// stencil's constant (kernels.hip, fold_unpack): the -q2 highpass taps (+3, -1) cancel for
// samples with x[k+1] = 3 x[k], and there the result has to be as precise as the samples are
// small, not as 128 + sample is.  The default for -q0 with a complex stage as well (build_plan:
// the folded form leaves the parity criterion at samples that filter to next to nothing).
// kCorrectIq (-q1): byte minus a running DC offset that the reference carries from sample pair to
// sample pair and from chunk to chunk (src/matrix.c:120-140); the offsets are replayed by a
// pass of their own into a table the other kernels read (kernels.hip, correct_iq_offsets).
// kShiftOriginNorm (streaming kernel only, round 3): the complex stencil divided by its first tap,
//   z / h0 = x[k] + rho_1 x[k+1] (+ rho_2 x[k+2] ...) + bias / h0,     rho_j = h_j / h_0,
// which is all the discriminator needs -- it takes the ARGUMENT of z[2n] conj(z[2n+1]), and a positive
// factor on both samples does not turn it.  The first tap costs no operation, and the constant
// rides on the unpack: X = alpha * (sample + 128) - (128 alpha, alpha (128 - s)) = alpha (x, x + s),
// s = bias / sum(h), one packed fused multiply-add where the plain form has a packed subtraction --
// exact, alpha chosen by build_plan so that its constants are floats (plan.cpp).  One fused
// multiply-add per tap less, ONE rounding per filtered sample instead of TIN, on intermediates no
// larger than the samples.  What it is not is the reference's own coefficient set: the rho_j are
// floats, so the taps applied are the design's up to their rounding -- build_plan prices that
// (Plan::norm_error) against the rounding the plain form commits and allows the form where it is smaller.
enum Conditioning : int { kShiftOrigin = 0, kShiftOriginPlain = 1, kShiftOriginNorm = 2, kNormalize = 3, kCorrectIq = 4 };

struct Stage {
    int sections = 0;                 // S = ceil(degree / 2)
    float sos[kMaxSections][6] = {};  // {b0,b1,b2,a0,a1,a2} per section, as the reference stores them
    int taps = 0;                     // T
    float h[kMaxTaps] = {};           // look-ahead FIR the sectioned recurrence collapses to
    float bias_im = 0.f;              // constant the imaginary lane of the complex filter carries
};

struct Plan {
    uint64_t chunk_bytes = 0;  // B
    int conditioning = kShiftOrigin;
    // -q0 with a complex stage: the folded form of the unpack (kShiftOrigin in the streaming kernel) may stand in for
    // the plain one, its constant -128 * sum(h) being exactly a float (build_plan); opt-in, demod_set_unpack_fold
    bool fold_exact = false;
    // -q0 with a complex stage: the stencil divided by its first tap (kShiftOriginNorm).  rho[j] = (float)(h[j] / h[0]);
    // the unpack is X = alpha * (byte + 128-complement) - (k_re, k_im) = alpha (x, x + s), exact, with k_re = 128 alpha and
    // alpha searched so that k_im = alpha (128 - s) is a float to within norm_bias_error (in the reference's units of a
    // filtered sample); norm_error / plain_error: what the remaining roundings of the two forms can move a full-scale
    // filtered sample by (plan.cpp, price_normalised_stencil); norm_ok: the form is allowed
    float rho[kMaxTaps] = {};
    float norm_alpha = 1.f, norm_k_re = 128.f, norm_k_im = 128.f;
    double norm_error = 0, plain_error = 0, norm_bias_error = 0;
    bool norm_ok = false;
    bool prefilter = false;
    // -q2: the complex stage is highpassDc (src/matrix.c:142-157), which reads a
    // buffer of its own: zeros behind the chunk's last sample where the -L filter
    // reads the head of its own output (the two buffers are adjacent there).
    bool in_separate = false;
    Stage in;    // complex input filter (-L, or the -q2 highpass), sections == 0 when absent
    // -q2 together with -L: highpassDc first (`pre`, a buffer of its own), then the -L filter
    // (`in`, reading what `pre` wrote).  The literal kernels replay both; the streaming kernel
    // applies ONE stencil, their composition: taps = pre.h * in.h (convolution),
    // bias = sum(in.h) * pre.bias_im + in.bias_im.  `stencil_in` is what the streaming kernel
    // uses in every case (a copy of `in` without `pre`).
    Stage pre;
    Stage stencil_in;
    Stage out;   // real output filter (-l)
    int head_outputs = 0;  // H: outputs per chunk that the closed form does not cover
    int head_quads = 0;    // ceil(H / 4): lanes of a chunk's first row whose outputs are chunk-head outputs (replayed literally)
    double magnitude_bound = 0;  // plan_magnitude_bound(): every float of the path stays below this
    // -q1: offsets move by (sum of the two corrected samples of an iteration) * esr, esr = (float)(50 / S)
    // (src/matrix.c:137-138,210); q1_warm: the first run-in correct_iq_offsets gives its bracket of the state to
    // close (about 26 time constants of 1 / (2 esr) iterations; doubled while it stays open, build_plan)
    float esr = 0.f;
    uint32_t q1_warm = 0, q1_retry = 0;
};

// Bound on the magnitude of every intermediate and result of the path for a plan (finite
// arithmetic is proven from the design; build_plan refuses plans beyond 1e30).
double plan_magnitude_bound(const Plan &pl);

// Pre-warped bilinear Butterworth / Chebyshev-I design -> float SOS rows
// (what processFilterOption leaves in sosf, /root/reference/src/matrix.c:25-80).
// kind: 0 Butterworth lowpass, 1 Chebyshev-I lowpass, 2 Butterworth highpass,
// 3 Chebyshev-I highpass.
void design_sections(unsigned kind, size_t degree, long double fc, long double fs,
                     long double epsilon, float (*sos)[6]);

// Collapse the sectioned in-place recurrence of applyFilter / applyComplexFilter
// (src/filter.c:212-259) into the forward-looking FIR it is equivalent to in a
// chunk interior.
void derive_taps(Stage &st);

// Validate the option set and build the plan.  Returns false with a message.
bool build_plan(const Options &o, Plan &plan, std::string &err);

}  // namespace demod
