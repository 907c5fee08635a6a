// ctx.h -- the context behind the C ABI (private to the library: api.cpp, host.cpp).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdio>
#include <string>
#include <utility>
#include <vector>

#include "demod_hip.h"
#include "kernels.h"
#include "plan.h"

struct HostPipe;  // host.cpp: streams, staging stages and copy workers of the host-buffer paths

struct demod_ctx {
    demod::Options opt;
    demod::Plan plan;
    int device = 0;
    std::string err;

    bool seamless = false;  // filter the whole stream as one chunk (look-ahead continues across chunk boundaries)
    // demod_set_stencil_form: which form of the complex stencil the streaming kernel runs behind the shiftOrigin
    // unpack -- DEMOD_STENCIL_AUTO: divided by its first tap where build_plan allows it (Plan::norm_ok), plain otherwise
    int stencil_form = DEMOD_STENCIL_AUTO;
    int head_mode = -1;     // demod_set_head_mode: -1 automatic, 0 demod_heads kernel, 1 inside the streaming kernel
    uint32_t seg_rows = 0;  // 0 = automatic
    uint32_t piece_rows = 0;           // rows per demod_rows dispatch (set at creation)
    bool out_s16 = false;              // results leave as int16 (demod_set_output_format)
    float out_scale = 1.f;
    uint32_t out_bytes() const { return out_s16 ? 2u : 4u; }
    int last_row_launches = 0;  // demod_rows dispatches issued by the most recent process call

    // kernel timing (HIP events on the launch stream)
    bool timing = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> timed;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> event_pool;

    // scratch for the whole-chunk literal path (chunks < 1 KiB): ONE buffer, wavefront w's
    // threads always use work + w * 64 * stride (interleaved), so launches on different streams (the host path's
    // stages) are ordered through tiny_done: each waits for the one before it
    float *tiny_work = nullptr;
    uint64_t tiny_stride = 0;
    uint32_t tiny_threads = 0;   // threads the scratch has room for
    hipEvent_t tiny_done = nullptr;
    bool tiny_pending = false;

    // -q1: the offsets table the correct_iq_offsets pass fills for one call (two floats per
    // iteration), and the two offset pairs that carry the reference's static state from call
    // to call (read from one, written to the other, swapped every call)
    float *q1_table = nullptr;
    uint64_t q1_pairs = 0;
    float *q1_carry = nullptr;
    int q1_flip = 0;

    HostPipe *pipe = nullptr;  // created by the first host-path call
    int rings = 0;             // streaming rings alive on this context (demod_ring_create)

    int err_code = DEMOD_OK;  // code of the failure `err` describes (demod_last_error_code)
    bool fail(int code, const char *what, hipError_t e = hipSuccess) {
        char buf[384];
        if (e != hipSuccess)
            snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
        else
            snprintf(buf, sizeof buf, "%s", what);
        err = buf;
        err_code = code;
        return false;
    }
};

#define HIP_TRY(ctx, call)                               \
    do {                                                 \
        hipError_t e_ = (call);                          \
        if (e_ != hipSuccess) {                          \
            (ctx)->fail(DEMOD_E_HIP, #call, e_);         \
            return DEMOD_E_HIP;                          \
        }                                                \
    } while (0)

// api.cpp: records the message demod_last_error(NULL) returns and hands back `code`
extern "C" int demod_create_fail(int code, const char *msg);

// host.cpp
void host_pipe_destroy(demod_ctx *ctx);
bool host_pipe_busy(const demod_ctx *ctx);   // submitted pieces not yet collected
