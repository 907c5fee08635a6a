// Host-only fuzz of the filter design / plan builder (no GPU): 200,000 random option sets, meant for
// g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -Iinclude -Idemodulator_amd/csrc \
//     tools/fuzz_plan.cpp demodulator_amd/csrc/plan.cpp -o /tmp/fuzz_plan && /tmp/fuzz_plan
#include <cstdio>
#include <random>
#include <string>
#include "plan.h"
using namespace demod;
int main() {
    std::mt19937_64 rng(7);
    int ok = 0, bad = 0;
    for (int it = 0; it < 200000; ++it) {
        Options o;
        o.sample_rate = std::pow(10.0, 1.0 + (rng() % 8000) / 1000.0);
        o.lowpass_in = (rng() & 1) ? std::pow(10.0, (rng() % 7000) / 1000.0) : 0.0;
        o.lowpass_out = std::pow(10.0, (rng() % 7000) / 1000.0);
        o.epsilon = std::pow(10.0, -4.0 + (rng() % 6000) / 1000.0);
        o.in_degree = rng() % 19;
        o.out_degree = rng() % 19;
        o.mode = (unsigned) (rng() % 64);
        o.buf_size = (rng() % 8 == 0) ? (size_t) (rng() % 100000) : (size_t) 1 << (rng() % 24);
        Plan p;
        std::string err;
        if (build_plan(o, p, err)) ++ok; else ++bad;
    }
    printf("accepted %d refused %d\n", ok, bad);
    return 0;
}
