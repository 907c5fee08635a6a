# Builds the MI355X-native demodulation library, the CLI host and the test oracle.
#
#   make lib      demodulator_amd/libdemod_hip.so      (HIP kernels + C ABI, gfx950)
#   make backend  demodulator_amd/libdemod_backend.so  (processMatrix/allocateBuffer adapter)
#   make cli      demodulator_amd/demod                (C host, reference CLI contract)
#   make oracle   oracle/libdemod_oracle.so (+ oracle/_ref when /root/reference exists)
#   make microbench  tools/microbench/{valu_rate,stream_copy}
#
# Everything is built in-tree so the binaries travel to the GPU box with the
# source snapshot.

HIPCC   ?= /opt/rocm/bin/hipcc
CC      ?= gcc
ARCH    ?= gfx950
PKG      = demodulator_amd
SRC      = $(PKG)/csrc
# -ffp-contract=off: the kernels spell every fused multiply-add explicitly
# -fno-slp-vectorize: the vectoriser pairs values that are not adjacent and pays in moves (-10 % in round 1);
#   the pairs that are (re/im of a sample, two neighbouring outputs) are written as such in kernels.hip (pf2)
# max-ilp scheduling: interleaves the four independent atan2 chains of a row
# (tools/microbench/valu_rate.hip, profiles/)
HIPFLAGS = --offload-arch=$(ARCH) -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize \
           -mllvm -amdgpu-sched-strategy=max-ilp -fPIC -Wall -Wno-pass-failed \
           -Wno-unused-function -Iinclude -I$(SRC)
CFLAGS   = -O2 -std=gnu99 -Wall -Iinclude

all: lib backend cli oracle
# (oracle is built after backend so that oracle/_ref/demodulator_refhost can link it)

lib: $(PKG)/libdemod_hip.so
$(PKG)/libdemod_hip.so: $(SRC)/kernels.hip $(SRC)/api.cpp $(SRC)/host.cpp $(SRC)/multi.cpp $(SRC)/plan.cpp $(SRC)/kernels.h \
                        $(SRC)/plan.h $(SRC)/ctx.h include/demod_hip.h
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(SRC)/kernels.hip $(SRC)/api.cpp $(SRC)/host.cpp $(SRC)/multi.cpp $(SRC)/plan.cpp -lpthread -ldl

backend: $(PKG)/libdemod_backend.so
$(PKG)/libdemod_backend.so: $(SRC)/backend.c include/demod_backend.h include/demod_hip.h $(PKG)/libdemod_hip.so
	$(CC) $(CFLAGS) -fPIC -shared -o $@ $(SRC)/backend.c -L$(PKG) -ldemod_hip \
	    -Wl,-rpath,'$$ORIGIN' -lpthread

cli: $(PKG)/demod
$(PKG)/demod: $(SRC)/main.c include/demod_backend.h include/demod_hip.h $(PKG)/libdemod_backend.so
	$(CC) $(CFLAGS) -o $@ $(SRC)/main.c -L$(PKG) -ldemod_backend -ldemod_hip \
	    -Wl,-rpath,'$$ORIGIN' -lpthread -lm

oracle:
	$(MAKE) -C oracle

# instruction-rate and streaming-pattern microbenchmarks behind DESIGN.md section 4.1 (run on the GPU box)
microbench: tools/microbench/valu_rate tools/microbench/stream_copy tools/microbench/chain_latency tools/microbench/policy_sweep
tools/microbench/%: tools/microbench/%.hip
	$(HIPCC) --offload-arch=$(ARCH) -O3 -o $@ $<

clean:
	rm -f $(PKG)/libdemod_hip.so $(PKG)/libdemod_backend.so $(PKG)/demod
	$(MAKE) -C oracle clean

.PHONY: all lib backend cli oracle microbench clean
