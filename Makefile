# Build the gfx950 C-ABI library (the product) and the C oracle (test infrastructure).
#   make lib      -> spatten_amd/lib/libspatten_hip.so
#   make oracle   -> oracle/liboracle.so
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
HIPFLAGS ?= -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -Wall -Wno-unused-function
CSRC := $(wildcard spatten_amd/csrc/*.hip)
OBJS := $(patsubst spatten_amd/csrc/%.hip,build/%.o,$(CSRC))
LIB := spatten_amd/lib/libspatten_hip.so

all: lib oracle

lib: $(LIB)

build/%.o: spatten_amd/csrc/%.hip spatten_amd/csrc/common.h include/spatten.h $(wildcard spatten_amd/csrc/*.h)
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) $(FLAGS_$*) -c $< -o $@

# the flash kernels: no SLP vectorisation — packed fp32 VALU (v_pk_mul/add/fma_f32) issued beside MFMAs costs ~+22
# cycles per instruction on gfx950 (measured: 650 -> 740 TFLOP/s with the packing gone)
FLAGS_prefill_attn := -fno-slp-vectorize
# the decode kernels: the first 16 kernel-argument dwords are preloaded into SGPRs at wave launch (decode_lean_kernel
# puts what the first K/V loads need there)
FLAGS_decode_attn := -mllvm -amdgpu-kernarg-preload-count=16

$(LIB): $(OBJS)
	@mkdir -p spatten_amd/lib
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS) -ldl

oracle: oracle/liboracle.so

oracle/liboracle.so: oracle/oracle.c
	gcc -O3 -march=x86-64-v3 -fopenmp -fPIC -shared -o $@ $< -lm

clean:
	rm -rf build $(LIB) oracle/liboracle.so

.PHONY: all lib oracle clean
