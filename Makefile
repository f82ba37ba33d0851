# Builds librenderbaby_hip.so (HIP kernels for gfx950 + C-ABI runtime) and the
# CPU oracle.  Explicit hipcc; no cmake, no JIT cache -- the .so stays in-tree so
# it travels to the GPU box.
HIPCC    ?= /opt/rocm/bin/hipcc
ARCH     ?= gfx950
CSRC     := renderbaby_amd/csrc
OUT      := renderbaby_amd/librenderbaby_hip.so
# Numerics contract: no FMA contraction, correctly rounded / and sqrt, no fast-math.
NUMERICS := -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math
# -fno-slp-vectorize: on gfx950 a packed f32 instruction (v_pk_mul_f32 ...) issues at half the rate of a plain one, so
# SLP-packing adjacent scalar operations gains no throughput and costs the v_mov's that build the register pairs:
# C2 26.4 -> 28.9 G segments/s, C1 21.8 -> 23.4, the mesh walks +4..9 % (profiles/r03_noslp.txt)
HIPFLAGS := -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) $(NUMERICS) -fno-slp-vectorize -Wall -Wno-unused-function
SRCS     := $(CSRC)/rb_kernels.hip $(CSRC)/rb_build.hip $(CSRC)/rb_runtime.cpp $(CSRC)/rb_bvh.cpp $(CSRC)/rb_rccl.cpp
HDRS     := $(CSRC)/rb_internal.hpp $(CSRC)/rb_device_common.hpp $(CSRC)/rb_device_math.hpp \
            $(CSRC)/rb_device_intersect.hpp $(CSRC)/rb_device_shade.hpp $(CSRC)/rb_rccl.hpp $(CSRC)/rb_chunk_math.hpp include/rb_abi.h
OBJS     := $(patsubst $(CSRC)/%,build/obj/%.o,$(SRCS))

all: $(OUT) oracle

# one object per source (the kernel TU takes a minute, the host files seconds).  RCCL is loaded with dlopen
# when a multi-device engine is first made (rb_rccl.cpp): no link-time dependency on librccl.
build/obj/%.o: $(CSRC)/% $(HDRS)
	@mkdir -p build/obj
	$(HIPCC) $(HIPFLAGS) -c -o $@ $<

$(OUT): $(OBJS)
	$(HIPCC) -fPIC --offload-arch=$(ARCH) -shared -o $@ $(OBJS) -ldl

oracle:
	$(MAKE) -C oracle

asm: $(CSRC)/rb_kernels.hip $(HDRS)
	mkdir -p build
	$(HIPCC) $(HIPFLAGS) -S --cuda-device-only -o build/rb_kernels.s $(CSRC)/rb_kernels.hip \
	    -Rpass-analysis=kernel-resource-usage 2> build/resource_usage.txt || true

clean:
	rm -f $(OUT) build/*.s build/*.txt
	$(MAKE) -C oracle clean

.PHONY: all oracle asm clean
