# Build the gfx950 kernels (libwan_hip.so, C ABI in include/wan_hip.h) and the dev harness.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
CSRC  := videocof_amd/csrc
SRCS  := $(wildcard $(CSRC)/*.cpp) $(wildcard $(CSRC)/*.hip)
HDRS  := $(CSRC)/common.hpp include/wan_hip.h $(wildcard $(CSRC)/*.inc)
LIB   := videocof_amd/libwan_hip.so
OBJD  := build/obj
OBJS  := $(patsubst $(CSRC)/%,$(OBJD)/%.o,$(SRCS))
# -fno-honor-nans: lets fmaxf/fminf lower to one v_max/v_min (no canonicalising v_max on MFMA outputs)
# EXPERIMENTS=1 compiles the timing-only variants behind the gemm_exp switch into the kernels (tools/kernel_check gemmx)
EXPERIMENTS ?= 0
FLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -fno-honor-nans -Iinclude -DWAN_DEV_EXPERIMENTS=$(EXPERIMENTS)
# attn_fwd.hip only: keep adjacent scalar f32 adds single-instruction -- under plain -O3 the SLP vectoriser packs the
# row-sum adds of the two query blocks into v_pk_add_f32, which beside MFMAs costs more than it saves (CDNA4 guide,
# per-instruction cycle constants).  The GEMM / norm epilogues keep their packed math.
FLAGS_attn_fwd.hip := -fno-slp-vectorize

all: $(LIB) tools/kernel_check

$(OBJD)/%.o: $(CSRC)/% $(HDRS)
	@mkdir -p $(OBJD)
	$(HIPCC) $(FLAGS) $(FLAGS_$*) -c $< -o $@

$(LIB): $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared $(OBJS) -o $@

tools/kernel_check: tools/kernel_check.cpp $(LIB) include/wan_hip.h
	$(HIPCC) -O2 -std=c++17 --offload-arch=$(ARCH) tools/kernel_check.cpp -Iinclude -Lvideocof_amd -lwan_hip -Wl,-rpath,'$$ORIGIN/../videocof_amd' -o $@

clean:
	rm -rf $(LIB) tools/kernel_check $(OBJD)

.PHONY: all clean
