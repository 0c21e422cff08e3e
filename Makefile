# Build the gfx950 kernels (libwan_hip.so, C ABI in include/wan_hip.h) and the dev harness.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
CSRC  := videocof_amd/csrc
SRCS  := $(CSRC)/api.cpp $(wildcard $(CSRC)/*.hip)
HDRS  := $(CSRC)/common.hpp include/wan_hip.h
LIB   := videocof_amd/libwan_hip.so
# -fno-honor-nans: lets fmaxf/fminf lower to one v_max/v_min (no canonicalising v_max on MFMA outputs)
FLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -fno-honor-nans -Iinclude

all: $(LIB) tools/kernel_check

$(LIB): $(SRCS) $(HDRS)
	$(HIPCC) $(FLAGS) -shared $(SRCS) -o $@

tools/kernel_check: tools/kernel_check.cpp $(LIB) include/wan_hip.h
	$(HIPCC) -O2 -std=c++17 --offload-arch=$(ARCH) tools/kernel_check.cpp -Iinclude -Lvideocof_amd -lwan_hip -Wl,-rpath,'$$ORIGIN/../videocof_amd' -o $@

clean:
	rm -f $(LIB) tools/kernel_check

.PHONY: all clean
