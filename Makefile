# Builds the product's shared libraries in-tree (they travel to the GPU box with the snapshot).
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
# -ffp-contract=off: the reference (Rust/LLVM) never fuses a*b+c; parity depends on it.
HIPFLAGS ?= --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-value
CSRC := portrayer_amd/csrc
HIP_HDRS := $(wildcard $(CSRC)/*.h) include/portrayer_hip.h

all: portrayer_amd/libportrayer_hip.so

portrayer_amd/libportrayer_hip.so: $(CSRC)/pt_api.hip $(HIP_HDRS)
	$(HIPCC) $(HIPFLAGS) -shared $(CSRC)/pt_api.hip -o $@

oracle:
	$(MAKE) -C oracle

clean:
	rm -f portrayer_amd/*.so
	$(MAKE) -C oracle clean
.PHONY: all oracle clean
