# Builds the product's shared libraries in-tree (they travel to the GPU box with the snapshot).
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
# -ffp-contract=off: the reference (Rust/LLVM) never fuses a*b+c; parity depends on it.
# (-Wall ... since round 5: the device code is warning-clean under them)
HIPFLAGS ?= --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wuninitialized -Wconditional-uninitialized -Wno-unused-value -Wno-unused-function -Wno-unused-command-line-argument $(EXTRA_HIPFLAGS)
CSRC := portrayer_amd/csrc
HIP_HDRS := $(wildcard $(CSRC)/*.h) include/portrayer_hip.h

CXX ?= g++
CXXFLAGS ?= -O2 -std=c++20 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wextra -Wno-unused-parameter -Wno-missing-field-initializers
HOST := portrayer_amd/host
HOST_SRCS := $(HOST)/portrayer.cpp $(HOST)/jpeg.cpp $(HOST)/capi.cpp $(wildcard examples/*.cpp)
HOST_HDRS := $(wildcard $(HOST)/*.hpp) examples/examples.hpp include/portrayer_host.h include/portrayer_hip.h
EXAMPLES := single-triangle primitives-simple macho-cows entering-the-mirror-dimension big-scene smooth-shading glossy-reflection soft-shadows hier instance antialiasing fish normal-mapping transmission-refraction water-glass \
            simple nonhier nonhier2 four-shapes graphics-poster simple-cows primitives texture-mapping cube-mapping graphics-castle graphics-temple monkeys-making-monkeys robot-alarm-clock

all: portrayer_amd/libportrayer_hip.so portrayer_amd/libportrayer_host.so $(addprefix examples/bin/,$(EXAMPLES))

# C++ host layer (scene API, flatten, k-d build, camera, Image) + the transliterated example scenes
portrayer_amd/libportrayer_host.so: $(HOST_SRCS) $(HOST_HDRS) portrayer_amd/libportrayer_hip.so
	$(CXX) $(CXXFLAGS) -shared $(HOST_SRCS) -o $@ -Lportrayer_amd -lportrayer_hip -lz -Wl,-rpath,'$$ORIGIN'

examples/bin/%: examples/%.cpp portrayer_amd/libportrayer_host.so
	@mkdir -p examples/bin
	$(CXX) $(CXXFLAGS) -fPIE -DPORTRAYER_EXAMPLE_MAIN $< -o $@ -Lportrayer_amd -lportrayer_host -lportrayer_hip -Wl,-rpath,'$$ORIGIN/../../portrayer_amd'

# Device objects: the C ABI + small kernels, the device-side tree build, and the render kernel instantiated per
# traversal mode (one source, six objects) - compiled side by side under make -j.
OBJDIR ?= $(CSRC)
HIPLIB ?= portrayer_amd/libportrayer_hip.so
RENDER_MODES := 1 2 3 4 5 6 7 8 9
HIP_OBJS := $(OBJDIR)/pt_api.o $(OBJDIR)/pt_build.o $(OBJDIR)/pt_node.o $(foreach m,$(RENDER_MODES),$(OBJDIR)/pt_render_m$(m).o)

# Every HIP translation unit is built in four steps instead of one `hipcc -c`, so that the device code can be CHECKED and REPAIRED between
# the compiler and the assembler (tools/check_exec_prologue.py; profiles/r05/notes.md section 1: the AMDGPU backend of this toolchain can put
# vector spill code in front of the instruction that re-converges a block's lanes - the wrong render of round 4, the hang of round 3):
#   1. device code to assembly            hipcc --offload-device-only -S
#   2. check + repair                     tools/check_exec_prologue.py --fix   (fails the build if a defect it cannot repair is left)
#   3. assemble, link, bundle             clang -x assembler, lld, clang-offload-bundler  - exactly what hipcc itself runs after its code generator
#   4. host code with the bundle embedded hipcc --cuda-host-only -Xclang -fcuda-include-gpubinary
LLVM_BIN ?= /opt/rocm/lib/llvm/bin
PYTHON ?= python3
define hip_four_steps
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) $(1) --offload-device-only -S $< -o $(basename $@).raw.s
	$(PYTHON) tools/check_exec_prologue.py --fix $(basename $@).raw.s -o $(basename $@).s
	$(LLVM_BIN)/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=$(ARCH) -c $(basename $@).s -o $(basename $@).dev.o
	$(LLVM_BIN)/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $(basename $@).co $(basename $@).dev.o
	$(LLVM_BIN)/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--$(ARCH) -input=/dev/null -input=$(basename $@).co -output=$(basename $@).hipfb
	$(HIPCC) $(HIPFLAGS) $(1) --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $(basename $@).hipfb -c $< -o $@
	@if [ -n "$(KEEP_ASM)" ]; then mkdir -p build/asm && cp $(basename $@).s build/asm/$(notdir $(basename $@)).s; fi
	@rm -f $(basename $@).raw.s $(basename $@).s $(basename $@).dev.o $(basename $@).co $(basename $@).hipfb
endef

$(OBJDIR)/%.o: $(CSRC)/%.hip $(HIP_HDRS) tools/check_exec_prologue.py
	$(call hip_four_steps,)

$(OBJDIR)/pt_render_m%.o: $(CSRC)/pt_render_inst.hip $(HIP_HDRS) tools/check_exec_prologue.py
	$(call hip_four_steps,-DPT_INST_MODE=$*)

$(HIPLIB): $(HIP_OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared $^ -o $@ -ldl

# the device objects alone (`make -B hipobjs` = the forced recompile of __graft_entry__.build())
hipobjs: $(HIP_OBJS)

# A/B builds made HERE (hipcc cross-compiles) and swapped in on the GPU box by profiles/*.sh:
#   make variant NAME=diag EXTRA_HIPFLAGS=-DPT_DIAG  ->  build/variants/diag/libportrayer_hip.so
variant:
	$(MAKE) OBJDIR=build/variants/$(NAME) HIPLIB=build/variants/$(NAME)/libportrayer_hip.so EXTRA_HIPFLAGS="$(EXTRA_HIPFLAGS)" build/variants/$(NAME)/libportrayer_hip.so

oracle:
	$(MAKE) -C oracle

# `make verify`: every HIP object rebuilt with its checked assembly kept (build/asm/*.s) and the checker run over all of it once more;
# `make verify-mi`: the nine render objects compiled under LLVM's machine verifier.
verify:
	$(MAKE) -B hipobjs KEEP_ASM=1
	$(PYTHON) tools/check_exec_prologue.py $(foreach o,$(HIP_OBJS),build/asm/$(notdir $(basename $(o))).s)
verify-mi:
	for m in $(RENDER_MODES); do $(HIPCC) $(HIPFLAGS) --offload-device-only -mllvm -verify-machineinstrs -DPT_INST_MODE=$$m -c $(CSRC)/pt_render_inst.hip -o /dev/null || exit 1; done

clean:
	rm -f portrayer_amd/*.so $(CSRC)/*.o
	rm -rf build/asm
	rm -rf build/variants
	rm -rf examples/bin
	$(MAKE) -C oracle clean
.PHONY: all oracle clean hipobjs variant verify verify-mi
