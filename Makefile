# Builds the product's shared libraries in-tree (they travel to the GPU box with the snapshot).
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
# -ffp-contract=off: the reference (Rust/LLVM) never fuses a*b+c; parity depends on it.
HIPFLAGS ?= --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-value $(EXTRA_HIPFLAGS)
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

$(OBJDIR)/%.o: $(CSRC)/%.hip $(HIP_HDRS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(OBJDIR)/pt_render_m%.o: $(CSRC)/pt_render_inst.hip $(HIP_HDRS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -DPT_INST_MODE=$* -c $< -o $@

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

clean:
	rm -f portrayer_amd/*.so $(CSRC)/*.o
	rm -rf build/variants
	rm -rf examples/bin
	$(MAKE) -C oracle clean
.PHONY: all oracle clean hipobjs variant
