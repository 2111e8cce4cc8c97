# Builds libpmc.so (HIP kernels + C ABI) for gfx950, in-tree.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
CSRC := parelagmc_amd/csrc
OBJDIR := build/obj
SRCS := $(CSRC)/kernels.hip $(CSRC)/sparse.hip $(CSRC)/solver.hip $(CSRC)/sampler.hip $(CSRC)/darcy.hip $(CSRC)/capi.hip $(CSRC)/hybrid_build.hip
OBJS := $(patsubst $(CSRC)/%.hip,$(OBJDIR)/%.o,$(SRCS))
HDRS := $(wildcard $(CSRC)/*.hpp) include/pmc.h
EXTRA ?=
CXXFLAGS := -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -Wall -Wno-unused-function $(EXTRA)
LIB := parelagmc_amd/lib/libpmc.so

all: $(LIB)

$(OBJDIR)/%.o: $(CSRC)/%.hip $(HDRS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(CXXFLAGS) -c $< -o $@

$(LIB): $(OBJS)
	@mkdir -p parelagmc_amd/lib
	$(HIPCC) -shared -fPIC --offload-arch=$(ARCH) -o $@ $(OBJS) -ldl

clean:
	rm -rf build parelagmc_amd/lib/libpmc.so

.PHONY: all clean

# host-side managers (plain C++, links against libpmc.so next to it)
HOSTLIB := parelagmc_amd/lib/libpmc_host.so
HOSTSRC := parelagmc_amd/host/host.cpp parelagmc_amd/host/mortar.cpp
$(HOSTLIB): $(HOSTSRC) parelagmc_amd/host/parelagmc.hpp include/pmc.h include/pmc_host.h $(LIB)
	g++ -O2 -std=c++17 -fPIC -shared -Wall -pthread -Iinclude -o $@ $(HOSTSRC) -Lparelagmc_amd/lib -lpmc -Wl,-rpath,'$$ORIGIN'

all: $(HOSTLIB)

# kernel laboratory (tuning only): links the library's objects with a main(); kept next to the library so that it travels
# to the GPU box (build/ does not)
LABBIN := parelagmc_amd/lib/k5_lab
$(OBJDIR)/k5_lab.o: scripts/lab/k5_lab.hip $(HDRS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(CXXFLAGS) -c $< -o $@
$(LABBIN): $(OBJDIR)/k5_lab.o $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -o $@ $(OBJDIR)/k5_lab.o $(OBJS) -ldl
lab: $(LABBIN)
.PHONY: lab

# laboratory build of the library: the same sources with -DPMC_LAB, i.e. with the PMC_* tuning overrides of csrc/common.hpp
# (lab_env) compiled in.  The product library (libpmc.so) reads none of them.  scripts/lab/*.sh copy it over libpmc.so on the
# GPU box's scratch tree for same-box A/B runs.
lab-lib:
	$(MAKE) OBJDIR=build/obj_lab LIB=parelagmc_amd/lib/libpmc_lab.so EXTRA="-DPMC_LAB $(LABEXTRA)" parelagmc_amd/lib/libpmc_lab.so
.PHONY: lab-lib

# Boundary tests compiled from C and C++ (tests/test_abi_binaries.py runs them): the C program sees include/pmc.h only,
# the C++ one the MFEM adapter (against tests/c/mfem_shim.hpp) and the mirror classes of parelagmc.hpp
ABIBIN := tests/c/bin
# (the libraries are order-only prerequisites: on a box that received the built .so files but not build/obj, nothing is rebuilt)
$(ABIBIN)/abi_smoke: tests/c/abi_smoke.c tests/c/prob_io.h include/pmc.h | $(LIB)
	@mkdir -p $(ABIBIN)
	gcc -std=c11 -O1 -Wall -Wextra -Werror -Iinclude -Itests/c -o $@ tests/c/abi_smoke.c -Lparelagmc_amd/lib -lpmc -lm -Wl,-rpath,'$$ORIGIN/../../../parelagmc_amd/lib'
$(ABIBIN)/adapter_smoke: tests/c/adapter_smoke.cpp tests/c/prob_io.h tests/c/mfem_shim.hpp parelagmc_amd/host/mfem_adapter.hpp parelagmc_amd/host/parelagmc.hpp include/pmc.h include/pmc_host.h | $(HOSTLIB)
	@mkdir -p $(ABIBIN)
	g++ -std=c++17 -O1 -Wall -Wextra -Iinclude -Itests/c -o $@ tests/c/adapter_smoke.cpp -Lparelagmc_amd/lib -lpmc_host -lpmc -pthread -Wl,-rpath,'$$ORIGIN/../../../parelagmc_amd/lib'
test-abi: $(ABIBIN)/abi_smoke $(ABIBIN)/adapter_smoke
.PHONY: test-abi
