# Builds the gfx950 HIP library (the product) and the CPU oracle (test infrastructure).
#   make            -> dba-fusion_amd/lib/libdba_hip.so + oracle/liboracle.so
#   make lib        -> HIP library only
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
PKG := dba-fusion_amd
CSRC := $(PKG)/csrc
BUILD := build/$(ARCH)
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wall -Wno-unused-function -Iinclude
SRCS := $(wildcard $(CSRC)/*.hip)
OBJS := $(patsubst $(CSRC)/%.hip,$(BUILD)/%.o,$(SRCS))
HDRS := $(wildcard $(CSRC)/*.h) include/dba_hip.h

all: lib ext oracle testbin
lib: $(PKG)/lib/libdba_hip.so
# test infrastructure: one cold start of the solvers per process (tests/test_gpu_solve_cold.py); host C++, loads the library by dlopen
testbin: tests/native/solve_cold
tests/native/solve_cold: tests/native/solve_cold.cpp
	g++ -O2 -std=c++17 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include $< -o $@ -L/opt/rocm/lib -lamdhip64 -ldl -Wl,-rpath,/opt/rocm/lib
ext: $(PKG)/droid_backends/_droid_backends_C.so
oracle:
	$(MAKE) -C oracle -s

# kernels whose results must be bit-identical to the reference arithmetic: no mul+add fusion
# (HIP's default -ffp-contract=fast fuses in the backend regardless of source pragmas)
EXACT := corr_lookup corr_sheared altcorr
$(foreach f,$(EXACT),$(eval $(BUILD)/$(f).o: EXTRA := -ffp-contract=off))

$(BUILD)/%.o: $(CSRC)/%.hip $(HDRS)
	@mkdir -p $(BUILD)
	$(HIPCC) $(HIPFLAGS) $(EXTRA) -c $< -o $@

$(PKG)/lib/libdba_hip.so: $(OBJS)
	@mkdir -p $(PKG)/lib
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS)

# the compiled `droid_backends` adapter (pybind11 over the C ABI; host C++ only: the torch headers, no device code).  The
# torch variables are recursively expanded: python only imports torch when the ext target is actually rebuilt
PY ?= python3
TORCH_INC = $(shell $(PY) -c "from torch.utils import cpp_extension as c; print(' '.join('-isystem ' + p for p in c.include_paths()))")
TORCH_LIB = $(shell $(PY) -c "import os, torch; print(os.path.join(os.path.dirname(torch.__file__), 'lib'))")
PY_INC = $(shell $(PY) -c "import sysconfig; print(sysconfig.get_paths()['include'])")
TORCH_ABI = $(shell $(PY) -c "import torch; print(int(torch._C._GLIBCXX_USE_CXX11_ABI))")
$(PKG)/droid_backends/_droid_backends_C.so: $(PKG)/csrc_ext/droid_backends_ext.cpp include/dba_hip.h $(PKG)/lib/libdba_hip.so
	g++ -O2 -std=c++17 -fPIC -shared -D__HIP_PLATFORM_AMD__ -DUSE_ROCM -D_GLIBCXX_USE_CXX11_ABI=$(TORCH_ABI) -DTORCH_EXTENSION_NAME=_droid_backends_C \
	    -Iinclude $(TORCH_INC) -isystem $(PY_INC) -isystem /opt/rocm/include $< -o $@ \
	    -L$(PKG)/lib -ldba_hip -L$(TORCH_LIB) -ltorch -ltorch_cpu -ltorch_python -lc10 -lc10_hip -L/opt/rocm/lib -lamdhip64 \
	    -Wl,-rpath,'$$ORIGIN/../lib' -Wl,-rpath,$(TORCH_LIB)

clean:
	rm -rf build $(PKG)/lib/libdba_hip.so $(PKG)/droid_backends/_droid_backends_C.so tests/native/solve_cold
	$(MAKE) -C oracle clean
.PHONY: all lib ext oracle testbin clean
