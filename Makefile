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

all: lib oracle
lib: $(PKG)/lib/libdba_hip.so
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

clean:
	rm -rf build $(PKG)/lib/libdba_hip.so
	$(MAKE) -C oracle clean
.PHONY: all lib oracle clean
