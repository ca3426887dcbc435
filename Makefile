# Builds libasr_amd.so (gfx950 HIP kernels + C ABI) in-tree.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
SRC := $(wildcard asr_amd/csrc/*.hip)
OBJ := $(patsubst asr_amd/csrc/%.hip,build/%.o,$(SRC))
HDR := $(wildcard asr_amd/csrc/*.h) include/asr_amd.h
LIB := asr_amd/lib/libasr_amd.so
CXXFLAGS := -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -Wall -Wno-unused-function -Wno-pass-failed -Wno-bitwise-instead-of-logical

all: $(LIB)

build/%.o: asr_amd/csrc/%.hip $(HDR)
	@mkdir -p build
	$(HIPCC) $(CXXFLAGS) -c $< -o $@

$(LIB): $(OBJ)
	@mkdir -p asr_amd/lib
	$(HIPCC) -shared -fPIC --offload-arch=$(ARCH) $(OBJ) -o $@

clean:
	rm -rf build $(LIB)

.PHONY: all clean
