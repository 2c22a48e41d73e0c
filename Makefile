# Builds libchase_hip.so (HIP kernels + C ABI + C++ host solver) for gfx950, in-tree.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
CSRC  := chase_amd/csrc
HOST  := chase_amd/host
OUT   := chase_amd/lib/libchase_hip.so
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Iinclude -I/opt/rocm/include -I$(CSRC) -I$(HOST) -Wno-unused-result
SRCS  := $(wildcard $(CSRC)/*.hip) $(wildcard $(CSRC)/*.cpp) $(wildcard $(HOST)/*.cpp)
OBJS  := $(patsubst %,build/%.o,$(SRCS))
HDRS  := $(wildcard include/*.h) $(wildcard $(CSRC)/*.h) $(wildcard $(HOST)/*.hpp)

all: $(OUT)

build/%.o: % $(HDRS)
	@mkdir -p $(dir $@)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(OUT): $(OBJS)
	@mkdir -p $(dir $@)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS) -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib -ldl -lpthread

clean:
	rm -rf build $(OUT)

.PHONY: all clean
