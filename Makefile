# Builds libchase_hip.so (HIP kernels + C ABI + C++ host solver) for gfx950, in-tree.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
CSRC  := chase_amd/csrc
HOST  := chase_amd/host
OUT   := chase_amd/lib/libchase_hip.so
EXTRA ?=
HIPFLAGS := $(EXTRA) --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Iinclude -I/opt/rocm/include -I$(CSRC) -I$(HOST) -Wno-unused-result -Wno-unused-value
SRCS  := $(wildcard $(CSRC)/*.hip) $(wildcard $(CSRC)/*.cpp) $(wildcard $(HOST)/*.cpp)
OBJS  := $(patsubst %,build/%.o,$(SRCS))
HDRS  := $(wildcard include/*.h) $(wildcard $(CSRC)/*.h) $(wildcard $(HOST)/*.hpp)

# the reference's MPI_Comm* entry points (interface/chase_c_interface.h:61-149) are built when an MPI is present
MPI_INC ?= /opt/conda/include
MPI_LIB ?= /opt/conda/lib
MPI_OUT := chase_amd/lib/libchase_hip_mpi.so
ifneq ($(wildcard $(MPI_INC)/mpi.h),)
ifneq ($(wildcard $(MPI_LIB)/libmpi.so),)
ALL_MPI := $(MPI_OUT)
endif
endif

all: $(OUT) $(ALL_MPI)

$(MPI_OUT): $(HOST)/c_interface_mpi.c $(OUT) $(HDRS)
	gcc -O2 -fPIC -shared -std=gnu11 -Iinclude -I$(MPI_INC) -o $@ $< -Lchase_amd/lib -lchase_hip $(MPI_LIB)/libmpi.so \
	    -Wl,--enable-new-dtags -Wl,-rpath,'$$ORIGIN' -Wl,-rpath,$(MPI_LIB)

build/%.o: % $(HDRS)
	@mkdir -p $(dir $@)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(OUT): $(OBJS)
	@mkdir -p $(dir $@)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS) -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib -ldl -lpthread

clean:
	rm -rf build $(OUT) $(MPI_OUT)

.PHONY: all clean
