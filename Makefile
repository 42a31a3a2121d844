# (-amdgpu-atomic-optimizer-strategy=None: the persistent GEMM keeps one returning atomic in flight per workgroup; the
# optimizer's wave reduction + readfirstlane would make the wave wait for it where it is issued.)
# (-amdgpu-mfma-vgpr-form: MFMA accumulators stay in VGPRs; the default put them in AccVGPRs and paid ~7 v_accvgpr moves
# per attention score element.)
# Builds libssak_hip.so (hand-written HIP kernels for gfx950 + the C ABI of include/ssak_hip.h).
# `make` cross-compiles without a GPU; the .so is built in-tree so that it travels with the snapshot.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
CSRC := ssak_amd/csrc
# (DEV objects in their own directory: a release build after a DEV build must not link objects that read the environment)
OBJ ?= build/obj$(if $(DEV),_dev,)
LIB ?= ssak_amd/lib/libssak_hip.so
# experiment builds next to the product: make OBJ=build/obj_x LIB=tools/ab_x.so EXTRA=-DSOME_VARIANT
EXTRA ?=
# `make DEV=1`: development switches read from the environment (SSAK_GEMM_P8, SSAK_ATTN_TILE, ...) are compiled in; the
# release library has none (common.h: SSAK_DEV_ENV)
DEVFLAGS := $(if $(DEV),-DSSAK_DEV,)
HIPFLAGS := $(DEVFLAGS) $(EXTRA) --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-lambda-capture -Iinclude -ffp-contract=fast -mllvm -amdgpu-mfma-vgpr-form -mllvm -amdgpu-atomic-optimizer-strategy=None
SRCS := $(wildcard $(CSRC)/*.hip) $(wildcard $(CSRC)/*.cpp)
OBJS := $(patsubst $(CSRC)/%,$(OBJ)/%.o,$(SRCS))

all: $(LIB)

$(OBJ)/%.hip.o: $(CSRC)/%.hip $(CSRC)/common.h $(CSRC)/gemm_common.h $(CSRC)/kernels.h include/ssak_hip.h
	@mkdir -p $(OBJ)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(OBJ)/%.cpp.o: $(CSRC)/%.cpp include/ssak_hip.h
	@mkdir -p $(OBJ)
	$(HIPCC) $(HIPFLAGS) -x hip -c $< -o $@

$(LIB): $(OBJS)
	@mkdir -p $(dir $(LIB))
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS)

clean:
	rm -rf build $(LIB)

.PHONY: all clean
