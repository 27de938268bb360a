# Builds libtcow_hip.so (gfx950 only) and the oracle's C/CPU helpers.  `python __graft_entry__.py` calls this.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
CSRC := tcow_amd/csrc
OBJ := build/obj
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Wno-unused-result -ffp-contract=off
SRCS := $(wildcard $(CSRC)/*.hip) $(wildcard $(CSRC)/*.cpp)
OBJS := $(patsubst $(CSRC)/%,$(OBJ)/%.o,$(SRCS))
LIB := tcow_amd/libtcow_hip.so
# the same sources with the 16-bit storage format set to IEEE binary16 (csrc/common.h): precision='fp16'
OBJ16 := build/obj_fp16
OBJS16 := $(patsubst $(CSRC)/%,$(OBJ16)/%.o,$(SRCS))
LIB16 := tcow_amd/libtcow_hip_fp16.so

all: $(LIB) $(LIB16)

# (attention: no NaN arithmetic on the path -- lets fmaxf chains become v_max3_f32 without canonicalising v_max instructions)
$(OBJ)/attention_bf16.hip.o: EXTRA := -fno-honor-nans -fno-slp-vectorize
$(OBJ)/%.hip.o: $(CSRC)/%.hip $(CSRC)/common.h $(CSRC)/gemm_f32.h $(CSRC)/gemm_nt_common.h $(CSRC)/attention_tiles.h $(CSRC)/attention_common.h include/tcow_hip.h
	@mkdir -p $(OBJ)
	$(HIPCC) $(HIPFLAGS) $(EXTRA) -x hip -c $< -o $@

$(OBJ)/%.cpp.o: $(CSRC)/%.cpp $(CSRC)/common.h include/tcow_hip.h
	@mkdir -p $(OBJ)
	$(HIPCC) $(HIPFLAGS) -x hip -c $< -o $@

$(LIB): $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS)

$(OBJ16)/attention_bf16.hip.o: EXTRA := -fno-honor-nans -fno-slp-vectorize
$(OBJ16)/%.hip.o: $(CSRC)/%.hip $(CSRC)/common.h $(CSRC)/gemm_f32.h $(CSRC)/gemm_nt_common.h $(CSRC)/attention_tiles.h $(CSRC)/attention_common.h include/tcow_hip.h
	@mkdir -p $(OBJ16)
	$(HIPCC) $(HIPFLAGS) $(EXTRA) -DTCOW_FP16 -x hip -c $< -o $@

$(OBJ16)/%.cpp.o: $(CSRC)/%.cpp $(CSRC)/common.h include/tcow_hip.h
	@mkdir -p $(OBJ16)
	$(HIPCC) $(HIPFLAGS) -DTCOW_FP16 -x hip -c $< -o $@

$(LIB16): $(OBJS16)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS16)

clean:
	rm -rf build $(LIB) $(LIB16)
.PHONY: all clean

# microbenchmarks (run on the GPU box: build/ubench_valu > profiles/rNN_ubench_valu.txt)
ubench: build/ubench_valu
build/ubench_gemm: tools/ubench_gemm.hip tools/gemm_p8.hip $(CSRC)/common.h
	@mkdir -p build
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 -ffp-contract=off -Wno-unused-result -x hip $< -o $@

ubench_gemm: build/ubench_gemm

build/ubench_tn: tools/ubench_tn.hip $(CSRC)/gemm_bf16.hip $(CSRC)/common.h
	@mkdir -p build
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 -ffp-contract=off -Wno-unused-result -x hip $< -o $@

ubench_tn: build/ubench_tn

build/ubench_valu: tools/ubench_valu.hip tools/attn_fwd_variants.inc $(CSRC)/attention_bf16.hip $(CSRC)/common.h
	@mkdir -p build
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 -ffp-contract=off -fno-honor-nans -fno-slp-vectorize -Wno-unused-result -x hip $< -o $@

build/ubench_c2: tools/ubench_c2.hip $(CSRC)/gemm_nt_c2.hip $(CSRC)/gemm_nt_common.h $(CSRC)/common.h
	@mkdir -p build
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 -ffp-contract=off -Wno-unused-result -x hip $< -o $@

build/ubench_glds: tools/ubench_glds.hip
	@mkdir -p build
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 -x hip $< -o $@

build/ubench_tn_ab: tools/ubench_tn_ab.hip $(CSRC)/gemm_bf16.hip $(CSRC)/gemm_nt_common.h $(CSRC)/common.h
	@mkdir -p build
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 -ffp-contract=off -Wno-unused-result -x hip $< -o $@

build/ubench_mfma: tools/ubench_mfma.hip
	@mkdir -p build
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 -x hip $< -o $@
