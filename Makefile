# Top-level build: libmisift.so (HIP kernels + C-ABI, hipcc, gfx950 only),
# libcudasift.so (C++ drop-in shim over the C-ABI, plain g++), the oracle (test infra).
ROCM     ?= /opt/rocm
HIPCC    ?= $(ROCM)/bin/hipcc
CXX      ?= g++
CSRC     := cudasift_amd/csrc
BUILD    := build
# -ffp-contract=off: fused multiply-adds are written explicitly (arithmetic contract with oracle/)
HIPFLAGS := --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude -I$(CSRC) \
            -Wno-unused-result -Wno-unused-value
HIPSRCS  := $(CSRC)/kernels_pyramid.hip $(CSRC)/kernels_dog.hip $(CSRC)/kernels_points.hip \
            $(CSRC)/kernels_match.hip $(CSRC)/misift_host.hip $(CSRC)/homography.hip $(CSRC)/pipeline.hip \
            $(CSRC)/multigpu.hip
HIPOBJS  := $(patsubst $(CSRC)/%.hip,$(BUILD)/%.o,$(HIPSRCS))

all: cudasift_amd/libmisift.so cudasift_amd/libcudasift.so cudasift_amd/libcudasift_managed.so oracle dropin build/pmc_calib build/valu_rates build/scan_rates build/single_call

$(BUILD)/%.o: $(CSRC)/%.hip $(CSRC)/common.hpp $(CSRC)/chain.hpp include/misift.h
	@mkdir -p $(BUILD)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

cudasift_amd/libmisift.so: $(HIPOBJS)
	$(HIPCC) --offload-arch=gfx950 -shared -fPIC -o $@ $(HIPOBJS) -ldl

cudasift_amd/libcudasift.so: $(CSRC)/shim_cudasift.cpp include/cudaSift.h include/cudaImage.h include/misift.h cudasift_amd/libmisift.so
	$(CXX) -O2 -std=c++17 -fPIC -shared -Iinclude -o $@ $(CSRC)/shim_cudasift.cpp -Lcudasift_amd -lmisift -Wl,-rpath,'$$ORIGIN'

# FETCH_SIZE calibration kernels (tools/pmc_calib.py runs them under rocprofv3 on the GPU box)
build/pmc_calib: tools/pmc_calib.hip
	@mkdir -p $(BUILD)
	$(HIPCC) --offload-arch=gfx950 -O3 -std=c++17 -o $@ $<

# instruction issue-cost microbenchmark (what the VALU-bound kernels are priced against)
build/valu_rates: tools/valu_rates.hip
	@mkdir -p $(BUILD)
	$(HIPCC) --offload-arch=gfx950 -O3 -std=c++17 -o $@ $<

# the instruction stream of one dog_scan row in isolation (why the scan stops where it does)
build/scan_rates: tools/scan_rates.hip
	@mkdir -p $(BUILD)
	$(HIPCC) --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o $@ $<

# the reference demo's inner loop on synthetic frames through the drop-in API (tools/single_call.sh: BASELINE configs 2-3)
build/single_call: tools/single_call.cpp include/cudaSift.h include/cudaImage.h cudasift_amd/libcudasift.so
	@mkdir -p $(BUILD)
	$(CXX) -O2 -std=c++17 -Iinclude -o $@ tools/single_call.cpp -Lcudasift_amd -lcudasift -lmisift -Wl,-rpath,'$$ORIGIN/../cudasift_amd'

# the MANAGEDMEM flavour of the drop-in API (cudaSift.h:27-32: SiftData holds ONE managed pointer, m_data)
cudasift_amd/libcudasift_managed.so: $(CSRC)/shim_cudasift.cpp include/cudaSift.h include/cudaImage.h include/misift.h cudasift_amd/libmisift.so
	$(CXX) -O2 -std=c++17 -fPIC -shared -DMANAGEDMEM -Iinclude -o $@ $(CSRC)/shim_cudasift.cpp -Lcudasift_amd -lmisift -Wl,-rpath,'$$ORIGIN'

oracle:
	$(MAKE) -C oracle all

clean:
	rm -rf $(BUILD) cudasift_amd/libmisift.so cudasift_amd/libcudasift.so
	$(MAKE) -C oracle clean

.PHONY: all oracle clean

# Drop-in demonstration: the reference's OWN mainSift.cpp + geomFuncs.cpp, compiled unchanged from
# $(REF) against include/cudaSift.h + libcudasift.so (mini-OpenCV stand-in because OpenCV is absent).
# Output under oracle/_ref/ (git-ignored, travels to the GPU box).  Skipped where $(REF) is absent.
REF ?= /root/reference
dropin: cudasift_amd/libcudasift.so cudasift_amd/libcudasift_managed.so
	@if [ -f $(REF)/mainSift.cpp ]; then mkdir -p oracle/_ref && \
	  $(CXX) -O2 -std=c++17 -Iinclude -Icudasift_amd/compat -o oracle/_ref/cudasift_dropin \
	    $(REF)/mainSift.cpp $(REF)/geomFuncs.cpp -Lcudasift_amd -lcudasift -lmisift \
	    -Wl,-rpath,'$$ORIGIN/../../cudasift_amd' && echo "built oracle/_ref/cudasift_dropin" && \
	  $(CXX) -O2 -std=c++17 -DMANAGEDMEM -Iinclude -Icudasift_amd/compat -o oracle/_ref/cudasift_dropin_managed \
	    $(REF)/mainSift.cpp $(REF)/geomFuncs.cpp -Lcudasift_amd -lcudasift_managed -lmisift \
	    -Wl,-rpath,'$$ORIGIN/../../cudasift_amd' && echo "built oracle/_ref/cudasift_dropin_managed"; \
	else echo "dropin: $(REF) absent, keeping prebuilt binary"; fi
.PHONY: dropin
