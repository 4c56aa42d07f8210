# Top-level convenience targets (everything can also be built by `python -c "import __graft_entry__ as g; g.build()"`).
PKG   := experimental_gpu_programming_for_a_spectral_numerical_integration_amd
HIPCC ?= /opt/rocm/bin/hipcc
HIPFLAGS := --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -I$(PKG)/csrc

all: lib oracle

lib:
	$(MAKE) -C $(PKG)/csrc

oracle:
	$(MAKE) -C oracle

# measurement tooling (not product code); binaries land in build/ (git-ignored, travels with gpurun)
tools: build/mfma_f64_probe build/apply_diag build/store_rate_probe build/small_diag build/stream_probe build/host_transfer_probe build/store_mfma_interleave build/valu_mfma_cost

build/mfma_f64_probe: tools/mfma_f64_probe.hip
	mkdir -p build && $(HIPCC) --offload-arch=gfx950 -O3 -include algorithm $< -o $@

build/store_rate_probe: tools/store_rate_probe.hip
	mkdir -p build && $(HIPCC) --offload-arch=gfx950 -O3 $< -o $@

GEOS := $(PKG)/csrc/sri_apply_tall.hip $(PKG)/csrc/sri_apply_mid.hip $(PKG)/csrc/sri_apply_strip.hip $(PKG)/csrc/sri_apply_flat.hip $(PKG)/csrc/sri_apply_wide.hip
build/apply_diag: tools/apply_diag.hip $(PKG)/csrc/sri_apply.hip $(PKG)/csrc/sri_apply_kernel.h $(PKG)/csrc/sri_apply_geometries.h $(PKG)/csrc/sri_apply_geometry_impl.h $(GEOS) $(PKG)/csrc/sri_host_panels.hip $(PKG)/csrc/sri_host_pipeline.h $(PKG)/csrc/sri_apply_small.hip $(PKG)/csrc/sri_host.cpp $(PKG)/csrc/sri_internal.h include/sri_capi.h
	mkdir -p build && $(HIPCC) $(HIPFLAGS) -DSRI_DIAG -Wno-pass-failed tools/apply_diag.hip $(GEOS) $(PKG)/csrc/sri_apply_small.hip $(PKG)/csrc/sri_host_panels.hip $(PKG)/csrc/sri_host.cpp -o $@

build/small_diag: tools/small_diag.hip $(PKG)/csrc/sri_apply_small.hip $(PKG)/csrc/sri_internal.h
	mkdir -p build && $(HIPCC) $(HIPFLAGS) tools/small_diag.hip -o $@

build/stream_probe: tools/stream_probe.hip
	mkdir -p build && $(HIPCC) --offload-arch=gfx950 -O3 -std=c++17 $< -o $@

build/host_transfer_probe: tools/host_transfer_probe.hip
	mkdir -p build && $(HIPCC) --offload-arch=gfx950 -O3 -std=c++17 -pthread $< -o $@

build/store_mfma_interleave: tools/store_mfma_interleave.hip
	mkdir -p build && $(HIPCC) --offload-arch=gfx950 -O3 -std=c++17 $< -o $@

build/valu_mfma_cost: tools/valu_mfma_cost.hip
	mkdir -p build && $(HIPCC) --offload-arch=gfx950 -O3 -std=c++17 $< -o $@

clean:
	$(MAKE) -C $(PKG)/csrc clean; $(MAKE) -C oracle clean; rm -rf build

.PHONY: all lib oracle tools clean
