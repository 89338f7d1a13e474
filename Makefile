# Builds the product library (HIP kernels + C ABI + host mirror) for gfx950
# and the CPU oracle.  `make` here == __graft_entry__.build().
HIPCC    ?= /opt/rocm/bin/hipcc
ARCH     ?= gfx950
# -ffp-contract=off: results are compared bit-for-bit with an unfused f64 reference.
EXTRA    ?=
HIPFLAGS ?= $(EXTRA) -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -ffp-contract=off -fno-fast-math -munsafe-fp-atomics -Wall -Wno-unused-result

LIB  := raytracing_amd/librtx.so
SRCS := raytracing_amd/csrc/rtx_kernels.hip raytracing_amd/csrc/rtx_build.hip raytracing_amd/csrc/rtx_output.hip raytracing_amd/csrc/rtx_noalloc.hip raytracing_amd/csrc/rtx_api.cpp raytracing_amd/csrc/rtx_multi.cpp raytracing_amd/host/rtx_host.cpp
OBJS := $(patsubst %,build/%.o,$(notdir $(SRCS)))
HDRS := raytracing_amd/csrc/rtx_internal.h include/rtx.h include/rtx_host.h raytracing_amd/csrc/rtx_device.h raytracing_amd/csrc/rtx_kernels.h raytracing_amd/csrc/rtx_scan.h raytracing_amd/host/raytracing.hpp raytracing_amd/host/examples.hpp
vpath %.hip raytracing_amd/csrc
vpath %.cpp raytracing_amd/csrc raytracing_amd/host

BIN  := bin/rtx_example

all: $(LIB) $(BIN) oracle build/row_regs.ok

# The render kernels that read a row of rays ahead load it with inline assembly the compiler does not track; that nothing
# reads those registers before the wait is a property of the compiler's register allocation, checked on its assembly output.
build/row_regs.ok: raytracing_amd/csrc/rtx_kernels.hip $(HDRS) tools/check_row_regs.py
	@mkdir -p build
	HIPCC=$(HIPCC) python3 tools/check_row_regs.py
	@touch $@

build/%.o: % $(HDRS)
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -c -o $@ -x hip $<

$(LIB): $(OBJS)
	$(HIPCC) -shared -fPIC --offload-arch=$(ARCH) -o $@ $(OBJS) -ldl -lpthread

# the reference's `cargo run --example <name>` (host code only; links the library)
$(BIN): examples/rtx_example.cpp raytracing_amd/host/examples.hpp raytracing_amd/host/raytracing.hpp include/rtx.h $(LIB)
	@mkdir -p bin
	g++ -O2 -std=c++17 -ffp-contract=off -Wall -o $@ examples/rtx_example.cpp -Lraytracing_amd -lrtx -Wl,-rpath,'$$ORIGIN/../raytracing_amd'

oracle:
	$(MAKE) -C oracle

clean:
	rm -rf $(LIB) build bin
	$(MAKE) -C oracle clean
.PHONY: all oracle clean
