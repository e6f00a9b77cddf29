# malva-hip build.  Everything is built in-tree so the artefacts travel with the
# repo snapshot to the GPU box; none of them is committed (.gitignore).
#
#   make lib      HIP kernels + C-ABI       -> malva_amd/lib/libmalva_hip.so   (hipcc, gfx950)
#   make cli      C++17 host driver         -> bin/malva-geno
#   make oracle   CPU restatement (checker) -> oracle/libmalva_oracle.so       (gcc)
#   make ref      the reference's own vendored xxhash.c, compiled where it lies
#                 under /root/reference     -> oracle/_ref/libxxhash_ref.so    (only when the reference is present)

HIPCC      ?= /opt/rocm/bin/hipcc
CC         ?= gcc
CXX        ?= g++
ARCH       ?= gfx950
REFERENCE  ?= /root/reference
# zstd for the reference's index container: the image has the runtime library system-wide and the header only under
# /opt/conda (searched AFTER the system directories, so nothing else is taken from there)
ZSTD_INC   ?= /opt/conda/include
ZSTD_LIB   ?= /usr/lib/x86_64-linux-gnu/libzstd.so.1

HIPFLAGS   := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude -Imalva_amd/csrc \
              -Wall -Wno-unused-function -Wno-unused-value -fvisibility=hidden
CSRC       := $(wildcard malva_amd/csrc/*.hip)
CHDR       := $(wildcard malva_amd/csrc/*.h) $(wildcard include/*.h)
HOSTSRC    := $(wildcard malva_amd/host/*.cpp)
HOSTHDR    := $(wildcard malva_amd/host/*.hpp)

all: lib oracle ref cli

lib: malva_amd/lib/libmalva_hip.so
malva_amd/lib/libmalva_hip.so: $(CSRC) $(CHDR)
	@mkdir -p malva_amd/lib
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(CSRC)

cli: bin/malva-geno
bin/malva-geno: $(HOSTSRC) $(HOSTHDR) malva_amd/lib/libmalva_hip.so
	@mkdir -p bin
	$(CXX) -std=c++17 -O2 -Wall -pthread -Iinclude -Imalva_amd/host -idirafter $(ZSTD_INC) -o $@ $(HOSTSRC) \
	    -Lmalva_amd/lib -lmalva_hip -lz $(ZSTD_LIB) -Wl,-rpath,'$$ORIGIN/../malva_amd/lib'

oracle: oracle/libmalva_oracle.so
oracle/libmalva_oracle.so: oracle/malva_oracle.c
	$(CC) -O2 -ffp-contract=off -fPIC -shared -fvisibility=hidden -Wall -pthread -o $@ $< -lm

# The reference build: its own xxhash.c (which includes its own xxhash.h),
# untouched, straight from the read-only checkout.  Skipped on machines that do
# not have the reference (the GPU box); the prebuilt .so travels instead.
ref:
	@if [ -f $(REFERENCE)/xxhash.c ]; then \
	    mkdir -p oracle/_ref && \
	    $(CC) -O2 -fPIC -shared -o oracle/_ref/libxxhash_ref.so $(REFERENCE)/xxhash.c && \
	    echo "built oracle/_ref/libxxhash_ref.so from $(REFERENCE)/xxhash.c"; \
	else echo "reference not present: keeping prebuilt oracle/_ref (if any)"; fi

# the microbenchmark behind DESIGN.md's bound for the filter kernel (run it on an MI355X)
microbench: bin/l2_gather_bench bin/ticket_gate_bench
bin/l2_gather_bench: tools/l2_gather_bench.hip
	@mkdir -p bin
	$(HIPCC) --offload-arch=$(ARCH) -O3 -o $@ $<
# the bare access pattern of the ticket form's pass two
bin/ticket_gate_bench: tools/ticket_gate_bench.hip
	@mkdir -p bin
	$(HIPCC) --offload-arch=$(ARCH) -O3 -Wno-unused-value -o $@ $<

clean:
	rm -rf malva_amd/lib bin oracle/libmalva_oracle.so oracle/_ref

.PHONY: all lib cli oracle ref microbench clean
