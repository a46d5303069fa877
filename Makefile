# Top-level build: hand-written HIP for gfx950 (libgcnhip.so, the C-ABI),
# the C++ host above it (libgcnhost.so, gcn-hip), and the CPU oracle (tests only).
HIPCC   ?= /opt/rocm/bin/hipcc
ARCH    ?= gfx950
PKG      = cuda_gcn_amd
LIBDIR   = $(PKG)/lib
BINDIR   = $(PKG)/bin
OBJDIR   = build/obj
# EXPERIMENTS=1: also compile the variants DESIGN.md records as built, bit-identical and slower (packed dH1 rows, the
# persistent index-prefetching aggregation, non-temporal row loads, the in-launch segment sum, the persistent weight
# gradient, W staged in LDS for the sparse forward, the backward pipeline of the host).  Their tests skip without it.
EXPFLAG  = $(if $(EXPERIMENTS),-DGCNHIP_EXPERIMENTS,)
HIPFLAGS = --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-pass-failed -Iinclude $(EXPFLAG)
CXXFLAGS = -O2 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-sign-compare -Iinclude -I$(PKG)/host -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ $(EXPFLAG)

KSRC = $(wildcard $(PKG)/csrc/*.hip)
KOBJ = $(patsubst $(PKG)/csrc/%.hip,$(OBJDIR)/%.o,$(KSRC))
HSRC = $(filter-out $(PKG)/host/main.cpp,$(wildcard $(PKG)/host/*.cpp))
HOBJ = $(patsubst $(PKG)/host/%.cpp,$(OBJDIR)/host_%.o,$(HSRC))

all: kernels host oracle

# the flavour the objects were compiled in: switching EXPERIMENTS rebuilds them (make does not see a changed flag by itself,
# and a library mixing the two flavours reports gcnhip_experiments() of whichever ctx.hip it happened to keep)
FLAVOUR = $(OBJDIR)/.flavour
$(FLAVOUR): FORCE
	@mkdir -p $(OBJDIR)
	@echo '$(EXPFLAG)' | cmp -s - $@ || echo '$(EXPFLAG)' > $@
FORCE:

kernels: $(LIBDIR)/libgcnhip.so
host: $(LIBDIR)/libgcnhost.so $(BINDIR)/gcn-hip

$(OBJDIR)/%.o: $(PKG)/csrc/%.hip $(wildcard $(PKG)/csrc/*.h) $(wildcard include/*.h) $(FLAVOUR)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIBDIR)/libgcnhip.so: $(KOBJ)
	@mkdir -p $(LIBDIR)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC $(KOBJ) -o $@

$(OBJDIR)/host_%.o: $(PKG)/host/%.cpp $(wildcard $(PKG)/host/*.h) $(wildcard include/*.h) $(FLAVOUR)
	@mkdir -p $(OBJDIR)
	$(CXX) $(CXXFLAGS) -c $< -o $@

# libgcnhost.so reaches the GPU only through the C-ABI of libgcnhip.so (+ RCCL for N > 1)
$(LIBDIR)/libgcnhost.so: $(HOBJ) $(LIBDIR)/libgcnhip.so
	$(CXX) -shared -fPIC $(HOBJ) -L$(LIBDIR) -lgcnhip -L/opt/rocm/lib -lrccl -lamdhip64 \
	    -Wl,-rpath,'$$ORIGIN' -Wl,-rpath,/opt/rocm/lib -o $@

# the program links the host objects directly (libgcnhost.so exports only its C entry points)
$(BINDIR)/gcn-hip: $(PKG)/host/main.cpp $(HOBJ) $(LIBDIR)/libgcnhip.so $(FLAVOUR)
	@mkdir -p $(BINDIR)
	$(CXX) $(CXXFLAGS) $< $(HOBJ) -L$(LIBDIR) -lgcnhip -L/opt/rocm/lib -lrccl -lamdhip64 -lpthread \
	    -Wl,-rpath,'$$ORIGIN/../lib' -Wl,-rpath,/opt/rocm/lib -o $@

oracle:
	$(MAKE) -s -C oracle

# measurement tools that are HIP programs of their own (not part of the product): the gather-ceiling microbenchmark
tools: build/libgatherpeak.so
build/libgatherpeak.so: tools/gather_peak.hip
	@mkdir -p build
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 -shared -fPIC $< -o $@

# CPU sanitizer build of the host logic and the oracle (the GPU box cannot run ASan; the reference has no
# sanitizer target at all, /root/reference/Makefile:6-7).  `make asan-test` runs the CPU tests of the parser,
# partition, RNG replay, R-MAT generator, C-ABI tables and the oracle pin against the instrumented libraries.
ASANDIR  = build/asan
SANFLAGS = -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -g -O1
asan: $(LIBDIR)/libgcnhip.so
	@mkdir -p $(ASANDIR)
	$(CXX) $(filter-out -O2,$(CXXFLAGS)) $(SANFLAGS) -shared $(HSRC) -L$(LIBDIR) -lgcnhip -L/opt/rocm/lib -lrccl -lamdhip64 \
	    -Wl,-rpath,'$(abspath $(LIBDIR))' -Wl,-rpath,/opt/rocm/lib -o $(ASANDIR)/libgcnhost.so
	cp $(LIBDIR)/libgcnhip.so $(ASANDIR)/libgcnhip.so
	$(CC) -std=gnu11 -Wall -Wno-unused-result $(SANFLAGS) -fPIC -shared oracle/gcn_oracle.c -o $(ASANDIR)/liboracle.so -lm
asan-test: asan
	LD_PRELOAD="$$($(CC) -print-file-name=libasan.so) $$($(CC) -print-file-name=libubsan.so)" \
	ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1 \
	GCN_LIBDIR=$(abspath $(ASANDIR)) GCN_ORACLE_LIB=$(abspath $(ASANDIR))/liboracle.so \
	python3 -m pytest tests/test_host_cpu.py tests/test_abi_cpu.py tests/test_oracle_pin.py tests/test_datagen_cpu.py -q -x -m "not gpu" -p no:cacheprovider

clean:
	rm -rf build $(LIBDIR) $(BINDIR)
	$(MAKE) -s -C oracle clean

.PHONY: all kernels host oracle tools clean asan asan-test
