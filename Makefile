# Top-level build: hand-written HIP for gfx950 (libgcnhip.so, the C-ABI),
# the C++ host above it (libgcnhost.so, gcn-hip), and the CPU oracle (tests only).
HIPCC   ?= /opt/rocm/bin/hipcc
ARCH    ?= gfx950
PKG      = cuda_gcn_amd
LIBDIR   = $(PKG)/lib
BINDIR   = $(PKG)/bin
OBJDIR   = build/obj
HIPFLAGS = --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-pass-failed -Iinclude
CXXFLAGS = -O2 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-sign-compare -Iinclude -I$(PKG)/host -I/opt/rocm/include -D__HIP_PLATFORM_AMD__

KSRC = $(wildcard $(PKG)/csrc/*.hip)
KOBJ = $(patsubst $(PKG)/csrc/%.hip,$(OBJDIR)/%.o,$(KSRC))
HSRC = $(filter-out $(PKG)/host/main.cpp,$(wildcard $(PKG)/host/*.cpp))
HOBJ = $(patsubst $(PKG)/host/%.cpp,$(OBJDIR)/host_%.o,$(HSRC))

all: kernels host oracle

kernels: $(LIBDIR)/libgcnhip.so
host: $(LIBDIR)/libgcnhost.so $(BINDIR)/gcn-hip

$(OBJDIR)/%.o: $(PKG)/csrc/%.hip $(wildcard $(PKG)/csrc/*.h) include/gcnhip.h
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIBDIR)/libgcnhip.so: $(KOBJ)
	@mkdir -p $(LIBDIR)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC $(KOBJ) -o $@

$(OBJDIR)/host_%.o: $(PKG)/host/%.cpp $(wildcard $(PKG)/host/*.h) include/gcnhip.h include/gcnhost.h
	@mkdir -p $(OBJDIR)
	$(CXX) $(CXXFLAGS) -c $< -o $@

# libgcnhost.so reaches the GPU only through the C-ABI of libgcnhip.so (+ RCCL for N > 1)
$(LIBDIR)/libgcnhost.so: $(HOBJ) $(LIBDIR)/libgcnhip.so
	$(CXX) -shared -fPIC $(HOBJ) -L$(LIBDIR) -lgcnhip -L/opt/rocm/lib -lrccl -lamdhip64 \
	    -Wl,-rpath,'$$ORIGIN' -Wl,-rpath,/opt/rocm/lib -o $@

# the program links the host objects directly (libgcnhost.so exports only its C entry points)
$(BINDIR)/gcn-hip: $(PKG)/host/main.cpp $(HOBJ) $(LIBDIR)/libgcnhip.so
	@mkdir -p $(BINDIR)
	$(CXX) $(CXXFLAGS) $< $(HOBJ) -L$(LIBDIR) -lgcnhip -L/opt/rocm/lib -lrccl -lamdhip64 -lpthread \
	    -Wl,-rpath,'$$ORIGIN/../lib' -Wl,-rpath,/opt/rocm/lib -o $@

oracle:
	$(MAKE) -s -C oracle

clean:
	rm -rf build $(LIBDIR) $(BINDIR)
	$(MAKE) -s -C oracle clean

.PHONY: all kernels host oracle clean
