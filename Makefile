# Builds the gfx950 hot-path library and the native parity driver.
HIPCC ?= hipcc
ARCH  ?= gfx950
CXXFLAGS = -O3 -std=c++17 --offload-arch=$(ARCH)

LIB = lcgp_amd/liblcgp_hip.so
SRC = lcgp_amd/csrc/lcgp_hip.hip
HDR = include/lcgp_hip.h
SCHED = lcgp_amd/csrc/fill_sched.h
# the same digest lcgp_amd/_hip.py::source_hash() computes: the binary carries it (lcgp_source_hash()), so a
# stale library is detected by content, not by mtime
HASH = $(shell cat $(SRC) $(HDR) $(SCHED) | sha256sum | cut -c1-16)

all: $(LIB) tests/native/test_kernels

$(LIB): $(SRC) $(HDR) $(SCHED)
	$(HIPCC) $(CXXFLAGS) -fPIC -shared '-DLCGP_SRC_HASH="LCGP_SRC_HASH=$(HASH)"' -o $@ $(SRC)

tests/native/test_kernels: tests/native/test_kernels.cpp $(LIB) $(HDR)
	$(HIPCC) $(CXXFLAGS) -o $@ tests/native/test_kernels.cpp -Llcgp_amd -llcgp_hip -Wl,-rpath,'$$ORIGIN/../../lcgp_amd'

# every native caller of the C ABI outside the test driver (signature drift shows up here, not at run time)
tools/graph_test: tools/graph_test.cpp $(LIB) $(HDR)
	$(HIPCC) $(CXXFLAGS) -o $@ tools/graph_test.cpp -Llcgp_amd -llcgp_hip -Wl,-rpath,'$$ORIGIN/../lcgp_amd'

tools: tools/graph_test

clean:
	rm -f $(LIB) tests/native/test_kernels tools/graph_test

.PHONY: all clean tools
