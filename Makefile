# Builds the gfx950 hot-path library and the native parity driver.
HIPCC ?= hipcc
ARCH  ?= gfx950
CXXFLAGS = -O3 -std=c++17 --offload-arch=$(ARCH)

LIB = lcgp_amd/liblcgp_hip.so
SRC = lcgp_amd/csrc/lcgp_hip.hip
HDR = include/lcgp_hip.h

all: $(LIB) tests/native/test_kernels

$(LIB): $(SRC) $(HDR)
	$(HIPCC) $(CXXFLAGS) -fPIC -shared -o $@ $(SRC)

tests/native/test_kernels: tests/native/test_kernels.cpp $(LIB) $(HDR)
	$(HIPCC) $(CXXFLAGS) -o $@ tests/native/test_kernels.cpp -Llcgp_amd -llcgp_hip -Wl,-rpath,'$$ORIGIN/../../lcgp_amd'

clean:
	rm -f $(LIB) tests/native/test_kernels

.PHONY: all clean
