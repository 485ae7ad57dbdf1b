# Builds the product library (HIP, gfx950) and the checker libraries.
HIPCC ?= /opt/rocm/bin/hipcc
CC ?= gcc
ARCH ?= gfx950
CSRC = quisk_amd/csrc
LIB = quisk_amd/libqkengine.so
HIPFLAGS = -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -Wall -Wno-unused-function -Wno-unused-variable -Wno-unused-value

PYINC := $(shell python3 -c "import sysconfig; print(sysconfig.get_paths()['include'])")
EXT = quisk_amd/ext/_quisk.so
PLUGIN = tests/plugin/qk_test_source.so

all: $(LIB) $(EXT) $(PLUGIN) oracle

# the `_quisk` CPython extension (quisk.c's module interface for the path) over the C ABI
$(EXT): $(CSRC)/quisk_module.c include/qk_engine.h $(LIB)
	mkdir -p quisk_amd/ext
	$(CC) -O2 -Wall -fPIC -shared -I$(PYINC) -o $@ $(CSRC)/quisk_module.c -Lquisk_amd -lqkengine -Wl,-rpath,'$$ORIGIN/..'

# a sample-source plug-in that binds through the capsule (test infrastructure)
$(PLUGIN): tests/plugin/qk_test_source.c
	$(CC) -O2 -Wall -fPIC -shared -I$(PYINC) -o $@ $<

$(CSRC)/qk_phasor.o: $(CSRC)/qk_phasor.c
	$(CC) -O2 -ffp-contract=off -fPIC -c -o $@ $<

$(CSRC)/qk_engine.o: $(CSRC)/qk_engine.hip $(CSRC)/qk_kernels.h $(CSRC)/qk_fft.h $(CSRC)/qk_coefs.inc include/qk_engine.h
	$(HIPCC) $(HIPFLAGS) -c -o $@ $(CSRC)/qk_engine.hip

$(LIB): $(CSRC)/qk_engine.o $(CSRC)/qk_phasor.o
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(CSRC)/qk_engine.o $(CSRC)/qk_phasor.o -lm -ldl

oracle:
	$(MAKE) -C oracle

clean:
	rm -f $(LIB) $(EXT) $(PLUGIN) $(CSRC)/*.o
	$(MAKE) -C oracle clean
.PHONY: all oracle clean
