# Top-level build: HIP library (gfx950 only), simulator, oracle.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
HIPFLAGS ?= -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -Wall -Wno-unused-function
CSRC := dentist_amd/csrc
LIB := dentist_amd/libdentist_hip.so
SIM := dentist_amd/sim/libdh_sim.so

DAZZ_TOOLS := fasta2DB fasta2DAM DBsplit DBrm DBdump DBshow DBdust LAmerge DAScover DASqv computeintrinsicqv daccord merge-insertions LAsplit Catrack TANmask
TOOLS := tools/daligner tools/damapper tools/datander tools/dazz_tools $(addprefix tools/,$(DAZZ_TOOLS))

all: $(LIB) $(SIM) oracle $(TOOLS)

tools/daligner: tools/aligner_main.cpp include/dentist_hip.h $(LIB)
	$(HIPCC) -O2 -std=c++17 -o $@ $< -Ldentist_amd -ldentist_hip -Wl,-rpath,'$$ORIGIN/../dentist_amd'

tools/damapper: tools/daligner
	cp $< $@

tools/datander: tools/daligner
	cp $< $@

tools/dazz_tools: tools/dazz_main.cpp include/dentist_hip.h $(LIB)
	$(HIPCC) -O2 -std=c++17 -o $@ $< -Ldentist_amd -ldentist_hip -Wl,-rpath,'$$ORIGIN/../dentist_amd'

$(addprefix tools/,$(DAZZ_TOOLS)): tools/dazz_tools
	cp $< $@

# one object per translation unit (build/ is git-ignored): `make -j8` rebuilds only what changed
OBJDIR := build/obj
SRCS := $(wildcard $(CSRC)/*.hip) $(wildcard $(CSRC)/*.cpp)
OBJS := $(patsubst $(CSRC)/%,$(OBJDIR)/%.o,$(SRCS))
HDRS := $(wildcard $(CSRC)/*.h) include/dentist_hip.h

$(OBJDIR)/%.o: $(CSRC)/% $(HDRS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -c -o $@ $<

$(LIB): $(OBJS)
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(OBJS)

$(SIM): dentist_amd/sim/sim.cpp
	g++ -O2 -fPIC -shared -fopenmp -o $@ $<

oracle:
	$(MAKE) -C oracle

clean:
	rm -f $(LIB) $(SIM) $(TOOLS); $(MAKE) -C oracle clean

.PHONY: all oracle clean

# the DH-2 lane code (dentist_amd/csrc/dh_tile.h) compiled for the CPU: test infrastructure
tests/native/libdh_tile_host.so: tests/native/tile_host.cpp dentist_amd/csrc/dh_tile.h dentist_amd/csrc/dh_device.h
	g++ -O2 -g -shared -fPIC -std=c++17 -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -Wno-unknown-pragmas -o $@ $<

# the host thread pool (dentist_amd/csrc/dh_parallel.h) on its own: test infrastructure
tests/native/libdh_pool_host.so: tests/native/pool_host.cpp dentist_amd/csrc/dh_parallel.h
	g++ -O2 -g -shared -fPIC -std=c++17 -pthread -o $@ $<
