# Build of the MI355X (gfx950) solver.
#   make            -> super-lattice-boltzmann-2d_amd/libslb2d.so, bin/boltzmann_solver, oracle
#   make lib        -> the library only
#   make EXTRA=-DSLB_TIMING_VARIANTS  -> also the kernel variants with the arithmetic removed
#                      (tuning variants 97-99: memory floors for scripts/tune.py, wrong results)
# The kernels need separately rounded mul/add in the reference's order:
# -ffp-contract=off, no fast-math.  Host C is built like the reference's host
# (gnu99, -O3, no -march: reference GNUmakefile:36).
ROCM    ?= /opt/rocm
HIPCC   ?= $(ROCM)/bin/hipcc
CC      ?= gcc
PKG     := super-lattice-boltzmann-2d_amd
LIB     := $(PKG)/libslb2d.so

EXTRA   ?=
HIPFLAGS := --offload-arch=gfx950 -O3 -fPIC -ffp-contract=off -std=c++17 -Wall $(EXTRA)
CFLAGS   := -std=gnu99 -O3 -fPIC -ffp-contract=off -Wall -D__HIP_PLATFORM_AMD__ -I$(ROCM)/include

HOST_SRC := $(PKG)/host/slb_globals.c $(PKG)/host/slb_bessel.c $(PKG)/host/slb_cli.c \
            $(PKG)/host/slb_solver.c $(PKG)/host/slb_main.c $(PKG)/host/slb_link.c $(PKG)/host/slb_slabs.c
HOST_OBJ := $(HOST_SRC:.c=.o)
HDRS     := include/slb_abi.h include/slb_slabs.h $(PKG)/host/slb_host.h

all: lib bin/boltzmann_solver bin/dropin_host bin/dropin_host_rdynamic oracle

lib: $(LIB)

$(PKG)/csrc/slb_kernels.o: $(PKG)/csrc/slb_kernels.hip $(wildcard $(PKG)/csrc/*.inc) $(HDRS)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(PKG)/host/%.o: $(PKG)/host/%.c $(HDRS)
	$(CC) $(CFLAGS) -c $< -o $@

$(LIB): $(PKG)/csrc/slb_kernels.o $(HOST_OBJ)
	$(HIPCC) --offload-arch=gfx950 -shared -fPIC -o $@ $^ -lm -ldl -lpthread -Wl,-rpath,$(ROCM)/lib

bin/boltzmann_solver: $(PKG)/host/boltzmann_solver.c $(LIB)
	@mkdir -p bin
	$(CC) $(CFLAGS) -o $@ $< -L$(PKG) -lslb2d -Wl,-rpath,'$$ORIGIN/../$(PKG)' -Wl,-rpath,$(ROCM)/lib -lm

# A host in the reference's shape (it defines the host_* parameters itself and owns the device
# memory), linked against the library the way the reference links its kernels' object:
# once as is, once with -rdynamic.  Used by tests/test_dropin_gpu.py.
DROPIN_LINK = -L$(PKG) -lslb2d -L$(ROCM)/lib -lamdhip64 -Wl,-rpath,'$$ORIGIN/../$(PKG)' -Wl,-rpath,$(ROCM)/lib -lm
bin/dropin_host: tests/dropin/dropin_host.c $(LIB)
	@mkdir -p bin
	$(CC) -std=gnu99 -O3 -D__HIP_PLATFORM_AMD__ -I$(ROCM)/include -o $@ $< $(DROPIN_LINK)
bin/dropin_host_rdynamic: tests/dropin/dropin_host.c $(LIB)
	@mkdir -p bin
	$(CC) -std=gnu99 -O3 -D__HIP_PLATFORM_AMD__ -I$(ROCM)/include -rdynamic -o $@ $< $(DROPIN_LINK)

oracle:
	$(MAKE) -C oracle

clean:
	rm -f $(PKG)/csrc/*.o $(PKG)/host/*.o $(LIB) bin/boltzmann_solver bin/dropin_host bin/dropin_host_rdynamic
	$(MAKE) -C oracle clean

.PHONY: all lib oracle clean
