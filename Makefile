# Builds the gfx950 C-ABI library (product) and the CPU oracle (test infrastructure).
HIPCC ?= hipcc
ARCH ?= gfx950
HIPFLAGS ?= -O3 -std=c++17 --offload-arch=$(ARCH) -fPIC -Wall -Wno-unused-result -Wno-unused-function -ffp-contract=fast
CSRC := small_gicp_amd/csrc
OBJDIR := build/obj
SRCS := $(wildcard $(CSRC)/*.hip)
OBJS := $(patsubst $(CSRC)/%.hip,$(OBJDIR)/%.o,$(SRCS))
LIB := small_gicp_amd/lib/libsmall_gicp_amd.so

all: lib oracle
lib: $(LIB)
$(OBJDIR)/%.o: $(CSRC)/%.hip $(wildcard $(CSRC)/*.hpp) include/small_gicp_amd.h
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
$(LIB): $(OBJS)
	@mkdir -p small_gicp_amd/lib
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS)
# diagnostics build: counts the loop-body executions of the kd walk (scripts/diag_trips.py); not part of the product
TRIPS_LIB := small_gicp_amd/lib/libsmall_gicp_amd_trips.so
trips:
	@mkdir -p build/obj_trips
	for f in $(SRCS); do $(HIPCC) $(HIPFLAGS) -DSGA_KD_TRIPS $(TRIPS_FLAGS) -c $$f -o build/obj_trips/$$(basename $$f .hip).o || exit 1; done
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $(TRIPS_LIB) build/obj_trips/*.o
oracle:
	$(MAKE) -C oracle
clean:
	rm -rf build $(LIB)
	$(MAKE) -C oracle clean
.PHONY: all lib oracle clean trips
# experiment builds: the library with other compile-time settings, e.g.  make variant V=w6 VFLAGS=-DSGA_SEARCH_WAVES=6
variant:
	@mkdir -p build/obj_$(V) small_gicp_amd/lib
	for f in $(SRCS); do o=build/obj_$(V)/$$(basename $$f .hip).o; if [ $$(basename $$f) = linearize.hip ] || [ ! -f $$o ]; then $(HIPCC) $(HIPFLAGS) $(VFLAGS) -c $$f -o $$o || exit 1; fi; done
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o small_gicp_amd/lib/libsmall_gicp_amd_$(V).so build/obj_$(V)/*.o
