#include "common.hpp"
using namespace sga;
extern "C" {
int sga_voxelgrid_sampling(sga_context*, const sga_cloud*, double, sga_cloud**) { return fail(SGA_ERR_UNSUPPORTED, "not built yet"); }
int sga_estimate_normals_covariances(sga_context*, sga_cloud*, const sga_index*, int, int) { return fail(SGA_ERR_UNSUPPORTED, "not built yet"); }
}
