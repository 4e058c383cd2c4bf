// Micro-benchmark (round 6): rocPRIM's key-value sort at the size of a LiDAR scan's voxel keys (115k x 28 bits) under other merge-sort
// configurations than the library default (1024 items per block sort -> 7 merge passes).  hipcc -O3 --offload-arch=gfx950 sort_small.hip
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <class Config>
float run(const char* name, unsigned* k_in, unsigned* k_out, unsigned* v_in, unsigned* v_out, size_t n, unsigned bits, std::vector<unsigned>* ref_k, std::vector<unsigned>* ref_v) {
  size_t tb = 0;
  CK(rocprim::radix_sort_pairs<Config>(nullptr, tb, k_in, k_out, v_in, v_out, n, 0, bits, 0));
  void* tmp = nullptr;
  CK(hipMalloc(&tmp, tb + 256));
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  for (int w = 0; w < 5; w++) CK(rocprim::radix_sort_pairs<Config>(tmp, tb, k_in, k_out, v_in, v_out, n, 0, bits, 0));
  CK(hipDeviceSynchronize());
  float best = 1e9f;
  for (int rep = 0; rep < 20; rep++) {
    CK(hipEventRecord(a, 0));
    CK(rocprim::radix_sort_pairs<Config>(tmp, tb, k_in, k_out, v_in, v_out, n, 0, bits, 0));
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    best = ms < best ? ms : best;
  }
  std::vector<unsigned> hk(n), hv(n);
  CK(hipMemcpy(hk.data(), k_out, n * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(hv.data(), v_out, n * 4, hipMemcpyDeviceToHost));
  bool same = true;
  if (ref_k->empty()) {
    *ref_k = hk;
    *ref_v = hv;
  } else {
    same = hk == *ref_k && hv == *ref_v;
  }
  printf("%-44s n=%zu  %.1f us (best of 20)  %s\n", name, n, best * 1e3f, same ? "== default" : "DIFFERS");
  CK(hipFree(tmp));
  return best;
}

int main(int argc, char** argv) {
  for (size_t n : {115884ul, 30000ul, 262144ul}) {
    const unsigned bits = 28;
    std::vector<unsigned> k(n), v(n);
    unsigned s = 12345u;
    for (size_t i = 0; i < n; i++) {
      s = s * 1664525u + 1013904223u;
      k[i] = (s >> 4) & ((1u << bits) - 1u) & ~0xffu;  // many equal keys (runs), like voxel keys
      v[i] = static_cast<unsigned>(i);
    }
    unsigned *k_in, *k_out, *v_in, *v_out;
    CK(hipMalloc(&k_in, n * 4)); CK(hipMalloc(&k_out, n * 4)); CK(hipMalloc(&v_in, n * 4)); CK(hipMalloc(&v_out, n * 4));
    CK(hipMemcpy(k_in, k.data(), n * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(v_in, v.data(), n * 4, hipMemcpyHostToDevice));
    std::vector<unsigned> rk, rv;
    using namespace rocprim;
    run<default_config>("default", k_in, k_out, v_in, v_out, n, bits, &rk, &rv);
    run<radix_sort_config<default_config, merge_sort_config<512, 512, 4>, default_config, 1 << 20>>("merge<512, 512x4 = 2048>", k_in, k_out, v_in, v_out, n, bits, &rk, &rv);
    run<radix_sort_config<default_config, merge_sort_config<512, 512, 8>, default_config, 1 << 20>>("merge<512, 512x8 = 4096>", k_in, k_out, v_in, v_out, n, bits, &rk, &rv);
    run<radix_sort_config<default_config, merge_sort_config<512, 512, 16>, default_config, 1 << 20>>("merge<512, 512x16 = 8192>", k_in, k_out, v_in, v_out, n, bits, &rk, &rv);
    run<radix_sort_config<default_config, merge_sort_config<1024, 1024, 8>, default_config, 1 << 20>>("merge<1024, 1024x8 = 8192>", k_in, k_out, v_in, v_out, n, bits, &rk, &rv);
    run<radix_sort_config<default_config, merge_sort_config<256, 256, 16>, default_config, 1 << 20>>("merge<256, 256x16 = 4096>", k_in, k_out, v_in, v_out, n, bits, &rk, &rv);
    run<radix_sort_config<default_config, merge_sort_config<512, 512, 8, 128, 256, 8, 1>, default_config, 1 << 20>>("merge<512x8, mergepath always 256x8>", k_in, k_out, v_in, v_out, n, bits, &rk, &rv);
    run<radix_sort_config<default_config, merge_sort_config<512, 512, 16, 128, 256, 8, 1>, default_config, 1 << 20>>("merge<512x16, mergepath always 256x8>", k_in, k_out, v_in, v_out, n, bits, &rk, &rv);
    run<radix_sort_config<default_config, default_config, default_config, 16384>>("onesweep", k_in, k_out, v_in, v_out, n, bits, &rk, &rv);
    CK(hipFree(k_in)); CK(hipFree(k_out)); CK(hipFree(v_in)); CK(hipFree(v_out));
  }
  return 0;
}
