// Key-value sort on the context's stream with the context's scratch (rocPRIM).
// rocPRIM's default sorts up to 2^20 items with a MERGE sort (block sort + log2(n / 1024) merge passes of two launches each) whatever the
// key width, larger inputs with its onesweep radix sort.  Forcing the radix sort for smaller inputs (SGA_SWEEP_MIN = the size from which
// the limit is lowered) was measured in round 6 and is OFF: a LiDAR scan's 115k keys 115 us against 54 us (five passes + seven state
// fills against ten launches); the 1M-point kd build 3.35 against 3.41 ms, the 1M-point source sort 484 against 443 us — no better.
#pragma once
#include <rocprim/rocprim.hpp>

#include "common.hpp"

namespace sga {
int ensure_temp(sga_context* ctx, size_t bytes);

template <typename Key, typename Val>
int sort_pairs(sga_context* ctx, Key* keys_in, Key* keys_out, Val* vals_in, Val* vals_out, size_t n, unsigned begin_bit, unsigned end_bit) {
  static const size_t sweep_min = getenv("SGA_SWEEP_MIN") ? static_cast<size_t>(atoll(getenv("SGA_SWEEP_MIN"))) : ~static_cast<size_t>(0);
  using Sweep = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 16384>;
  // a LiDAR scan's keys: block sorts of 2048 items (512 x 4) instead of the default 1024 — one merge pass less; same (stable) result
  // (scripts/ubench/sort_small.hip: 115k pairs 55.0 -> 48.7 us, 30k 38.2 -> 34.1; at 262k the default wins again)
  using Small = rocprim::radix_sort_config<rocprim::default_config, rocprim::merge_sort_config<512, 512, 4>, rocprim::default_config, 1 << 20>;
  size_t tb = 0;
  if (n > 2048 && n <= 200000 && n < sweep_min) {
    SGA_HIP(rocprim::radix_sort_pairs<Small>(nullptr, tb, keys_in, keys_out, vals_in, vals_out, n, begin_bit, end_bit, ctx->stream));
    SGA_TRY(ensure_temp(ctx, tb));
    SGA_HIP(rocprim::radix_sort_pairs<Small>(ctx->d_temp.p, tb, keys_in, keys_out, vals_in, vals_out, n, begin_bit, end_bit, ctx->stream));
  } else if (n >= sweep_min) {
    SGA_HIP(rocprim::radix_sort_pairs<Sweep>(nullptr, tb, keys_in, keys_out, vals_in, vals_out, n, begin_bit, end_bit, ctx->stream));
    SGA_TRY(ensure_temp(ctx, tb));
    SGA_HIP(rocprim::radix_sort_pairs<Sweep>(ctx->d_temp.p, tb, keys_in, keys_out, vals_in, vals_out, n, begin_bit, end_bit, ctx->stream));
  } else {
    SGA_HIP(rocprim::radix_sort_pairs(nullptr, tb, keys_in, keys_out, vals_in, vals_out, n, begin_bit, end_bit, ctx->stream));
    SGA_TRY(ensure_temp(ctx, tb));
    SGA_HIP(rocprim::radix_sort_pairs(ctx->d_temp.p, tb, keys_in, keys_out, vals_in, vals_out, n, begin_bit, end_bit, ctx->stream));
  }
  return SGA_OK;
}
}  // namespace sga
