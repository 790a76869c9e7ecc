// Internal (not exported): exclusive scan of int32 flags with a device-side live length.
#pragma once
#include <hip/hip_runtime.h>

// number of 2048-item chunks (= ints of chunk_sums scratch) for up to n_max items
int aq3d_scan_chunks_for(int n_max);

// prefix[i] = sum(flags[0..i)) for i < min(*count * mult, n_max); *total = sum of all live flags.
int aq3d_scan_flags(const int *flags, const int *count, int mult, int n_max, int *prefix, int *total,
                    int *chunk_sums, hipStream_t stream);
