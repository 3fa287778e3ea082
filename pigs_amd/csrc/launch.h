// Host-side argument block shared by the launchers behind the C ABI (include/pigs_amd.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pigs_amd.h"

namespace pigs {

struct SampleArgs {
    int dtype, d, c, orders_mask;
    int64_t N, M;
    const void *means, *conics, *values, *samples;
    void* out[4];          // forward outputs (orders 0..3)
    const void* gout[4];   // backward: incoming gradients
    void *g_means, *g_conics, *g_values;
};

int dense_dispatch(bool backward, const SampleArgs& a, hipStream_t stream);

}  // namespace pigs
