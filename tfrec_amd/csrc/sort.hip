// Stable LSD radix sort of (key, value) pairs on the device.  The duplicate-index reduction of
// the training steps is built on it: occurrences sorted by row id become contiguous segments.
// rocPRIM's device radix sort is used for this index-only step (8 bytes per element and pass);
// the row traffic of a step is two orders of magnitude larger.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "common.hpp"

size_t tfrec_sort_temp_bytes(size_t n, int end_bit)
{
    size_t bytes = 0;
    uint32_t *k = nullptr, *v = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, k, k, v, v, n, 0, (unsigned)end_bit, (hipStream_t) nullptr);
    return bytes ? bytes : 256;
}

int tfrec_sort_pairs(void *d_temp, size_t temp_bytes, const uint32_t *keys_in, uint32_t *keys_out,
                     const uint32_t *vals_in, uint32_t *vals_out, size_t n, int end_bit, hipStream_t stream)
{
    hipError_t e = rocprim::radix_sort_pairs(d_temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, n, 0,
                                             (unsigned)end_bit, stream);
    if (e != hipSuccess) {
        tfrec_set_error("radix_sort_pairs failed: %s", hipGetErrorString(e));
        return TFREC_EHIP;
    }
    return TFREC_OK;
}
