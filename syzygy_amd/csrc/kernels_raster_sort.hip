// kernels_raster_sort.hip — the one library call of the rasteriser: rocPRIM's device radix sort of the primitive
// sort keys (kernels_raster.hip k_raster_setup), kept in its own translation unit because of the header's weight.
// Temp storage is sized from the element count alone, so no host round trip is needed between setup and sort.
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "szg_launch.hpp"

namespace szg
{
hipError_t raster_sort_temp_bytes(unsigned n, size_t& bytes)
{
    bytes = 0;
    unsigned* none = nullptr;
    return rocprim::radix_sort_pairs(nullptr, bytes, none, none, none, none, n, 0, 32, hipStream_t{});
}

hipError_t raster_sort_pairs(hipStream_t s, void* temp, size_t tempBytes, const unsigned* keysIn, unsigned* keysOut,
                             const unsigned* valsIn, unsigned* valsOut, unsigned n)
{
    // stable LSD radix sort: equal keys keep the submission order
    return rocprim::radix_sort_pairs(temp, tempBytes, keysIn, keysOut, valsIn, valsOut, n, 0, 32, s);
}
} // namespace szg
