// Device radix sort of (Hilbert key, storage slot) pairs for the re-sort of the particle storage order
// (plmpm_capi.hip: resort_frame_t on one GPU, migrate_finish_t on slab engines).  Kept in its own translation unit:
// it is the only place that uses a library primitive -- rocPRIM's radix_sort_pairs, called directly (no CUB-compat
// layer) -- once per re-sort, off the substep hot path.
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

extern "C" size_t plmpm_sort_temp_bytes(int n) {
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, (const unsigned*)nullptr, (unsigned*)nullptr, (const int*)nullptr, (int*)nullptr, (size_t)n, 0u, 32u,
                                    (hipStream_t) nullptr);
    return bytes;
}
// keys use bits [0, key_bits); padding keys have the top bit of that range set, so they sort last; stable
extern "C" int plmpm_sort_pairs(void* tmp, size_t bytes, const unsigned* kin, unsigned* kout, const int* vin, int* vout, int n, int key_bits,
                                void* stream) {
    return (int)rocprim::radix_sort_pairs(tmp, bytes, kin, kout, vin, vout, (size_t)n, 0u, (unsigned)key_bits, (hipStream_t)stream);
}
