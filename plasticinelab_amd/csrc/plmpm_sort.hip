// Device radix sort of (Hilbert key, storage slot) pairs for the re-sort of the particle storage order
// (plmpm_capi.hip: resort_frame_t on one GPU, migrate_finish_t on slab engines).  Kept in its own translation unit:
// it is the only place that uses a library primitive -- rocPRIM's radix_sort_pairs, called directly (no CUB-compat
// layer) -- once per re-sort, off the substep hot path.
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

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

// exclusive prefix sum over the per-cell particle counts of the counting-sort flavour of the re-sort
extern "C" size_t plmpm_scan_temp_bytes(size_t n) {
    size_t bytes = 0;
    (void)rocprim::exclusive_scan(nullptr, bytes, (const unsigned*)nullptr, (unsigned*)nullptr, 0u, n, rocprim::plus<unsigned>(), (hipStream_t) nullptr);
    return bytes;
}
extern "C" int plmpm_exclusive_scan(void* tmp, size_t bytes, const unsigned* in, unsigned* out, size_t n, void* stream) {
    return (int)rocprim::exclusive_scan(tmp, bytes, in, out, 0u, n, rocprim::plus<unsigned>(), (hipStream_t)stream);
}
