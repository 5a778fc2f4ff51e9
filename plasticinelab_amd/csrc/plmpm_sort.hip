// Device radix sort of (Hilbert key, storage slot) pairs for the per-env-step re-sort of the particle storage
// order (plmpm_capi.hip, resort_step).  Kept in its own translation unit: it is the only place that uses a library
// primitive (hipCUB / rocPRIM DeviceRadixSort), once per env step, off the substep hot path.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

extern "C" size_t plmpm_sort_temp_bytes(int n) {
    size_t bytes = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, (const unsigned*)nullptr, (unsigned*)nullptr, (const int*)nullptr,
                                             (int*)nullptr, n, 0, 32, (hipStream_t) nullptr);
    return bytes;
}
// keys use bits [0, key_bits); padding keys have every bit set, so they still sort last
extern "C" int plmpm_sort_pairs(void* tmp, size_t bytes, const unsigned* kin, unsigned* kout, const int* vin, int* vout, int n,
                                int key_bits, void* stream) {
    return (int)hipcub::DeviceRadixSort::SortPairs(tmp, bytes, kin, kout, vin, vout, n, 0, key_bits, (hipStream_t)stream);
}
