// The device-wide primitives the library still takes from a library — radix sort and prefix sum — called through rocPRIM's own interface
// (r04: the hipCUB layer, CUB's interface over rocPRIM, is gone from csrc/).  The steady-state MI pass uses neither (selection without a
// sort: k_sel_*); what does: the general selection path of blocks without a bucket guess, the short-range model's per-(cluster, len)
// quantiles, the ARACNE adjacency (CSR by one sort), the long-range Tukey thresholds and the LD map's rank scan.
#pragma once
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

namespace ldw {

// storage == nullptr: size query (bytes is set); keys ascending over bits [begin_bit, end_bit), stable
template <class K, class V>
inline hipError_t prim_sort_pairs(void *storage, size_t &bytes, const K *kin, K *kout, const V *vin, V *vout, size_t n, unsigned begin_bit, unsigned end_bit,
                                  hipStream_t st) {
    return rocprim::radix_sort_pairs(storage, bytes, kin, kout, vin, vout, n, begin_bit, end_bit, st);
}
template <class K>
inline hipError_t prim_sort_keys(void *storage, size_t &bytes, const K *kin, K *kout, size_t n, unsigned begin_bit, unsigned end_bit, hipStream_t st) {
    return rocprim::radix_sort_keys(storage, bytes, kin, kout, n, begin_bit, end_bit, st);
}
template <class T>
inline hipError_t prim_exclusive_sum(void *storage, size_t &bytes, const T *in, T *out, size_t n, hipStream_t st) {
    return rocprim::exclusive_scan(storage, bytes, in, out, T(0), n, rocprim::plus<T>(), st);
}

}  // namespace ldw
