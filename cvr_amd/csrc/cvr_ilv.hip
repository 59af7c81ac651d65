// cvr_ilv.hip -- CSR -> CVR64 for INTERLEAVED images (cvr_options.interleave; DESIGN.md section 3, "interleaved chunks").
//
// The reference hands a lane one row at a time (spmv.cpp:821-868), so at every step the 64 lanes of a chunk gather x at the columns of
// 64 different rows: on scattered columns that is one L1->L2 request per non-zero, which is what bounds the power-law shapes
// (profiles/r03_locality_livejournal.txt: 70.1 M requests for 69.0 M non-zeros).  An interleaved chunk keeps the chunk = consecutive
// rows, equal-length lane streams, row sums in LDS, but deals its non-zeros to the lanes in COLUMN order: element e of the chunk's
// column-sorted list (ties: by row) stands at step e / 64, lane e % 64, so one gather instruction reads 64 column-sorted neighbours and
// lanes share 128-byte lines of x (the requests-per-non-zero model and the prototype: profiles/r04_request_model.log,
// r04_sorted_prototype_lj.log).  Every slot is a piece of its own: its column word carries the end flag and the row inside the chunk (or
// the 16-bit tag does), which is the column-phase format in the limit of one column per phase and pieces of one element -- the image
// runs through spmv_seg_kernel unchanged (per row the products are added in column order, as the CSR loop of spmv.cpp:1843-1850 does).
//   slots e <  n (the chunk's non-zeros)     : column | end flag [| row << col_bits], value / code, [tag = row]
//   slots e >= n (padding up to 64 S; the pad slots the planner counts for empty rows among them): pad column (x_ext[ncols] = 0),
//                                              value 0, row = the dump entry behind the chunk's rows
// Preprocessing: a stable segmented radix sort of (column, position) over the chunks' element ranges (hipCUB), then one pass that writes
// the groups.
#include "cvr_kernels.h"

#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <type_traits>

namespace cvr {
namespace {

__global__ __launch_bounds__(256) void ilv_prepare_kernel(const int64_t *__restrict__ nzb, uint32_t nchunks, uint32_t *__restrict__ off32, int64_t n0, int64_t n1,
                                                          uint32_t *__restrict__ idx, uint2 *__restrict__ desc2)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nt = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = t; i <= (int64_t)nchunks; i += nt) off32[i] = (uint32_t)(nzb[i] - n0);
    for (int64_t i = t; i < (int64_t)nchunks; i += nt) desc2[i].x = (uint32_t)((nzb[i + 1] - nzb[i] + 255) / 256);      // the groups that hold non-zeros (the SpMV kernel stops there)
    for (int64_t j = n0 + t; j < n1; j += nt) idx[j - n0] = (uint32_t)(j - n0);
}

// one workgroup per group of 256 slots: thread t writes the slot of lane t & 63 at step t >> 6 of the group
template <typename T, bool DICT, bool TAG>
__global__ __launch_bounds__(256) void ilv_emit_kernel(uint8_t *__restrict__ stream, const uint4 *__restrict__ desc, const uint2 *__restrict__ desc2, const int64_t *__restrict__ nzb,
                                                       const int64_t *__restrict__ rp, const uint32_t *__restrict__ skey, const uint32_t *__restrict__ sidx, const T *__restrict__ vals,
                                                       const uint8_t *__restrict__ codes, const T *__restrict__ dict, uint32_t ndict, int G, int64_t n0, uint32_t pad_col, uint32_t col_bits,
                                                       uint32_t col_base, uint32_t *__restrict__ err)
{
    constexpr uint32_t GB = (DICT ? kGroupBytesDict : sizeof(T) == 8 ? kGroupBytes64 : kGroupBytes32) + (TAG ? kTagBytes : 0);
    constexpr uint32_t VB = kColsBytes + (TAG ? kTagBytes : 0);
    const uint32_t k = blockIdx.x / (uint32_t)G, g = blockIdx.x - k * (uint32_t)G;
    const uint32_t lane = threadIdx.x & 63u, j = threadIdx.x >> 6;
    const int64_t  b = nzb[k], n = nzb[k + 1] - b;
    const int64_t  e = (int64_t)g * 256 + (int64_t)j * 64 + lane;
    const uint32_t row_first = desc[k].x, nri = desc2[k].y;
    uint32_t col = pad_col, row = nri;
    T        v = T(0);
    uint32_t code = 0;
    if (e < n) {
        const int64_t p = n0 + (int64_t)sidx[b - n0 + e];         // position in the part's CSR arrays
        col = skey[b - n0 + e] - col_base;                         // (a column panel keeps its columns relative to its first)
        // the chunk's row of position p: the last of its rows that starts at or before p (a row cut over chunks begins before the chunk)
        uint32_t lo = 0, hi = nri;                                // answer in [0, nri)
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (rp[row_first + mid] <= p) lo = mid; else hi = mid;
        }
        row = lo;
        if constexpr (DICT) {
            if (codes) code = codes[p];
            else {
                const T   val = vals[p];
                uint32_t  a = 0, z = ndict;                        // the dictionary is sorted by bit pattern
                typedef typename std::conditional<sizeof(T) == 8, uint64_t, uint32_t>::type U;
                const U   bits = __builtin_bit_cast(U, val);
                while (a < z) { const uint32_t m = (a + z) >> 1; if (__builtin_bit_cast(U, dict[m]) < bits) a = m + 1; else z = m; }
                if (a >= ndict || __builtin_bit_cast(U, dict[a]) != bits) { atomicOr(err, 4u); a = 0; }
                code = a;
            }
        } else v = vals[p];
    } else if constexpr (DICT) {
        // the code of +0.0 (cvr_create puts it into every dictionary)
        uint32_t a = 0;
        typedef typename std::conditional<sizeof(T) == 8, uint64_t, uint32_t>::type U;
        while (a < ndict && __builtin_bit_cast(U, dict[a]) != (U)0) a++;
        if (a >= ndict) { atomicOr(err, 4u); a = 0; }
        code = a;
    }
    uint8_t *grp = stream + ((size_t)k * G + g) * GB;
    uint32_t cw = col | kEndBit;
    if constexpr (!TAG) cw |= row << col_bits;
    reinterpret_cast<uint32_t *>(grp)[lane * 4 + j] = cw;
    if constexpr (TAG) reinterpret_cast<uint16_t *>(grp + kColsBytes)[lane * 4 + j] = (uint16_t)row;
    if constexpr (DICT) (grp + VB)[lane * 4 + j] = (uint8_t)code;
    else if constexpr (sizeof(T) == 8) reinterpret_cast<double *>(grp + VB + (j >> 1) * (kLanes * 16))[lane * 2 + (j & 1)] = v;
    else reinterpret_cast<float *>(grp + VB)[lane * 4 + j] = v;
}

}  // namespace

size_t convert_interleaved_scratch(int64_t nnz, uint32_t nchunks)
{
    size_t tmp = 0;
    (void)hipcub::DeviceSegmentedRadixSort::SortPairs(nullptr, tmp, (const uint32_t *)nullptr, (uint32_t *)nullptr, (const uint32_t *)nullptr, (uint32_t *)nullptr,
                                                      (int)std::max<int64_t>(nnz, 1), (int)std::max<uint32_t>(nchunks, 1), (const uint32_t *)nullptr, (const uint32_t *)nullptr, 0, 32, nullptr);
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    return up(tmp) + 3 * up(sizeof(uint32_t) * (size_t)std::max<int64_t>(nnz, 1)) + up(sizeof(uint32_t) * ((size_t)nchunks + 1)) + 256;
}

// scratch: convert_interleaved_scratch(nnz of the part, nchunks) bytes of device memory
hipError_t launch_convert_interleaved(const DeviceImage &img, const DeviceCsr &csr, int64_t n0, int64_t n1, uint32_t *err_flag, void *scratch, size_t scratch_bytes, hipStream_t st)
{
    if (img.nchunks == 0) return hipSuccess;
    const int64_t nnz = n1 - n0;
    if (nnz >= (int64_t)0x7fffffff) return hipErrorInvalidValue;
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    size_t tmp = 0;
    hipError_t e = hipcub::DeviceSegmentedRadixSort::SortPairs(nullptr, tmp, (const uint32_t *)nullptr, (uint32_t *)nullptr, (const uint32_t *)nullptr, (uint32_t *)nullptr,
                                                               (int)std::max<int64_t>(nnz, 1), (int)img.nchunks, (const uint32_t *)nullptr, (const uint32_t *)nullptr, 0, 32, st);
    if (e != hipSuccess) return e;
    const size_t nn = up(sizeof(uint32_t) * (size_t)std::max<int64_t>(nnz, 1));
    if (up(tmp) + 3 * nn + up(sizeof(uint32_t) * ((size_t)img.nchunks + 1)) > scratch_bytes) return hipErrorInvalidValue;
    uint8_t  *a = static_cast<uint8_t *>(scratch);
    void     *d_tmp = a;
    uint32_t *idx = reinterpret_cast<uint32_t *>(a + up(tmp)), *skey = reinterpret_cast<uint32_t *>(a + up(tmp) + nn), *sidx = reinterpret_cast<uint32_t *>(a + up(tmp) + 2 * nn),
             *off32 = reinterpret_cast<uint32_t *>(a + up(tmp) + 3 * nn);
    const uint32_t pb = (uint32_t)std::min<int64_t>(4096, (std::max<int64_t>(nnz, (int64_t)img.nchunks + 1) + 255) / 256);
    hipLaunchKernelGGL(ilv_prepare_kernel, dim3(std::max(pb, 1u)), dim3(256), 0, st, csr.nz_begin, img.nchunks, off32, (long long)n0, (long long)n1, idx, img.desc2);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    if (nnz > 0) {
        int bits = 1;
        while (bits < 32 && (1ull << bits) <= (unsigned long long)img.col_base + img.pad_col) bits++;
        e = hipcub::DeviceSegmentedRadixSort::SortPairs(d_tmp, tmp, reinterpret_cast<const uint32_t *>(csr.col_idx + n0), skey, idx, sidx, (int)nnz, (int)img.nchunks, off32, off32 + 1, 0, bits, st);
        if (e != hipSuccess) return e;
    }
    const dim3 grid(img.nchunks * (uint32_t)img.G), block(256);
    const bool dict = img.dict != nullptr;
#define CVR_ILV(T, DI, TG)                                                                                                                           \
    hipLaunchKernelGGL((ilv_emit_kernel<T, DI, TG>), grid, block, 0, st, img.stream, img.desc, img.desc2, csr.nz_begin, csr.row_ptr, skey, sidx, static_cast<const T *>(csr.vals), \
                       csr.codes, static_cast<const T *>(img.dict), img.ndict, img.G, (long long)n0, img.pad_col, img.col_bits, img.col_base, err_flag)
    if (img.f32) { if (dict) { if (img.tag16) CVR_ILV(float, true, true); else CVR_ILV(float, true, false); } else { if (img.tag16) CVR_ILV(float, false, true); else CVR_ILV(float, false, false); } }
    else { if (dict) { if (img.tag16) CVR_ILV(double, true, true); else CVR_ILV(double, true, false); } else { if (img.tag16) CVR_ILV(double, false, true); else CVR_ILV(double, false, false); } }
#undef CVR_ILV
    return hipGetLastError();
}

}  // namespace cvr
