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

constexpr int kIlvMaxParts = 64;
// what the kernels need of one image; the images of a handle (column panels) are converted together: one sort, one pass that writes
struct IlvPartDev {
    uint8_t       *stream;
    const uint4   *desc;
    uint2         *desc2;
    const int64_t *nzb, *rp;
    const int32_t *ci;
    const void    *vals;
    int64_t        n0;          // the part's first position in its CSR arrays
    int64_t        e0;          // where its elements start in the concatenated key array
    uint32_t       nchunks, chunk0;      // its chunks are numbered chunk0 .. chunk0 + nchunks - 1 over all parts
    uint32_t       pad_col, col_base;
};
struct IlvTable { IlvPartDev part[kIlvMaxParts]; int64_t e_end[kIlvMaxParts]; uint32_t chunk_end[kIlvMaxParts]; uint32_t nparts; };

__device__ __forceinline__ uint32_t part_of_chunk(const IlvTable *__restrict__ t, uint32_t kk)
{
    uint32_t p = 0;
    while (p + 1 < t->nparts && t->chunk_end[p] <= kk) p++;
    return p;
}

// key = chunk (numbered over all parts) << cbits | column inside the image (relative to col_base), value = position in the concatenation:
// one stable radix sort over these keys sorts every chunk's non-zeros by column (ties: by position, i.e. by row), the chunks staying in
// place.  One workgroup per chunk (its part and number are the workgroup's: no search per element).  Also desc2[k].x = the groups of
// chunk k that hold non-zeros (the SpMV kernel stops there).
template <typename K>
__global__ __launch_bounds__(256) void ilv_keys_kernel(const IlvTable *__restrict__ t, uint32_t cbits, K *__restrict__ key, uint32_t *__restrict__ idx)
{
    const uint32_t    kk = blockIdx.x;
    const IlvPartDev &q = t->part[part_of_chunk(t, kk)];
    const uint32_t    k = kk - q.chunk0;
    const int64_t     b = q.nzb[k], n = q.nzb[k + 1] - b, E0 = q.e0 + (b - q.n0);
    if (threadIdx.x == 0) q.desc2[k].x = (uint32_t)((n + 255) / 256);
    const K hi = (K)kk << cbits;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
        key[E0 + i] = hi | (K)((uint32_t)q.ci[b + i] - q.col_base);
        idx[E0 + i] = (uint32_t)(E0 + i);
    }
}

// One workgroup per chunk: the starts of the chunk's rows (relative to its first position, clipped to the chunk) go to LDS once, then the
// threads walk the chunk's slots -- thread t of a group of 256 slots writes lane t & 63 at step t >> 6 -- and find the row of an element
// by a search in LDS.
constexpr int kEmitThreads = 1024;
template <typename T, bool DICT, bool TAG, typename K>
__global__ __launch_bounds__(kEmitThreads) void ilv_emit_kernel(const IlvTable *__restrict__ t, const K *__restrict__ skey, const uint32_t *__restrict__ sidx, const T *__restrict__ dict,
                                                                uint32_t ndict, int G, uint32_t col_bits, uint32_t cbits, uint32_t *__restrict__ err)
{
    constexpr uint32_t GB = (DICT ? kGroupBytesDict : sizeof(T) == 8 ? kGroupBytes64 : kGroupBytes32) + (TAG ? kTagBytes : 0);
    constexpr uint32_t VB = kColsBytes + (TAG ? kTagBytes : 0);
    extern __shared__ uint32_t rstart[];                         // [nri]: where row i of the chunk starts among the chunk's elements
    const uint32_t    kk = blockIdx.x;
    const IlvPartDev &q = t->part[part_of_chunk(t, kk)];
    const uint32_t    k = kk - q.chunk0;
    const int64_t     b = q.nzb[k], n = q.nzb[k + 1] - b;
    const uint32_t    row_first = q.desc[k].x, nri = q.desc2[k].y;
    const int64_t     E0 = q.e0 + (b - q.n0);                    // the chunk's first element in the sorted arrays
    for (uint32_t i = threadIdx.x; i < nri; i += blockDim.x) {
        const int64_t r = q.rp[row_first + i] - b;               // (a row cut over chunks begins before the chunk)
        rstart[i] = (uint32_t)(r < 0 ? 0 : r > n ? n : r);
    }
    __syncthreads();
    uint32_t code0 = 0;                                           // the code of +0.0 (cvr_create puts it into every dictionary)
    if constexpr (DICT) {
        typedef typename std::conditional<sizeof(T) == 8, uint64_t, uint32_t>::type U;
        while (code0 < ndict && __builtin_bit_cast(U, dict[code0]) != (U)0) code0++;
        if (code0 >= ndict) { if (threadIdx.x == 0) atomicOr(err, 4u); code0 = 0; }
    }
    const int64_t nslots = (int64_t)G * 256;
    for (int64_t e = threadIdx.x; e < nslots; e += blockDim.x) {
        const uint32_t g = (uint32_t)(e >> 8), j = (uint32_t)(e >> 6) & 3u, lane = (uint32_t)e & 63u;
        uint32_t col = q.pad_col, row = nri, code = code0;
        T        v = T(0);
        if (e < n) {
            const uint32_t pe = (uint32_t)((int64_t)sidx[E0 + e] - E0);      // the element's place among the chunk's, in CSR order
            col = (uint32_t)(skey[E0 + e] & (((K)1 << cbits) - 1));          // (relative to the image's first column)
            uint32_t lo = 0, hi = nri;                                        // its row: the last one that starts at or before it
            while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (rstart[mid] <= pe) lo = mid; else hi = mid; }
            row = lo;
            const T val = static_cast<const T *>(q.vals)[b + pe];
            if constexpr (DICT) {
                uint32_t a = 0, z = ndict;                                    // the dictionary is sorted by bit pattern
                typedef typename std::conditional<sizeof(T) == 8, uint64_t, uint32_t>::type U;
                const U  bits = __builtin_bit_cast(U, val);
                while (a < z) { const uint32_t m = (a + z) >> 1; if (__builtin_bit_cast(U, dict[m]) < bits) a = m + 1; else z = m; }
                if (a >= ndict || __builtin_bit_cast(U, dict[a]) != bits) { atomicOr(err, 4u); a = 0; }
                code = a;
            } else v = val;
        }
        uint8_t *grp = q.stream + ((size_t)k * G + g) * GB;
        uint32_t cw = col | kEndBit;
        if constexpr (!TAG) cw |= row << col_bits;
        reinterpret_cast<uint32_t *>(grp)[lane * 4 + j] = cw;
        if constexpr (TAG) reinterpret_cast<uint16_t *>(grp + kColsBytes)[lane * 4 + j] = (uint16_t)row;
        if constexpr (DICT) (grp + VB)[lane * 4 + j] = (uint8_t)code;
        else if constexpr (sizeof(T) == 8) reinterpret_cast<double *>(grp + VB + (j >> 1) * (kLanes * 16))[lane * 2 + (j & 1)] = v;
        else reinterpret_cast<float *>(grp + VB)[lane * 4 + j] = v;
    }
}

inline size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }
inline uint32_t bits_of(uint64_t v) { uint32_t b = 1; while (b < 63 && ((uint64_t)1 << b) <= v) b++; return b; }      // bits that hold 0 .. v

template <typename K> size_t sort_bytes(int64_t nnz)
{
    size_t tmp = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, tmp, (const K *)nullptr, (K *)nullptr, (const uint32_t *)nullptr, (uint32_t *)nullptr, (unsigned int)std::max<int64_t>(nnz, 1), 0, (int)sizeof(K) * 8, nullptr);
    return tmp;
}

template <typename K>
hipError_t convert_t(const IlvTable &tab, int64_t ntot, uint32_t nchunks_tot, const DeviceImage &common, uint32_t ystage_max, uint32_t cbits, uint32_t kbits, uint32_t *err_flag, uint8_t *a, size_t scratch_bytes, hipStream_t st)
{
    const size_t nk = up256(sizeof(K) * (size_t)std::max<int64_t>(ntot, 1)), nv = up256(sizeof(uint32_t) * (size_t)std::max<int64_t>(ntot, 1));
    size_t       tmp = sort_bytes<K>(ntot);
    if (up256(sizeof(IlvTable)) + up256(tmp) + 2 * nk + 2 * nv > scratch_bytes) return hipErrorInvalidValue;
    IlvTable *d_tab = reinterpret_cast<IlvTable *>(a);
    a += up256(sizeof(IlvTable));
    void     *d_tmp = a;
    K        *key = reinterpret_cast<K *>(a + up256(tmp)), *skey = reinterpret_cast<K *>(a + up256(tmp) + nk);
    uint32_t *idx = reinterpret_cast<uint32_t *>(a + up256(tmp) + 2 * nk), *sidx = reinterpret_cast<uint32_t *>(a + up256(tmp) + 2 * nk + nv);
    hipError_t e = hipMemcpyAsync(d_tab, &tab, sizeof(IlvTable), hipMemcpyHostToDevice, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(ilv_keys_kernel<K>, dim3(nchunks_tot), dim3(256), 0, st, d_tab, cbits, key, idx);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    if (ntot > 0) {
        e = hipcub::DeviceRadixSort::SortPairs(d_tmp, tmp, key, skey, idx, sidx, (unsigned int)ntot, 0, (int)(cbits + kbits), st);      // stable: ties keep their positions' order
        if (e != hipSuccess) return e;
    }
    const dim3   grid(nchunks_tot), block(kEmitThreads);
    const size_t lds = sizeof(uint32_t) * (size_t)std::max<uint32_t>(ystage_max, 1u);      // (the rows of a chunk fit its accumulators: ystage - 1 at most)
    const bool   dict = common.dict != nullptr;
#define CVR_ILV(T, DI, TG)                                                                                                                           \
    hipLaunchKernelGGL((ilv_emit_kernel<T, DI, TG, K>), grid, block, lds, st, d_tab, skey, sidx, static_cast<const T *>(common.dict), common.ndict, common.G, common.col_bits, cbits, err_flag)
    if (common.f32) { if (dict) { if (common.tag16) CVR_ILV(float, true, true); else CVR_ILV(float, true, false); } else { if (common.tag16) CVR_ILV(float, false, true); else CVR_ILV(float, false, false); } }
    else { if (dict) { if (common.tag16) CVR_ILV(double, true, true); else CVR_ILV(double, true, false); } else { if (common.tag16) CVR_ILV(double, false, true); else CVR_ILV(double, false, false); } }
#undef CVR_ILV
    return hipGetLastError();
}
}  // namespace

// device scratch of a conversion of `nnz` non-zeros in all (room for either key width: which one is decided at conversion time)
size_t convert_interleaved_scratch(int64_t nnz, uint32_t nchunks)
{
    (void)nchunks;
    const size_t n = (size_t)std::max<int64_t>(nnz, 1);
    return up256(sizeof(IlvTable)) + up256(std::max(sort_bytes<uint32_t>(nnz), sort_bytes<uint64_t>(nnz))) + 2 * up256(sizeof(uint64_t) * n) + 2 * up256(sizeof(uint32_t) * n) + 256;
}

// The interleaved images imgs[0 .. n) (all of one handle: same chunk length, value type, dictionary, tag width and row field) converted
// together -- one sort over all their non-zeros, one pass that writes all their groups; csrs[i], [n0[i], n1[i]) = image i's CSR arrays and
// positions.  scratch: convert_interleaved_scratch(sum of the non-zeros, ..) bytes.  *err_flag bit 2: a value that is not in the dictionary.
hipError_t launch_convert_interleaved(const DeviceImage *const *imgs, const DeviceCsr *csrs, const int64_t *n0, const int64_t *n1, int n, uint32_t *err_flag, void *scratch,
                                      size_t scratch_bytes, hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    if (n > kIlvMaxParts) return hipErrorInvalidValue;
    IlvTable tab;
    memset(&tab, 0, sizeof(tab));
    int64_t  e = 0;
    uint32_t c = 0, pad_max = 0, ystage_max = 0;
    for (int i = 0; i < n; i++) {
        const DeviceImage &g = *imgs[i];
        if (g.G != imgs[0]->G || g.f32 != imgs[0]->f32 || g.dict != imgs[0]->dict || g.tag16 != imgs[0]->tag16 || g.col_bits != imgs[0]->col_bits) return hipErrorInvalidValue;
        IlvPartDev &q = tab.part[i];
        q.stream = g.stream; q.desc = g.desc; q.desc2 = g.desc2; q.nzb = csrs[i].nz_begin; q.rp = csrs[i].row_ptr; q.ci = csrs[i].col_idx; q.vals = csrs[i].vals;
        q.n0 = n0[i]; q.e0 = e; q.nchunks = g.nchunks; q.chunk0 = c; q.pad_col = g.pad_col; q.col_base = g.col_base;
        e += n1[i] - n0[i]; c += g.nchunks;
        tab.e_end[i] = e; tab.chunk_end[i] = c;
        pad_max = std::max(pad_max, g.pad_col);
        ystage_max = std::max(ystage_max, g.ystage);
    }
    tab.nparts = (uint32_t)n;
    if (c == 0) return hipSuccess;
    if (e >= (int64_t)0xffffffffll) return hipErrorInvalidValue;        // positions in the concatenation are 32-bit
    const uint32_t cbits = bits_of(pad_max), kbits = bits_of(c - 1);
    if (cbits + kbits <= 32) return convert_t<uint32_t>(tab, e, c, *imgs[0], ystage_max, cbits, kbits, err_flag, static_cast<uint8_t *>(scratch), scratch_bytes, st);
    return convert_t<uint64_t>(tab, e, c, *imgs[0], ystage_max, cbits, kbits, err_flag, static_cast<uint8_t *>(scratch), scratch_bytes, st);
}

}  // namespace cvr
