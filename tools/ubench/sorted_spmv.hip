// tools/ubench/sorted_spmv.hip -- prototype of the INTERLEAVED chunk layout (round 4): a chunk is a range of consecutive rows whose
// non-zeros are sorted by column and dealt to the 64 lanes step by step (element e -> step e / 64, lane e % 64), every slot carrying
// its row inside the chunk; the wavefront gathers x for 64 column-sorted neighbours per instruction (lanes share 128-byte lines: fewer
// L1->L2 requests than non-zeros) and adds every product into the row's accumulator in LDS (ds_add, lanes in order, steps in order).
// Measures what that buys before converter, planner, mirror and C ABI are changed for it.
//   hipcc --offload-arch=gfx950 -O3 -fopenmp sorted_spmv.hip -o sorted_spmv
//   sorted_spmv csr.bin f32|f64 R W Smax P mode depth iters [dict]
//     R rows per accumulator set, W wavefronts per workgroup, Smax steps per chunk at most, P column panels (dealt to the XCDs),
//     mode 0: a wavefront owns a chunk (reproducible), mode 1: the W wavefronts of a workgroup share one chunk of W * R rows
#include <hip/hip_runtime.h>
#include <omp.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef double   f64x2 __attribute__((ext_vector_type(2)));
typedef float    f32x4 __attribute__((ext_vector_type(4)));

struct ChunkDesc {
    uint64_t stream_off;      // bytes
    uint32_t G;               // groups of 4 steps
    uint32_t nrows;           // accumulators this chunk uses
    uint64_t zoff;            // where its sums go
    uint32_t xbase;           // first column of its panel
    uint32_t xcols;           // columns of its panel
    uint32_t nacc, pad_;      // mode 3: accumulators (rows + 3 per replicated row)
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *p, uint32_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, bytes, 0x00020000);
}

template <typename T, bool DICT> struct Grp;
template <> struct Grp<double, false> { u32x4 c; f64x2 lo, hi; u32x2 t; };
template <> struct Grp<float, false>  { u32x4 c; f32x4 v; u32x2 t; };
template <typename T> struct Grp<T, true> { u32x4 c; uint32_t codes; u32x2 t; };

template <typename T, bool DICT, bool TAG> constexpr uint32_t group_bytes() { return 1024u + (TAG ? 512u : 0u) + (DICT ? 256u : sizeof(T) == 8 ? 2048u : 1024u); }

template <typename T, bool DICT, bool TAG>
__device__ __forceinline__ Grp<T, DICT> load_grp(__amdgpu_buffer_rsrc_t r, uint32_t lane, uint32_t soff)
{
    Grp<T, DICT> g;
    const uint32_t voff = lane * 16;
    constexpr uint32_t VB = 1024u + (TAG ? 512u : 0u);
    g.c = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
    if constexpr (TAG) g.t = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(r, (voff >> 1) + 1024u, soff, 0));
    else g.t = u32x2{0, 0};
    if constexpr (DICT) g.codes = __builtin_amdgcn_raw_buffer_load_b32(r, (voff >> 2) + VB, soff, 0);
    else if constexpr (sizeof(T) == 8) {
        g.lo = __builtin_bit_cast(f64x2, __builtin_amdgcn_raw_buffer_load_b128(r, voff + VB, soff, 0));
        g.hi = __builtin_bit_cast(f64x2, __builtin_amdgcn_raw_buffer_load_b128(r, voff + VB + 1024u, soff, 0));
    } else g.v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff + VB, soff, 0));
    return g;
}

template <typename T> struct X4 { T v[4]; };

template <typename T>
__device__ __forceinline__ X4<T> gather4(__amdgpu_buffer_rsrc_t rx, const u32x4 c, uint32_t cmask)
{
    X4<T> r;
    const uint32_t col[4] = {c.x & cmask, c.y & cmask, c.z & cmask, c.w & cmask};
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if constexpr (sizeof(T) == 8) r.v[j] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rx, col[j] * 8u, 0, 0));
        else r.v[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, col[j] * 4u, 0, 0));
    }
    return r;
}

template <typename T, bool DICT>
__device__ __forceinline__ T val_of(const Grp<T, DICT> &g, int j, const T *dict)
{
    if constexpr (DICT) return dict[(g.codes >> (8 * j)) & 0xffu];
    else if constexpr (sizeof(T) == 8) return j == 0 ? g.lo.x : j == 1 ? g.lo.y : j == 2 ? g.hi.x : g.hi.y;
    else return j == 0 ? g.v.x : j == 1 ? g.v.y : j == 2 ? g.v.z : g.v.w;
}

// Loop-carried load results are only ever read through an empty asm on the whole vector: the compiler then keeps them as the register
// tuples the loads write (it had split them into scalars in other registers, with copies -- behind s_waitcnt vmcnt(0) -- at the back-edge)
template <typename T, bool DICT> __device__ __forceinline__ void pin(Grp<T, DICT> &g)
{
    asm volatile("" : "+v"(g.c));
    asm volatile("" : "+v"(g.t));
    if constexpr (DICT) asm volatile("" : "+v"(g.codes));
    else if constexpr (sizeof(T) == 8) { asm volatile("" : "+v"(g.lo)); asm volatile("" : "+v"(g.hi)); }
    else asm volatile("" : "+v"(g.v));
}
template <typename T> __device__ __forceinline__ void pin(X4<T> &x)
{
#pragma unroll
    for (int j = 0; j < 4; j++) asm volatile("" : "+v"(x.v[j]));
}

template <typename T, bool SHARED> __device__ __forceinline__ void lds_add(T *p, T v)
{
    if constexpr (SHARED) (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}

// wg_first[b], wg_count[b]: the chunks of workgroup b (mode 0: one per wavefront; mode 1: one chunk, all wavefronts)
template <typename T, bool DICT, bool TAG, int DEPTH, bool SHARED, int NOADD>
__global__ __launch_bounds__(1024) void sorted_spmv_kernel(const uint8_t *__restrict__ stream, const ChunkDesc *__restrict__ desc, const uint32_t *__restrict__ wg_first,
                                                           const uint32_t *__restrict__ wg_count, const T *__restrict__ x, T *__restrict__ z, uint32_t col_bits,
                                                           uint32_t R, const T *__restrict__ dict_g, uint32_t ndict)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr uint32_t GB = group_bytes<T, DICT, TAG>();
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6, nwv = blockDim.x >> 6;
    T *const dict = reinterpret_cast<T *>(smem);
    T *const acc_all = dict + (DICT ? 256 : 0);
    const uint32_t first = wg_first[blockIdx.x], cnt = wg_count[blockIdx.x];
    const uint32_t wv_u = __builtin_amdgcn_readfirstlane(wv);            // (wave-uniform: the descriptors below must live in SGPRs)
    if constexpr (DICT) for (uint32_t i = threadIdx.x; i < 256u; i += blockDim.x) dict[i] = i < ndict ? dict_g[i] : T(0);
    uint32_t k;
    bool     live;
    T       *acc;
    if constexpr (SHARED) { k = first; live = cnt > 0; acc = acc_all; }
    else { k = first + wv_u; live = wv_u < cnt; acc = acc_all + (size_t)wv_u * (R + 1); }
    ChunkDesc d = live ? desc[k] : ChunkDesc{0, 0, 0, 0, 0, 0, 0, 0};
    if constexpr (SHARED) { for (uint32_t i = threadIdx.x; i <= d.nrows; i += blockDim.x) acc[i] = T(0); }
    else if (live) for (uint32_t i = lane; i <= d.nrows; i += 64u) acc[i] = T(0);
    if constexpr (SHARED || DICT) __syncthreads();
    if (!live) return;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(stream + d.stream_off, d.G * GB);
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(x + d.xbase, d.xcols * (uint32_t)sizeof(T));
    const uint32_t cmask = col_bits >= 32 ? 0xffffffffu : (1u << col_bits) - 1u;
    const uint32_t g0 = SHARED ? wv_u : 0u, gs = SHARED ? nwv : 1u;       // shared chunk: the wavefronts take the groups in turn
    // software pipeline as a ring with compile-time slots (register moves of values still in flight would force vmcnt(0)):
    // Q[j] holds the stream group g with g % QN == j, xs[j] the gathered x of the group with g % DEPTH == j; QN % DEPTH == 0
    constexpr int QA = DEPTH, QN = DEPTH + QA;
    Grp<T, DICT> Q[QN];
    X4<T>        xs[DEPTH];
    // the ring starts empty (all-zero groups: column 0, tag 0 = the dump accumulator) and the loop runs QN groups early: no prologue
    // whose registers the loop's would have to be matched with (that match failed: a rotation of the whole ring behind vmcnt(0) per trip)
    memset(Q, 0, sizeof(Q));
    memset(xs, 0, sizeof(xs));
    T sink = 0;
    for (int32_t gb = (int32_t)g0 - (int32_t)(gs * QN); gb < (int32_t)d.G; gb += (int32_t)(gs * QN)) {
#pragma unroll
        for (int u = 0; u < QN; u++) {
            const uint32_t g = (uint32_t)(gb + (int32_t)((uint32_t)u * gs));      // (negative: the run-in, its stream offsets wrap far beyond the chunk: zeros)
            Grp<T, DICT> q = Q[u];
            X4<T>        xv = xs[u % DEPTH];
            pin(q); pin(xv);                        // (whole registers tuples across the back-edge: see pin())
            u32x4 cnext = Q[(u + DEPTH) % QN].c;
            asm volatile("" : "+v"(cnext));
            xs[u % DEPTH] = gather4<T>(rx, cnext, cmask);
            Q[u] = load_grp<T, DICT, TAG>(rs, lane, (g + (uint32_t)QN * gs) * GB);
            __builtin_amdgcn_sched_barrier(0);       // (the scheduler must not move loads over uses: the counted vmcnt waits follow program order)
            {
                T av[4];
#pragma unroll
                for (int j = 0; j < 4; j++) av[j] = val_of<T, DICT>(q, j, dict);
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const uint32_t cw = j == 0 ? q.c.x : j == 1 ? q.c.y : j == 2 ? q.c.z : q.c.w;
                    uint32_t row;
                    if constexpr (TAG) row = j == 0 ? q.t.x & 0xffffu : j == 1 ? q.t.x >> 16 : j == 2 ? q.t.y & 0xffffu : q.t.y >> 16;
                    else row = col_bits >= 32 ? 0u : cw >> col_bits;
                    const T prod = av[j] * xv.v[j];
                    if constexpr (NOADD == 0) lds_add<T, SHARED>(acc + row, prod);
                    else sink += prod + (T)row;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if constexpr (NOADD != 0) { if (sink == (T)12345.678) acc[0] = sink; }
    if constexpr (SHARED) {
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < d.nrows; i += blockDim.x) z[d.zoff + i] = acc[i + 1];
    } else {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (uint32_t i = lane; i < d.nrows; i += 64u) z[d.zoff + i] = acc[i + 1];
    }
}

// ---- the same kernel with a HAND-PIPELINED loop -------------------------------------------------------------------------------
// hipcc would not keep a ring of in-flight loads in fixed registers (it rotates the ring with register copies behind s_waitcnt
// vmcnt(0) at the loop's back-edge, or splits the loaded vectors), so the ring lives in registers the compiler does not allocate:
// the kernel is limited to v0..v95 (amdgpu_num_vgpr), the loads are issued by asm statements into v96..v255 and waited for with
// counted s_waitcnt vmcnt; what a step consumes is copied out with v_mov.  Every vector-memory instruction of the loop is issued
// here (a compiler-issued one would be counted by the compiler without these).
//   x ring : 4 slots of 8 registers  v[96 + 8 s ...]   (four gathered values of a group)
//   Q ring : 8 slots of 16 registers v[128 + 16 s ...] ([0:3] column words, [4:5] tags, [6:13] values / [6] codes)
#if defined(STREAM_POL)
#define NTS STREAM_POL
#elif defined(STREAM_NT)
#define NTS " nt"
#else
#define NTS ""
#endif
#ifndef XPOL
#define XPOL ""          // cache policy bits of the x gathers (-DXPOL='" sc1"': probe, profiles/r06_gather_policy_probe.log)
#endif
#ifndef TOK_SLEEP
#define TOK_SLEEP 1
#endif
#define S_(x) #x
#define S(x) S_(x)
#define RING_CLOBBER "memory", "v96","v97","v98","v99","v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115","v116","v117","v118","v119","v120","v121","v122","v123","v124","v125","v126","v127", \
    "v128","v129","v130","v131","v132","v133","v134","v135","v136","v137","v138","v139","v140","v141","v142","v143","v144","v145","v146","v147","v148","v149","v150","v151","v152","v153","v154","v155","v156","v157","v158","v159", \
    "v160","v161","v162","v163","v164","v165","v166","v167","v168","v169","v170","v171","v172","v173","v174","v175","v176","v177","v178","v179","v180","v181","v182","v183","v184","v185","v186","v187","v188","v189","v190","v191", \
    "v192","v193","v194","v195","v196","v197","v198","v199","v200","v201","v202","v203","v204","v205","v206","v207","v208","v209","v210","v211","v212","v213","v214","v215","v216","v217","v218","v219","v220","v221","v222","v223", \
    "v224","v225","v226","v227","v228","v229","v230","v231","v232","v233","v234","v235","v236","v237","v238","v239","v240","v241","v242","v243","v244","v245","v246","v247","v248","v249","v250","v251","v252","v253","v254","v255"
#define QR(u, o, n) "v[128+16*" S(u) "+" S(o) ":128+16*" S(u) "+" S(o) "+" S(n) "-1]"
#define QR1(u, o) "v[128+16*" S(u) "+" S(o) "]"
#define XR(s, j, n) "v[96+8*" S(s) "+" S(n) "*" S(j) ":96+8*" S(s) "+" S(n) "*" S(j) "+" S(n) "-1]"
#define XR1(s, o) "v[96+8*" S(s) "+" S(o) "]"

// TOK = U > 0 (mode 4, round 6): the W wavefronts of a workgroup share one chunk and take its groups in UNITS of U groups in turn (unit n = groups
// [n U, (n + 1) U) belongs to wavefront n % W); a unit's LDS additions are issued only once the token word in LDS says it is unit n's turn, and the
// token moves on behind them -- LDS operations execute in the order the LDS receives them, so every row's products are added in the order of the
// chunk's sorted list whatever the wavefronts' timing: bitwise reproducible, the sums of the CSR loop.  Loads, gathers and products run ahead freely.
// GBASE (mode 5): no 16-bit tag block -- the column word holds the column's offset from its group's first (smallest) column in its low 17 bits and the
// row above them; the groups' first columns stand in a table per wavefront (scalar loads, a revolution of the ring ahead).
// PACK4 (mode 6): GBASE with the dictionary code in the column word too -- offset (13 bits) | tag (15 bits) << 13 | code (4 bits) << 28: 4 bytes per slot, one
// stream load per group (dictionaries of at most 16 entries)
template <typename T, bool DICT, bool TAG, bool SHARED, bool BAR = false, bool REP = false, int TOK = 0, bool GBASE = false, bool PACK4 = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_num_vgpr(96))) void sorted_spmv_ring_kernel(const uint8_t *__restrict__ stream, const ChunkDesc *__restrict__ desc, const uint32_t *__restrict__ wg_first,
                                                           const uint32_t *__restrict__ wg_count, const T *__restrict__ x, T *__restrict__ z, uint32_t col_bits,
                                                           uint32_t R, const T *__restrict__ dict_g, uint32_t ndict, const uint32_t *__restrict__ rowslot = nullptr,
                                                           const uint32_t *__restrict__ gbase = nullptr, uint32_t gb_stride = 0)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr uint32_t GB = PACK4 ? 1024u : group_bytes<T, DICT, TAG>();
    constexpr uint32_t VB = 1024u + (TAG ? 512u : 0u);
    constexpr int      NS = PACK4 ? 1 : 1 + (TAG ? 1 : 0) + (DICT ? 1 : sizeof(T) == 8 ? 2 : 1);       // stream loads per group
    constexpr int      D = 4, QN = 8, K = (D - 1) * (4 + NS);
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6, nwv = blockDim.x >> 6;
    T *const dict = reinterpret_cast<T *>(smem);
    T *const acc_all = dict + (DICT ? 256 : 0);
    const uint32_t first = wg_first[blockIdx.x], cnt = wg_count[blockIdx.x];
    const uint32_t wv_u = __builtin_amdgcn_readfirstlane(wv);
    if constexpr (DICT) for (uint32_t i = threadIdx.x; i < 256u; i += blockDim.x) dict[i] = i < ndict ? dict_g[i] : T(0);
    uint32_t k;
    bool     live;
    T       *acc;
    if constexpr (SHARED) { k = first; live = cnt > 0; acc = acc_all; }
    else { k = first + wv_u; live = wv_u < cnt; acc = acc_all + (size_t)wv_u * (R + 1); }
    ChunkDesc d = live ? desc[k] : ChunkDesc{0, 0, 0, 0, 0, 0, 0, 0};
    if constexpr (SHARED) { for (uint32_t i = threadIdx.x; i <= (REP ? d.nacc : d.nrows); i += blockDim.x) acc[i] = T(0); }
    else if (live) for (uint32_t i = lane; i <= d.nrows; i += 64u) acc[i] = T(0);
    if constexpr (TOK > 0) if (threadIdx.x == 0) *reinterpret_cast<uint32_t *>(acc_all + R + 1) = 0u;
    if constexpr (SHARED || DICT) __syncthreads();
    if (!live) return;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(stream + d.stream_off, d.G * GB);
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(x + d.xbase, d.xcols * (uint32_t)sizeof(T));
    const uint32_t cmask = col_bits >= 32 ? 0xffffffffu : (1u << col_bits) - 1u;
    const uint32_t g0 = SHARED ? wv_u : 0u, gs = SHARED ? nwv : 1u;
    const uint32_t vo_c = lane * 16u, vo_t = lane * 8u + 1024u, vo_code = lane * 4u + VB, vo_v0 = lane * 16u + VB, vo_v1 = lane * 16u + VB + 1024u;
    (void)vo_t; (void)vo_code; (void)vo_v0; (void)vo_v1;

#define LOADQ(u, grp)                                                                                                                   \
    do {                                                                                                                                \
        const uint32_t so_ = __builtin_amdgcn_readfirstlane((grp) * GB);                                                               \
        asm volatile("buffer_load_dwordx4 " QR(u, 0, 4) ", %0, %1, %2 offen" NTS ::"v"(vo_c), "s"(rs), "s"(so_) : RING_CLOBBER);           \
        if constexpr (TAG) asm volatile("buffer_load_dwordx2 " QR(u, 4, 2) ", %0, %1, %2 offen" NTS ::"v"(vo_t), "s"(rs), "s"(so_) : RING_CLOBBER); \
        if constexpr (PACK4) {}                                                                                                         \
        else if constexpr (DICT) asm volatile("buffer_load_dword " QR1(u, 6) ", %0, %1, %2 offen" NTS ::"v"(vo_code), "s"(rs), "s"(so_) : RING_CLOBBER); \
        else if constexpr (sizeof(T) == 8) {                                                                                            \
            asm volatile("buffer_load_dwordx4 " QR(u, 6, 4) ", %0, %1, %2 offen" NTS ::"v"(vo_v0), "s"(rs), "s"(so_) : RING_CLOBBER);      \
            asm volatile("buffer_load_dwordx4 " QR(u, 10, 4) ", %0, %1, %2 offen" NTS ::"v"(vo_v1), "s"(rs), "s"(so_) : RING_CLOBBER);     \
        } else asm volatile("buffer_load_dwordx4 " QR(u, 6, 4) ", %0, %1, %2 offen" NTS ::"v"(vo_v0), "s"(rs), "s"(so_) : RING_CLOBBER);   \
    } while (0)
    // gather the x of the group in Q slot `un` into x slot `xsl`
#define GATHER(un, xsl, gbv)                                                                                                            \
    do {                                                                                                                                \
        uint32_t c0_, c1_, c2_, c3_;                                                                                                    \
        asm volatile("v_mov_b32 %0, " QR1(un, 0) "\n\tv_mov_b32 %1, " QR1(un, 1) "\n\tv_mov_b32 %2, " QR1(un, 2) "\n\tv_mov_b32 %3, " QR1(un, 3)   \
                     : "=v"(c0_), "=v"(c1_), "=v"(c2_), "=v"(c3_)::"memory");                                                       \
        const uint32_t gb__ = GBASE ? (gbv) : 0u;                                                                                       \
        c0_ = ((c0_ & cmask) + gb__) * (uint32_t)sizeof(T); c1_ = ((c1_ & cmask) + gb__) * (uint32_t)sizeof(T); c2_ = ((c2_ & cmask) + gb__) * (uint32_t)sizeof(T); c3_ = ((c3_ & cmask) + gb__) * (uint32_t)sizeof(T); \
        if constexpr (sizeof(T) == 8) {                                                                                                 \
            asm volatile("buffer_load_dwordx2 " XR(xsl, 0, 2) ", %0, %4, 0 offen" XPOL "\n\tbuffer_load_dwordx2 " XR(xsl, 1, 2) ", %1, %4, 0 offen" XPOL "\n\t"         \
                         "buffer_load_dwordx2 " XR(xsl, 2, 2) ", %2, %4, 0 offen" XPOL "\n\tbuffer_load_dwordx2 " XR(xsl, 3, 2) ", %3, %4, 0 offen" XPOL             \
                         ::"v"(c0_), "v"(c1_), "v"(c2_), "v"(c3_), "s"(rx) : RING_CLOBBER);                                              \
        } else {                                                                                                                        \
            asm volatile("buffer_load_dword " XR1(xsl, 0) ", %0, %4, 0 offen" XPOL "\n\tbuffer_load_dword " XR1(xsl, 1) ", %1, %4, 0 offen" XPOL "\n\t"               \
                         "buffer_load_dword " XR1(xsl, 2) ", %2, %4, 0 offen" XPOL "\n\tbuffer_load_dword " XR1(xsl, 3) ", %3, %4, 0 offen" XPOL                   \
                         ::"v"(c0_), "v"(c1_), "v"(c2_), "v"(c3_), "s"(rx) : RING_CLOBBER);                                              \
        }                                                                                                                               \
    } while (0)
    // copy the group of Q slot u and its x out of the rings
#define TAKE(u, xsl)                                                                                                                    \
    uint32_t cw_[4], tg_[2] = {0, 0}, vv_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, xx_[8] = {0, 0, 0, 0, 0, 0, 0, 0};                                 \
    asm volatile("v_mov_b32 %0, " QR1(u, 0) "\n\tv_mov_b32 %1, " QR1(u, 1) "\n\tv_mov_b32 %2, " QR1(u, 2) "\n\tv_mov_b32 %3, " QR1(u, 3)           \
                 : "=v"(cw_[0]), "=v"(cw_[1]), "=v"(cw_[2]), "=v"(cw_[3])::"memory");                                                  \
    if constexpr (TAG) asm volatile("v_mov_b32 %0, " QR1(u, 4) "\n\tv_mov_b32 %1, " QR1(u, 5) : "=v"(tg_[0]), "=v"(tg_[1])::"memory");       \
    if constexpr (PACK4) {}                                                                                                             \
    else if constexpr (DICT) asm volatile("v_mov_b32 %0, " QR1(u, 6) : "=v"(vv_[0])::"memory");                                             \
    else if constexpr (sizeof(T) == 8)                                                                                                  \
        asm volatile("v_mov_b32 %0, " QR1(u, 6) "\n\tv_mov_b32 %1, " QR1(u, 7) "\n\tv_mov_b32 %2, " QR1(u, 8) "\n\tv_mov_b32 %3, " QR1(u, 9) "\n\t"   \
                     "v_mov_b32 %4, " QR1(u, 10) "\n\tv_mov_b32 %5, " QR1(u, 11) "\n\tv_mov_b32 %6, " QR1(u, 12) "\n\tv_mov_b32 %7, " QR1(u, 13)     \
                     : "=v"(vv_[0]), "=v"(vv_[1]), "=v"(vv_[2]), "=v"(vv_[3]), "=v"(vv_[4]), "=v"(vv_[5]), "=v"(vv_[6]), "=v"(vv_[7])::"memory"); \
    else asm volatile("v_mov_b32 %0, " QR1(u, 6) "\n\tv_mov_b32 %1, " QR1(u, 7) "\n\tv_mov_b32 %2, " QR1(u, 8) "\n\tv_mov_b32 %3, " QR1(u, 9) \
                      : "=v"(vv_[0]), "=v"(vv_[1]), "=v"(vv_[2]), "=v"(vv_[3])::"memory");                                            \
    if constexpr (sizeof(T) == 8)                                                                                                       \
        asm volatile("v_mov_b32 %0, " XR1(xsl, 0) "\n\tv_mov_b32 %1, " XR1(xsl, 1) "\n\tv_mov_b32 %2, " XR1(xsl, 2) "\n\tv_mov_b32 %3, " XR1(xsl, 3) "\n\t" \
                     "v_mov_b32 %4, " XR1(xsl, 4) "\n\tv_mov_b32 %5, " XR1(xsl, 5) "\n\tv_mov_b32 %6, " XR1(xsl, 6) "\n\tv_mov_b32 %7, " XR1(xsl, 7)     \
                     : "=v"(xx_[0]), "=v"(xx_[1]), "=v"(xx_[2]), "=v"(xx_[3]), "=v"(xx_[4]), "=v"(xx_[5]), "=v"(xx_[6]), "=v"(xx_[7])::"memory"); \
    else asm volatile("v_mov_b32 %0, " XR1(xsl, 0) "\n\tv_mov_b32 %1, " XR1(xsl, 1) "\n\tv_mov_b32 %2, " XR1(xsl, 2) "\n\tv_mov_b32 %3, " XR1(xsl, 3) \
                      : "=v"(xx_[0]), "=v"(xx_[1]), "=v"(xx_[2]), "=v"(xx_[3])::"memory")
#define STEP(u, un, xsl)                                                                                                                \
    do {                                                                                                                                \
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(K) : "memory");                                                                       \
        TAKE(u, xsl);                                                                                                                   \
        GATHER(un, xsl, (u < 4 ? bcur[(u + 4) & 7] : bnext[(u + 4) & 7]));                                                              \
        LOADQ(u, GG(tb + (uint32_t)(u + QN)));                                                                                          \
        T pr_[4]; uint32_t rw_[4];            /* products and rows first: with a token nothing but the additions themselves happens while it is held */ \
        _Pragma("unroll") for (int j = 0; j < 4; j++) {                                                                                \
            T av, xv;                                                                                                                   \
            if constexpr (PACK4) av = dict[cw_[j] >> 28];                                                                               \
            else if constexpr (DICT) av = dict[(vv_[0] >> (8 * j)) & 0xffu];                                                            \
            else if constexpr (sizeof(T) == 8) av = __builtin_bit_cast(double, (uint64_t)vv_[2 * j] | ((uint64_t)vv_[2 * j + 1] << 32)); \
            else av = __builtin_bit_cast(float, vv_[j]);                                                                                \
            if constexpr (sizeof(T) == 8) xv = __builtin_bit_cast(double, (uint64_t)xx_[2 * j] | ((uint64_t)xx_[2 * j + 1] << 32));    \
            else xv = __builtin_bit_cast(float, xx_[j]);                                                                                \
            if constexpr (PACK4) rw_[j] = (cw_[j] >> 13) & 0x7fffu;                                                                      \
            else if constexpr (TAG) rw_[j] = (tg_[j >> 1] >> (16 * (j & 1))) & 0xffffu;                                                  \
            else rw_[j] = col_bits >= 32 ? 0u : cw_[j] >> col_bits;                                                                     \
            pr_[j] = av * xv;                                                                                                           \
        }                                                                                                                               \
        if constexpr (TOK > 0) {          /* the unit's products wait in registers; its last group takes the token, adds them all in order, passes it on */ \
            _Pragma("unroll") for (int j = 0; j < 4; j++) { prb[(u) % TOK][j] = pr_[j]; rwb[(u) % TOK][j] = rw_[j]; }                    \
            if ((u) % TOK == TOK - 1) {                                                                                                 \
                _Pragma("unroll") for (int q = 0; q < TOK; q++) _Pragma("unroll") for (int j = 0; j < 4; j++) { asm volatile("" : "+v"(prb[q][j])); asm volatile("" : "+v"(rwb[q][j])); } \
                if (tb + (u) < Tw) {                                                                                                    \
                    const uint32_t n_ = ((tb + (u)) / (uint32_t)TOK) * nwv + wv_u;                                                      \
                    asm volatile("" ::: "memory");                                                                                      \
                    while (__hip_atomic_load(tok, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != n_) { if (TOK_SLEEP > 0) __builtin_amdgcn_s_sleep(TOK_SLEEP); } \
                    asm volatile("" ::: "memory");                                                                                      \
                    _Pragma("unroll") for (int q = 0; q < TOK; q++) _Pragma("unroll") for (int j = 0; j < 4; j++) lds_add<T, SHARED>(acc + rwb[q][j], prb[q][j]); \
                    asm volatile("" ::: "memory");                                                                                      \
                    if (lane == 0) __hip_atomic_store(tok, n_ + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);                    \
                    asm volatile("" ::: "memory");                                                                                      \
                }                                                                                                                       \
            }                                                                                                                           \
        } else {                                                                                                                        \
            _Pragma("unroll") for (int j = 0; j < 4; j++) lds_add<T, SHARED>(acc + rw_[j], pr_[j]);                                     \
        }                                                                                                                               \
        if constexpr (BAR) __builtin_amdgcn_s_barrier();      /* mode 2: every wavefront has added its group of the round */         \
    } while (0)

    // the wavefront's t-th group is group GG(t) of the chunk (private: t; shared: the groups, or units of TOK groups, in turn)
#define GG(t) (TOK > 0 ? ((t) / (uint32_t)TOK) * (nwv * (uint32_t)TOK) + wv_u * (uint32_t)TOK + (t) % (uint32_t)TOK : g0 + (t) * gs)
    // groups every wavefront walks (the same count in all of them with a barrier or a token: every unit's turn must be taken)
    const uint32_t Tw = TOK > 0 ? (d.G + nwv * (uint32_t)TOK - 1u) / (nwv * (uint32_t)TOK) * (uint32_t)TOK : BAR ? (d.G + gs - 1u) / gs : d.G > g0 ? (d.G - g0 + gs - 1u) / gs : 0u;
    uint32_t *const tok = reinterpret_cast<uint32_t *>(acc_all + R + 1);
    (void)tok;
    T        prb[TOK > 0 ? TOK : 1][4];
    uint32_t rwb[TOK > 0 ? TOK : 1][4];
    (void)prb; (void)rwb;
    const uint32_t *gbw = GBASE ? gbase + d.nacc + wv_u * gb_stride : nullptr;
    uint32_t bcur[8] = {0, 0, 0, 0, 0, 0, 0, 0}, bnext[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if constexpr (GBASE) {
#pragma unroll
        for (int i = 0; i < 8; i++) bcur[i] = gbw[i];
    }
    // run-in: the stream of the first QN groups, then virtual steps -D .. -1 issue what steps of the loop would have issued
    {
        LOADQ(0, GG(0u)); LOADQ(1, GG(1u)); LOADQ(2, GG(2u)); LOADQ(3, GG(3u));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        GATHER(0, 0, bcur[0]); LOADQ(4, GG(4u));
        GATHER(1, 1, bcur[1]); LOADQ(5, GG(5u));
        GATHER(2, 2, bcur[2]); LOADQ(6, GG(6u));
        GATHER(3, 3, bcur[3]); LOADQ(7, GG(7u));
    }
    for (uint32_t tb = 0; tb < Tw; tb += QN) {
        if constexpr (GBASE) {
#pragma unroll
            for (int i = 0; i < 8; i++) bnext[i] = gbw[tb + 8u + (uint32_t)i];
        }
        STEP(0, 4, 0); STEP(1, 5, 1); STEP(2, 6, 2); STEP(3, 7, 3);
        STEP(4, 0, 0); STEP(5, 1, 1); STEP(6, 2, 2); STEP(7, 3, 3);
        if constexpr (GBASE) {
#pragma unroll
            for (int i = 0; i < 8; i++) bcur[i] = bnext[i];
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef STEP
#undef GG
#undef TAKE
#undef GATHER
#undef LOADQ
    if constexpr (SHARED && REP) {          // mode 3: a replicated row has one accumulator per wavefront, added up in wavefront order
        __syncthreads();
        constexpr uint32_t kB = 8;                       // (eight table loads in flight per thread: the loop is a chain of round trips otherwise)
        for (uint32_t i0 = threadIdx.x; i0 < d.nrows; i0 += blockDim.x * kB) {
            uint32_t rs_[kB];
#pragma unroll
            for (uint32_t u = 0; u < kB; u++) { const uint32_t i = i0 + u * blockDim.x; rs_[u] = i < d.nrows ? rowslot[d.zoff + i] : 0u; }
#pragma unroll
            for (uint32_t u = 0; u < kB; u++) {
                const uint32_t i = i0 + u * blockDim.x, sl = rs_[u] & 0x7fffffffu;
                if (i < d.nrows) z[d.zoff + i] = (rs_[u] >> 31) ? ((acc[sl] + acc[sl + 1]) + acc[sl + 2]) + acc[sl + 3] : acc[sl];
            }
        }
    } else if constexpr (SHARED) {
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < d.nrows; i += blockDim.x) z[d.zoff + i] = acc[i + 1];
    } else {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (uint32_t i = lane; i < d.nrows; i += 64u) z[d.zoff + i] = acc[i + 1];
    }
}

struct Csr { int64_t nrows, ncols, nnz; std::vector<int64_t> rp; std::vector<int32_t> ci; std::vector<double> va; };

static Csr read_csr(const char *path, bool f32)
{
    Csr A;
    FILE *f = fopen(path, "rb");
    if (!f) { perror(path); exit(1); }
    int64_t h[3];
    if (fread(h, 8, 3, f) != 3) exit(1);
    A.nrows = h[0]; A.ncols = h[1]; A.nnz = h[2];
    A.rp.resize(A.nrows + 1); A.ci.resize(A.nnz); A.va.resize(A.nnz);
    if (fread(A.rp.data(), 8, A.nrows + 1, f) != (size_t)A.nrows + 1) exit(1);
    if (fread(A.ci.data(), 4, A.nnz, f) != (size_t)A.nnz) exit(1);
    if (f32) { std::vector<float> t(A.nnz); if (fread(t.data(), 4, A.nnz, f) != (size_t)A.nnz) exit(1); for (int64_t i = 0; i < A.nnz; i++) A.va[i] = t[i]; }
    else if (fread(A.va.data(), 8, A.nnz, f) != (size_t)A.nnz) exit(1);
    fclose(f);
    return A;
}

struct HostChunk { int panel; int64_t sub0, sub1; int64_t e0, e1; };       // sub-rows [sub0, sub1) of the panel, elements [e0, e1) of the panel's arrays

template <typename T>
static int run(const Csr &A, uint32_t R, uint32_t W, uint32_t Smax, uint32_t P, int mode, int depth, int iters, bool want_dict, int noadd)
{
    const uint32_t per_line = 128 / sizeof(T);
    uint32_t pw = (uint32_t)((A.ncols + P - 1) / P);
    pw = (pw + per_line - 1) / per_line * per_line;
    P = (uint32_t)((A.ncols + pw - 1) / pw);
    // panel p = columns [pb[p], pb[p + 1]): equal widths, or (BALANCE=1) cut where the running count of non-zeros passes p / P of them, on
    // 128-byte lines of x -- R-MAT's hot low columns then make a narrow first panel and every XCD gets the same number of non-zeros
    std::vector<uint32_t> pb(P + 1);
    for (uint32_t p = 0; p <= P; p++) pb[p] = (uint32_t)std::min<int64_t>((int64_t)p * pw, A.ncols);
    if (getenv("BALANCE") && atoi(getenv("BALANCE"))) {
        std::vector<int64_t> cc((size_t)A.ncols + 1, 0);
        for (int64_t i = 0; i < A.nnz; i++) cc[(size_t)A.ci[i] + 1]++;
        for (int64_t c = 0; c < A.ncols; c++) cc[(size_t)c + 1] += cc[(size_t)c];
        for (uint32_t p = 1; p < P; p++) {
            const int64_t want = A.nnz * (int64_t)p / P;
            int64_t       c = std::lower_bound(cc.begin(), cc.end(), want) - cc.begin();
            c = (c + per_line - 1) / per_line * per_line;
            pb[p] = (uint32_t)std::min<int64_t>(std::max<int64_t>(c, (int64_t)pb[p - 1] + per_line), A.ncols);
        }
        pb[P] = (uint32_t)A.ncols;
        pw = 0;
        for (uint32_t p = 0; p < P; p++) pw = std::max(pw, pb[p + 1] - pb[p]);
        printf("# nnz-balanced panels, first columns:");
        for (uint32_t p = 0; p <= P; p++) printf(" %u", pb[p]);
        printf("\n");
    }
    auto panel_of = [&](int32_t c) { return (int)(std::upper_bound(pb.begin() + 1, pb.end() - 1, (uint32_t)c) - (pb.begin() + 1)); };
    const uint32_t Reff = mode >= 1 ? R * W : R;          // (mode 3: rows per chunk; its accumulators -- rows + 3 per replicated row -- must fit the LDS)
    // ---- split into panels: per panel the sub-rows (row, begin) and the elements, in row order
    struct Panel { std::vector<int64_t> sp; std::vector<int32_t> srow; std::vector<int32_t> col; std::vector<double> val; };
    std::vector<Panel> pan(P);
    {
        std::vector<std::vector<int64_t>> cnt(P);
        std::vector<int64_t> pn(P, 0), ps(P, 0);
        for (int64_t r = 0; r < A.nrows; r++) {
            int lastp = -1;
            for (int64_t j = A.rp[r]; j < A.rp[r + 1]; j++) { const int p = panel_of(A.ci[j]); pn[p]++; if (p != lastp) { ps[p]++; lastp = p; } }
            if (P == 1 && A.rp[r] == A.rp[r + 1]) ps[0]++;      // one panel: empty rows stay (they are written as 0 by their chunk)
        }
        for (uint32_t p = 0; p < P; p++) { pan[p].sp.reserve(ps[p] + 1); pan[p].srow.reserve(ps[p]); pan[p].col.reserve(pn[p]); pan[p].val.reserve(pn[p]); pan[p].sp.push_back(0); }
        for (int64_t r = 0; r < A.nrows; r++) {
            int lastp = -1;
            for (int64_t j = A.rp[r]; j < A.rp[r + 1]; j++) {
                const int p = panel_of(A.ci[j]);
                if (p != lastp) { if (lastp >= 0) pan[lastp].sp.push_back((int64_t)pan[lastp].col.size()); pan[p].srow.push_back((int32_t)r); lastp = p; }
                pan[p].col.push_back(A.ci[j] - (int32_t)pb[p]); pan[p].val.push_back(A.va[j]);
            }
            if (lastp >= 0) pan[lastp].sp.push_back((int64_t)pan[lastp].col.size());
            else if (P == 1) { pan[0].srow.push_back((int32_t)r); pan[0].sp.push_back((int64_t)pan[0].col.size()); }
        }
    }
    // ---- chunks: consecutive sub-rows, at most Reff of them, at most 64 * Smax elements; longer rows are cut
    const int64_t budget = (int64_t)64 * Smax;
    std::vector<HostChunk> chunks;
    int64_t npairs = 0;
    for (uint32_t p = 0; p < P; p++) {
        const auto &sp = pan[p].sp;
        const int64_t ns = (int64_t)pan[p].srow.size();
        npairs += ns;
        int64_t s = 0;
        while (s < ns) {
            if (sp[s + 1] - sp[s] > budget) {        // a long row: pieces of `budget` elements, each a chunk with one row (partial sums)
                for (int64_t e = sp[s]; e < sp[s + 1]; e += budget) chunks.push_back({(int)p, s, s + 1, e, std::min(sp[s + 1], e + budget)});
                s++;
                continue;
            }
            int64_t t = s;
            while (t < ns && t - s < (int64_t)Reff && sp[t + 1] - sp[s] <= budget && sp[t + 1] - sp[t] <= budget) t++;
            chunks.push_back({(int)p, s, t, sp[s], sp[t]});
            s = t;
        }
    }
    const size_t nch = chunks.size();
    // ---- dictionary
    std::vector<T> dict;
    bool use_dict = false;
    if (want_dict) {
        std::vector<double> u;
        for (int64_t i = 0; i < A.nnz && u.size() <= 256; i++) if (std::find(u.begin(), u.end(), A.va[i]) == u.end()) u.push_back(A.va[i]);
        if (std::find(u.begin(), u.end(), 0.0) == u.end()) u.push_back(0.0);
        if (u.size() <= 256) { use_dict = true; std::sort(u.begin(), u.end()); for (double v : u) dict.push_back((T)v); }
    }
    // ---- formats
    uint32_t col_bits = 1; while ((1ull << col_bits) < pw) col_bits++;
    uint32_t row_bits = 1; while ((1ull << row_bits) < (uint64_t)Reff + 1) row_bits++;      // tags 0 .. Reff
    const bool pack4 = mode == 6;                                 // group-base packing with the code in the word: 13 + 15 + 4 bits
    const bool gbm = mode == 5 || pack4;                          // group-base packing: 17 bits of column offset + the row in one word
    const uint32_t TOKU = getenv("TOK_U") ? (uint32_t)atoi(getenv("TOK_U")) : 1u;      // modes 4 / 5: groups per unit (1, 2, 4, 8)
    if (gbm && row_bits > 15) { fprintf(stderr, "rows per chunk beyond 15 bits\n"); return 1; }
    const bool tag = !gbm && (col_bits + row_bits > 32 || mode == 3);
    if (tag && Reff + 1 > 65536) { fprintf(stderr, "rows per chunk beyond 16-bit tags\n"); return 1; }
    if (pack4 && (!use_dict || dict.size() > 16)) { fprintf(stderr, "mode 6 needs a dictionary of at most 16 entries\n"); return 1; }
    const uint32_t GB = pack4 ? 1024u : 1024u + (tag ? 512u : 0u) + (use_dict ? 256u : sizeof(T) == 8 ? 2048u : 1024u);
    std::vector<ChunkDesc> desc(nch);
    uint64_t soff = 0, zoff = 0;
    for (size_t k = 0; k < nch; k++) {
        const auto &c = chunks[k];
        const uint32_t G = (uint32_t)((c.e1 - c.e0 + 255) / 256);
        desc[k] = {soff, G, (uint32_t)(c.sub1 - c.sub0), zoff, pb[c.panel], pb[c.panel + 1] - pb[c.panel], 0u, 0u};
        soff += (uint64_t)G * GB; zoff += (uint64_t)(c.sub1 - c.sub0);
    }
    std::vector<uint8_t> stream(soff + 8 * GB, 0);
    std::vector<int32_t> zrow(zoff);
    std::vector<uint32_t> rowslot(zoff + 1, 0);
    uint32_t Gmax = 0; for (auto &dd : desc) Gmax = std::max(Gmax, dd.G);
    const uint32_t gb_stride = ((Gmax + W * TOKU - 1) / (W * TOKU) * TOKU + 7) / 8 * 8 + 16;      // a wavefront's table: its groups' first columns, then zeros for the run-ahead
    std::vector<uint32_t> gbase(gbm ? (size_t)nch * W * gb_stride + 64 : 64, 0);
    int64_t gb_bad = 0;
    if (mode >= 4) for (size_t k = 0; k < nch; k++) desc[k].nacc = (uint32_t)(k * W * gb_stride);
    const long hotT = getenv("HOT_T") ? atol(getenv("HOT_T")) : 16;
    int64_t nrep_tot = 0, nconf_tot = 0; uint32_t nacc_max = 0;
    double t0 = omp_get_wtime();
#pragma omp parallel for schedule(dynamic, 8)
    for (size_t k = 0; k < nch; k++) {
        const auto &c = chunks[k];
        const Panel &pp = pan[c.panel];
        const int64_t n = c.e1 - c.e0;
        std::vector<uint64_t> key(n);          // (col, row in chunk) -> sort
        {
            int64_t s = c.sub0;
            for (int64_t e = c.e0; e < c.e1; e++) {
                while (pp.sp[s + 1] <= e) s++;
                key[e - c.e0] = ((uint64_t)(uint32_t)pp.col[e] << 24) | (uint64_t)(s - c.sub0);
            }
        }
        // key = col << 24 | row in chunk; stable sort by column keeps a column's rows ascending
        std::vector<uint32_t> idx(n);
        std::iota(idx.begin(), idx.end(), 0u);
        std::stable_sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return (key[a] >> 24) < (key[b] >> 24); });
        std::vector<uint32_t> slot;          // mode 3: accumulator of a row (+ the wavefront for a replicated row)
        std::vector<uint8_t>  rep;
        if (mode == 3) {
            const uint32_t nr = (uint32_t)(c.sub1 - c.sub0);
            std::vector<uint32_t> deg(nr, 0), orow(nr, 0xffffffffu);
            std::vector<uint8_t>  owav(nr, 0);
            rep.assign(nr, 0);
            for (int64_t e = 0; e < n; e++) deg[key[idx[e]] & 0xffffffu]++;
            for (uint32_t r = 0; r < nr; r++) rep[r] = deg[r] >= (uint32_t)hotT;
            int64_t nconf = 0;
            for (int64_t e = 0; e < n; e++) {                      // a row that two wavefronts touch in one round is replicated as well
                const uint32_t r = (uint32_t)(key[idx[e]] & 0xffffffu), round = (uint32_t)(e / (256 * W)), wave = (uint32_t)(e / 256) % W;
                if (rep[r]) continue;
                if (orow[r] == round && owav[r] != wave && !getenv("NOREP")) { rep[r] = 1; nconf++; }
                else { orow[r] = round; owav[r] = (uint8_t)wave; }
            }
            slot.resize(nr);
            uint32_t s = 1, nrep = 0;
            for (uint32_t r = 0; r < nr; r++) { slot[r] = s; s += rep[r] ? W : 1; nrep += rep[r]; rowslot[desc[k].zoff + r] = slot[r] | ((uint32_t)rep[r] << 31); }
            desc[k].nacc = s;
#pragma omp critical
            { nrep_tot += nrep; nconf_tot += nconf; nacc_max = std::max(nacc_max, s); }
        }
        uint8_t *base = stream.data() + desc[k].stream_off;
        for (int64_t e = 0; e < (int64_t)desc[k].G * 256; e++) {
            const uint32_t g = (uint32_t)(e / 256), j = (uint32_t)(e % 256) / 64, ln = (uint32_t)(e % 64);
            uint8_t *gp = base + (size_t)g * GB;
            uint32_t col, row; double v;
            if (e < n) { const uint32_t i = idx[e]; col = (uint32_t)(key[i] >> 24); row = (uint32_t)(key[i] & 0xffffffu); v = pp.val[c.e0 + i]; }
            else { col = n ? (uint32_t)(key[idx[n - 1]] >> 24) : 0u; row = 0xffffffffu; v = 0.0; }      // pad: the last column again, the dump accumulator (tag 0)
            row += 1u;            // tags are biased by one: tag 0 (what a load past the end returns) is the dump accumulator
            if (mode == 3) row = e < n ? slot[row - 1] + (rep[row - 1] ? (uint32_t)(e / 256) % W : 0u) : 0u;
            uint32_t cw = tag ? col : (col_bits >= 32 ? col : col | (row << col_bits));
            if (gbm) {
                const uint32_t e256 = g * 256u, bcol = e256 < (uint32_t)n ? (uint32_t)(key[idx[e256]] >> 24) : 0u;      // the group's first = smallest column
                if (e % 256 == 0 && e < n) { const uint32_t wq = (g / TOKU) % W, tq = (g / (TOKU * W)) * TOKU + g % TOKU; gbase[desc[k].nacc + wq * gb_stride + tq] = bcol; }
                const uint32_t off = e < n ? col - bcol : 0u;
                if (off >= (pack4 ? 1u << 13 : 1u << 17)) {
#pragma omp atomic
                    gb_bad++;
                }
                cw = e < n ? (off & 0x1ffffu) | (row << 17) : 0u;
                if (pack4) {
                    const T tv = (T)v;
                    const uint32_t code = (uint32_t)(std::lower_bound(dict.begin(), dict.end(), tv) - dict.begin());
                    cw = e < n ? (off & 0x1fffu) | (row << 13) | (code << 28) : (uint32_t)(std::lower_bound(dict.begin(), dict.end(), (T)0) - dict.begin()) << 28;
                }
            }
            reinterpret_cast<uint32_t *>(gp)[ln * 4 + j] = cw;
            if (tag) reinterpret_cast<uint16_t *>(gp + 1024)[ln * 4 + j] = (uint16_t)row;
            uint8_t *vp = gp + 1024 + (tag ? 512 : 0);
            if (pack4) {}
            else if (use_dict) { const T tv = (T)v; const uint32_t code = (uint32_t)(std::lower_bound(dict.begin(), dict.end(), tv) - dict.begin()); vp[ln * 4 + j] = (uint8_t)code; }
            else if (sizeof(T) == 8) reinterpret_cast<double *>(vp + (j >= 2 ? 1024 : 0))[ln * 2 + (j & 1)] = v;
            else reinterpret_cast<float *>(vp)[ln * 4 + j] = (float)v;
        }
        for (int64_t s = c.sub0; s < c.sub1; s++) zrow[desc[k].zoff + (s - c.sub0)] = pp.srow[s];
    }
    double t1 = omp_get_wtime();
    // ---- workgroups: W consecutive chunks of one panel (mode 0) or one chunk (mode 1); panels dealt to the XCDs round-robin
    std::vector<std::vector<std::pair<uint32_t, uint32_t>>> xq(P > 1 ? 8 : 1);
    {
        size_t k = 0;
        while (k < nch) {
            const int p = chunks[k].panel;
            uint32_t c = 1;
            if (mode == 0) while (c < W && k + c < nch && chunks[k + c].panel == p) c++;
            xq[P > 1 ? p % 8 : 0].push_back({(uint32_t)k, c});
            k += c;
        }
    }
    std::vector<uint32_t> wg_first, wg_count;
    if (P > 1) {
        size_t mx = 0; for (auto &q : xq) mx = std::max(mx, q.size());
        for (size_t i = 0; i < mx; i++) for (int xc = 0; xc < 8; xc++) {
            if (i < xq[xc].size()) { wg_first.push_back(xq[xc][i].first); wg_count.push_back(xq[xc][i].second); }
            else { wg_first.push_back(0); wg_count.push_back(0); }
        }
    } else for (auto &q : xq[0]) { wg_first.push_back(q.first); wg_count.push_back(q.second); }
    const uint32_t nwg = (uint32_t)wg_first.size();
    const size_t lds = (size_t)(use_dict ? 256 : 0) * sizeof(T) + (mode == 3 ? (size_t)(nacc_max + 1) : (size_t)(mode >= 1 ? 1 : W) * (Reff + 1)) * sizeof(T) + 16;
    if (mode >= 4) printf("# mode %d: token units of %u groups%s; column offsets beyond 17 bits: %ld\n", mode, TOKU, gbm ? ", group-base packing" : "", (long)gb_bad);
    if (mode == 3) printf("# mode 3: hot rows from %ld non-zeros; replicated rows %ld of %ld (%.1f %%), of them for a conflict %ld; most accumulators in a chunk %u\n", hotT, (long)nrep_tot, (long)npairs, 100.0 * nrep_tot / npairs, (long)nconf_tot, nacc_max);
    int64_t slots = 0; for (auto &d : desc) slots += (int64_t)d.G * 256;
    printf("# %s nrows %ld nnz %ld | P %u (%u cols, %.2f MB) pairs %.2fM chunks %zu wgs %u slots/nnz %.3f | R %u W %u Smax %u mode %d depth %d tag %d dict %d(%zu) bits %u+%u GB %u stream %.1f MB lds %zu | build %.1fs\n",
           sizeof(T) == 8 ? "f64" : "f32", (long)A.nrows, (long)A.nnz, P, pw, pw * sizeof(T) / 1e6, npairs / 1e6, nch, nwg, (double)slots / A.nnz, R, W, Smax, mode, depth, (int)tag, (int)use_dict,
           dict.size(), col_bits, row_bits, GB, soff / 1e6, lds, t1 - t0);
    if (lds > 160 * 1024) { fprintf(stderr, "LDS %zu too large\n", lds); return 1; }
    // ---- x, reference
    std::vector<T> x(A.ncols + 64);
    for (int64_t j = 0; j < A.ncols; j++) {
        uint64_t zz = 0xC0FFEEull + ((uint64_t)j + 1) * 0x9E3779B97F4A7C15ull;
        zz = (zz ^ (zz >> 30)) * 0xBF58476D1CE4E5B9ull; zz = (zz ^ (zz >> 27)) * 0x94D049BB133111EBull; zz ^= zz >> 31;
        x[j] = (T)((double)(zz >> 11) * (1.0 / 9007199254740992.0) * 2.0 - 1.0);
    }
    uint8_t *d_stream; ChunkDesc *d_desc; uint32_t *d_first, *d_count, *d_rowslot, *d_gbase; T *d_x, *d_z, *d_dict;
    CK(hipMalloc(&d_gbase, gbase.size() * 4)); CK(hipMemcpy(d_gbase, gbase.data(), gbase.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_rowslot, rowslot.size() * 4)); CK(hipMemcpy(d_rowslot, rowslot.data(), rowslot.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_stream, stream.size())); CK(hipMemcpy(d_stream, stream.data(), stream.size(), hipMemcpyHostToDevice));
    // SAME_STREAM=M (timing only, wrong sums): every chunk streams the image of one of the first M chunks of its panel (its own x slice and
    // accumulators), so the matrix stream is L2-resident -- what a perfect prefetch of the stream into the L2s would give; the gathers keep
    // the pattern of real chunks (round 5: how much of the kernel is the stream's HBM misses holding the L1's miss queue?)
    if (getenv("SAME_STREAM") && atoi(getenv("SAME_STREAM")) > 0) {
        const size_t M = (size_t)atoi(getenv("SAME_STREAM"));
        std::vector<ChunkDesc> d2 = desc;
        size_t base = 0;
        for (size_t k = 0; k < nch; k++) {
            if (k > 0 && chunks[k].panel != chunks[k - 1].panel) base = k;
            const ChunkDesc &src = desc[base + (k - base) % M < nch && chunks[base + (k - base) % M].panel == chunks[k].panel ? base + (k - base) % M : base];
            d2[k].stream_off = src.stream_off; d2[k].G = std::min(desc[k].G, src.G);
        }
        desc = d2;
    }
    CK(hipMalloc(&d_desc, nch * sizeof(ChunkDesc))); CK(hipMemcpy(d_desc, desc.data(), nch * sizeof(ChunkDesc), hipMemcpyHostToDevice));
    CK(hipMalloc(&d_first, nwg * 4)); CK(hipMemcpy(d_first, wg_first.data(), nwg * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_count, nwg * 4)); CK(hipMemcpy(d_count, wg_count.data(), nwg * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_x, x.size() * sizeof(T))); CK(hipMemcpy(d_x, x.data(), x.size() * sizeof(T), hipMemcpyHostToDevice));
    CK(hipMalloc(&d_z, (zoff + 64) * sizeof(T))); CK(hipMemset(d_z, 0xff, (zoff + 64) * sizeof(T)));
    CK(hipMalloc(&d_dict, 256 * sizeof(T))); if (use_dict) CK(hipMemcpy(d_dict, dict.data(), dict.size() * sizeof(T), hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto launch = [&]() {
#define L(DI, TG, DP, SH, NA) hipLaunchKernelGGL((sorted_spmv_kernel<T, DI, TG, DP, SH, NA>), dim3(nwg), dim3(64 * W), lds, 0, d_stream, d_desc, d_first, d_count, d_x, d_z, tag ? 32u : col_bits, Reff, d_dict, (uint32_t)dict.size())
#define L_NA(DI, TG, DP, SH) do { if (noadd) L(DI, TG, DP, SH, 1); else L(DI, TG, DP, SH, 0); } while (0)
#define L_SH(DI, TG, DP) do { if (mode >= 1) L_NA(DI, TG, DP, true); else L_NA(DI, TG, DP, false); } while (0)
#define L_TOK(DI, TG, U, GBM) do { if (pack4) { if constexpr (DI && GBM) hipLaunchKernelGGL((sorted_spmv_ring_kernel<T, true, false, true, false, false, U, true, true>), dim3(nwg), dim3(64 * W), lds, 0, d_stream, d_desc, d_first, d_count, d_x, d_z, 13u, Reff, d_dict, (uint32_t)dict.size(), d_rowslot, d_gbase, gb_stride); } \
    else hipLaunchKernelGGL((sorted_spmv_ring_kernel<T, DI, TG, true, false, false, U, GBM>), dim3(nwg), dim3(64 * W), lds, 0, d_stream, d_desc, d_first, d_count, d_x, d_z, GBM ? 17u : tag ? 32u : col_bits, Reff, d_dict, (uint32_t)dict.size(), d_rowslot, d_gbase, gb_stride); } while (0)
#define L_TOKU(DI, TG, GBM) do { if (TOKU == 1) L_TOK(DI, TG, 1, GBM); else if (TOKU == 2) L_TOK(DI, TG, 2, GBM); else if (TOKU == 4) L_TOK(DI, TG, 4, GBM); else L_TOK(DI, TG, 8, GBM); } while (0)
#define L_RING(DI, TG) do { if (mode == 5 || mode == 6) { if constexpr (!TG) L_TOKU(DI, false, true); } else if (mode == 4) L_TOKU(DI, TG, false); else if (mode == 3) hipLaunchKernelGGL((sorted_spmv_ring_kernel<T, DI, TG, true, true, true>), dim3(nwg), dim3(64 * W), lds, 0, d_stream, d_desc, d_first, d_count, d_x, d_z, tag ? 32u : col_bits, Reff, d_dict, (uint32_t)dict.size(), d_rowslot); \
        else if (mode == 2) hipLaunchKernelGGL((sorted_spmv_ring_kernel<T, DI, TG, true, true>), dim3(nwg), dim3(64 * W), lds, 0, d_stream, d_desc, d_first, d_count, d_x, d_z, tag ? 32u : col_bits, Reff, d_dict, (uint32_t)dict.size()); \
        else if (mode == 1) hipLaunchKernelGGL((sorted_spmv_ring_kernel<T, DI, TG, true>), dim3(nwg), dim3(64 * W), lds, 0, d_stream, d_desc, d_first, d_count, d_x, d_z, tag ? 32u : col_bits, Reff, d_dict, (uint32_t)dict.size()); \
        else hipLaunchKernelGGL((sorted_spmv_ring_kernel<T, DI, TG, false>), dim3(nwg), dim3(64 * W), lds, 0, d_stream, d_desc, d_first, d_count, d_x, d_z, tag ? 32u : col_bits, Reff, d_dict, (uint32_t)dict.size()); } while (0)
#define L_DP(DI, TG) do { if (depth == 9) L_RING(DI, TG); else if (depth >= 8) L_SH(DI, TG, 8); else if (depth >= 4) L_SH(DI, TG, 4); else if (depth >= 2) L_SH(DI, TG, 2); else L_SH(DI, TG, 1); } while (0)
#define L_TG(DI) do { if (tag) L_DP(DI, true); else L_DP(DI, false); } while (0)
        if (use_dict) L_TG(true); else L_TG(false);
    };
    for (int i = 0; i < 3; i++) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; i++) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / iters;
    // ---- check: y[row] = sum of the chunks' partial sums, against the CSR loop in double
    std::vector<T> z(zoff);
    CK(hipMemcpy(z.data(), d_z, zoff * sizeof(T), hipMemcpyDeviceToHost));
    // bitwise: a second launch gives the same bits, and every chunk's row sum is the sequential sum of its rounded products in column order
    // (the order of the CSR loop) -- what a reproducible layout must deliver whatever the wavefronts' timing
    if (!noadd && !getenv("SAME_STREAM")) {
        launch(); CK(hipDeviceSynchronize());
        std::vector<T> z2(zoff);
        CK(hipMemcpy(z2.data(), d_z, zoff * sizeof(T), hipMemcpyDeviceToHost));
        const bool same = memcmp(z.data(), z2.data(), zoff * sizeof(T)) == 0;
        int64_t nord = 0;
        if (mode != 3)
#pragma omp parallel for reduction(+ : nord) schedule(dynamic, 8)
        for (size_t k = 0; k < nch; k++) {
            const auto &c = chunks[k];
            const Panel &pp = pan[c.panel];
            for (int64_t sidx = c.sub0; sidx < c.sub1; sidx++) {
                const int64_t b0 = std::max(pp.sp[sidx], c.e0), b1 = std::min(pp.sp[sidx + 1], c.e1);
                T acc = 0;
                for (int64_t e = b0; e < b1; e++) { const T pr = (T)pp.val[e] * x[(size_t)pb[c.panel] + pp.col[e]]; acc += pr; }
                if (memcmp(&acc, &z[desc[k].zoff + (sidx - c.sub0)], sizeof(T)) != 0) nord++;
            }
        }
        printf("# rerun bitwise %s; partial sums that differ from the column-ordered sum: %ld of %lu\n", same ? "EQUAL" : "DIFFERENT", (long)nord, (unsigned long)zoff);
    }
    std::vector<double> y(A.nrows, 0.0);
    for (uint64_t i = 0; i < zoff; i++) y[zrow[i]] += (double)z[i];
    int64_t bad = 0; double worst = 0;
    if (!noadd)
#pragma omp parallel for reduction(+ : bad) reduction(max : worst)
    for (int64_t r = 0; r < A.nrows; r++) {
        double s = 0, a = 0;
        for (int64_t j = A.rp[r]; j < A.rp[r + 1]; j++) { const double pv = (double)(T)A.va[j] * (double)x[A.ci[j]]; s += pv; a += std::fabs(pv); }
        const double err = std::fabs(y[r] - s), tol = (sizeof(T) == 8 ? 1e-12 : 1e-5) * a + 1e-300;
        if (err > tol) bad++;
        if (a > 0) worst = std::max(worst, err / a);
    }
    const double balg = (double)A.nnz * (sizeof(T) + 4) + (A.nrows + 1) * 4.0 + (double)A.ncols * sizeof(T) + (double)A.nrows * sizeof(T);
    printf("RESULT us %.1f  frac %.3f  (B_alg %.1f MB, %.0f GB/s)  wrong %ld worst %.2e  stream %.0f GB/s\n", us, balg / (us * 1e-6) / 8e12, balg / 1e6, balg / us / 1e3, (long)bad, worst, soff / us / 1e3);
    hipFree(d_stream); hipFree(d_desc); hipFree(d_first); hipFree(d_count); hipFree(d_x); hipFree(d_z); hipFree(d_dict);
    return 0;
}

int main(int argc, char **argv)
{
    if (argc < 10) { fprintf(stderr, "usage: %s csr.bin f32|f64 R W Smax P mode depth iters [dict] [noadd] [rank] [maxdeg] [mindeg]\n", argv[0]); return 1; }
    const bool f32 = !strcmp(argv[2], "f32");
    const Csr  A = read_csr(argv[1], f32);
    const uint32_t R = atoi(argv[3]), W = atoi(argv[4]), Smax = atoi(argv[5]), P = atoi(argv[6]);
    const int mode = atoi(argv[7]), depth = atoi(argv[8]), iters = atoi(argv[9]);
    const bool dict = argc > 10 && atoi(argv[10]) != 0;
    const int  noadd = argc > 11 ? atoi(argv[11]) : 0;
    const int  rank = argc > 12 ? atoi(argv[12]) : 0;          // 1: columns renumbered by popularity (most popular first), rows re-sorted
    const long maxdeg = argc > 13 ? atol(argv[13]) : 0;        // > 0: rows of that many non-zeros or more are emptied (what a long-row kernel would take)
    const long mindeg = argc > 14 ? atol(argv[14]) : 0;        // > 0: rows of fewer non-zeros are emptied
    if (rank || maxdeg || mindeg) {
        Csr &M = const_cast<Csr &>(A);
        if (rank) {
            std::vector<int64_t> cnt(M.ncols, 0);
            for (int64_t i = 0; i < M.nnz; i++) cnt[M.ci[i]]++;
            std::vector<int32_t> order(M.ncols), rk(M.ncols);
            std::iota(order.begin(), order.end(), 0);
            std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return cnt[a] > cnt[b]; });
            for (int64_t i = 0; i < M.ncols; i++) rk[order[i]] = (int32_t)i;
#pragma omp parallel for schedule(dynamic, 1024)
            for (int64_t r = 0; r < M.nrows; r++) {
                const int64_t b = M.rp[r], e = M.rp[r + 1];
                std::vector<std::pair<int32_t, double>> t((size_t)(e - b));
                for (int64_t j = b; j < e; j++) t[(size_t)(j - b)] = {rk[M.ci[j]], M.va[j]};
                std::stable_sort(t.begin(), t.end(), [](const std::pair<int32_t, double> &a, const std::pair<int32_t, double> &c) { return a.first < c.first; });
                for (int64_t j = b; j < e; j++) { M.ci[j] = t[(size_t)(j - b)].first; M.va[j] = t[(size_t)(j - b)].second; }
            }
        }
        if (maxdeg || mindeg) {
            std::vector<int64_t> rp(M.nrows + 1, 0);
            int64_t w = 0;
            for (int64_t r = 0; r < M.nrows; r++) {
                const int64_t b = M.rp[r], e = M.rp[r + 1];
                rp[r] = w;
                if ((maxdeg && e - b >= maxdeg) || (mindeg && e - b < mindeg)) continue;
                for (int64_t j = b; j < e; j++) { M.ci[w] = M.ci[j]; M.va[w] = M.va[j]; w++; }
            }
            rp[M.nrows] = w;
            M.rp = rp; M.nnz = w; M.ci.resize(w); M.va.resize(w);
        }
        printf("# columns %s, rows kept: degree in [%ld, %ld): nnz %ld\n", rank ? "ranked by popularity" : "natural", mindeg, maxdeg ? maxdeg : -1, (long)M.nnz);
    }
    return f32 ? run<float>(A, R, W, Smax, P, mode, depth, iters, dict, noadd) : run<double>(A, R, W, Smax, P, mode, depth, iters, dict, noadd);
}
