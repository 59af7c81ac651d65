/* oracle/ref_shim.h -- TEST INFRASTRUCTURE, container-only.
 *
 * Force-included (-include) when compiling the UNMODIFIED reference translation unit
 * /root/reference/spmv.cpp with g++.  The reference was written for Intel icpc, which
 * accepts four legacy (KNC-era) spellings that g++'s <immintrin.h> does not carry.
 * Each is aliased here to the ISA-identical AVX-512F intrinsic g++ does provide; no
 * behaviour is emulated or stubbed:
 *   _MM_SCALE_8 / _MM_SCALE_4          -> the literal scale factors 8 / 4
 *   _mm512_i32logather_pd(idx,b,s)     -> vgatherdpd on the LOW 8 dword indices of idx
 *   _mm512_permute4f128_epi32(v,perm)  -> vshufi32x4 v,v,perm  (128-bit lane permute)
 * Uses in the reference: spmv.cpp:963, 973-974, 1226-1227 and the four loop copies.
 */
#include <immintrin.h>
#define _MM_SCALE_8 8
#define _MM_SCALE_4 4
#if !defined(__clang__)
#define _mm512_i32logather_pd(idx, base, scale) \
    _mm512_i32gather_pd(_mm512_castsi512_si256(idx), (base), (scale))
#endif
#define _mm512_permute4f128_epi32(v, perm) _mm512_shuffle_i32x4((v), (v), (perm))
