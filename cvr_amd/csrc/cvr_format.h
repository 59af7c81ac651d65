// cvr_format.h -- the CVR64 device format.  Shared by the host planner, the
// device converter and the SpMV kernel.  Re-derives the reference's CVR layout
// (pre_processing, /root/reference/spmv.cpp:565-1014) for 64-lane wavefronts.
//
//   chunk      = 64 lane streams x S steps, handled by ONE wavefront (the reference: one OpenMP
//                thread x 8 AVX-512 lanes, spmv.cpp:584-627).  S is a multiple of 4.
//   slot       = one (column, value) element of a lane stream.  Every matrix row owns
//                max(1, nnz_row) consecutive slots: an EMPTY row owns one pad slot
//                (column = ncols -> x_ext[ncols] == 0, value 0), so that the row a lane writes is
//                implied by the order rows are handed out and no (pos, wb) record list
//                (spmv.cpp:832-834) is needed.
//   segment    = the slots of one row inside one chunk (rows are cut only when longer than the
//                split threshold), plus at most one trailing pad segment that fills the chunk to
//                exactly 64*S slots (the reference pads nnz to 16 instead, spmv.cpp:474-482).
//   group      = 4 consecutive steps of all 64 lanes, stored as
//                  [64 lanes][4 x u32 column words]                  1024 B   (one dwordx4 / lane)
//                  fp64: [2 halves][64 lanes][2 x f64]               2048 B   (two dwordx4 / lane)
//                  fp32: [64 lanes][4 x f32]                         1024 B   (one dwordx4 / lane)
//                value dictionary (matrices with <= 256 distinct values, e.g. pattern matrices):
//                  [64 lanes][4 x u8 code]  256 B after the column words instead of the values; the
//                  dictionary (sorted by bit pattern, contains +0.0 for the pad slots) sits in LDS
//                bit 31 of a column word = "last slot of this lane's current segment"
//                narrow chunks (every chunk of the matrix spans fewer than 32 767 columns: banded matrices):
//                  [64 lanes][4 x u16 column - cbase[k]]  512 B instead of the 1024 B of column words (bit 15 = end
//                  of segment, 0x7fff = the pad column): 10 instead of 12 bytes per fp64 slot
//                column phases: the LAST column word of every piece of a lane stream (a (row, phase) segment or what a lane stole
//                  of one) carries the chunk's row of the piece in bits [col_bits, 31); when column index and row do not fit 31
//                  bits together (wide matrices, long chunks) the rows stand in a block of their own instead (wide row tags):
//                  [64 lanes][4 x u16 row]  512 B between the column words and the values / codes
//   desc[k]    = {row_first, nseg, head_dest, last_dest}: segment q of chunk k writes
//                y_ext[q == 0 ? head_dest : q == nseg-1 ? last_dest : row_first + q]
//   y_ext      = [ y[0..nrows) | dump | carry_head(0), carry_tail(0), carry_head(1), ... ]
//   target[k]  = per lane: the lane it stole from (itself if it never stole); spmv.cpp:900, 982-999
//   shared[]   = rows cut over chunks c0..c1: y[row] = carry_tail(c0) + sum_{c0<c<=c1} carry_head(c)
#pragma once
#include <cstddef>
#include <cstdint>

namespace cvr {

constexpr int      kLanes        = 64;
constexpr uint32_t kEndBit       = 0x80000000u;
constexpr uint32_t kColMask      = 0x7fffffffu;
constexpr uint32_t kHubBit       = 0x40000000u;   // hub table: the column field is an index into the LDS copy of x[hub columns]
constexpr int      kGroupSteps   = 4;
constexpr int      kColsBytes    = kLanes * 16;            // 1024
constexpr int      kGroupBytes64 = kColsBytes + kLanes * 32;  // 3072
constexpr int      kGroupBytes32 = kColsBytes + kLanes * 16;  // 2048
constexpr int      kGroupBytesDict = kColsBytes + kLanes * 4;  // 1280: column words + one code byte per slot
constexpr int      kCols16Bytes  = kLanes * 8;             // 512: narrow chunks store 16-bit column offsets (bit 15 = end of segment, 0x7fff = pad column)
constexpr int      kGroupBytes64C16 = kCols16Bytes + kLanes * 32;   // 2560: 10 bytes per slot
constexpr int      kGroupBytes32C16 = kCols16Bytes + kLanes * 16;   // 1536:  6 bytes per slot
constexpr uint32_t kC16Pad = 0x7fffu;
constexpr int      kTagBytes = kLanes * 8;               // 512: wide row tags (column phases): [64 lanes][4 x u16 row of the chunk] between the column words and the values
constexpr int      kDictMax = 256;
constexpr int      kYStageMax = 4096;   // most row sums a wavefront stages in LDS and writes out coalesced at the end of its chunk (32 KB of fp64)
constexpr int      kWavesPerBlock = 1;   // converter / fix-up launches; the SpMV default: 1 wave per workgroup spreads the chunks most evenly over the CUs (profiles/r01_waves_per_block.log)
constexpr int64_t  kPlanRowBlock = 65536;   // the planner restarts a chunk at every multiple of this many rows (blocks are planned in parallel)
constexpr int      kMaxWavesPerBlock = 16;   // SpMV workgroups of several consecutive chunks share an LDS window of x (cvr_options.waves_per_block)
constexpr size_t   kLdsBytes = 160 * 1024;   // LDS of one gfx950 CU
// gang chunks (cvr_options.gang; cvr_spmv.hip: spmv_gang_kernel): the chunks of an interleaved workgroup sorted TOGETHER -- element e of the gang's list
// sorted by (column, position) stands in group e / 256 of the gang's stream (the chunks' allocations, one behind the other) -- and walked by the workgroup's
// wavefronts in turn, units of kGangUnit groups each; a slot's row tag = chunk inside the gang * accumulators per chunk + row inside the chunk (kGangTagBits
// bits), and its column word holds the column's offset from the group's first (smallest) column in the kGangOffBits bits below the tag (gbase[] keeps the
// groups' first columns); with 16-bit tags of their own the column word keeps the panel's column
constexpr int      kGangOffBits = 17, kGangTagBits = 15;
constexpr int      kGangUnit = 2;        // groups a wavefront takes in a row (their products wait in registers until the unit's turn: the token)
constexpr int      kIlvMaxSteps = 576;      // longest interleaved chunk: the converter sorts a chunk's 64 S (column, position) pairs in one workgroup's LDS, up to 36 per thread (cvr_ilv.hip)

inline int group_bytes(bool f32, bool dict = false, bool c16 = false, bool tag16 = false) { return (dict ? kGroupBytesDict : c16 ? (f32 ? kGroupBytes32C16 : kGroupBytes64C16) : f32 ? kGroupBytes32 : kGroupBytes64) + (tag16 ? kTagBytes : 0); }

struct Shared { int64_t row, c0, c1; };

}  // namespace cvr
