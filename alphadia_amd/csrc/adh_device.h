// Device-side views shared by the kernels and the C-ABI host layer.
// Plain structs passed to kernels by value; every pointer is a device pointer.
//
// HBM layout:
//   entries uint2[n_peaks]             the run transposed: sorted by (cycle block, cycle row, m/z bin,
//                                      cycle, m/z); (cycle in block << 9 | low m/z bits, intensity)
//   tab     uint32[blocks*rows*bins+1] first entry of every (block, row, bin).  4-byte entries on
//                                      purpose: an 8-byte self-describing table (single entries
//                                      inline) measured 20 % slower in the gather (bigger footprint)
//   lib     LibRec[n_fragments]        32-byte fragment records (one vector load per lane)
//   plan    CandRec[n_candidates]      80-byte candidate records in PROCESSING order
//   scratch per-candidate blocks       selected fragments + XIC tile, written by the gather
//                                      kernel, read by the feature kernel
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/alphadia_hip.h"

#define ADH_WAVE 64
#define ADH_MAX_OBS 8
#define ADH_MAX_MS1_OBS 16

// m/z bin of the transposed run = float32 bit pattern >> ADH_BIN_SHIFT: exactly monotone,
// relative width 2^-14 (30.5 .. 61 ppm)
#ifndef ADH_BIN_SHIFT
#define ADH_BIN_SHIFT 9
#endif
// Cycles are cut into blocks of 2^block_shift cycles (the granularity of the bin table) and blocks into
// groups of ADH_SUB blocks (the granularity of the sort): inside a (group, cycle row, m/z bin) the entries
// of consecutive blocks follow each other, so the two or three blocks a candidate spans are ONE run of
// entries per bin and their table words are neighbours - half the random line fetches of a layout sorted
// block by block (what bounds the gather).
#define ADH_SUB_SHIFT 3
#define ADH_SUB (1 << ADH_SUB_SHIFT)

struct DevRun {
    // peaks sorted by (group of ADH_SUB blocks, cycle row, m/z bin, cycle, m/z):
    // .x = (cycle inside the GROUP << ADH_BIN_SHIFT) | low m/z bits, .y = intensity bits
    const uint2 *entries;
    // [((group * cycle_len + row) * n_bins + bin) * ADH_SUB + block in group + group] (adh_tab_row): first entry of
    // the (group, row, bin) that lies in that block or a later one; the word after a bin's last block is the
    // next bin's first, so entries of blocks sb0 .. sb1 of a bin are [t[bin * ADH_SUB + sb0], t[bin * ADH_SUB + sb1 + 1])
    const uint32_t *tab;
    // Round 4: table words are RELATIVE to the first entry of their group of blocks (grp_entry0[group], 64-bit), so
    // that a run may hold 2^32 peaks and more (a group holds ~1e7); every group's segment of the table ends with one
    // extra word, the group's entry count, which is what the last cell's "next word" reads.
    const int64_t *grp_entry0;  // [n_blocks / ADH_SUB + 1]
    const float *rt;          // [n_spectra]
    const float *mobility;    // [n_mobility]
    const double *cycle;      // [cycle_len * cycle_scans * 2]
    const int32_t *ms1_obs;   // rows of the cycle selected by quadrupole (-1,-1)
    int64_t n_spectra;
    int64_t n_peaks;
    int32_t cycle_len;
    int32_t cycle_scans;
    int32_t n_ms1_obs;
    int32_t n_bins;           // m/z bins per (block, row)
    int32_t bin0;             // bit pattern >> ADH_BIN_SHIFT of the smallest m/z
    int32_t block_shift;      // log2(cycles per block)
    int32_t n_blocks;         // blocks, padded to whole groups
    float mz_min, mz_max;     // smallest / largest m/z of the run
};

// one library fragment (fragment_container.py:11-46), 32 bytes
struct __attribute__((aligned(16))) LibRec {
    float mz_library;
    float mz;
    float intensity;
    uint8_t type, loss_type, charge, number;
    uint8_t position, cardinality, pad0, pad1;
    uint32_t pad2[3];
};
static_assert(sizeof(LibRec) == 32, "LibRec must be 32 bytes");

// one candidate in processing order (score_group.py:145-229 columns + the plan)
struct __attribute__((aligned(16))) CandRec {
    uint32_t precursor_idx, frag_start, frag_stop;
    int32_t frame_start, frame_stop, frame_center;
    int32_t scan_start, scan_stop, scan_center;
    float precursor_mz;
    uint8_t charge, rank, flags, n_obs;
    uint16_t obs[ADH_MAX_OBS];  // cycle rows overlapping the precursor isolation range
    uint32_t row;               // row of the output tables
    uint64_t scratch_off;       // byte offset of this candidate's scratch block
    uint32_t k_cap;             // min(top_k, library slice length)
    uint32_t pad;
};
static_assert(sizeof(CandRec) == 80, "CandRec must be 80 bytes");

// scratch block of one candidate (all sizes multiples of 32 bytes):
//   [0, 32)                      header: uint32 K (0 = failed before the gather), uint32 hits
//   [32, 32 + k_cap * 32)        selected fragments, ascending m/z (LibRec)
//   then  float2[k_cap * O * F]  fragment cells, index ((o * F + f) * K + k)
//   then  float2[I * F]          precursor cells, index (i * F + f)
__host__ __device__ inline uint64_t adh_scratch_frag_off(uint32_t k_cap) {
    return 32 + (uint64_t)k_cap * 32;
}
__host__ __device__ inline uint64_t adh_scratch_prec_off(uint32_t k_cap, int O, int F) {
    return adh_scratch_frag_off(k_cap) + (uint64_t)k_cap * O * F * 8;
}
__host__ __device__ inline uint64_t adh_scratch_bytes(uint32_t k_cap, int O, int F, int I) {
    uint64_t b = adh_scratch_prec_off(k_cap, O, F) + (uint64_t)I * F * 8;
    return (b + 31) / 32 * 32;
}

// Ion-mobility run staged in HBM: TimsTOFTransposeJIT arrays (bruker_jit.py:22-137)
struct DevTims {
    const int64_t *tof_indptr;   // [n_tof + 1]
    const uint32_t *push;        // [n_events] frame * scan_max + scan, ascending inside a TOF bin
    const uint16_t *inten;       // [n_events]
    const double *mz;            // [n_tof]
    const double *cycle;         // [cycle_len * scan_max * 2]
    const int32_t *dpc;          // [cycle_len * scan_max] dia_precursor_cycle
    const double *rt;            // [n_frames]
    const double *mobility;      // [scan_max]
    int64_t n_tof, n_events, n_frames;
    int32_t cycle_len, scan_max, zeroth;
    // Search indices built when the run is staged (adh_index_im.hip); NULL when they do not fit.
    //   mz_lut[b]  = first TOF bin with mz >= lut_min + b / lut_inv_step        (lut_n + 1 entries)
    //   cyc_idx[cb * n_tof + tof] = first event of the bin with push >= the first push of cycle
    //   cb << cyc_shift, counted from the bin's first event (cyc_cols = blocks + 1; the last column is the end of
    //   the bin's cycles).  Column-major since round 3: the TOF bins of an m/z window are neighbours and the
    //   lanes that look them up take ONE 128-byte line per column, where the row-major table cost a line per bin
    //   (the ion-mobility gather is bound by the lines it makes HBM fill: 35 GB per 600 000 candidates at
    //   4.8 TB/s, for 8 GB of events, index words and rows it needs)
    // They replace the two ~19-step searches over mz and the two ~10-step searches over a bin's pushes,
    // i.e. most of the dependent-load chain of a candidate, by one load each.
    const uint32_t *mz_lut;
    const uint32_t *cyc_idx;
    double lut_min, lut_inv_step;
    int32_t lut_n, cyc_shift, cyc_cols, n_cycles;
    __device__ __forceinline__ uint32_t cyc_word(int tof, int col) const { return cyc_idx[(size_t)col * (size_t)n_tof + (size_t)tof]; }
    // Second copy of the events for the scoring gather (round 4), in TILE order: a tile is a block of
    // 2^tile_cshift cycles x 2^tile_sshift scans (tile = cycle block * tile_sblocks + scan block), its events are
    // sorted by TOF bin, then push.  In the TOF-major order above a candidate reads, per TOF bin of a window, all
    // scans and quadrupole windows of its cycles (3 % of what it reads is its own) and every bin costs its own
    // lines; here the bins of a window are neighbours INSIDE a tile, so a (window, tile) pair is one short
    // contiguous run and a third of it is the candidate's.
    //   tile_ev[e]  = {frame << tile_sbits | scan, intensity | (TOF bin & 0xFFFF) << 16}   (frame and scan by a
    //   shift and a mask where the TOF-major push needs a division; tile_sbits = bits of scan_max - 1)
    //   tile_idx[tile * (n_tof + 1) + tof] = first event of the tile with TOF bin >= tof (absolute; the layout is
    //   only built for runs below 2^31 events)
    // NULL when the layout is not built (ADH_IM_TILED=0, no room, too many events): the kernels use the bin ranges.
    const uint2 *tile_ev;
    const uint32_t *tile_idx;
    // Round 6: with tile_frames = cycle_len the frame of the cycle is part of the tile (tile = (cycle block *
    // tile_sblocks + scan block) * tile_frames + frame inside the cycle): a candidate reads the frames its quadrupole
    // windows are in - one of nine in configs[3] - instead of all of them, at the price of a tile_frames times larger
    // index for the same (cycles x scans) footprint: HBM spent on an index (6.6 GB on configs[3]).  1: frames share a tile.
    int32_t tile_cshift, tile_sshift, tile_cblocks, tile_sblocks, tile_sbits, tile_frames;
    __device__ __forceinline__ uint32_t tile_word(int tile, int tof) const { return tile_idx[(size_t)tile * (size_t)(n_tof + 1) + (size_t)tof]; }
};

// candidate record of the ion-mobility plan (processing order)
struct __attribute__((aligned(16))) CandRecIM {
    uint32_t precursor_idx, frag_start, frag_stop;
    int32_t frame_start, frame_stop, frame_center;
    int32_t scan_start, scan_stop, scan_center;
    float precursor_mz;
    uint8_t charge, rank, flags, n_obs;
    uint16_t obs[ADH_MAX_OBS];   // sorted unique dia_precursor_cycle values hit by the fragment quad range
    uint8_t n_ms1, pad8[3];
    uint16_t ms1_obs[ADH_MAX_MS1_OBS];  // the same for the (-1, -1) precursor query
    uint32_t row, k_cap;
    uint64_t scratch_off;
    // frames of the cycle (bit fr) with a push of the candidate's scans whose quadrupole window meets the fragment range /
    // is the MS1 window (-1, -1): the tiles the gather visits when the tile layout is keyed by frame (cycle_len <= 32)
    uint32_t frames_f, frames_p;
};
static_assert(sizeof(CandRecIM) == 128, "CandRecIM must be 128 bytes");

// ion-mobility scratch block: header (32 B; [0] K, [1] hits, [2] entry count, [3] mode, [4] precursor
// entry count), selected fragments (k_cap x 32 B), fragment cells float2[k_cap][O][S][F], precursor cells
// float2[I][Op][S][F].  Ion-mobility tiles are ~1 % full, so the gather kernel normally never
// materialises the tiles (mode ADH_IM_MODE_COMPACT): the non-zero cells leave as (cell, intensity, m/z)
// entries sorted by cell - the reference's summation order - written where the tiles would be:
// header[2] fragment entries (cell = ((k * O + o) * S + scan) * F + cycle), then header[4] precursor
// entries with the MS1 rows already collapsed (cell = (scan * F + cycle) * I + isotope).  A candidate
// with more than ADH_IM_SORT_CAP events in one window (or in its isotope windows together), more than
// ADH_IM_PAIR_CAP TOF bins in all windows, or more entries than fit where the tiles would be (a small
// tile full of signal) gets the dense tiles (zero fill + scatter; mode ADH_IM_MODE_DENSE).
#define ADH_IM_SORT_CAP 512
#ifndef ADH_IM_PAIR_CAP
#define ADH_IM_PAIR_CAP 256  // (a build switch since round 6: LDS per wavefront of the gathers)
#endif
#define ADH_IM_MODE_DENSE 0u
#define ADH_IM_MODE_COMPACT 1u
struct ImEntry {
    uint32_t cell;
    float x, y;     // accumulated intensity, running intensity-weighted m/z
};
__host__ __device__ inline uint64_t adh_im_prec_off(uint32_t k_cap, int O, int S, int F) {
    return adh_scratch_frag_off(k_cap) + (uint64_t)k_cap * O * S * F * 8;
}
__host__ __device__ inline uint64_t adh_im_tiles_end(uint32_t k_cap, int O, int S, int F, int I, int Op) {
    uint64_t b = adh_im_prec_off(k_cap, O, S, F) + (uint64_t)I * Op * S * F * 8;
    return (b + 31) / 32 * 32;
}
__host__ __device__ inline uint64_t adh_im_scratch_bytes(uint32_t k_cap, int O, int S, int F, int I, int Op) {
    return adh_im_tiles_end(k_cap, O, S, F, I, Op);
}

// Hand-over record of the split ion-mobility feature path (round 4): adh_feature_im_kernel<LAY, true> stops
// after its passes over the tiles and leaves, per candidate, what the profile phase
// (adh_features_im2.hip: four candidates per wavefront) needs - a few KB of profiles instead of the tile.
// Rows are zero-padded to the capacities, the frame axis is stored CENTRED (entry r <-> cycle r - FM/2 + F/2),
// so the reader takes whole rows with vector loads and constant register indices.
#define ADH_IM_PROF_K 12
template <int FM, int SM, int NO = 1>
struct __attribute__((aligned(16))) ImProfRec {
    uint32_t K0, pad0[3];
    double ohe[ADH_IM_PROF_K][NO], omz[ADH_IM_PROF_K][NO];  // weighted centre means of the (fragment, observation) planes
    double hp[4], omzp[4];                                  // ... of the isotope planes
    float tsum[NO], pad1[4 - NO];                           // template sums
    float spi[4], iso_int[4], iso_mz[4];
    float tfp_raw[NO][FM];                                  // template frame profiles (sum over scans), centred
    float tsp_raw[NO][SM];                                  // template scan profiles (sum over cycles)
    float ffp[ADH_IM_PROF_K][NO][FM];                       // fragment frame profiles before the presence mask, centred
    float fsp[ADH_IM_PROF_K][NO][SM];                       // fragment scan profiles
};

// Barrier of a ONE-wavefront block whose lanes talk through LDS.  A wavefront issues its LDS
// instructions in order and the LDS unit completes them in order, so a read that follows a write in
// program order sees the write, whichever lane made it: nothing has to be waited for.  What is
// needed is that the compiler keeps the order - the two wavefront-scope fences (no instruction) and the
// scheduling barrier do that.  __syncthreads() adds s_waitcnt vmcnt(0) lgkmcnt(0) + s_barrier, i.e.
// every global load in flight has to land first: a prefetch issued before a hand-over point would be
// waited for right there.  (Measured: by itself the swap changes nothing - the kernels are bound by
// dependent LDS / ALU chains, not by the barriers - it only keeps prefetches alive.)  NOT for data
// handed over through global memory.
__device__ __forceinline__ void adh_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// table row of (group of blocks, cycle row): index it with bin * ADH_SUB + block inside the group
__device__ __forceinline__ const uint32_t *adh_tab_row(const DevRun &run, int row, int blk) {
    const int64_t grp = blk >> ADH_SUB_SHIFT;
    return run.tab + ((grp * run.cycle_len + row) * (int64_t)run.n_bins << ADH_SUB_SHIFT) + grp;  // (+ one spare word per group)
}
// entries of a group of blocks: table words count from here
__device__ __forceinline__ const uint2 *adh_group_entries(const DevRun &run, int blk) {
    return run.entries + run.grp_entry0[blk >> ADH_SUB_SHIFT];
}

typedef adh_output_t DevOut;

// parameters of the quadrupole transfer function (SimpleQuadrupoleJit.predict, quadrupole.py:94-113)
struct QuadParams {
    double sigma_lo, sigma_hi, delta_lo, delta_hi;
};
__host__ __device__ inline QuadParams adh_quad_params(const adh_scoring_config_t &c) {
    QuadParams q;
    const bool set = c.quadrupole_sigma[0] > 0.0 && c.quadrupole_sigma[1] > 0.0;
    q.sigma_lo = set ? c.quadrupole_sigma[0] : 0.2;
    q.sigma_hi = set ? c.quadrupole_sigma[1] : 0.2;
    q.delta_lo = set ? c.quadrupole_delta_mu[0] : 0.0;
    q.delta_hi = set ? c.quadrupole_delta_mu[1] : 0.0;
    return q;
}

// LDS capacities of one launch (maxima over the launch's candidates)
struct Caps {
    int32_t k;       // fragments kept (<= top_k)
    int32_t o;       // observations
    int32_t f;       // cycles
    int32_t i;       // isotopes
    int32_t n_lib;   // longest library slice (gather kernel scratch)
    int32_t stop_phase;  // developer ablation switch (0 = run everything)
    int32_t s;       // scans (ion-mobility kernels only)
    int32_t op;      // MS1 observations (ion-mobility kernels only)
    // adh_debug_get_dense only (stop_phase == ADH_DEBUG_DENSE): explicit quadrupole range of the
    // query instead of the one derived from the precursor isotopes; no "<= 3 fragments" exit
    float dbg_q_lo, dbg_q_hi;
    int dbg_drop_dense;  // developer ablation (ADH_DEBUG_IM_DROP_DENSE): ion-mobility candidates that overflow the sparse lists are dropped
};
#define ADH_DEBUG_DENSE 9

