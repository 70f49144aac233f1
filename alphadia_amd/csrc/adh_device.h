// Device-side views shared by the kernels and the C-ABI host layer.
// Plain structs passed to kernels by value; every pointer is a device pointer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/alphadia_hip.h"

#define ADH_WAVE 64

// Run staged in HBM: AlphaRawJIT arrays (alpharaw_jit.py:78-138) + our m/z bucket index.
struct DevRun {
    const float *mz;          // [n_peaks]
    const float *intensity;   // [n_peaks]
    const int64_t *pstart;    // [n_spectra]
    const int64_t *pstop;     // [n_spectra]
    const float *rt;          // [n_spectra]
    const float *mobility;    // [n_mobility]
    const double *cycle;      // [cycle_len * cycle_scans * 2]
    const int32_t *ms1_obs;   // rows of the cycle selected by quadrupole (-1,-1)
    const uint32_t *bucket;   // [n_spectra * (n_buckets + 1)] offsets relative to pstart
    int64_t n_spectra;
    int64_t n_peaks;
    int32_t cycle_len;
    int32_t cycle_scans;
    int32_t n_ms1_obs;
    int32_t n_buckets;
    float bucket_min;         // m/z of bucket 0
    float bucket_inv_width;   // 1 / width
};

// FragmentContainer arrays (fragment_container.py:11-46)
struct DevLib {
    const float *mz_library;
    const float *mz;
    const float *intensity;
    const uint8_t *type;
    const uint8_t *loss_type;
    const uint8_t *charge;
    const uint8_t *number;
    const uint8_t *position;
    const uint8_t *cardinality;
    int64_t n;
};

// Candidate SoA (score_group.py:145-229)
struct DevCands {
    int64_t n;
    const uint32_t *precursor_idx;
    const uint8_t *rank;
    const uint8_t *flags;
    const uint32_t *frag_start;
    const uint32_t *frag_stop;
    const int64_t *scan_start;
    const int64_t *scan_stop;
    const int64_t *scan_center;
    const int64_t *frame_start;
    const int64_t *frame_stop;
    const int64_t *frame_center;
    const uint8_t *charge;
    const float *precursor_mz;
    const float *isotope_intensity;
    const uint32_t *order;    // processing permutation or nullptr
    int32_t n_isotope_cols;
};

// OutputPsmDF (output.py:17-70); same member order as adh_output_t
typedef adh_output_t DevOut;

// LDS capacities for one launch (maxima over the batch, from the plan kernel)
struct Caps {
    int32_t n_lib;   // longest library slice
    int32_t k;       // fragments kept (<= top_k)
    int32_t o;       // observations
    int32_t f;       // cycles
    int32_t i;       // isotopes
    int32_t stop_phase;  // developer ablation switch (0 = run everything)
};

// monotone bucket function shared by index build and lookup
__host__ __device__ inline int adh_bucket_of(float mz, float bmin, float binv, int nb) {
    float t = (mz - bmin) * binv;
    if (!(t > 0.0f)) return 0;
    if (t >= (float)nb) return nb;
    return (int)t;
}
