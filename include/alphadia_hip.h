/*
 * alphadia_hip.h - C ABI of libalphadia_hip.so, the MI355X-native drop-in for
 * alphaDIA's peptide-centric candidate scoring hot path.
 *
 * Nothing like this exists in the reference (it is pure Python + Numba); each
 * entry point below names the reference interface whose work it replaces.
 * All paths are relative to the MannLabs/alphadia source tree.
 *
 * Conventions
 *   - plain pointers and sizes only; no torch / numpy types cross this ABI
 *   - every function returns ADH_OK (0) or a negative error code; the message
 *     is available from adh_last_error() (thread local)
 *   - per-candidate soft failures are DATA (valid[i] == 0), never errors,
 *     exactly like `Candidate.failed` (search/scoring/containers/candidate.py:190-325)
 *   - host buffers are owned by the caller; device buffers live behind the
 *     opaque handle until adh_destroy()
 *   - a handle is bound to one GPU and is not re-entrant (one host thread per GPU)
 */
#ifndef ALPHADIA_HIP_H
#define ALPHADIA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ADH_OK 0
#define ADH_ERR_INVALID_ARGUMENT (-1)
#define ADH_ERR_HIP (-2)
#define ADH_ERR_NOT_STAGED (-3)
#define ADH_ERR_UNSUPPORTED (-4)
#define ADH_ERR_OUT_OF_MEMORY (-5)

#define ADH_NUM_FEATURES 46 /* alphadia/constants/settings.py:5 */

/* candidate flag bits */
#define ADH_FLAG_SKIP 1u /* score group without its reference channel: score_group.py:50-64 */

/*
 * Raw data of a non-ion-mobility run: the fields of AlphaRawJIT
 * (search/jitclasses/alpharaw_jit.py:78-138) that the scoring path reads.
 * Peaks are CSR by spectrum, m/z ascending inside each spectrum.
 */
typedef struct adh_alpharaw {
    const double *cycle;            /* (1, cycle_len, cycle_scans, 2) float64, C order */
    int32_t cycle_len;              /* cycle.shape[1] */
    int32_t cycle_scans;            /* cycle.shape[2] (1 for AlphaRaw) */
    const float *rt_values;         /* [n_spectra] seconds */
    int64_t n_spectra;
    const float *mobility_values;   /* [n_mobility]; {1e-6, 0} for AlphaRaw */
    int64_t n_mobility;
    const int64_t *peak_start_idx;  /* [n_spectra] */
    const int64_t *peak_stop_idx;   /* [n_spectra] */
    const float *mz_values;         /* [n_peaks] */
    const float *intensity_values;  /* [n_peaks] */
    int64_t n_peaks;
} adh_alpharaw_t;

/*
 * Raw data of an ion-mobility run in the transposed (TOF-major) layout: the fields of
 * TimsTOFTransposeJIT (search/jitclasses/bruker_jit.py:22-137) that the scoring path reads.
 * push index = frame * scan_max_index + scan; events are ascending in push inside a TOF bin.
 */
typedef struct adh_timstof {
    const double *cycle;                /* (1, cycle_len, scan_max_index, 2) float64 */
    int32_t cycle_len;
    int32_t scan_max_index;
    const int64_t *dia_precursor_cycle; /* [cycle_len * scan_max_index] cycle row of a push */
    const double *rt_values;            /* [n_frames] */
    int64_t n_frames;
    const double *mobility_values;      /* [scan_max_index] */
    const double *mz_values;            /* [n_tof] m/z of a TOF index, ascending */
    int64_t n_tof;
    const int64_t *tof_indptr;          /* [n_tof + 1] */
    const uint32_t *push_indices;       /* [n_events] */
    const uint16_t *intensity_values;   /* [n_events] */
    int64_t n_events;
    int32_t zeroth_frame;               /* 1 when frame 0 is the empty alphatims frame */
} adh_timstof_t;

/*
 * Flat fragment library: the nine arrays of FragmentContainer
 * (search/jitclasses/fragment_container.py:11-46) as assembled by
 * CandidateScoring.assemble_fragments (search/scoring/scoring.py:355-392).
 */
typedef struct adh_fragments {
    int64_t n;
    const float *mz_library;
    const float *mz;          /* the configured fragment m/z column (library or calibrated) */
    const float *intensity;
    const uint8_t *type;
    const uint8_t *loss_type;
    const uint8_t *charge;
    const uint8_t *number;
    const uint8_t *position;
    const uint8_t *cardinality;
} adh_fragments_t;

/*
 * Candidate table in struct-of-arrays form: the columns
 * ScoreGroupContainer.build_from_df receives
 * (search/scoring/containers/score_group.py:145-229), already in
 * score-group order; row i of every output belongs to candidate i.
 */
typedef struct adh_candidates {
    int64_t n;
    const uint32_t *precursor_idx;
    const uint8_t *rank;
    const uint8_t *flags;            /* ADH_FLAG_* or NULL */
    const uint32_t *frag_start_idx;
    const uint32_t *frag_stop_idx;
    const int64_t *scan_start;
    const int64_t *scan_stop;
    const int64_t *scan_center;
    const int64_t *frame_start;
    const int64_t *frame_stop;
    const int64_t *frame_center;
    const uint8_t *charge;
    const float *precursor_mz;
    const float *isotope_intensity;  /* [n, n_isotope_cols] row major */
    int32_t n_isotope_cols;
} adh_candidates_t;

/* CandidateScoringConfigJIT (search/scoring/config.py:13-60) */
typedef struct adh_scoring_config {
    int32_t collect_fragments;
    int32_t score_grouped;
    int32_t exclude_shared_ions;
    uint32_t top_k_fragments;
    uint32_t top_k_isotopes;
    int32_t reference_channel;
    uint32_t quant_window;
    int32_t quant_all;
    float precursor_mz_tolerance;
    float fragment_mz_tolerance;
    int32_t experimental_xic;
    /* SimpleQuadrupoleJit.sigma / .delta_mu (search/scoring/quadrupole.py:72-76,110-113): the quadrupole
     * transfer function of a fitted calibration, logistic(x, lower + delta_mu[0], sigma[0]) -
     * logistic(x, upper + delta_mu[1], sigma[1]).  sigma[0] <= 0 (a zeroed struct) stands for the class
     * defaults the reference workflow runs with: sigma 0.2, delta_mu 0. */
    double quadrupole_sigma[2];
    double quadrupole_delta_mu[2];
} adh_scoring_config_t;

/*
 * OutputPsmDF (search/scoring/output.py:17-70): `valid`, `precursor_idx`,
 * `rank`, `features[n,46]` and thirteen per-fragment tables [n, top_k].
 * Buffers must be zero-initialised by the caller for the *_device entry
 * point; the host entry point zero-fills them itself.
 */
typedef struct adh_output {
    int64_t n;
    int32_t top_k;
    uint8_t *valid;
    uint32_t *precursor_idx;
    uint8_t *rank;
    float *features;               /* [n, 46] */
    uint32_t *fragment_precursor_idx;
    uint8_t *fragment_rank;
    float *fragment_mz_library;
    float *fragment_mz;
    float *fragment_mz_observed;
    float *fragment_height;
    float *fragment_intensity;
    float *fragment_mass_error;
    float *fragment_correlation;
    uint8_t *fragment_position;
    uint8_t *fragment_number;
    uint8_t *fragment_type;
    uint8_t *fragment_charge;
    uint8_t *fragment_loss_type;
    uint32_t *stat_matched_peaks;  /* optional [n]: peaks accumulated by get_dense, or NULL */
    /* optional [n, top_k], or NULL: 1 + position of the slot's fragment inside the candidate's library
     * slice [frag_start_idx, frag_stop_idx), 0 for an empty slot.  With it the library columns of the
     * fragment tables (mz_library, mz, position, number, type, charge, loss_type) can be rebuilt from
     * the staged library, so they need not travel in the all-gather (alphadia_amd/distributed.py). */
    uint16_t *fragment_lib_slot;
} adh_output_t;

typedef struct adh_handle adh_handle_t;

/* Thread-local message of the last failing call. */
const char *adh_last_error(void);

/* Number of visible HIP devices. */
int adh_device_count(int *count);

/* Create / destroy a per-GPU context. */
int adh_create(adh_handle_t **handle, int device);
int adh_destroy(adh_handle_t *handle);

/*
 * Stage the run in HBM once (replaces DiaData.to_jitclass(),
 * raw_data/alpharaw_wrapper.py:124-142, as consumed at scoring.py:639).
 * The run is kept as a time-major transposed copy (sorted by cycle block, cycle
 * row, m/z bin, cycle, m/z) that replaces the reference's per-spectrum binary
 * search (alpharaw_jit.py:53-64,293-297): the XIC of a fragment is contiguous.
 */
int adh_stage_alpharaw(adh_handle_t *handle, const adh_alpharaw_t *dia);

/*
 * Stage an ion-mobility run (replaces TimsTOFTranspose.to_jitclass(),
 * raw_data/bruker.py:119-152).  A handle holds ONE run: staging either layout
 * replaces the previous one.
 */
int adh_stage_timstof(adh_handle_t *handle, const adh_timstof_t *dia);

/* Stage the flat fragment library (replaces assemble_fragments, scoring.py:355-392).  Besides the copy in HBM
 * (32 bytes per fragment) the handle keeps a host copy of the packed records (another 32 bytes per fragment of
 * host memory, per handle = per rank): adh_score_candidates rebuilds the library columns of the fragment tables
 * from it on the host instead of copying them over PCIe.  ADH_DEBUG_COPY_ALL=1 does without the rebuild. */
int adh_stage_fragments(adh_handle_t *handle, const adh_fragments_t *fragments);

/*
 * Score candidates: host table in, host OutputPsmDF out (upload + kernels +
 * download).  Replaces the pjit loop `_process_score_groups`
 * (scoring.py:114-137,634-643), i.e. ScoreGroup.process -> Candidate.process
 * for every candidate.  `out` buffers are zero-filled by the call.
 */
int adh_score_candidates(adh_handle_t *handle, const adh_candidates_t *candidates,
                         const adh_scoring_config_t *config, adh_output_t *out);

/*
 * The same call for the DataFrame operator: only what `collect_candidates` / `collect_fragments`
 * (search/scoring/scoring.py:394-467,520-580; output.py:89-97) keep of the padded tables comes back - the VALID
 * candidates (`valid`), column by column, and the FILLED fragment slots (`fragment_mz_library > 0`), in the order
 * the padded tables hold them.  Per chunk of the pipeline the device counts and scans, a pack kernel writes the
 * compacted columns - features transposed to [feature][row], library columns of a slot from the staged library -
 * densely into a device block, ONE DMA copy of exactly the used bytes moves the block into a page-locked twin of the
 * handle, and host threads unpack finished blocks into the caller's arrays (which may be pageable) while later
 * chunks are scored.  No padded host table, no host pass over invalid rows or empty slots: 1.11 GB instead of
 * 1.35 GB cross PCIe per 3 M candidates and the DataFrame operator takes 85 ms instead of 150.
 *
 * Capacities: `rows_capacity` >= number of valid candidates, `slots_capacity` >= number of filled slots
 * (candidates and candidates * top_k always suffice).  A call that would overflow either writes nothing beyond
 * them, sets n_rows / n_slots to what it needs and fails with ADH_ERR_INVALID_ARGUMENT.
 * The padded device tables are left in HBM as by adh_score_candidates (adh_get_device_tables, the FDR stage).
 */
typedef struct adh_compact_output {
    int64_t rows_capacity;          /* in */
    int64_t slots_capacity;         /* in */
    int32_t top_k;                  /* in: fragment slots per candidate (width of the padded tables) */
    int32_t reserved;
    int64_t n_rows;                 /* out: valid candidates */
    int64_t n_slots;                /* out: filled fragment slots */
    /* per valid candidate, ascending candidate row */
    uint32_t *row;                  /* [rows_capacity] row of the candidate in adh_candidates_t */
    uint32_t *precursor_idx;        /* [rows_capacity] */
    uint8_t *rank;                  /* [rows_capacity] */
    float *features;                /* [46][rows_capacity]: feature f of the j-th valid candidate at f * rows_capacity + j */
    /* per filled fragment slot, ascending (candidate row, slot) */
    uint32_t *fragment_row;         /* [slots_capacity] row of the slot's candidate in adh_candidates_t */
    uint32_t *fragment_precursor_idx;
    uint8_t *fragment_rank;
    float *fragment_mz_library;
    float *fragment_mz;
    float *fragment_mz_observed;
    float *fragment_height;
    float *fragment_intensity;
    float *fragment_mass_error;
    float *fragment_correlation;
    uint8_t *fragment_position;
    uint8_t *fragment_number;
    uint8_t *fragment_type;
    uint8_t *fragment_charge;
    uint8_t *fragment_loss_type;
} adh_compact_output_t;

int adh_score_candidates_compact(adh_handle_t *handle, const adh_candidates_t *candidates,
                                 const adh_scoring_config_t *config, adh_compact_output_t *out);

/*
 * The same work split so that tables can stay in HBM:
 *   adh_upload_candidates  - copy the candidate SoA to the GPU (kept in the handle)
 *   adh_score_uploaded     - enqueue the kernels on `hip_stream` (a hipStream_t taken
 *                            literally: NULL is HIP's default stream; use
 *                            adh_get_stream for the handle's own stream) writing into
 *                            DEVICE buffers `out_device` (zero-initialised by the
 *                            caller ON THAT STREAM, on the handle's GPU); does not
 *                            synchronise.
 * Used when the per-GPU tables are reassembled with an RCCL all-gather before
 * they leave HBM.
 */
int adh_upload_candidates(adh_handle_t *handle, const adh_candidates_t *candidates);
int adh_score_uploaded(adh_handle_t *handle, const adh_scoring_config_t *config,
                       adh_output_t *out_device, void *hip_stream);

/*
 * Device view of the tables the last adh_score_candidates call filled (they stay in HBM until the
 * next call): the hand-over to a following on-device stage (classifier, q-values, fragment
 * competition; alphadia/workflow/peptidecentric/peptidecentric.py:219-243 hands DataFrames over
 * on the host).  device_view->n is the number of live rows.
 */
int adh_get_device_tables(adh_handle_t *handle, adh_output_t *device_view);

/*
 * Layout of the packed device buffer that holds the tables of `rows` candidates (the buffer
 * adh_get_device_tables / adh_comm_gathered give views of): one entry per OutputPsmDF column
 * (alphadia/search/scoring/output.py:17-97) in buffer order, every table 256-byte aligned.  The computed
 * tables come first and form the contiguous "wire" prefix of `wire_bytes` bytes - what the all-gather of
 * adh_comm_init and the D2H copies of adh_score_candidates move; the columns behind it (wire = 0) repeat the
 * candidate table or the staged library and are rebuilt where they are needed.  Pure function: needs no
 * GPU.  fields may be NULL to query n_fields / the sizes only.
 */
typedef struct adh_table_field {
    char name[40];
    uint64_t offset;      /* bytes from the start of the buffer */
    uint32_t row_elems;   /* elements per candidate row: 1, 46 or top_k */
    uint32_t elem_bytes;
    int32_t wire;
    int32_t reserved;
} adh_table_field_t;
int adh_table_layout(int64_t rows, int32_t top_k, int32_t capacity, adh_table_field_t *fields, int32_t *n_fields,
                     uint64_t *total_bytes, uint64_t *wire_bytes);
/* zero those tables on `hip_stream` (for callers that refill them with adh_score_uploaded) */
int adh_zero_device_tables(adh_handle_t *handle, void *hip_stream);

/*
 * Page-locked host memory for candidate columns and output tables: with it the H2D / D2H copies
 * of adh_score_candidates run asynchronously at full PCIe rate and overlap the kernels.  Pageable
 * buffers are accepted everywhere, they are just slower.
 */
int adh_host_alloc(void **ptr, uint64_t bytes);
/* blocking copy of `bytes` bytes from a device pointer of one of the views above to host memory */
int adh_copy_to_host(adh_handle_t *handle, void *dst, const void *src_device, uint64_t bytes);
int adh_host_free(void *ptr);

/*
 * Device buffers of 256 MB and more that the library no longer needs (sort temporaries of a staging call, tables and
 * scratch it outgrew, what a destroyed handle held) are parked and reused instead of freed - freeing gigabytes of
 * device memory leaves the runtime's DMA copies at half the link rate for the rest of the process.  This
 * call returns everything parked (at most ADH_DEV_CACHE_GB, default 48) to the runtime.
 */
int adh_trim_device_cache(void);

/*
 * How many host threads a scoring call over a table of n_rows candidates starts for its host-side work (rebuilding the
 * id / library columns behind the copy-out, unpacking the compact blocks), and the CPU budget that number is cut from:
 * the smallest of the hardware threads, the scheduler affinity mask and the cgroup CPU quota (cpu.max), divided by
 * LOCAL_WORLD_SIZE - the ranks of a node share one quota -, at most 16, a thread per 16 384 rows at least;
 * ADH_HOST_THREADS overrides it.  Replaces the `thread_count` argument of CandidateScoring.__call__
 * (alphadia/search/scoring/scoring.py:583-661, `alphatims.utils.set_threads`), which the reference leaves to the caller.
 * Needs no GPU.  Either pointer may be NULL.
 */
int adh_host_threads(int64_t n_rows, int32_t *threads, int32_t *cpu_budget);

/*
 * dst[i] = src[idx[i]] for n entries of arrays of CPython object pointers (NumPy dtype=object) - the string columns
 * the features frame takes over from the precursor table (scoring.py:430-445 merges them in) - on `threads` host
 * threads, reference counts raised atomically.  The caller must hold the GIL for the whole call (ctypes.PyDLL) and
 * pass a fresh dst - every entry NULL or `fill` (np.empty fills an object array with references to None; they are
 * given back); idx outside [0, n_src) fails.  Host helper of the Python operator: no GPU involved.
 */
int adh_host_take_objects(void **dst, void *const *src, const int64_t *idx, int64_t n, int64_t n_src, void *fill,
                          int32_t threads);

/*
 * Test entry: the dense tile the gather kernels build for ONE query, i.e. what
 * AlphaRawJIT.get_dense (search/jitclasses/alpharaw_jit.py:208-337) /
 * TimsTOFTransposeJIT.get_dense (search/jitclasses/bruker_jit.py:273-504) return with
 * absolute_masses=True: dense[2][n_query][O][S][F] (intensity plane, then m/z plane; S = 1 for an
 * AlphaRaw run, whose two scan slots are copies, alpharaw_jit.py:326-333) and the cycle rows
 * (`precursor_idx_list`) in obs[0..*n_obs).  mz_query must be ascending.  Runs the production
 * gather kernel on a one-candidate plan built from the query.
 */
int adh_debug_get_dense(adh_handle_t *handle, int64_t frame_start, int64_t frame_stop, int64_t scan_start,
                        int64_t scan_stop, const float *mz_query, int32_t n_query, float mass_tolerance,
                        float quad_lo, float quad_hi, float *dense, int64_t dense_capacity, int32_t *obs,
                        int32_t *n_obs, int32_t *n_scans, int32_t *n_cycles);

/* The handle's own (non-blocking) stream as a hipStream_t. */
int adh_get_stream(adh_handle_t *handle, void **hip_stream);

/* Block until all work enqueued on the handle's stream has finished. */
int adh_synchronize(adh_handle_t *handle);

/*
 * Average durations (milliseconds per adh_score_uploaded call) of the two scoring
 * kernels since the last reset, measured with HIP events on the launch stream:
 * `gather_ms` = fragment selection + XIC gather, `feature_ms` = the feature stack.
 */
int adh_kernel_time_ms(adh_handle_t *handle, double *gather_ms, double *feature_ms,
                       int64_t *launches, int reset);

/* ------------------------------------------------------------------------
 * Multi-GPU: one process per GPU, candidates sharded by contiguous score-group ranges
 * (score groups are independent, search/scoring/containers/score_group.py:66-75), ONE RCCL
 * all-gather of the computed tables over xGMI.  librccl is loaded on first use.
 * ---------------------------------------------------------------------- */

/* 128-byte RCCL unique id (rank 0 creates it; the caller hands it to every rank). */
int adh_comm_unique_id(void *id128);
/*
 * Attach a communicator to the handle.  From now on every adh_score_candidates call lays its
 * device tables out for `max_rows_per_rank` rows (the largest shard, so that all ranks agree on the
 * layout) and, once its kernels are enqueued, all-gathers the computed tables of all ranks into
 * HBM on a stream of its own.  The call returns when the host tables are complete; the gather may
 * still run and overlaps the next call (two table slots).  adh_comm_wait blocks until it is done.
 */
int adh_comm_init(adh_handle_t *handle, int rank, int world, const void *id128, int64_t max_rows_per_rank);
int adh_comm_destroy(adh_handle_t *handle);
int adh_comm_wait(adh_handle_t *handle);
/* Device view of rank `rank`'s computed tables after the last call's gather (waits for it). */
int adh_comm_gathered(adh_handle_t *handle, int rank, adh_output_t *device_view, int64_t *rows);
/* max over ranks of *value (in place); also the barrier of the benchmark.  No-op without a communicator. */
int adh_comm_all_reduce_max(adh_handle_t *handle, double *value);
int adh_comm_barrier(adh_handle_t *handle);
/* Every rank passes `bytes` bytes of host memory (the same count on all ranks) and receives world x bytes in
 * rank order (recv holds world x bytes; without a communicator recv = send).  The exchange of the two other
 * stages of the path that shard without touching each other's rows - candidate selection by precursor range
 * (selection.py:620-660) and fragment competition by DIA window (fragcomp/fragcomp.py:204-229,278): one gather
 * of the per-rank results (alphadia_amd/selection.py, alphadia_amd/fragcomp.py).  The reference runs both on
 * the threads of one process and has no counterpart. */
int adh_comm_all_gather_host(adh_handle_t *handle, const void *send, uint64_t bytes, void *recv);
/* What RCCL itself reports for the attached communicator (ncclCommUserRank / ncclCommCount): rank 0 of 1
 * without one.  bench.py prints it, so that a run on N GPUs shows that N ranks met. */
int adh_comm_info(adh_handle_t *handle, int *rank, int *world);
/* hipDeviceSynchronize on the handle's GPU. */
int adh_device_synchronize(adh_handle_t *handle);

/*
 * Fragment competition inside one run (replaces `_compete_for_fragments`,
 * fragcomp/fragcomp.py:51-143).  PSMs are sorted by (window, proba, precursor)
 * and windows are given as [start, stop) row ranges exactly as
 * FragmentCompetition.__call__ prepares them (fragcomp.py:268-289); the row ranges of
 * different windows must not overlap.  valid[] is input/output (all ones on entry; a PSM
 * that enters with 0 neither removes nor is looked at, as in the reference's loops).
 */
int adh_fragcomp(adh_handle_t *handle, int64_t n_windows, const int64_t *window_start,
                 const int64_t *window_stop, int64_t n_psm, const float *rt,
                 const int64_t *frag_start_idx, const int64_t *frag_stop_idx,
                 int64_t n_frag, const float *fragment_mz, double rt_tol_seconds,
                 double mass_tol_ppm, uint8_t *valid);
/*
 * FragmentCompetition.__call__ (fragcomp/fragcomp.py:204-299) from the columns of its two frames: the preparation the
 * reference does with pandas - candidate keys (fragcomp/utils.py:48-58), the fragment range of every PSM
 * (add_frag_start_stop_idx, utils.py:11-45; PSMs without fragment rows leave), the DIA window of every PSM
 * (_add_window_idx, fragcomp.py:170-202; `window_lower` / `window_upper` = per cycle row the lowest / highest isolation
 * limit over its scans), the processing order (window, proba, precursor_idx, input position) - runs on the device, then
 * the competition.  Out: `rows[0 .. *n_rows)` = input position of every processed PSM in processing order, `valid` its
 * flag.  `*grouped` = 0 (and nothing else done) when a candidate's fragment rows are not contiguous in the fragment
 * table or a table has 2^31 rows or more: the caller then prepares the plan itself and calls adh_fragcomp.
 */
int adh_fragcomp_frames(adh_handle_t *handle, int64_t n_psm, const uint32_t *psm_precursor_idx, const uint8_t *psm_rank,
                        const float *psm_mz_observed, const float *psm_rt_observed, const float *psm_proba, int64_t n_frag,
                        const uint32_t *frag_precursor_idx, const uint8_t *frag_rank, const float *frag_mz_observed,
                        int32_t n_cycle_rows, const double *window_lower, const double *window_upper, double rt_tol_seconds,
                        double mass_tol_ppm, int64_t *rows, uint8_t *valid, int64_t *n_rows, int32_t *grouped);

/* What the last competition on this handle (adh_fragcomp or inside adh_fdr_resident) did: HIP-event
 * duration of its device work, (PSM, RT neighbour) pairs whose fragment lists were compared, PSMs whose
 * fate hung on an earlier PSM, rounds that settled those, and whether the one-workgroup-per-window
 * kernel had to run (the neighbour bitmap did not fit).  Any pointer may be NULL.  No reference
 * counterpart (the reference times `_compete_for_fragments` with its logger, fragcomp.py:283-289). */
int adh_fragcomp_stats(adh_handle_t *handle, double *kernel_ms, int64_t *pairs, int64_t *waiting,
                       int32_t *rounds, int32_t *serial);

/* ------------------------------------------------------------------------
 * Candidate selection - the step before scoring.
 * Non-ion-mobility (AlphaRaw) runs.
 * ---------------------------------------------------------------------- */

/*
 * Precursor table of the selection step: the columns of PrecursorFlatContainer
 * (search/selection/config_df.py; filled by
 * CandidateSelection._assemble_precursor_container, selection.py:712-737),
 * sorted by precursor_idx.
 */
typedef struct adh_precursors {
    int64_t n;
    const uint32_t *precursor_idx;
    const uint32_t *frag_start_idx;
    const uint32_t *frag_stop_idx;
    const uint8_t *charge;
    const float *rt;                 /* the configured rt column, seconds */
    const float *mobility;           /* unused for AlphaRaw runs */
    const float *mz;                 /* the configured precursor m/z column */
    const float *isotope_intensity;  /* [n][n_isotope_cols], row-major (columns i_0, i_1, ...) */
    int32_t n_isotope_cols;
} adh_precursors_t;

/* CandidateSelectionConfigJIT (search/selection/config_df.py:15-110), single score feature. */
typedef struct adh_selection_config {
    double rt_tolerance;
    double precursor_mz_tolerance;
    double fragment_mz_tolerance;
    int64_t candidate_count;
    int64_t top_k_precursors;        /* isotopes used */
    int64_t kernel_size;
    double f_mobility, f_rt, center_fraction;
    int64_t min_size_mobility, min_size_rt, max_size_mobility, max_size_rt;
    double join_close_candidates_scan_threshold;
    double join_close_candidates_cycle_threshold;
    double feature_mean, feature_std, feature_weight;  /* used when use_weighted_score */
    uint8_t exclude_shared_ions;
    uint8_t use_weighted_score;
    uint8_t join_close_candidates;
    uint8_t pad0;
    uint8_t pad1[4];
    double mobility_tolerance;       /* ion-mobility runs only */
} adh_selection_config_t;

/*
 * CandidateContainer (search/selection/config_df.py:227-254): n = precursors x
 * candidate_count rows, row (i * candidate_count + rank); rows that hold no
 * candidate keep score = 0 (candidate_container_to_df drops them, :270-298).
 */
typedef struct adh_candidate_table {
    int64_t n;
    uint32_t *precursor_idx;
    uint8_t *rank;
    float *score;
    uint32_t *scan_center, *scan_start, *scan_stop;
    uint32_t *frame_center, *frame_start, *frame_stop;
} adh_candidate_table_t;

/*
 * Select up to candidate_count (rt) boxes per precursor on the staged AlphaRaw
 * run and fragment library.  Replaces the pjit loop `_select_candidates_pjit`
 * (search/selection/selection.py:78-203): dense XICs of isotopes and fragments
 * over rt +- rt_tolerance, circular convolution with `kernel`
 * (GaussianKernel.get_dense_matrix, selection/kernel.py:141-218; the reference
 * does it by FFT, selection/fft.py:119-212), log-sum score, peak picking and
 * symmetric limits (selection/utils.py:46-312).  Host buffers in and out; `out`
 * is zero-filled by the call.
 */
int adh_select_candidates(adh_handle_t *handle, const adh_precursors_t *precursors,
                          const adh_selection_config_t *config, const float *kernel,
                          int32_t kernel_rows, int32_t kernel_cols, adh_candidate_table_t *out);

/*
 * Transpose timsTOF detector events from the frame-major alphatims layout (rows = pushes:
 * push_indptr[n_push + 1], tof_indices[n], values[n]) into the TOF-major layout the scoring and
 * selection operators read (tof_indptr[n_tof + 1], push_indices[n], values[n]; pushes ascending
 * inside a TOF bin).  Drop-in for `_transpose` (alphadia/raw_data/bruker.py:201-280), which does
 * this on 20 CPU threads when a run is loaded.  Host buffers in and out; n < 2^31.
 */
int adh_transpose_timstof(adh_handle_t *handle, const uint32_t *tof_indices, const int64_t *push_indptr,
                          int64_t n_push, int64_t n_tof, const uint16_t *values, int64_t n_events,
                          uint32_t *push_indices_out, int64_t *tof_indptr_out, uint16_t *values_out);

/* Duration (ms, HIP events) of the selection kernel of the last adh_select_candidates call. */
int adh_select_time_ms(adh_handle_t *handle, double *kernel_ms);

/* ------------------------------------------------------------------------------------------
 * FDR stage: alphadia/fdr/fdr.py + alphadia/fdr/classifiers.py.
 * ------------------------------------------------------------------------------------------ */

/*
 * q-values of n rows.  Replaces `get_q_values` + `_fdr_to_q_values` (fdr/fdr.py:215-297): rows are
 * ordered by (score, decoy, tiebreak) ascending (stable; NaN scores last), fdr = cumulative decoys /
 * cumulative targets (float64, 1/0 = inf as in numpy), q = running minimum of fdr from the back.
 * `order_out[i]` = input row at sorted position i, `qval_out[i]` = its q-value (sorted order, as
 * the reference returns the frame).  `tiebreak` (the reference's precursor_idx) may be NULL.
 */
int adh_fdr_q_values(adh_handle_t *handle, int64_t n, const double *score, const uint8_t *decoy,
                     const int64_t *tiebreak, int64_t *order_out, double *qval_out);

/*
 * Best row per group.  Replaces `keep_best` (fdr/fdr.py:181-213): of the rows sharing
 * (group_a[, group_b]) the one with the LOWEST score stays (score = decoy probability), the
 * earliest row on ties; keep[i] = 1 for the rows that stay, rows keep their input order.
 */
int adh_fdr_keep_best(adh_handle_t *handle, int64_t n, const double *score, const int64_t *group_a,
                      const int64_t *group_b, uint8_t *keep);

/*
 * The target/decoy classifier: BatchNorm1d -> (Linear -> ReLU -> Dropout) x hidden -> Linear ->
 * Softmax trained with BCELoss and Adam (`FeedForwardNN`, fdr/classifiers.py:497-532;
 * `BinaryClassifierLegacyNewBatching.fit/predict_proba`, :316-495).  The feature matrix is staged
 * once and stays in HBM; a training step is three kernels (batch statistics, fused
 * forward/backward per 16-row tile, gradient reduction + Adam) with no host round trip.
 */
#define ADH_MLP_MAX_LINEAR 8
typedef struct adh_mlp adh_mlp_t;

typedef struct adh_mlp_arch {
    int32_t n_linear;                       /* Linear layers: hidden layers + the output layer */
    int32_t dims[ADH_MLP_MAX_LINEAR + 1];   /* input_dim, hidden sizes ..., output_dim */
    float bn_eps, bn_momentum;              /* torch.nn.BatchNorm1d defaults: 1e-5, 0.1 */
} adh_mlp_arch_t;

typedef struct adh_mlp_fit {
    const int64_t *train_rows;   /* rows of the staged matrix in training order (x_train) */
    int64_t n_train;
    const int64_t *batch_start;  /* per optimiser step: first position of its batch in train_rows */
    int64_t n_steps;
    int32_t batch_size;
    float learning_rate, weight_decay, dropout;
    float beta1, beta2, eps;     /* torch.optim.Adam defaults: 0.9, 0.999, 1e-8 */
    uint64_t seed;               /* dropout masks */
    int64_t first_step;          /* optimiser steps taken before this call; 0 resets the Adam moments */
} adh_mlp_fit_t;

int adh_mlp_create(adh_handle_t *handle, const adh_mlp_arch_t *arch, adh_mlp_t **mlp);
int adh_mlp_destroy(adh_mlp_t *mlp);
/* floats in the trainable parameter vector: bn weight[d], bn bias[d], then per Linear weight[out][in], bias[out] */
int adh_mlp_param_count(const adh_mlp_arch_t *arch, int64_t *n_params);
int adh_mlp_set_state(adh_mlp_t *mlp, const float *params, const float *running_mean, const float *running_var,
                      int64_t num_batches_tracked);
int adh_mlp_get_state(adh_mlp_t *mlp, float *params, float *running_mean, float *running_var,
                      int64_t *num_batches_tracked);
/* copy x[n][d] (row-major float32) and, if not NULL, the class-1 target of every row to HBM */
int adh_mlp_stage_rows(adh_mlp_t *mlp, const float *x, int64_t n, int32_t d, const float *y);
/* n_steps training steps (network.train() semantics); train_loss[n_steps] may be NULL */
int adh_mlp_fit(adh_mlp_t *mlp, const adh_mlp_fit_t *fit, float *train_loss);
/* network.eval() forward of `rows` (NULL = all staged rows): proba[n][output_dim] */
int adh_mlp_predict(adh_mlp_t *mlp, const int64_t *rows, int64_t n, float *proba);
/* HIP-event time (ms) of the kernels of the last adh_mlp_fit / adh_mlp_predict call */
int adh_mlp_time_ms(adh_mlp_t *mlp, double *fit_ms, double *predict_ms);

/* ------------------------------------------------------------------------------------------
 * The FDR stage fed from the scoring tables that adh_score_candidates left in HBM: the 46-float
 * feature rows and the fragment tables never cross PCIe between scoring and FDR
 * (the reference hands DataFrames over on the host: workflow/peptidecentric/peptidecentric.py:219-243
 * -> fdr/fdr.py:24-178).  Only row-sized metadata travels.
 * ------------------------------------------------------------------------------------------ */

/*
 * Classifier rows straight from the device tables (fdr.py:86-105): of the `n_rows` candidate rows of
 * the last adh_score_candidates call, those that are valid and have no NaN in a classifier column are
 * staged, targets first, then decoys, each in row order (np.concatenate([targets, decoys])).
 * Column j of the network input is, by src_cols[j]:  0..45 = that feature;  46 + e = extra_cols[e]
 * (host float[n_rows], e.g. mz_library);  -(1 + e) = rt_observed - extra_cols[e] (delta_rt,
 * scoring.py:458).  `decoy` is a host uint8[n_rows].  Returns the staged counts.
 */
int adh_mlp_stage_rows_device(adh_mlp_t *mlp, const int32_t *src_cols, int32_t d, const float *const *extra_cols,
                              int32_t n_extra, const uint8_t *decoy, int64_t n_rows, int64_t *n_targets,
                              int64_t *n_decoys);
/* candidate row of every staged row (8 bytes per row; the host needs it to label its metrics) */
int adh_mlp_staged_rows(adh_mlp_t *mlp, int64_t *rows_out, int64_t capacity);
/* network.eval() forward of all staged rows; the probabilities stay in HBM */
int adh_mlp_predict_resident(adh_mlp_t *mlp);
/*
 * fdr.py:134-178 on the device: q-values of the staged rows (sorted by proba, decoy, tiebreak) ->
 * if `cycle` is given (float64 [cycle_len][cycle_scans][2], cycle_scans <= 2): the rows below
 * `fdr_heuristic` compete for fragments (fragcomp/fragcomp.py:231-299; fragment m/z from the
 * fragment_mz_observed table, rt / m/z from features 2 / 10), the others are dropped -> best row per
 * (group_a[, group_b]) -> q-values.  group_a / group_b / tiebreak are host int64[n_rows of the
 * device tables] (e.g. elution_group_idx, channel, precursor_idx).  Out (host, capacity = staged
 * rows): candidate row, class-1 probability and q-value of the surviving PSMs in the final order.
 */
int adh_fdr_resident(adh_handle_t *handle, adh_mlp_t *mlp, const int64_t *group_a, const int64_t *group_b,
                     const int64_t *tiebreak, const double *cycle, int32_t cycle_len, int32_t cycle_scans,
                     double rt_tol_seconds, double mass_tol_ppm, double fdr_heuristic, int64_t *n_out,
                     int64_t *row_out, float *proba_out, double *qval_out);
/* bytes this library has copied device -> host on the handle's GPU since the last reset */
int adh_transfer_counters(adh_handle_t *handle, uint64_t *d2h_bytes, int reset);

#ifdef __cplusplus
}
#endif
#endif /* ALPHADIA_HIP_H */
