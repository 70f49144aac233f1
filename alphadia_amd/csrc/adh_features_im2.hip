// adh_features_im2.hip - profile phase of the ion-mobility feature stack, FOUR candidates per wavefront.
//
// adh_feature_im_kernel (adh_features_im.hip) spends its first 6.7 of 15.7 ms per 600 000 candidates on the tiles
// (precursor pass, template, fragment tile pass) and the other 9 ms on what follows - presence, envelopes,
// quantification, the ~45 sums of the feature assembly, scan correlation, median profile, frame statistics:
// candidate.py:319-481, fragment_features.py:198-480, profile_features.py:18-206 - all of it work on a few KB of
// PROFILES with one lane per fragment: 12 busy lanes of 64, and the kernel is bound by instruction issue
// (DESIGN.md section 4.3).  Round 4 splits the kernel there: the tile part leaves an ImProfRec (adh_device.h)
// per candidate, and this kernel scores four candidates per wavefront, 16 lanes each - the shape of
// adh_fused_kernel (adh_fused.hip), whose device code for the shared steps it reuses:
//   * lane k < 12 owns fragment k: its frame profile (FM registers, centred as in adh_fused.hip) and its
//     scan profile (SM registers); nothing per-fragment lives in LDS but the centred scan rows of the K x K
//     scan correlation, which the MFMA tile reads (one v_mfma_f32_16x16x4_f32 chain per candidate, issued by
//     the whole wavefront for each of its four candidates in turn)
//   * sums over fragments: lane k provides its term of every sum, lane j adds up sum j in fragment order
//   * the arithmetic is that of adh_feature_im_kernel expression by expression (float64 rt / mobility typing of
//     the ion-mobility arrays, bruker_jit.py:35,45): the two paths are held to identical bits by the GPU suite
// One or two observations (NO; with two, the rows of an observation are taken from the record where they are
// needed - the profile sum over the observations is what stays in registers), <= 12 fragments, <= 3 isotopes,
// experimental_xic: plan classes 0, 1 and ADH_CLASS_IM_SMALL with the fixed layouts.  Everything else keeps the
// one-kernel path.
#include "adh_device.h"
#include "adh_feature_common.h"

namespace featim2 {

using fused::GS;
using fused::Recip;

constexpr int KMAX = ADH_IM_PROF_K;

template <int FM, int SM, int NO>
struct __attribute__((aligned(16))) GroupLds {
    union {
        struct {               // per-fragment terms of the feature sums: [fragment][sum]
            double t64[16][6];
            float t32[16][6];
        } at;
        float cen[16][SM + 1];  // centred scan rows of the kept fragments (MFMA operands) ...
        float gram[16][17];     // ... whose Gram matrix takes their place
        float nrmT[16][17];     // transpose buffer of the per-cycle median
    } u;
    double frt[FM];             // frame times (float64), centred
    double merr[16];
    double red64[14];
    double hp[4], omzp[4];
    float tfp_raw[NO][FM], tfp[NO][FM], med[FM];
    float tsp_raw[NO][SM], tsp[NO][SM];
    float g_int[16], g_fin[16], corr[16];
    float mn[16], sd[NO][16], ml[16], ftc2[NO][16];
    float iso_int[4], iso_mz[4], spi[4];
    float feat[ADH_NUM_FEATURES + 2];
    float red32[4];
    int fpeak[16][NO];
    int ord[16];
    int medlo[NO], medhi[NO];
};

#define F2_FOR_R _Pragma("unroll") for (int r = 0; r < FM; ++r)
#define F2_FOR_S _Pragma("unroll") for (int i = 0; i < SM; ++i)
#define F2_OK(r) ((unsigned)((r) + shift) < (unsigned)F)

template <int N>
__device__ __forceinline__ void load_row4(float (&x)[N], const float *p) {
    static_assert(N % 4 == 0, "rows are whole float4s");
#pragma unroll
    for (int i = 0; i < N / 4; ++i) {
        const float4 v = reinterpret_cast<const float4 *>(p)[i];
        x[4 * i] = v.x, x[4 * i + 1] = v.y, x[4 * i + 2] = v.z, x[4 * i + 3] = v.w;
    }
}

// mean and population standard deviation of the first n entries of a zero-padded register row, as
// featim::row_moments takes them (scoring/utils.py:545-559)
template <int N>
__device__ __forceinline__ void moments(const float (&x)[N], int n, float &mean, float &sd) {
    float sx = 0.0f;
#pragma unroll
    for (int i = 0; i < N; ++i) sx += x[i];
    mean = sx / (float)n;
    float q = 0.0f;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        float d = x[i] - mean;
        d = (i < n) ? d : 0.0f;
        q += d * d;
    }
    sd = sqrtf(q / (float)n);
}

}  // namespace featim2

// What the profile phase starts from: the hand-over record of the tile kernels (ImProfRec in HBM), or - FUSED, one
// observation - the LDS arrays and lane results of featim4::tile4_phase of the SAME wavefront (adh_features_im4.hip).
// FUSED: the group LDS of this phase (`lds`) lies over the tile phase's arrays, so everything the phase needs of them
// is taken into registers before the first write.
template <int FM, int SM, int NO, bool FUSED>
__device__ __forceinline__ void adh_im_profiles_body(
    featim2::GroupLds<FM, SM, NO> *const lds, const int ci, const bool in_range, const DevTims &run,
    const CandRecIM *__restrict__ plan, const adh_scoring_config_t &cfg, int32_t n_iso_cols,
    const unsigned char *__restrict__ scratch, const unsigned char *__restrict__ prof, const DevOut &out,
    const featim4::GroupTile<FM, SM, 1> *const tile, const featim4::TileOut *const tout) {
    using namespace featim2;
    constexpr int RC = FM / 2;
    typedef ImProfRec<FM, SM, NO> Rec;
    static_assert(!FUSED || NO == 1, "the fused path takes one observation");
    const int lane = threadIdx.x;
    const int g = lane / GS, sub = lane % GS;
    const unsigned gsh = (unsigned)(g * GS);
    GroupLds<FM, SM, NO> &Q = lds[g];
    bool alive = in_range;
    const CandRecIM &cand = plan[alive ? ci : 0];
    alive = alive && !(cand.flags & ADH_FLAG_SKIP);
    const unsigned char *block = scratch + cand.scratch_off;
    const int K0 = alive ? (FUSED ? tout->K0 : (int)reinterpret_cast<const uint32_t *>(block)[0]) : 0;
    alive = alive && K0 > 0;
    const Rec &rec = reinterpret_cast<const Rec *>(prof)[(alive && !FUSED) ? ci : 0];
    const uint32_t row = cand.row;
    const int L = run.cycle_len, z = run.zeroth;
    const int c0 = (cand.frame_start - z) / L;
    const int F = alive ? (cand.frame_stop - z) / L - c0 : 0;
    const int S = alive ? cand.scan_stop - cand.scan_start : 0;
    const int c = F / 2;
    const int shift = c - RC;  // f = r + shift
    const int I = min(min(n_iso_cols, (int)cfg.top_k_isotopes), 4);
    const int top_k = out.top_k;
    auto gmask = [&](bool p) -> unsigned { return (unsigned)((__ballot(p) >> gsh) & 0xFFFFull); };

    // ---- the small arrays into the group's LDS; a fragment lane's rows are taken from the record where needed
    const bool frag_lane0 = alive && sub < K0;
    const int slot = frag_lane0 ? sub : 0;
    // (FUSED) everything of the tile phase's LDS arrays, into registers, before this phase writes its own
    float P0[FUSED ? FM : 1], Sp0[FUSED ? SM : 1], tfpv[(FM + 15) / 16], tspv[(SM + 15) / 16], isov[2] = {0.0f, 0.0f};
    if constexpr (FUSED) {
        F2_FOR_R P0[r] = (frag_lane0 && F2_OK(r)) ? tile->ffp[0][min(max(r + shift, 0), FM - 1)][slot] : 0.0f;
        F2_FOR_S Sp0[i] = (frag_lane0 && i < S) ? tile->v.fsp[0][i][slot] : 0.0f;
#pragma unroll
        for (int pass = 0; pass < (FM + 15) / 16; ++pass) {
            const int f = min(sub + 16 * pass, FM - 1) + shift;
            tfpv[pass] = (alive && f >= 0 && f < F) ? tile->tfp_raw[0][min(max(f, 0), FM - 1)] : 0.0f;
        }
#pragma unroll
        for (int pass = 0; pass < (SM + 15) / 16; ++pass) {
            const int i = min(sub + 16 * pass, SM - 1);
            tspv[pass] = (alive && i < S) ? tile->tsp_raw[0][i] : 0.0f;
        }
        if (sub < 4) {
            isov[0] = (alive && sub < I) ? tile->iso_int[sub] : 0.0f;
            isov[1] = (alive && sub < I) ? tile->iso_mz[sub] : 0.0f;
        }
        adh_wave_sync();
        // (the frame profile is not needed before the quantification: it waits in the rows of the scan-correlation
        // operands, idle until then - 32 registers less across the presence / envelope steps, where the kernel spilled)
        static_assert(SM + 1 >= FM, "a frame profile fits a row of the centred scan rows");
        F2_FOR_R Q.u.cen[sub][r] = P0[r];
    }
    // frame profile of observation o (centred), raw scan profile of observation o
    auto load_P = [&](int o, float (&P)[FM]) {
        if constexpr (FUSED) {
            F2_FOR_R P[r] = Q.u.cen[sub][r];
        } else {
            F2_FOR_R P[r] = 0.0f;
            if (frag_lane0) load_row4<FM>(P, rec.ffp[slot][o]);
        }
    };
    auto load_Sp = [&](int o, float (&Sp)[SM]) {
        if constexpr (FUSED) {
            F2_FOR_S Sp[i] = Sp0[i];
        } else {
            F2_FOR_S Sp[i] = 0.0f;
            if (frag_lane0) load_row4<SM>(Sp, rec.fsp[slot][o]);
        }
    };
    // OR-envelope of a scan profile (scoring/utils.py:56-66): reads the raw neighbours, writes a copy
    auto scan_envelope = [&](const float (&Sp)[SM], float (&Se)[SM]) {
        F2_FOR_S {
            float v = Sp[i];
            if (i >= 1 && i + 1 < SM) {
                const bool inner = i < S - 1;
                const float xl = Sp[i - 1], xr = Sp[i + 1];
                if (inner && (v < xl || v < xr)) {
                    const float sm = xl + xr;
                    v = (float)((double)sm / 2.0);
                }
            }
            Se[i] = v;
        }
    };
    double ohe_l[NO], omz_l[NO];
    fused::RawRec lrec;
    lrec.a = make_uint4(0u, 0u, 0u, 0u);
    lrec.b = 0u;
    uint32_t lib_slot = 0;
#pragma unroll
    for (int o = 0; o < NO; ++o) {
        if constexpr (FUSED) {
            ohe_l[o] = frag_lane0 ? tout->ohe[o] : 0.0;
            omz_l[o] = frag_lane0 ? tout->omz[o] : 0.0;
        } else {
            ohe_l[o] = frag_lane0 ? rec.ohe[slot][o] : 0.0;
            omz_l[o] = frag_lane0 ? rec.omz[slot][o] : 0.0;
        }
    }
    if (frag_lane0) {
        const LibRec *sel = reinterpret_cast<const LibRec *>(block + 32) + sub;
        lrec = fused::load_rec(sel);
        lib_slot = 1u + (uint32_t)sel->pad0 + 256u * (uint32_t)sel->pad1;
    }
    const float lrec_mz = fused::rec_mz(lrec), lrec_int = fused::rec_intensity(lrec);
#pragma unroll
    for (int pass = 0; pass < (FM + 15) / 16; ++pass) {
        const int rr = min(sub + 16 * pass, FM - 1);
        const int f = rr + shift;
        const bool ok = alive && f >= 0 && f < F;
#pragma unroll
        for (int o = 0; o < NO; ++o) Q.tfp_raw[o][rr] = FUSED ? tfpv[pass] : (alive ? rec.tfp_raw[o][rr] : 0.0f);
        Q.frt[rr] = ok ? run.rt[cand.frame_start + f * L] : 0.0;
    }
#pragma unroll
    for (int pass = 0; pass < (SM + 15) / 16; ++pass) {
        const int i = min(sub + 16 * pass, SM - 1);
#pragma unroll
        for (int o = 0; o < NO; ++o) Q.tsp_raw[o][i] = FUSED ? tspv[pass] : (alive ? rec.tsp_raw[o][i] : 0.0f);
    }
    if (sub < 4) {
        if constexpr (FUSED) {
            const bool on = alive && sub < I;
            Q.hp[sub] = on ? tout->hp : 0.0;
            Q.omzp[sub] = on ? tout->omzp : 0.0;
            Q.spi[sub] = on ? tout->spi : 0.0f;
            Q.iso_int[sub] = isov[0];
            Q.iso_mz[sub] = isov[1];
        } else {
            Q.hp[sub] = alive ? rec.hp[sub] : 0.0;
            Q.omzp[sub] = alive ? rec.omzp[sub] : 0.0;
            Q.spi[sub] = alive ? rec.spi[sub] : 0.0f;
            Q.iso_int[sub] = alive ? rec.iso_int[sub] : 0.0f;
            Q.iso_mz[sub] = alive ? rec.iso_mz[sub] : 0.0f;
        }
    }
    float tsum[NO];
#pragma unroll
    for (int o = 0; o < NO; ++o) tsum[o] = FUSED ? (alive ? tout->tsum[o] : 0.0f) : (alive ? rec.tsum[o] : 0.0f);
    // location_features.py:8-33 with float64 mobility / rt arrays
    float loc = 0.0f;
    double rt_width = 0.0, mob_width = 0.0;
    if (alive) {
        rt_width = run.rt[cand.frame_stop - 1] - run.rt[cand.frame_start];
        mob_width = run.mobility[cand.scan_start] - run.mobility[cand.scan_stop - 1];
        loc = sub == 0 ? (float)mob_width : (sub == 1 ? (float)rt_width : (sub == 2 ? (float)run.rt[cand.frame_center] : (float)run.mobility[cand.scan_center]));
    }
#pragma unroll
    for (int j = 0; j < 3; ++j)
        if (sub + 16 * j < ADH_NUM_FEATURES + 2) Q.feat[sub + 16 * j] = 0.0f;
    adh_wave_sync();

    // ---- observation importance (quadrupole.py:327-335), fragment presence (candidate.py:319-329)
    float oi[NO];
    {
        float tot = 0.0f;
#pragma unroll
        for (int o = 0; o < NO; ++o) tot += tsum[o];
#pragma unroll
        for (int o = 0; o < NO; ++o) oi[o] = (tot == 0.0f) ? 1.0f / (float)NO : tsum[o] / tot;
    }
    int best_obs = 0;  // np.argmax: the first maximum (fragment_features.py:245)
#pragma unroll
    for (int o = 1; o < NO; ++o)
        if (oi[o] > oi[best_obs]) best_obs = o;
    // the lane's enveloped scan profile: kept in registers with one observation, taken again per use with two
    float Se[SM];
    float rowsum_l[NO], so = 0.0f;
#pragma unroll
    for (int o = 0; o < NO; ++o) {
        float Sp[SM];
        load_Sp(o, Sp);
        float ss = 0.0f;
        F2_FOR_S ss += Sp[i];  // sum of the per-scan sums, in scan order
        rowsum_l[o] = frag_lane0 ? ss : 0.0f;
        so += rowsum_l[o];
        if (NO == 1) scan_envelope(Sp, Se);
    }
    bool present = frag_lane0 && so > 0.0f;
    const unsigned gm = gmask(present);
    int K = __popc(gm);
    const int kk = __popc(gm & ((1u << sub) - 1u));
    const int n_present = K;
    if (K < 2) {  // candidate.py:323
        alive = false;
        present = false;
        K = 0;
    }
    if (present) Q.g_fin[kk] = lrec_int;  // raw intensity, normalised below
    adh_wave_sync();
    float g_int_l = 0.0f, g_fin_l = 0.0f;
    {
        float sum1 = 0.0f;
#pragma unroll
        for (int j = 0; j < KMAX; ++j) sum1 += (j < K) ? Q.g_fin[j] : 0.0f;
        if (present) {
            g_int_l = lrec_int / sum1;
            Q.g_int[kk] = g_int_l;
        }
    }
    adh_wave_sync();
    {
        float sum2 = 0.0f;
#pragma unroll
        for (int j = 0; j < KMAX; ++j) sum2 += (j < K) ? Q.g_int[j] : 0.0f;
        if (present) g_fin_l = g_int_l / sum2;
    }
    adh_wave_sync();
    if (present) Q.g_fin[kk] = g_fin_l;

    // ---- OR-envelopes of the template profiles (scoring/utils.py:46-66)
#pragma unroll
    for (int o = 0; o < NO; ++o) {
#pragma unroll
        for (int pass = 0; pass < (FM + 15) / 16; ++pass) {
            const int rr = min(sub + 16 * pass, FM - 1);
            const int f = rr + shift;
            const float x = Q.tfp_raw[o][rr];
            float v = x;
            if (f >= 1 && f < F - 1) {
                const float xl = Q.tfp_raw[o][rr - 1], xr = Q.tfp_raw[o][rr + 1];
                if (x < xl || x < xr) {
                    const float sm = xl + xr;
                    v = (float)((double)sm / 2.0);
                }
            }
            Q.tfp[o][rr] = (f >= 0 && f < F) ? v : 0.0f;
        }
#pragma unroll
        for (int pass = 0; pass < (SM + 15) / 16; ++pass) {
            const int i = min(sub + 16 * pass, SM - 1);
            const float x = Q.tsp_raw[o][i];
            float v = x;
            if (i >= 1 && i < S - 1) {
                const float xl = Q.tsp_raw[o][i - 1], xr = Q.tsp_raw[o][i + 1];
                if (x < xl || x < xr) {
                    const float sm = xl + xr;
                    v = (float)((double)sm / 2.0);
                }
            }
            Q.tsp[o][i] = (i < S) ? v : 0.0f;
        }
    }

    // ---- envelope, quantification (fragment_features.py:240-273; rt_values are float64 here).  P = the frame
    // profile summed over the observations, as the profile features see it: with quant_all the sum is quantified
    // (a copy); without, the most important observation's profile is - a view, its envelope edit stays
    double area = 0.0;
    float obs_int = 0.0f;
    const int qw = min(c - 1, (int)cfg.quant_window);
    auto quantify = [&](const float (&E)[FM]) {
        double ar = 0.0;
#pragma unroll
        for (int r = 1; r < FM - 1; ++r) {
            const bool in = r >= RC - qw && r + 1 <= RC + qw;
            const float sm = E[r + 1] + E[r];
            const double drt = Q.frt[r + 1] - Q.frt[r];
            const double m = (double)sm * drt;
            ar += in ? m * 0.5 : 0.0;
        }
        area = ar * (double)qw;
        float acc = 0.0f;
        F2_FOR_R acc += (r >= RC - qw && r <= RC + qw) ? E[r] : 0.0f;
        obs_int = acc;
    };
    float P[FM];
    if (NO == 1) {
        load_P(0, P);
        float E[FM];
        F2_FOR_R E[r] = P[r];  // (quant_all: 0 + x for the single observation)
        fused::center_envelope<FM>(E, F);
        quantify(E);
        if (!cfg.quant_all) {
            F2_FOR_R P[r] = E[r];
        }
    } else {
        F2_FOR_R P[r] = 0.0f;
#pragma unroll
        for (int o = 0; o < NO; ++o) {
            float A[FM];
            load_P(o, A);
            if (!cfg.quant_all && o == best_obs) {  // (per candidate: the lanes of a group agree)
                fused::center_envelope<FM>(A, F);
                quantify(A);
            }
            F2_FOR_R P[r] += A[r];
        }
        if (cfg.quant_all) {
            float E[FM];
            F2_FOR_R E[r] = P[r];
            fused::center_envelope<FM>(E, F);
            quantify(E);
        }
    }
    double m1 = 0.0, m2 = 0.0, merr_l = 0.0;
    bool hrow = false;
    if (present) {
        // importance-weighted means over observations (fragment_features.py:311-336)
        if (NO == 1 && oi[0] == 1.0f) {
            hrow = ohe_l[0] > 0;
            if (hrow) {
                m1 = omz_l[0];
                m2 = ohe_l[0];
            }
        } else {
            float ws = 0.0f;
#pragma unroll
            for (int o = 0; o < NO; ++o) {
                const bool m = ohe_l[o] > 0;
                hrow = hrow || m;
                const float w32 = m ? oi[o] : oi[o] * 0.0f;
                ws += w32;
            }
            double msum = 0.0;
            int nm = 0;
#pragma unroll
            for (int o = 0; o < NO; ++o) {
                const bool m = ohe_l[o] > 0;
                const float w32 = m ? oi[o] : oi[o] * 0.0f;
                const double wd = (double)w32 / ((double)ws + 1e-20);
                if (wd > 0) {
                    msum += wd;
                    ++nm;
                }
            }
            if (nm > 0) {
#pragma unroll
                for (int o = 0; o < NO; ++o) {
                    const bool m = ohe_l[o] > 0;
                    const float w32 = m ? oi[o] : oi[o] * 0.0f;
                    const double wd = (double)w32 / ((double)ws + 1e-20);
                    if (wd > 0) {
                        const double lw = wd / msum;
                        m1 += omz_l[o] * lw;
                        m2 += ohe_l[o] * lw;
                    }
                }
            }
        }
        merr_l = (m1 - (double)lrec_mz) / (double)lrec_mz * 1e6;  // fragment_features.py:387
        Q.merr[kk] = merr_l;
        int rk = 0;
#pragma unroll
        for (int j = 0; j < KMAX; ++j) {
            const float ib = Q.g_int[j];
            rk += (j < K) && ((ib > g_int_l) || (ib == g_int_l && j > kk));
        }
        Q.ord[rk] = kk;  // position in argsort(intensity)[::-1]
    }
    adh_wave_sync();

    if (alive && sub < 4) Q.feat[sub] = loc;
    // ---- fragment features 17-27, 41-45 and precursor features 4-16, 28: as in adh_fused.hip
    const bool ipos = present && obs_int > 0.0f;
    const bool hpos = present && m2 > 0.0;
    const bool isb = present && fused::rec_type(lrec) == 98, isy = present && fused::rec_type(lrec) == 121;
    const unsigned b_isb = gmask(isb), b_isy = gmask(isy);
    const int n_int = __popc(gmask(ipos)), n_hei = __popc(gmask(hpos)), n_hrows = __popc(gmask(present && hrow));
    const int nb = __popc(b_isb), ny = __popc(b_isy);
    const int lpos = fused::rec_position(lrec);
    int min_y = isy ? lpos : 255, max_b = isb ? lpos : 0;
#pragma unroll
    for (int m = 8; m > 0; m >>= 1) {
        min_y = min(min_y, __shfl_xor(min_y, m, GS));
        max_b = max(max_b, __shfl_xor(max_b, m, GS));
    }
    const bool ov = (isy && lpos < max_b) || (isb && lpos > min_y);
    const int n_ov = __popc(gmask(ov));
    const int n3 = min(K, 3);
    if (present) {
        double *t = Q.u.at.t64[kk];
        t[0] = area;
        t[1] = m2;
        t[2] = (double)g_fin_l;
        t[3] = merr_l;
        t[4] = ov ? area : 0.0;
        t[5] = ov ? merr_l : 0.0;
        // cosine_similarity_a1 (features_utils.py:40-47) of the observation sums
        float tn = 0.0f, fn = 0.0f, dot = 0.0f;
#pragma unroll
        for (int o = 0; o < NO; ++o) tn += tsum[o] * tsum[o];
        tn = sqrtf(tn);
#pragma unroll
        for (int o = 0; o < NO; ++o) fn += rowsum_l[o] * rowsum_l[o];
        fn = sqrtf(fn);
#pragma unroll
        for (int o = 0; o < NO; ++o) dot += rowsum_l[o] * tsum[o];
        const float pr = fn * tn;
        const float score = (float)((double)dot / ((double)pr + 0.0001));
        float *u = Q.u.at.t32[kk];
        u[0] = ipos ? g_fin_l : 0.0f;
        u[1] = hpos ? g_fin_l : 0.0f;
        u[2] = ipos ? score : 0.0f;
        u[3] = isb ? obs_int : 0.0f;
        u[4] = isy ? obs_int : 0.0f;
    }
    adh_wave_sync();
    {
        double s64 = 0.0;
        float s32 = 0.0f;
        {
            double v64[KMAX];
            float v32[KMAX];
            const int c64 = min(sub, 5), c32 = min(max(sub - 6, 0), 4);
#pragma unroll
            for (int k = 0; k < KMAX; ++k) {
                v64[k] = Q.u.at.t64[k][c64];
                v32[k] = Q.u.at.t32[k][c32];
            }
            if (sub == 11) {
#pragma unroll
                for (int i = 0; i < 3; ++i) v64[i] = Q.merr[i < n3 ? Q.ord[i] : 0];
            }
            const int n64 = sub == 11 ? n3 : K;
#pragma unroll
            for (int k = 0; k < KMAX; ++k) {
                s64 += (k < n64) ? v64[k] : 0.0;
                s32 += (k < K) ? v32[k] : 0.0f;
            }
        }
        // the quotients, lane -> (numerator, denominator): see adh_fused.hip
        double num = s64, den = (double)K;
        if (sub == 4 || sub == 5) den = (double)n_ov;
        if (sub == 6) num = (double)n_int;
        if (sub == 7) num = (double)n_hei;
        if (sub == 8) num = (double)s32, den = (double)n_int;
        if (sub == 11) den = (double)n3;
        double omzp_i = 0.0;
        if (sub >= 12) {
            const int i = sub - 12;
            omzp_i = Q.omzp[i];
            num = omzp_i - (double)Q.iso_mz[i];
            den = (double)Q.iso_mz[i];
        }
        if (sub == 9) num = (double)n_present, den = (double)K0;
        const double quo = num / den;
        if (sub < 3) Q.red64[sub] = quo;
        if (sub == 1) Q.red64[3] = s64;
        if (sub >= 12) {
            const double me = quo * 1e6;
            Q.red64[4 + sub - 12] = (omzp_i > 0) ? me * (double)Q.iso_int[sub - 12] : 0.0;
        }
        if (alive) {
            float *ft = Q.feat;
            if (sub == 3) ft[42] = (float)quo;
            if ((sub == 4 || sub == 5) && nb > 0 && ny > 0) {
                if (sub == 4) ft[43] = (float)n_ov;
                ft[40 + sub] = n_ov > 0 ? (float)quo : (sub == 4 ? 0.0f : 15.0f);
            }
            if (sub == 6) ft[22] = s32, ft[20] = (float)quo;
            if (sub == 7) ft[23] = s32, ft[21] = (float)quo;
            if (sub == 8 && n_int > 0) ft[24] = (float)quo;
            if (sub == 9 || sub == 10) {
                const float lg = (float)log((double)s32 + 1.0);
                ft[16 + sub] = ((sub == 9 ? nb : ny) > 0) ? lg : 0.0f;
            }
            if (sub == 11) ft[41] = (float)quo;
            if (sub == 9) ft[28] = (float)quo, ft[17] = (float)NO;
        }
    }
    adh_wave_sync();
    {
        const double mx_a = Q.red64[0], mx_h = Q.red64[1], my = Q.red64[2];
        if (present) {
            const double a = area - mx_a, h = m2 - mx_h, b = (double)g_fin_l - my;
            double *t = Q.u.at.t64[kk];
            t[0] = a * a;
            t[1] = b * b;
            t[2] = a * b;
            t[3] = h * h;
            t[4] = h * b;
        }
    }
    adh_wave_sync();
    {
        double v64[KMAX], s64 = 0.0;
        const int c64 = min(sub, 4);
#pragma unroll
        for (int k = 0; k < KMAX; ++k) v64[k] = Q.u.at.t64[k][c64];
#pragma unroll
        for (int k = 0; k < KMAX; ++k) s64 += (k < K) ? v64[k] : 0.0;
        if (sub < 5) Q.red64[8 + sub] = s64;
    }
    adh_wave_sync();
    if (alive && sub < 2) {
        const double fact = fmax((double)K - 1.0, 0.0);
        const double inv = 1.0 / fact;
        const double cxx = Q.red64[sub ? 11 : 8] * inv, cyy = Q.red64[9] * inv;
        const double cxy = Q.red64[sub ? 12 : 10] * inv;
        const double s0 = sqrt(cxx), s1 = sqrt(cyy);
        double cc = cxy / s1 / s0;
        if (fabs(cc) > 1.0) cc = (cc > 0) ? 1.0 : -1.0;
        const bool on = sub ? (Q.red64[3] > 0.0) : (n_hrows > 0);
        if (on) Q.feat[18 + sub] = (float)cc;
    }
    if (alive && sub == 2) Q.feat[27] = Q.feat[25] - Q.feat[26];
    if (alive && sub == 3) fused::precursor_features<NO>(Q.feat, I, Q.iso_int, Q.iso_mz, Q.spi, Q.hp, &Q.red64[4], oi);
    adh_wave_sync();  // (the term tables are dead: the centred scan rows take their place)

    // =========================== fragment_mobility_correlation (fragment_features.py:430-480) ============
    // fragments whose (enveloped) scan profiles hold any signal over the observations, in order
    // (fragment_features.py:447-452)
    float so_m = 0.0f;
#pragma unroll
    for (int o = 0; o < NO; ++o) {
        if (NO > 1) {
            float Sp[SM];
            load_Sp(o, Sp);
            scan_envelope(Sp, Se);
        }
        float ss = 0.0f;
        F2_FOR_S ss += Se[i];
        so_m += ss;
    }
    const bool keep = present && so_m > 0.0f;
    const unsigned km = gmask(keep);
    const int Km = __popc(km);
    const int am = __popc(km & ((1u << sub) - 1u));
    const bool corr_on = alive && Km >= 3;
    if (keep) Q.mn[am] = g_int_l;
    adh_wave_sync();
    float mnorm_l = 0.0f;
    {
        float isum = 0.0f;
#pragma unroll
        for (int a = 0; a < KMAX; ++a) isum += (a < Km) ? Q.mn[a] : 0.0f;
        if (keep) mnorm_l = g_int_l / isum;
    }
    adh_wave_sync();
    if (keep) Q.mn[am] = mnorm_l;
    float red_row[KMAX];  // row `am` of the normalised correlation matrix, summed over the observations
#pragma unroll
    for (int b = 0; b < KMAX; ++b) red_row[b] = 0.0f;
#pragma unroll
    for (int o = 0; o < NO; ++o) {
        if (NO > 1) {
            adh_wave_sync();  // (the previous observation's Gram matrix has been read)
            float Sp[SM];
            load_Sp(o, Sp);
            scan_envelope(Sp, Se);
        }
        {
            // centred row + std over the scan axis (scoring/utils.py:545-559) and, with the centred row still in
            // registers, its correlation with the template's scan profile (scoring/utils.py:574-647)
            float py[SM];
            F2_FOR_S py[i] = Q.tsp[o][i];
            float mean, sd, ym, ysd;
            moments<SM>(Se, S, mean, sd);
            moments<SM>(py, S, ym, ysd);
            float dot = 0.0f;
            F2_FOR_S {
                float d = Se[i] - mean;
                d = (i < S) ? d : 0.0f;
                if (keep && corr_on) Q.u.cen[am][i] = d;
                dot += d * (py[i] - ym);
            }
            const float cov = dot / (float)S;
            const float smm = sd * ysd;
            if (keep) {
                Q.sd[o][am] = sd;
                Q.ftc2[o][am] = (float)((double)cov / ((double)smm + 1e-12));
            }
        }
        adh_wave_sync();
        // np.dot(profile_centered, profile_centered.T) over the scan axis (BLAS SGEMM in the reference): one MFMA
        // chain per candidate and observation, as adh_feature_im_kernel issues it (same operands, same order of the
        // scan steps), by all 64 lanes for each of the wavefront's four candidates in turn.  Every lane gets here: no
        // early return above, a candidate without a correlation contributes zero operands and ignores the result.
        {
            typedef float floatx4 __attribute__((ext_vector_type(4)));
            const int i = lane & 15, kq = lane >> 4;
            floatx4 d4[ADH_WAVE / GS];
#pragma unroll
            for (int gg = 0; gg < ADH_WAVE / GS; ++gg) {
                const int S_g = __shfl(corr_on ? S : 0, gg * GS);
                const int Km_g = __shfl(Km, gg * GS);
                floatx4 d = {0.0f, 0.0f, 0.0f, 0.0f};
                // (a rolled loop: see adh_features_im.hip on what hipcc 7.2 does to the unrolled chain)
                for (int s0 = 0; s0 < S_g; s0 += 4) {
                    const int sc = s0 + kq;
                    const float v = (i < Km_g && sc < S_g) ? lds[gg].u.cen[i][sc] : 0.0f;
                    d = __builtin_amdgcn_mfma_f32_16x16x4f32(v, v, d, 0, 0, 0);
                }
                d4[gg] = d;
            }
            adh_wave_sync();  // every chain has read its rows: the Gram matrices take their place
#pragma unroll
            for (int gg = 0; gg < ADH_WAVE / GS; ++gg) {
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) lds[gg].u.gram[4 * kq + rr][i] = d4[gg][rr];
            }
        }
        adh_wave_sync();
        if (keep && corr_on) {
            // one (a, b) pair per step, every pair its own float64 division; weighted by the observation importance
            const float sda = Q.sd[o][am];
#pragma unroll
            for (int b = 0; b < KMAX; ++b) {
                const float cov = Q.u.gram[am][b] / (float)S;
                const float smm = sda * Q.sd[o][min(b, max(Km - 1, 0))];
                const float cm = (float)((double)cov / ((double)smm + 1e-12));
                red_row[b] += cm * oi[o];
            }
        }
    }
    if (keep && corr_on) {
        float acc = 0.0f;
#pragma unroll
        for (int b = 0; b < KMAX; ++b) acc += (b < Km) ? red_row[b] * Q.mn[b] : 0.0f;
        Q.ml[am] = acc;
    }
    adh_wave_sync();
    if (corr_on && sub == 0) {
        float lsum = 0.0f;
#pragma unroll
        for (int a = 0; a < KMAX; ++a) lsum += (a < Km) ? Q.ml[a] : 0.0f;
        Q.feat[29] = (float)((double)lsum / (double)Km);
        float dot = 0.0f;
#pragma unroll
        for (int a = 0; a < KMAX; ++a) {
            float rr = 0.0f;
#pragma unroll
            for (int o = 0; o < NO; ++o) rr += Q.ftc2[o][min(a, max(Km - 1, 0))] * oi[o];
            dot += (a < Km) ? rr * Q.mn[a] : 0.0f;
        }
        Q.feat[30] = dot;
    }
    adh_wave_sync();

    // =========================== profile features (profile_features.py:18-206), experimental_xic ==========
    {
        // normalize_profiles (scoring_utils.py:71-117): centre +- 1 are registers RC-1, RC, RC+1
        float sm = 0.0f;
        sm += P[RC - 1];
        sm += P[RC];
        sm += P[RC + 1];
        const double cn = (double)sm / 3.0;
        const bool cpos = cn > 0;
        // median over fragments per cycle (scoring_utils.py:120-152): 16 x 16 transposes via LDS
#pragma unroll
        for (int half = 0; half < (FM + 15) / 16; ++half) {
            adh_wave_sync();
            if (present) {
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    if (half * 16 + t >= FM) break;
                    const float x = P[half * 16 + t];
                    Q.u.nrmT[t][kk] = cpos ? (float)((double)x / cn) : 0.0f;
                }
            }
            adh_wave_sync();
            float v[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = (j < K) ? Q.u.nrmT[sub][j] : INFINITY;
            fast::sort16(v);
            const int r_lo = (K - 1) / 2, r_hi = K / 2;
            float lo_v = 0.0f, hi_v = 0.0f;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                lo_v = (j == r_lo) ? v[j] : lo_v;
                hi_v = (j == r_hi) ? v[j] : hi_v;
            }
            float m;
            if (K & 1) {
                m = hi_v;
            } else {
                const float s2 = lo_v + hi_v;
                m = (float)((double)s2 / 2.0);
            }
            const int rr = half * 16 + sub;
            const int f = rr + shift;
            if (rr < FM) Q.med[rr] = (alive && f >= 0 && f < F) ? m : 0.0f;
        }
    }
    adh_wave_sync();
    float corr_l = 0.0f;
    {
        // correlation_coefficient (scoring_utils.py:14-68)
        float sx = 0.0f;
        F2_FOR_R sx += Q.med[r];
        const float mx = (float)((double)sx / (double)F);
        float sxx = 0.0f, sy = 0.0f;
        F2_FOR_R {
            float xm = Q.med[r] - mx;
            xm = F2_OK(r) ? xm : 0.0f;
            sxx += xm * xm;
        }
        const double var_x = (double)sxx / (double)F;
        F2_FOR_R sy += P[r];
        const float my = (float)((double)sy / (double)F);
        float sxy = 0.0f, syy = 0.0f;
        F2_FOR_R {
            float xm = Q.med[r] - mx;
            const float ym = P[r] - my;
            xm = F2_OK(r) ? xm : 0.0f;  // (one zero factor is enough)
            sxy += xm * ym;
        }
        F2_FOR_R {
            float ym = P[r] - my;
            ym = F2_OK(r) ? ym : 0.0f;
            syy += ym * ym;
        }
        const double cov = (double)sxy / (double)F;
        const double var_y = (double)syy / (double)F;
        const double var_xy = var_x * var_y;
        corr_l = (var_xy == 0) ? 0.0f : (float)(cov / sqrt(var_xy));
    }
    // frame statistics of every observation against its template frame profile, FWHM in RT / mobility, apex
    // (scoring/utils.py:574-647, profile_features.py:117-193)
    float ftc_l[NO], fw_l[NO], mfw_l[NO];
    int fpeak_l[NO];
#pragma unroll
    for (int o = 0; o < NO; ++o) {
        float A[FM];
        if (NO == 1) {
            F2_FOR_R A[r] = P[r];
        } else {
            load_P(o, A);
            if (!cfg.quant_all && o == best_obs) fused::center_envelope<FM>(A, F);  // (the view's edit, taken again)
            float Sp[SM];
            load_Sp(o, Sp);
            scan_envelope(Sp, Se);
        }
        const float Ff = (float)F;
        float syt = 0.0f;
        F2_FOR_R syt += Q.tfp[o][r];
        const float ym = syt / Ff;
        float qy = 0.0f;
        F2_FOR_R {
            float d = Q.tfp[o][r] - ym;
            d = F2_OK(r) ? d : 0.0f;
            qy += d * d;
        }
        const float ysd = sqrtf(qy / Ff);
        float sy = 0.0f;
        F2_FOR_R sy += A[r];
        const float xmn = sy / Ff;
        float qx = 0.0f, dot = 0.0f;
        F2_FOR_R {
            float d = A[r] - xmn;
            d = F2_OK(r) ? d : 0.0f;
            qx += d * d;
        }
        const float xsd = sqrtf(qx / Ff);
        F2_FOR_R {
            float dx = A[r] - xmn;
            const float dy = Q.tfp[o][r] - ym;
            dx = F2_OK(r) ? dx : 0.0f;
            dot += dx * dy;
        }
        const float cv = dot / Ff;
        const float smm = xsd * ysd;
        ftc_l[o] = (float)((double)cv / ((double)smm + 1e-12));
        float mxv = -1.0f;
        int am_r = 0;
        F2_FOR_R {
            const bool up = F2_OK(r) && A[r] > mxv;
            mxv = up ? A[r] : mxv;
            am_r = up ? r : am_r;
        }
        const double half = (double)mxv / 2.0;
        int n_above = 0;
        F2_FOR_R n_above += (F2_OK(r) && (double)A[r] > half) ? 1 : 0;
        const double frac = (double)n_above / (double)F;
        fw_l[o] = (float)(frac * rt_width);
        fpeak_l[o] = am_r + shift;
        // mobility FWHM (profile_features.py:151-188) on the enveloped scan profile
        float mxs = Se[0];
        F2_FOR_S mxs = (i >= 1 && i < S && Se[i] > mxs) ? Se[i] : mxs;
        const double halfs = (double)mxs / 2.0;
        int n_ab = 0;
        F2_FOR_S n_ab += (i < S && (double)Se[i] > halfs) ? 1 : 0;
        const double fracs = (double)n_ab / (double)S;
        mfw_l[o] = (float)(fracs * mob_width);
        Q.fpeak[sub][o] = fpeak_l[o];
    }
    if (present) Q.corr[kk] = corr_l;
    adh_wave_sync();
    // ---- features 31-40: sums organised as above
    {
        const int r_lo = (K - 1) / 2, r_hi = K / 2;
        if (present) {
            // median apex per observation (profile_features.py:196-198): rank of this fragment's apex
#pragma unroll
            for (int o = 0; o < NO; ++o) {
                const int va = fpeak_l[o];
                int rk = 0;
#pragma unroll
                for (int b = 0; b < 16; ++b) {
                    if (!((gm >> b) & 1u)) continue;
                    const int vb = Q.fpeak[b][o];
                    rk += (vb < va) || (vb == va && b < sub);
                }
                if (rk == r_lo) Q.medlo[o] = va;
                if (rk == r_hi) Q.medhi[o] = va;
            }
            const float cr = Q.corr[Q.ord[kk]];  // correlation of the fragment with intensity rank kk
            // b / y: mask in original order applied to the sorted index array (profile_features.py:94-113)
            const bool b3 = isb && __popc(b_isb & ((1u << sub) - 1u)) < 3;
            const bool y3 = isy && __popc(b_isy & ((1u << sub) - 1u)) < 3;
            float rr = 0.0f, ml = 0.0f, mm = 0.0f;
#pragma unroll
            for (int o = 0; o < NO; ++o) rr += ftc_l[o] * oi[o];
#pragma unroll
            for (int o = 0; o < NO; ++o) ml += fw_l[o] * oi[o];
#pragma unroll
            for (int o = 0; o < NO; ++o) mm += mfw_l[o] * oi[o];
            float *u = Q.u.at.t32[kk];  // (the Gram matrix / transpose buffer are dead)
            u[0] = corr_l;
            u[1] = rr * g_int_l;
            u[2] = ml * g_int_l;
            u[3] = b3 ? cr : 0.0f;
            u[4] = y3 ? cr : 0.0f;
            u[5] = (kk < n3) ? cr : 0.0f;
            Q.u.at.t64[kk][0] = (double)(mm * g_int_l);  // (a float kept in the float64 table: feature 39)
        }
    }
    adh_wave_sync();
    {
        float v32[KMAX], s32 = 0.0f, s39 = 0.0f;
        const int c32 = min(sub, 5);
#pragma unroll
        for (int k = 0; k < KMAX; ++k) v32[k] = Q.u.at.t32[k][c32];
#pragma unroll
        for (int k = 0; k < KMAX; ++k) s32 += (k < K) ? v32[k] : 0.0f;
        if (sub == 7) {
#pragma unroll
            for (int k = 0; k < KMAX; ++k) s39 += (k < K) ? (float)Q.u.at.t64[k][0] : 0.0f;
        }
        const int dn = sub == 0 ? K : (sub == 5 ? n3 : (sub == 3 ? min(nb, 3) : min(ny, 3)));
        const double quo = (double)s32 / (double)dn;
        if (alive) {
            float *ft = Q.feat;
            if (sub == 0) ft[31] = (float)quo;
            if (sub == 5) ft[32] = (float)quo;
            if (sub == 1) ft[33] = s32;
            if (sub == 2) ft[38] = s32;
            if (sub == 3 && nb > 0) ft[34] = (float)quo, ft[35] = (float)nb;
            if (sub == 4 && ny > 0) ft[36] = (float)quo, ft[37] = (float)ny;
            if (sub == 7) ft[39] = s39;
            if (sub == 6) {
                double acc = 0.0;
#pragma unroll
                for (int o = 0; o < NO; ++o) {
                    const double med = (K & 1) ? (double)Q.medhi[o] : (double)(Q.medlo[o] + Q.medhi[o]) / 2.0;
                    const float medpk = (float)med;
                    acc += ((double)medpk - floor((double)F / 2.0)) * (double)oi[o];
                }
                ft[40] = (float)acc;
            }
        }
    }
    adh_wave_sync();

    // ---- output row (candidate.py:403-481)
    if (alive) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int idx = sub + 16 * j;
            if (idx < ADH_NUM_FEATURES) out.features[(int64_t)row * ADH_NUM_FEATURES + idx] = Q.feat[idx];
        }
        if (cfg.collect_fragments && present && kk < top_k) {
            const int64_t o = (int64_t)row * top_k + kk;
            if (out.fragment_precursor_idx) {  // (NULL: the columns that repeat ids / the library are rebuilt later)
                out.fragment_precursor_idx[o] = cand.precursor_idx;
                out.fragment_rank[o] = cand.rank;
                out.fragment_mz_library[o] = __uint_as_float(lrec.a.x);
                out.fragment_mz[o] = lrec_mz;
                out.fragment_position[o] = (uint8_t)lpos;
                out.fragment_number[o] = (uint8_t)(lrec.a.w >> 24);
                out.fragment_type[o] = (uint8_t)(lrec.a.w & 0xFFu);
                out.fragment_charge[o] = (uint8_t)((lrec.a.w >> 16) & 0xFFu);
                out.fragment_loss_type[o] = (uint8_t)((lrec.a.w >> 8) & 0xFFu);
            }
            out.fragment_mz_observed[o] = (float)m1;
            out.fragment_height[o] = (float)m2;
            out.fragment_intensity[o] = (float)area;
            out.fragment_mass_error[o] = (float)merr_l;
            out.fragment_correlation[o] = corr_l;
            if (out.fragment_lib_slot) out.fragment_lib_slot[o] = (uint16_t)lib_slot;
        }
        if (sub == 0) out.valid[row] = 1;
    }
}

template <int FM, int SM, int NO>
__global__ __launch_bounds__(ADH_WAVE, 2) void adh_feature_im_profiles_kernel(
    DevTims run, const CandRecIM *__restrict__ plan, int32_t n_cand, adh_scoring_config_t cfg, int32_t n_iso_cols,
    const unsigned char *__restrict__ scratch, const unsigned char *__restrict__ prof, DevOut out) {
    __shared__ featim2::GroupLds<FM, SM, NO> lds[ADH_WAVE / featim2::GS];
    const int ci = (int)blockIdx.x * (ADH_WAVE / featim2::GS) + (int)threadIdx.x / featim2::GS;
    adh_im_profiles_body<FM, SM, NO, false>(lds, ci, ci < n_cand, run, plan, cfg, n_iso_cols, scratch, prof, out, nullptr, nullptr);
}

// Tile phase and profile phase of four one-observation candidates in ONE kernel (round 6): the profiles never leave
// the CU - no ImProfRec (4 KB per candidate written and read), one set of fixed costs.  The grid is that of
// adh_feature_im_tile4_kernel: the first `list_blocks` blocks take the materialised tiles, here through the
// one-kernel body (adh_feature_im_body<LAY, false>: tiles -> output row); `side`: order of work, see adh_features_im4.hip.
template <int FM, int SM, class LAY>
__global__ __launch_bounds__(ADH_WAVE, 2) void adh_feature_im_fused4_kernel(
    DevTims run, const CandRecIM *__restrict__ plan, int32_t n_cand, const float *__restrict__ iso_table,
    int32_t n_iso_cols, adh_scoring_config_t cfg, const unsigned char *__restrict__ scratch, DevOut out,
    const uint32_t *__restrict__ side, Caps caps, int32_t list_blocks) {
    using namespace featim4;
    extern __shared__ __align__(16) unsigned char smem[];
    if ((int32_t)blockIdx.x < list_blocks) {
        const uint32_t n = side[0];
        const uint32_t *list = side + SIDE_HEAD + n_cand;
        for (uint32_t j = blockIdx.x; j < n; j += (uint32_t)list_blocks) {
            adh_feature_im_body<LAY, false, true>((int)list[j], run, plan, iso_table, n_iso_cols, cfg, scratch, out, caps, nullptr);
            __syncthreads();  // (the next candidate reuses the LDS arrays)
        }
        return;
    }
    WaveTile<FM, SM, 1> &W = *reinterpret_cast<WaveTile<FM, SM, 1> *>(smem);
    const int32_t block = (int32_t)blockIdx.x - list_blocks;
    const int32_t n_order = (int32_t)side[1];
    if (block * NG >= n_order) return;
    int ci = 0;
    const TileOut t = tile4_phase<FM, SM, 1>(W, run, plan, n_order, block, iso_table, n_iso_cols, cfg, scratch, out, side + SIDE_HEAD, ci);
    adh_im_profiles_body<FM, SM, 1, true>(reinterpret_cast<featim2::GroupLds<FM, SM, 1> *>(smem), ci, t.alive, run, plan, cfg,
                                          n_iso_cols, scratch, nullptr, out, &W.g[threadIdx.x / GS], &t);
}
