// Scalar feature assembly shared by the generic and the register-resident feature kernels:
// short sequential sums executed by ONE lane per candidate, in the reference's order.
// Sources: precursor_features.py:13-102, fragment_features.py:198-427,
// profile_features.py:70-113,141-146,196-204, location_features.py:8-33, candidate.py:362.
#pragma once
#include "adh_device.h"

namespace feat {

// np.corrcoef(x, y)[0, 1] in float64 (sequential sums)
__device__ inline double corrcoef01(const double *x, const float *y, int n) {
    double sx = 0, sy = 0;
    for (int i = 0; i < n; ++i) sx += x[i];
    for (int i = 0; i < n; ++i) sy += (double)y[i];
    double mx = sx / (double)n, my = sy / (double)n;
    double cxx = 0, cyy = 0, cxy = 0;
    for (int i = 0; i < n; ++i) {
        double a = x[i] - mx, b = (double)y[i] - my;
        cxx += a * a;
        cyy += b * b;
        cxy += a * b;
    }
    double fact = fmax((double)n - 1.0, 0.0);
    double inv = 1.0 / fact;
    cxx *= inv;
    cyy *= inv;
    cxy *= inv;
    double s0 = sqrt(cxx), s1 = sqrt(cyy);
    double c = cxy / s1 / s0;
    if (fabs(c) > 1.0) c = (c > 0) ? 1.0 : -1.0;
    return c;
}

// Views of one candidate's LDS arrays.  Per-(fragment, observation) arrays are [k * O + o],
// except ftc which is [o * K + k]; rowsum is indexed through kmap (pre-compaction rows).
struct Assemble {
    const DevRun *run;
    const CandRec *rec;
    float *featv;
    const float *iso_int, *iso_mz, *spi, *oi, *tsum, *rowsum, *g_fin, *g_int, *obs_int, *corr, *ftc,
        *fw, *medpk;
    const double *omzp, *hp, *ohe, *area, *height, *merr;
    const int *kmap, *ord;
    const uint8_t *g_type, *g_pos;
    int n_present, K0;
    float top3;
};

// features 0-16, 28
__device__ __forceinline__ void assemble_precursor(const Assemble &q, int I, int O) {
        float *ft = q.featv;
        ft[28] = (float)((double)q.n_present / (double)q.K0);  // candidate.py:362
        // location_features.py:8-33
        if (q.run) {  // the ion-mobility kernel fills 0-3 itself (float64 rt / mobility arrays)
            ft[0] = q.run->mobility[q.rec->scan_start] - q.run->mobility[q.rec->scan_stop - 1];
            ft[1] = q.run->rt[q.rec->frame_stop - 1] - q.run->rt[q.rec->frame_start];
            ft[2] = q.run->rt[q.rec->frame_center];
            ft[3] = q.run->mobility[q.rec->scan_center];
        }

        // precursor_features.py:13-102
        int amax = 0;
        for (int i = 1; i < I; ++i)
            if (q.iso_int[i] > q.iso_int[amax]) amax = i;
        float w4 = 0, w5 = 0, f6 = 0, f7 = 0;
        for (int i = 0; i < I; ++i) {
            float a = 0;
            for (int o = 0; o < O; ++o) a += q.spi[i] * q.oi[o];
            if (i == 0) w4 = a;
            if (i == amax) w5 = a;
            f6 += a;
            f7 += a * q.iso_int[i];
        }
        ft[4] = w4;
        ft[5] = w5;
        ft[6] = f6;
        ft[7] = f7;
        double wme = 0;
        for (int i = 0; i < I; ++i)
            if (q.omzp[i] > 0) {
                double me = (q.omzp[i] - (double)q.iso_mz[i]) / (double)q.iso_mz[i] * 1e6;
                wme += me * (double)q.iso_int[i];
            }
        ft[8] = (float)wme;
        ft[9] = (float)fabs(wme);
        ft[10] = (float)((double)q.iso_mz[0] + wme * 1e-6 * (double)q.iso_mz[0]);
        ft[11] = (float)q.hp[0];
        ft[12] = (float)q.hp[amax];
        {
            double a = 0, b = 0;
            for (int i = 0; i < I; ++i) a += q.hp[i];
            for (int i = 0; i < I; ++i) b += q.hp[i] * (double)q.iso_int[i];
            ft[13] = (float)a;
            ft[14] = (float)b;
        }
        {
            // save_corrcoeff (scoring/utils.py:478-510): (f32, f32) and (f32, f64)
            float sx = 0, sy = 0;
            double sh = 0;
            for (int i = 0; i < I; ++i) sx += q.iso_int[i];
            for (int i = 0; i < I; ++i) sy += q.spi[i];
            for (int i = 0; i < I; ++i) sh += q.hp[i];
            float xb = (float)((double)sx / (double)I), yb = (float)((double)sy / (double)I);
            double hb = sh / (double)I;
            float num = 0, sxx = 0, syy = 0;
            for (int i = 0; i < I; ++i) num += (q.iso_int[i] - xb) * (q.spi[i] - yb);
            for (int i = 0; i < I; ++i) sxx += (q.iso_int[i] - xb) * (q.iso_int[i] - xb);
            for (int i = 0; i < I; ++i) syy += (q.spi[i] - yb) * (q.spi[i] - yb);
            float den = sqrtf(sxx * syy);
            ft[15] = (float)((double)num / ((double)den + 1e-12));
            double numd = 0, shh = 0;
            for (int i = 0; i < I; ++i) numd += (double)(q.iso_int[i] - xb) * (q.hp[i] - hb);
            for (int i = 0; i < I; ++i) shh += (q.hp[i] - hb) * (q.hp[i] - hb);
            double dend = sqrt((double)sxx * shh);
            ft[16] = (float)(numd / (dend + 1e-12));
        }
}

// assemble_precursor for at most four isotopes and OM observations with every table read ONCE, up front, and the loops
// over isotopes / observations unrolled under predicates - the same operations in the same order.  The loop form above
// is a chain of ~80 dependent LDS reads of a single lane, which the register kernels' wavefronts (two per SIMD) cannot
// hide: half of the "fragment features" step of the wide kernels (round 6).  Features 0-3 are the caller's.
template <int OM>
__device__ __forceinline__ void assemble_precursor_regs(const Assemble &q, int I, int O) {
    float *ft = q.featv;
    ft[28] = (float)((double)q.n_present / (double)q.K0);  // candidate.py:362
    float ii[4], im[4], sp[4], oi[OM];
    double oz[4], hp[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {  // (entries at and above I are never used)
        ii[i] = q.iso_int[i];
        im[i] = q.iso_mz[i];
        sp[i] = q.spi[i];
        oz[i] = q.omzp[i];
        hp[i] = q.hp[i];
    }
#pragma unroll
    for (int o = 0; o < OM; ++o) oi[o] = q.oi[o];
    // precursor_features.py:13-102
    int amax = 0;
    float best = ii[0];
#pragma unroll
    for (int i = 1; i < 4; ++i)
        if (i < I && ii[i] > best) best = ii[i], amax = i;
    float w4 = 0, w5 = 0, f6 = 0, f7 = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (i < I) {
            float a = 0;
#pragma unroll
            for (int o = 0; o < OM; ++o)
                if (o < O) a += sp[i] * oi[o];
            if (i == 0) w4 = a;
            if (i == amax) w5 = a;
            f6 += a;
            f7 += a * ii[i];
        }
    ft[4] = w4;
    ft[5] = w5;
    ft[6] = f6;
    ft[7] = f7;
    double wme = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (i < I && oz[i] > 0) {
            double me = (oz[i] - (double)im[i]) / (double)im[i] * 1e6;
            wme += me * (double)ii[i];
        }
    ft[8] = (float)wme;
    ft[9] = (float)fabs(wme);
    ft[10] = (float)((double)im[0] + wme * 1e-6 * (double)im[0]);
    ft[11] = (float)hp[0];
    double hp_amax = hp[0];
#pragma unroll
    for (int i = 1; i < 4; ++i) hp_amax = i == amax ? hp[i] : hp_amax;
    ft[12] = (float)hp_amax;
    {
        double a = 0, b = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < I) a += hp[i];
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < I) b += hp[i] * (double)ii[i];
        ft[13] = (float)a;
        ft[14] = (float)b;
    }
    {
        // save_corrcoeff (scoring/utils.py:478-510): (f32, f32) and (f32, f64)
        float sx = 0, sy = 0;
        double sh = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < I) sx += ii[i];
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < I) sy += sp[i];
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < I) sh += hp[i];
        float xb = (float)((double)sx / (double)I), yb = (float)((double)sy / (double)I);
        double hb = sh / (double)I;
        float num = 0, sxx = 0, syy = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < I) num += (ii[i] - xb) * (sp[i] - yb);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < I) sxx += (ii[i] - xb) * (ii[i] - xb);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < I) syy += (sp[i] - yb) * (sp[i] - yb);
        float den = sqrtf(sxx * syy);
        ft[15] = (float)((double)num / ((double)den + 1e-12));
        double numd = 0, shh = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < I) numd += (double)(ii[i] - xb) * (hp[i] - hb);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < I) shh += (hp[i] - hb) * (hp[i] - hb);
        double dend = sqrt((double)sxx * shh);
        ft[16] = (float)(numd / (dend + 1e-12));
    }
}

// features 17-27, 41-45 (fragment_features.py:198-427)
__device__ __forceinline__ void assemble_fragments(const Assemble &q, int O, int K) {
        float *ft = q.featv;
        ft[17] = (float)O;
        int n_height_rows = 0;
        for (int k = 0; k < K; ++k) {
            int cnt = 0;
            for (int o = 0; o < O; ++o) cnt += q.ohe[k * O + o] > 0;
            n_height_rows += cnt > 0;
        }
        if (n_height_rows > 0) ft[18] = (float)corrcoef01(q.area, q.g_fin, K);
        {
            double sh = 0;
            for (int k = 0; k < K; ++k) sh += q.height[k];
            if (sh > 0.0) ft[19] = (float)corrcoef01(q.height, q.g_fin, K);
        }
        int n_int = 0, n_hei = 0;
        float w_int = 0, w_hei = 0;
        for (int k = 0; k < K; ++k)
            if (q.obs_int[k] > 0.0f) {
                ++n_int;
                w_int += q.g_fin[k];
            }
        for (int k = 0; k < K; ++k)
            if (q.height[k] > 0.0) {
                ++n_hei;
                w_hei += q.g_fin[k];
            }
        ft[20] = (float)((double)n_int / (double)K);
        ft[21] = (float)((double)n_hei / (double)K);
        ft[22] = w_int;
        ft[23] = w_hei;
        if (n_int > 0) {
            // cosine_similarity_a1 (features_utils.py:40-47)
            float tn = 0;
            for (int o = 0; o < O; ++o) tn += q.tsum[o] * q.tsum[o];
            tn = sqrtf(tn);
            float acc = 0;
            int cnt = 0;
            for (int k = 0; k < K; ++k) {
                if (!(q.obs_int[k] > 0.0f)) continue;
                const float *rs = q.rowsum + q.kmap[k] * O;
                float fn = 0, dot = 0;
                for (int o = 0; o < O; ++o) fn += rs[o] * rs[o];
                fn = sqrtf(fn);
                for (int o = 0; o < O; ++o) dot += rs[o] * q.tsum[o];
                float pr = fn * tn;
                float score = (float)((double)dot / ((double)pr + 0.0001));
                acc += score;
                ++cnt;
            }
            ft[24] = (float)((double)acc / (double)cnt);
        }
        float sb = 0, sy = 0;
        int nb = 0, ny = 0;
        for (int k = 0; k < K; ++k)
            if (q.g_type[k] == 98) {
                sb += q.obs_int[k];
                ++nb;
            }
        for (int k = 0; k < K; ++k)
            if (q.g_type[k] == 121) {
                sy += q.obs_int[k];
                ++ny;
            }
        ft[25] = nb > 0 ? (float)log((double)sb + 1.0) : 0.0f;
        ft[26] = ny > 0 ? (float)log((double)sy + 1.0) : 0.0f;
        ft[27] = ft[25] - ft[26];
        {
            int n3 = min(K, 3);
            double a = 0, b = 0;
            for (int i = 0; i < n3; ++i) a += q.merr[q.ord[i]];
            for (int k = 0; k < K; ++k) b += q.merr[k];
            ft[41] = (float)(a / (double)n3);
            ft[42] = (float)(b / (double)K);
        }
        if (nb > 0 && ny > 0) {
            int min_y = 255, max_b = 0;
            for (int k = 0; k < K; ++k) {
                if (q.g_type[k] == 121) min_y = min(min_y, (int)q.g_pos[k]);
                if (q.g_type[k] == 98) max_b = max(max_b, (int)q.g_pos[k]);
            }
            int n_ov = 0;
            double sa = 0, se = 0;
            for (int k = 0; k < K; ++k) {
                bool ov = (q.g_type[k] == 121 && (int)q.g_pos[k] < max_b) ||
                          (q.g_type[k] == 98 && (int)q.g_pos[k] > min_y);
                if (ov) {
                    ++n_ov;
                    sa += q.area[k];
                    se += q.merr[k];
                }
            }
            ft[43] = (float)n_ov;
            if (n_ov > 0) {
                ft[44] = (float)(sa / (double)n_ov);
                ft[45] = (float)(se / (double)n_ov);
            } else {
                ft[44] = 0.0f;
                ft[45] = 15.0f;
            }
        }
}

// features 0-28, 41-45
__device__ __forceinline__ void assemble_part1(const Assemble &q, int I, int O, int K) {
    assemble_precursor(q, I, O);
    assemble_fragments(q, O, K);
}

// features 31-38, 40
__device__ __forceinline__ void assemble_part2(const Assemble &q, int O, int K, int F) {
        float *ft = q.featv;
        float sm = 0;
        for (int k = 0; k < K; ++k) sm += q.corr[k];
        ft[31] = (float)((double)sm / (double)K);
        ft[32] = q.top3;
        float dot = 0;
        for (int k = 0; k < K; ++k) {
            float rr = 0;
            for (int o = 0; o < O; ++o) rr += q.ftc[o * K + k] * q.oi[o];
            dot += rr * q.g_int[k];
        }
        ft[33] = dot;
        // b / y: mask in original order applied to the sorted index array (profile_features.py:94-113)
        int nbi = 0, nyi = 0;
        float sbv = 0, syv = 0;
        for (int k = 0; k < K; ++k) {
            if (q.g_type[k] == 98) {
                if (nbi < 3) sbv += q.corr[q.ord[k]];
                ++nbi;
            }
        }
        for (int k = 0; k < K; ++k) {
            if (q.g_type[k] == 121) {
                if (nyi < 3) syv += q.corr[q.ord[k]];
                ++nyi;
            }
        }
        if (nbi > 0) {
            ft[34] = (float)((double)sbv / (double)min(nbi, 3));
            ft[35] = (float)nbi;
        }
        if (nyi > 0) {
            ft[36] = (float)((double)syv / (double)min(nyi, 3));
            ft[37] = (float)nyi;
        }
        float agg = 0;
        for (int k = 0; k < K; ++k) {
            float ml = 0;
            for (int o = 0; o < O; ++o) ml += q.fw[k * O + o] * q.oi[o];
            agg += ml * q.g_int[k];
        }
        ft[38] = agg;
        double acc = 0;
        for (int o = 0; o < O; ++o) {
            double delta = (double)q.medpk[o] - floor((double)F / 2.0);
            acc += delta * (double)q.oi[o];
        }
        ft[40] = (float)acc;
}

}  // namespace feat
