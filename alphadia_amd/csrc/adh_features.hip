// adh_features.hip - kernel 2 of the scoring path: the 46-feature stack.
//
// One 64-lane wavefront per candidate; the candidate's XIC tile (written by the
// gather kernel) is loaded coalesced into LDS and everything from the quadrupole
// transfer function onwards is computed there.  Replaces
//   Candidate.process (after get_dense)  alphadia/search/scoring/containers/candidate.py:278-481
//   quadrupole transfer fn / template    alphadia/search/scoring/quadrupole.py:261-335
//   profiles and envelopes               alphadia/search/scoring/utils.py:26-66
//   location / precursor / fragment / profile features
//                                        alphadia/search/scoring/features/ (all modules)
//   correlation helpers                  alphadia/search/scoring/scoring_utils.py:14-152,
//                                        alphadia/search/scoring/utils.py:478-647
//
// Design notes
//   * the duplicated "scan" axis of non-IM data (alpharaw_jit.py:326-333) is never
//     materialised: every sum over the two identical scan slots is x + x
//   * weighted_center_mean's exp() weights (features_utils.py:9-25) depend only on
//     (observation, scan, cycle): one LDS table per candidate instead of one exp per
//     non-zero cell per fragment
//   * float32 reductions keep the reference's sequential order (one lane walks the
//     short axis), float64 where Numba's typing makes the expression float64, so the
//     kernel agrees with the CPU restatement bit for bit wherever libm agrees
//   * compile with -ffp-contract=off
#include "adh_device.h"
#include "adh_feature_common.h"

namespace feat {

// python slice(start, stop) on length n
__device__ __forceinline__ void py_slice(int start, int stop, int n, int &a, int &b) {
    if (start < 0) start += n;
    if (stop < 0) stop += n;
    a = min(max(start, 0), n);
    b = min(max(stop, 0), n);
    if (b < a) b = a;
}

__device__ __forceinline__ double logistic(double x, double mu, double sigma) {
    double a = (x - mu) / sigma;
    return 1.0 / (1.0 + exp(-a));
}

// LDS regions.  Element counts depend only on the launch capacities.
struct Layout {
    int Kc, Oc, Fc, Ic;
    __host__ __device__ Layout(const Caps &c) : Kc(c.k), Oc(c.o), Fc(c.f), Ic(c.i) {}
    // doubles
    __host__ __device__ int d_wt() const { return 0; }
    __host__ __device__ int d_wtp() const { return d_wt() + Oc * 2 * Fc; }
    __host__ __device__ int d_qtf() const { return d_wtp() + 2 * Fc; }
    __host__ __device__ int d_omz() const { return d_qtf() + Ic * Oc; }
    __host__ __device__ int d_ohe() const { return d_omz() + Kc * Oc; }
    __host__ __device__ int d_pk() const { return d_ohe() + Kc * Oc; }   // [4][Kc]: mzmean,height,area,merr
    __host__ __device__ int d_po() const { return d_pk() + 4 * Kc; }     // [2][Oc]: esc, efc
    __host__ __device__ int d_pi() const { return d_po() + 2 * Oc; }     // [2][Ic]: hp, omzp
    __host__ __device__ int n_double() const { return d_pi() + 2 * Ic; }
    // floats (after the doubles)
    __host__ __device__ int f_tile() const { return 0; }                 // [3][Kc*Oc*Fc]: fi, fm, ffp
    __host__ __device__ int f_prec() const { return f_tile() + 3 * Kc * Oc * Fc; }  // [2][Ic*Fc]
    __host__ __device__ int f_tpl() const { return f_prec() + 2 * Ic * Fc; }        // [2][Oc*Fc]
    __host__ __device__ int f_bp() const { return f_tpl() + 2 * Oc * Fc; }          // [Kc*Fc]
    __host__ __device__ int f_pk() const { return f_bp() + Kc * Fc; }    // [6][Kc]
    __host__ __device__ int f_pko() const { return f_pk() + 6 * Kc; }    // [3][Kc*Oc]
    __host__ __device__ int f_po() const { return f_pko() + 3 * Kc * Oc; }  // [4][Oc]
    __host__ __device__ int f_pi() const { return f_po() + 4 * Oc; }     // [3][Ic]
    __host__ __device__ int f_pf() const { return f_pi() + 3 * Ic; }     // [3][Fc]
    __host__ __device__ int f_feat() const { return f_pf() + 3 * Fc; }
    __host__ __device__ int n_float() const { return f_feat() + ADH_NUM_FEATURES; }
    // ints (after the floats)
    __host__ __device__ int i_pk() const { return 0; }                   // [3][Kc]: present, kmap, ord
    __host__ __device__ int i_pko() const { return i_pk() + 3 * Kc; }    // [Kc*Oc]: fpeak
    __host__ __device__ int i_obs() const { return i_pko() + Kc * Oc; }  // [Oc]
    __host__ __device__ int n_int() const { return i_obs() + Oc; }
    // bytes (after the ints): [5][Kc]
    __host__ __device__ int n_byte() const { return ((5 * Kc + 7) / 8) * 8; }
    __host__ __device__ size_t bytes() const {
        size_t b = (size_t)n_double() * 8;
        b += ((size_t)n_float() * 4 + 7) / 8 * 8;
        b += ((size_t)n_int() * 4 + 7) / 8 * 8;
        b += n_byte();
        return b;
    }
};

}  // namespace feat

size_t adh_feature_lds_bytes(const Caps &c) { return feat::Layout(c).bytes(); }

__global__ __launch_bounds__(ADH_WAVE) void adh_feature_kernel(
    DevRun run, const CandRec *__restrict__ plan, const float *__restrict__ iso_table,
    int32_t n_iso_cols, adh_scoring_config_t cfg, const unsigned char *__restrict__ scratch,
    DevOut out, Caps caps) {
    using namespace feat;
    extern __shared__ __align__(16) unsigned char smem[];
    const Layout lay(caps);
    const int Kc = lay.Kc, Oc = lay.Oc, Fc = lay.Fc, Ic = lay.Ic;
    double *const D = reinterpret_cast<double *>(smem);
    float *const Fl = reinterpret_cast<float *>(smem + (size_t)lay.n_double() * 8);
    int *const In = reinterpret_cast<int *>(reinterpret_cast<unsigned char *>(Fl) +
                                           ((size_t)lay.n_float() * 4 + 7) / 8 * 8);
    uint8_t *const By = reinterpret_cast<uint8_t *>(In) + ((size_t)lay.n_int() * 4 + 7) / 8 * 8;

    const int lane = threadIdx.x;
    const CandRec &r = plan[blockIdx.x];
    if (r.flags & ADH_FLAG_SKIP) return;
    const unsigned char *block = scratch + r.scratch_off;
    const uint32_t *header = reinterpret_cast<const uint32_t *>(block);
    const int K0 = (int)header[0];
    if (K0 == 0) return;  // failed before / in the gather kernel
    const uint32_t row = r.row;
    const int L = run.cycle_len;
    const int c0 = r.frame_start / L;
    const int F = r.frame_stop / L - c0;
    const int O = r.n_obs;
    const int I = min(n_iso_cols, (int)cfg.top_k_isotopes);
    const int OF = O * F;
    const int top_k = out.top_k;
    if (lane == 0 && out.stat_matched_peaks) out.stat_matched_peaks[row] = header[1];

    // tiles
    float *const fi = Fl + lay.f_tile();
    float *const fm = fi + Kc * Oc * Fc;
    float *const ffp = fm + Kc * Oc * Fc;
    float *const pi = Fl + lay.f_prec();
    float *const pm = pi + Ic * Fc;
    float *const tpl = Fl + lay.f_tpl();
    float *const tfp = tpl + Oc * Fc;
    float *const bp = Fl + lay.f_bp();
    // per-o / per-i / per-f floats
    float *const oi = Fl + lay.f_po();
    float *const tsum = oi + Oc;
    float *const qmask = tsum + Oc;
    float *const medpk = qmask + Oc;
    float *const iso_mz = Fl + lay.f_pi();
    float *const iso_int = iso_mz + Ic;
    float *const spi = iso_int + Ic;
    float *const frame_rt = Fl + lay.f_pf();
    float *const med = frame_rt + Fc;
    float *const xm = med + Fc;
    float *const featv = Fl + lay.f_feat();
    int *const present = In + lay.i_pk();
    int *const kmap = present + Kc;
    int *const ord = kmap + Kc;
    int *const fpeak = In + lay.i_pko();
    int *const obs = In + lay.i_obs();
    double *const qtf = D + lay.d_qtf();

    // ---- load the tile: scratch cell ((o*F + f)*K0 + k) -> LDS [k][o][f]
    {
        const float2 *fcells = reinterpret_cast<const float2 *>(block + adh_scratch_frag_off(r.k_cap));
        const int n_fc = K0 * OF;
        for (int c = lane; c < n_fc; c += ADH_WAVE) {
            float2 v = fcells[c];
            int k = c % K0, of = c / K0;
            fi[k * OF + of] = v.x;
            fm[k * OF + of] = v.y;
        }
        const float2 *pcells =
            reinterpret_cast<const float2 *>(block + adh_scratch_prec_off(r.k_cap, O, F));
        for (int c = lane; c < I * F; c += ADH_WAVE) {
            float2 v = pcells[c];
            pi[c] = v.x;
            pm[c] = v.y;
        }
        if (lane < I) {
            iso_int[lane] = iso_table[(int64_t)row * n_iso_cols + lane];
            double off = (double)lane * 1.0033548350700006 / (double)r.charge;  // candidate.py:158-163
            iso_mz[lane] = (float)off + r.precursor_mz;
        }
        if (lane < O) obs[lane] = r.obs[lane];
        if (lane < ADH_NUM_FEATURES) featv[lane] = 0.0f;
    }
    __syncthreads();

    // ---- quadrupole transfer function (quadrupole.py:261-301), n_scans == 1 (non-IM)
    for (int c = lane; c < I * O; c += ADH_WAVE) {
        int i = c / O, o = c - i * O;
        const double *cy = run.cycle + 2 * ((int64_t)obs[o] * run.cycle_scans + r.scan_start);
        double x = (double)iso_mz[i];
        const QuadParams qp = adh_quad_params(cfg);
        qtf[c] = logistic(x, cy[0] + qp.delta_lo, qp.sigma_lo) - logistic(x, cy[1] + qp.delta_hi, qp.sigma_hi);
    }
    __syncthreads();
    if (lane < O) {
        double sum = 0;
        for (int i = 0; i < I; ++i) sum += qtf[i * O + lane];
        qmask[lane] = (float)(sum / (double)I);  // candidate.py:287-289
    }
    __syncthreads();
    for (int c = lane; c < K0 * OF; c += ADH_WAVE) {
        int o = (c % OF) / F;
        fi[c] = fi[c] * qmask[o];  // candidate.py:290
    }
    // template (quadrupole.py:304-324); both scan slots are identical for non-IM data
    for (int c = lane; c < OF; c += ADH_WAVE) {
        int o = c / F, f = c - o * F;
        double acc = 0;
        for (int i = 0; i < I; ++i) {
            float a = pi[i * F + f] * iso_int[i];
            acc += (double)a * qtf[i * O + o];
        }
        tpl[c] = (float)acc;
    }
    __syncthreads();

    // ---- observation importance (quadrupole.py:327-335) and fragment presence (candidate.py:319-329)
    float *const rowsum = Fl + lay.f_pko();
    float *const fw = rowsum + Kc * Oc;
    float *const ftc = fw + Kc * Oc;
    if (lane < O) {
        float sf = 0;
        for (int f = 0; f < F; ++f) sf += tpl[lane * F + f];
        tsum[lane] = sf + sf;  // sum over the two identical scan slots
    }
    for (int k = lane; k < K0; k += ADH_WAVE) {
        float so = 0;
        for (int o = 0; o < O; ++o) {
            float sf = 0;
            for (int f = 0; f < F; ++f) sf += fi[(k * O + o) * F + f];
            float ss = sf + sf;
            rowsum[k * O + o] = ss;
            so += ss;
        }
        present[k] = so > 0.0f;
    }
    __syncthreads();
    {
        float tot = 0;
        for (int o = 0; o < O; ++o) tot += tsum[o];
        if (lane < O) oi[lane] = (tot == 0.0f) ? 1.0f / (float)O : tsum[lane] / tot;
    }
    int K = 0;
    for (int k = 0; k < K0; ++k) {
        if (present[k]) {
            if (lane == 0) kmap[K] = k;
            ++K;
        }
    }
    if (K < 2) return;  // candidate.py:323
    const int n_present = K;
    __syncthreads();

    // ---- surviving fragments (fragment_container.py:104-120)
    float *const g_mzlib = Fl + lay.f_pk();
    float *const g_mz = g_mzlib + Kc;
    float *const g_int = g_mz + Kc;
    float *const g_fin = g_int + Kc;
    float *const obs_int = g_fin + Kc;
    float *const corr = obs_int + Kc;
    uint8_t *const g_type = By;
    uint8_t *const g_loss = g_type + Kc;
    uint8_t *const g_charge = g_loss + Kc;
    uint8_t *const g_number = g_charge + Kc;
    uint8_t *const g_pos = g_number + Kc;
    {
        const LibRec *sel = reinterpret_cast<const LibRec *>(block + 32);
        for (int k = lane; k < K; k += ADH_WAVE) {
            LibRec rec = sel[kmap[k]];
            g_mzlib[k] = rec.mz_library;
            g_mz[k] = rec.mz;
            g_int[k] = rec.intensity;
            g_type[k] = rec.type;
            g_loss[k] = rec.loss_type;
            g_charge[k] = rec.charge;
            g_number[k] = rec.number;
            g_pos[k] = rec.position;
        }
    }
    __syncthreads();
    {
        // apply_mask renormalisation, then the second normalisation of
        // fragment_features (fragment_features.py:218)
        float sum1 = 0;
        for (int k = 0; k < K; ++k) sum1 += g_int[k];
        __syncthreads();
        for (int k = lane; k < K; k += ADH_WAVE) g_int[k] = g_int[k] / sum1;
        __syncthreads();
        float sum2 = 0;
        for (int k = 0; k < K; ++k) sum2 += g_int[k];
        for (int k = lane; k < K; k += ADH_WAVE) g_fin[k] = g_int[k] / sum2;
    }

    // ---- profiles (candidate.py:333-347; scoring/utils.py:26-66)
    for (int c = lane; c < K * OF; c += ADH_WAVE) {
        int k = c / OF, rem = c - k * OF;
        float v = fi[kmap[k] * OF + rem];
        ffp[c] = v + v;
    }
    for (int c = lane; c < OF; c += ADH_WAVE) {
        int f = c % F;
        float x = tpl[c] + tpl[c];
        float rr = x;
        if (f >= 1 && f < F - 1) {
            float xl = tpl[c - 1] + tpl[c - 1];
            float xr = tpl[c + 1] + tpl[c + 1];
            if (x < xl || x < xr) {
                float sm = xl + xr;
                rr = (float)((double)sm / 2.0);
            }
        }
        tfp[c] = rr;
    }
    const int n_frame_rt = F;  // frame_stop - frame_start is a multiple of the cycle length
    for (int f = lane; f < F; f += ADH_WAVE) frame_rt[f] = run.rt[r.frame_start + f * L];
    if (caps.stop_phase == 3) return;

    // =========================== features ===========================
    double *const wt = D + lay.d_wt();
    double *const wtp = D + lay.d_wtp();
    double *const esc = D + lay.d_po();
    double *const efc = esc + Oc;
    double *const hp = D + lay.d_pi();
    double *const omzp = hp + Ic;
    // precursor weight table around (scan, frame) = (S, 1) = (2, 1)
    // (precursor_features.py:52-57, features_utils.py:9-25)
    for (int c = lane; c < 2 * F; c += ADH_WAVE) {
        int sc = c / F, f = c - sc * F;
        double ds = (double)(sc - 2), df = (double)(f - 1);
        double dist = sqrt(ds * ds + df * df);
        wtp[c] = exp(-0.1 * dist);
    }
    // template centre of mass per observation (fragment_features.py:20-68)
    if (lane < O) {
        double isum = 0, ssum = 0, fsum = 0;
        bool any = false;
        for (int sc = 0; sc < 2; ++sc)
            for (int f = 0; f < F; ++f) {
                float v = tpl[lane * F + f];
                if (v > 0.0f) {
                    any = true;
                    isum += (double)v;
                    ssum += (double)sc * (double)v;
                    fsum += (double)f * (double)v;
                }
            }
        esc[lane] = (any && isum > 0) ? ssum / isum : 0.0;
        efc[lane] = (any && isum > 0) ? fsum / isum : 0.0;
    }
    if (lane < I) {
        float sf = 0;
        for (int f = 0; f < F; ++f) sf += pi[lane * F + f];
        spi[lane] = sf + sf;
    }
    __syncthreads();
    for (int c = lane; c < O * 2 * F; c += ADH_WAVE) {
        int o = c / (2 * F), rem = c - o * 2 * F;
        int sc = rem / F, f = rem - sc * F;
        double ds = (double)sc - esc[o], df = (double)f - efc[o];
        double dist = sqrt(ds * ds + df * df);
        wt[c] = exp(-0.1 * dist);
    }
    // precursor heights / observed m/z
    for (int c = lane; c < 2 * I; c += ADH_WAVE) {
        int i = c >> 1, plane = c & 1;
        const float *p = (plane ? pm : pi) + i * F;
        double values = 0, weights = 0;
        bool any = false;
        for (int sc = 0; sc < 2; ++sc)
            for (int f = 0; f < F; ++f) {
                float v = p[f];
                if (v > 0.0f) {
                    any = true;
                    double w = wtp[sc * F + f];
                    values += (double)v * w;
                    weights += w;
                }
            }
        double res = (any && weights > 0) ? values / weights : 0.0;
        if (plane)
            omzp[i] = res;
        else
            hp[i] = res;
    }
    __syncthreads();
    if (caps.stop_phase == 4) return;

    // ---- best profile + centre envelope (fragment_features.py:240-250)
    double *const omz = D + lay.d_omz();
    double *const ohe = D + lay.d_ohe();
    double *const mzmean = D + lay.d_pk();
    double *const height = mzmean + Kc;
    double *const area = height + Kc;
    double *const merr = area + Kc;
    int best_obs = 0;
    if (!cfg.quant_all)
        for (int o = 1; o < O; ++o)
            if (oi[o] > oi[best_obs]) best_obs = o;
    for (int k = lane; k < K; k += ADH_WAVE) {
        float *x;
        if (cfg.quant_all) {
            x = bp + k * F;
            for (int f = 0; f < F; ++f) {
                float a = 0;
                for (int o = 0; o < O; ++o) a += ffp[(k * O + o) * F + f];
                x[f] = a;
            }
        } else {
            x = ffp + (k * O + best_obs) * F;  // a VIEW in the reference: mutated in place
        }
        // center_envelope_1d (fragment_features.py:71-159)
        const int n = F;
        if (n >= 2) {
            if (n % 2 == 0) {
                int cr = n / 2, cl = cr - 1;
                double left = x[cl], right = x[cr];
                for (int i = 1; i <= cl; ++i) {
                    x[cl - i] = (float)fmin(left, (double)x[cl - i]);
                    left = (double)(x[cl - i] + x[cl - i + 1]) * 0.5;
                    x[cr + i] = (float)fmin(right, (double)x[cr + i]);
                    right = (double)(x[cr + i] + x[cr + i - 1]) * 0.5;
                }
            } else {
                int cc = n / 2;
                double left = (double)(x[cc - 1] + x[cc]) * 0.5;
                double right = (double)(x[cc + 1] + x[cc]) * 0.5;
                for (int i = 1; i <= cc; ++i) {
                    x[cc - i] = (float)fmin(left, (double)x[cc - i]);
                    left = (double)(x[cc - i] + x[cc - i + 1]) * 0.5;
                    x[cc + i] = (float)fmin(right, (double)x[cc + i]);
                    right = (double)(x[cc + i] + x[cc + i - 1]) * 0.5;
                }
            }
        }
        if (!cfg.quant_all)
            for (int f = 0; f < F; ++f) bp[k * F + f] = x[f];
        // quantification window, trapezoid area (fragment_features.py:252-273)
        int qw = min(F / 2 - 1, (int)cfg.quant_window);
        int center = F / 2;
        int a, b, ra, rb;
        py_slice(center - qw, center + qw + 1, F, a, b);
        py_slice(center - qw, center + qw + 1, n_frame_rt, ra, rb);
        const float *p = bp + k * F + a;
        int W = b - a;
        double ar = 0;
        for (int i = 0; i + 1 < W && ra + i + 1 < rb; ++i) {
            float sm = p[i + 1] + p[i];
            float drt = frame_rt[ra + i + 1] - frame_rt[ra + i];
            float m = sm * drt;
            ar += (double)m * 0.5;
        }
        area[k] = ar * (double)qw;
        float t = 0;
        for (int i = 0; i < W; ++i) t += p[i];
        obs_int[k] = t;
    }
    // ---- per (fragment, observation) weighted centre means (features_utils.py:9-37)
    for (int c = lane; c < 2 * K * O; c += ADH_WAVE) {
        int plane = c & 1, ko = c >> 1;
        int k = ko / O, o = ko - k * O;
        const float *p = (plane ? fm : fi) + (kmap[k] * O + o) * F;
        const double *w = wt + o * 2 * F;
        double values = 0, weights = 0;
        bool any = false;
        for (int sc = 0; sc < 2; ++sc)
            for (int f = 0; f < F; ++f) {
                float v = p[f];
                if (v > 0.0f) {
                    any = true;
                    double ww = w[sc * F + f];
                    values += (double)v * ww;
                    weights += ww;
                }
            }
        double res = (any && weights > 0) ? values / weights : 0.0;
        if (plane)
            omz[ko] = res;
        else
            ohe[ko] = res;
    }
    __syncthreads();
    // importance-weighted means over observations (fragment_features.py:311-336)
    for (int k = lane; k < K; k += ADH_WAVE) {
        float ws = 0;
        for (int o = 0; o < O; ++o) {
            bool m = ohe[k * O + o] > 0;
            float w32 = m ? oi[o] : oi[o] * 0.0f;
            ws += w32;
        }
        double msum = 0;
        int nm = 0;
        for (int o = 0; o < O; ++o) {
            bool m = ohe[k * O + o] > 0;
            float w32 = m ? oi[o] : oi[o] * 0.0f;
            double w = (double)w32 / ((double)ws + 1e-20);
            if (w > 0) {
                msum += w;
                ++nm;
            }
        }
        double m1 = 0, m2 = 0;
        if (nm > 0)
            for (int o = 0; o < O; ++o) {
                bool m = ohe[k * O + o] > 0;
                float w32 = m ? oi[o] : oi[o] * 0.0f;
                double w = (double)w32 / ((double)ws + 1e-20);
                if (w > 0) {
                    double lw = w / msum;
                    m1 += omz[k * O + o] * lw;
                    m2 += ohe[k * O + o] * lw;
                }
            }
        mzmean[k] = m1;
        height[k] = m2;
        merr[k] = (m1 - (double)g_mz[k]) / (double)g_mz[k] * 1e6;  // fragment_features.py:387
        // position of k in argsort(intensity)[::-1]
        int rk = 0;
        float ia = g_int[k];
        for (int b = 0; b < K; ++b) {
            float ib = g_int[b];
            rk += (ib > ia) || (ib == ia && b > k);
        }
        ord[rk] = k;
    }
    __syncthreads();
    if (caps.stop_phase == 5) return;

    // ---- scalar feature assembly by lane 0 (short sequential float sums)
    Assemble asmv;
    asmv.run = &run;
    asmv.rec = &r;
    asmv.featv = featv;
    asmv.iso_int = iso_int; asmv.iso_mz = iso_mz; asmv.spi = spi; asmv.oi = oi; asmv.tsum = tsum;
    asmv.rowsum = rowsum; asmv.g_fin = g_fin; asmv.g_int = g_int; asmv.obs_int = obs_int;
    asmv.corr = corr; asmv.ftc = ftc; asmv.fw = fw; asmv.medpk = medpk;
    asmv.omzp = omzp; asmv.hp = hp; asmv.ohe = ohe; asmv.area = area; asmv.height = height;
    asmv.merr = merr; asmv.kmap = kmap; asmv.ord = ord; asmv.g_type = g_type; asmv.g_pos = g_pos;
    asmv.n_present = n_present; asmv.K0 = K0; asmv.top3 = 0.0f;
    if (lane == 0) assemble_part1(asmv, I, O, K);
    if (caps.stop_phase == 6) return;

    // =========================== profile features (profile_features.py:18-206)
    // fi / fm are dead from here on: reuse them as isl[K][F] and nrm[K][F]
    __syncthreads();
    float *isl = fi, *nrm = fm;
    float *cen = fm;  // non-xic path: centred profiles [K][O][F] (nrm unused there)
    float top3_nonxic = 0.0f;
    if (cfg.experimental_xic) {
        for (int c = lane; c < K * F; c += ADH_WAVE) {
            int k = c / F, f = c - k * F;
            float a = 0;
            for (int o = 0; o < O; ++o) a += ffp[(k * O + o) * F + f];
            isl[c] = a;
        }
        __syncthreads();
        // normalize_profiles (scoring_utils.py:71-117)
        int cidx = F / 2, wa, wb;
        py_slice(cidx - 1, cidx + 2, F, wa, wb);
        for (int k = lane; k < K; k += ADH_WAVE) {
            float sm = 0;
            for (int i = wa; i < wb; ++i) sm += isl[k * F + i];
            double ci = (double)sm / (double)(wb - wa);
            for (int f = 0; f < F; ++f)
                nrm[k * F + f] = (ci > 0) ? (float)((double)isl[k * F + f] / ci) : 0.0f;
        }
        __syncthreads();
        // median over fragments per cycle (scoring_utils.py:120-152) by rank selection
        for (int f = lane; f < F; f += ADH_WAVE) {
            float lo_v = 0, hi_v = 0;
            int r_lo = (K - 1) / 2, r_hi = K / 2;
            for (int a = 0; a < K; ++a) {
                float va = nrm[a * F + f];
                int rk = 0;
                for (int b = 0; b < K; ++b) {
                    float vb = nrm[b * F + f];
                    rk += (vb < va) || (vb == va && b < a);
                }
                if (rk == r_lo) lo_v = va;
                if (rk == r_hi) hi_v = va;
            }
            float m;
            if (K & 1)
                m = hi_v;
            else {
                float sm = lo_v + hi_v;
                m = (float)((double)sm / 2.0);
            }
            med[f] = m;
        }
        __syncthreads();
        // correlation_coefficient (scoring_utils.py:14-68)
        float sx = 0;
        for (int f = 0; f < F; ++f) sx += med[f];
        float mx = (float)((double)sx / (double)F);
        for (int f = lane; f < F; f += ADH_WAVE) xm[f] = med[f] - mx;
        __syncthreads();
        float sxx = 0;
        for (int f = 0; f < F; ++f) sxx += xm[f] * xm[f];
        double var_x = (double)sxx / (double)F;
        for (int k = lane; k < K; k += ADH_WAVE) {
            float sy = 0;
            for (int f = 0; f < F; ++f) sy += isl[k * F + f];
            float my = (float)((double)sy / (double)F);
            float sxy = 0, syy = 0;
            for (int f = 0; f < F; ++f) sxy += xm[f] * (isl[k * F + f] - my);
            for (int f = 0; f < F; ++f) {
                float ym = isl[k * F + f] - my;
                syy += ym * ym;
            }
            double cov = (double)sxy / (double)F;
            double var_y = (double)syy / (double)F;
            double var_xy = var_x * var_y;
            corr[k] = (var_xy == 0) ? 0.0f : (float)(cov / sqrt(var_xy));
        }
    } else {
        // fragment_correlation (scoring/utils.py:513-571): centred rows + std per (k, o)
        for (int c = lane; c < K * O; c += ADH_WAVE) {
            const float *p = ffp + c * F;
            float sm = 0;
            for (int f = 0; f < F; ++f) sm += p[f];
            float mean = sm / (float)F;
            float q = 0;
            for (int f = 0; f < F; ++f) cen[c * F + f] = p[f] - mean;
            for (int f = 0; f < F; ++f) q += cen[c * F + f] * cen[c * F + f];
            fw[c] = sqrtf(q / (float)F);  // std, parked in fw until the FWHM step
        }
        __syncthreads();
        // covariance_matrix = np.dot(profile_centered, profile_centered.T) (scoring/utils.py:559): the
        // one dense contraction of the AlphaRaw path, a K x F x K product per observation.  The
        // reference hands it to BLAS (SGEMM, summation order implementation defined); here it is
        // one MFMA tile: v_mfma_f32_16x16x4_f32 accumulates the 16 x 16 Gram matrix over 4 cycles
        // per instruction (lane l supplies row l % 16, cycle l / 16 of the centred profiles as both
        // operands).  More than 16 fragments fall back to ordered scalar dot products.
        // red = sum_o corr_o * importance_o, list[a] = sum_b red[a][b] * intensity[b]
        __shared__ float gram[16][17], redm[16][17];
        const bool use_mfma = K <= 16;
        if (use_mfma) {
            for (int c = lane; c < 16 * 16; c += ADH_WAVE) redm[c / 16][c % 16] = 0.0f;
            for (int o = 0; o < O; ++o) {
                typedef float floatx4 __attribute__((ext_vector_type(4)));
                floatx4 d = {0.0f, 0.0f, 0.0f, 0.0f};
                const int i = lane & 15, kq = lane >> 4;
                for (int f0 = 0; f0 < F; f0 += 4) {
                    const int f = f0 + kq;
                    const float v = (i < K && f < F) ? cen[(i * O + o) * F + f] : 0.0f;
                    d = __builtin_amdgcn_mfma_f32_16x16x4f32(v, v, d, 0, 0, 0);
                }
                __syncthreads();  // the previous observation's Gram matrix was consumed
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) gram[4 * kq + rr][i] = d[rr];
                __syncthreads();
                for (int a = lane; a < K; a += ADH_WAVE)
                    for (int b = 0; b < K; ++b) {
                        float cov = gram[a][b] / (float)F;
                        float sm = fw[a * O + o] * fw[b * O + o];
                        float cm = (float)((double)cov / ((double)sm + 1e-12));
                        redm[a][b] += cm * oi[o];
                    }
            }
            __syncthreads();
            for (int a = lane; a < K; a += ADH_WAVE) {
                float acc = 0;
                for (int b = 0; b < K; ++b) acc += redm[a][b] * g_int[b];
                corr[a] = acc;
            }
        } else {
            for (int a = lane; a < K; a += ADH_WAVE) {
                float acc = 0;
                for (int b = 0; b < K; ++b) {
                    float red = 0;
                    for (int o = 0; o < O; ++o) {
                        float dot = 0;
                        for (int f = 0; f < F; ++f)
                            dot += cen[(a * O + o) * F + f] * cen[(b * O + o) * F + f];
                        float cov = dot / (float)F;
                        float sm = fw[a * O + o] * fw[b * O + o];
                        float cm = (float)((double)cov / ((double)sm + 1e-12));
                        red += cm * oi[o];
                    }
                    acc += red * g_int[b];
                }
                corr[a] = acc;
            }
        }
        __syncthreads();
        if (lane == 0) {
            // mean correlation of the three most intense fragments (profile_features.py:70-92)
            int n3 = min(K, 3);
            float sm = 0;
            for (int i = 0; i < n3; ++i)
                for (int j = 0; j < n3; ++j) {
                    int a = ord[i], b = ord[j];
                    float red = 0;
                    if (use_mfma) {
                        red = redm[a][b];
                    } else {
                        for (int o = 0; o < O; ++o) {
                            float dot = 0;
                            for (int f = 0; f < F; ++f)
                                dot += cen[(a * O + o) * F + f] * cen[(b * O + o) * F + f];
                            float cov = dot / (float)F;
                            float sd = fw[a * O + o] * fw[b * O + o];
                            float cm = (float)((double)cov / ((double)sd + 1e-12));
                            red += cm * oi[o];
                        }
                    }
                    sm += red;
                }
            top3_nonxic = (float)((double)sm / (double)(n3 * n3));
        }
    }
    __syncthreads();
    float top3 = 0.0f;
    if (lane == 0) {
        int n3 = min(K, 3);
        if (cfg.experimental_xic) {
            float sm = 0;
            for (int i = 0; i < n3; ++i) sm += corr[ord[i]];
            top3 = (float)((double)sm / (double)n3);
        } else {
            top3 = top3_nonxic;
        }
    }
    __syncthreads();
    // fragment-vs-template frame correlation (scoring/utils.py:574-647), FWHM and apex
    const float rt_width = run.rt[r.frame_stop - 1] - run.rt[r.frame_start];
    for (int c = lane; c < K * O; c += ADH_WAVE) {
        int k = c / O, o = c - k * O;
        const float *px = ffp + c * F;
        const float *py = tfp + o * F;
        float sy = 0;
        for (int f = 0; f < F; ++f) sy += py[f];
        float ym = sy / (float)F;
        float qy = 0;
        for (int f = 0; f < F; ++f) {
            float d = py[f] - ym;
            qy += d * d;
        }
        float ysd = sqrtf(qy / (float)F);
        float sx = 0;
        for (int f = 0; f < F; ++f) sx += px[f];
        float xmn = sx / (float)F;
        float qx = 0;
        for (int f = 0; f < F; ++f) {
            float d = px[f] - xmn;
            qx += d * d;
        }
        float xsd = sqrtf(qx / (float)F);
        float dot = 0;
        for (int f = 0; f < F; ++f) dot += (px[f] - xmn) * (py[f] - ym);
        float cov = dot / (float)F;
        float sm = xsd * ysd;
        ftc[o * K + k] = (float)((double)cov / ((double)sm + 1e-12));
        // FWHM in RT (profile_features.py:117-146) and apex (profile_features.py:192-193)
        float mxv = px[0];
        int am = 0;
        for (int f = 1; f < F; ++f)
            if (px[f] > mxv) {
                mxv = px[f];
                am = f;
            }
        double half = (double)mxv / 2.0;
        int n_above = 0;
        for (int f = 0; f < F; ++f) n_above += ((double)px[f] > half);
        double frac = (double)n_above / (double)F;
        fpeak[c] = am;
        fw[c] = (float)(frac * (double)rt_width);  // std values parked here are dead by now
    }
    __syncthreads();
    if (lane < O) {
        // median of the apex index over fragments (profile_features.py:196-198)
        int o = lane;
        int lo_v = 0, hi_v = 0, r_lo = (K - 1) / 2, r_hi = K / 2;
        for (int a = 0; a < K; ++a) {
            int va = fpeak[a * O + o];
            int rk = 0;
            for (int b = 0; b < K; ++b) {
                int vb = fpeak[b * O + o];
                rk += (vb < va) || (vb == va && b < a);
            }
            if (rk == r_lo) lo_v = va;
            if (rk == r_hi) hi_v = va;
        }
        double m = (K & 1) ? (double)hi_v : (double)(lo_v + hi_v) / 2.0;
        medpk[o] = (float)m;
    }
    __syncthreads();
    if (lane == 0) {
        asmv.top3 = top3;
        assemble_part2(asmv, O, K, F);
    }
    __syncthreads();

    // ---- write the row: features, fragment table, valid flag (candidate.py:403-481)
    if (lane < ADH_NUM_FEATURES) out.features[(int64_t)row * ADH_NUM_FEATURES + lane] = featv[lane];
    if (cfg.collect_fragments) {
        const int n = min(K, top_k);
        const int64_t base = (int64_t)row * top_k;
        for (int k = lane; k < n; k += ADH_WAVE) {
            if (out.fragment_precursor_idx) {  // (NULL: the columns that repeat ids / the library are rebuilt later)
                out.fragment_precursor_idx[base + k] = r.precursor_idx;
                out.fragment_rank[base + k] = r.rank;
                out.fragment_mz_library[base + k] = g_mzlib[k];
                out.fragment_mz[base + k] = g_mz[k];
                out.fragment_position[base + k] = g_pos[k];
                out.fragment_number[base + k] = g_number[k];
                out.fragment_type[base + k] = g_type[k];
                out.fragment_charge[base + k] = g_charge[k];
                out.fragment_loss_type[base + k] = g_loss[k];
            }
            out.fragment_mz_observed[base + k] = (float)mzmean[k];
            out.fragment_height[base + k] = (float)height[k];
            out.fragment_intensity[base + k] = (float)area[k];
            out.fragment_mass_error[base + k] = (float)merr[k];
            out.fragment_correlation[base + k] = corr[k];
            if (out.fragment_lib_slot) {
                const LibRec pick = reinterpret_cast<const LibRec *>(block + 32)[kmap[k]];
                out.fragment_lib_slot[base + k] = (uint16_t)(1 + pick.pad0 + 256 * pick.pad1);
            }
        }
    }
    if (lane == 0) out.valid[row] = 1;
}
