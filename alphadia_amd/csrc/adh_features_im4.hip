// adh_features_im4.hip - tile phase of the ion-mobility feature stack, FOUR candidates per wavefront (round 6).
//
// adh_feature_im_kernel<LAY, true> (adh_features_im.hip) reduces a candidate's sparse tiles - the sorted
// (cell, intensity, m/z) entries the gather leaves - to an ImProfRec with ONE candidate per wavefront: its serial
// walks keep 12 - 15 lanes of 64 busy (the precursor entries: one lane per (sum, isotope); the fragment planes: one
// lane per plane; the template list: one lane).  This kernel does the same arithmetic, expression by expression and
// in the same order, in the shape of adh_feature_im_profiles_kernel and adh_fused_kernel: 16 lanes per candidate,
//   * transfer function (quadrupole.py:261-301) and its scan mask: the group's 16 lanes share the (isotope, scan) cells
//   * precursor entries (candidate.py:248-269 collapsed by the gather; precursor_features.py:52-66, quadrupole.py:304-324):
//     staged 16 at a time - every lane decodes one entry and computes its weight exp(-0.1 * distance) and its template
//     term - then lane 3 * role + isotope (15 lanes) walks the staged entries for its one sum and lane 15 folds the
//     template cells, in cell order, into scan profile, frame profile and centre of mass as they complete (no template
//     tile, no cell list)
//   * fragment planes (scoring/utils.py:26-66 inputs, features_utils.py:9-37): lane k < 12 owns plane k and walks ITS
//     entries straight from the scratch block (they are sorted by cell, a plane is a range), with the running sums in
//     registers and the two profiles as [row][fragment] columns in LDS
// so a wavefront's serial chain is the longest list of FOUR candidates, where the old kernel's was one candidate's.
// One observation (plan classes 0 and ADH_CLASS_IM_SMALL with the fixed layouts), up to three isotopes, sparse
// tiles.  A candidate whose tiles had to be materialised (ADH_IM_MODE_DENSE, rare) is put on a list and handled by
// the old kernel's body (adh_feature_im_list_kernel); two observations keep the old kernel.  ADH_DEBUG_IM_TILE1=1
// brings the old kernel back everywhere; the GPU suite holds both to identical records.
#include "adh_device.h"
#include "adh_feature_common.h"

namespace featim4 {

constexpr int GS = 16;
constexpr int KMAX = ADH_IM_PROF_K;

template <int FM, int SM>
struct __attribute__((aligned(16))) TileLds {
    union {
        double qtf[3][SM];        // transfer function per (isotope, scan): dead once the precursor entries are folded ...
        float ffp[FM][KMAX];      // ... fragment frame profiles [cycle][fragment]
    } u;
    float fsp[SM][KMAX];          // fragment scan profiles [scan][fragment]
    double s_w[GS], s_t[GS];      // staged precursor entries: weight, template term
    uint32_t s_cell[GS];          //   scan << 16 | cycle << 4 | isotope
    float s_x[GS], s_y[GS];
    float qmask[SM];
    float tsp_raw[SM], tfp_raw[FM];
    float iso_int[4], iso_mz[4];
    int pl_beg[GS], pl_end[GS];
};

}  // namespace featim4

// `dense_list`: [0] = number of candidates left to adh_feature_im_list_kernel, [1 ...] their positions in `plan`
template <int FM, int SM>
__global__ __launch_bounds__(ADH_WAVE, 2) void adh_feature_im_tile4_kernel(
    DevTims run, const CandRecIM *__restrict__ plan, int32_t n_cand, const float *__restrict__ iso_table,
    int32_t n_iso_cols, adh_scoring_config_t cfg, const unsigned char *__restrict__ scratch, DevOut out,
    unsigned char *__restrict__ prof, uint32_t *__restrict__ dense_list) {
    using namespace featim4;
    typedef ImProfRec<FM, SM, 1> Rec;
    __shared__ TileLds<FM, SM> lds[ADH_WAVE / GS];
    const int lane = threadIdx.x;
    const int g = lane / GS, sub = lane % GS;
    const int gbase = g * GS;
    TileLds<FM, SM> &Q = lds[g];
    const int ci = (int)blockIdx.x * (ADH_WAVE / GS) + g;
    bool alive = ci < n_cand;
    const CandRecIM &r = plan[alive ? ci : 0];
    alive = alive && !(r.flags & ADH_FLAG_SKIP);
    const unsigned char *block = scratch + r.scratch_off;
    const uint32_t *header = reinterpret_cast<const uint32_t *>(block);
    uint4 h4 = make_uint4(0u, 0u, 0u, 0u);
    uint32_t h_pe = 0u;
    if (alive) {
        h4 = *reinterpret_cast<const uint4 *>(header);
        h_pe = header[4];
    }
    const int K0 = (int)h4.x;
    alive = alive && K0 > 0;
    if (alive && sub == 0 && out.stat_matched_peaks) out.stat_matched_peaks[r.row] = h4.y;
    if (alive && h4.w != ADH_IM_MODE_COMPACT) {  // materialised tiles: the old kernel's body takes this candidate
        if (sub == 0) dense_list[1u + atomicAdd(dense_list, 1u)] = (uint32_t)ci;
        alive = false;
    }
    const int n_fe = alive ? (int)h4.z : 0, n_pe = alive ? (int)h_pe : 0;
    const int L = run.cycle_len, z = run.zeroth;
    const int c0 = (r.frame_start - z) / L;
    const int F = alive ? (r.frame_stop - z) / L - c0 : 1;
    const int S = alive ? r.scan_stop - r.scan_start : 1;
    const int I = min(min(n_iso_cols, (int)cfg.top_k_isotopes), 3);
    const int SF = S * F;
    const ImEntry *const entries = reinterpret_cast<const ImEntry *>(block + adh_scratch_frag_off(r.k_cap));

    if (sub < I) {
        Q.iso_int[sub] = alive ? iso_table[(int64_t)r.row * n_iso_cols + sub] : 0.0f;
        const double off = (double)sub * 1.0033548350700006 / (double)r.charge;
        Q.iso_mz[sub] = (float)off + r.precursor_mz;
    }
    adh_wave_sync();

    // ---- quadrupole transfer function per (isotope, scan) (quadrupole.py:261-301), its mean over the isotopes
    // (candidate.py:287-289: the mask the fragment tile is multiplied with, candidate.py:290)
    {
        // (the quadrupole rows of the candidate's scans first, all loads in flight together: the scan profiles' bytes
        // are idle until the fragment planes)
        double *const cy_lo = reinterpret_cast<double *>(&Q.fsp[0][0]), *const cy_hi = cy_lo + SM;
        static_assert(sizeof(Q.fsp) >= 2 * SM * sizeof(double), "the cycle rows fit the scan profiles");
        const int obs0 = (int)r.obs[0];
        double2 cyv[(SM + GS - 1) / GS];
#pragma unroll
        for (int j = 0; j < (SM + GS - 1) / GS; ++j) {
            const int sc = sub + j * GS;
            cyv[j] = make_double2(0.0, 0.0);
            if (alive && sc < S)
                cyv[j] = *reinterpret_cast<const double2 *>(run.cycle + 2 * ((int64_t)obs0 * run.scan_max + (r.scan_start + sc)));
        }
#pragma unroll
        for (int j = 0; j < (SM + GS - 1) / GS; ++j) {
            const int sc = sub + j * GS;
            if (sc < SM) {
                cy_lo[sc] = cyv[j].x;
                cy_hi[sc] = cyv[j].y;
            }
        }
        adh_wave_sync();
        if (alive) {
            const QuadParams qp = adh_quad_params(cfg);
            for (int c = sub; c < I * S; c += GS) {
                const int i = c / S, sc = c - i * S;
                const double x = (double)Q.iso_mz[i];
                Q.u.qtf[i][sc] = featim::logistic(x, cy_lo[sc] + qp.delta_lo, qp.sigma_lo) - featim::logistic(x, cy_hi[sc] + qp.delta_hi, qp.sigma_hi);
            }
        }
    }
    adh_wave_sync();
    for (int sc = sub; sc < SM; sc += GS) {
        double sum = 0;
        if (sc < S && alive)
            for (int i = 0; i < I; ++i) sum += Q.u.qtf[i][sc];
        Q.qmask[sc] = (float)(sum / (double)I);
        Q.tsp_raw[sc] = 0.0f;
    }
    for (int f = sub; f < FM; f += GS) Q.tfp_raw[f] = 0.0f;
    adh_wave_sync();

    // ---- the precursor entries: non-zero (scan, cycle, isotope) cells in that order.
    //   lanes 0 .. 5I-1 (role = lane / I: 0 intensity sum per scan and over the scans; 1, 2 weighted intensity mean;
    //   3, 4 weighted m/z mean around (scan, frame) = (S, 1), precursor_features.py:52-66): sequential sums, one each
    //   lane 15: the template cell (quadrupole.py:304-324) of consecutive entries of one (scan, cycle), folded when the
    //   cell is complete - scan profile (np.sum over the cycle axis), frame profile (over the scan axis), centre of
    //   mass over the cells v > 0 (fragment_features.py:20-68)
    const ImEntry *const pent = entries + n_fe;
    const int role = sub / I, iso = sub - role * I;
    double acc = 0.0;
    float part = 0.0f, tot = 0.0f;
    int cur_sc = -1;
    // (lane 15)
    double t_a = 0.0, t_isum = 0.0, t_ssum = 0.0, t_fsum = 0.0;
    float t_srow = 0.0f, t_tsum = 0.0f;
    int t_cell = -1, t_row = -1;
    auto template_cell_done = [&]() {  // the finished cell t_cell (scan << 12 | cycle) with value (float)t_a
        const int sc = t_cell >> 12, f = t_cell & 0xFFF;
        const float v = (float)t_a;
        if (sc != t_row) {
            if (t_row >= 0) {
                Q.tsp_raw[t_row] = t_srow;
                t_tsum += t_srow;  // (the sum of the scan profile in scan order: untouched scans add +0)
            }
            t_srow = 0.0f;
            t_row = sc;
        }
        t_srow += v;
        Q.tfp_raw[f] = Q.tfp_raw[f] + v;
        if (v > 0.0f) {
            t_isum += (double)v;
            t_ssum += (double)sc * (double)v;
            t_fsum += (double)f * (double)v;
        }
    };
    int n_pe_max = n_pe;
#pragma unroll
    for (int off = 32; off >= GS; off >>= 1) n_pe_max = max(n_pe_max, __shfl_xor(n_pe_max, off));
    ImEntry pnext;
    pnext.cell = 0u, pnext.x = 0.0f, pnext.y = 0.0f;
    if (sub < n_pe) pnext = pent[sub];
    for (int base = 0; base < n_pe_max; base += GS) {
        const int cnt = min(GS, n_pe - base);  // (<= 0: this group is through)
        const ImEntry en = pnext;
        if (base + GS + sub < n_pe) pnext = pent[base + GS + sub];  // (the next round's entry, in flight meanwhile)
        if (sub < cnt) {
            const int sf = (int)en.cell / I, i = (int)en.cell - sf * I, sc = sf / F, f = sf - sc * F;
            const double ds = (double)(sc - S), df = (double)(f - 1);
            Q.s_cell[sub] = (uint32_t)(sc << 16 | f << 4 | i);
            Q.s_x[sub] = en.x;
            Q.s_y[sub] = en.y;
            Q.s_w[sub] = exp(-0.1 * sqrt(ds * ds + df * df));
            const float t = en.x * Q.iso_int[i];
            Q.s_t[sub] = (double)t * Q.u.qtf[i][sc];
        }
        adh_wave_sync();
        if (sub < 5 * I) {
#pragma unroll 4
            for (int u = 0; u < GS; ++u) {
                const uint32_t cp = Q.s_cell[u];
                const float xs = Q.s_x[u], ys = Q.s_y[u];
                const double ws = Q.s_w[u];
                const bool mine = u < cnt && (int)(cp & 15u) == iso;
                const int sc = (int)(cp >> 16);
                const bool fresh = mine && sc != cur_sc;  // role 0: per-scan sums, added up in scan order
                tot += fresh ? part : 0.0f;
                part = fresh ? 0.0f : part;
                cur_sc = fresh ? sc : cur_sc;
                part += mine ? xs : 0.0f;
                const float flag = role <= 2 ? xs : ys;
                const double term = role == 1 ? (double)xs * ws : (role == 3 ? (double)ys * ws : ws);
                acc += mine && flag > 0.0f ? term : 0.0;
            }
        } else if (sub == GS - 1) {
            for (int u = 0; u < cnt; ++u) {
                const uint32_t cp = Q.s_cell[u];
                const int cs = (int)((cp >> 16) << 12 | ((cp >> 4) & 0xFFFu));
                if (cs != t_cell) {
                    if (t_cell >= 0) template_cell_done();
                    t_cell = cs;
                    t_a = 0.0;
                }
                t_a += Q.s_t[u];
            }
        }
        adh_wave_sync();
    }
    double esc = 0.0, efc = 0.0;
    float tsum = 0.0f;
    if (sub == GS - 1) {
        if (t_cell >= 0) template_cell_done();
        if (t_row >= 0) {
            Q.tsp_raw[t_row] = t_srow;
            t_tsum += t_srow;
        }
        esc = (t_isum > 0) ? t_ssum / t_isum : 0.0;
        efc = (t_isum > 0) ? t_fsum / t_isum : 0.0;
        tsum = t_tsum;
    }
    esc = __shfl(esc, gbase + GS - 1);
    efc = __shfl(efc, gbase + GS - 1);
    tsum = __shfl(tsum, gbase + GS - 1);
    // isotope lane i < I: the sums of roles 1 .. 4 of its isotope
    const int il = sub < I ? sub : 0;
    const double vh = __shfl(acc, gbase + I + il), wh = __shfl(acc, gbase + 2 * I + il);
    const double vmz = __shfl(acc, gbase + 3 * I + il), wmz = __shfl(acc, gbase + 4 * I + il);
    const float spi_l = tot + part;
    const double hp_l = (wh > 0) ? vh / wh : 0.0;  // weights are exp(...) > 0: "any non-zero cell" == "w sum > 0"
    const double omzp_l = (wmz > 0) ? vmz / wmz : 0.0;
    adh_wave_sync();

    // ---- the fragment planes.  qtf is dead: its bytes become the frame profiles.
    for (int c = sub; c < FM * KMAX; c += GS) (&Q.u.ffp[0][0])[c] = 0.0f;
    for (int c = sub; c < SM * KMAX; c += GS) (&Q.fsp[0][0])[c] = 0.0f;
    Q.pl_beg[sub] = 0;
    Q.pl_end[sub] = 0;
    adh_wave_sync();
    {
        // a plane's entries are a range of the list (cell = (k * S + scan) * F + cycle, sorted): the group looks at
        // every entry's plane once
        const float inv_sf = 1.0f / (float)SF;
        auto plane_of = [&](uint32_t cell) -> int {
            int q = (int)((float)cell * inv_sf);
            const int rem = (int)cell - q * SF;
            q += rem >= SF ? 1 : (rem < 0 ? -1 : 0);
            return q;
        };
        for (int e0 = sub; e0 < n_fe; e0 += 4 * GS) {
            uint32_t cc[4], cp[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = e0 + u * GS;
                cc[u] = e < n_fe ? entries[e].cell : 0u;
                cp[u] = (e < n_fe && e > 0) ? entries[e - 1].cell : 0u;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = e0 + u * GS;
                if (e < n_fe) {
                    const int pc = plane_of(cc[u]);
                    const int pp = e > 0 ? plane_of(cp[u]) : -1;
                    if (pc != pp) {
                        Q.pl_beg[pc] = e;
                        if (pp >= 0) Q.pl_end[pp] = e;
                    }
                    if (e == n_fe - 1) Q.pl_end[pc] = n_fe;
                }
            }
        }
    }
    adh_wave_sync();
    double ohe_l = 0.0, omz_l = 0.0;
    {
        const bool fl = alive && sub < K0;
        const int beg = fl ? Q.pl_beg[sub] : 0, end = fl ? Q.pl_end[sub] : 0;
        const int psf = sub * SF;
        const double inv_f = 1.0 / (double)F;
        double vi = 0.0, wi = 0.0, vm = 0.0, wm = 0.0;
        float fs = 0.0f;
        int cur = -1;
        // a lane's entries arrive through a ring of PF registers: PF loads in flight per lane (the walk is a chain of
        // memory latencies otherwise: one dependent 12-byte load per entry)
        constexpr int PF = 8;
        ImEntry ring[PF];
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            ring[j].cell = 0u, ring[j].x = 0.0f, ring[j].y = 0.0f;
            if (beg + j < end) ring[j] = entries[beg + j];
        }
        for (int e0 = beg; __any(e0 < end); e0 += PF) {
#pragma unroll
            for (int j = 0; j < PF; ++j) {
                const int e = e0 + j;
                const ImEntry en = ring[j];
                if (e + PF < end) ring[j] = entries[e + PF];
                if (e < end) {
                    const int rem = (int)en.cell - psf;
                    int sc = (int)((double)rem * inv_f);  // exact quotient: float64 estimate, one fix-up
                    if (rem - sc * F >= F) ++sc;
                    const int f = rem - sc * F;
                    const double ds = (double)sc - esc, df = (double)f - efc;
                    const double w = exp(-0.1 * sqrt(ds * ds + df * df));
                    const float v = en.x * Q.qmask[sc];  // candidate.py:290
                    const float y = en.y;
                    if (sc != cur) {  // the cells of a scan are consecutive: its sum is complete
                        if (cur >= 0) Q.fsp[cur][sub] = fs;
                        fs = 0.0f;
                        cur = sc;
                    }
                    fs += v;
                    Q.u.ffp[f][sub] = Q.u.ffp[f][sub] + v;
                    const double tm = (double)y * w;  // (w > 0: the product is > 0 exactly when the m/z channel is)
                    vi += v > 0.0f ? (double)v * w : 0.0;  // (adding 0.0 leaves a sum as it is)
                    wi += v > 0.0f ? w : 0.0;
                    vm += tm > 0.0 ? tm : 0.0;
                    wm += tm > 0.0 ? w : 0.0;
                }
            }
        }
        if (fl) {
            if (cur >= 0) Q.fsp[cur][sub] = fs;
            ohe_l = (wi > 0) ? vi / wi : 0.0;  // weights are exp(...) > 0: "any non-zero cell" == "wi > 0"
            omz_l = (wm > 0) ? vm / wm : 0.0;
        }
    }
    adh_wave_sync();

    // ---- hand-over: the record adh_feature_im_profiles_kernel reads (frame axis centred: entry r <-> cycle r + shift)
    if (!alive) return;
    Rec &rec = reinterpret_cast<Rec *>(prof)[ci];
    const int shift = F / 2 - FM / 2;
    for (int rr = sub; rr < FM; rr += GS) {
        const int f = rr + shift;
        rec.tfp_raw[0][rr] = (f >= 0 && f < F) ? Q.tfp_raw[f] : 0.0f;
    }
    for (int sc = sub; sc < SM; sc += GS) rec.tsp_raw[0][sc] = sc < S ? Q.tsp_raw[sc] : 0.0f;
    for (int c = sub; c < K0 * FM; c += GS) {
        const int k = c / FM, rr = c - k * FM, f = rr + shift;
        rec.ffp[k][0][rr] = (f >= 0 && f < F) ? Q.u.ffp[f][k] : 0.0f;
    }
    for (int c = sub; c < K0 * SM; c += GS) {
        const int k = c / SM, sc = c - k * SM;
        rec.fsp[k][0][sc] = sc < S ? Q.fsp[sc][k] : 0.0f;
    }
    if (sub < K0) {
        rec.ohe[sub][0] = ohe_l;
        rec.omz[sub][0] = omz_l;
    }
    if (sub < 4) {
        const bool on = sub < I;
        rec.hp[sub] = on ? hp_l : 0.0;
        rec.omzp[sub] = on ? omzp_l : 0.0;
        rec.spi[sub] = on ? spi_l : 0.0f;
        rec.iso_int[sub] = on ? Q.iso_int[sub] : 0.0f;
        rec.iso_mz[sub] = on ? Q.iso_mz[sub] : 0.0f;
    }
    if (sub == 0) {
        rec.tsum[0] = tsum;
        rec.K0 = (uint32_t)K0;
    }
}
