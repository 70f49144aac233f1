// adh_features_im4.hip - tile phase of the ion-mobility feature stack, FOUR candidates per wavefront (round 6).
//
// adh_feature_im_kernel<LAY, true> (adh_features_im.hip) reduces a candidate's sparse tiles - the sorted
// (cell, intensity, m/z) entries the gather leaves - to an ImProfRec with ONE candidate per wavefront: its serial
// walks keep 12 - 15 lanes of 64 busy and every candidate pays the transfer function of all its scans.  This file
// does the same arithmetic, expression by expression and in the same order, in the shape of
// adh_feature_im_profiles_kernel and adh_fused_kernel - 16 lanes per candidate - and splits every step into a
// BALANCED part (all 64 lanes, whichever candidate has the work) and a serial part that only adds:
//   * precursor entries (candidate.py:248-269 collapsed by the gather; precursor_features.py:52-66,
//     quadrupole.py:261-324): staged 16 at a time per candidate - every lane decodes one entry and computes its weight
//     exp(-0.1 * distance), its transfer-function value (one logistic pair for ITS (isotope, scan): a candidate has
//     ~40 precursor entries and 120 (isotope, scan) cells) and its template term - then lane 3 * role + isotope
//     (15 lanes) walks the staged entries for its one sum and lane 15 folds the template cells, in cell order, into
//     scan profile, frame profile and centre of mass as they complete (no template tile, no cell list)
//   * fragment entries (scoring/utils.py:26-66 inputs, features_utils.py:9-37), the entries of the wavefront's FOUR
//     candidates as one list: lane l takes entries l, l + 64, ... whichever candidate they belong to, decodes
//     (plane, scan, cycle), computes the weight around the template centre and leaves (weight, intensity, m/z,
//     scan << 8 | cycle) in an LDS pool; the scans that occur are collected in a bit mask per candidate and the scan mask of the transfer
//     function (candidate.py:287-290) is evaluated for THOSE scans only (~10 of 40), again one (candidate, scan) per
//     lane; then lane k < 12 of a candidate folds plane k - its entries are a range of the list - with the running
//     sums in registers, the weights from the pool and the two profiles as [row][fragment] columns in LDS
// so the expensive float64 exp / logistic work runs on full wavefronts and a wavefront's serial chain is the longest
// plane of four candidates, a dozen instructions per entry.
// One observation (plan classes 0 and ADH_CLASS_IM_SMALL with the fixed layouts), up to three isotopes, sparse
// tiles.  A candidate whose tiles had to be materialised (ADH_IM_MODE_DENSE, rare) is put on a list and handled by
// the old kernel's body in the first blocks of the same grid; two observations keep the old kernel.  ADH_DEBUG_IM_TILE1=1
// brings the old kernel back everywhere; the GPU suite holds both to identical output tables.
#include "adh_device.h"
#include "adh_feature_common.h"

namespace featim4 {

constexpr int GS = 16;
constexpr int NG = ADH_WAVE / GS;
constexpr int KMAX = ADH_IM_PROF_K;
constexpr int NP = 144;  // fragment entries of a wavefront's candidates per pass of the pool (LDS is the kernel's occupancy)

// precursor entries of the wavefront's candidates, PP per pass, as the balanced pass leaves them (PrecPoolT below; they
// live in the pool's bytes: the precursor entries come first)
constexpr int PP = 56;

template <int FM, int SM, int NO>
struct __attribute__((aligned(16))) GroupTile {
    float ffp[NO][FM][KMAX];      // fragment frame profiles [observation][cycle][fragment]
    union {
        float fsp[NO][SM][KMAX];  // fragment scan profiles [observation][scan][fragment] ...
        double cy[NO][2][SM];     // ... before them: the quadrupole rows of the candidate's scans (lower, upper limit) per
    } v;                          //     observation - dead once the scan masks are known (LDS is this kernel's occupancy)
    float qmask[NO][SM];
    float tsp_raw[NO][SM], tfp_raw[NO][FM];
    float iso_int[4], iso_mz[4];
    int pl_beg[NO * GS], pl_end[NO * GS];  // entries of plane p = fragment * O + observation
    // what the balanced passes need to know about the candidate
    const ImEntry *entries;
    double esc[NO], efc[NO], inv_f;
    float inv_sf;
    int n_fe, off, F, SF, S, O;
    const ImEntry *pent;
    int n_pe, poff;
    unsigned long long scans[NO], scans_done[NO];
};

template <int NO>
struct __attribute__((aligned(16))) PrecPoolT {
    double t[NO][PP];             // template term of the entry: (intensity * library isotope intensity) * transfer function
    double r[4][PP];              // x > 0 ? x * w : 0, x > 0 ? w : 0, y > 0 ? y * w : 0, y > 0 ? w : 0  (w: weight around (S, 1))
    uint32_t cell[PP];            // scan << 16 | cycle << 4 | isotope
    float x[PP];
};

template <int FM, int SM, int NO>
struct __attribute__((aligned(16))) WaveTile {
    static_assert(SM <= 64 && FM <= 256 && NO <= 2, "scan bit masks are 64 bits wide, a pool entry names its cycle by a byte and its observation by a bit");
    GroupTile<FM, SM, NO> g[NG];
    union {
        struct {
            double w[NP];         // weight of a fragment entry around the template centre
            float x[NP], y[NP];   // its intensity and m/z
            uint16_t scf[NP];     // observation << 15 | scan << 8 | cycle
        } pool;
        PrecPoolT<NO> prec;
    } u;
    uint16_t need[NG * NO * SM];  // (candidate, observation, scan) triples whose scan mask is due
    uint8_t owner[NP];            // lane (candidate, fragment) a pool slot belongs to
    uint16_t seg_base[ADH_WAVE];  // first pool slot of a lane's planes in this pass ...
    int seg_src[ADH_WAVE];        // ... and the entry it holds
};

// what a group's lanes hold when the tile phase is over (beside the LDS arrays of GroupTile)
struct TileOut {
    bool alive;
    int K0, F, S, I, O;
    double ohe[2], omz[2];  // lane k < K0: weighted centre means of planes (k, o)
    double hp, omzp;        // lane i < I: ... of isotope plane i
    float spi;              // lane i < I: isotope intensity sum
    float tsum[2];          // template sums
};

// The tile phase of four candidates (group g of the wavefront = candidate order[4 * block + g] of `plan`).
template <int FM, int SM, int NO>
__device__ __forceinline__ TileOut tile4_phase(WaveTile<FM, SM, NO> &W, const DevTims &run, const CandRecIM *__restrict__ plan,
                                               int32_t n_cand, int32_t block, const float *__restrict__ iso_table,
                                               int32_t n_iso_cols, const adh_scoring_config_t &cfg,
                                               const unsigned char *__restrict__ scratch, const DevOut &out,
                                               const uint32_t *__restrict__ order, int &ci_out, const int stop = 0) {
    // (`stop`: developer ablation, ADH_DEBUG_IM4 = 1 tables, 6 + the balanced pass over the precursor entries, 2 + their folds, 3 + weights, 4 + scan masks, 5 + folds)
    const int lane = threadIdx.x;
    const int g = lane / GS, sub = lane % GS;
    const int gbase = g * GS;
    GroupTile<FM, SM, NO> &Q = W.g[g];
    // (`order`: the candidates with sparse tiles, heaviest first - adh_im_order_*_kernel below; n_cand of them)
    // Four of a kind per wavefront (one observation) or the ordered candidates dealt out column by column (two
    // observations: a wavefront holds one of every quarter of the order, so a long list has the wavefront's pools to
    // itself).  The one-observation launches are bound by their throughput - a light candidate beside a long list waits
    // for it: fused kernels 841 / 723 -> 942 / 828 us dealt out -, the two-observation launch by its longest
    // wavefront, whose lanes fold two planes each: 444 -> 277 us dealt out (round 6; ADH_IM4_SPREAD: 0 = four of a
    // kind everywhere, 1 = dealt out everywhere, 2 = by observation count)
#ifndef ADH_IM4_SPREAD
#define ADH_IM4_SPREAD 2
#endif
    int oi = block * NG + g;
    const int n_waves = (n_cand + NG - 1) / NG;  // (the grid may hold more: it is sized before the order is known)
    if (ADH_IM4_SPREAD == 1 || (ADH_IM4_SPREAD >= 2 && NO == 2)) oi = g * n_waves + block;
    // (3, measured: one observation in pairs - two of the front half of the order, two of the back half per wavefront)
    if (ADH_IM4_SPREAD == 3 && NO == 1) oi = (g >> 1) * 2 * n_waves + 2 * block + (g & 1);
    bool alive = block < n_waves && oi < n_cand;
    const int ci = alive ? (int)order[oi] : 0;
    ci_out = ci;
    const CandRecIM &r = plan[ci];
    alive = alive && !(r.flags & ADH_FLAG_SKIP);
    const unsigned char *block_p = scratch + r.scratch_off;
    uint4 h4 = make_uint4(0u, 0u, 0u, 0u);
    uint32_t h_pe = 0u;
    if (alive) {
        h4 = *reinterpret_cast<const uint4 *>(block_p);
        h_pe = reinterpret_cast<const uint32_t *>(block_p)[4];
    }
    const int K0 = (int)h4.x;
    alive = alive && K0 > 0;
    if (alive && sub == 0 && out.stat_matched_peaks) out.stat_matched_peaks[r.row] = h4.y;
    alive = alive && h4.w == ADH_IM_MODE_COMPACT;  // (materialised tiles are not in `order`)
    const int n_fe = alive ? (int)h4.z : 0, n_pe = alive ? (int)h_pe : 0;
    const int L = run.cycle_len, z = run.zeroth;
    const int c0 = (r.frame_start - z) / L;
    const int F = alive ? (r.frame_stop - z) / L - c0 : 1;
    const int S = alive ? r.scan_stop - r.scan_start : 1;
    const int O = NO == 1 ? 1 : (alive ? min((int)r.n_obs, NO) : 1);
    const int I = max(min(min(n_iso_cols, (int)cfg.top_k_isotopes), 3), 1);
    const int SF = S * F;
    const ImEntry *const entries = reinterpret_cast<const ImEntry *>(block_p + adh_scratch_frag_off(r.k_cap));
    const QuadParams qp = adh_quad_params(cfg);

    // ---- the candidate's small tables: isotopes, the quadrupole rows of its scans (all loads in flight together)
    {
        constexpr int NJ = (SM + GS - 1) / GS;
        double2 cyv[NO][NJ];
#pragma unroll
        for (int o = 0; o < NO; ++o) {
            const int obs_o = (int)r.obs[o < O ? o : 0];
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int sc = sub + j * GS;
                cyv[o][j] = make_double2(0.0, 0.0);
                if (alive && sc < S)
                    cyv[o][j] = *reinterpret_cast<const double2 *>(run.cycle + 2 * ((int64_t)obs_o * run.scan_max + (r.scan_start + sc)));
            }
        }
        if (sub < I) {
            Q.iso_int[sub] = alive ? iso_table[(int64_t)r.row * n_iso_cols + sub] : 0.0f;
            const double off = (double)sub * 1.0033548350700006 / (double)r.charge;
            Q.iso_mz[sub] = (float)off + r.precursor_mz;
        }
#pragma unroll
        for (int o = 0; o < NO; ++o) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int sc = sub + j * GS;
                if (sc < SM) {
                    Q.v.cy[o][0][sc] = cyv[o][j].x;
                    Q.v.cy[o][1][sc] = cyv[o][j].y;
                    Q.tsp_raw[o][sc] = 0.0f;
                    Q.qmask[o][sc] = 0.0f;
                }
            }
            for (int f = sub; f < FM; f += GS) Q.tfp_raw[o][f] = 0.0f;
        }
        for (int c = sub; c < NO * GS; c += GS) {
            Q.pl_beg[c] = 0x7FFFFFFF;
            Q.pl_end[c] = 0;
        }
        if (sub == 0) {
            Q.entries = entries;
            Q.n_fe = n_fe;
            Q.F = F;
            Q.SF = SF;
            Q.S = S;
            Q.O = O;
            Q.inv_f = 1.0 / (double)F;
            Q.inv_sf = 1.0f / (float)SF;
#pragma unroll
            for (int o = 0; o < NO; ++o) {
                Q.scans[o] = 0ull;
                Q.scans_done[o] = 0ull;
                Q.esc[o] = 0.0;
                Q.efc[o] = 0.0;
            }
        }
    }
    adh_wave_sync();
    TileOut res;
    res.alive = false;
    if (stop == 1) return res;
    // transfer function of (isotope i, observation o, scan sc) (quadrupole.py:261-301)
    auto qtf_at = [&](const GroupTile<FM, SM, NO> &T, int i, int o, int sc) -> double {
        const double x = (double)T.iso_mz[i];
        return featim::logistic(x, T.v.cy[o][0][sc] + qp.delta_lo, qp.sigma_lo) - featim::logistic(x, T.v.cy[o][1][sc] + qp.delta_hi, qp.sigma_hi);
    };

    // ---- the precursor entries: non-zero (scan, cycle, isotope) cells in that order.
    //   lanes 0 .. 5I-1 (role = lane / I: 0 intensity sum per scan and over the scans; 1, 2 weighted intensity mean;
    //   3, 4 weighted m/z mean around (scan, frame) = (S, 1), precursor_features.py:52-66): sequential sums, one each
    //   lane 15: the template cell (quadrupole.py:304-324) of consecutive entries of one (scan, cycle), per
    //   observation, folded when the cell is complete - scan profile (np.sum over the cycle axis), frame profile
    //   (over the scan axis), centre of mass over the cells v > 0 (fragment_features.py:20-68)
    const int role = sub / I, iso = sub - role * I;
    double acc = 0.0;
    float part = 0.0f, tot = 0.0f;
    int cur_sc = -1;
    // (lane 15)
    double t_a[NO], t_isum[NO], t_ssum[NO], t_fsum[NO];
    float t_srow[NO], t_tsum[NO];
#pragma unroll
    for (int o = 0; o < NO; ++o) t_a[o] = t_isum[o] = t_ssum[o] = t_fsum[o] = 0.0, t_srow[o] = t_tsum[o] = 0.0f;
    int t_cell = -1, t_row = -1;
    auto template_cell_done = [&]() {  // the finished cell t_cell (scan << 12 | cycle) with the values (float)t_a[o]
        const int sc = t_cell >> 12, f = t_cell & 0xFFF;
        const bool new_row = sc != t_row;
#pragma unroll
        for (int o = 0; o < NO; ++o) {
            const float v = (float)t_a[o];
            if (new_row) {
                if (t_row >= 0) {
                    Q.tsp_raw[o][t_row] = t_srow[o];
                    t_tsum[o] += t_srow[o];  // (the sum of the scan profile in scan order: untouched scans add +0)
                }
                t_srow[o] = 0.0f;
            }
            t_srow[o] += v;
            // (this lane alone adds to the template's frame profile: through ds_add_f32 - no answer to wait for, the adds
            // to one address arrive in program order - instead of a read, an add and a write in the lane's chain; a cell
            // that is not positive enters the centre-of-mass sums as + 0: round 6, the chain of a long list's wavefront)
            (void)__hip_atomic_fetch_add(&Q.tfp_raw[o][f], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            const double vd = (double)__builtin_fmaxf(v, 0.0f);
            t_isum[o] += vd;
            t_ssum[o] += (double)sc * vd;
            t_fsum[o] += (double)f * vd;
        }
        t_row = sc;
    };
    {
        PrecPoolT<NO> &P = W.u.prec;
        // passes over the four candidates SIDE BY SIDE: every candidate that has entries left puts its next
        // PP / (candidates still busy) of them into the pool - slots [q * share, q * share + take_q) - so that the
        // serial folds of the four run at once
        int pat = 0;  // next precursor entry of this group's candidate
        for (;;) {
            const int prem = max(n_pe - pat, 0);
            const unsigned long long busy = __ballot(prem > 0);
            if (busy == 0ull) break;
            int n_busy = 0;
#pragma unroll
            for (int q = 0; q < NG; ++q) n_busy += (int)((busy >> (q * GS)) & 1ull);
            const int share = PP / n_busy;
            const int take = min(prem, share);
            int my_rank = 0;  // this group's place among the busy ones
#pragma unroll
            for (int q = 0; q < NG; ++q) my_rank += (q < g) ? (int)((busy >> (q * GS)) & 1ull) : 0;
            const int slot0 = my_rank * share;
            if (sub == 0) {
                Q.poff = slot0;      // (first pool slot of the candidate in this pass ...)
                Q.n_pe = take;       // (... and how many it holds)
                Q.pent = entries + n_fe + pat;
            }
            adh_wave_sync();
            // (a) balanced: lane l takes slots l, l + 64, ... whichever candidate's they are
            for (int s_ = lane; s_ < n_busy * share; s_ += ADH_WAVE) {
                // the s_ / share-th busy candidate
                int q = 0, seen = -1;
#pragma unroll
                for (int c = 0; c < NG; ++c) {
                    const int on = (int)((busy >> (c * GS)) & 1ull);
                    seen += on;
                    if (on && seen == s_ / share) q = c;
                }
                const GroupTile<FM, SM, NO> &T = W.g[q];
                const int e = s_ - T.poff;
                if (e >= T.n_pe) continue;
                const ImEntry en = T.pent[e];
                const int tF = T.F;
                const int sf = (int)en.cell / I, i = (int)en.cell - sf * I, sc = sf / tF, f = sf - sc * tF;
                const double ds = (double)(sc - T.S), df = (double)(f - 1);
                const double w = exp(-0.1 * sqrt(ds * ds + df * df));
                const float t = en.x * T.iso_int[i];
#pragma unroll
                for (int o = 0; o < NO; ++o) P.t[o][s_] = o < T.O ? (double)t * qtf_at(T, i, o, sc) : 0.0;
                P.r[0][s_] = en.x > 0.0f ? (double)en.x * w : 0.0;
                P.r[1][s_] = en.x > 0.0f ? w : 0.0;
                P.r[2][s_] = en.y > 0.0f ? (double)en.y * w : 0.0;
                P.r[3][s_] = en.y > 0.0f ? w : 0.0;
                P.cell[s_] = (uint32_t)(sc << 16 | f << 4 | i);
                P.x[s_] = en.x;
            }
            pat += take;
            adh_wave_sync();
            if (stop == 6) continue;
            // (b) serial, adds only: the candidate's entries of this pass, four per step with their loads first
            const int lo = slot0, hi = slot0 + take;
            if (sub < 5 * I) {
                const int ri = max(role - 1, 0);
                for (int u0 = lo; __any(u0 < hi); u0 += 4) {
                    uint32_t cp[4];
                    float xs[4];
                    double rr[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int u = min(max(min(u0 + j, hi - 1), 0), PP - 1);
                        cp[j] = P.cell[u];
                        xs[j] = P.x[u];
                        rr[j] = P.r[ri][u];
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const bool mine = u0 + j < hi && (int)(cp[j] & 15u) == iso;
                        const int sc = (int)(cp[j] >> 16);
                        const bool fresh = mine && sc != cur_sc;  // role 0: per-scan sums, added up in scan order
                        tot += fresh ? part : 0.0f;
                        part = fresh ? 0.0f : part;
                        cur_sc = fresh ? sc : cur_sc;
                        part += mine ? xs[j] : 0.0f;
                        acc += mine ? rr[j] : 0.0;  // (roles 1 .. 4; the zero terms of cells without signal add nothing)
                    }
                }
            } else if (sub == GS - 1) {
                for (int u0 = lo; __any(u0 < hi); u0 += 4) {
                    uint32_t cp[4];
                    double tt[NO][4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int u = min(max(min(u0 + j, hi - 1), 0), PP - 1);
                        cp[j] = P.cell[u];
#pragma unroll
                        for (int o = 0; o < NO; ++o) tt[o][j] = P.t[o][u];
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (u0 + j < hi) {
                            const int cs = (int)((cp[j] >> 16) << 12 | ((cp[j] >> 4) & 0xFFFu));
                            if (cs != t_cell) {
                                if (t_cell >= 0) template_cell_done();
                                t_cell = cs;
#pragma unroll
                                for (int o = 0; o < NO; ++o) t_a[o] = 0.0;
                            }
#pragma unroll
                            for (int o = 0; o < NO; ++o) t_a[o] += tt[o][j];
                        }
                    }
                }
            }
            adh_wave_sync();
        }
    }
    float tsum[NO];
#pragma unroll
    for (int o = 0; o < NO; ++o) tsum[o] = 0.0f;
    if (sub == GS - 1) {
        if (t_cell >= 0) template_cell_done();
#pragma unroll
        for (int o = 0; o < NO; ++o) {
            if (t_row >= 0) {
                Q.tsp_raw[o][t_row] = t_srow[o];
                t_tsum[o] += t_srow[o];
            }
            Q.esc[o] = (t_isum[o] > 0) ? t_ssum[o] / t_isum[o] : 0.0;  // template centre of mass
            Q.efc[o] = (t_isum[o] > 0) ? t_fsum[o] / t_isum[o] : 0.0;
            tsum[o] = t_tsum[o];
        }
    }
#pragma unroll
    for (int o = 0; o < NO; ++o) res.tsum[o] = __shfl(tsum[o], gbase + GS - 1);
    if (NO == 1) res.tsum[1] = 0.0f;
    // isotope lane i < I: the sums of roles 1 .. 4 of its isotope
    const int il = sub < I ? sub : 0;
    const double vh = __shfl(acc, gbase + I + il), wh = __shfl(acc, gbase + 2 * I + il);
    const double vmz = __shfl(acc, gbase + 3 * I + il), wmz = __shfl(acc, gbase + 4 * I + il);
    res.spi = tot + part;
    res.hp = (wh > 0) ? vh / wh : 0.0;  // weights are exp(...) > 0: "any non-zero cell" == "w sum > 0"
    res.omzp = (wmz > 0) ? vmz / wmz : 0.0;
    if (stop == 2 || stop == 6) return res;
    // where the candidates' fragment entries start in the wavefront's list
    if (lane == 0) {
        int o = 0;
#pragma unroll
        for (int q = 0; q < NG; ++q) {
            W.g[q].off = o;
            o += W.g[q].n_fe;
        }
    }
    for (int c = sub; c < NO * FM * KMAX; c += GS) (&Q.ffp[0][0][0])[c] = 0.0f;
    adh_wave_sync();

    // ---- the fragment entries.  First every entry's plane (fragment * O + observation), balanced over the
    // wavefront's list: a plane is a range.  When the four candidates' entries fit the pool together (nearly always)
    // this pass is also pass (a) below: the list's order IS the pool's order then, and an entry is fetched once.
    const int total = W.g[NG - 1].off + W.g[NG - 1].n_fe;
    const bool single = total <= NP;
    // (more than one pass: the scan masks of ALL the candidates' scans are taken in the first pass - the quadrupole rows
    // they are computed from share their bytes with the scan profiles, which the first fold writes)
    if (!single && sub < NO) Q.scans[sub] = S >= 64 ? ~0ull : ((1ull << S) - 1ull);
    const int off1 = W.g[1].off, off2 = W.g[2].off, off3 = W.g[3].off;
    // plane of a cell (float estimate of cell / SF, one fix-up either way)
    auto plane_of = [](uint32_t c, int tSF, float isf) -> int {
        int k = (int)((float)c * isf);
        const int rem = (int)c - k * tSF;
        k += rem >= tSF ? 1 : (rem < 0 ? -1 : 0);
        return k;
    };
    // pool slot `at` <- entry `en` of plane `pc` of candidate T
    auto to_pool = [&](GroupTile<FM, SM, NO> &T, const ImEntry &en, int pc, int at) {
        const int tF = T.F;
        const int o = NO == 1 ? 0 : pc - (pc / T.O) * T.O;
        const int rem_c = (int)en.cell - pc * T.SF;
        int sc = (int)((double)rem_c * T.inv_f);  // exact quotient: float64 estimate, one fix-up
        if (rem_c - sc * tF >= tF) ++sc;
        const int f = rem_c - sc * tF;
        const double ds = (double)sc - T.esc[o], df = (double)f - T.efc[o];
        W.u.pool.w[at] = exp(-0.1 * sqrt(ds * ds + df * df));
        W.u.pool.x[at] = en.x;
        W.u.pool.y[at] = en.y;
        W.u.pool.scf[at] = (uint16_t)(o << 15 | sc << 8 | f);
        atomicOr(&T.scans[o], 1ull << sc);
    };
    for (int j = lane; j < total; j += ADH_WAVE) {
        const int q = (j >= off1) + (j >= off2) + (j >= off3);
        GroupTile<FM, SM, NO> &T = W.g[q];
        const int e = j - T.off;
        const ImEntry *E = T.entries;
        ImEntry en;
        en.cell = E[e].cell, en.x = 0.0f, en.y = 0.0f;
        if (single) en = E[e];
        const uint32_t pcell = e > 0 ? E[e - 1].cell : 0u;
        const int pc = plane_of(en.cell, T.SF, T.inv_sf), pp = e > 0 ? plane_of(pcell, T.SF, T.inv_sf) : -1;
        if (pc != pp) {
            T.pl_beg[pc] = e;
            if (pp >= 0) T.pl_end[pp] = e;
        }
        if (e == T.n_fe - 1) T.pl_end[pc] = T.n_fe;
        if (single) {
            to_pool(T, en, pc, j);
            W.owner[j] = (uint8_t)(q * GS + (NO == 1 ? pc : pc / T.O));
        }
    }
    adh_wave_sync();
    // Then passes over the fragments SIDE BY SIDE: every fragment lane that has entries left puts its next R of them into
    // the pool (R = pool size / lanes still busy), so the serial folds below advance all planes of all four candidates
    // at once - a contiguous piece of the list would hold three planes of one signal-rich candidate and the other lanes
    // would wait - and a pass is: (a) balanced over the pool's slots: decode, weight around the template centre,
    // scans that occur; (b) balanced: the scan mask of the transfer function (candidate.py:287-290) for the scans
    // that are new, applied to the intensities; (c) serial per fragment lane (its O planes are neighbours), adds only.
    const bool fl = alive && sub < K0;
    double vi = 0.0, wi = 0.0, vm = 0.0, wm = 0.0;
    float fs = 0.0f;
    int cur = -1;      // scan of the running scan sum
    int cur_o = 0;     // observation of the plane the lane is in
#pragma unroll
    for (int o = 0; o < 2; ++o) res.ohe[o] = res.omz[o] = 0.0;
    int at = 0x7FFFFFFF, at_end = 0;  // entries of the lane's planes
    if (fl) {
#pragma unroll
        for (int o = 0; o < NO; ++o) {
            if (o < O) {
                at = min(at, Q.pl_beg[sub * O + o]);
                at_end = max(at_end, Q.pl_end[sub * O + o]);
            }
        }
    }
    if (at == 0x7FFFFFFF) at = 0, at_end = 0;
    bool fsp_ready = false;
    auto plane_done = [&]() {  // the lane leaves plane (sub, cur_o)
        if (cur >= 0) Q.v.fsp[cur_o][cur][sub] = fs;
        if (NO == 1) {
            res.ohe[0] = (wi > 0) ? vi / wi : 0.0;  // weights are exp(...) > 0: "any non-zero cell" == "wi > 0"
            res.omz[0] = (wm > 0) ? vm / wm : 0.0;
        } else {
            const double a = (wi > 0) ? vi / wi : 0.0, b = (wm > 0) ? vm / wm : 0.0;
            res.ohe[0] = cur_o == 0 ? a : res.ohe[0];
            res.ohe[1] = cur_o == 1 ? a : res.ohe[1];
            res.omz[0] = cur_o == 0 ? b : res.omz[0];
            res.omz[1] = cur_o == 1 ? b : res.omz[1];
        }
        vi = wi = vm = wm = 0.0;
        fs = 0.0f;
        cur = -1;
    };
    for (;;) {
        const int rem = max(at_end - at, 0);
        const int n_busy = __popcll(__ballot(rem > 0));
        if (n_busy == 0) break;
        int take = rem, base = Q.off + at, n_slots = total;  // (one pass: the pool was filled above)
        if (!single) {
            const int R = min(NP / n_busy, 64);
            take = min(rem, R);
            int incl = take;  // slots [base, base + take) of the pool are this lane's
#pragma unroll
            for (int o = 1; o < ADH_WAVE; o <<= 1) {
                const int u = __shfl_up(incl, o);
                if (lane >= o) incl += u;
            }
            base = incl - take;
            n_slots = __shfl(incl, ADH_WAVE - 1);
            for (int rr = 0; rr < take; ++rr) W.owner[base + rr] = (uint8_t)lane;
            W.seg_base[lane] = (uint16_t)base;
            W.seg_src[lane] = at;
            adh_wave_sync();
            // (a)
            for (int s_ = lane; s_ < n_slots; s_ += ADH_WAVE) {
                const int o = (int)W.owner[s_];
                GroupTile<FM, SM, NO> &T = W.g[o / GS];
                const int e = W.seg_src[o] + (s_ - (int)W.seg_base[o]);
                const ImEntry en = T.entries[e];
                to_pool(T, en, plane_of(en.cell, T.SF, T.inv_sf), s_);
            }
            adh_wave_sync();
        }
        if (stop == 3) {
            at += take;
            continue;
        }
        // (b)
        {
            int n_need = 0;
            const unsigned long long lt = (1ull << lane) - 1ull;
            constexpr int NSLOT = NG * NO * SM;
#pragma unroll
            for (int k = 0; k < (NSLOT + ADH_WAVE - 1) / ADH_WAVE; ++k) {
                const int slot = k * ADH_WAVE + lane;
                const int qo = min(slot / SM, NG * NO - 1), sc = slot - qo * SM;
                const int q = qo / NO, o = qo - q * NO;
                const bool due = slot < NSLOT && (((W.g[q].scans[o] & ~W.g[q].scans_done[o]) >> sc) & 1ull);
                const unsigned long long m = __ballot(due);
                if (due) W.need[n_need + __popcll(m & lt)] = (uint16_t)slot;
                n_need += __popcll(m);
            }
            adh_wave_sync();
            for (int t = lane; t < n_need; t += ADH_WAVE) {
                const int slot = (int)W.need[t];
                const int qo = slot / SM, sc = slot - qo * SM;
                const int q = qo / NO, o = qo - q * NO;
                GroupTile<FM, SM, NO> &T = W.g[q];
                double sum = 0;
                for (int i = 0; i < I; ++i) sum += qtf_at(T, i, o, sc);
                T.qmask[o][sc] = (float)(sum / (double)I);
            }
            adh_wave_sync();
            if (sub < NO) Q.scans_done[sub] = Q.scans[sub];
            if (!fsp_ready) {  // (the quadrupole rows are dead: their bytes become the scan profiles)
                for (int c = sub; c < NO * SM * KMAX; c += GS) (&Q.v.fsp[0][0][0])[c] = 0.0f;
                fsp_ready = true;
            }
            for (int s_ = lane; s_ < n_slots; s_ += ADH_WAVE) {  // candidate.py:290
                const int scf = (int)W.u.pool.scf[s_];
                W.u.pool.x[s_] = W.u.pool.x[s_] * W.g[(int)W.owner[s_] / GS].qmask[scf >> 15][(scf >> 8) & 0x7F];
            }
            adh_wave_sync();
        }
        if (stop == 4) {
            at += take;
            continue;
        }
        // (c) four entries per step, their loads first (the walk is a chain of LDS latencies otherwise); the frame
        // profile takes its terms through ds_add_f32 (no answer to wait for; a lane's adds to one address arrive in
        // program order)
        for (int e = 0; __any(e < take); e += 4) {
            double w4[4];
            float v4[4], y4[4];
            int s4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int slot = min(base + min(e + j, max(take - 1, 0)), NP - 1);
                w4[j] = W.u.pool.w[slot];
                v4[j] = W.u.pool.x[slot];
                y4[j] = W.u.pool.y[slot];
                s4[j] = (int)W.u.pool.scf[slot];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (e + j < take) {
                    const double w = w4[j];
                    const float v = v4[j], y = y4[j];
                    const int o = NO == 1 ? 0 : s4[j] >> 15, sc = (s4[j] >> 8) & 0x7F, f = s4[j] & 0xFF;
                    if (NO > 1 && o != cur_o) {  // (the lane's planes follow each other: observation 0, then 1)
                        plane_done();
                        cur_o = o;
                    }
                    if (sc != cur) {  // the cells of a scan are consecutive: its sum is complete
                        if (cur >= 0) Q.v.fsp[cur_o][cur][sub] = fs;
                        fs = 0.0f;
                        cur = sc;
                    }
                    fs += v;
                    (void)__hip_atomic_fetch_add(&Q.ffp[cur_o][f][sub], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    const double tm = (double)y * w;  // (w > 0: the product is > 0 exactly when the m/z channel is)
                    // (a cell without signal enters as 0 - adding + 0.0 leaves a sum as it is - and its weight through
                    // an indicator: w x 1.0 + wi rounds once, like wi + w; one select on the indicator's high word
                    // instead of two on the weight, as in the register kernels)
                    vi += (double)__builtin_fmaxf(v, 0.0f) * w;
                    wi = __builtin_fma(w, __hiloint2double(v > 0.0f ? 0x3ff00000 : 0, 0), wi);
                    vm += __builtin_fmax(tm, 0.0);
                    wm = __builtin_fma(w, __hiloint2double(tm > 0.0 ? 0x3ff00000 : 0, 0), wm);
                }
            }
        }
        at += take;
        adh_wave_sync();
    }
    if (!fsp_ready) {  // (no fragment entry in any of the four candidates, or an ablation stop)
        for (int c = sub; c < NO * SM * KMAX; c += GS) (&Q.v.fsp[0][0][0])[c] = 0.0f;
        adh_wave_sync();
    }
    if (fl) plane_done();
    adh_wave_sync();
    res.alive = alive && !(stop >= 3 && stop <= 5);
    res.K0 = K0;
    res.F = F;
    res.S = S;
    res.I = I;
    res.O = O;
    return res;
}

}  // namespace featim4

// ---- the order of work.  A wavefront's serial walks take as long as the longest list of its four candidates, and
// the lists are very unequal (a candidate on a planted peptide holds a few hundred entries, one on noise a handful):
// in plan order three of four wavefronts hold a long list and all wait for it.  Two small kernels bucket the
// candidates of a launch by the length of their lists (log scale, longest first: long wavefronts start first) and
// leave aside what adh_feature_im_tile4_kernel does not take - candidates that failed in the gather (nothing to do)
// and materialised tiles (-> `dense`, for the first blocks of the grid: the one-candidate body).
// side[0] materialised candidates, side[1] ordered candidates, side[2 .. 2 + NB) bucket sizes, side[2 + NB .. 2 + 2 NB)
// bucket fill, then order[n] and dense[n]
namespace featim4 {
constexpr int NB = 32;
constexpr int SIDE_HEAD = 2 + 2 * NB;
__host__ __device__ inline uint64_t side_bytes(int64_t n) { return ((uint64_t)(SIDE_HEAD + 2 * n) * 4 + 255) / 256 * 256; }
// -2: nothing to do, -1: materialised tiles, else the bucket
__device__ __forceinline__ int order_class(const CandRecIM &r, const unsigned char *__restrict__ scratch) {
    if (r.flags & ADH_FLAG_SKIP) return -2;
    const uint32_t *h = reinterpret_cast<const uint32_t *>(scratch + r.scratch_off);
    const uint4 h4 = *reinterpret_cast<const uint4 *>(h);
    if (h4.x == 0u) return -2;
    if (h4.w != ADH_IM_MODE_COMPACT) return -1;
    const float key = (float)(h[4] + (h4.z >> 2));  // precursor entries + a quarter of the fragment entries: ~ the longest walk
    return min(NB - 1, (int)(__log2f(key + 1.0f) * 3.0f));
}
}  // namespace featim4

__global__ __launch_bounds__(256) void adh_im_order_hist_kernel(const CandRecIM *__restrict__ plan, int32_t n,
                                                                const unsigned char *__restrict__ scratch,
                                                                uint32_t *__restrict__ side) {
    using namespace featim4;
    __shared__ uint32_t h[NB];
    if (threadIdx.x < NB) h[threadIdx.x] = 0u;
    __syncthreads();
    const int ci = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    const int c = ci < n ? order_class(plan[ci], scratch) : -2;
    if (c >= 0) atomicAdd(&h[c], 1u);
    __syncthreads();
    if (threadIdx.x < NB && h[threadIdx.x]) atomicAdd(&side[2 + threadIdx.x], h[threadIdx.x]);
}

__global__ __launch_bounds__(256) void adh_im_order_scatter_kernel(const CandRecIM *__restrict__ plan, int32_t n,
                                                                   const unsigned char *__restrict__ scratch,
                                                                   uint32_t *__restrict__ side) {
    using namespace featim4;
    __shared__ uint32_t cnt[NB], base[NB];
    if (threadIdx.x < NB) cnt[threadIdx.x] = 0u;
    __syncthreads();
    const int ci = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    const int c = ci < n ? order_class(plan[ci], scratch) : -2;
    uint32_t rank = 0u;
    if (c >= 0) rank = atomicAdd(&cnt[c], 1u);
    __syncthreads();
    if (threadIdx.x < NB) {
        const int b = (int)threadIdx.x;
        uint32_t start = 0u;  // buckets in descending order: the longest lists first
        for (int q = b + 1; q < NB; ++q) start += side[2 + q];
        base[b] = cnt[b] ? start + atomicAdd(&side[2 + NB + b], cnt[b]) : 0u;
        if (blockIdx.x == 0 && b == 0) {
            uint32_t tot = side[2];
            for (int q = 1; q < NB; ++q) tot += side[2 + q];
            side[1] = tot;
        }
    }
    __syncthreads();
    uint32_t *order = side + SIDE_HEAD, *dense = order + n;
    if (c >= 0) order[base[c] + rank] = (uint32_t)ci;
    else if (c == -1) dense[atomicAdd(&side[0], 1u)] = (uint32_t)ci;
}

// `side`: see above (side[1] candidates in `order`).  The first `list_blocks` blocks of the grid take the candidates
// with materialised tiles through the one-candidate body (adh_feature_im_body<LAY, true>, adh_features_im.hip) - they
// are the longest walks of a launch (a dense tile is 1 152 cells per plane) and start first, beside the others,
// instead of in a launch of their own behind this one; the LDS of a block is the larger of the two layouts.
template <int FM, int SM, int NO, class LAY>
__global__ __launch_bounds__(ADH_WAVE, 2) void adh_feature_im_tile4_kernel(
    DevTims run, const CandRecIM *__restrict__ plan, int32_t n_cand, const float *__restrict__ iso_table,
    int32_t n_iso_cols, adh_scoring_config_t cfg, const unsigned char *__restrict__ scratch, DevOut out,
    unsigned char *__restrict__ prof, const uint32_t *__restrict__ side, Caps caps, int32_t list_blocks, int32_t stop) {
    using namespace featim4;
    typedef ImProfRec<FM, SM, NO> Rec;
    extern __shared__ __align__(16) unsigned char smem[];
    if ((int32_t)blockIdx.x < list_blocks) {
        if (stop) return;  // (developer ablation: the four-candidate path alone)
#ifndef ADH_IM4_STANDALONE
        const uint32_t n = side[0];
        const uint32_t *list = side + SIDE_HEAD + n_cand;
        for (uint32_t j = blockIdx.x; j < n; j += (uint32_t)list_blocks) {
            adh_feature_im_body<LAY, true, true>((int)list[j], run, plan, iso_table, n_iso_cols, cfg, scratch, out, caps, prof);
            __syncthreads();  // (the next candidate reuses the LDS arrays)
        }
#endif
        return;
    }
    WaveTile<FM, SM, NO> &W = *reinterpret_cast<WaveTile<FM, SM, NO> *>(smem);
    const int32_t block = (int32_t)blockIdx.x - list_blocks;
    const int32_t n_order = (int32_t)side[1];
    if (block * NG >= n_order) return;
    int ci = 0;
    const TileOut t = tile4_phase<FM, SM, NO>(W, run, plan, n_order, block, iso_table, n_iso_cols, cfg, scratch, out,
                                              side + SIDE_HEAD, ci, stop);
    // ---- hand-over: the record adh_feature_im_profiles_kernel reads (frame axis centred: entry r <-> cycle r + shift);
    // observations beyond the candidate's own (a launch of up to two) are zero rows
    if (!t.alive) return;
    const int lane = threadIdx.x;
    const int g = lane / GS, sub = lane % GS;
    const GroupTile<FM, SM, NO> &Q = W.g[g];
    const int K0 = t.K0, F = t.F, S = t.S, I = t.I, O = t.O;
    Rec &rec = reinterpret_cast<Rec *>(prof)[ci];
    const int shift = F / 2 - FM / 2;
    for (int c = sub; c < NO * FM; c += GS) {
        const int o = c / FM, rr = c - o * FM, f = rr + shift;
        rec.tfp_raw[o][rr] = (o < O && f >= 0 && f < F) ? Q.tfp_raw[o][f] : 0.0f;
    }
    for (int c = sub; c < NO * SM; c += GS) {
        const int o = c / SM, sc = c - o * SM;
        rec.tsp_raw[o][sc] = (o < O && sc < S) ? Q.tsp_raw[o][sc] : 0.0f;
    }
    for (int c = sub; c < K0 * NO * FM; c += GS) {
        const int k = c / (NO * FM), rem = c - k * NO * FM, o = rem / FM, rr = rem - o * FM, f = rr + shift;
        rec.ffp[k][o][rr] = (o < O && f >= 0 && f < F) ? Q.ffp[o][f][k] : 0.0f;
    }
    for (int c = sub; c < K0 * NO * SM; c += GS) {
        const int k = c / (NO * SM), rem = c - k * NO * SM, o = rem / SM, sc = rem - o * SM;
        rec.fsp[k][o][sc] = (o < O && sc < S) ? Q.v.fsp[o][sc][k] : 0.0f;
    }
    if (sub < K0) {
#pragma unroll
        for (int o = 0; o < NO; ++o) {
            rec.ohe[sub][o] = o < O ? t.ohe[o] : 0.0;
            rec.omz[sub][o] = o < O ? t.omz[o] : 0.0;
        }
    }
    if (sub < 4) {
        const bool on = sub < I;
        rec.hp[sub] = on ? t.hp : 0.0;
        rec.omzp[sub] = on ? t.omzp : 0.0;
        rec.spi[sub] = on ? t.spi : 0.0f;
        rec.iso_int[sub] = on ? Q.iso_int[sub] : 0.0f;
        rec.iso_mz[sub] = on ? Q.iso_mz[sub] : 0.0f;
    }
    if (sub < NO) rec.tsum[sub] = sub < O ? (sub == 0 ? t.tsum[0] : t.tsum[1]) : 0.0f;
    if (sub == 0) rec.K0 = (uint32_t)K0;
}
