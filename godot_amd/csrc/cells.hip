// Per-cell weighted-Poisson likelihood kernels (gfx950).
//
// Layout: photon weights of all cells in one array `we`; cell c owns
// we[lo[c] .. hi[c]-1].  A "team" of T lanes (T = 4, 16 or 64, aligned inside
// a 64-lane wavefront) owns one cell.  The team streams the cell's weights
// from HBM exactly once into registers (R per lane) and runs every pass of
// the fixed-schedule Newton/Halley maximiser, the TS and the support search
// out of registers; per-pass sums are combined with wavefront butterflies.
// Cells that do not fit T*R registers re-read their weights per pass (they
// then come from L2, which a cell of that size still fits).  Divisions are a
// hardware reciprocal estimate plus two Newton steps (the operands are never
// near the ends of the exponent range), reductions use DPP lane permutes.

#include "cells_common.hpp"

namespace {

// One team per cell.  Cells whose first pass of the schedule asks for a
// restart (about one in seven of sparse cells) are not iterated in place,
// where they would hold up the other teams of their wavefront: the team posts
// (cell, new start) to a queue in LDS and the workgroup then serves the queue
// with densely packed teams that reload the cell's weights (L2-resident).
// B0: the background level is fixed at its nominal value for every cell
// (beta = 1, the common case): t2 = 1 is folded at compile time.
template <int T, int R, bool B0>
__device__ __forceinline__ void cell_fit_body(const CellArgs &p, const double *guess, double guess0,
                                              const double *beta, double beta0, double *alpha, double *ts,
                                              int32_t *flags)
{
    constexpr int NTEAM = 256 / T;
    __shared__ int q_count[2];
    __shared__ int q_cell[2][NTEAM];
    __shared__ double q_g[2][NTEAM], q_b[2][NTEAM];
    const int lane = threadIdx.x & (T - 1);
    const int team = threadIdx.x / T;
    const int64_t cell0 = (int64_t)blockIdx.x * NTEAM;
    if (threadIdx.x < 2) q_count[threadIdx.x] = 0;
    __syncthreads();

    // round 0 serves the workgroup's own cells, rounds 1..3 the restart queue
    // (the reference recurses at most three deep); one copy of the schedule.
    for (int depth = 0; depth <= 3; ++depth) {
        const int qr = (depth + 1) & 1, qw = depth & 1;
        int64_t cell = cell0 + team;
        bool active = cell < p.ncell;
        double g = 0.0, b = 0.0;
        if (depth > 0) {
            __syncthreads();
            const int n = q_count[qr];
            if (n == 0) break;
            __syncthreads();
            if (threadIdx.x == 0) q_count[qw] = 0;
            __syncthreads();
            active = team < n;
            if (active) {
                cell = cell0 + q_cell[qr][team];
                g = q_g[qr][team];
                b = q_b[qr][team];
            }
        } else if (active) {
            g = (guess ? guess[cell] : guess0) - 1;
            b = B0 ? 0.0 : (beta ? beta[cell] : beta0) - 1;
        }
        if (!active) continue;

        const int64_t lo = p.lo[cell], hi = p.hi[cell];
        const bool bad = (lo < 0) || (hi < lo) || (hi > p.nph);
        const double S = p.swap ? p.B[cell] : p.S[cell];
        CellData<T, R> cd;
        cd.load(p.we + (bad ? 0 : lo), bad ? 0 : hi - lo, lane, p.swap != 0);
        int fl = depth << 1;
        double a = qnan();
        bool at_zero = false, done = true;
        if (bad) {
            fl = GODOT_F_FAILED | GODOT_F_BAD_RANGE;
        } else if (fit_attempt_impl<CellData<T, R>, B0>(cd, S, g, b, a, at_zero)) {
            if (at_zero) {
                a = 0.0;
                fl |= GODOT_F_AT_ZERO;
            } else {
                a = a + 1;
            }
        } else if (depth > 2) {
            // bracketing fallback: out of line, on a streamed view of the cell
            StreamCell<T> sc;
            stream_load(sc, p.we + lo, hi - lo, lane, p.swap != 0);
            a = max_numerical(sc, S, fl);
        } else {
            done = false;
            if (lane == 0) {
                int slot = atomicAdd(&q_count[qw], 1);
                q_cell[qw][slot] = (int)(cell - cell0);
                q_g[qw][slot] = (a + 1) - 1;
                q_b[qw][slot] = (b + 1) - 1;
            }
        }
        if (done) {
            double t = qnan();
            if (ts && !(fl & GODOT_F_FAILED)) t = test_statistic(cd, S, a);
            if (lane == 0) {
                alpha[cell] = a;
                if (ts) ts[cell] = t;
                if (flags) flags[cell] = fit_flags(fl, a);
            }
        }
    }
}

template <int T, int R>
__global__ __launch_bounds__(256, 4) void cell_fit_b0_kernel(CellArgs p, const double *guess, double guess0,
                                                             const double *beta, double beta0, double *alpha,
                                                             double *ts, int32_t *flags)
{
    cell_fit_body<T, R, true>(p, guess, guess0, beta, beta0, alpha, ts, flags);
}

template <int T, int R>
__global__ __launch_bounds__(256, 4) void cell_fit_bx_kernel(CellArgs p, const double *guess, double guess0,
                                                             const double *beta, double beta0, double *alpha,
                                                             double *ts, int32_t *flags)
{
    cell_fit_body<T, R, false>(p, guess, guess0, beta, beta0, alpha, ts, flags);
}

template <int T, int R>
__global__ __launch_bounds__(256) void cell_loglike_kernel(CellArgs p, const double *alpha, double *out)
{
    int64_t cell, lo, n;
    int lane;
    double S, B;
    bool bad;
    if (!cell_setup<T>(p, cell, lane, lo, n, S, B, bad)) return;
    CellData<T, R> cd;
    cd.load(p.we + lo, n, lane, p.swap != 0);
    double v = bad ? qnan() : loglike(cd, S, alpha[cell]);
    if (lane == 0) out[cell] = v;
}

// The sums behind the reference's small per-cell functions (core.py:427-542:
// log_full_likelihood, fmin_tnc_func, f1/f2/f3, nr, halley, f1_profile,
// log_profile_likelihood_approx), for flux factors alpha, beta per cell, with
// t_i = 1 + (beta - 1)(1 - w_i) + (alpha - 1) w_i:
//   out[0] sum log t      out[1] sum w/t         out[2] sum (1-w)/t
//   out[3] sum (w/t)^2    out[4] sum (w/t)^3     out[5] sum ((1-w)/t)^2
//   out[6] sum (w/t)((1-w)/t)                    out[7] sum (w/t)((1-w)/t)^2
template <int T, int R>
__global__ __launch_bounds__(256) void cell_sums_kernel(CellArgs p, const double *alpha, double alpha0,
                                                        const double *beta, double beta0, double *out)
{
    int64_t cell, lo, n;
    int lane;
    double S, B;
    bool bad;
    p.S = alpha ? alpha : out;   // S, B are not used here; keeps cell_setup's loads in bounds
    p.B = nullptr;
    if (!cell_setup<T>(p, cell, lane, lo, n, S, B, bad)) return;
    CellData<T, R> cd;
    cd.load(p.we + lo, n, lane, p.swap != 0);
    const double a = (alpha ? alpha[cell] : alpha0) - 1.0;
    const double b = (beta ? beta[cell] : beta0) - 1.0;
    const double ab = a - b, b1 = 1.0 + b;
    double acc[7];
    cd.template sums<7>([=](double w, double (&s)[7]) {
        const double t = fma(w, ab, b1);
        const double x = w / t, y = (1.0 - w) / t;
        s[0] += x;
        s[1] += y;
        s[2] = fma(x, x, s[2]);
        s[3] = fma(x * x, x, s[3]);
        s[4] = fma(y, y, s[4]);
        s[5] = fma(x, y, s[5]);
        s[6] = fma(x * y, y, s[6]);
    }, acc);
    // a negative t makes the reference's sum of logarithms NaN; two of them would hide
    // each other in the product, so they are counted
    double ls, nneg;
    cd.logsum([=](double w, double &t, double &g) {
        t = fma(w, ab, b1);
        g = t < 0.0 ? 1.0 : 0.0;
    }, ls, nneg);
    if (nneg > 0.0) ls = qnan();
    if (lane == 0) {
        double *o = out + cell * 8;
        o[0] = bad ? qnan() : ls;
#pragma unroll
        for (int k = 0; k < 7; ++k) o[1 + k] = bad ? qnan() : acc[k];
    }
}

template <int T, int R>
__global__ __launch_bounds__(256) void cell_moments_kernel(CellArgs p, double *wsum, double *w2sum)
{
    int64_t cell, lo, n;
    int lane;
    double S, B;
    bool bad;
    p.S = wsum;   // unused by this kernel; keeps cell_setup's loads in bounds
    p.B = nullptr;
    if (!cell_setup<T>(p, cell, lane, lo, n, S, B, bad)) return;
    CellData<T, R> cd;
    cd.load(p.we + lo, n, lane, false);
    double acc[2];
    cd.template sums<2>([](double w, double (&s)[2]) {
        s[0] += w;
        s[1] = fma(w, w, s[1]);
    }, acc);
    if (lane == 0) {
        wsum[cell] = bad ? qnan() : acc[0];
        w2sum[cell] = bad ? qnan() : acc[1];
    }
}

// The series power_spectrum_fft transforms, from the bins of a time series (core.py:2178-2183,
// 241-244, 279-285, 2654-2661): weight moments W, WW of every bin, emptied where the bin has no
// exposure (as are its source and background exposures, in place), and the background moments
// Wb = counts - W, WbWb = counts - 2 W + WW, each difference rounded as numpy rounds it.
template <int T, int R>
__global__ __launch_bounds__(256) void cell_series_kernel(CellArgs p, const double *exp, double *sexp, double *bexp,
                                                          double *W, double *WW, double *Wb, double *WbWb)
{
    int64_t cell, lo, n;
    int lane;
    double S, B;
    bool bad;
    p.S = exp;    // unused by cell_setup's callers here; keeps its loads in bounds
    p.B = nullptr;
    if (!cell_setup<T>(p, cell, lane, lo, n, S, B, bad)) return;
    CellData<T, R> cd;
    cd.load(p.we + lo, n, lane, false);
    double acc[2];
    cd.template sums<2>([](double w, double (&s)[2]) {
        s[0] += w;
        s[1] = fma(w, w, s[1]);
    }, acc);
    if (lane == 0) {
        double w1 = bad ? qnan() : acc[0], w2 = bad ? qnan() : acc[1], cnt = (double)n;
        if (!(exp[cell] > 0)) {          // minimum_exposure = 0: NaN exposure empties the bin too
            w1 = w2 = cnt = 0.0;
            sexp[cell] = 0.0;
            bexp[cell] = 0.0;
        }
        W[cell] = w1;
        WW[cell] = w2;
        Wb[cell] = __dsub_rn(cnt, w1);
        WbWb[cell] = __dadd_rn(__dsub_rn(cnt, __dmul_rn(2.0, w1)), w2);
    }
}

// The two moments of a cell by ONE thread, for lists whose cells hold a few photons at most
// (the 300-s bins of a time series: one photon a bin): four lanes a cell spend three quarters
// of a wavefront on bins that hold nothing.  The additions are those of a team of four lanes
// to the bit -- photon j goes to partial sum j mod 4, the partial sums are added as
// (0 + 1) + (2 + 3) -- so that the series of one source do not depend on which kernel formed them.
struct SparseMoments {
    double w1, w2;
    int64_t n;
    bool bad;
};

__device__ __forceinline__ SparseMoments sparse_moments(const CellArgs &p, int64_t cell)
{
    const int64_t lo = p.lo[cell], hi = p.hi[cell];
    SparseMoments m;
    m.bad = (lo < 0) || (hi < lo) || (hi > p.nph);
    m.n = m.bad ? 0 : hi - lo;
    const double *w = p.we + (m.bad ? 0 : lo);
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0, b0 = 0.0, b1 = 0.0, b2 = 0.0, b3 = 0.0;
    int64_t j = 0;
    for (; j + 4 <= m.n; j += 4) {
        const double x0 = w[j], x1 = w[j + 1], x2 = w[j + 2], x3 = w[j + 3];
        a0 += x0;
        b0 = fma(x0, x0, b0);
        a1 += x1;
        b1 = fma(x1, x1, b1);
        a2 += x2;
        b2 = fma(x2, x2, b2);
        a3 += x3;
        b3 = fma(x3, x3, b3);
    }
    if (j < m.n) {
        const double x = w[j];
        a0 += x;
        b0 = fma(x, x, b0);
    }
    if (j + 1 < m.n) {
        const double x = w[j + 1];
        a1 += x;
        b1 = fma(x, x, b1);
    }
    if (j + 2 < m.n) {
        const double x = w[j + 2];
        a2 += x;
        b2 = fma(x, x, b2);
    }
    m.w1 = (a0 + a1) + (a2 + a3);
    m.w2 = (b0 + b1) + (b2 + b3);
    return m;
}

__global__ __launch_bounds__(256) void cell_moments_sparse_kernel(CellArgs p, double *wsum, double *w2sum)
{
    const int64_t cell = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (cell >= p.ncell) return;
    const SparseMoments m = sparse_moments(p, cell);
    wsum[cell] = m.bad ? qnan() : m.w1;
    w2sum[cell] = m.bad ? qnan() : m.w2;
}

__global__ __launch_bounds__(256) void cell_series_sparse_kernel(CellArgs p, const double *exp, double *sexp,
                                                                 double *bexp, double *W, double *WW, double *Wb,
                                                                 double *WbWb)
{
    const int64_t cell = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (cell >= p.ncell) return;
    const SparseMoments m = sparse_moments(p, cell);
    double w1 = m.bad ? qnan() : m.w1, w2 = m.bad ? qnan() : m.w2, cnt = (double)m.n;
    if (!(exp[cell] > 0)) {          // minimum_exposure = 0: NaN exposure empties the bin too
        w1 = w2 = cnt = 0.0;
        sexp[cell] = 0.0;
        bexp[cell] = 0.0;
    }
    W[cell] = w1;
    WW[cell] = w2;
    Wb[cell] = __dsub_rn(cnt, w1);
    WbWb[cell] = __dadd_rn(__dsub_rn(cnt, __dmul_rn(2.0, w1)), w2);
}

template <int T, int R>
__global__ __launch_bounds__(256) void cell_beta_kernel(CellArgs p, const double *alpha, double alpha0,
                                                        const double *guess, double guess0,
                                                        double *beta, double *lprof, int32_t *flags)
{
    int64_t cell, lo, n;
    int lane;
    double S, B;
    bool bad;
    if (!cell_setup<T>(p, cell, lane, lo, n, S, B, bad)) return;
    CellData<T, R> cd;
    cd.load(p.we + lo, n, lane, p.swap != 0);
    int fl = 0;
    double bt, lp = 0.0;
    if (bad) {
        fl = GODOT_F_FAILED;
        bt = lp = qnan();
    } else {
        double al = alpha ? alpha[cell] : alpha0;
        bt = fit_beta(cd, B, al, guess ? guess[cell] : guess0, fl);
        if (lprof) {
            double b = bt - 1, a = al - 1;
            double ls, unused;
            cd.logsum([=](double w, double &t, double &g) {
                t = fma(w, a - b, 1.0 + b);
                g = 0.0;
            }, ls, unused);
            lp = ls - S * (1 + a) - B * (1 + b);
        }
    }
    if (lane == 0) {
        beta[cell] = bt;
        if (lprof) lprof[cell] = lp;
        if (flags) flags[cell] = fl;
    }
}

// ---------------------------------------------------------------------------
// get_logpdf (core.py:851-932): adaptive support + npt-point profile.
// One workgroup = LP_CH cells.  Phase 1 (support search) runs one 16-lane
// team per cell; phase 2 spreads the LP_CH*npt grid points over all 256
// threads, each thread summing over its cell's photons; phase 3 checks the
// grid maximum against llmax and re-centres the cell if they disagree.
// ---------------------------------------------------------------------------
constexpr int LP_CH = 16;
constexpr int LP_T = 16;
constexpr int LP_R = 4;
constexpr int LP_PPL = 4;    // grid points per lane in the evaluation phase
constexpr int LP_MAX_RECENTRE = 50;

// wts = p.we once more, as a parameter that promises not to alias the outputs:
// that is what lets the compiler read weights at wave-uniform addresses with
// scalar loads.
__global__ __launch_bounds__(256) void cell_logpdf_kernel(CellArgs p, const double *__restrict__ wts,
                                                          const double *__restrict__ aopt_in, double dlogl,
                                                          int npt, int include_zero, double *__restrict__ dom,
                                                          double *__restrict__ cod, int32_t *__restrict__ flags)
{
    extern __shared__ double s_cod[];   // [LP_CH][npt]
    __shared__ double s_amin[LP_CH], s_amax[LP_CH], s_step[LP_CH], s_llmax[LP_CH], s_S[LP_CH],
        s_aopt[LP_CH];
    __shared__ int64_t s_lo[LP_CH], s_n[LP_CH];
    __shared__ int s_state[LP_CH];   // 0 = to do, 1 = done, 2 = absent/failed
    __shared__ int s_flags[LP_CH];

    const int tid = threadIdx.x;
    const int team = tid / LP_T;
    const int lane = tid % LP_T;
    const int64_t cell0 = (int64_t)blockIdx.x * LP_CH;
    const int64_t cell = cell0 + team;
    const bool have = cell < p.ncell;
    const bool swap = p.swap != 0;

    CellData<LP_T, LP_R> cd;
    double S = 0.0;
    int fl = 0;
    bool bad = false;
    if (have) {
        int64_t lo = p.lo[cell], hi = p.hi[cell];
        bad = (lo < 0) || (hi < lo) || (hi > p.nph);
        int64_t n = bad ? 0 : hi - lo;
        S = swap ? p.B[cell] : p.S[cell];
        cd.load(p.we + lo, n, lane, swap);
        double a0;
        if (bad) {
            a0 = qnan();
            fl = GODOT_F_FAILED | GODOT_F_BAD_RANGE;
        } else if (aopt_in) {
            a0 = aopt_in[cell];
        } else {
            // the bracketing fallback on a streamed view of the cell: handed the register
            // copy by reference, it would make every team keep a second copy in scratch memory
            bool numerical = false;
            a0 = fit_alpha(cd, S, 1.0, 1.0, fl, &numerical);
            if (numerical) {
                StreamCell<LP_T> sc;
                stream_load(sc, p.we + lo, n, lane, swap);
                a0 = max_numerical(sc, S, fl);
            }
            if (fl & GODOT_F_FAILED) bad = true;
        }
        if (lane == 0) {
            s_lo[team] = lo;
            s_n[team] = n;
            s_S[team] = S;
            s_aopt[team] = a0;
            s_state[team] = bad ? 2 : 0;
            s_flags[team] = fl;
        }
    } else {
        cd.load(p.we, 0, lane, swap);
        if (lane == 0) {
            s_state[team] = 2;
            s_n[team] = 0;
            s_lo[team] = 0;
            s_flags[team] = 0;
        }
    }
    __syncthreads();

    for (int round = 0; round <= LP_MAX_RECENTRE; ++round) {
        // phase 1: support of the cells still to do
        if (s_state[team] == 0) {
            Support sp = support_search(cd, S, s_aopt[team], dlogl, include_zero != 0);
            if (lane == 0) {
                s_amin[team] = sp.amin;
                s_amax[team] = sp.amax;
                s_llmax[team] = sp.llmax;
                s_step[team] = (npt > 1) ? (sp.amax - sp.amin) / (double)(npt - 1) : 0.0;
            }
        }
        __syncthreads();
        // phase 2: grid evaluation.  A wavefront takes one cell at a time, so the
        // cell's weights are wave-uniform (scalar loads, no vector-memory traffic)
        // and a lane carries LP_PPL grid points as independent product chains.
        // alpha w + (1 - w) is formed as 1 + (alpha - 1) w; on complemented
        // weights that is alpha - (alpha - 1) w.
        {
            const int wv = tid >> 6, ln = tid & 63;
            for (int c = wv; c < LP_CH; c += 4) {
                if (s_state[c] != 0) continue;
                const double amin_c = s_amin[c], amax_c = s_amax[c], step_c = s_step[c], S_c = s_S[c];
                const int64_t lo_c = __builtin_amdgcn_readfirstlane((int)s_lo[c]) |
                                     ((int64_t)__builtin_amdgcn_readfirstlane((int)(s_lo[c] >> 32)) << 32);
                const int n_c = __builtin_amdgcn_readfirstlane((int)s_n[c]);
                const double *w = wts + lo_c;
                for (int i0 = 0; i0 < npt; i0 += 64 * LP_PPL) {
                    double av[LP_PPL], am1[LP_PPL], pr[LP_PPL];
                    int ex[LP_PPL];
#pragma unroll
                    for (int u = 0; u < LP_PPL; ++u) {
                        int i = i0 + u * 64 + ln;
                        i = i < npt ? i : npt - 1;
                        av[u] = grid_point(amin_c, amax_c, step_c, i, npt);
                        am1[u] = av[u] - 1.0;
                        pr[u] = 1.0;
                        ex[u] = 0;
                    }
                    if (!swap) {
                        grid_products<LP_PPL>(w, n_c, p.nph - lo_c, pr, ex,
                                              [&](double wj, int u) { return fma(am1[u], wj, 1.0); });
                    } else {
                        grid_products<LP_PPL>(w, n_c, p.nph - lo_c, pr, ex,
                                              [&](double wj, int u) { return fma(-am1[u], wj, av[u]); });
                    }
#pragma unroll
                    for (int u = 0; u < LP_PPL; ++u) {
                        const int i = i0 + u * 64 + ln;
                        if (i < npt) s_cod[c * npt + i] = log_scaled(pr[u], ex[u]) - S_c * av[u];
                    }
                }
            }
        }
        __syncthreads();
        // phase 3: agreement of the grid maximum with llmax (core.py:923-929)
        int pending = 0;
        if (s_state[team] == 0) {
            double best = -INFINITY;
            int ibest = npt;
            bool anynan = false;
            for (int i = lane; i < npt; i += LP_T) {
                double v = s_cod[team * npt + i];
                if (isnan(v)) anynan = true;
                if (v > best) {   // strided scan keeps the first maximum per lane
                    best = v;
                    ibest = i;
                }
            }
#pragma unroll
            for (int o = LP_T / 2; o > 0; o >>= 1) {
                double ob = __shfl_xor(best, o, WAVE);
                int oi = __shfl_xor(ibest, o, WAVE);
                int on = __shfl_xor((int)anynan, o, WAVE);
                anynan = anynan || on;
                if (ob > best || (ob == best && oi < ibest)) {
                    best = ob;
                    ibest = oi;
                }
            }
            // numpy's argmax returns the first NaN if there is one, and the
            // reference's comparison is then false: the grid is accepted.
            if (ibest >= npt) ibest = 0;
            bool agree = anynan || !(fabs(best - s_llmax[team]) > 0.05);
            if (agree) {
                if (lane == 0) s_state[team] = 1;
            } else if (round == LP_MAX_RECENTRE) {
                if (lane == 0) {
                    s_state[team] = 2;
                    s_flags[team] |= GODOT_F_FAILED;
                }
            } else {
                if (lane == 0) {
                    s_aopt[team] = grid_point(s_amin[team], s_amax[team], s_step[team], ibest, npt);
                    s_flags[team] |= GODOT_F_RECENTRED;
                }
                pending = 1;
            }
        }
        if (!__syncthreads_or(pending)) break;
    }

    // write-out, coalesced over (cell, point)
    for (int k = tid; k < LP_CH * npt; k += blockDim.x) {
        int c = k / npt, i = k - c * npt;
        int64_t cg = cell0 + c;
        if (cg >= p.ncell) break;
        double d, v;
        if (s_state[c] == 1) {
            d = grid_point(s_amin[c], s_amax[c], s_step[c], i, npt);
            v = s_cod[k] - s_llmax[c];
        } else {
            d = v = qnan();
        }
        dom[cg * npt + i] = d;
        cod[cg * npt + i] = v;
    }
    if (flags && have && lane == 0) flags[cell] = s_flags[team];
}

// ---------------------------------------------------------------------------
// Quantiles of the cached profiles (core.py:1229, 1247-1256); one wavefront
// per cell, profile staged in LDS.
// ---------------------------------------------------------------------------
constexpr int PQ_MAX_NPT = 1024;

__global__ __launch_bounds__(256) void profile_quantiles_kernel(const double *dom, const double *cod,
                                                                int64_t ncell, int npt, double c_lo,
                                                                double c_hi, double *aguess,
                                                                double *xlo, double *xhi)
{
    extern __shared__ double sm[];   // per wave: dom[npt], pdf[npt], cdf[npt]
    const int wave = threadIdx.x / WAVE, lane = threadIdx.x % WAVE;
    const int64_t cell = (int64_t)blockIdx.x * (blockDim.x / WAVE) + wave;
    const bool active = cell < ncell;
    double *s_dom = sm + (size_t)wave * 3 * npt;
    double *s_pdf = s_dom + npt;
    double *s_cdf = s_pdf + npt;

    // argmax of cod (first maximum) and pdf = exp(cod)
    double best = -INFINITY;
    int ibest = npt;
    if (active) {
        const double *d = dom + cell * npt;
        const double *c = cod + cell * npt;
        for (int i = lane; i < npt; i += WAVE) {
            double v = c[i];
            s_dom[i] = d[i];
            s_pdf[i] = exp(v);
            if (v > best) {
                best = v;
                ibest = i;
            }
        }
    }
#pragma unroll
    for (int o = WAVE / 2; o > 0; o >>= 1) {
        double ob = __shfl_xor(best, o, WAVE);
        int oi = __shfl_xor(ibest, o, WAVE);
        if (ob > best || (ob == best && oi < ibest)) {
            best = ob;
            ibest = oi;
        }
    }
    // the slice belongs to this wavefront alone and LDS serves a wavefront's accesses in
    // order: only the compiler has to be kept from reordering
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // cumulative trapezoid: cdf[0] = 0, cdf[i+1] = cdf[i] + (x[i+1]-x[i]) (y[i+1]+y[i]) / 2
    double carry = 0.0;
    const int nseg = npt - 1;
    if (active) {
        if (lane == 0) s_cdf[0] = 0.0;
        for (int base = 0; base < nseg; base += WAVE) {
            int i = base + lane;
            double seg = 0.0;
            if (i < nseg) seg = (s_dom[i + 1] - s_dom[i]) * (s_pdf[i + 1] + s_pdf[i]) / 2.0;
#pragma unroll
            for (int o = 1; o < WAVE; o <<= 1) {
                double u = __shfl_up(seg, o, WAVE);
                if (lane >= o) seg += u;
            }
            double incl = seg + carry;
            carry = __shfl(incl, WAVE - 1, WAVE);
            if (i < nseg) s_cdf[i + 1] = incl;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (!active) return;
    // normalise, locate the quantiles: searchsorted(cdf, conf) (side='left')
    const double inv = 1. / carry;
    const double conf[2] = {c_lo, c_hi};
    double res[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        int idx = npt;
        for (int i = lane; i < npt; i += WAVE) {
            if (s_cdf[i] * inv >= conf[q]) {
                idx = i;
                break;
            }
        }
#pragma unroll
        for (int o = WAVE / 2; o > 0; o >>= 1) {
            int oi = __shfl_xor(idx, o, WAVE);
            idx = oi < idx ? oi : idx;
        }
        if (idx <= 0 || idx >= npt) {
            res[q] = qnan();
        } else {
            double chi = s_cdf[idx] * inv, clo = s_cdf[idx - 1] * inv;
            double m = (chi - clo) / (s_dom[idx] - s_dom[idx - 1]);
            res[q] = s_dom[idx - 1] + (conf[q] - clo) / m;
        }
    }
    if (lane == 0) {
        aguess[cell] = (ibest < npt) ? s_dom[ibest] : qnan();
        xlo[cell] = res[0];
        xhi[cell] = res[1];
    }
}

// The same for profiles of at most 256 points (the light curves use 200 and 100): FOUR rows a
// wavefront, a row of sixteen lanes (one DPP row) each.  Lane p of a row keeps points
// PPL p .. PPL p + PPL - 1 in registers (PPL = 7, 13, 16 for rows of up to 112, 208, 256 points),
// adds its own trapezoids serially, and the sixteen lane totals are scanned by DPP shifts inside
// the row: every cross-lane step serves four rows, all lanes hold points (a wavefront per row of
// 200 points left 14 lanes idle and spent 256 exponentials on it), and the kernel, which is
// bound by its arithmetic and not by the bytes it reads, takes half the instructions a row.
// The four rows (contiguous in memory) are fetched with consecutive lanes on consecutive
// numbers and turned through LDS.  Same results as the kernel above up to the order of the
// additions.
// LINSPACE: the abscissae of a row are numpy.linspace(first, last, npt) of its ends to the bit
// (what the profile kernels write): only the two ends are read, the row's abscissae are formed
// as those kernels form them, and the kernel moves half the bytes.
constexpr int PQ16_ROWS = 16;    // rows a workgroup of 256 threads
template <int PPL, bool LINSPACE>
__global__ __launch_bounds__(256) void profile_quantiles16_kernel(const double *dom, const double *cod,
                                                                  int64_t ncell, int npt, double c_lo,
                                                                  double c_hi, double *aguess,
                                                                  double *xlo, double *xhi)
{
    extern __shared__ double sm[];   // per wave: cod of four rows (+ dom of four rows unless LINSPACE)
    const int wave = threadIdx.x / WAVE, lane = threadIdx.x % WAVE;
    const int q = lane >> 4, p = lane & 15;
    const int64_t row0 = (int64_t)blockIdx.x * PQ16_ROWS + wave * 4;
    if (row0 >= ncell) return;                  // whole wavefronts leave together
    const int nrow = ncell - row0 < 4 ? (int)(ncell - row0) : 4;
    const int per = LINSPACE ? 1 : 2;
    double *s_cod = sm + (size_t)wave * per * 4 * npt;
    double *s_dom = s_cod + 4 * npt;
    for (int i = lane; i < nrow * npt; i += WAVE) {
        s_cod[i] = cod[row0 * npt + i];
        if (!LINSPACE) s_dom[i] = dom[row0 * npt + i];
    }
    // the slice belongs to this wavefront alone and LDS serves a wavefront's accesses in
    // order: only the compiler has to be kept from reordering
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const bool have = q < nrow;
    const int64_t cell = row0 + (have ? q : 0);
    const double *cp = s_cod + (have ? q : 0) * npt;
    const double *dp = s_dom + (have ? q : 0) * npt;
    const int i0 = PPL * p;
    const int nseg = npt - 1;
    double first = 0.0, last = 0.0, step = 0.0;
    if (LINSPACE) {
        first = dom[cell * npt];
        last = dom[cell * npt + npt - 1];
        step = (last - first) / (double)(npt - 1);
    }
    auto abscissa = [&](int i) {
        const int ii = i < npt ? i : npt - 1;
        return LINSPACE ? grid_point(first, last, step, ii, npt) : dp[ii];
    };
    // this lane's points: pdf = exp(cod), the first maximum of cod, the trapezoids
    double pdf[PPL + 1];
    double best = -INFINITY;
    int ibest = npt;
#pragma unroll
    for (int k = 0; k < PPL; ++k) {
        const int i = i0 + k;
        const bool in = i < npt;
        const double v = cp[in ? i : npt - 1];
        pdf[k] = in ? exp(v) : 0.0;
        if (in && v > best) {
            best = v;
            ibest = i;
        }
    }
    {
        // (DPP reductions inside the row of sixteen lanes; a NaN is never a lane's best)
        const double gbest = team_max<16>(best);
        ibest = team_min_i32<16>(best == gbest ? ibest : npt);
    }
    pdf[PPL] = dpp_f64<0x101>(pdf[0]);          // row_shl:1: the next lane's first point (0 past the row's end)
    // inclusive sums of this lane's trapezoids: s[k] = seg(i0) + ... + seg(i0 + k)
    double s[PPL];
    double run = 0.0;
    {
        double dk = abscissa(i0);
#pragma unroll
        for (int k = 0; k < PPL; ++k) {
            const int i = i0 + k;
            const double dn = abscissa(i + 1);
            if (i < nseg) run += (dn - dk) * (pdf[k + 1] + pdf[k]) / 2.0;
            s[k] = run;
            dk = dn;
        }
    }
    // running sum of the lane totals inside the row (zeros come in at its start);
    // cdf[i0] = before, cdf[i0 + k + 1] = before + s[k]
    double inrow = run;
    inrow += dpp_f64<0x111>(inrow);             // row_shr:1, 2, 4, 8
    inrow += dpp_f64<0x112>(inrow);
    inrow += dpp_f64<0x114>(inrow);
    inrow += dpp_f64<0x118>(inrow);
    const double before = dpp_f64<0x111>(inrow);
    const double total = __shfl(inrow, 15, 16);
    const double inv = 1. / total;
    const double conf[2] = {c_lo, c_hi};
    double res[2];
    double ag = qnan();
    if (ibest < npt) ag = abscissa(ibest);      // every lane of the row forms it (or reads it from LDS)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        // searchsorted(cdf, conf) (side='left'): the first index whose cdf reaches conf; this
        // lane's first such point, with the cdf at it and before it
        int kfirst = PPL;
        double chi = 0.0, clo = 0.0;
#pragma unroll
        for (int k = PPL - 1; k >= 0; --k) {
            const double ck = (before + s[k]) * inv;
            if (i0 + k < nseg && ck >= conf[c]) {
                kfirst = k;
                chi = ck;
                clo = (k == 0 ? before : before + s[k > 0 ? k - 1 : 0]) * inv;
            }
        }
        const unsigned long long m = __ballot(kfirst < PPL ? 1 : 0);
        const unsigned mine_row = (unsigned)(m >> (16 * q)) & 0xffffu;
        double r = qnan();
        const bool at0 = (0.0 * inv >= conf[c]);              // cdf[0] = 0 reaches it: index 0, no interval
        double mine = qnan();
        if (kfirst < PPL) {
            const double dlo = abscissa(i0 + kfirst), dhi = abscissa(i0 + kfirst + 1);
            const double slope = (chi - clo) / (dhi - dlo);
            mine = dlo + (conf[c] - clo) / slope;
        }
        const int src = mine_row ? __builtin_ctz(mine_row) : 0;
        const double got = __shfl(mine, src, 16);
        if (mine_row != 0 && !at0) r = got;
        res[c] = r;
    }
    if (have && p == 0) {
        aguess[cell] = ag;
        xlo[cell] = res[0];
        xhi[cell] = res[1];
    }
}

template <bool LINSPACE>
void launch_quantiles16(const double *dom, const double *cod, int64_t ncell, int npt, double c_lo, double c_hi,
                        double *aguess, double *xlo, double *xhi, hipStream_t st)
{
    const dim3 grid((unsigned)ceil_div(ncell, (int64_t)PQ16_ROWS));
    const size_t smem = (size_t)(256 / WAVE) * (LINSPACE ? 1 : 2) * 4 * npt * sizeof(double);
    if (npt <= 16 * 7)
        hipLaunchKernelGGL((profile_quantiles16_kernel<7, LINSPACE>), grid, dim3(256), smem, st, dom, cod, ncell, npt,
                           c_lo, c_hi, aguess, xlo, xhi);
    else if (npt <= 16 * 13)
        hipLaunchKernelGGL((profile_quantiles16_kernel<13, LINSPACE>), grid, dim3(256), smem, st, dom, cod, ncell, npt,
                           c_lo, c_hi, aguess, xlo, xhi);
    else
        hipLaunchKernelGGL((profile_quantiles16_kernel<16, LINSPACE>), grid, dim3(256), smem, st, dom, cod, ncell, npt,
                           c_lo, c_hi, aguess, xlo, xhi);
}

// rows 1..5 of get_raw_lightcurve (core.py:1318-1334)
__global__ void lightcurve_rows_kernel(const double *tstart, const double *tstop, const double *alpha,
                                       const double *ts, const double *xlo, const double *xhi,
                                       const double *S, int64_t ncell, double tsmin, double *rows)
{
    int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncell) return;
    double tmid = 0.5 * (tstart[c] + tstop[c]);
    double terr = (tmid - tstart[c]) / 86400;
    double a = alpha[c], t = ts[c];
    double *r = rows + c * 5;
    r[0] = terr;
    if (isnan(a)) {
        // a cell that could not be fitted (the reference stops with an exception there): no
        // number in its row
        r[1] = r[2] = r[3] = r[4] = a;
        return;
    }
    // A profile that holds only NaN does not stop the reference (the search for the upper end
    // of the support of a cell with alpha-hat = 0 can leave the domain of the logarithm,
    // core.py:869-880): the quantiles are NaN and take their places in the row like numbers.
    if (t <= tsmin) {
        r[1] = xhi[c];
        r[2] = 0;
        r[3] = -1;
    } else {
        r[1] = a;
        r[2] = a - xlo[c];
        r[3] = xhi[c] - a;
    }
    r[4] = t;
    (void)S;
}

// met * (1 / 86400) + mjdref as the reference forms it (core.py:24-26): in the 80-bit
// extended precision of numpy's longdouble on x86 (64-bit significand, round to nearest
// even after the product and after the sum), then rounded to double where the value is
// stored.  Restated in integer arithmetic: exact 106-bit product, each rounding done on
// the integer significand.  Positive normal operands whose sum fits 128 bits after
// alignment; anything else yields NaN and is left to the caller.
__device__ __forceinline__ int top_bit(unsigned __int128 v)
{
    const unsigned long long hi = (unsigned long long)(v >> 64), lo = (unsigned long long)v;
    return hi ? 128 - __builtin_clzll(hi) : (lo ? 64 - __builtin_clzll(lo) : 0);
}

// v (a significand of more than `keep` bits) rounded to nearest even at `keep` bits;
// e is the exponent of v's unit and follows the shift
__device__ __forceinline__ unsigned __int128 round_bits(unsigned __int128 v, int keep, int &e)
{
    const int nb = top_bit(v);
    if (nb <= keep) return v;
    int sh = nb - keep;
    unsigned __int128 q = v >> sh;
    const unsigned __int128 rem = v & ((((unsigned __int128)1) << sh) - 1);
    const unsigned __int128 half = ((unsigned __int128)1) << (sh - 1);
    if (rem > half || (rem == half && (q & 1))) ++q;
    if (top_bit(q) > keep) {
        q >>= 1;        // the carry ran through: a power of two, nothing is lost
        ++sh;
    }
    e += sh;
    return q;
}

__device__ __forceinline__ bool split_positive_normal(double x, unsigned long long &mant, int &e)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(x);
    const int be = (int)((b >> 52) & 0x7ff);
    if ((b >> 63) || be == 0 || be == 0x7ff) return false;
    mant = (b & ((1ull << 52) - 1)) | (1ull << 52);
    e = be - 1075;
    return true;
}

__global__ void met2mjd_kernel(const double *met, int64_t n, double per_day, double mjdref, double *mjd)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned long long mt, mc, mm;
    int et, ec, em;
    double out = qnan();
    if (split_positive_normal(met[i], mt, et) && split_positive_normal(per_day, mc, ec) &&
        split_positive_normal(mjdref, mm, em)) {
        int ep = et + ec;
        const unsigned __int128 p = round_bits((unsigned __int128)mt * mc, 64, ep);    // fl80(t * c)
        const int elow = ep < em ? ep : em;
        const int dp = ep - elow, dm = em - elow;
        if (dp <= 60 && dm <= 60) {
            int es = elow;
            unsigned __int128 sum = (p << dp) + (((unsigned __int128)mm) << dm);
            sum = round_bits(sum, 64, es);                                             // fl80(. + mjdref)
            sum = round_bits(sum, 53, es);                                             // stored as double
            out = ldexp((double)(unsigned long long)sum, es);
        }
    }
    mjd[i] = out;
}

}   // namespace

extern "C" int godot_met2mjd(const double *met, int64_t n, double mjdref, double *mjd, godot_stream_t stream)
{
    GODOT_ENTRY();
    GODOT_REQUIRE(n >= 0, GODOT_E_INVALID, "godot_met2mjd: negative n");
    if (n == 0) return GODOT_OK;
    GODOT_REQUIRE(met && mjd, GODOT_E_INVALID, "godot_met2mjd: null pointer");
    hipLaunchKernelGGL(met2mjd_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, met, n,
                       1. / 86400, mjdref, mjd);
    GODOT_LAUNCH_CHECK("godot_met2mjd");
    return GODOT_OK;
}

extern "C" int godot_cell_fit(const double *we, int64_t nph, const int64_t *lo, const int64_t *hi,
                              const double *S, const double *B, int64_t ncell, const double *guess,
                              double guess0, const double *beta, double beta0, int swap,
                              double *alpha, double *ts, int32_t *flags, godot_stream_t stream)
{
    GODOT_ENTRY();
    int rc = check_cells(we, nph, lo, hi, ncell, "godot_cell_fit");
    if (rc) return rc;
    if (ncell == 0) return GODOT_OK;
    GODOT_REQUIRE(S && alpha, GODOT_E_INVALID, "godot_cell_fit: null S or alpha");
    GODOT_REQUIRE(B || !swap, GODOT_E_INVALID, "godot_cell_fit: swap needs B");
    hipStream_t st = (hipStream_t)stream;
    CellArgs p{we, nph, lo, hi, S, B, ncell, swap};
    int gl, gr;
    if (godot_detail::fit_block_cells(nph, ncell)) {
        godot_detail::fit_block_launch(we, nph, lo, hi, S, B, ncell, guess, guess0, beta, beta0, swap, alpha, ts,
                                       flags, st);
    } else if (godot_detail::fit_group_shape(nph, ncell, gl, gr, false)) {
        // groups of 1 to 64 lanes per cell (cells_group.hip)
        GODOT_REQUIRE(godot_detail::fit_group_launch(gl, gr, we, nph, lo, hi, S, B, ncell, guess, guess0, beta, beta0,
                                                     swap, alpha, ts, flags, st),
                      GODOT_E_UNSUPPORTED, "godot_cell_fit: no kernel for the chosen shape of lane groups");
    } else if (!beta && beta0 == 1.0) {
        DISPATCH_TEAM(pick_team(nph, ncell), cell_fit_b0_kernel, 0, p, guess, guess0, beta, beta0, alpha, ts, flags);
    } else {
        DISPATCH_TEAM(pick_team(nph, ncell), cell_fit_bx_kernel, 0, p, guess, guess0, beta, beta0, alpha, ts, flags);
    }
    GODOT_LAUNCH_CHECK("godot_cell_fit");
    return GODOT_OK;
}

extern "C" int godot_cell_fit_shaped(const double *we, int64_t nph, const int64_t *lo, const int64_t *hi,
                                     const double *S, const double *B, int64_t ncell, int64_t photons_in_cells,
                                     const double *guess, double guess0, const double *beta, double beta0,
                                     int swap, double *alpha, double *ts, int32_t *flags, godot_stream_t stream)
{
    GODOT_REQUIRE(photons_in_cells >= 0, GODOT_E_INVALID, "godot_cell_fit_shaped: negative photon count");
    GodotShapeHint hint(photons_in_cells);
    return godot_cell_fit(we, nph, lo, hi, S, B, ncell, guess, guess0, beta, beta0, swap, alpha, ts, flags, stream);
}

extern "C" int godot_cell_loglike(const double *we, int64_t nph, const int64_t *lo, const int64_t *hi,
                                  const double *S, const double *B, int64_t ncell,
                                  const double *alpha, int swap, double *logl, godot_stream_t stream)
{
    GODOT_ENTRY();
    int rc = check_cells(we, nph, lo, hi, ncell, "godot_cell_loglike");
    if (rc) return rc;
    if (ncell == 0) return GODOT_OK;
    GODOT_REQUIRE(S && alpha && logl, GODOT_E_INVALID, "godot_cell_loglike: null argument");
    GODOT_REQUIRE(B || !swap, GODOT_E_INVALID, "godot_cell_loglike: swap needs B");
    hipStream_t st = (hipStream_t)stream;
    CellArgs p{we, nph, lo, hi, S, B, ncell, swap};
    DISPATCH_TEAM(pick_team(nph, ncell), cell_loglike_kernel, 0, p, alpha, logl);
    GODOT_LAUNCH_CHECK("godot_cell_loglike");
    return GODOT_OK;
}

extern "C" int godot_cell_sums(const double *we, int64_t nph, const int64_t *lo, const int64_t *hi, int64_t ncell,
                               const double *alpha, double alpha0, const double *beta, double beta0, int swap,
                               double *sums, godot_stream_t stream)
{
    GODOT_ENTRY();
    int rc = check_cells(we, nph, lo, hi, ncell, "godot_cell_sums");
    if (rc) return rc;
    if (ncell == 0) return GODOT_OK;
    GODOT_REQUIRE(sums, GODOT_E_INVALID, "godot_cell_sums: null output");
    hipStream_t st = (hipStream_t)stream;
    CellArgs p{we, nph, lo, hi, nullptr, nullptr, ncell, swap};
    DISPATCH_TEAM(pick_team(nph, ncell), cell_sums_kernel, 0, p, alpha, alpha0, beta, beta0, sums);
    GODOT_LAUNCH_CHECK("godot_cell_sums");
    return GODOT_OK;
}

// fewer than three photons a cell on average and teams of four lanes otherwise: a thread a cell
// (the same additions as the team's, see sparse_moments)
static bool sparse_cells(int64_t nph, int64_t ncell)
{
    return pick_team(nph, ncell) == 0 && (double)godot_shape_photons(nph) < 3.0 * (double)ncell;
}

extern "C" int godot_cell_moments(const double *we, int64_t nph, const int64_t *lo, const int64_t *hi,
                                  int64_t ncell, double *wsum, double *w2sum, godot_stream_t stream)
{
    GODOT_ENTRY();
    int rc = check_cells(we, nph, lo, hi, ncell, "godot_cell_moments");
    if (rc) return rc;
    if (ncell == 0) return GODOT_OK;
    GODOT_REQUIRE(wsum && w2sum, GODOT_E_INVALID, "godot_cell_moments: null output");
    hipStream_t st = (hipStream_t)stream;
    CellArgs p{we, nph, lo, hi, nullptr, nullptr, ncell, 0};
    if (sparse_cells(nph, ncell))
        hipLaunchKernelGGL(cell_moments_sparse_kernel, dim3((unsigned)ceil_div(ncell, (int64_t)256)), dim3(256), 0, st,
                           p, wsum, w2sum);
    else
        DISPATCH_TEAM(pick_team(nph, ncell), cell_moments_kernel, 0, p, wsum, w2sum);
    GODOT_LAUNCH_CHECK("godot_cell_moments");
    return GODOT_OK;
}

extern "C" int godot_cell_series(const double *we, int64_t nph, const int64_t *lo, const int64_t *hi,
                                 const double *exp, int64_t ncell, double *sexp, double *bexp, double *W,
                                 double *WW, double *Wb, double *WbWb, godot_stream_t stream)
{
    GODOT_ENTRY();
    int rc = check_cells(we, nph, lo, hi, ncell, "godot_cell_series");
    if (rc) return rc;
    if (ncell == 0) return GODOT_OK;
    GODOT_REQUIRE(exp && sexp && bexp && W && WW && Wb && WbWb, GODOT_E_INVALID, "godot_cell_series: null argument");
    hipStream_t st = (hipStream_t)stream;
    CellArgs p{we, nph, lo, hi, nullptr, nullptr, ncell, 0};
    if (sparse_cells(nph, ncell))
        hipLaunchKernelGGL(cell_series_sparse_kernel, dim3((unsigned)ceil_div(ncell, (int64_t)256)), dim3(256), 0, st,
                           p, exp, sexp, bexp, W, WW, Wb, WbWb);
    else
        DISPATCH_TEAM(pick_team(nph, ncell), cell_series_kernel, 0, p, exp, sexp, bexp, W, WW, Wb, WbWb);
    GODOT_LAUNCH_CHECK("godot_cell_series");
    return GODOT_OK;
}

extern "C" int godot_cell_beta_max(const double *we, int64_t nph, const int64_t *lo, const int64_t *hi,
                                   const double *S, const double *B, int64_t ncell,
                                   const double *alpha, double alpha0, const double *guess,
                                   double guess0, int swap, double *beta, double *logl_profile,
                                   int32_t *flags, godot_stream_t stream)
{
    GODOT_ENTRY();
    int rc = check_cells(we, nph, lo, hi, ncell, "godot_cell_beta_max");
    if (rc) return rc;
    if (ncell == 0) return GODOT_OK;
    GODOT_REQUIRE(S && B && beta, GODOT_E_INVALID, "godot_cell_beta_max: null argument");
    hipStream_t st = (hipStream_t)stream;
    CellArgs p{we, nph, lo, hi, S, B, ncell, swap};
    DISPATCH_TEAM(pick_team(nph, ncell), cell_beta_kernel, 0, p, alpha, alpha0, guess, guess0, beta,
                  logl_profile, flags);
    GODOT_LAUNCH_CHECK("godot_cell_beta_max");
    return GODOT_OK;
}

extern "C" int godot_cell_logpdf(const double *we, int64_t nph, const int64_t *lo, const int64_t *hi,
                                 const double *S, const double *B, int64_t ncell, const double *aopt,
                                 double dlogl, int npt, int include_zero, int swap, double *dom,
                                 double *cod, int32_t *flags, godot_stream_t stream)
{
    GODOT_ENTRY();
    int rc = check_cells(we, nph, lo, hi, ncell, "godot_cell_logpdf");
    if (rc) return rc;
    GODOT_REQUIRE(npt >= 2 && npt <= PQ_MAX_NPT, GODOT_E_UNSUPPORTED,
                  "godot_cell_logpdf: npt must be in [2, %d]", PQ_MAX_NPT);
    if (ncell == 0) return GODOT_OK;
    GODOT_REQUIRE(S && dom && cod, GODOT_E_INVALID, "godot_cell_logpdf: null argument");
    GODOT_REQUIRE(B || !swap, GODOT_E_INVALID, "godot_cell_logpdf: swap needs B");
    hipStream_t st = (hipStream_t)stream;
    CellArgs p{we, nph, lo, hi, S, B, ncell, swap};
    int gl = 0, gr = 0;
    const bool group = !godot_detail::block_cells(nph, ncell) && godot_detail::fit_group_shape(nph, ncell, gl, gr, true);
    // lane groups (cells_group.hip) wherever a shape holds the cells and there are wavefronts
    // enough to give every SIMD two; a workgroup per cell for very large cells and for few
    // moderately large ones
    if (group && ncell * gl >= 64 * 2048) {
        GODOT_REQUIRE(godot_detail::logpdf_group_launch(gl, gr, we, nph, lo, hi, S, B, ncell, aopt, dlogl, npt,
                                                        include_zero, swap, dom, cod, flags, st),
                      GODOT_E_UNSUPPORTED, "godot_cell_logpdf: no kernel for the chosen shape of lane groups");
    } else if (godot_detail::logpdf_block_cells(nph, ncell)) {
        godot_detail::logpdf_block_launch(we, nph, lo, hi, S, B, ncell, aopt, dlogl, npt, include_zero, swap, dom,
                                          cod, flags, st);
    } else if (group) {
        GODOT_REQUIRE(godot_detail::logpdf_group_launch(gl, gr, we, nph, lo, hi, S, B, ncell, aopt, dlogl, npt,
                                                        include_zero, swap, dom, cod, flags, st),
                      GODOT_E_UNSUPPORTED, "godot_cell_logpdf: no kernel for the chosen shape of lane groups");
    } else {
        size_t smem = (size_t)LP_CH * npt * sizeof(double);
        dim3 grid((unsigned)ceil_div(ncell, LP_CH));
        hipLaunchKernelGGL(cell_logpdf_kernel, grid, dim3(256), smem, st, p, we, aopt, dlogl, npt,
                           include_zero, dom, cod, flags);
    }
    GODOT_LAUNCH_CHECK("godot_cell_logpdf");
    return GODOT_OK;
}

extern "C" int godot_cell_logpdf_shaped(const double *we, int64_t nph, const int64_t *lo, const int64_t *hi,
                                        const double *S, const double *B, int64_t ncell, int64_t photons_in_cells,
                                        const double *aopt, double dlogl, int npt, int include_zero, int swap,
                                        double *dom, double *cod, int32_t *flags, godot_stream_t stream)
{
    GODOT_REQUIRE(photons_in_cells >= 0, GODOT_E_INVALID, "godot_cell_logpdf_shaped: negative photon count");
    GodotShapeHint hint(photons_in_cells);
    return godot_cell_logpdf(we, nph, lo, hi, S, B, ncell, aopt, dlogl, npt, include_zero, swap, dom, cod, flags,
                             stream);
}

extern "C" int godot_profile_quantiles(const double *dom, const double *cod, int64_t ncell, int npt,
                                       double conf_lo, double conf_hi, double *aguess, double *xlo,
                                       double *xhi, godot_stream_t stream)
{
    GODOT_ENTRY();
    GODOT_REQUIRE(ncell >= 0, GODOT_E_INVALID, "godot_profile_quantiles: negative ncell");
    GODOT_REQUIRE(npt >= 2 && npt <= PQ_MAX_NPT, GODOT_E_UNSUPPORTED,
                  "godot_profile_quantiles: npt must be in [2, %d]", PQ_MAX_NPT);
    if (ncell == 0) return GODOT_OK;
    GODOT_REQUIRE(dom && cod && aguess && xlo && xhi, GODOT_E_INVALID,
                  "godot_profile_quantiles: null argument");
    hipStream_t st = (hipStream_t)stream;
    size_t smem = (size_t)4 * 3 * npt * sizeof(double);
    dim3 grid((unsigned)ceil_div(ncell, 4));
    if (npt <= 256)
        launch_quantiles16<false>(dom, cod, ncell, npt, conf_lo, conf_hi, aguess, xlo, xhi, st);
    else
        hipLaunchKernelGGL(profile_quantiles_kernel, grid, dim3(256), smem, st, dom, cod, ncell, npt,
                           conf_lo, conf_hi, aguess, xlo, xhi);
    GODOT_LAUNCH_CHECK("godot_profile_quantiles");
    return GODOT_OK;
}

extern "C" int godot_profile_quantiles_linspace(const double *dom, const double *cod, int64_t ncell, int npt,
                                                double conf_lo, double conf_hi, double *aguess, double *xlo,
                                                double *xhi, godot_stream_t stream)
{
    if (npt > 256)      // long rows: the general kernel (its rows are staged in LDS either way)
        return godot_profile_quantiles(dom, cod, ncell, npt, conf_lo, conf_hi, aguess, xlo, xhi, stream);
    GODOT_ENTRY();
    GODOT_REQUIRE(ncell >= 0, GODOT_E_INVALID, "godot_profile_quantiles_linspace: negative ncell");
    GODOT_REQUIRE(npt >= 2, GODOT_E_UNSUPPORTED, "godot_profile_quantiles_linspace: npt must be at least 2");
    if (ncell == 0) return GODOT_OK;
    GODOT_REQUIRE(dom && cod && aguess && xlo && xhi, GODOT_E_INVALID,
                  "godot_profile_quantiles_linspace: null argument");
    launch_quantiles16<true>(dom, cod, ncell, npt, conf_lo, conf_hi, aguess, xlo, xhi, (hipStream_t)stream);
    GODOT_LAUNCH_CHECK("godot_profile_quantiles_linspace");
    return GODOT_OK;
}

extern "C" int godot_lightcurve_rows(const double *tstart, const double *tstop, const double *alpha,
                                     const double *ts, const double *xlo, const double *xhi,
                                     const double *S, int64_t ncell, double tsmin, double *rows,
                                     godot_stream_t stream)
{
    GODOT_ENTRY();
    GODOT_REQUIRE(ncell >= 0, GODOT_E_INVALID, "godot_lightcurve_rows: negative ncell");
    if (ncell == 0) return GODOT_OK;
    GODOT_REQUIRE(tstart && tstop && alpha && ts && xlo && xhi && rows, GODOT_E_INVALID,
                  "godot_lightcurve_rows: null argument");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(lightcurve_rows_kernel, dim3((unsigned)ceil_div(ncell, 256)), dim3(256), 0, st,
                       tstart, tstop, alpha, ts, xlo, xhi, S, ncell, tsmin, rows);
    GODOT_LAUNCH_CHECK("godot_lightcurve_rows");
    return GODOT_OK;
}
