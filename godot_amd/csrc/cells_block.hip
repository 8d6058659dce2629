// Very large cells (tens of thousands to millions of photons each, few of
// them): what merging the cells of a Bayesian block produces (reference
// core.py:200-210, 1366-1381).  One workgroup of 1024 threads owns one cell and
// streams its weights in every pass; sums over the cell are reduced inside
// the wavefronts and then across the sixteen wavefronts through LDS, in a
// fixed order, so every thread holds the same value and the schedule's
// branches are uniform.  The schedule itself (get_max, get_ts, the support
// search and the profile grid of get_logpdf) is the shared code of
// cells_common.hpp.

#include "cells_common.hpp"

namespace {

constexpr int BK_THREADS = 1024;
constexpr int BK_WAVES = BK_THREADS / 64;

// The cell as the whole workgroup sees it.  Same interface as CellData.
struct BlockCell {
    const double *p;
    int64_t n;
    int tid;
    bool swap;
    double *red;       // LDS: 3 * BK_WAVES doubles
    int *redi;         // LDS: BK_WAVES ints

    __device__ __forceinline__ double ld(int64_t i) const
    {
        const double v = p[i];
        return swap ? 1.0 - v : v;
    }

    // sum of NS per-thread values over the workgroup, identical in every thread
    template <int NS>
    __device__ __forceinline__ void block_sum(double (&acc)[NS]) const
    {
        const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const double s = team_sum<64>(acc[k]);
            if (lane == 0) red[k * BK_WAVES + wave] = s;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            double s = 0.0;
#pragma unroll
            for (int w = 0; w < BK_WAVES; ++w) s += red[k * BK_WAVES + w];
            acc[k] = s;
        }
        __syncthreads();
    }

    // product of per-thread mantissas in [0.5, 1) (times 2^ex), over the workgroup
    __device__ __forceinline__ void block_prod(double &pr, int &ex) const
    {
        const int wave = tid >> 6, lane = tid & 63;
        pr = team_prod<64>(pr);          // at worst 2^-64
        ex = team_sum_i32<64>(ex);
        ex += __builtin_amdgcn_frexp_exp(pr);
        pr = __builtin_amdgcn_frexp_mant(pr);
        if (lane == 0) {
            red[wave] = pr;
            redi[wave] = ex;
        }
        __syncthreads();
        double m = 1.0;
        int e = 0;
#pragma unroll
        for (int w = 0; w < BK_WAVES; ++w) {
            m *= red[w];                 // sixteen factors in [0.5, 1)
            e += redi[w];
        }
        e += __builtin_amdgcn_frexp_exp(m);
        m = __builtin_amdgcn_frexp_mant(m);
        pr = m;
        ex = e;
        __syncthreads();
    }

    template <int NS, class DEN, class ACC>
    __device__ __forceinline__ void sums_rcp(DEN den, ACC accf, double (&acc)[NS]) const
    {
#pragma unroll
        for (int k = 0; k < NS; ++k) acc[k] = 0.0;
        int64_t i = tid;
        // four loads in flight per thread, their reciprocals from one hardware rcp
        for (; i + 3 * BK_THREADS < n; i += 4 * BK_THREADS) {
            const double w0 = ld(i), w1 = ld(i + BK_THREADS), w2 = ld(i + 2 * BK_THREADS),
                         w3 = ld(i + 3 * BK_THREADS);
            const double d0 = den(w0), d1 = den(w1), d2 = den(w2), d3 = den(w3);
            double x0, x1, x2, x3;
            rcp4(d0, d1, d2, d3, x0, x1, x2, x3);
            accf(w0, d0, x0, acc);
            accf(w1, d1, x1, acc);
            accf(w2, d2, x2, acc);
            accf(w3, d3, x3, acc);
        }
        for (; i < n; i += BK_THREADS) {
            const double wv = ld(i), d = den(wv);
            accf(wv, d, rcp_fast1(d), acc);
        }
        block_sum<NS>(acc);
    }

    template <class DEN, class GF>
    __device__ __forceinline__ void logsum_rcp(DEN den, GF gf, double &logs, double &gsum) const
    {
        double pr = 1.0;
        double gs[1] = {0.0};
        int ex = 0, k = 0;
        int64_t i = tid;
        for (; i + 3 * BK_THREADS < n; i += 4 * BK_THREADS) {
            const double w0 = ld(i), w1 = ld(i + BK_THREADS), w2 = ld(i + 2 * BK_THREADS),
                         w3 = ld(i + 3 * BK_THREADS);
            const double d0 = den(w0), d1 = den(w1), d2 = den(w2), d3 = den(w3);
            double x0, x1, x2, x3;
            rcp4(d0, d1, d2, d3, x0, x1, x2, x3);
            gs[0] += gf(w0, x0);
            gs[0] += gf(w1, x1);
            gs[0] += gf(w2, x2);
            gs[0] += gf(w3, x3);
            pr *= (d0 * d1) * (d2 * d3);
            if ((++k & 1) == 0) {
                ex += __builtin_amdgcn_frexp_exp(pr);
                pr = __builtin_amdgcn_frexp_mant(pr);
            }
        }
        ex += __builtin_amdgcn_frexp_exp(pr);
        pr = __builtin_amdgcn_frexp_mant(pr);
        for (; i < n; i += BK_THREADS) {
            const double wv = ld(i), d = den(wv);
            gs[0] += gf(wv, rcp_fast1(d));
            pr *= d;
        }
        ex += __builtin_amdgcn_frexp_exp(pr);
        pr = __builtin_amdgcn_frexp_mant(pr);
        block_prod(pr, ex);
        block_sum<1>(gs);
        gsum = gs[0];
        logs = log_mant(pr, ex);
    }

    template <bool ZS = false, class F>
    __device__ __forceinline__ double logratio(F f) const
    {
        double p1 = 1.0, p2 = 1.0;
        int e1 = 0, e2 = 0, k = 0;
        int64_t i = tid;
        for (; i + 3 * BK_THREADS < n; i += 4 * BK_THREADS) {
            const double w0 = ld(i), w1 = ld(i + BK_THREADS), w2 = ld(i + 2 * BK_THREADS),
                         w3 = ld(i + 3 * BK_THREADS);
            double a1, a2, b1, b2, c1, c2, d1, d2;
            f(w0, a1, a2);
            f(w1, b1, b2);
            f(w2, c1, c2);
            f(w3, d1, d2);
            p1 *= (a1 * b1) * (c1 * d1);
            p2 *= (a2 * b2) * (c2 * d2);
            if ((++k & 1) == 0) {
                e1 += __builtin_amdgcn_frexp_exp(p1);
                p1 = __builtin_amdgcn_frexp_mant(p1);
                e2 += __builtin_amdgcn_frexp_exp(p2);
                p2 = __builtin_amdgcn_frexp_mant(p2);
            }
        }
        e1 += __builtin_amdgcn_frexp_exp(p1);
        p1 = __builtin_amdgcn_frexp_mant(p1);
        e2 += __builtin_amdgcn_frexp_exp(p2);
        p2 = __builtin_amdgcn_frexp_mant(p2);
        k = 0;
        for (; i < n; i += BK_THREADS) {
            double t1, t2;
            f(ld(i), t1, t2);
            p1 *= t1;
            p2 *= t2;
            if ((++k & 7) == 0) {
                e1 += __builtin_amdgcn_frexp_exp(p1);
                p1 = __builtin_amdgcn_frexp_mant(p1);
                e2 += __builtin_amdgcn_frexp_exp(p2);
                p2 = __builtin_amdgcn_frexp_mant(p2);
            }
        }
        e1 += __builtin_amdgcn_frexp_exp(p1);
        p1 = __builtin_amdgcn_frexp_mant(p1);
        e2 += __builtin_amdgcn_frexp_exp(p2);
        p2 = __builtin_amdgcn_frexp_mant(p2);
        block_prod(p1, e1);
        block_prod(p2, e2);
        return log_scaled(p1 * rcp_fast(p2), e1 - e2);
    }

    __device__ __forceinline__ double wmax() const
    {
        double m = -INFINITY;
        int64_t i = tid;
        for (; i + 3 * BK_THREADS < n; i += 4 * BK_THREADS)
            m = fmax(fmax(m, fmax(ld(i), ld(i + BK_THREADS))),
                     fmax(ld(i + 2 * BK_THREADS), ld(i + 3 * BK_THREADS)));
        for (; i < n; i += BK_THREADS) m = fmax(m, ld(i));
        const int wave = tid >> 6, lane = tid & 63;
        m = team_max<64>(m);
        if (lane == 0) red[wave] = m;
        __syncthreads();
        double r = -INFINITY;
#pragma unroll
        for (int w = 0; w < BK_WAVES; ++w) r = fmax(r, red[w]);
        __syncthreads();
        return r;
    }
};

// a value every lane holds alike, as a scalar
__device__ __forceinline__ int64_t uniform_i64(int64_t v)
{
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)v);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32));
    return (int64_t)(((unsigned long long)hi << 32) | lo);
}

struct BlockSetup {
    int64_t lo, n;
    double S;
    bool bad;
};

__device__ __forceinline__ BlockSetup block_setup(const CellArgs &p, int64_t cell)
{
    BlockSetup b;
    b.lo = p.lo[cell];
    const int64_t hi = p.hi[cell];
    b.bad = (b.lo < 0) || (hi < b.lo) || (hi > p.nph);
    b.n = b.bad ? 0 : hi - b.lo;
    if (b.bad) b.lo = 0;
    b.S = p.swap ? p.B[cell] : p.S[cell];
    return b;
}

// get_max + get_ts (core.py:632-739, 982-994), one workgroup per cell
__global__ __launch_bounds__(BK_THREADS) void cell_fit_block_kernel(CellArgs p, const double *guess, double guess0,
                                                                    const double *beta, double beta0,
                                                                    double *alpha, double *ts, int32_t *flags)
{
    __shared__ double s_red[3 * BK_WAVES];
    __shared__ int s_redi[BK_WAVES];
    const int64_t cell = blockIdx.x;
    const BlockSetup b = block_setup(p, cell);
    BlockCell cd{p.we + b.lo, b.n, (int)threadIdx.x, p.swap != 0, s_red, s_redi};
    int fl = 0;
    double a = qnan(), t = qnan();
    if (b.bad) {
        fl = GODOT_F_FAILED;
    } else {
        a = fit_alpha(cd, b.S, guess ? guess[cell] : guess0, beta ? beta[cell] : beta0, fl);
        if (ts && !(fl & GODOT_F_FAILED)) t = test_statistic(cd, b.S, a);
    }
    if (threadIdx.x == 0) {
        alpha[cell] = a;
        if (ts) ts[cell] = t;
        if (flags) flags[cell] = fl;
    }
}

// get_logpdf (core.py:851-932), one workgroup per cell.  The grid is evaluated
// with the photons split over the sixteen wavefronts (each reads its share at
// wave-uniform addresses: scalar loads) and four grid points per lane; the
// sixteen partial products of a point are multiplied together through LDS.
constexpr int BK_PPL = 4;
constexpr int BK_SWEEP = 64 * BK_PPL;     // grid points per sweep

__global__ __launch_bounds__(BK_THREADS) void cell_logpdf_block_kernel(CellArgs p, const double *__restrict__ wts,
                                                                       const double *__restrict__ aopt_in,
                                                                       double dlogl, int npt, int include_zero,
                                                                       double *__restrict__ dom,
                                                                       double *__restrict__ cod,
                                                                       int32_t *__restrict__ flags)
{
    __shared__ double s_red[3 * BK_WAVES];
    __shared__ int s_redi[BK_WAVES];
    __shared__ double s_pm[BK_WAVES][BK_SWEEP];
    __shared__ int s_pe[BK_WAVES][BK_SWEEP];
    __shared__ double s_bv[BK_WAVES];
    __shared__ int s_bk[BK_WAVES], s_nan;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t cell = blockIdx.x;
    const bool swap = p.swap != 0;
    const BlockSetup b = block_setup(p, cell);
    BlockCell cd{p.we + b.lo, b.n, tid, swap, s_red, s_redi};
    int fl = 0;
    bool ok = !b.bad;
    double aopt = qnan();
    if (b.bad) {
        fl = GODOT_F_FAILED;
    } else if (aopt_in) {
        aopt = aopt_in[cell];
    } else {
        aopt = fit_alpha(cd, b.S, 1.0, 1.0, fl);
        if (fl & GODOT_F_FAILED) ok = false;
    }
    // this wavefront's share of the photons
    const int64_t n_u = uniform_i64(b.n), lo_u = uniform_i64(b.lo);
    const int64_t share = (n_u + BK_WAVES - 1) / BK_WAVES;
    const int64_t j0 = share * wave < n_u ? share * wave : n_u;
    const int64_t j1 = j0 + share < n_u ? j0 + share : n_u;
    const double *w = wts + lo_u;

    bool done = !ok;
    for (int round = 0; !done && round <= 50; ++round) {
        const Support sp = support_search(cd, b.S, aopt, dlogl, include_zero != 0);
        const double step = (npt > 1) ? (sp.amax - sp.amin) / (double)(npt - 1) : 0.0;
        double best = -INFINITY;
        int ibest = npt;
        bool anynan = false;
        for (int i0 = 0; i0 < npt; i0 += BK_SWEEP) {
            double av[BK_PPL], am1[BK_PPL], pr[BK_PPL];
            int ex[BK_PPL];
#pragma unroll
            for (int u = 0; u < BK_PPL; ++u) {
                int i = i0 + u * 64 + lane;
                i = i < npt ? i : npt - 1;
                av[u] = grid_point(sp.amin, sp.amax, step, i, npt);
                am1[u] = av[u] - 1.0;
                pr[u] = 1.0;
                ex[u] = 0;
            }
            // alpha w + (1 - w) as 1 + (alpha - 1) w; on complemented weights
            // alpha (1 - w) + w = alpha - (alpha - 1) w: one form, sign and offset chosen once
            double sg[BK_PPL], of[BK_PPL];
#pragma unroll
            for (int u = 0; u < BK_PPL; ++u) {
                sg[u] = swap ? -am1[u] : am1[u];
                of[u] = swap ? av[u] : 1.0;
            }
            int64_t j = j0;
            // 32 photons per trip: four scalar loads go out together and their latency
            // (the whole cell streams from HBM) is waited out once, not once per eight photons
            for (; j + 32 <= j1; j += 32) {
                double wv[32];
#pragma unroll
                for (int k = 0; k < 32; ++k) wv[k] = w[j + k];
#pragma unroll
                for (int g = 0; g < 32; g += 8) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
#pragma unroll
                        for (int u = 0; u < BK_PPL; ++u) pr[u] *= fma(sg[u], wv[g + k], of[u]);
                    }
#pragma unroll
                    for (int u = 0; u < BK_PPL; ++u) {
                        ex[u] += __builtin_amdgcn_frexp_exp(pr[u]);
                        pr[u] = __builtin_amdgcn_frexp_mant(pr[u]);
                    }
                }
            }
            for (; j + 8 <= j1; j += 8) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const double wj = w[j + k];
#pragma unroll
                    for (int u = 0; u < BK_PPL; ++u) pr[u] *= fma(sg[u], wj, of[u]);
                }
#pragma unroll
                for (int u = 0; u < BK_PPL; ++u) {
                    ex[u] += __builtin_amdgcn_frexp_exp(pr[u]);
                    pr[u] = __builtin_amdgcn_frexp_mant(pr[u]);
                }
            }
            for (; j < j1; ++j) {
                const double wj = w[j];
#pragma unroll
                for (int u = 0; u < BK_PPL; ++u) pr[u] *= fma(sg[u], wj, of[u]);
            }
#pragma unroll
            for (int u = 0; u < BK_PPL; ++u) {
                s_pm[wave][u * 64 + lane] = __builtin_amdgcn_frexp_mant(pr[u]);
                s_pe[wave][u * 64 + lane] = ex[u] + __builtin_amdgcn_frexp_exp(pr[u]);
            }
            __syncthreads();
            if (tid < BK_SWEEP) {
                const int i = i0 + tid;
                if (i < npt) {
                    double m = 1.0;
                    int e = 0;
#pragma unroll
                    for (int wv = 0; wv < BK_WAVES; ++wv) {
                        m *= s_pm[wv][tid];
                        e += s_pe[wv][tid];
                    }
                    const double a = grid_point(sp.amin, sp.amax, step, i, npt);
                    const double v = log_scaled(m, e) - b.S * a;
                    dom[cell * npt + i] = a;
                    cod[cell * npt + i] = v - sp.llmax;
                    if (isnan(v)) anynan = true;
                    if (v > best) {
                        best = v;
                        ibest = i;
                    }
                }
            }
            __syncthreads();
        }
        // maximum of the grid against the likelihood at the centre (core.py:923-929)
        {
            double gb = team_max<64>(best);
            int cand = (best == gb) ? ibest : npt;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int other = __shfl_xor(cand, o, WAVE);
                cand = other < cand ? other : cand;
            }
            if (tid == 0) s_nan = 0;
            if (lane == 0) {
                s_bv[wave] = gb;
                s_bk[wave] = cand;
            }
            __syncthreads();
            if (anynan) s_nan = 1;
            __syncthreads();
            gb = -INFINITY;
            cand = npt;
            // only the first BK_SWEEP / 64 wavefronts hold grid points
#pragma unroll
            for (int wv = 0; wv < BK_SWEEP / 64; ++wv) {
                if (s_bv[wv] > gb || (s_bv[wv] == gb && s_bk[wv] < cand)) {
                    gb = s_bv[wv];
                    cand = s_bk[wv];
                }
            }
            const bool agree = (s_nan != 0) || !(fabs(gb - sp.llmax) > 0.05);
            __syncthreads();
            if (agree) {
                done = true;
            } else if (round == 50) {
                fl |= GODOT_F_FAILED;
                ok = false;
                done = true;
            } else {
                aopt = grid_point(sp.amin, sp.amax, step, cand >= npt ? 0 : cand, npt);
                fl |= GODOT_F_RECENTRED;
            }
        }
    }
    if (!ok) {
        for (int i = tid; i < npt; i += BK_THREADS) {
            dom[cell * npt + i] = qnan();
            cod[cell * npt + i] = qnan();
        }
    }
    if (flags && tid == 0) flags[cell] = fl;
}

}   // namespace

namespace godot_detail {

// cells of this mean size and above get a workgroup each
bool block_cells(int64_t nph, int64_t ncell)
{
    return ncell > 0 && (double)nph >= 16384.0 * (double)ncell;
}

void fit_block_launch(const double *we, int64_t nph, const int64_t *lo, const int64_t *hi, const double *S,
                      const double *B, int64_t ncell, const double *guess, double guess0, const double *beta,
                      double beta0, int swap, double *alpha, double *ts, int32_t *flags, hipStream_t st)
{
    CellArgs p{we, nph, lo, hi, S, B, ncell, swap};
    hipLaunchKernelGGL(cell_fit_block_kernel, dim3((unsigned)ncell), dim3(BK_THREADS), 0, st, p, guess, guess0,
                       beta, beta0, alpha, ts, flags);
}

void logpdf_block_launch(const double *we, int64_t nph, const int64_t *lo, const int64_t *hi, const double *S,
                         const double *B, int64_t ncell, const double *aopt, double dlogl, int npt,
                         int include_zero, int swap, double *dom, double *cod, int32_t *flags, hipStream_t st)
{
    CellArgs p{we, nph, lo, hi, S, B, ncell, swap};
    hipLaunchKernelGGL(cell_logpdf_block_kernel, dim3((unsigned)ncell), dim3(BK_THREADS), 0, st, p, we, aopt,
                       dlogl, npt, include_zero, dom, cod, flags);
}

}   // namespace godot_detail
