// Very large cells (tens of thousands to millions of photons each, few of
// them): what merging the cells of a Bayesian block produces (reference
// core.py:200-210, 1366-1381).  One workgroup of 1024 threads owns one cell and
// streams its weights in every pass; sums over the cell are reduced inside
// the wavefronts and then across the sixteen wavefronts through LDS, in a
// fixed order, so every thread holds the same value and the schedule's
// branches are uniform.  The schedule itself (get_max, get_ts, the support
// search and the profile grid of get_logpdf) is the shared code of
// cells_common.hpp.
//
// When the cells are fewer than the compute units, each cell is spread over G
// workgroups (at most one per compute unit, so that all of them are resident
// together): a workgroup streams its share of the photons, publishes its
// partial sums in a slot of device memory, waits for the other G - 1 at a
// barrier of the cell's own, and adds the G partial results in a fixed order, so
// that every workgroup of a cell holds the same bits and walks the same path
// through the schedule.  One compute unit streams at most some tens of GB/s;
// a cell of millions of photons needs many of them.

#include "cells_common.hpp"

#include <mutex>

namespace {

constexpr int BK_THREADS = 1024;
constexpr int BK_WAVES = BK_THREADS / 64;

// spread cells: workgroups per cell, slots, barrier counters
constexpr int SP_MAX_G = 64;          // workgroups per cell
constexpr int SP_MAX_WG = 512;        // workgroups per launch
constexpr int SP_SLOT = 4;            // doubles a workgroup publishes per exchange
constexpr int64_t SP_MIN_SHARE = 8192;   // photons per workgroup below which spreading does not pay
constexpr long SP_SPIN_LIMIT = 1L << 22;   // polls of a waiting workgroup before it gives the cell up (seconds)

// Partial results of the workgroups (two generations, so that a workgroup may
// publish exchange k + 1 while another still reads exchange k) and one arrival
// counter per cell.  One copy per device; launches that use it are ordered by
// the host (spread_order below).
__device__ double g_sp_slots[2 * SP_MAX_WG * SP_SLOT];
__device__ unsigned g_sp_count[SP_MAX_WG];
// Set for a cell by any of its workgroups that gave up waiting: the results of that launch
// are not to be trusted for the cell, whichever workgroup wrote them, and the cell is fitted
// again by one workgroup (the retry kernels below look at this word).
__device__ unsigned g_sp_gave_up[SP_MAX_WG];
__device__ long g_sp_spin_limit = SP_SPIN_LIMIT;

__global__ void spread_reset_kernel()
{
    for (int i = threadIdx.x; i < SP_MAX_WG; i += blockDim.x) {
        g_sp_count[i] = 0u;
        g_sp_gave_up[i] = 0u;
    }
}

// What a workgroup of a spread cell needs to talk to the others.
struct Spread {
    double *slots;     // this launch's slots: [2][ncell * G][SP_SLOT]
    unsigned *count;   // this cell's arrival counter
    unsigned *gave_up; // this cell's word in g_sp_gave_up
    int64_t first;     // index of the cell's first workgroup among the launch's
    int g, G;
    double *xs;        // LDS: SP_SLOT * SP_MAX_G doubles
    int *broken;       // LDS: set when a barrier timed out
};

// The cell as the whole workgroup sees it.  Same interface as CellData.
// SPREAD: p, n describe this workgroup's share of the cell and every reduction
// ends with an exchange between the cell's workgroups.
template <bool SPREAD>
struct BlockCellT {
    const double *p;
    int64_t n;
    int tid;
    bool swap;
    double *red;       // LDS: 3 * BK_WAVES doubles
    int *redi;         // LDS: BK_WAVES ints
    Spread sp;
    mutable unsigned gen = 0;

    // Publishes NS values (the same in every thread), waits for all workgroups
    // of the cell and leaves value k of workgroup j in sp.xs[k * G + j].
    template <int NS>
    __device__ __forceinline__ void gather(const double (&v)[NS]) const
    {
        static_assert(NS <= SP_SLOT, "slot too small");
        const int G = sp.G;
        double *gen_slots = sp.slots + (size_t)(gen & 1u) * SP_MAX_WG * SP_SLOT;
        if (tid == 0) {
            double *mine = gen_slots + (sp.first + sp.g) * SP_SLOT;
#pragma unroll
            for (int k = 0; k < NS; ++k)
                __hip_atomic_store(mine + k, v[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(sp.count, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (gen + 1u) * (unsigned)G;
            if (!*sp.broken) {
                const long limit = g_sp_spin_limit;
                long spins = 0;
                while (__hip_atomic_load(sp.count, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
                    __builtin_amdgcn_s_sleep(4);
                    if (++spins > limit) {
                        // The others never came (not all workgroups resident: the GPU is shared, or
                        // busy with another launch).  Do not hang: tell the whole cell, whose other
                        // workgroups may still finish and write results that lack this one's part.
                        *sp.broken = 1;
                        __hip_atomic_store(sp.gave_up, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                }
            }
        }
        ++gen;
        __syncthreads();
        if (tid < NS * G) {
            const int k = tid / G, j = tid - k * G;
            sp.xs[tid] = __hip_atomic_load(gen_slots + (sp.first + j) * SP_SLOT + k, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
    }

    template <int NS>
    __device__ __forceinline__ void spread_sum(double (&acc)[NS]) const
    {
        gather<NS>(acc);
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            double s = 0.0;
            for (int j = 0; j < sp.G; ++j) s += sp.xs[k * sp.G + j];
            acc[k] = s;
        }
    }

    // product of the workgroups' mantissas in xs[m0 * G ..] and sum of their
    // exponents in xs[e0 * G ..]
    __device__ __forceinline__ void spread_prod_from(int m0, int e0, double &pr, int &ex) const
    {
        double m = 1.0;
        int e = 0;
        for (int j = 0; j < sp.G; ++j) {       // at most 64 factors in [0.5, 1)
            m *= sp.xs[m0 * sp.G + j];
            e += (int)sp.xs[e0 * sp.G + j];
        }
        ex = e + __builtin_amdgcn_frexp_exp(m);
        pr = __builtin_amdgcn_frexp_mant(m);
    }

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
        if constexpr (SPREAD) spread_sum<NS>(acc);
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
        if constexpr (SPREAD) {
            const double v[3] = {pr, (double)ex, gs[0]};
            gather<3>(v);
            spread_prod_from(0, 1, pr, ex);
            double sg = 0.0;
            for (int j = 0; j < sp.G; ++j) sg += sp.xs[2 * sp.G + j];
            gs[0] = sg;
        }
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
        if constexpr (SPREAD) {
            const double v[4] = {p1, (double)e1, p2, (double)e2};
            gather<4>(v);
            spread_prod_from(0, 1, p1, e1);
            spread_prod_from(2, 3, p2, e2);
        }
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
        if constexpr (SPREAD) {
            const double v[1] = {r};
            gather<1>(v);
            r = -INFINITY;
            for (int j = 0; j < sp.G; ++j) r = fmax(r, sp.xs[j]);
        }
        return r;
    }
};

using BlockCell = BlockCellT<false>;
using SpreadCell = BlockCellT<true>;

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
// retry: fit only the cells whose spread launch gave up (g_sp_gave_up) and mark them
__global__ __launch_bounds__(BK_THREADS) void cell_fit_block_kernel(CellArgs p, const double *guess, double guess0,
                                                                    const double *beta, double beta0,
                                                                    double *alpha, double *ts, int32_t *flags,
                                                                    int retry)
{
    __shared__ double s_red[3 * BK_WAVES];
    __shared__ int s_redi[BK_WAVES];
    const int64_t cell = blockIdx.x;
    if (retry && g_sp_gave_up[cell] == 0u) return;
    const BlockSetup b = block_setup(p, cell);
    BlockCell cd{p.we + b.lo, b.n, (int)threadIdx.x, p.swap != 0, s_red, s_redi, {}};
    int fl = 0;
    double a = qnan(), t = qnan();
    if (b.bad) {
        fl = GODOT_F_FAILED | GODOT_F_BAD_RANGE;
    } else {
        a = fit_alpha(cd, b.S, guess ? guess[cell] : guess0, beta ? beta[cell] : beta0, fl);
        if (ts && !(fl & GODOT_F_FAILED)) t = test_statistic(cd, b.S, a);
    }
    if (retry) fl |= GODOT_F_RETRIED;
    if (threadIdx.x == 0) {
        alpha[cell] = a;
        if (ts) ts[cell] = t;
        if (flags) flags[cell] = fit_flags(fl, a);
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
                                                                       int32_t *__restrict__ flags, int retry)
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
    if (retry && g_sp_gave_up[cell] == 0u) return;
    const bool swap = p.swap != 0;
    const BlockSetup b = block_setup(p, cell);
    BlockCell cd{p.we + b.lo, b.n, tid, swap, s_red, s_redi, {}};
    int fl = retry ? GODOT_F_RETRIED : 0;
    bool ok = !b.bad;
    double aopt = qnan();
    if (b.bad) {
        fl = GODOT_F_FAILED | GODOT_F_BAD_RANGE;
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

// ---------------------------------------------------------------------------
// Spread cells: G workgroups per cell
// ---------------------------------------------------------------------------

struct SpreadSetup {
    BlockSetup b;
    int64_t j0, nshare;
    Spread sp;
};

__device__ __forceinline__ SpreadSetup spread_setup(const CellArgs &p, int G, double *xs, int *broken)
{
    SpreadSetup s;
    const int64_t cell = blockIdx.x / G;
    const int g = (int)(blockIdx.x - cell * G);
    s.b = block_setup(p, cell);
    const int64_t share = (s.b.n + G - 1) / G;
    s.j0 = share * g < s.b.n ? share * g : s.b.n;
    s.nshare = s.j0 + share < s.b.n ? share : s.b.n - s.j0;
    s.sp.slots = g_sp_slots;
    s.sp.count = g_sp_count + cell;
    s.sp.gave_up = g_sp_gave_up + cell;
    s.sp.first = cell * G;
    s.sp.g = g;
    s.sp.G = G;
    s.sp.xs = xs;
    s.sp.broken = broken;
    if (threadIdx.x == 0) *broken = 0;
    __syncthreads();
    return s;
}

// get_max + get_ts (core.py:632-739, 982-994), G workgroups per cell
__global__ __launch_bounds__(BK_THREADS) void cell_fit_spread_kernel(CellArgs p, int G, const double *guess,
                                                                     double guess0, const double *beta,
                                                                     double beta0, double *alpha, double *ts,
                                                                     int32_t *flags)
{
    __shared__ double s_red[3 * BK_WAVES];
    __shared__ int s_redi[BK_WAVES];
    __shared__ double s_xs[SP_SLOT * SP_MAX_G];
    __shared__ int s_broken;
    const SpreadSetup s = spread_setup(p, G, s_xs, &s_broken);
    const int64_t cell = blockIdx.x / G;
    SpreadCell cd{p.we + s.b.lo + s.j0, s.nshare, (int)threadIdx.x, p.swap != 0, s_red, s_redi, s.sp};
    int fl = 0;
    double a = qnan(), t = qnan();
    if (s.b.bad) {
        fl = GODOT_F_FAILED | GODOT_F_BAD_RANGE;
    } else {
        a = fit_alpha(cd, s.b.S, guess ? guess[cell] : guess0, beta ? beta[cell] : beta0, fl);
        if (ts && !(fl & GODOT_F_FAILED)) t = test_statistic(cd, s.b.S, a);
    }
    __syncthreads();
    if (s_broken) {
        // this cell is fitted again by cell_fit_block_kernel, launched right behind
        fl |= GODOT_F_FAILED;
        a = t = qnan();
    }
    if (threadIdx.x == 0 && s.sp.g == 0) {
        alpha[cell] = a;
        if (ts) ts[cell] = t;
        if (flags) flags[cell] = fit_flags(fl, a);
    }
}

// get_logpdf (core.py:851-932), G workgroups per cell.  Maximum and support
// come from the shared schedule on the workgroup's share of the photons; the
// grid is split by points: a workgroup takes four points at a time and runs
// over all photons of the cell for them.
__global__ __launch_bounds__(BK_THREADS) void cell_logpdf_spread_kernel(CellArgs p, int G,
                                                                        const double *__restrict__ aopt_in,
                                                                        double dlogl, int npt, int include_zero,
                                                                        double *__restrict__ dom,
                                                                        double *__restrict__ cod,
                                                                        int32_t *__restrict__ flags)
{
    __shared__ double s_red[3 * BK_WAVES];
    __shared__ int s_redi[BK_WAVES];
    __shared__ double s_xs[SP_SLOT * SP_MAX_G];
    __shared__ int s_broken;
    const SpreadSetup s = spread_setup(p, G, s_xs, &s_broken);
    const int tid = threadIdx.x;
    const int64_t cell = blockIdx.x / G;
    const bool swap = p.swap != 0;
    SpreadCell cd{p.we + s.b.lo + s.j0, s.nshare, tid, swap, s_red, s_redi, s.sp};
    // the whole cell, for the products of the grid (no exchange: block_prod only)
    const BlockCell whole{p.we + s.b.lo, s.b.n, tid, swap, s_red, s_redi, {}};
    const double *w = p.we + s.b.lo;
    const int64_t n = s.b.n;
    const double S = s.b.S;
    int fl = 0;
    bool ok = !s.b.bad;
    double aopt = qnan();
    if (s.b.bad) {
        fl = GODOT_F_FAILED | GODOT_F_BAD_RANGE;
    } else if (aopt_in) {
        aopt = aopt_in[cell];
    } else {
        aopt = fit_alpha(cd, S, 1.0, 1.0, fl);
        if (fl & GODOT_F_FAILED) ok = false;
    }
    bool done = !ok;
    for (int round = 0; !done && round <= 50; ++round) {
        const Support sp = support_search(cd, S, aopt, dlogl, include_zero != 0);
        const double step = (npt > 1) ? (sp.amax - sp.amin) / (double)(npt - 1) : 0.0;
        double best = -INFINITY;
        int ibest = npt;
        bool anynan = false;
        for (int base = s.sp.g * BK_PPL; base < npt; base += G * BK_PPL) {
            double av[BK_PPL], sg[BK_PPL], of[BK_PPL], pr[BK_PPL];
            int ex[BK_PPL];
#pragma unroll
            for (int u = 0; u < BK_PPL; ++u) {
                const int i = base + u < npt ? base + u : npt - 1;
                av[u] = grid_point(sp.amin, sp.amax, step, i, npt);
                // alpha w + (1 - w) as 1 + (alpha - 1) w; on complemented weights alpha - (alpha - 1) w
                sg[u] = swap ? 1.0 - av[u] : av[u] - 1.0;
                of[u] = swap ? av[u] : 1.0;
                pr[u] = 1.0;
                ex[u] = 0;
            }
            // The photons stream in 16 bytes a lane, four loads of a thread in flight (a compute
            // unit holds one such workgroup: twice the bytes on their way to it that 8-byte
            // loads give; 3.3 10^6-photon cells: 1.99 -> 1.73 ms for the whole kernel, most of
            // which is the search for the support with its exchanges between the workgroups).
            const int head = (int)(((uintptr_t)w >> 3) & 1);    // w[0] on an odd 8-byte slot
            const int64_t m = n > head ? n - head : 0;          // photons from the 16-byte boundary on
            const double2 *wv = reinterpret_cast<const double2 *>(w + head);
            const int64_t n2 = m >> 1;
            auto fac = [&](int u, double x) { return fma(sg[u], x, of[u]); };
            if (tid == 0 && head && n > 0) {
#pragma unroll
                for (int u = 0; u < BK_PPL; ++u) pr[u] *= fac(u, w[0]);
            }
            if (tid == 1 && (m & 1)) {
#pragma unroll
                for (int u = 0; u < BK_PPL; ++u) pr[u] *= fac(u, w[n - 1]);
            }
            int64_t i = tid;
            for (; i + 3 * BK_THREADS < n2; i += 4 * BK_THREADS) {
                const double2 a = wv[i], b = wv[i + BK_THREADS], c = wv[i + 2 * BK_THREADS],
                              d = wv[i + 3 * BK_THREADS];
#pragma unroll
                for (int u = 0; u < BK_PPL; ++u) {
                    pr[u] *= ((fac(u, a.x) * fac(u, a.y)) * (fac(u, b.x) * fac(u, b.y))) *
                             ((fac(u, c.x) * fac(u, c.y)) * (fac(u, d.x) * fac(u, d.y)));
                    ex[u] += __builtin_amdgcn_frexp_exp(pr[u]);
                    pr[u] = __builtin_amdgcn_frexp_mant(pr[u]);
                }
            }
            for (; i < n2; i += BK_THREADS) {     // at most three pairs per thread
                const double2 a = wv[i];
#pragma unroll
                for (int u = 0; u < BK_PPL; ++u) pr[u] *= fac(u, a.x) * fac(u, a.y);
            }
#pragma unroll
            for (int u = 0; u < BK_PPL; ++u) {
                ex[u] += __builtin_amdgcn_frexp_exp(pr[u]);
                pr[u] = __builtin_amdgcn_frexp_mant(pr[u]);
                whole.block_prod(pr[u], ex[u]);
                const int ip = base + u;
                if (ip < npt) {
                    const double v = log_mant(pr[u], ex[u]) - S * av[u];
                    if (tid == 0) {
                        dom[cell * npt + ip] = av[u];
                        cod[cell * npt + ip] = v - sp.llmax;
                    }
                    if (isnan(v)) anynan = true;
                    if (v > best) {
                        best = v;
                        ibest = ip;
                    }
                }
            }
        }
        // maximum of the grid against the likelihood at the centre (core.py:923-929)
        {
            const double v[3] = {best, (double)ibest, anynan ? 1.0 : 0.0};
            cd.template gather<3>(v);
            double gb = -INFINITY;
            int cand = npt;
            bool nan = false;
            for (int j = 0; j < G; ++j) {
                const double bj = s_xs[j];
                const int kj = (int)s_xs[G + j];
                if (bj > gb || (bj == gb && kj < cand)) {
                    gb = bj;
                    cand = kj;
                }
                nan = nan || (s_xs[2 * G + j] != 0.0);
            }
            const bool agree = nan || !(fabs(gb - sp.llmax) > 0.05);
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
    __syncthreads();
    if (s_broken) {
        fl |= GODOT_F_FAILED;
        ok = false;
    }
    if (!ok) {
        // every workgroup of the cell comes here together: after the last exchange
        for (int i = s.sp.g * BK_THREADS + tid; i < npt; i += G * BK_THREADS) {
            dom[cell * npt + i] = qnan();
            cod[cell * npt + i] = qnan();
        }
    }
    if (flags && tid == 0 && s.sp.g == 0) flags[cell] = fl;
}

// Launches that use the device's slots and counters run one after the other,
// whatever streams they are on: each waits for the event the one before it
// recorded.
struct SpreadOrder {
    std::mutex mu;
    hipEvent_t last[64] = {};
    int cus[64] = {};
};
SpreadOrder g_order;

// workgroups per cell, or 0 when the cells stay with one workgroup each
int spread_width(int dev, int64_t nph, int64_t ncell)
{
    if (dev < 0 || dev >= 64) return 0;
    if (g_order.cus[dev] == 0) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = -1;
        // every compute unit must be able to hold a workgroup of each spread kernel (the waiting
        // workgroups have to be resident together): ask the runtime rather than assume it
        int fit_per_cu = 0, grid_per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&fit_per_cu, (const void *)cell_fit_spread_kernel,
                                                         BK_THREADS, 0) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&grid_per_cu, (const void *)cell_logpdf_spread_kernel,
                                                         BK_THREADS, 0) != hipSuccess ||
            fit_per_cu < 1 || grid_per_cu < 1) {
            (void)hipGetLastError();
            cus = -1;
        }
        g_order.cus[dev] = cus > 0 ? cus : -1;
    }
    if (g_order.cus[dev] <= 0) return 0;
    int64_t width = g_order.cus[dev] / ncell;     // one workgroup per compute unit at most
    const int64_t by_size = godot_shape_photons(nph) / ncell / SP_MIN_SHARE;
    if (width > by_size) width = by_size;
    if (width > SP_MAX_G) width = SP_MAX_G;
    if (width * ncell > SP_MAX_WG) width = SP_MAX_WG / ncell;
    return width >= 2 ? (int)width : 0;
}

// false: nothing was launched (the caller falls back to one workgroup per cell)
template <class Retry>
bool spread_launch(const void *kernel, int64_t nph, int64_t ncell, void **args, int *width_arg, hipStream_t st,
                   Retry retry)
{
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    std::lock_guard<std::mutex> lock(g_order.mu);
    const int G = spread_width(dev, nph, ncell);
    if (G == 0) return false;
    *width_arg = G;
    if (!g_order.last[dev] &&
        hipEventCreateWithFlags(&g_order.last[dev], hipEventDisableTiming) != hipSuccess) {
        g_order.last[dev] = nullptr;
        (void)hipGetLastError();
        return false;
    }
    if (hipStreamWaitEvent(st, g_order.last[dev], 0) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    hipLaunchKernelGGL(spread_reset_kernel, dim3(1), dim3(256), 0, st);
    // An ordinary launch: ncell * G is at most the number of compute units, each of which holds
    // such a workgroup, and no other launch that waits inside a kernel runs beside this one, so
    // every workgroup becomes resident without any of those already waiting having to leave.
    // hipLaunchCooperativeKernel gives the same residency (it only checks the grid against the
    // occupancy query, which spread_width has asked already) and a process that has used it dies
    // in an exit handler after rocprofv3 has finalised its tool.  Should the workgroups not become
    // resident together all the same (a GPU shared with another process), the waits run out, the
    // cell is marked and the retry launch below fits it.
    const hipError_t e = hipLaunchKernel(kernel, dim3((unsigned)(ncell * G)), dim3(BK_THREADS), args, 0, st);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    // cells whose workgroups gave up waiting for each other (a shared or busy GPU) are fitted
    // again, one workgroup each: a launch whose workgroups leave at once when no cell did
    retry();
    (void)hipEventRecord(g_order.last[dev], st);
    return true;
}

}   // namespace

// Test hook: how many polls a workgroup of a spread cell waits for the others before it
// gives the cell up to the retry launch (default 2^22: seconds).  0 restores the default.
extern "C" int godot_debug_spread_spin_limit(long polls)
{
    GODOT_ENTRY();
    const long v = polls > 0 ? polls : SP_SPIN_LIMIT;
    const hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(g_sp_spin_limit), &v, sizeof(v));
    if (e != hipSuccess) {
        godot_set_error("godot_debug_spread_spin_limit: %s", hipGetErrorString(e));
        return GODOT_E_LAUNCH;
    }
    return GODOT_OK;
}

namespace godot_detail {

// cells of this mean size and above get a workgroup each
bool block_cells(int64_t nph, int64_t ncell)
{
    return ncell > 0 && (double)godot_shape_photons(nph) >= 16384.0 * (double)ncell;
}

// Fits take a workgroup per cell also where the cells are well beyond the registers of a
// 64-lane team (2 048 photons) and few: the dense-cell kernel then streams each cell with one
// wavefront, a chain of loads as long as the cell, and a few thousand wavefronts leave most
// SIMDs with a single one.  10^7 photons: 4 107 per cell 0.084 ms dense, 0.168 ms by
// workgroups; 5 476: 0.096 and 0.129; 6 496: 0.101 dense; 6 971: 0.112 by workgroups;
// 9 588: 0.102 by workgroups (dense about 0.17); 13 699: 0.076 (dense about 0.24).
bool fit_block_cells(int64_t nph, int64_t ncell)
{
    return block_cells(nph, ncell) || (ncell > 0 && ncell <= 4096 && (double)godot_shape_photons(nph) >= 6500.0 * (double)ncell);
}

// Profile grids take a workgroup per cell also where the cells are only moderately large but
// few: the dense-cell kernel puts 16 cells into a workgroup, and with fewer than a thousand
// such workgroups most SIMDs hold a single wavefront and nothing hides its latencies.
bool logpdf_block_cells(int64_t nph, int64_t ncell)
{
    return block_cells(nph, ncell) || (ncell > 0 && ncell <= 4096 && (double)godot_shape_photons(nph) >= 1024.0 * (double)ncell);
}

void fit_block_launch(const double *we, int64_t nph, const int64_t *lo, const int64_t *hi, const double *S,
                      const double *B, int64_t ncell, const double *guess, double guess0, const double *beta,
                      double beta0, int swap, double *alpha, double *ts, int32_t *flags, hipStream_t st)
{
    CellArgs p{we, nph, lo, hi, S, B, ncell, swap};
    {
        int G = 0;
        void *args[] = {&p, &G, &guess, &guess0, &beta, &beta0, &alpha, &ts, &flags};
        auto retry = [&]() {
            hipLaunchKernelGGL(cell_fit_block_kernel, dim3((unsigned)ncell), dim3(BK_THREADS), 0, st, p, guess,
                               guess0, beta, beta0, alpha, ts, flags, 1);
        };
        if (spread_launch((const void *)cell_fit_spread_kernel, nph, ncell, args, &G, st, retry)) return;
    }
    hipLaunchKernelGGL(cell_fit_block_kernel, dim3((unsigned)ncell), dim3(BK_THREADS), 0, st, p, guess, guess0,
                       beta, beta0, alpha, ts, flags, 0);
}

void logpdf_block_launch(const double *we, int64_t nph, const int64_t *lo, const int64_t *hi, const double *S,
                         const double *B, int64_t ncell, const double *aopt, double dlogl, int npt,
                         int include_zero, int swap, double *dom, double *cod, int32_t *flags, hipStream_t st)
{
    CellArgs p{we, nph, lo, hi, S, B, ncell, swap};
    {
        int G = 0;
        void *args[] = {&p, &G, &aopt, &dlogl, &npt, &include_zero, &dom, &cod, &flags};
        auto retry = [&]() {
            hipLaunchKernelGGL(cell_logpdf_block_kernel, dim3((unsigned)ncell), dim3(BK_THREADS), 0, st, p, we,
                               aopt, dlogl, npt, include_zero, dom, cod, flags, 1);
        };
        if (spread_launch((const void *)cell_logpdf_spread_kernel, nph, ncell, args, &G, st, retry)) return;
    }
    hipLaunchKernelGGL(cell_logpdf_block_kernel, dim3((unsigned)ncell), dim3(BK_THREADS), 0, st, p, we, aopt,
                       dlogl, npt, include_zero, dom, cod, flags, 0);
}

}   // namespace godot_detail
