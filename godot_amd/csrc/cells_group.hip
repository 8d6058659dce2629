// Fit kernel for sparse cells (a few to a few tens of photons per cell): the
// shape the ten-minute light curves of the reference produce.
//
// A group of L adjacent lanes (L = 1 ... 64) owns one cell and keeps its
// weights in R registers per lane for the whole fixed schedule of
// CellLogLikelihood.get_max (reference core.py:632-691) and for get_ts
// (core.py:982-994), so a wavefront fits 64/L cells at once and the per-cell
// scalar work (Newton updates, clamps, the one logarithm) is shared by that
// many cells per instruction.  Sums over a cell's photons are combined across
// its lanes with one or two DPP permutes.
//
// A wavefront (= one workgroup) reads the photons of its cells -- one
// contiguous run of the weights array, as binning produces -- with coalesced
// loads into LDS; lanes then pick their weights out of it with unconditional
// reads at fixed offsets.  No barrier anywhere: wavefronts on a CU drift apart
// and the loads of one overlap the arithmetic of the others.  Photons beyond
// the L*R register slots of a cell are read from the LDS copy in every pass,
// in rounds of four more slots a lane that the whole wavefront makes (as many
// as its largest cell needs), with the reciprocals of a round from one rcp.
//
// Cells whose first pass of the schedule asks for a restart (1 to 10 in 100,
// more the sparser the cells) are handed to a second kernel through the sign
// bit of their alpha; it packs them densely and runs the restarts, skipping
// the pass that only establishes the sign of the derivative at zero.  Cells
// that end in the bracketing fallback, and cells far larger than the register
// slots, are fitted one at a time by all 64 lanes.

#include "cells_common.hpp"

namespace {

// Weights of one cell spread over a group of L lanes: lane q of the group
// holds photons q, q + L, q + 2L, ... in registers, zero-padded.  A zero
// weight adds nothing to any sum of the fit (w / (t2 + a w) = 0) and
// multiplies the likelihood products by exactly 1.
template <int L, int R>
struct GroupCell {
    static constexpr int SLOTS = L * R;
    double w[R];
    const double *tp;   // photon 0 of the cell for reads beyond the registers (LDS slice or global)
    int n;              // photons in the cell
    int q;              // lane inside the group
    bool tail_swap;     // tp points to raw weights that are to be complemented
    int tg;             // photons beyond the register slots: rounds of four more slots a lane that the
                        // wavefront's largest cell needs (the same in every lane: a scalar loop count)

    __device__ __forceinline__ double tail(int j) const
    {
        double v = tp[j];
        return tail_swap ? 1.0 - v : v;
    }

    // photon j of the cell, or the zero of an empty slot
    __device__ __forceinline__ double tail0(int j) const
    {
        double v = 0.0;
        if (j < n) v = tail(j);
        return v;
    }

    // rounds of the wavefront for cells of at most nmax photons
    static __device__ __forceinline__ int tail_rounds(int nmax)
    {
        const int over = __builtin_amdgcn_readfirstlane(team_max_i32<64>(nmax)) - SLOTS;
        return over > 0 ? (over + 4 * L - 1) / (4 * L) : 0;
    }

    template <int NS, bool ZS = false, class F>
    __device__ __forceinline__ void sums(F f, double (&acc)[NS]) const
    {
#pragma unroll
        for (int k = 0; k < NS; ++k) acc[k] = 0.0;
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (ZS || r * L + q < n) f(w[r], acc);
        for (int g = 0; g < tg; ++g) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int j = SLOTS + (4 * g + k) * L + q;
                if (j < n) f(tail(j), acc);
            }
        }
#pragma unroll
        for (int k = 0; k < NS; ++k) acc[k] = team_sum<L>(acc[k]);
    }

    // acc[k] = sum over the cell's photons of what accf(w, d, 1/d, acc) adds, d = den(w):
    // the register slots four at a time with one hardware reciprocal
    template <int NS, class DEN, class ACC>
    __device__ __forceinline__ void sums_rcp(DEN den, ACC accf, double (&acc)[NS]) const
    {
        static_assert(R % 4 == 0, "register slots are taken four at a time");
#pragma unroll
        for (int k = 0; k < NS; ++k) acc[k] = 0.0;
#pragma unroll
        for (int r = 0; r < R; r += 4) {
            const double d0 = den(w[r]), d1 = den(w[r + 1]), d2 = den(w[r + 2]), d3 = den(w[r + 3]);
            double x0, x1, x2, x3;
            rcp4(d0, d1, d2, d3, x0, x1, x2, x3);
            accf(w[r], d0, x0, acc);
            accf(w[r + 1], d1, x1, acc);
            accf(w[r + 2], d2, x2, acc);
            accf(w[r + 3], d3, x3, acc);
        }
        for (int g = 0; g < tg; ++g) {
            const int j = SLOTS + 4 * g * L + q;
            const double w0 = tail0(j), w1 = tail0(j + L), w2 = tail0(j + 2 * L), w3 = tail0(j + 3 * L);
            const double d0 = den(w0), d1 = den(w1), d2 = den(w2), d3 = den(w3);
            double x0, x1, x2, x3;
            rcp4(d0, d1, d2, d3, x0, x1, x2, x3);
            accf(w0, d0, x0, acc);
            accf(w1, d1, x1, acc);
            accf(w2, d2, x2, acc);
            accf(w3, d3, x3, acc);
        }
#pragma unroll
        for (int k = 0; k < NS; ++k) acc[k] = team_sum<L>(acc[k]);
    }

    // logs = sum over the cell of log(t_i), gsum = sum of gf(w_i, 1/t_i), t_i = den(w_i):
    // one logarithm of the running product, reciprocals four at a time
    template <class DEN, class GF>
    __device__ __forceinline__ void logsum_rcp(DEN den, GF gf, double &logs, double &gsum) const
    {
        double pr = 1.0, gs = 0.0;
        int ex = 0;
        auto renorm = [&]() {
            ex += __builtin_amdgcn_frexp_exp(pr);
            pr = __builtin_amdgcn_frexp_mant(pr);
        };
#pragma unroll
        for (int r = 0; r < R; r += 4) {
            const double d0 = den(w[r]), d1 = den(w[r + 1]), d2 = den(w[r + 2]), d3 = den(w[r + 3]);
            double x0, x1, x2, x3;
            rcp4(d0, d1, d2, d3, x0, x1, x2, x3);
            gs += gf(w[r], x0);
            gs += gf(w[r + 1], x1);
            gs += gf(w[r + 2], x2);
            gs += gf(w[r + 3], x3);
            pr *= (d0 * d1) * (d2 * d3);
            if ((r & 4) != 0) renorm();
        }
        for (int g = 0; g < tg; ++g) {
            const int j = SLOTS + 4 * g * L + q;
            const double w0 = tail0(j), w1 = tail0(j + L), w2 = tail0(j + 2 * L), w3 = tail0(j + 3 * L);
            const double d0 = den(w0), d1 = den(w1), d2 = den(w2), d3 = den(w3);
            double x0, x1, x2, x3;
            rcp4(d0, d1, d2, d3, x0, x1, x2, x3);
            gs += gf(w0, x0);
            gs += gf(w1, x1);
            gs += gf(w2, x2);
            gs += gf(w3, x3);
            pr *= (d0 * d1) * (d2 * d3);
            if ((g & 1) != 0) renorm();
        }
        renorm();
        reduce_prod(pr, ex);
        gsum = team_sum<L>(gs);
        logs = log_mant(pr, ex);
    }

    // largest weight of the cell (-inf for an empty cell)
    __device__ __forceinline__ double wmax() const
    {
        double m = -INFINITY;
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (r * L + q < n) m = fmax(m, w[r]);
        for (int g = 0; g < tg; ++g) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int j = SLOTS + (4 * g + k) * L + q;
                if (j < n) m = fmax(m, tail(j));
            }
        }
        return team_max<L>(m);
    }

    // product over the cell of f(w) as a mantissa in [0.5, 1) and an exponent
    template <class F>
    __device__ __forceinline__ void prod(F f, double &m, int &e) const
    {
        double pr = 1.0;
        int ex = 0;
        auto step = [&](double wv, int k) {
            pr *= f(wv);
            if ((k & 7) == 7) {
                ex += __builtin_amdgcn_frexp_exp(pr);
                pr = __builtin_amdgcn_frexp_mant(pr);
            }
        };
#pragma unroll
        for (int r = 0; r < R; ++r) step(w[r], r);
        for (int g = 0; g < tg; ++g) {
            // an empty slot multiplies by one
            const int j = SLOTS + 4 * g * L + q;
            auto fac = [&](int jj) { return jj < n ? f(tail(jj)) : 1.0; };
            pr *= (fac(j) * fac(j + L)) * (fac(j + 2 * L) * fac(j + 3 * L));
            if ((g & 1) != 0) {
                ex += __builtin_amdgcn_frexp_exp(pr);
                pr = __builtin_amdgcn_frexp_mant(pr);
            }
        }
        ex += __builtin_amdgcn_frexp_exp(pr);
        pr = __builtin_amdgcn_frexp_mant(pr);
        reduce_prod(pr, ex);
        m = pr;
        e = ex;
    }

    // combines the lanes' partial products (each a mantissa in [0.5, 1) and an exponent)
    __device__ __forceinline__ void reduce_prod(double &pr, int &ex) const
    {
        pr = team_prod<L>(pr);   // at most four factors in [0.5, 1)
        ex = team_sum_i32<L>(ex);
        if (L > 1) {
            ex += __builtin_amdgcn_frexp_exp(pr);
            pr = __builtin_amdgcn_frexp_mant(pr);
        }
    }

    // sum over the cell of log(t1_i) - log(t2_i), f(w, t1, t2): one log of the
    // ratio of two running products (mantissas renormalised every eight factors)
    template <bool ZS = false, class F>
    __device__ __forceinline__ double logratio(F f) const
    {
        double p1 = 1.0, p2 = 1.0;
        int ex = 0;
        auto renorm = [&]() {
            ex += __builtin_amdgcn_frexp_exp(p1) - __builtin_amdgcn_frexp_exp(p2);
            p1 = __builtin_amdgcn_frexp_mant(p1);
            p2 = __builtin_amdgcn_frexp_mant(p2);
        };
        auto step = [&](double wv, int k) {
            double t1, t2;
            f(wv, t1, t2);
            p1 *= t1;
            p2 *= t2;
            if ((k & 7) == 7) renorm();
        };
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (ZS || r * L + q < n) step(w[r], r);
        for (int g = 0; g < tg; ++g) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int j = SLOTS + (4 * g + k) * L + q;
                if (j < n) step(tail(j), 4 * g + k + 3);   // renormalised after the fourth of a round
            }
        }
        renorm();
        p1 = team_prod<L>(p1);   // at most four factors in [0.5, 1)
        p2 = team_prod<L>(p2);
        ex = team_sum_i32<L>(ex);
        return log_scaled(p1 * rcp_fast(p2), ex);
    }
};

// a 64-bit value of one lane as a wave-uniform (scalar) value
__device__ __forceinline__ int64_t readlane_i64(int64_t v, int l)
{
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)v, l);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(v >> 32), l);
    return (int64_t)(((unsigned long long)hi << 32) | lo);
}

// get_ts (core.py:982-994) when log L(0) = sum log(1 - w_i) is known already:
// one product of alpha w + (1 - w) = 1 + (alpha - 1) w and one logarithm.
template <class GC>
__device__ __forceinline__ double test_statistic_l0(const GC &gc, double S, double aopt, double azb,
                                                    const LogProd &l0)
{
    if (S == 0 || aopt == 0) return 0.0;
    double m;
    int e;
    gc.prod([=](double w) { return fma(azb, w, 1.0); }, m, e);
    const double dl = log_scaled(m * rcp_fast(l0.m), e - l0.e);
    return 2 * (dl - S * aopt);
}

// every lane of a group reads its photons straight from global memory
template <int L, int R>
__device__ __forceinline__ void gather_cell(GroupCell<L, R> &gc, const double *gp, int n, int q, bool swap)
{
    gc.tp = gp;
    gc.n = n;
    gc.q = q;
    gc.tail_swap = swap;
    gc.tg = GroupCell<L, R>::tail_rounds(n);
#pragma unroll
    for (int r = 0; r < R; ++r) gc.w[r] = 0.0;
    if (n > 0) {
        const int last = n - 1;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int j = r * L + q;
            double v = gp[j < last ? j : last];
            if (swap) v = 1.0 - v;
            gc.w[r] = (j < n) ? v : 0.0;
        }
    }
}

// One cell fitted by a whole wavefront that streams its weights (from LDS or
// from L2) in every pass: the schedule from restart depth `depth` on, with the
// reference's restart rule (core.py:684-689) and bracketing fallback.
template <bool B0>
__device__ __noinline__ void slow_cell_fit(const double *wp, int64_t n, bool swap, double S, double g, double b,
                                           int depth, bool want_ts, double &a_out, double &t_out, int &fl_out)
{
    using SC = StreamCell<64>;
    SC sc;
    stream_load(sc, wp, n, (int)(threadIdx.x & 63), swap);
    int fl = 0;
    double a = qnan();
    if (depth > 3) {
        // the restarts have run out already: straight to the bracketing fallback
        fl = 3 << 1;
        a = max_numerical(sc, S, fl);
    } else for (;; ++depth) {
        fl = depth << 1;
        bool at_zero, ok;
        // a restart knows that the derivative at zero is not negative
        if (depth == 0) ok = fit_attempt_impl<SC, B0, true>(sc, S, g, b, a, at_zero);
        else ok = fit_attempt_impl<SC, B0, false>(sc, S, g, b, a, at_zero);
        if (ok) {
            if (at_zero) {
                a = 0.0;
                fl |= GODOT_F_AT_ZERO;
            } else {
                a = a + 1;
            }
            break;
        }
        if (depth > 2) {
            a = max_numerical(sc, S, fl);
            break;
        }
        g = (a + 1) - 1;
        b = (b + 1) - 1;
    }
    a_out = a;
    fl_out = fl;
    t_out = qnan();
    if (want_ts && !(fl & GODOT_F_FAILED)) t_out = test_statistic(sc, S, a);
}

// The test statistic of a restarted cell, as a function of its own: inlined into the loop
// over a wavefront's restarts, the constants of its logarithm were formed once ahead of the
// loop and, with no register to spare there, pushed to scratch memory and fetched back --
// 28 MB of the 37 MB that a launch over 4.4e5 ten-minute cells (1e7 photons) wrote, where its
// results are 9 MB.
template <class CD>
__device__ __noinline__ double test_statistic_apart(CD cd, double S, double aopt)
{
    return test_statistic(cd, S, aopt);
}

template <int L, int R, bool B0, bool INPLACE>
__device__ __forceinline__ void cell_fit_group_body(const CellArgs &p, const double *guess, double guess0,
                                                    const double *beta, double beta0, double *alpha,
                                                    double *ts, int32_t *flags)
{
    constexpr int CW = 64 / L;                 // cells per wavefront
    constexpr int SLOTS = L * R;
    // photons in the wavefront's LDS slice: the run of its 64 / L cells.  Two lanes a cell:
    // 32 cells of at most R + R - 3 sigma photons on average; four lanes: 16 cells whose
    // mean may come to within two sigma of the 4 R slots, so that the run needs more room.
    // (the largest shape also takes cells half again as long as its 2 048 slots: what does not
    // fit the registers is read from the slice in every pass)
    constexpr int BOUNCE = (L == 1 && R >= 24) ? 128 * ((42 * R + 64) / 128) : (L == 64 && R == 32 ? 96 : L >= 4 ? 64 : 52) * R;
    constexpr int NLOAD = (BOUNCE + 63) / 64;
    constexpr int NWIDE = (BOUNCE + 127) / 128;   // rows of 128 photons: 16 bytes a lane
    using GC = GroupCell<L, R>;

    __shared__ __align__(16) double sb[NWIDE * 128 + SLOTS];

    const int lane = threadIdx.x;
    const int q = lane % L, ci = lane / L;
    const int64_t cell0 = (int64_t)blockIdx.x * CW;
    const int64_t cell = cell0 + ci;
    const bool have = cell < p.ncell;
    const bool swap = p.swap != 0;

    int64_t lo = 0, hi = 0;
    bool bad = false;
    double S = 0.0;
    if (have) {
        lo = p.lo[cell];
        hi = p.hi[cell];
        bad = (lo < 0) || (hi < lo) || (hi > p.nph);
        S = swap ? p.B[cell] : p.S[cell];
    }
    const int64_t n64 = (have && !bad) ? hi - lo : 0;
    const bool big = n64 > 8 * SLOTS;          // fitted by the whole wavefront at the end
    const int n = big ? 0 : (int)n64;

    // are the wavefront's cells one contiguous run of photons that fits the slice?
    // (everything wave-uniform is kept in scalar registers)
    const int64_t left = p.ncell - cell0;
    const int nvalid = left < CW ? (int)left : CW;      // >= 1 by the grid size
    const int64_t next_lo = __shfl_down((long long)lo, L, WAVE);
    bool link_ok = !bad;
    if (have && !bad && ci + 1 < nvalid) link_ok = (next_lo == hi);
    const int64_t first = readlane_i64(lo, 0);
    // the LDS copy begins on a 16-byte boundary of the weights, one photon before the run if
    // need be, so that its rows can be loaded 16 bytes a lane
    const int64_t base = first - (int64_t)((((uintptr_t)p.we >> 3) + (uintptr_t)first) & 1);
    const int64_t end = readlane_i64(hi, (nvalid - 1) * L);
    const bool lds_ok = __all(link_ok ? 1 : 0) && (end - base) <= BOUNCE;
    const int span = lds_ok ? (int)(end - base) : 0;

    GC gc;
    if (lds_ok) {
        if (span > 0) {
            if (base >= 0 && base + NWIDE * 128 <= p.nph) {
                // whole 128-photon rows, the last ones possibly past the run (never past the
                // array); rows past the run hold zeros: stored all the same, never read as photons
                const double2 *vp = reinterpret_cast<const double2 *>(p.we + base) + lane;
                double2 v[NWIDE];
#pragma unroll
                for (int u = 0; u < NWIDE; ++u) {
                    v[u] = double2{0.0, 0.0};
                    if (__builtin_expect(u * 128 < span, 1)) v[u] = vp[u * 64];
                }
                if (swap) {
#pragma unroll
                    for (int u = 0; u < NWIDE; ++u) v[u] = double2{1.0 - v[u].x, 1.0 - v[u].y};
                }
                double2 *sb2 = reinterpret_cast<double2 *>(sb);
#pragma unroll
                for (int u = 0; u < NWIDE; ++u) sb2[u * 64 + lane] = v[u];
            } else {
                // the ends of the array: 8 bytes a lane, indices kept inside it
                double v[NLOAD];
#pragma unroll
                for (int u = 0; u < NLOAD; ++u) {
                    const int k = u * 64 + lane;
                    const int64_t i = base + (k < span ? k : span - 1);
                    v[u] = 0.0;
                    if (u * 64 < span) v[u] = p.we[i < 0 ? 0 : i];
                }
                if (swap) {
#pragma unroll
                    for (int u = 0; u < NLOAD; ++u) v[u] = 1.0 - v[u];
                }
#pragma unroll
                for (int u = 0; u < NLOAD; ++u) {
                    const int k = u * 64 + lane;
                    if (u < NLOAD - 1 || k < BOUNCE) sb[k] = v[u];
                }
            }
        }
        // the slice belongs to this wavefront alone and LDS serves a wavefront's
        // accesses in order: only the compiler has to be kept from reordering
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const int off = have ? (int)(lo - base) : 0;
        const double *src = sb + off + q;
        const int nr = (n - q + L - 1) / L;    // register slots of this lane that hold a photon
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const double x = src[r * L];       // may lie beyond the cell (never beyond the slice)
            gc.w[r] = (r < nr) ? x : 0.0;
        }
        gc.tp = sb + off;
        gc.n = n;
        gc.q = q;
        gc.tail_swap = false;
        gc.tg = GC::tail_rounds(n);
    } else {
        gather_cell(gc, p.we + (bad ? 0 : lo), n, q, swap);
    }

    // the schedule, for every cell of the wavefront at once
    double g = 0.0, b = 0.0;
    bool restart = false;
    LogProd l0;
    l0.valid = false;
    double o_a = qnan(), o_t = 0.0;           // what this lane's cell puts out
    int o_fl = 0;
    bool o_full = false, o_hand = false;      // alpha + TS + flags, or alpha alone (a restart handed on)
    if (have) {
        g = (guess ? guess[cell] : guess0) - 1;
        b = B0 ? 0.0 : (beta ? beta[cell] : beta0) - 1;
        if (bad) {
            o_t = qnan();
            o_fl = GODOT_F_FAILED | GODOT_F_BAD_RANGE;
            o_full = true;
        } else if (!big) {
            double a;
            bool at_zero;
            if (fit_attempt_impl<GC, B0, true, true>(gc, S, g, b, a, at_zero, &l0)) {
                const double azb = at_zero ? -1.0 : a;
                a = at_zero ? 0.0 : a + 1;
                if (ts) o_t = l0.valid ? test_statistic_l0(gc, S, a, azb, l0) : test_statistic(gc, S, a);
                o_a = isnan(a) ? qnan() : a;      // sign bit clear: finished
                o_fl = at_zero ? GODOT_F_AT_ZERO : 0;
                o_full = true;
            } else if (INPLACE) {
                // the reference calls itself again with guess = a + 1 (core.py:684-689)
                g = (a + 1) - 1;
                b = (b + 1) - 1;
                restart = true;
            } else {
                // Restarts are left to the second kernel, which packs them densely:
                // the new guess (never negative) is handed over with its sign bit set.
                o_a = -(a + 1);
                o_hand = true;
            }
        }
    }

    // results: the first lanes of the groups hand theirs to lanes 0 .. CW-1, which
    // store whole rows (before the restarts run: what a finished cell puts out then no
    // longer occupies registers across them, and nothing of it is pushed to scratch memory)
    if (L > 1) {
        const int from = (lane * L) & 63;
        o_a = __shfl(o_a, from, WAVE);
        if (ts) o_t = __shfl(o_t, from, WAVE);
        o_fl = __shfl(o_fl | (o_full ? 0x100 : 0) | (o_hand ? 0x200 : 0), from, WAVE);
        o_full = (o_fl & 0x100) != 0;
        o_hand = (o_fl & 0x200) != 0;
        o_fl &= 0xff;
    }
    if (lane < CW && cell0 + lane < p.ncell) {
        const int64_t c = cell0 + lane;
        if (o_full || o_hand) alpha[c] = o_a;
        if (o_full) {
            if (ts) ts[c] = o_t;
            if (flags) flags[c] = fit_flags(o_fl, o_a);
        }
    }

    // Restarts in place (about 1 cell in 100 here, one or two per wavefront that has
    // any): four at a time by teams of 16 lanes, which take the cell's weights from
    // the LDS copy -- a fraction of the instructions that sending the whole wavefront
    // through the schedule again for one cell would cost.
    if (INPLACE) {
        constexpr int QT = 16;
        using TC = CellData<QT, 4>;
        const int team = lane / QT, tl = lane % QT;
        unsigned long long todo = __ballot((restart && q == 0) ? 1 : 0);
        while (todo) {
            // the team's cell: the team-th of the next four restarts
            int src = -1;
#pragma unroll
            for (int t = 0; t < 64 / QT; ++t) {
                if (todo) {
                    const int sl = __builtin_ctzll(todo);
                    todo &= todo - 1;
                    if (team == t) src = sl;
                }
            }
            const int from = src < 0 ? lane : src;
            const int64_t lo_s = __shfl((long long)lo, from, WAVE);
            const int n_s = __shfl(n, from, WAVE);
            const double S_s = __shfl(S, from, WAVE);
            double g_s = __shfl(g, from, WAVE), b_s = __shfl(b, from, WAVE);
            if (src >= 0) {
                const int64_t c = cell0 + src / L;
                TC cd;
                if (lds_ok) cd.load(sb + (int)(lo_s - base), n_s, tl, false);
                else cd.load(p.we + lo_s, n_s, tl, swap);
                int depth = 1, fl = 0;
                double a = qnan();
                for (;; ++depth) {
                    bool at_zero;
                    fl = depth << 1;
                    if (fit_attempt_impl<TC, B0, false>(cd, S_s, g_s, b_s, a, at_zero)) {
                        a = a + 1;
                        break;
                    }
                    if (depth > 2) {
                        // bracketing fallback: out of line, on a streamed view of the cell
                        StreamCell<QT> sc;
                        stream_load(sc, p.we + lo_s, (int64_t)n_s, tl, swap);
                        a = max_numerical(sc, S_s, fl);
                        break;
                    }
                    g_s = (a + 1) - 1;
                    b_s = (b_s + 1) - 1;
                }
                double t = qnan();
                if (ts && !(fl & GODOT_F_FAILED)) t = test_statistic_apart(cd, S_s, a);
                if (tl == 0) {
                    alpha[c] = isnan(a) ? qnan() : a;
                    if (ts) ts[c] = t;
                    if (flags) flags[c] = fit_flags(fl, a);
                }
            }
        }
    }

    // cells far larger than a group's registers: one at a time, all 64 lanes on the
    // cell (out of line: keeps the common path's registers free)
    unsigned long long bigs = __ballot((big && q == 0) ? 1 : 0);
    while (bigs) {
        const int src = __builtin_ctzll(bigs);
        bigs &= bigs - 1;
        const int64_t c = cell0 + src / L;
        const int64_t lo_s = readlane_i64(lo, src), hi_s = readlane_i64(hi, src);
        const double S_s = __longlong_as_double(readlane_i64(__double_as_longlong(S), src));
        const double g_s = (guess ? guess[c] : guess0) - 1;
        const double b_s = B0 ? 0.0 : (beta ? beta[c] : beta0) - 1;
        const int d_s = 0;
        // the cell's weights: the LDS copy (already complemented) when there is one
        const double *wp = lds_ok ? sb + (int)(lo_s - base) : p.we + lo_s;
        double a_s, t_s;
        int fl_s;
        slow_cell_fit<B0>(wp, hi_s - lo_s, lds_ok ? false : swap, S_s, g_s, b_s, d_s, ts != nullptr, a_s, t_s, fl_s);
        if (lane == 0) {
            alpha[c] = a_s;
            if (ts) ts[c] = t_s;
            if (flags) flags[c] = fit_flags(fl_s, a_s);
        }
    }
}

// Second kernel: the restarts.  A wavefront scans GK_SCAN cells for the sign
// bit that the first kernel left on alpha, packs the cells it finds into a
// list and fits them like the first kernel does, one group of lanes per cell,
// with the reference's restart rule (at most three restarts, then the
// bracketing fallback).  With restarts at a rate of 1 to 10 in 100 cells the
// lanes of this kernel are well filled, where repeating the schedule in place
// would run whole wavefronts for one or two cells.
constexpr int GK_SCAN = 1024;

template <int L, int R, bool B0>
__device__ __forceinline__ void cell_fit_restart_body(const CellArgs &p, const double *beta, double beta0,
                                                      double *alpha, double *ts, int32_t *flags)
{
    constexpr int CW = 64 / L;
    using GC = GroupCell<L, R>;
    __shared__ short s_list[GK_SCAN];

    const int lane = threadIdx.x;
    const int q = lane % L, ci = lane / L;
    const int64_t c0 = (int64_t)blockIdx.x * GK_SCAN;
    const bool swap = p.swap != 0;

    int count = 0;
    // every load of the scan in flight at once
    double av[GK_SCAN / 64];
#pragma unroll
    for (int k = 0; k < GK_SCAN / 64; ++k) {
        const int64_t c = c0 + k * 64 + lane;
        av[k] = c < p.ncell ? alpha[c] : 0.0;
    }
#pragma unroll
    for (int k = 0; k < GK_SCAN / 64; ++k) {
        const double v = av[k];
        const bool f = (__double_as_longlong(v) < 0) && !isnan(v);
        const unsigned long long m = __ballot(f ? 1 : 0);
        if (f) s_list[count + __popcll(m & ((1ull << lane) - 1))] = (short)(k * 64 + lane);
        count += __popcll(m);
    }
    if (count == 0) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");

    for (int e0 = 0; e0 < count; e0 += CW) {
        const int e = e0 + ci;
        const bool act = e < count;
        int64_t c = 0, l = 0;
        double S = 0.0, g = 0.0, b = 0.0;
        int n = 0;
        if (act) {
            c = c0 + s_list[e];
            l = p.lo[c];
            n = (int)(p.hi[c] - l);
            S = swap ? p.B[c] : p.S[c];
            g = (-alpha[c]) - 1;
            b = B0 ? 0.0 : (((beta ? beta[c] : beta0) - 1) + 1) - 1;
        }
        GC gc;
        gather_cell(gc, p.we + l, n, q, swap);
        bool pend = act, brent = false;
        int depth = 1;
        double a = qnan();
        while (__any(pend ? 1 : 0)) {
            if (pend) {
                bool at_zero;
                if (fit_attempt_impl<GC, B0, false>(gc, S, g, b, a, at_zero)) {
                    pend = false;
                } else if (depth > 2) {
                    brent = true;
                    pend = false;
                } else {
                    g = (a + 1) - 1;
                    b = (b + 1) - 1;
                    ++depth;
                }
            }
        }
        if (act && !brent) {
            a = a + 1;
            double t = 0.0;
            if (ts) t = test_statistic(gc, S, a);
            if (q == 0) {
                alpha[c] = isnan(a) ? qnan() : a;
                if (ts) ts[c] = t;
                if (flags) flags[c] = fit_flags(depth << 1, a);
            }
        }
        // the bracketing fallback: all 64 lanes on one cell, out of line
        unsigned long long todo = __ballot((brent && q == 0) ? 1 : 0);
        while (todo) {
            const int src = __builtin_ctzll(todo);
            todo &= todo - 1;
            const int64_t c_s = readlane_i64(c, src), l_s = readlane_i64(l, src);
            const int n_s = __builtin_amdgcn_readlane(n, src);
            const double S_s = __longlong_as_double(readlane_i64(__double_as_longlong(S), src));
            double a_s, t_s;
            int fl_s;
            slow_cell_fit<B0>(p.we + l_s, (int64_t)n_s, swap, S_s, 0.0, 0.0, 4, ts != nullptr, a_s, t_s, fl_s);
            if (lane == 0) {
                alpha[c_s] = a_s;
                if (ts) ts[c_s] = t_s;
                if (flags) flags[c_s] = fit_flags(fl_s, a_s);
            }
        }
    }
}

// register budget: five wavefronts per SIMD (twenty LDS slices per CU); the
// variant that repeats restarts in place needs the registers of four per SIMD
// to stay clear of scratch memory (measured: faster so); 32 slots a lane and
// more need those of three (their LDS slices allow ten to twelve a CU anyway)
template <int L, int R, bool INPLACE>
__global__ __launch_bounds__(64, (R >= 32 && !(L == 64 && R == 32)) ? 3 : INPLACE ? 4 : 5) void cell_fit_group_b0_kernel(CellArgs p, const double *guess,
                                                                  double guess0, const double *beta,
                                                                  double beta0, double *alpha, double *ts,
                                                                  int32_t *flags)
{
    cell_fit_group_body<L, R, true, INPLACE>(p, guess, guess0, beta, beta0, alpha, ts, flags);
}

template <int L, int R, bool INPLACE>
__global__ __launch_bounds__(64, (R >= 32 && !(L == 64 && R == 32)) ? 3 : INPLACE ? 4 : 5) void cell_fit_group_bx_kernel(CellArgs p, const double *guess,
                                                                  double guess0, const double *beta,
                                                                  double beta0, double *alpha, double *ts,
                                                                  int32_t *flags)
{
    cell_fit_group_body<L, R, false, INPLACE>(p, guess, guess0, beta, beta0, alpha, ts, flags);
}

template <int L, int R>
__global__ __launch_bounds__(64, 4) void cell_fit_restart_b0_kernel(CellArgs p, const double *beta, double beta0,
                                                                    double *alpha, double *ts, int32_t *flags)
{
    cell_fit_restart_body<L, R, true>(p, beta, beta0, alpha, ts, flags);
}

template <int L, int R>
__global__ __launch_bounds__(64, 4) void cell_fit_restart_bx_kernel(CellArgs p, const double *beta, double beta0,
                                                                    double *alpha, double *ts, int32_t *flags)
{
    cell_fit_restart_body<L, R, false>(p, beta, beta0, alpha, ts, flags);
}

// ---------------------------------------------------------------------------
// get_logpdf (core.py:851-932) for sparse cells: the maximum (when none is
// given), the adaptive support and the npt-point profile of every cell.
// A wavefront owns 64/L cells.  Maximum and support are found with one group
// of L lanes per cell, as in the fit kernel; the grid is then evaluated one
// cell at a time by all 64 lanes (the cell's weights are wave-uniform there:
// scalar loads), LP_PPL points per lane, and written straight to memory in
// rows.  The maximum of the grid is compared with the likelihood at the centre
// and the cell is re-centred and done again if they disagree (core.py:923-929).
// No LDS, no barrier.
// ---------------------------------------------------------------------------
constexpr int GP_PPL = 4;
constexpr int GP_MAX_RECENTRE = 50;

// three wavefronts per SIMD: the registers of a cell's weights stay live across the
// grid evaluation (measured 3 % faster than four with spills)
template <int L, int R>
__global__ __launch_bounds__(64, 3) void cell_logpdf_group_kernel(CellArgs p, const double *__restrict__ wts,
                                                                  const double *__restrict__ aopt_in,
                                                                  double dlogl, int npt, int include_zero,
                                                                  double *__restrict__ dom,
                                                                  double *__restrict__ cod,
                                                                  int32_t *__restrict__ flags)
{
    constexpr int CW = 64 / L;
    constexpr int SLOTS = L * R;
    using GC = GroupCell<L, R>;
    const int lane = threadIdx.x;
    const int q = lane % L, ci = lane / L;
    const int64_t cell0 = (int64_t)blockIdx.x * CW;
    const int64_t cell = cell0 + ci;
    const bool have = cell < p.ncell;
    const bool swap = p.swap != 0;

    int64_t lo = 0, hi = 0;
    bool bad = false;
    double S = 0.0;
    if (have) {
        lo = p.lo[cell];
        hi = p.hi[cell];
        bad = (lo < 0) || (hi < lo) || (hi > p.nph);
        S = swap ? p.B[cell] : p.S[cell];
    }
    const int64_t n64 = (have && !bad) ? hi - lo : 0;
    const bool big = n64 > 8 * SLOTS;          // maximum and support by the whole wavefront
    const int n = big ? 0 : (int)n64;
    GC gc;
    gather_cell(gc, p.we + (bad ? 0 : lo), n, q, swap);

    // state of this lane's cell: 0 = to do, 1 = done, 2 = absent or failed
    int state = (have && !bad) ? 0 : 2;
    int fl = (have && bad) ? (GODOT_F_FAILED | GODOT_F_BAD_RANGE) : 0;
    double aopt = qnan();
    bool numerical = false;
    if (state == 0) {
        if (aopt_in) {
            aopt = aopt_in[cell];
        } else if (!big) {
            aopt = fit_alpha(gc, S, 1.0, 1.0, fl, &numerical);
        }
    }
    // big cells, and cells whose restarts have run out (the bracketing fallback): all 64
    // lanes on one cell at a time
    if (!aopt_in) {
        unsigned long long bigs = __ballot(((big || numerical) && state == 0 && q == 0) ? 1 : 0);
        while (bigs) {
            const int src = __builtin_ctzll(bigs);
            bigs &= bigs - 1;
            const int64_t lo_s = readlane_i64(lo, src), hi_s = readlane_i64(hi, src);
            const double S_s = __longlong_as_double(readlane_i64(__double_as_longlong(S), src));
            StreamCell<64> sc;
            stream_load(sc, p.we + lo_s, hi_s - lo_s, lane, swap);
            int bfl = 0;
            const double ab = fit_alpha(sc, S_s, 1.0, 1.0, bfl);
            if (lane / L == src / L) {
                aopt = ab;
                fl = bfl;
                if (bfl & GODOT_F_FAILED) state = 2;
            }
        }
    }

    double amin = 0.0, amax = 0.0, llmax = 0.0, step = 0.0;
    for (int round = 0; round <= GP_MAX_RECENTRE; ++round) {
        // support of the cells still to do
        if (state == 0 && !big) {
            const Support sp = support_search(gc, S, aopt, dlogl, include_zero != 0);
            amin = sp.amin;
            amax = sp.amax;
            llmax = sp.llmax;
        }
        unsigned long long bigs = __ballot((big && state == 0 && q == 0) ? 1 : 0);
        while (bigs) {
            const int src = __builtin_ctzll(bigs);
            bigs &= bigs - 1;
            const int64_t lo_s = readlane_i64(lo, src), hi_s = readlane_i64(hi, src);
            const double S_s = __longlong_as_double(readlane_i64(__double_as_longlong(S), src));
            const double a_s = __longlong_as_double(readlane_i64(__double_as_longlong(aopt), src));
            StreamCell<64> sc;
            stream_load(sc, p.we + lo_s, hi_s - lo_s, lane, swap);
            const Support sp = support_search(sc, S_s, a_s, dlogl, include_zero != 0);
            if (lane / L == src / L) {
                amin = sp.amin;
                amax = sp.amax;
                llmax = sp.llmax;
            }
        }
        if (state == 0) step = (npt > 1) ? (amax - amin) / (double)(npt - 1) : 0.0;

        // grid evaluation, one cell at a time
        unsigned long long todo = __ballot((state == 0 && q == 0) ? 1 : 0);
        if (!todo) break;
        while (todo) {
            const int src = __builtin_ctzll(todo);
            todo &= todo - 1;
            const int64_t cg = cell0 + src / L;
            const int64_t lo_c = readlane_i64(lo, src);
            const int n_c = (int)(readlane_i64(hi, src) - lo_c);
            const double amin_c = __longlong_as_double(readlane_i64(__double_as_longlong(amin), src));
            const double amax_c = __longlong_as_double(readlane_i64(__double_as_longlong(amax), src));
            const double step_c = __longlong_as_double(readlane_i64(__double_as_longlong(step), src));
            const double ll_c = __longlong_as_double(readlane_i64(__double_as_longlong(llmax), src));
            const double S_c = __longlong_as_double(readlane_i64(__double_as_longlong(S), src));
            const double *w = wts + lo_c;
            double best = -INFINITY;
            int ibest = npt;
            bool anynan = false;
            for (int i0 = 0; i0 < npt; i0 += 64 * GP_PPL) {
                double av[GP_PPL], am1[GP_PPL], pr[GP_PPL];
                int ex[GP_PPL];
#pragma unroll
                for (int u = 0; u < GP_PPL; ++u) {
                    int i = i0 + u * 64 + lane;
                    i = i < npt ? i : npt - 1;
                    av[u] = grid_point(amin_c, amax_c, step_c, i, npt);
                    am1[u] = av[u] - 1.0;
                    pr[u] = 1.0;
                    ex[u] = 0;
                }
                if (!swap) {
                    // alpha w + (1 - w) as 1 + (alpha - 1) w
                    grid_products<GP_PPL>(w, n_c, p.nph - lo_c, pr, ex,
                                          [&](double wj, int u) { return fma(am1[u], wj, 1.0); });
                } else {
                    // on complemented weights: alpha (1 - w) + w = alpha - (alpha - 1) w
                    grid_products<GP_PPL>(w, n_c, p.nph - lo_c, pr, ex,
                                          [&](double wj, int u) { return fma(-am1[u], wj, av[u]); });
                }
#pragma unroll
                for (int u = 0; u < GP_PPL; ++u) {
                    const int i = i0 + u * 64 + lane;
                    if (i < npt) {
                        const double v = log_scaled(pr[u], ex[u]) - S_c * av[u];
                        dom[cg * npt + i] = av[u];
                        cod[cg * npt + i] = v - ll_c;
                        if (isnan(v)) anynan = true;
                        if (v > best) {        // ascending i per lane: keeps the first maximum
                            best = v;
                            ibest = i;
                        }
                    }
                }
            }
            // agreement of the grid maximum with the likelihood at the centre (core.py:923-929);
            // numpy's argmax returns the first NaN if there is one, and the reference's
            // comparison is then false: the grid is accepted.
            const bool nan_c = __any(anynan ? 1 : 0);
            // Rounding is monotone, so |max - llmax| > 0.05 is decided lane by lane: some
            // lane's best lies more than 0.05 above, or every lane's more than 0.05 below.
            const double dbest = best - ll_c;
            const bool far = __any(dbest > 0.05 ? 1 : 0) || __all(dbest < -0.05 ? 1 : 0);   // false on NaN
            const bool agree = nan_c || !far;
            int inew = 0;
            if (!agree) {
                const double gbest = team_max<64>(best);
                int cand = (best == gbest) ? ibest : npt;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const int other = __shfl_xor(cand, o, WAVE);
                    cand = other < cand ? other : cand;
                }
                inew = cand >= npt ? 0 : cand;
            }
            if (lane / L == src / L) {
                if (agree) {
                    state = 1;
                } else if (round == GP_MAX_RECENTRE) {
                    state = 2;
                    fl |= GODOT_F_FAILED;
                } else {
                    aopt = grid_point(amin, amax, step, inew, npt);
                    fl |= GODOT_F_RECENTRED;
                }
            }
        }
    }

    // cells without a profile: rows of NaN
    unsigned long long nans = __ballot((have && state == 2 && q == 0) ? 1 : 0);
    while (nans) {
        const int src = __builtin_ctzll(nans);
        nans &= nans - 1;
        const int64_t cg = cell0 + src / L;
        for (int i = lane; i < npt; i += 64) {
            dom[cg * npt + i] = qnan();
            cod[cg * npt + i] = qnan();
        }
    }
    if (flags && have && q == 0) flags[cell] = fl;
}

}   // namespace

namespace godot_detail {

// Shape of the lane groups (L lanes a cell, R register slots a lane) for a mean
// occupancy.  A wavefront spends about 700 vector instructions on what does not depend on
// the number of slots (ranges, staging, the steps of the schedule, the logarithm, results,
// restarts) and about 29 on every slot of a lane, whatever L is: the fewer lanes a cell
// takes, the more cells share those 700.  The fit (grid = false) therefore gives cells of up
// to 23 photons on average ONE lane of 24, 32 or 40 slots (the mean plus 3.5 sqrt(mean):
// the cells of a light curve spread wider than Poisson because the exposure of a cell varies
// along the orbit -- ten-minute cells of 22.6 photons on average have a standard deviation
// of 7, not 4.8); what a cell holds beyond its slots is taken in rounds of four more slots
// that the whole wavefront makes (GroupCell::tg).  Measured on 10^7 photons, 1 x 40 against
// 2 x 20 at 22.6 photons a cell: 1.47e7 against 1.78e7 vector instructions a launch; single
// launches 43-46 us both (ten to eleven wavefronts of twice the length a compute unit: the
// launch ends on a longer tail), four lists in one launch 36 against 39 us a list, launches
// overlapping on three streams 33 against 36 us; 1 x 32 against 2 x 12 / 2 x 16 at 12, 15
// and 17 photons a cell: 61, 55, 45 us against 69, 60, 49 us; 2 x 32 against 4 x 12 at 36
// photons a cell: 39-42 against 44-49 us.  The profile grid keeps the two-lane shapes: its
// lanes fetch their cells straight from memory, 64 runs a wavefront instead of 32, and it
// measured 0.81 against 0.68 ms with one lane a cell.  Two lanes hold the mean plus three
// sigma, larger cells take 4 to 64 lanes: measured
// on 10^7 photons these shapes fit cells of 40 to 1 400 photons in 35-45 us and form their
// profile grids in 0.32-0.52 ms, where the team-per-cell kernels of cells.hip take 40-66 us
// and 0.48-0.72 ms.  The largest shape keeps 3 072 photons in its LDS slice and reads what
// its 2 048 slots do not hold from there (daily cells of 2 738 photons: fit 94 -> 47 us,
// grid 0.70 -> 0.42 ms); for the fit alone the 64 x 32 team kernel is the faster up to its
// own 2 048 slots (grid = false leaves those cells to it).
// Returns false when no shape holds the cells.
bool fit_group_shape(int64_t nph, int64_t ncell, int &L, int &R, bool grid)
{
    const double mean = ncell > 0 ? (double)godot_shape_photons(nph) / (double)ncell : 0.0;
    const double need = mean + 2.0 * sqrt(mean), wide = mean + 3.0 * sqrt(mean);
    const double one = mean + 3.5 * sqrt(mean);
    const bool fit = !grid;
    if (need <= 9.0) { L = 1; R = 8; }
    else if (need <= 17.0) { L = 1; R = 16; }
    else if (one <= 24.0 && fit) { L = 1; R = 24; }
    else if (one <= 32.0 && fit) { L = 1; R = 32; }
    else if (one <= 40.0 && fit) { L = 1; R = 40; }
    else if (need <= 25.0) { L = 2; R = 12; }
    else if (wide <= 33.0) { L = 2; R = 16; }
    else if (wide <= (fit ? 44.0 : 41.0)) { L = 2; R = 20; }
    else if (wide <= 60.0 && fit) { L = 2; R = 32; }
    else if (need <= 49.0) { L = 4; R = 12; }
    else if (need <= 66.0) { L = 4; R = 16; }
    else if (need <= 82.0) { L = 4; R = 20; }
    else if (need <= 98.0) { L = 4; R = 24; }
    else if (need <= 114.0) { L = 4; R = 28; }
    else if (need <= 130.0) { L = 4; R = 32; }
    else if (need <= 162.0) { L = 8; R = 20; }
    else if (need <= 194.0) { L = 8; R = 24; }
    else if (need <= 258.0) { L = 16; R = 16; }
    else if (need <= 322.0) { L = 16; R = 20; }
    else if (need <= 386.0) { L = 16; R = 24; }
    else if (need <= 514.0) { L = 32; R = 16; }
    else if (need <= 642.0) { L = 32; R = 20; }
    else if (need <= 770.0) { L = 32; R = 24; }
    else if (need <= 1026.0) { L = 64; R = 16; }
    else if (need <= 1282.0) { L = 64; R = 20; }
    else if (need <= 1538.0) { L = 64; R = 24; }
    else if (need <= 3000.0 && (grid || need > 2050.0)) { L = 64; R = 32; }
    else return false;
    return true;
}

bool fit_group_launch(int L, int R, const double *we, int64_t nph, const int64_t *lo, const int64_t *hi,
                      const double *S, const double *B, int64_t ncell, const double *guess, double guess0,
                      const double *beta, double beta0, int swap, double *alpha, double *ts, int32_t *flags,
                      hipStream_t st)
{
    bool launched = false;   // a shape that fit_group_shape names and no kernel below takes must not pass in silence
    CellArgs p{we, nph, lo, hi, S, B, ncell, swap};
    const bool b0 = !beta && beta0 == 1.0;
    // Sparse cells restart often (1 in 10 at four photons per cell, 1 in 40 at
    // twelve): their restarts go to the second kernel.  From about sixteen photons
    // per cell on about 1 in 100 restarts and repeating in place is cheaper.
    const bool inplace = (double)godot_shape_photons(nph) >= 16.0 * (double)ncell;
    dim3 grid((unsigned)ceil_div(ncell, (int64_t)(64 / L))), block(64);
    dim3 grid2((unsigned)ceil_div(ncell, (int64_t)GK_SCAN));
#define GK_LAUNCH_INPLACE(L_, R_)                                                                   \
    if (L == L_ && R == R_ && inplace) {                                                            \
        launched = true;                                                                            \
        if (b0)                                                                                     \
            hipLaunchKernelGGL((cell_fit_group_b0_kernel<L_, R_, true>), grid, block, 0, st, p,     \
                               guess, guess0, beta, beta0, alpha, ts, flags);                       \
        else                                                                                        \
            hipLaunchKernelGGL((cell_fit_group_bx_kernel<L_, R_, true>), grid, block, 0, st, p,     \
                               guess, guess0, beta, beta0, alpha, ts, flags);                       \
    }
#define GK_LAUNCH_TWO(L_, R_)                                                                       \
    if (L == L_ && R == R_ && !inplace) {                                                           \
        launched = true;                                                                            \
        if (b0) {                                                                                   \
            hipLaunchKernelGGL((cell_fit_group_b0_kernel<L_, R_, false>), grid, block, 0, st, p,    \
                               guess, guess0, beta, beta0, alpha, ts, flags);                       \
            hipLaunchKernelGGL((cell_fit_restart_b0_kernel<L_, R_>), grid2, block, 0, st, p, beta,  \
                               beta0, alpha, ts, flags);                                            \
        } else {                                                                                    \
            hipLaunchKernelGGL((cell_fit_group_bx_kernel<L_, R_, false>), grid, block, 0, st, p,    \
                               guess, guess0, beta, beta0, alpha, ts, flags);                       \
            hipLaunchKernelGGL((cell_fit_restart_bx_kernel<L_, R_>), grid2, block, 0, st, p, beta,  \
                               beta0, alpha, ts, flags);                                            \
        }                                                                                           \
    }
    GK_LAUNCH_TWO(1, 8)
    GK_LAUNCH_TWO(1, 16)
    GK_LAUNCH_INPLACE(2, 20)
    GK_LAUNCH_INPLACE(1, 24)
    GK_LAUNCH_TWO(1, 24)
    GK_LAUNCH_INPLACE(1, 32)
    GK_LAUNCH_TWO(1, 32)
    GK_LAUNCH_INPLACE(1, 40)
    GK_LAUNCH_INPLACE(2, 32)
    GK_LAUNCH_INPLACE(4, 16)
    GK_LAUNCH_INPLACE(4, 20)
    GK_LAUNCH_INPLACE(4, 24)
    GK_LAUNCH_INPLACE(4, 28)
    GK_LAUNCH_INPLACE(4, 32)
    GK_LAUNCH_INPLACE(8, 20)
    GK_LAUNCH_INPLACE(8, 24)
    GK_LAUNCH_INPLACE(16, 16)
    GK_LAUNCH_INPLACE(16, 20)
    GK_LAUNCH_INPLACE(16, 24)
    GK_LAUNCH_INPLACE(32, 16)
    GK_LAUNCH_INPLACE(32, 20)
    GK_LAUNCH_INPLACE(32, 24)
    GK_LAUNCH_INPLACE(64, 16)
    GK_LAUNCH_INPLACE(64, 20)
    GK_LAUNCH_INPLACE(64, 24)
    GK_LAUNCH_INPLACE(64, 32)
#undef GK_LAUNCH_INPLACE
#undef GK_LAUNCH_TWO
    return launched;
}

bool logpdf_group_launch(int L, int R, const double *we, int64_t nph, const int64_t *lo, const int64_t *hi,
                         const double *S, const double *B, int64_t ncell, const double *aopt, double dlogl,
                         int npt, int include_zero, int swap, double *dom, double *cod, int32_t *flags,
                         hipStream_t st)
{
    bool launched = false;
    CellArgs p{we, nph, lo, hi, S, B, ncell, swap};
    dim3 grid((unsigned)ceil_div(ncell, (int64_t)(64 / L))), block(64);
#define GP_LAUNCH(L_, R_)                                                                             \
    if (L == L_ && R == R_) {                                                                         \
        launched = true;                                                                              \
        hipLaunchKernelGGL((cell_logpdf_group_kernel<L_, R_>), grid, block, 0, st, p, we, aopt, dlogl, \
                           npt, include_zero, dom, cod, flags);                                       \
    }
    GP_LAUNCH(1, 8)
    GP_LAUNCH(1, 16)
    GP_LAUNCH(2, 12)
    GP_LAUNCH(2, 16)
    GP_LAUNCH(2, 20)
    GP_LAUNCH(4, 12)
    GP_LAUNCH(4, 16)
    GP_LAUNCH(4, 20)
    GP_LAUNCH(4, 24)
    GP_LAUNCH(4, 28)
    GP_LAUNCH(4, 32)
    GP_LAUNCH(8, 20)
    GP_LAUNCH(8, 24)
    GP_LAUNCH(16, 16)
    GP_LAUNCH(16, 20)
    GP_LAUNCH(16, 24)
    GP_LAUNCH(32, 16)
    GP_LAUNCH(32, 20)
    GP_LAUNCH(32, 24)
    GP_LAUNCH(64, 16)
    GP_LAUNCH(64, 20)
    GP_LAUNCH(64, 24)
    GP_LAUNCH(64, 32)
#undef GP_LAUNCH
    return launched;
}

}   // namespace godot_detail
