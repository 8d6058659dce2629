// Bayesian Blocks on the cached likelihood profiles (gfx950).
//
// The reference (core.py:1108-1208) builds, for every cell i, the fitness of
// every block that ends with cell i: it sums the log profiles of the cells of
// the block, each interpolated (numpy.interp, -inf outside the support) onto
// a 1000-point grid over the support of cell i, and takes the maximum.  That
// is O(ncell^2 / 2) interpolations of 1000 points, sequential in the block
// start for one i but independent between the i's.
//
// Here one workgroup owns one i ("row").  Its 256 threads hold the 1000 grid
// points and their running sums in registers (four points per thread) and
// walk back over the cells; the maximum over the grid is reduced inside each
// wavefront per cell and across the four wavefronts once per 32 cells.
// Rows are scheduled longest first.  The slopes numpy.interp divides for are
// formed once per cell by a kernel of their own, with the same IEEE division,
// so the interpolated values are the reference's to the last bit (this file
// is compiled without FMA contraction: numpy multiplies, rounds, then adds).
// The dynamic programme over the rows (core.py:1146-1161) goes in blocks of 256 rows:
// the candidates below a block on all CUs, those inside it in one workgroup, 64 rows
// at a time (see bb_below_kernel / bb_within_kernel).

#include "common.hpp"

namespace {

constexpr int BB_THREADS = 256;
constexpr int BB_PPT = 4;                       // grid points per thread
constexpr int BB_MAX_GRID = BB_THREADS * BB_PPT;
constexpr int BB_RING = 32;                     // cells between two cross-wavefront reductions

// pairs[c][j] = (cod[c][j], slope) with slope = (cod[c][j+1] - cod[c][j]) / (dom[c][j+1] - dom[c][j])
// (numpy's arr_interp): the value and the slope of an interval come with one 16-byte load
__global__ void bb_slopes_kernel(const double *__restrict__ dom, const double *__restrict__ cod, int64_t ncell,
                                 int npt, double2 *__restrict__ pairs)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= ncell * npt) return;
    const int j = (int)(k % npt);
    double s = 0.0;
    if (j + 1 < npt) s = (cod[k + 1] - cod[k]) / (dom[k + 1] - dom[k]);
    pairs[k] = make_double2(cod[k], s);
}

// numpy.linspace(a0, a1, n)[i]
__device__ __forceinline__ double linspace_at(double a0, double a1, double step, int i, int n)
{
    if (i == n - 1 && n > 1) return a1;
    return (double)i * step + a0;
}

// Per-cell facts the fitness kernel branches on:
//   BB_UNIFORM  the abscissae of cell c are, to the bit, numpy.linspace of their end points
//               (what get_logpdf produces): they are computed instead of loaded;
//   BB_CLEAN    no NaN among the abscissae, no NaN and no +inf among the values: adding the
//               cell's interpolated profile to -inf leaves -inf, so a grid point that has
//               fallen out of some cell's support needs no further interpolation.
// *unclean counts the cells without BB_CLEAN (zeroed by the caller).
constexpr int BB_UNIFORM = 1, BB_CLEAN = 2;

__global__ void bb_uniform_kernel(const double *__restrict__ dom, const double *__restrict__ cod, int64_t ncell,
                                  int npt, int32_t *facts, int32_t *unclean)
{
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncell) return;
    const double *xp = dom + c * npt, *fp = cod + c * npt;
    const double x0 = xp[0], xl = xp[npt - 1];
    const double step = (xl - x0) / (double)(npt - 1);
    bool uni = true, clean = true;
    for (int j = 0; j < npt; ++j) {
        uni = uni && (xp[j] == linspace_at(x0, xl, step, j, npt));
        clean = clean && !isnan(xp[j]) && (fp[j] < INFINITY);      // false for NaN and +inf
    }
    facts[c] = (uni ? BB_UNIFORM : 0) | (clean ? BB_CLEAN : 0);
    if (!clean) atomicAdd(unclean, 1);
}

// numpy.interp(x, xp, fp, left=-inf, right=-inf) for one x; xp is a uniform
// grid (numpy.linspace), so the interval is guessed arithmetically and then
// settled by the comparisons numpy's search makes.  UNI: the abscissae are
// known to equal linspace(x0, xl, npt) exactly and are computed, not loaded.
template <bool UNI>
__device__ __forceinline__ double interp_one(double x, const double *__restrict__ xp,
                                             const double2 *__restrict__ fs, int npt, double x0, double xl,
                                             double inv_step, double step)
{
    if (isnan(x)) return x;
    if (x < x0 || x > xl) return -INFINITY;
    if (x == xl) return fs[npt - 1].x;
    int j = (int)((x - x0) * inv_step);
    j = j < 0 ? 0 : (j > npt - 2 ? npt - 2 : j);
    auto at = [&](int q) { return UNI ? linspace_at(x0, xl, step, q, npt) : xp[q]; };
    double xj = at(j), xj1 = at(j + 1);
    while (j > 0 && x < xj) {
        --j;
        xj1 = xj;
        xj = at(j);
    }
    while (j < npt - 2 && x >= xj1) {
        ++j;
        xj = xj1;
        xj1 = at(j + 1);
    }
    const double2 v = fs[j];
    const double fj = v.x;
    if (xj == x) return fj;
    const double s = v.y;
    double r = s * (x - xj) + fj;
    if (isnan(r)) {
        const double fj1 = fs[j + 1].x;
        r = s * (x - xj1) + fj1;
        if (isnan(r) && fj == fj1) r = fj;
    }
    return r;
}

struct Best {
    double v;      // maximum (NaN once a NaN was seen, as numpy.argmax treats NaN as the maximum)
    int k;         // its first grid index
};

// order of numpy.argmax: a NaN beats everything, then the larger value, then the smaller index
__device__ __forceinline__ Best best_of(Best a, Best b)
{
    const bool an = isnan(a.v), bn = isnan(b.v);
    bool take_b;
    if (an || bn) take_b = bn && (!an || b.k < a.k);
    else take_b = (b.v > a.v) || (b.v == a.v && b.k < a.k);
    return take_b ? b : a;
}

// Row i (cells 0 .. i): fit_val / fit_arg[i (i + 1) / 2 + j] = 2 * (max over the grid, alpha
// there) of the summed profiles of cells j .. i  (core.py:1108-1136 with i0 = 0, i1 = i + 1).
__global__ __launch_bounds__(BB_THREADS) void bb_fitness_kernel(const double *__restrict__ dom,
                                                                const double2 *__restrict__ pairs,
                                                                const int32_t *__restrict__ facts,
                                                                const int32_t *__restrict__ unclean,
                                                                int64_t ncell, int npt, int ngrid,
                                                                int64_t row_begin, int64_t row_end,
                                                                double *__restrict__ fit_val,
                                                                double *__restrict__ fit_arg)
{
    __shared__ double s_v[BB_RING][BB_THREADS / 64];
    __shared__ int s_k[BB_RING][BB_THREADS / 64];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int64_t i = row_end - 1 - (int64_t)blockIdx.x;    // longest rows first
    const double a0 = dom[i * npt], a1 = dom[i * npt + npt - 1];
    const double gstep = ngrid > 1 ? (a1 - a0) / (double)(ngrid - 1) : 0.0;
    double x[BB_PPT], acc[BB_PPT];
#pragma unroll
    for (int u = 0; u < BB_PPT; ++u) {
        const int k = u * BB_THREADS + tid;
        x[u] = linspace_at(a0, a1, gstep, k < ngrid ? k : ngrid - 1, ngrid);
        acc[u] = 0.0;
    }
    const int64_t obase = i * (i + 1) / 2 - row_begin * (row_begin + 1) / 2;
    const bool all_clean = *unclean == 0;
    int pending = 0;
    for (int64_t c = i; c >= 0; --c) {
        const double *xp = dom + c * npt;
        const double2 *fs = pairs + c * npt;
        const double x0 = xp[0], xl = xp[npt - 1];
        const double inv_step = (double)(npt - 1) / (xl - x0);
        const double step = (xl - x0) / (double)(npt - 1);
        const int fc = facts[c];                // the same for the whole workgroup
        const bool clean = (fc & BB_CLEAN) != 0;
        if (fc & BB_UNIFORM) {
#pragma unroll
            for (int u = 0; u < BB_PPT; ++u)
                if (!(clean && acc[u] == -INFINITY))
                    acc[u] += interp_one<true>(x[u], xp, fs, npt, x0, xl, inv_step, step);
        } else {
#pragma unroll
            for (int u = 0; u < BB_PPT; ++u)
                if (!(clean && acc[u] == -INFINITY))
                    acc[u] += interp_one<false>(x[u], xp, fs, npt, x0, xl, inv_step, step);
        }
        // numpy.argmax over the grid: this thread's points first (ascending index, so a
        // strict comparison keeps the first of equals; slots past the grid repeat its last
        // point and are left out)
        double m = acc[0];
        int ku = 0;
        bool nan = isnan(acc[0]);
#pragma unroll
        for (int u = 1; u < BB_PPT; ++u) {
            if (u * BB_THREADS + tid < ngrid) {
                nan = nan || isnan(acc[u]);
                if (acc[u] > m) {
                    m = acc[u];
                    ku = u;
                }
            }
        }
        if (tid >= ngrid) m = -INFINITY;
        Best b{m, tid < ngrid ? ku * BB_THREADS + tid : ngrid};
        if (__ballot(nan ? 1 : 0) == 0) {
            // no NaN in the wavefront: the maximum, then the lowest index that has it
            const double mw = team_fmax<64>(m);
            b.k = team_min_i32<64>(m == mw ? b.k : ngrid);
            b.v = mw;
        } else {
            // a NaN counts as the maximum (its first index): the general order, lane by lane
            b = Best{-INFINITY, ngrid};
#pragma unroll
            for (int u = 0; u < BB_PPT; ++u) {
                const int k = u * BB_THREADS + tid;
                if (k < ngrid) b = best_of(b, Best{acc[u], k});
            }
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                Best other{__shfl_xor(b.v, o, WAVE), __shfl_xor(b.k, o, WAVE)};
                b = best_of(b, other);
            }
        }
        const int slot = (int)((i - c) % BB_RING);
        if (lane == 0) {
            s_v[slot][wave] = b.v;
            s_k[slot][wave] = b.k;
        }
        ++pending;
        if (pending == BB_RING || c == 0) {
            __syncthreads();
            if (tid < pending) {
                // slot tid belongs to cell c + (pending - 1 - tid)
                Best r{s_v[tid][0], s_k[tid][0]};
#pragma unroll
                for (int w = 1; w < BB_THREADS / 64; ++w) r = best_of(r, Best{s_v[tid][w], s_k[tid][w]});
                const int64_t cc = c + (pending - 1 - tid);
                const int kk = r.k >= ngrid ? 0 : r.k;
                fit_val[obase + cc] = r.v * 2;
                fit_arg[obase + cc] = linspace_at(a0, a1, gstep, kk, ngrid) * 2;
            }
            // Every grid point at -inf after cell c (its maximum over the grid is -inf, in
            // all four wavefronts) and nothing but clean cells anywhere: all longer blocks
            // have fitness -inf at the first grid point, as numpy.argmax of all -inf gives.
            bool dead = all_clean;
#pragma unroll
            for (int w = 0; w < BB_THREADS / 64; ++w) dead = dead && (s_v[slot][w] == -INFINITY);
            __syncthreads();
            pending = 0;
            if (dead) {
                const double arg0 = linspace_at(a0, a1, gstep, 0, ngrid) * 2;
                for (int64_t cc = tid; cc < c; cc += BB_THREADS) {
                    fit_val[obase + cc] = -INFINITY;
                    fit_arg[obase + cc] = arg0;
                }
                return;
            }
        }
    }
}

// best[0] = 0; best[i+1] = max over r <= i of (fit_val[row i][r] - prior) + best[r];
// last[i] = the first r that attains it  (core.py:1146-1161).
// The recurrence is sequential in i, but row i needs best[r] only for r <= i:
// the rows are taken in blocks of BP_BLOCK.  For a block starting at i0,
//   bb_below_kernel   (one workgroup per row, all CUs): the candidates r < i0,
//                     whose best[r] are final -- the bulk of the ncell^2 / 2
//                     compare-adds and of the bytes read; its result is parked
//                     in best[i + 1] / last[i], which are not final yet;
//   bb_within_kernel  (one workgroup): the candidates i0 <= r <= i, in
//                     stretches of 64 rows: those below the stretch are
//                     reduced for all its rows at once by the sixteen
//                     wavefronts, those inside it by one wavefront row after
//                     row without a barrier.
constexpr int BP_K = 64;
constexpr int BP_BLOCK = 256;
constexpr int BP_NONE = 0x7fffffff;

__global__ __launch_bounds__(256) void bb_below_kernel(const double *__restrict__ fit_val, int64_t i0, double prior,
                                                       double *best, int64_t *last)
{
    __shared__ double s_v[4];
    __shared__ int s_k[4];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int64_t i = i0 + blockIdx.x;
    const double *row = fit_val + i * (i + 1) / 2;
    Best b{-INFINITY, BP_NONE};
    for (int64_t r = tid; r < i0; r += 256) b = best_of(b, Best{(row[r] - prior) + best[r], (int)r});
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        Best other{__shfl_xor(b.v, o, WAVE), __shfl_xor(b.k, o, WAVE)};
        b = best_of(b, other);
    }
    if (lane == 0) {
        s_v[wave] = b.v;
        s_k[wave] = b.k;
    }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; ++w) b = best_of(b, Best{s_v[w], s_k[w]});
        best[i + 1] = b.v;
        last[i] = b.k;
    }
}

// value and first lane of the maximum over a wavefront, in numpy.argmax's order (a NaN
// beats everything, ties go to the lower lane): one max-reduction and two ballots
// maximum over the 64 lanes of values that hold no NaN, without a trip through LDS: four
// DPP steps inside the rows of sixteen lanes, the four row maxima read as scalars
__device__ __forceinline__ double wave_max_no_nan(double v)
{
    v = fmax(v, dpp_f64<0xB1>(v));
    v = fmax(v, dpp_f64<0x4E>(v));
    v = fmax(v, dpp_f64<0x141>(v));
    v = fmax(v, dpp_f64<0x140>(v));
    auto row = [&](int l) {
        const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
        const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
        return __hiloint2double(hi, lo);
    };
    return fmax(fmax(row(0), row(16)), fmax(row(32), row(48)));
}

__device__ __forceinline__ Best wave_best_by_lane(double v, int base)
{
    const unsigned long long nan_lanes = __ballot(isnan(v) ? 1 : 0);
    if (nan_lanes) return Best{qnan(), base + (int)__builtin_ctzll(nan_lanes)};
    const double m = wave_max_no_nan(v);
    const unsigned long long at = __ballot(v == m ? 1 : 0);
    return Best{m, base + (int)__builtin_ctzll(at)};
}

__global__ __launch_bounds__(1024) void bb_within_kernel(const double *__restrict__ fit_val, int64_t ncell,
                                                         int64_t b0, int64_t b1, double prior, double *best,
                                                         int64_t *last)
{
    __shared__ double s_v[BP_K];
    __shared__ int s_k[BP_K];
    // (fit_val - prior) of the candidates inside the stretch: row t, candidates i0 .. i0 + t.
    // Fetched by all wavefronts before the row-after-row part, which then waits for no memory.
    __shared__ double s_tri[BP_K][BP_K + 1];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (b0 == 0 && tid == 0) best[0] = 0.0;
    __syncthreads();
    for (int64_t i0 = b0; i0 < b1; i0 += BP_K) {
        const int kmax = (int)((b1 - i0) < BP_K ? (b1 - i0) : BP_K);
        // candidates of this block below the stretch, on top of what bb_below_kernel parked
        // (a wavefront takes rows wave, wave + 16, wave + 32, wave + 48 together: their loads do
        // not depend on one another, and best[r] is read once for the four)
        {
            const double *rows[4];
            Best bs[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int t = wave + 16 * u;
                const int64_t i = i0 + (t < kmax ? t : 0);
                rows[u] = fit_val + i * (i + 1) / 2;
                bs[u] = Best{-INFINITY, BP_NONE};
                if (t < kmax) s_tri[t][lane] = lane <= t ? rows[u][i0 + lane] - prior : -INFINITY;
            }
            for (int64_t r = b0 + lane; r < i0; r += 64) {
                const double br = best[r];
                double fv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) fv[u] = rows[u][r];
#pragma unroll
                for (int u = 0; u < 4; ++u) bs[u] = best_of(bs[u], Best{(fv[u] - prior) + br, (int)r});
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int t = wave + 16 * u;
                Best b = bs[u];
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    Best other{__shfl_xor(b.v, o, WAVE), __shfl_xor(b.k, o, WAVE)};
                    b = best_of(b, other);
                }
                if (lane == 0 && t < kmax) {
                    const int64_t i = i0 + t;
                    if (b0 > 0) b = best_of(Best{best[i + 1], (int)last[i]}, b);
                    s_v[t] = b.v;
                    s_k[t] = b.k;
                }
            }
        }
        __syncthreads();
        // candidates inside the stretch: one wavefront, row after row.  Lane l holds best[i0 + l]
        // once row l - 1 has produced it; everything else a row needs was fetched above.
        if (wave == 0) {
            double mine = lane == 0 ? best[i0] : 0.0;     // best[i0 + lane], when known
            double tri = s_tri[0][lane], sv = s_v[0];
            int sk = s_k[0];
            for (int t = 0; t < kmax; ++t) {
                const int64_t i = i0 + t;
                const double v = lane <= t ? tri + mine : -INFINITY;
                const Best below{sv, sk};
                if (t + 1 < kmax) {                       // the next row's, while this one is reduced
                    tri = s_tri[t + 1][lane];
                    sv = s_v[t + 1];
                    sk = s_k[t + 1];
                }
                // lanes are in the order of r: the first lane of the maximum is its first r
                Best b = best_of(below, wave_best_by_lane(v, (int)i0));
                if (b.k == BP_NONE) b.k = 0;
                if (lane == t + 1) mine = b.v;
                if (lane == 0) {
                    best[i + 1] = b.v;
                    last[i] = b.k;
                }
            }
        }
        __syncthreads();
    }
}

}   // namespace

extern "C" size_t godot_bb_workspace_bytes(int64_t ncell, int npt)
{
    if (ncell < 0 || npt < 0) return 0;
    // (value, slope) pairs, then one word of facts per cell and the count of unclean cells
    return (size_t)ncell * (size_t)npt * sizeof(double2) + ((size_t)ncell + 1) * sizeof(int32_t);
}

extern "C" int godot_bb_fitness(const double *dom, const double *cod, int64_t ncell, int npt, int ngrid,
                                int64_t row_begin, int64_t row_end, void *workspace, double *fit_val,
                                double *fit_arg, godot_stream_t stream)
{
    GODOT_ENTRY();
    GODOT_REQUIRE(ncell >= 0, GODOT_E_INVALID, "godot_bb_fitness: negative ncell");
    GODOT_REQUIRE(npt >= 2, GODOT_E_INVALID, "godot_bb_fitness: npt must be at least 2");
    GODOT_REQUIRE(ngrid >= 2 && ngrid <= BB_MAX_GRID, GODOT_E_UNSUPPORTED,
                  "godot_bb_fitness: ngrid must be in [2, %d]", BB_MAX_GRID);
    GODOT_REQUIRE(ncell < ((int64_t)1 << 31), GODOT_E_UNSUPPORTED, "godot_bb_fitness: too many cells");
    GODOT_REQUIRE(0 <= row_begin && row_begin <= row_end && row_end <= ncell, GODOT_E_INVALID,
                  "godot_bb_fitness: rows must satisfy 0 <= row_begin <= row_end <= ncell");
    if (row_begin == row_end) return GODOT_OK;
    GODOT_REQUIRE(dom && cod && workspace && fit_val && fit_arg, GODOT_E_INVALID,
                  "godot_bb_fitness: null argument");
    hipStream_t st = (hipStream_t)stream;
    double2 *pairs = (double2 *)workspace;
    hipLaunchKernelGGL(bb_slopes_kernel, dim3((unsigned)ceil_div(ncell * npt, 256)), dim3(256), 0, st, dom, cod,
                       ncell, npt, pairs);
    int32_t *facts = (int32_t *)(pairs + (size_t)ncell * (size_t)npt);
    int32_t *unclean = facts + ncell;
    GODOT_REQUIRE(hipMemsetAsync(unclean, 0, sizeof(int32_t), st) == hipSuccess, GODOT_E_LAUNCH,
                  "godot_bb_fitness: clearing the workspace failed");
    hipLaunchKernelGGL(bb_uniform_kernel, dim3((unsigned)ceil_div(ncell, 256)), dim3(256), 0, st, dom, cod, ncell,
                       npt, facts, unclean);
    hipLaunchKernelGGL(bb_fitness_kernel, dim3((unsigned)(row_end - row_begin)), dim3(BB_THREADS), 0, st, dom,
                       (const double2 *)pairs, (const int32_t *)facts, (const int32_t *)unclean, ncell, npt, ngrid,
                       row_begin, row_end, fit_val, fit_arg);
    GODOT_LAUNCH_CHECK("godot_bb_fitness");
    return GODOT_OK;
}

extern "C" int godot_bb_partition(const double *fit_val, int64_t ncell, double prior, double *best,
                                  int64_t *last, godot_stream_t stream)
{
    GODOT_ENTRY();
    GODOT_REQUIRE(ncell >= 0, GODOT_E_INVALID, "godot_bb_partition: negative ncell");
    GODOT_REQUIRE(ncell < ((int64_t)1 << 31), GODOT_E_UNSUPPORTED, "godot_bb_partition: too many cells");
    if (ncell == 0) return GODOT_OK;
    GODOT_REQUIRE(fit_val && best && last, GODOT_E_INVALID, "godot_bb_partition: null argument");
    hipStream_t st = (hipStream_t)stream;
    for (int64_t b0 = 0; b0 < ncell; b0 += BP_BLOCK) {
        const int64_t b1 = b0 + BP_BLOCK < ncell ? b0 + BP_BLOCK : ncell;
        if (b0 > 0)
            hipLaunchKernelGGL(bb_below_kernel, dim3((unsigned)(b1 - b0)), dim3(256), 0, st, fit_val, b0, prior,
                               best, last);
        hipLaunchKernelGGL(bb_within_kernel, dim3(1), dim3(1024), 0, st, fit_val, ncell, b0, b1, prior, best,
                           last);
    }
    GODOT_LAUNCH_CHECK("godot_bb_partition");
    return GODOT_OK;
}
