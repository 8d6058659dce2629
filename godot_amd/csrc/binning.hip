// Exposure lookup and photon -> cell binning (gfx950).
//
// HBM-bound integer / fp64 streaming work: binary searches over sorted arrays
// by one thread per query, tiled integer prefix scans (2048 elements per
// 256-thread workgroup, three launches) for stream compaction, index gathers,
// and one deliberately serial pass (the exposure running sums, see below).
#include "common.hpp"
#include <type_traits>

namespace {

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

struct Quad {
    double a, b, c, d;
};
__device__ __forceinline__ Quad operator+(const Quad &x, const Quad &y)
{
    return Quad{x.a + y.a, x.b + y.b, x.c + y.c, x.d + y.d};
}
__device__ __forceinline__ Quad zero_of(Quad *) { return Quad{0, 0, 0, 0}; }
__device__ __forceinline__ int64_t zero_of(int64_t *) { return 0; }

__device__ __forceinline__ Quad shfl_up_v(const Quad &v, int o)
{
    return Quad{__shfl_up(v.a, o, WAVE), __shfl_up(v.b, o, WAVE), __shfl_up(v.c, o, WAVE),
                __shfl_up(v.d, o, WAVE)};
}
__device__ __forceinline__ int64_t shfl_up_v(int64_t v, int o)
{
    return (int64_t)__shfl_up((long long)v, o, WAVE);
}

// Inclusive scan of one value per thread over the workgroup; returns the
// inclusive prefix and writes the workgroup total to `total`.
template <class V>
__device__ V block_scan(V v, V &total)
{
    __shared__ V wave_tot[SCAN_THREADS / WAVE];
    const int lane = threadIdx.x % WAVE, wave = threadIdx.x / WAVE;
#pragma unroll
    for (int o = 1; o < WAVE; o <<= 1) {
        V u = shfl_up_v(v, o);
        if (lane >= o) v = u + v;
    }
    if (lane == WAVE - 1) wave_tot[wave] = v;
    __syncthreads();
    V pre = zero_of((V *)nullptr);
    V tot = zero_of((V *)nullptr);
#pragma unroll
    for (int k = 0; k < SCAN_THREADS / WAVE; ++k) {
        if (k < wave) pre = pre + wave_tot[k];
        tot = tot + wave_tot[k];
    }
    __syncthreads();
    total = tot;
    return pre + v;
}

// pass 1: per-tile totals
template <class V, class Load>
__global__ __launch_bounds__(SCAN_THREADS) void tile_reduce_kernel(Load ld, int64_t n, V *partials)
{
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    V s = zero_of((V *)nullptr);
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k)
        if (base + k < n) s = s + ld(base + k);
    V tot;
    block_scan(s, tot);
    if (threadIdx.x == 0) partials[blockIdx.x] = tot;
}

// pass 2: exclusive scan of the tile totals by one workgroup
template <class V>
__global__ __launch_bounds__(SCAN_THREADS) void scan_partials_kernel(V *partials, int64_t ntiles, V *grand)
{
    V carry = zero_of((V *)nullptr);
    for (int64_t base = 0; base < ntiles; base += SCAN_THREADS) {
        int64_t i = base + threadIdx.x;
        V v = (i < ntiles) ? partials[i] : zero_of((V *)nullptr);
        V tot;
        V incl = block_scan(v, tot);
        // exclusive prefix = inclusive prefix of the previous thread
        __shared__ V incl_s[SCAN_THREADS];
        incl_s[threadIdx.x] = incl;
        __syncthreads();
        V excl = (threadIdx.x == 0) ? zero_of((V *)nullptr) : incl_s[threadIdx.x - 1];
        if (i < ntiles) partials[i] = carry + excl;
        carry = carry + tot;
        __syncthreads();
    }
    if (grand && threadIdx.x == 0) *grand = carry;
}

// pass 3: scan inside each tile, add the tile prefix, hand each inclusive
// value to `st`
template <class V, class Load, class Store>
__global__ __launch_bounds__(SCAN_THREADS) void tile_scan_kernel(Load ld, int64_t n, const V *partials,
                                                                 Store st)
{
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    V x[SCAN_ITEMS];
    V s = zero_of((V *)nullptr);
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        x[k] = (base + k < n) ? ld(base + k) : zero_of((V *)nullptr);
        s = s + x[k];
        x[k] = s;   // thread-local inclusive
    }
    V tot;
    V incl = block_scan(s, tot);
    __shared__ V incl_s[SCAN_THREADS];
    incl_s[threadIdx.x] = incl;
    __syncthreads();
    V pre = partials[blockIdx.x];
    if (threadIdx.x > 0) pre = pre + incl_s[threadIdx.x - 1];
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k)
        if (base + k < n) st(base + k, pre + x[k]);
}

// first index i in [0, n) with a[i] >= v (numpy searchsorted side='left')
__device__ __forceinline__ int64_t lower_bound(const double *a, int64_t n, double v)
{
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if (a[mid] < v) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}
// first index with a[i] > v (side='right')
__device__ __forceinline__ int64_t upper_bound(const double *a, int64_t n, double v)
{
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if (a[mid] <= v) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

struct ExposureLoad {
    const double *tstart, *tstop, *exposure, *livetime;
    double s_over_e, b_over_e;
    // per-row source- and background-scaled exposures when the caller has rescaled
    // them (apply_scalers, core.py:2282-2306); null: the plain (S/E), (B/E) multiples
    const double *sexposure = nullptr, *bexposure = nullptr;
    __device__ Quad operator()(int64_t i) const
    {
        double e = exposure[i];
        // (S/E)*exposure, (B/E)*exposure: core.py:1821-1824
        // dead time (TSTOP-TSTART)-LIVETIME: core.py:1831
        return Quad{e, sexposure ? sexposure[i] : __dmul_rn(s_over_e, e),
                    bexposure ? bexposure[i] : __dmul_rn(b_over_e, e),
                    __dsub_rn(__dsub_rn(tstop[i], tstart[i]), livetime[i])};
    }
};
struct QuadStore {
    Quad *out;
    __device__ void operator()(int64_t i, const Quad &v) const { out[i] = v; }
};

// numpy.cumsum adds left to right, and which cells end up with an exposure of
// exactly zero (and are trimmed, core.py:2110-2122) depends on that rounding
// order.  To reproduce it bit for bit the running sums are formed serially:
// four lanes of the first wavefront (one per series) walk a tile of the four
// series in LDS carrying their running sum, sixteen elements per trip (the
// chain of dependent fp64 additions is the floor: measured about 22 cycles
// per table row), while the other three wavefronts store the tile scanned before and
// fetch the next one into the buffer it leaves.  One workgroup.  The table
// depends only on the exposure table, so callers keep it across queries.
constexpr int SEQ_TILE = 2048;
constexpr int SEQ_THREADS = 256;
constexpr int SEQ_MOVERS = SEQ_THREADS - 64;

__global__ __launch_bounds__(SEQ_THREADS) void exposure_cumsum_kernel(ExposureLoad ld, int64_t n, Quad *cum)
{
    __shared__ __align__(16) double s_x[2][4][SEQ_TILE];
    const int tid = threadIdx.x;
    const int64_t ntile = (n + SEQ_TILE - 1) / SEQ_TILE;
    // all loads of a thread's share of the tile are issued before the first is used
    auto fetch = [&](int64_t tile, int buf, int first, auto step_c) {
        constexpr int STEP = decltype(step_c)::value;
        constexpr int TRIPS = (SEQ_TILE + STEP - 1) / STEP;
        const int64_t base = tile * SEQ_TILE;
        const int m = (int)((n - base) < SEQ_TILE ? (n - base) : SEQ_TILE);
        Quad q[TRIPS];
#pragma unroll
        for (int j = 0; j < TRIPS; ++j) {
            const int k = first + j * STEP;
            q[j] = Quad{0.0, 0.0, 0.0, 0.0};
            if (k < m) q[j] = ld(base + k);
        }
#pragma unroll
        for (int j = 0; j < TRIPS; ++j) {
            const int k = first + j * STEP;
            if (k < SEQ_TILE) {
                s_x[buf][0][k] = q[j].a;
                s_x[buf][1][k] = q[j].b;
                s_x[buf][2][k] = q[j].c;
                s_x[buf][3][k] = q[j].d;
            }
        }
    };
    fetch(0, 0, tid, std::integral_constant<int, SEQ_THREADS>{});
    __syncthreads();
    double carry = 0.0;   // lanes 0..3 of wave 0: running sum of series `lane`
    for (int64_t t = 0; t < ntile; ++t) {
        const int buf = (int)(t & 1);
        if (tid < 64) {
            if (tid < 4) {
                double *row = s_x[buf][tid];
                double s = carry;
                // two register batches: the LDS reads of one are in flight while the
                // other goes through the chain of additions
                double a[16], b[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) a[u] = row[u];
                for (int k = 0; k < SEQ_TILE; k += 32) {
#pragma unroll
                    for (int u = 0; u < 16; ++u) b[u] = row[k + 16 + u];
#pragma unroll
                    for (int u = 0; u < 16; ++u) {
                        s = __dadd_rn(s, a[u]);
                        a[u] = s;
                    }
#pragma unroll
                    for (int u = 0; u < 16; ++u) row[k + u] = a[u];
                    if (k + 32 < SEQ_TILE) {
#pragma unroll
                        for (int u = 0; u < 16; ++u) a[u] = row[k + 32 + u];
                    }
#pragma unroll
                    for (int u = 0; u < 16; ++u) {
                        s = __dadd_rn(s, b[u]);
                        b[u] = s;
                    }
#pragma unroll
                    for (int u = 0; u < 16; ++u) row[k + 16 + u] = b[u];
                }
                carry = s;
            }
        } else {
            // the other buffer: out with tile t-1, in with tile t+1
            const int ob = buf ^ 1, mt = tid - 64;
            if (t > 0) {
                const int64_t base = (t - 1) * SEQ_TILE;
                const int m = (int)((n - base) < SEQ_TILE ? (n - base) : SEQ_TILE);
                for (int k = mt; k < m; k += SEQ_MOVERS)
                    cum[base + k] = Quad{s_x[ob][0][k], s_x[ob][1][k], s_x[ob][2][k], s_x[ob][3][k]};
            }
            if (t + 1 < ntile) fetch(t + 1, ob, mt, std::integral_constant<int, SEQ_MOVERS>{});
        }
        __syncthreads();
    }
    // the last tile scanned
    {
        const int ob = (int)((ntile - 1) & 1);
        const int64_t base = (ntile - 1) * SEQ_TILE;
        const int m = (int)((n - base) < SEQ_TILE ? (n - base) : SEQ_TILE);
        for (int k = tid; k < m; k += SEQ_THREADS)
            cum[base + k] = Quad{s_x[ob][0][k], s_x[ob][1][k], s_x[ob][2][k], s_x[ob][3][k]};
    }
}

// cumulative exposures at the query times: core.py:1806-1832
__global__ void exposure_query_kernel(ExposureLoad ld, const Quad *cum, int64_t n_ft2, const double *times,
                                      int64_t n_t, Quad *at)
{
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_t) return;
    double t = times[k];
    int64_t idx = lower_bound(ld.tstop, n_ft2, t);
    if (idx >= n_ft2) idx = n_ft2 - 1;   // the reference would index out of range here
    double t0 = ld.tstart[idx], t1 = ld.tstop[idx];
    double frac = __ddiv_rn(__dsub_rn(t, t0), __dsub_rn(t1, t0));
    frac = frac < 0 ? 0.0 : (frac > 1 ? 1.0 : frac);   // np.clip keeps NaN
    frac = 1 - frac;
    Quad x = ld(idx), c = cum[idx];
    at[k] = Quad{__dsub_rn(c.a, __dmul_rn(x.a, frac)), __dsub_rn(c.b, __dmul_rn(x.b, frac)),
                 __dsub_rn(c.c, __dmul_rn(x.c, frac)), __dsub_rn(c.d, __dmul_rn(x.d, frac))};
}

__global__ void exposure_out_kernel(const Quad *at, int64_t n_out, int cumulative, double *o_exp,
                                    double *o_sexp, double *o_bexp, double *o_dead)
{
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_out) return;
    Quad v = at[k];
    if (!cumulative) {
        Quad u = at[k + 1];
        v = Quad{u.a - v.a, u.b - v.b, u.c - v.c, u.d - v.d};
    }
    o_exp[k] = v.a;
    o_sexp[k] = v.b;
    o_bexp[k] = v.c;
    if (o_dead) o_dead[k] = v.d;
}

// photon range of every cell (see godot_bin_photons in the header)
__global__ void bin_photons_kernel(const double *ti, int64_t nph, double start0, const double *stops,
                                   int64_t ncell, int64_t *lo, int64_t *hi, int64_t *counts)
{
    int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncell) return;
    int64_t l = (c == 0) ? lower_bound(ti, nph, start0) : upper_bound(ti, nph, stops[c - 1]);
    int64_t h = (c == ncell - 1) ? lower_bound(ti, nph, stops[c]) : upper_bound(ti, nph, stops[c]);
    if (h < l) h = l;   // start0 beyond the first stop: empty first cell
    lo[c] = l;
    hi[c] = h;
    if (counts) counts[c] = h - l;
}

struct MaskLoad {
    const double *exp;
    double thr;
    __device__ int64_t operator()(int64_t i) const { return exp[i] > thr ? 1 : 0; }
};
struct SelectStore {
    const double *exp;
    double thr;
    int64_t *keep;
    __device__ void operator()(int64_t i, int64_t incl) const
    {
        if (exp[i] > thr) keep[incl - 1] = i;
    }
};

template <class T>
__global__ void gather_kernel(const T *src, const int64_t *idx, int64_t n, T *dst)
{
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) dst[k] = src[idx[k]];
}

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// numpy.searchsorted(a, v, side): out[k] = first index with a[i] >= v[k] (left) or > v[k] (right)
__global__ void search_sorted_kernel(const double *a, int64_t n, const double *v, int64_t nv, int right,
                                     int64_t *out)
{
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nv) return;
    out[k] = right ? upper_bound(a, n, v[k]) : lower_bound(a, n, v[k]);
}

// scales[searchsorted(stops, t)]: the value of a piecewise-constant function at event
// times (PiecewiseScaler.__call__ with one argument, scalers.py:46-51)
__global__ void piecewise_lookup_kernel(const double *stops, const double *scales, int64_t nseg,
                                        const double *t, int64_t n, double *out)
{
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    int64_t i = lower_bound(stops, nseg, t[k]);
    out[k] = scales[i < nseg ? i : nseg - 1];
}

// w' = a w / (a w + b (1 - w))  (apply_scalers, core.py:2304); a, b per photon or one value
__global__ void rescale_weights_kernel(const double *w, const double *a, double a0, const double *b,
                                       double b0, int64_t n, double *out)
{
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const double wk = w[k], ak = a ? a[k] : a0, bk = b ? b[k] : b0;
    const double num = __dmul_rn(ak, wk);
    out[k] = num / __dadd_rn(num, __dmul_rn(bk, __dsub_rn(1.0, wk)));
}

int exposure_table_impl(const ExposureLoad &ld, int64_t n_ft2, void *table, size_t table_bytes,
                        godot_stream_t stream, const char *fn)
{
    GODOT_REQUIRE(n_ft2 > 0, GODOT_E_INVALID, "%s: n_ft2 must be positive", fn);
    GODOT_REQUIRE(ld.tstart && ld.tstop && ld.exposure && ld.livetime, GODOT_E_INVALID,
                  "%s: null table pointer", fn);
    GODOT_REQUIRE(table && table_bytes >= align256(sizeof(Quad) * (size_t)n_ft2), GODOT_E_WORKSPACE,
                  "%s: table smaller than godot_exposure_table_bytes(n_ft2)", fn);
    hipLaunchKernelGGL(exposure_cumsum_kernel, dim3(1), dim3(SEQ_THREADS), 0, (hipStream_t)stream, ld, n_ft2,
                       (Quad *)table);
    GODOT_LAUNCH_CHECK(fn);
    return GODOT_OK;
}

int exposure_at_impl(const ExposureLoad &ld, int64_t n_ft2, const void *table, const double *times, int64_t n_t,
                     int cumulative, double *out_exp, double *out_sexp, double *out_bexp, double *out_dead,
                     void *workspace, size_t workspace_bytes, godot_stream_t stream, const char *fn)
{
    GODOT_REQUIRE(n_ft2 > 0 && n_t >= 0, GODOT_E_INVALID, "%s: bad sizes", fn);
    GODOT_REQUIRE(ld.tstart && ld.tstop && ld.exposure && ld.livetime && table, GODOT_E_INVALID,
                  "%s: null table pointer", fn);
    if (n_t == 0) return GODOT_OK;
    GODOT_REQUIRE(times && out_exp && out_sexp && out_bexp, GODOT_E_INVALID, "%s: null query/output pointer", fn);
    GODOT_REQUIRE(workspace && workspace_bytes >= align256(sizeof(Quad) * (size_t)n_t), GODOT_E_WORKSPACE,
                  "%s: workspace smaller than godot_exposure_workspace_bytes(n_t)", fn);
    hipStream_t st = (hipStream_t)stream;
    Quad *at = (Quad *)workspace;
    hipLaunchKernelGGL(exposure_query_kernel, dim3((unsigned)ceil_div(n_t, 256)), dim3(256), 0, st, ld,
                       (const Quad *)table, n_ft2, times, n_t, at);
    int64_t n_out = cumulative ? n_t : n_t - 1;
    if (n_out > 0)
        hipLaunchKernelGGL(exposure_out_kernel, dim3((unsigned)ceil_div(n_out, 256)), dim3(256), 0, st,
                           (const Quad *)at, n_out, cumulative, out_exp, out_sexp, out_bexp, out_dead);
    GODOT_LAUNCH_CHECK(fn);
    return GODOT_OK;
}

}   // namespace

extern "C" size_t godot_exposure_table_bytes(int64_t n_ft2)
{
    return n_ft2 < 0 ? 0 : align256(sizeof(Quad) * (size_t)n_ft2);
}

extern "C" int godot_exposure_table(const double *tstart, const double *tstop, const double *exposure,
                                    const double *livetime, int64_t n_ft2, double s_over_e,
                                    double b_over_e, void *table, size_t table_bytes,
                                    godot_stream_t stream)
{
    ExposureLoad ld{tstart, tstop, exposure, livetime, s_over_e, b_over_e};
    return exposure_table_impl(ld, n_ft2, table, table_bytes, stream, "godot_exposure_table");
}

extern "C" int godot_exposure_table_rows(const double *tstart, const double *tstop, const double *exposure,
                                         const double *sexposure, const double *bexposure,
                                         const double *livetime, int64_t n_ft2, void *table,
                                         size_t table_bytes, godot_stream_t stream)
{
    GODOT_REQUIRE(sexposure && bexposure, GODOT_E_INVALID, "godot_exposure_table_rows: null scaled exposure");
    ExposureLoad ld{tstart, tstop, exposure, livetime, 0.0, 0.0, sexposure, bexposure};
    return exposure_table_impl(ld, n_ft2, table, table_bytes, stream, "godot_exposure_table_rows");
}

extern "C" size_t godot_exposure_workspace_bytes(int64_t n_t)
{
    return n_t < 0 ? 0 : align256(sizeof(Quad) * (size_t)n_t);
}

extern "C" int godot_exposure_at(const double *tstart, const double *tstop, const double *exposure,
                                 const double *livetime, int64_t n_ft2, double s_over_e, double b_over_e,
                                 const void *table, const double *times, int64_t n_t, int cumulative,
                                 double *out_exp, double *out_sexp, double *out_bexp, double *out_dead,
                                 void *workspace, size_t workspace_bytes, godot_stream_t stream)
{
    ExposureLoad ld{tstart, tstop, exposure, livetime, s_over_e, b_over_e};
    return exposure_at_impl(ld, n_ft2, table, times, n_t, cumulative, out_exp, out_sexp, out_bexp, out_dead,
                            workspace, workspace_bytes, stream, "godot_exposure_at");
}

extern "C" int godot_exposure_at_rows(const double *tstart, const double *tstop, const double *exposure,
                                      const double *sexposure, const double *bexposure,
                                      const double *livetime, int64_t n_ft2, const void *table,
                                      const double *times, int64_t n_t, int cumulative, double *out_exp,
                                      double *out_sexp, double *out_bexp, double *out_dead, void *workspace,
                                      size_t workspace_bytes, godot_stream_t stream)
{
    GODOT_REQUIRE(sexposure && bexposure, GODOT_E_INVALID, "godot_exposure_at_rows: null scaled exposure");
    ExposureLoad ld{tstart, tstop, exposure, livetime, 0.0, 0.0, sexposure, bexposure};
    return exposure_at_impl(ld, n_ft2, table, times, n_t, cumulative, out_exp, out_sexp, out_bexp, out_dead,
                            workspace, workspace_bytes, stream, "godot_exposure_at_rows");
}

extern "C" int godot_search_sorted(const double *sorted, int64_t n, const double *values, int64_t nv,
                                   int right, int64_t *out, godot_stream_t stream)
{
    GODOT_REQUIRE(n >= 0 && nv >= 0, GODOT_E_INVALID, "godot_search_sorted: negative size");
    if (nv == 0) return GODOT_OK;
    GODOT_REQUIRE((sorted || n == 0) && values && out, GODOT_E_INVALID, "godot_search_sorted: null argument");
    hipLaunchKernelGGL(search_sorted_kernel, dim3((unsigned)ceil_div(nv, 256)), dim3(256), 0,
                       (hipStream_t)stream, sorted, n, values, nv, right, out);
    GODOT_LAUNCH_CHECK("godot_search_sorted");
    return GODOT_OK;
}

extern "C" int godot_piecewise_lookup(const double *stops, const double *scales, int64_t nseg,
                                      const double *t, int64_t n, double *out, godot_stream_t stream)
{
    GODOT_REQUIRE(nseg > 0 && n >= 0, GODOT_E_INVALID, "godot_piecewise_lookup: bad sizes");
    if (n == 0) return GODOT_OK;
    GODOT_REQUIRE(stops && scales && t && out, GODOT_E_INVALID, "godot_piecewise_lookup: null argument");
    hipLaunchKernelGGL(piecewise_lookup_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0,
                       (hipStream_t)stream, stops, scales, nseg, t, n, out);
    GODOT_LAUNCH_CHECK("godot_piecewise_lookup");
    return GODOT_OK;
}

extern "C" int godot_rescale_weights(const double *w, const double *a, double a0, const double *b, double b0,
                                     int64_t n, double *out, godot_stream_t stream)
{
    GODOT_REQUIRE(n >= 0, GODOT_E_INVALID, "godot_rescale_weights: negative size");
    if (n == 0) return GODOT_OK;
    GODOT_REQUIRE(w && out, GODOT_E_INVALID, "godot_rescale_weights: null argument");
    hipLaunchKernelGGL(rescale_weights_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0,
                       (hipStream_t)stream, w, a, a0, b, b0, n, out);
    GODOT_LAUNCH_CHECK("godot_rescale_weights");
    return GODOT_OK;
}

extern "C" int godot_bin_photons(const double *ti, int64_t nph, double start0, const double *stops,
                                 int64_t ncell, int64_t *lo, int64_t *hi, int64_t *counts,
                                 godot_stream_t stream)
{
    GODOT_REQUIRE(nph >= 0 && ncell >= 0, GODOT_E_INVALID, "godot_bin_photons: negative size");
    if (ncell == 0) return GODOT_OK;
    GODOT_REQUIRE((ti || nph == 0) && stops && lo && hi, GODOT_E_INVALID,
                  "godot_bin_photons: null argument");
    hipLaunchKernelGGL(bin_photons_kernel, dim3((unsigned)ceil_div(ncell, 256)), dim3(256), 0,
                       (hipStream_t)stream, ti, nph, start0, stops, ncell, lo, hi, counts);
    GODOT_LAUNCH_CHECK("godot_bin_photons");
    return GODOT_OK;
}

extern "C" size_t godot_select_workspace_bytes(int64_t ncell)
{
    if (ncell < 0) return 0;
    return align256(sizeof(int64_t) * ((size_t)ceil_div(ncell, SCAN_TILE) + 1));
}

extern "C" int godot_select_cells(const double *exp, int64_t ncell, double threshold, int64_t *keep_idx,
                                  int64_t *n_keep, void *workspace, size_t workspace_bytes,
                                  godot_stream_t stream)
{
    GODOT_REQUIRE(ncell >= 0, GODOT_E_INVALID, "godot_select_cells: negative ncell");
    GODOT_REQUIRE(n_keep, GODOT_E_INVALID, "godot_select_cells: null n_keep");
    hipStream_t st = (hipStream_t)stream;
    if (ncell == 0) {
        (void)hipMemsetAsync(n_keep, 0, sizeof(int64_t), st);
        return GODOT_OK;
    }
    GODOT_REQUIRE(exp && keep_idx, GODOT_E_INVALID, "godot_select_cells: null argument");
    GODOT_REQUIRE(workspace && workspace_bytes >= godot_select_workspace_bytes(ncell), GODOT_E_WORKSPACE,
                  "godot_select_cells: workspace too small");
    int64_t ntiles = ceil_div(ncell, SCAN_TILE);
    int64_t *part = (int64_t *)workspace;
    MaskLoad ld{exp, threshold};
    hipLaunchKernelGGL((tile_reduce_kernel<int64_t, MaskLoad>), dim3((unsigned)ntiles), dim3(SCAN_THREADS), 0,
                       st, ld, ncell, part);
    hipLaunchKernelGGL((scan_partials_kernel<int64_t>), dim3(1), dim3(SCAN_THREADS), 0, st, part, ntiles,
                       n_keep);
    hipLaunchKernelGGL((tile_scan_kernel<int64_t, MaskLoad, SelectStore>), dim3((unsigned)ntiles),
                       dim3(SCAN_THREADS), 0, st, ld, ncell, (const int64_t *)part,
                       SelectStore{exp, threshold, keep_idx});
    GODOT_LAUNCH_CHECK("godot_select_cells");
    return GODOT_OK;
}

extern "C" int godot_gather_f64(const double *src, const int64_t *idx, int64_t n, double *dst,
                                godot_stream_t stream)
{
    GODOT_REQUIRE(n >= 0, GODOT_E_INVALID, "godot_gather_f64: negative n");
    if (n == 0) return GODOT_OK;
    GODOT_REQUIRE(src && idx && dst, GODOT_E_INVALID, "godot_gather_f64: null argument");
    hipLaunchKernelGGL(gather_kernel<double>, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0,
                       (hipStream_t)stream, src, idx, n, dst);
    GODOT_LAUNCH_CHECK("godot_gather_f64");
    return GODOT_OK;
}

extern "C" int godot_gather_i64(const int64_t *src, const int64_t *idx, int64_t n, int64_t *dst,
                                godot_stream_t stream)
{
    GODOT_REQUIRE(n >= 0, GODOT_E_INVALID, "godot_gather_i64: negative n");
    if (n == 0) return GODOT_OK;
    GODOT_REQUIRE(src && idx && dst, GODOT_E_INVALID, "godot_gather_i64: null argument");
    hipLaunchKernelGGL(gather_kernel<int64_t>, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0,
                       (hipStream_t)stream, src, idx, n, dst);
    GODOT_LAUNCH_CHECK("godot_gather_i64");
    return GODOT_OK;
}
