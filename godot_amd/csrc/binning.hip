// Exposure lookup and photon -> cell binning (gfx950).
//
// HBM-bound integer / fp64 streaming work: binary searches over sorted arrays
// by one thread per query, tiled integer prefix scans (2048 elements per
// 256-thread workgroup, three launches) for stream compaction, index gathers,
// and one deliberately serial pass (the exposure running sums, see below).
#include "common.hpp"
#include <type_traits>
#include <mutex>
#include <vector>

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

// The same searches inside [lo, hi): for callers that know the answer lies there (the
// threads of a workgroup whose queries are sorted: the answers of its first and last thread
// bracket all the others, and 256 neighbouring cells span a few thousand elements, not 10^7:
// half the probes, and all of them in a few tens of KB that stay in cache).
__device__ __forceinline__ int64_t lower_bound_in(const double *a, int64_t lo, int64_t hi, double v)
{
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if (a[mid] < v) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}
__device__ __forceinline__ int64_t upper_bound_in(const double *a, int64_t lo, int64_t hi, double v)
{
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if (a[mid] <= v) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

// lower_bound (UPPER = false) or upper_bound of a value that is the same in every lane,
// found by the 64 lanes of a wavefront together: a step probes 64 evenly spaced elements,
// one a lane, and keeps the 64th of the range the answer lies in -- four load latencies for
// 10^7 elements where a lane on its own takes twenty-four.  All 64 lanes must be active.
template <bool UPPER>
__device__ __forceinline__ int64_t bound_by_wave(const double *a, int64_t n, double v)
{
    const int lane = threadIdx.x & 63;
    // invariant: the predicate (a[i] < v, or a[i] <= v) holds for every i < lo and for no i >= hi
    int64_t lo = 0, hi = n;
    while (hi - lo > 64) {
        const int64_t step = (hi - lo + 63) >> 6;
        const int64_t p = lo + step * (lane + 1) - 1;
        bool t = false;
        if (p < hi) {
            const double x = a[p];
            t = UPPER ? x <= v : x < v;
        }
        const int cnt = __popcll(__ballot(t));              // the first cnt probes hold
        const int64_t failed = lo + step * (cnt + 1) - 1;   // the first probe that does not, if in range
        lo += step * cnt;
        if (failed < hi) hi = failed;
    }
    const int64_t p = lo + lane;
    bool t = false;
    if (p < hi) {
        const double x = a[p];
        t = UPPER ? x <= v : x < v;
    }
    return lo + __popcll(__ballot(t));
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
// order.  Short tables: the running sums are formed serially, bit for bit:
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
    // lanes 0..3 of wave 0: running sum of series `lane`; -0.0 + x = x for every x, so that the
    // first sum is the first row itself, as numpy.cumsum has it
    double carry = -0.0;
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

// ---------------------------------------------------------------------------
// The same running sums, bit for bit, in parallel.
//
// While the running sum s stays inside one binade [2^k, 2^(k+1)) it is a
// multiple S u of u = 2^(k-52), and fl(s + x) for x >= 0 is S u plus x rounded
// to a multiple of u: an INTEGER addition S += rn(x / u), where only an exact
// tie (x / u = X + 1/2) looks at S, through its parity (ties go to even).  An
// element is therefore a map S -> S + (S even ? e : o), and such maps compose
// into maps of the same form, so a chunk of rows reduces to one pair (e, o) and
// can be scanned in parallel in exact integer arithmetic.
//
//   1. per chunk of 2048 rows: plain fp64 totals (estimates only);
//   2. their prefix: an estimate of s at the start of every chunk, from which
//      the binade k of the chunk is guessed;
//   3. per chunk: the composed pair (e, o) for that k, or "irregular" (negative
//      or non-finite x, no estimate);
//   4. one workgroup per series walks the chunks with the exact s: a regular
//      chunk is accepted if s really lies in binade k and the sum stays below
//      2^(k+1) (x >= 0: then every partial sum does) and costs one integer
//      addition -- runs of sixteen such chunks are composed beforehand and
//      taken in one step; any other chunk is added row by row, as numpy does;
//   5. per accepted chunk: the scan inside the chunk from its exact start.
// A binade is crossed some forty times in a ten-year table, so all but a few
// dozen chunks per series go the parallel way.
// (Rows of -0.0 count as +0.0 here: a running sum that numpy would still show
// as -0.0, because every row so far was -0.0, comes out as +0.0.  The serial
// sums of short tables keep that sign.)
// ---------------------------------------------------------------------------
constexpr int CS_CHUNK = 2048;
constexpr int CS_THREADS = 256;
constexpr int CS_RPT = CS_CHUNK / CS_THREADS;
constexpr int CS_INFO_TILE = 2048;          // chunk records staged in LDS per trip of the walk
constexpr int CS_GROUP = 16;                // chunks composed into one step of the walk
constexpr int CS_K_ANY = INT32_MIN;         // all-zero chunk: leaves any s unchanged
constexpr int64_t CS_2P53 = (int64_t)1 << 53;

struct Adder {
    int64_t even, odd;                      // S -> S + (S even ? even : odd)
};
__device__ __forceinline__ Adder adder_then(const Adder &f, const Adder &g)   // f first, then g
{
    Adder h;
    h.even = f.even + ((f.even & 1) ? g.odd : g.even);
    h.odd = f.odd + (((f.odd + 1) & 1) ? g.odd : g.even);
    return h;
}
__device__ __forceinline__ int64_t adder_apply(int64_t S, const Adder &f) { return S + ((S & 1) ? f.odd : f.even); }

struct ChunkInfo {
    int64_t even, odd;
    int32_t k;
    int32_t regular;
};

__device__ __forceinline__ double quad_get(const Quad &q, int c)
{
    return c == 0 ? q.a : (c == 1 ? q.b : (c == 2 ? q.c : q.d));
}

// x as an adder in units of u = 2^(k-52); false when x does not qualify (negative, not a
// number, or 2^(k+1) and more).  x = M 2^(e-1075) with its significand M as an integer, so
// x / u = M / 2^sh, sh = k + 1023 - e: the quotient is the adder, the remainder against half
// a unit says which way it rounds.  Integer work on the bits of x, no branches.
__device__ __forceinline__ bool element_adder(double x, int k, Adder &f)
{
    constexpr int64_t MANT = ((int64_t)1 << 52) - 1;
    const int64_t b = __double_as_longlong(x);
    const int e = (int)(b >> 52) & 0x7ff;
    const int64_t M = e ? ((b & MANT) | (MANT + 1)) : (b & MANT);    // subnormals: no leading bit, e = 1
    const int sh = k + 1023 - (e ? e : 1);
    const int shc = sh > 63 ? 63 : (sh < 0 ? 0 : sh);                 // 54 and more: x is below half a unit
    const int64_t X0 = M >> shc;
    const int64_t rem = M - (X0 << shc);
    const int64_t half = shc ? (int64_t)1 << (shc - 1) : INT64_MAX;  // no shift, no remainder
    const bool up = rem > half, tie = rem == half;
    const bool x0_odd = X0 & 1;
    f.even = X0 + ((up || (tie && x0_odd)) ? 1 : 0);                  // a tie goes to the even sum
    f.odd = X0 + ((up || (tie && !x0_odd)) ? 1 : 0);
    return x >= 0.0 && sh >= 0;
}

__device__ __forceinline__ Quad quad_block_sum(Quad v, Quad *sm)
{
    v.a = team_sum<WAVE>(v.a);
    v.b = team_sum<WAVE>(v.b);
    v.c = team_sum<WAVE>(v.c);
    v.d = team_sum<WAVE>(v.d);
    if (threadIdx.x % WAVE == 0) sm[threadIdx.x / WAVE] = v;
    __syncthreads();
    Quad t = sm[0];
    for (int q = 1; q < CS_THREADS / WAVE; ++q) t = t + sm[q];
    __syncthreads();
    return t;
}

// step 1 (the bodies take the chunk or series a workgroup works on, so that launches over the
// tables of many sources can call them with blockIdx.y as the source)
__device__ __forceinline__ void cs_totals_body(const ExposureLoad &ld, int64_t n, Quad *totals, int64_t chunk)
{
    __shared__ Quad sm[CS_THREADS / WAVE];
    const int64_t base = chunk * CS_CHUNK;
    Quad v{0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < CS_RPT; ++j) {
        const int64_t r = base + threadIdx.x + j * CS_THREADS;
        if (r < n) v = v + ld(r);
    }
    v = quad_block_sum(v, sm);
    if (threadIdx.x == 0) totals[chunk] = v;
}
__global__ __launch_bounds__(CS_THREADS) void cs_totals_kernel(ExposureLoad ld, int64_t n, Quad *totals)
{
    cs_totals_body(ld, n, totals, blockIdx.x);
}

// step 2: est[c] = sum of totals[0 .. c-1] (one workgroup)
__device__ __forceinline__ void cs_estimate_body(const Quad *totals, int64_t nchunk, Quad *est)
{
    __shared__ Quad part[1024];
    const int64_t per = (nchunk + 1023) / 1024;
    const int64_t c0 = (int64_t)threadIdx.x * per;
    const int64_t c1 = c0 + per < nchunk ? c0 + per : nchunk;
    Quad v{0, 0, 0, 0};
    for (int64_t c = c0; c < c1; ++c) v = v + totals[c];
    part[threadIdx.x] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        Quad run{0, 0, 0, 0};
        for (int t = 0; t < 1024; ++t) {
            Quad nx = run + part[t];
            part[t] = run;
            run = nx;
        }
    }
    __syncthreads();
    v = part[threadIdx.x];
    for (int64_t c = c0; c < c1; ++c) {
        est[c] = v;
        v = v + totals[c];
    }
}
__global__ __launch_bounds__(1024) void cs_estimate_kernel(const Quad *totals, int64_t nchunk, Quad *est)
{
    cs_estimate_body(totals, nchunk, est);
}

// A chunk of the four series in LDS, one padded row of doubles per series: filled
// with coalesced loads (consecutive lanes, consecutive rows), read back by the
// thread that owns CS_RPT consecutive rows.  One spare slot per 8 rows keeps those
// reads (stride 8 rows across lanes) off a common bank.
constexpr int CS_PAD = CS_CHUNK + CS_CHUNK / 8;
__device__ __forceinline__ int cs_slot(int r) { return r + (r >> 3); }

__device__ __forceinline__ void cs_stage_chunk(const ExposureLoad &ld, int64_t base, int64_t n,
                                               double (&s_v)[4][CS_PAD])
{
#pragma unroll
    for (int j = 0; j < CS_RPT; ++j) {
        const int r = threadIdx.x + j * CS_THREADS;
        const Quad q = (base + r < n) ? ld(base + r) : Quad{0, 0, 0, 0};
        const int o = cs_slot(r);
        s_v[0][o] = q.a;
        s_v[1][o] = q.b;
        s_v[2][o] = q.c;
        s_v[3][o] = q.d;
    }
    __syncthreads();
}

// step 3
__device__ __forceinline__ void cs_adders_body(const ExposureLoad &ld, int64_t n, const Quad *est, ChunkInfo *info,
                                               int64_t chunk)
{
    __shared__ double s_v[4][CS_PAD];
    __shared__ Adder sm[CS_THREADS];
    __shared__ int sm_flag[2];
    const int tid = threadIdx.x;
    cs_stage_chunk(ld, chunk * CS_CHUNK, n, s_v);
    const Quad e = est[chunk];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const double ec = quad_get(e, c);
        const bool have_k = ec > 0.0 && ec < INFINITY;
        const int k = have_k ? ilogb(ec) : 0;
        Adder f{0, 0};
        bool ok = true, allzero = true;
#pragma unroll
        for (int j = 0; j < CS_RPT; ++j) {
            const double x = s_v[c][cs_slot(tid * CS_RPT + j)];
            allzero = allzero && (x == 0.0);
            Adder g{0, 0};
            ok = ok && element_adder(x, k, g);
            f = adder_then(f, g);
        }
        if (tid == 0) sm_flag[0] = sm_flag[1] = 1;
        __syncthreads();
        if (!ok) sm_flag[0] = 0;
        if (!allzero) sm_flag[1] = 0;
        sm[tid] = f;
        __syncthreads();
        for (int st = 1; st < CS_THREADS; st <<= 1) {
            if ((tid & (2 * st - 1)) == 0) sm[tid] = adder_then(sm[tid], sm[tid + st]);
            __syncthreads();
        }
        if (tid == 0) {
            ChunkInfo ci;
            if (sm_flag[1]) {
                ci = ChunkInfo{0, 0, CS_K_ANY, 1};
            } else {
                const bool reg = have_k && sm_flag[0] && sm[0].even < CS_2P53 && sm[0].odd < CS_2P53 &&
                                 sm[0].even >= 0 && sm[0].odd >= 0;
                ci = ChunkInfo{sm[0].even, sm[0].odd, k, reg ? 1 : 0};
            }
            info[chunk * 4 + c] = ci;
        }
        __syncthreads();
    }
}
__global__ __launch_bounds__(CS_THREADS) void cs_adders_kernel(ExposureLoad ld, int64_t n, const Quad *est,
                                                               ChunkInfo *info)
{
    cs_adders_body(ld, n, est, info, blockIdx.x);
}

// step 4: blockIdx.x = series.  The exact sum s is carried as its bit pattern:
// for s in binade k the integer S = s / u is the significand with its leading
// bit, so entering and leaving a regular chunk is integer work only.
// COMPACT: cum holds one record per sixteen rows, cum[j] = the sums through row 16 j + 15 (or
// through the last row of the table): godot_exposure_checkpoints.
template <bool COMPACT>
__device__ __forceinline__ void cs_walk_body(const ExposureLoad &ld, int64_t n, int64_t nchunk, const ChunkInfo *info,
                                             double *entry, int32_t *by_walk, Quad *cum, int col)
{
    __shared__ ChunkInfo s_info[CS_INFO_TILE];
    __shared__ double s_entry[CS_INFO_TILE];
    __shared__ int32_t s_by[CS_INFO_TILE];
    __shared__ double s_x[CS_CHUNK];
    __shared__ int64_t s_chunk;
    // runs of CS_GROUP chunks that share a binade, composed into one adder each: thread 0
    // steps over a whole run when the sum stays inside the binade to its end
    __shared__ ChunkInfo s_group[CS_INFO_TILE / CS_GROUP];
    __shared__ int64_t s_gbits[CS_INFO_TILE / CS_GROUP];
    __shared__ int32_t s_gtaken[CS_INFO_TILE / CS_GROUP];
    constexpr int64_t MANT = ((int64_t)1 << 52) - 1;
    const int tid = threadIdx.x;
    double *cumd = (double *)cum;
    double s = 0.0;                       // thread 0: the exact running sum
    for (int64_t tile0 = 0; tile0 < nchunk; tile0 += CS_INFO_TILE) {
        const int64_t tile1 = tile0 + CS_INFO_TILE < nchunk ? tile0 + CS_INFO_TILE : nchunk;
        for (int64_t c = tile0 + tid; c < tile1; c += CS_THREADS) s_info[c - tile0] = info[c * 4 + col];
        __syncthreads();
        const int tlen = (int)(tile1 - tile0);
        for (int g = tid; g * CS_GROUP < tlen; g += CS_THREADS) {
            const int q1 = (g + 1) * CS_GROUP < tlen ? (g + 1) * CS_GROUP : tlen;
            Adder f{0, 0};
            int k = CS_K_ANY;
            bool ok = true;
            for (int q = g * CS_GROUP; q < q1; ++q) {
                const ChunkInfo in = s_info[q];
                ok = ok && in.regular;
                if (in.k != CS_K_ANY) {
                    ok = ok && (k == CS_K_ANY || k == in.k);
                    k = in.k;
                    f = adder_then(f, Adder{in.even, in.odd});
                }
            }
            ok = ok && f.even >= 0 && f.odd >= 0 && f.even < CS_2P53 && f.odd < CS_2P53;
            s_group[g] = ChunkInfo{f.even, f.odd, k, ok ? 1 : 0};
            s_gtaken[g] = 0;
        }
        __syncthreads();
        int64_t chunk = tile0;
        while (true) {
            if (tid == 0) {
                // over the chunks that go the parallel way
                int64_t bits = __double_as_longlong(s);
                while (chunk < tile1) {
                    const int ci = (int)(chunk - tile0);
                    if ((ci % CS_GROUP) == 0 && s_group[ci / CS_GROUP].regular) {
                        // the whole run at once, if the sum is in its binade and stays there
                        const ChunkInfo gi = s_group[ci / CS_GROUP];
                        int64_t nb = bits;
                        bool ok = true;
                        if (gi.k != CS_K_ANY) {
                            const int64_t ex = bits >> 52;
                            ok = ex == (int64_t)gi.k + 1023 && ex > 0 && ex < 2047;
                            if (ok) {
                                const int64_t S2 = adder_apply((bits & MANT) | (MANT + 1), Adder{gi.even, gi.odd});
                                ok = S2 < CS_2P53;
                                nb = (ex << 52) | (S2 & MANT);
                            }
                        }
                        if (ok) {
                            s_gbits[ci / CS_GROUP] = bits;
                            s_gtaken[ci / CS_GROUP] = 1;
                            bits = nb;
                            chunk += (tlen - ci) < CS_GROUP ? (tlen - ci) : CS_GROUP;
                            continue;
                        }
                    }
                    const ChunkInfo in = s_info[chunk - tile0];
                    if (!in.regular) break;
                    if (in.k != CS_K_ANY) {
                        // s positive, finite, normal and in binade k <=> sign 0, biased exponent k + 1023
                        const int64_t ex = bits >> 52;
                        if (ex != (int64_t)in.k + 1023 || ex <= 0 || ex >= 2047) break;
                        const int64_t S = (bits & MANT) | (MANT + 1);
                        const int64_t S2 = adder_apply(S, Adder{in.even, in.odd});
                        if (S2 >= CS_2P53) break;
                        s_entry[chunk - tile0] = __longlong_as_double(bits);
                        bits = (ex << 52) | (S2 & MANT);
                    } else {
                        s_entry[chunk - tile0] = __longlong_as_double(bits);
                    }
                    s_by[chunk - tile0] = 0;
                    ++chunk;
                }
                s = __longlong_as_double(bits);
                s_chunk = chunk;
            }
            __syncthreads();
            chunk = s_chunk;
            if (chunk >= tile1) break;
            // this chunk row by row: everyone fetches, thread 0 adds, everyone stores
            const int64_t base = chunk * CS_CHUNK;
            const int m = (int)((n - base) < CS_CHUNK ? (n - base) : CS_CHUNK);
#pragma unroll
            for (int j = 0; j < CS_RPT; ++j) {
                const int r = tid + j * CS_THREADS;
                s_x[r] = (r < m) ? quad_get(ld(base + r), col) : 0.0;
            }
            __syncthreads();
            if (tid == 0) {
                double a[16];
                for (int r0 = 0; r0 < CS_CHUNK; r0 += 16) {
#pragma unroll
                    for (int u = 0; u < 16; ++u) a[u] = s_x[r0 + u];
#pragma unroll
                    for (int u = 0; u < 16; ++u) {
                        s = __dadd_rn(s, a[u]);
                        a[u] = s;
                    }
#pragma unroll
                    for (int u = 0; u < 16; ++u) s_x[r0 + u] = a[u];
                    if (r0 + 16 >= m) break;
                }
                // rows past the end of the table were zeros: s is the sum of the m rows
                s = s_x[m - 1];
                s_by[chunk - tile0] = 1;
                s_entry[chunk - tile0] = 0.0;
            }
            __syncthreads();
            for (int r = tid; r < m; r += CS_THREADS) {
                if (!COMPACT) cumd[(base + r) * 4 + col] = s_x[r];
                else if ((r & 15) == 15 || base + r == n - 1) cumd[((base + r) >> 4) * 4 + col] = s_x[r];
            }
            ++chunk;
            __syncthreads();
        }
        __syncthreads();
        // the chunks inside the runs taken whole: their starts follow from the run's start
        for (int ci = tid; ci < tlen; ci += CS_THREADS) {
            const int g = ci / CS_GROUP;
            if (!s_gtaken[g]) continue;
            int64_t bits = s_gbits[g];
            if (s_group[g].k != CS_K_ANY) {
                const int64_t ex = bits >> 52;
                int64_t S = (bits & MANT) | (MANT + 1);
                for (int q = g * CS_GROUP; q < ci; ++q) {
                    const ChunkInfo in = s_info[q];
                    if (in.k != CS_K_ANY) S = adder_apply(S, Adder{in.even, in.odd});
                }
                bits = (ex << 52) | (S & MANT);
            }
            s_entry[ci] = __longlong_as_double(bits);
            s_by[ci] = 0;
        }
        __syncthreads();
        for (int64_t c = tile0 + tid; c < tile1; c += CS_THREADS) {
            entry[c * 4 + col] = s_entry[c - tile0];
            by_walk[c * 4 + col] = s_by[c - tile0];
        }
        __syncthreads();
    }
}
__global__ __launch_bounds__(CS_THREADS) void cs_walk_kernel(ExposureLoad ld, int64_t n, int64_t nchunk,
                                                             const ChunkInfo *info, double *entry, int32_t *by_walk,
                                                             Quad *cum)
{
    cs_walk_body<false>(ld, n, nchunk, info, entry, by_walk, cum, blockIdx.x);
}

// step 5
__device__ __forceinline__ void cs_fill_body(const ExposureLoad &ld, int64_t n, const ChunkInfo *info,
                                             const double *entry, const int32_t *by_walk, Quad *cum, int64_t chunk)
{
    __shared__ double s_v[4][CS_PAD];
    __shared__ Adder sm[2][CS_THREADS];
    const int tid = threadIdx.x;
    const int64_t base = chunk * CS_CHUNK;
    cs_stage_chunk(ld, base, n, s_v);
    int done = 0;                                     // series whose sums now stand in s_v
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int64_t ic = chunk * 4 + c;
        if (by_walk[ic]) continue;                    // uniform over the workgroup
        done |= 1 << c;
        const ChunkInfo in = info[ic];
        const double s0 = entry[ic];
        if (in.k == CS_K_ANY) {
#pragma unroll
            for (int j = 0; j < CS_RPT; ++j) s_v[c][cs_slot(tid * CS_RPT + j)] = s0;
            continue;
        }
        Adder g[CS_RPT];
        Adder f{0, 0};
#pragma unroll
        for (int j = 0; j < CS_RPT; ++j) {
            g[j] = Adder{0, 0};
            element_adder(s_v[c][cs_slot(tid * CS_RPT + j)], in.k, g[j]);
            f = adder_then(f, g[j]);
        }
        // inclusive scan of the threads' adders (Hillis-Steele, order kept)
        int cur = 0;
        sm[0][tid] = f;
        __syncthreads();
        for (int off = 1; off < CS_THREADS; off <<= 1) {
            Adder v = sm[cur][tid];
            if (tid >= off) v = adder_then(sm[cur][tid - off], v);
            sm[cur ^ 1][tid] = v;
            cur ^= 1;
            __syncthreads();
        }
        int64_t S = (int64_t)ldexp(s0, 52 - in.k);
        if (tid > 0) S = adder_apply(S, sm[cur][tid - 1]);
#pragma unroll
        for (int j = 0; j < CS_RPT; ++j) {
            S = adder_apply(S, g[j]);
            s_v[c][cs_slot(tid * CS_RPT + j)] = ldexp((double)S, in.k - 52);
        }
        __syncthreads();
    }
    __syncthreads();
    // out in whole rows when all four series were scanned here, else series by series
    double *cumd = (double *)cum;
#pragma unroll
    for (int j = 0; j < CS_RPT; ++j) {
        const int r = tid + j * CS_THREADS;
        if (base + r >= n) continue;
        const int64_t row = base + r;
        const int o = cs_slot(r);
        if (done == 15) {
            cum[row] = Quad{s_v[0][o], s_v[1][o], s_v[2][o], s_v[3][o]};
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (done & (1 << c)) cumd[row * 4 + c] = s_v[c][o];
        }
    }
}
__global__ __launch_bounds__(CS_THREADS) void cs_fill_kernel(ExposureLoad ld, int64_t n, const ChunkInfo *info,
                                                             const double *entry, const int32_t *by_walk,
                                                             Quad *cum)
{
    cs_fill_body(ld, n, info, entry, by_walk, cum, blockIdx.x);
}

__host__ __device__ inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
inline int64_t cs_chunks(int64_t n) { return (n + CS_CHUNK - 1) / CS_CHUNK; }
inline size_t cs_scratch_bytes(int64_t n)
{
    const size_t nc = (size_t)cs_chunks(n);
    return 2 * align256(sizeof(Quad) * nc) + align256(sizeof(ChunkInfo) * 4 * nc) +
           align256(sizeof(double) * 4 * nc) + align256(sizeof(int32_t) * 4 * nc);
}

// What a query reads of the row it falls in, side by side: the six numbers lie in five
// arrays of 8 bytes a row, cell edges fall some twenty rows apart, and every one of them cost
// a cache line of its own -- a call over 5 10^5 edges streamed the whole table (0.5 GB).
// (The running sums stay in an array of their own: with them in the same record, 80 bytes a
// row, the query was slower, 94 us against 67 us.)
struct ExpRow {
    double t0, t1, e, se, be, dead;
};
__global__ void exposure_rows_kernel(ExposureLoad ld, int64_t n, ExpRow *rows)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Quad x = ld(i);
    rows[i] = ExpRow{ld.tstart[i], ld.tstop[i], x.a, x.b, x.c, x.d};
}

// Where a kernel's sorted values (query times, cell stops) come from: an array, or the
// uniform cell edges first + i, i = 0, 1, ..., formed as numpy forms tstart + arange(n) * tcell
// (core.py:2071): the product rounded, then the sum rounded.
struct ArrayAt {
    const double *p;
    __device__ __forceinline__ double operator()(int64_t i) const { return p[i]; }
};
struct UniformEdge {
    double t0, tcell;
    int64_t first;
    __device__ __forceinline__ double operator()(int64_t i) const
    {
        return __dadd_rn(__dmul_rn((double)(i + first), tcell), t0);
    }
};

// cumulative exposures at the query times: core.py:1806-1832.  Called by every thread of a
// 256-thread workgroup (it synchronises); the workgroup serves queries 256 * blockIdx.x ...
// The row a query falls in: false for the threads past the last query.
// `bracket`: null, or the rows of this workgroup's first query and of the next workgroup's first
// (the last query's for the last workgroup), found by a launch ahead: sources_brackets_kernel.
// find_row = false leaves idx to the caller (who searches a copy of the bracket's rows in LDS).
template <class TimeAt>
__device__ __forceinline__ bool exposure_query_row(const ExposureLoad &ld, int64_t n_ft2, const TimeAt times,
                                                   int64_t n_t, int64_t &k, double &t, int64_t &idx, int64_t &row_first,
                                                   int64_t &row_last, const int64_t *bracket = nullptr,
                                                   bool find_row = true, int64_t wg = -1)
{
    if (wg < 0) wg = blockIdx.x;   // which 256 (blockDim.x) queries the workgroup serves
    // sorted queries (cell edges): the rows of the workgroup's first and last query bracket
    // the rows of all the others; those two are searched for by a wavefront each
    __shared__ double s_t[2];
    __shared__ int64_t s_i[2];
    k = wg * blockDim.x + threadIdx.x;
    const int64_t kb = wg * blockDim.x;
    const int64_t kl = (kb + blockDim.x < n_t ? kb + blockDim.x : n_t) - 1;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (bracket) {
        if (threadIdx.x < 2) {
            const int64_t e = threadIdx.x == 0 ? kb : (kb + blockDim.x < n_t ? kb + blockDim.x : n_t - 1);
            s_t[threadIdx.x] = times(e);
            s_i[threadIdx.x] = bracket[threadIdx.x];
        }
    } else if (wave == 0) {
        const double tb = times(kb);
        const int64_t r = bound_by_wave<false>(ld.tstop, n_ft2, tb);
        if (lane == 0) {
            s_t[0] = tb;
            s_i[0] = r;
        }
    }
    if (!bracket && wave == (int)((kl - kb) >> 6)) {
        const double tl = times(kl);
        const int64_t r = bound_by_wave<false>(ld.tstop, n_ft2, tl);
        if (lane == 0) {
            s_t[1] = tl;
            s_i[1] = r;
        }
    }
    __syncthreads();
    row_first = s_i[0];          // the rows of the workgroup's first and last query
    row_last = s_i[1];
    if (k >= n_t) return false;
    t = times(k);
    if (!find_row) return true;
    if (s_t[0] <= t && t <= s_t[1]) {
        // The rows of a spacecraft file are evenly spaced but for the gaps: the row is guessed
        // by proportion inside the bracket and the guess and its two neighbours are looked at
        // with loads that do not depend on one another; only where that does not settle it
        // (a gap inside the bracket) the bracket is searched.
        const int64_t i0 = s_i[0], i1 = s_i[1];
        int64_t g = i0;
        if (i1 > i0 && s_t[1] > s_t[0]) {
            g = i0 + (int64_t)((t - s_t[0]) / (s_t[1] - s_t[0]) * (double)(i1 - i0));
            g = g < i0 ? i0 : (g > i1 ? i1 : g);
        }
        const int64_t last = n_ft2 - 1;
        const double below = ld.tstop[g > 0 ? g - 1 : 0];
        const double at_g = ld.tstop[g < last ? g : last];
        const double above = ld.tstop[g + 1 < last ? g + 1 : last];
        // lower_bound(t) = g  iff  tstop[g - 1] < t <= tstop[g]  (with the ends of the table open)
        if ((g == 0 || below < t) && (g > last || t <= at_g)) idx = g;
        else if (g <= last && at_g < t && (g + 1 > last || t <= above)) idx = g + 1;
        else idx = lower_bound_in(ld.tstop, i0, i1, t);
    } else {
        idx = lower_bound(ld.tstop, n_ft2, t);   // unsorted queries, NaN
    }
    if (idx >= n_ft2) idx = n_ft2 - 1;   // the reference would index out of range here
    return true;
}
// the sums at time t inside row r, whose running sums through the row are c: core.py:1826-1832
__device__ __forceinline__ Quad exposure_within_row(double t, const ExpRow &r, const Quad &c)
{
    double frac = __ddiv_rn(__dsub_rn(t, r.t0), __dsub_rn(r.t1, r.t0));
    frac = frac < 0 ? 0.0 : (frac > 1 ? 1.0 : frac);   // np.clip keeps NaN
    frac = 1 - frac;
    return Quad{__dsub_rn(c.a, __dmul_rn(r.e, frac)), __dsub_rn(c.b, __dmul_rn(r.se, frac)),
                __dsub_rn(c.c, __dmul_rn(r.be, frac)), __dsub_rn(c.d, __dmul_rn(r.dead, frac))};
}
template <class TimeAt>
__device__ __forceinline__ void exposure_query_body(const ExposureLoad &ld, const Quad *cum, const ExpRow *rows,
                                                    int64_t n_ft2, const TimeAt times, int64_t n_t, Quad *at,
                                                    const int64_t *bracket = nullptr, int64_t wg = -1)
{
    int64_t k, idx, i0, i1;
    double t;
    if (!exposure_query_row(ld, n_ft2, times, n_t, k, t, idx, i0, i1, bracket, true, wg)) return;
    at[k] = exposure_within_row(t, rows[idx], cum[idx]);
}

// The same from the running sums kept for every sixteenth row only (godot_exposure_checkpoints:
// ck[j] = the sums through row 16 j + 15): the sums through the query's row are the record
// before its group of sixteen plus the rows of the group up to it, added one after another
// from the left as numpy.cumsum adds them -- the same roundings, so the same bits as the full
// table gives, for 2 bytes a row instead of 80.
constexpr int CK_ROWS = 16;
__host__ __device__ inline int64_t ck_records(int64_t n) { return (n + CK_ROWS - 1) / CK_ROWS; }
// Cells of a few rows each (300-s cells over 30-s rows): the 256 queries of a workgroup fall in
// a short run of rows, and the up to sixteen rows a query adds overlap those of its neighbours.
// The workgroup then fetches the run once, with whole cache lines, into LDS (exposure and
// dead time of a row; the two multiples of the exposure are formed as ExposureLoad forms
// them) and the queries add from there; a thread that walks its rows in memory waits for
// every row in turn (measured 4.3 ms against 1.1 ms for 8.4 10^6 cells).
constexpr int CK_LDS_ROWS = 1536;
template <class TimeAt>
__device__ __forceinline__ void exposure_query_ck_body(const ExposureLoad &ld, const Quad *ck, int64_t n_ft2,
                                                       const TimeAt times, int64_t n_t, Quad *at,
                                                       const int64_t *bracket = nullptr, int64_t wg = -1)
{
    // a spare slot after every sixteen rows: the queries of neighbouring lanes start their walks at
    // the heads of neighbouring groups of sixteen, which would otherwise all lie on one LDS bank
    // (measured: four cycles of bank conflict for every cycle of LDS work)
    __shared__ double2 s_row[CK_LDS_ROWS + CK_LDS_ROWS / CK_ROWS];
    __shared__ double s_stop[CK_LDS_ROWS];
    auto slot = [](int64_t i) { return i + (i >> 4); };
    int64_t k, idx = 0, i0, i1;
    double t;
    // with the bracket known ahead, the rows are fetched first and every query finds its row in
    // the LDS copy of their TSTOP: no chain of dependent loads from memory per query
    const bool valid = exposure_query_row(ld, n_ft2, times, n_t, k, t, idx, i0, i1, bracket, bracket == nullptr, wg);
    const int64_t r_lo = (i0 < n_ft2 ? i0 : n_ft2 - 1) / CK_ROWS * CK_ROWS;
    const int64_t r_hi = i1 < n_ft2 ? i1 : n_ft2 - 1;
    const bool staged = !ld.sexposure && r_hi >= r_lo && r_hi - r_lo < CK_LDS_ROWS;   // the same for all threads
    if (staged) {
        for (int64_t r = r_lo + threadIdx.x; r <= r_hi; r += blockDim.x) {
            const double stop = ld.tstop[r];
            s_row[slot(r - r_lo)] = double2{ld.exposure[r], __dsub_rn(__dsub_rn(stop, ld.tstart[r]), ld.livetime[r])};
            s_stop[r - r_lo] = stop;
        }
        __syncthreads();
    }
    if (!valid) return;
    if (bracket) {
        // lower_bound(TSTOP, t) lies between the rows of the bracket's ends
        int64_t lo = i0 < n_ft2 ? i0 : n_ft2 - 1, hi = r_hi;
        if (staged) {
            while (lo < hi) {
                const int64_t mid = (lo + hi) >> 1;
                if (s_stop[mid - r_lo] < t) lo = mid + 1;
                else hi = mid;
            }
            idx = lo;
        } else {
            idx = lower_bound_in(ld.tstop, lo, hi, t);
        }
    }
    const int64_t j = idx / CK_ROWS;
    int64_t r = j * CK_ROWS;
    Quad x, c;
    if (staged && r >= r_lo && idx <= r_hi) {
        auto row = [&](int64_t i) {
            const double2 v = s_row[slot(i - r_lo)];
            return Quad{v.x, __dmul_rn(ld.s_over_e, v.x), __dmul_rn(ld.b_over_e, v.x), v.y};
        };
        x = row(r);
        c = j > 0 ? ck[j - 1] + x : x;
        for (++r; r <= idx; ++r) {
            x = row(r);
            c = c + x;
        }
    } else {
        x = ld(r);
        c = j > 0 ? ck[j - 1] + x : x;
        for (++r; r <= idx; ++r) {
            x = ld(r);
            c = c + x;
        }
    }
    at[k] = exposure_within_row(t, ExpRow{ld.tstart[idx], ld.tstop[idx], x.a, x.b, x.c, x.d}, c);
}

__global__ __launch_bounds__(256) void exposure_query_kernel(ExposureLoad ld, const Quad *cum,
                                                             const ExpRow *rows, int64_t n_ft2,
                                                             const double *times, int64_t n_t, Quad *at)
{
    exposure_query_body(ld, cum, rows, n_ft2, ArrayAt{times}, n_t, at);
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

// photon range of every cell (see godot_bin_photons in the header); `shift` is added to the
// indices written.  Called by every thread of a 256-thread workgroup (it synchronises).
template <class StopAt>
__device__ __forceinline__ void bin_photons_body(const double *ti, int64_t nph, double start0, const StopAt stops,
                                                 int64_t ncell, int64_t *lo, int64_t *hi, int64_t *counts,
                                                 int64_t shift)
{
    // one search per cell: where its stop falls among the photon times; a cell starts where
    // the cell before it ends, which the neighbouring thread has found.  The stops are
    // sorted, so the workgroup's first and last answers bracket the rest.
    __shared__ int64_t ends[257];
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t cb = (int64_t)blockIdx.x * blockDim.x;
    const int64_t cl = (cb + blockDim.x < ncell ? cb + blockDim.x : ncell) - 1;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave == 0) {   // where the workgroup's first cell starts
        const int64_t r = (cb == 0) ? bound_by_wave<false>(ti, nph, start0) : bound_by_wave<true>(ti, nph, stops(cb - 1));
        if (lane == 0) ends[0] = r;
    }
    if (wave == (int)((cl - cb) >> 6)) {   // where its last cell ends
        const int64_t r = (cl == ncell - 1) ? bound_by_wave<false>(ti, nph, stops(cl)) : bound_by_wave<true>(ti, nph, stops(cl));
        if (lane == 0) ends[cl - cb + 1] = r;
    }
    __syncthreads();
    if (c < cl) {
        const double sc = stops(c);
        // holds for sorted stops; a thread for which it does not searches the whole list
        const bool bracketed = (cb == 0 || stops(cb - 1) <= sc) && sc < stops(cl);
        const int64_t wlo = cb == 0 ? 0 : ends[0];   // the first cell's start may lie beyond its stop
        const int64_t whi = ends[cl - cb + 1];
        ends[threadIdx.x + 1] =
            (bracketed && wlo <= whi) ? upper_bound_in(ti, wlo, whi, sc) : upper_bound(ti, nph, sc);
    }
    __syncthreads();
    if (c >= ncell) return;
    const int64_t l = ends[threadIdx.x];
    int64_t h = ends[threadIdx.x + 1];
    if (h < l) h = l;   // start0 beyond the first stop: empty first cell
    lo[c] = l + shift;
    hi[c] = h + shift;
    if (counts) counts[c] = h - l;
}

__global__ __launch_bounds__(256) void bin_photons_kernel(const double *ti, int64_t nph, double start0,
                                                          const double *stops, int64_t ncell, int64_t *lo,
                                                          int64_t *hi, int64_t *counts)
{
    bin_photons_body(ti, nph, start0, ArrayAt{stops}, ncell, lo, hi, counts, 0);
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

// The same selection applied to several per-cell arrays of 8-byte elements at once
// (the columns of a cell table: starts, stops, exposures, photon ranges, ...).
constexpr int GATHER_MAX_COLUMNS = 16;
struct GatherColumns {
    const uint64_t *src[GATHER_MAX_COLUMNS];
    uint64_t *dst[GATHER_MAX_COLUMNS];
};
__global__ void gather_columns_kernel(GatherColumns c, int ncol, const int64_t *idx, int64_t n)
{
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int64_t i = idx[k];
    for (int a = 0; a < ncol; ++a) c.dst[a][k] = c.src[a][i];
}

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

// Mean of a piecewise-constant function over the intervals t0[k] .. t1[k]
// (PiecewiseScaler.__call__ with two arguments, scalers.py:52-85): the value of the
// segment when the interval lies inside one, else the two partial segments at its ends
// plus the whole segments between them (one: its own product; several: a difference of
// the running sums of the products), divided by the length.  starts/stops/scales are the
// scaler's padded segment arrays, prod = (stops - starts) * scales and cprod its running
// sum, both rounded from extended precision as the reference stores them.  An interval
// that begins exactly where a segment ends belongs to the next segment.
__global__ void piecewise_average_kernel(const double *starts, const double *stops, const double *scales,
                                         const double *prod, const double *cprod, int64_t nseg,
                                         const double *t0, const double *t1, int64_t n, double *out)
{
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const double a = t0[k], b = t1[k];
    int64_t i0 = lower_bound(stops, nseg, a);
    const int64_t i1 = lower_bound(stops, nseg, b);
    if (i0 < nseg && a == stops[i0]) ++i0;
    const int64_t last = nseg - 1;
    const int64_t j0 = i0 < last ? i0 : last, j1 = i1 < last ? i1 : last;
    if (i0 == i1) {
        out[k] = scales[j0];
        return;
    }
    const double ends = scales[j1] * (b - starts[j1]) + scales[j0] * (stops[j0] - a);
    const int64_t between = i1 - i0 - 1;
    const int64_t f = j0 + 1 < last ? j0 + 1 : last;
    double whole = 0.0;
    if (between == 1) whole = prod[f];
    else if (between > 1) whole = cprod[j1] - cprod[f];
    out[k] = (ends + whole) / (b - a);
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

// the table: running sums (32 bytes a row), the rows as the queries read them (48 bytes a
// row), then what the parallel scan needs while the sums are built
__host__ __device__ inline size_t table_sums_bytes(int64_t n) { return align256(sizeof(Quad) * (size_t)n); }
inline size_t table_rows_bytes(int64_t n) { return align256(sizeof(ExpRow) * (size_t)n); }

// Many sources at once (see godot_bin_sources in the header): blockIdx.y is the source, counted
// from `first`; its descriptor says where its tables and photons are and where its cells go.
// Pass one: the running sums at the ncell + 1 edges of every source (source k's stand at
// at[cell_offset + k ...], one more entry than it has cells).
// Workgroups go to the eight XCDs (each with an L2 of its own) in turn: the workgroups that one XCD
// receives one after the other take the same piece (chunk, stretch of edges) for table after
// table, so that what the tables share is fetched into that L2 once.  Grids are multiples of 8 n.
__device__ __forceinline__ void xcd_deal(int64_t w, int64_t n, int64_t &which, int64_t &piece)
{
    const int64_t xcd = w & 7, q = w >> 3;
    which = q % n;
    piece = (q / n) * 8 + xcd;
}
// Where the brackets of source k's workgroups stand: one entry per workgroup of 256 edges and one
// more; the regions of consecutive sources do not overlap (their edges stand at cell_offset + k,
// ascending, at least ncell + 1 apart).
constexpr int SQ_THREADS = 128;   // edges per workgroup of sources_query_kernel
__host__ __device__ inline int64_t bracket_offset(int64_t cell_offset, int64_t k)
{
    return (cell_offset + k) / SQ_THREADS + 2 * k;
}
__host__ __device__ inline size_t bracket_entries(int64_t nsrc, int64_t total_cells)
{
    return (size_t)((total_cells + nsrc) / SQ_THREADS + 2 * nsrc + 2);
}
// The table row of the first edge of every workgroup of sources_query_kernel (and of the last
// edge of all): a search over the whole table each, all of them at once, so that the workgroups
// themselves start from a bracket instead of two searches of their own.
__global__ __launch_bounds__(256) void sources_brackets_kernel(const godot_source_cells *src, int64_t first,
                                                               int64_t *brackets)
{
    const int64_t k = first + blockIdx.y;
    const godot_source_cells s = src[k];
    const int64_t n_t = s.ncell + 1;
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s.ncell <= 0 || g > (n_t + SQ_THREADS - 1) / SQ_THREADS) return;
    const int64_t e = SQ_THREADS * g < n_t ? SQ_THREADS * g : n_t - 1;
    const UniformEdge edge{s.t0, s.tcell, 0};
    brackets[bracket_offset(s.cell_offset, k) + g] = lower_bound(s.tstop, s.n_ft2, edge(e));
}
// Consecutive workgroups serve the same stretch of edges for different sources (xcd_deal): sources
// over one spacecraft file read the same pieces of TSTART, TSTOP and LIVETIME then (from the cache;
// dealt source after source, every source streams all three again).
__global__ __launch_bounds__(SQ_THREADS) void sources_query_kernel(const godot_source_cells *src, int64_t first,
                                                                   int64_t nsrc, Quad *at, const int64_t *brackets)
{
    int64_t which, wg;
    xcd_deal(blockIdx.x, nsrc, which, wg);
    const int64_t k = first + which;
    const godot_source_cells s = src[k];
    const int64_t n_t = s.ncell + 1;
    if (s.ncell <= 0 || wg * blockDim.x >= n_t) return;   // the whole workgroup
    ExposureLoad ld{s.tstart, s.tstop, s.exposure, s.livetime, s.s_over_e, s.b_over_e, s.sexposure, s.bexposure};
    const int64_t *br = brackets + bracket_offset(s.cell_offset, k) + wg;
    if (s.checkpoints)   // uniform over the workgroup
        exposure_query_ck_body(ld, (const Quad *)s.table, s.n_ft2, UniformEdge{s.t0, s.tcell, 0}, n_t,
                               at + s.cell_offset + k, br, wg);
    else
        exposure_query_body(ld, (const Quad *)s.table,
                            (const ExpRow *)((const char *)s.table + table_sums_bytes(s.n_ft2)), s.n_ft2,
                            UniformEdge{s.t0, s.tcell, 0}, n_t, at + s.cell_offset + k, br, wg);
}

// Checkpointed running sums of many tables at once (see godot_exposure_checkpoints in the
// header): the five steps of the parallel running sums above, blockIdx.y = the table.
struct CkSource {
    ExposureLoad ld;
    int64_t n, nchunk;
    Quad *totals, *est;
    ChunkInfo *info;
    double *entry;
    int32_t *by_walk;
    Quad *ck;
};
// Consecutive workgroups of these launches work on the same chunk of different tables (xcd_deal):
// the tables of one spacecraft file share TSTART, TSTOP and LIVETIME, and workgroups that run side
// by side then read the same piece of them (from the cache, once the first has fetched it).
// Built for throughput where the single-table kernels above are built for latency: a thread
// takes eight consecutive rows straight from memory (no staging of the chunk in LDS, which
// held those kernels to one workgroup a CU) and the threads' adders are composed by wavefront
// shuffles.
__device__ __forceinline__ bool ck_which(const CkSource *src, int64_t ntab, const CkSource *&s, int64_t &chunk)
{
    int64_t t;
    xcd_deal(blockIdx.x, ntab, t, chunk);
    s = src + t;
    return chunk < s->nchunk;
}
__global__ __launch_bounds__(CS_THREADS) void ck_totals_kernel(const CkSource *src, int64_t ntab)
{
    const CkSource *s;
    int64_t chunk;
    if (!ck_which(src, ntab, s, chunk)) return;
    cs_totals_body(s->ld, s->n, s->totals, chunk);
}
__global__ __launch_bounds__(1024) void ck_estimate_kernel(const CkSource *src)
{
    const CkSource &s = src[blockIdx.x];
    if (s.nchunk == 0) return;
    cs_estimate_body(s.totals, s.nchunk, s.est);
}
__device__ __forceinline__ Adder adder_shfl_down(const Adder &f, int d)
{
    return Adder{__shfl_down(f.even, d, WAVE), __shfl_down(f.odd, d, WAVE)};
}
__device__ __forceinline__ Adder adder_shfl_up(const Adder &f, int d)
{
    return Adder{__shfl_up(f.even, d, WAVE), __shfl_up(f.odd, d, WAVE)};
}
// The eight consecutive rows of every thread of a wavefront (thread `lane` gets rows
// wbase + 8 lane ... + 7; zeros past the end of the table).  Loaded with consecutive lanes on
// consecutive rows (whole cache lines per instruction; a thread that loads its own 64 bytes
// makes every instruction touch 32 lines, which then fall out of the cache before the next
// instruction comes for them) and turned through `stage`, CK_STAGE doubles of LDS that belong
// to the wavefront.
constexpr int CK_STAGE = WAVE * CS_RPT + WAVE;
__device__ __forceinline__ void ck_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ void ck_load(const double *p, int64_t wbase, int64_t n, int lane, double (&t)[CS_RPT])
{
    if (wbase + WAVE * CS_RPT <= n) {
#pragma unroll
        for (int j = 0; j < CS_RPT; ++j) t[j] = p[wbase + WAVE * j + lane];
    } else {
#pragma unroll
        for (int j = 0; j < CS_RPT; ++j) {
            const int64_t r = wbase + WAVE * j + lane;
            t[j] = r < n ? p[r] : 0.0;
        }
    }
}
// from a row per lane and trip (row 64 j + lane in t[j]) to eight consecutive rows per lane
__device__ __forceinline__ void ck_turn(double (&t)[CS_RPT], int lane, double *stage)
{
    ck_wave_sync();                                   // the column before has been read
#pragma unroll
    for (int j = 0; j < CS_RPT; ++j) stage[cs_slot(WAVE * j + lane)] = t[j];
    ck_wave_sync();
#pragma unroll
    for (int j = 0; j < CS_RPT; ++j) t[j] = stage[cs_slot(CS_RPT * lane + j)];
}
__device__ __forceinline__ void ck_rows(const ExposureLoad &ld, int64_t wbase, int64_t n, int lane, double *stage,
                                        Quad (&q)[CS_RPT])
{
    double e[CS_RPT], d[CS_RPT];
    {
        double t0[CS_RPT], t1[CS_RPT], lt[CS_RPT];
        ck_load(ld.exposure, wbase, n, lane, e);
        ck_load(ld.tstart, wbase, n, lane, t0);
        ck_load(ld.tstop, wbase, n, lane, t1);
        ck_load(ld.livetime, wbase, n, lane, lt);
#pragma unroll
        for (int j = 0; j < CS_RPT; ++j) d[j] = __dsub_rn(__dsub_rn(t1[j], t0[j]), lt[j]);   // as ExposureLoad
    }
    ck_turn(e, lane, stage);
    ck_turn(d, lane, stage);
    if (ld.sexposure) {
        double se[CS_RPT], be[CS_RPT];
        ck_load(ld.sexposure, wbase, n, lane, se);
        ck_load(ld.bexposure, wbase, n, lane, be);
        ck_turn(se, lane, stage);
        ck_turn(be, lane, stage);
#pragma unroll
        for (int j = 0; j < CS_RPT; ++j) q[j] = Quad{e[j], se[j], be[j], d[j]};
    } else {
#pragma unroll
        for (int j = 0; j < CS_RPT; ++j)
            q[j] = Quad{e[j], __dmul_rn(ld.s_over_e, e[j]), __dmul_rn(ld.b_over_e, e[j]), d[j]};
    }
}
template <int CTRL>
__device__ __forceinline__ int64_t dpp_i64(int64_t v)
{
    const int lo = __builtin_amdgcn_mov_dpp((int)v, CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp((int)(v >> 32), CTRL, 0xf, 0xf, true);
    return ((int64_t)hi << 32) | (uint32_t)lo;
}
__device__ __forceinline__ int64_t lane_i64(int64_t v, int l)
{
    const int lo = __builtin_amdgcn_readlane((int)v, l);
    const int hi = __builtin_amdgcn_readlane((int)(v >> 32), l);
    return ((int64_t)hi << 32) | (uint32_t)lo;
}
GODOT_TEAM_REDUCE(team_sum_i64, int64_t, dpp_i64, lane_i64, GODOT_OP_ADD)

// Running sum over the lanes of a wavefront, every lane active: inside the rows of sixteen
// lanes by DPP shifts (zeros come in at the row's start), then the totals of the rows before
// a lane's own, read as scalars.
template <int N>
__device__ __forceinline__ int64_t dpp_row_shr_i64(int64_t v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, (int)v, 0x110 + N, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(v >> 32), 0x110 + N, 0xf, 0xf, true);
    return ((int64_t)hi << 32) | (uint32_t)lo;
}
__device__ __forceinline__ int64_t wave_running_sum_i64(int64_t v, int lane)
{
    v += dpp_row_shr_i64<1>(v);
    v += dpp_row_shr_i64<2>(v);
    v += dpp_row_shr_i64<4>(v);
    v += dpp_row_shr_i64<8>(v);
    const int64_t r0 = lane_i64(v, 15), r1 = lane_i64(v, 31), r2 = lane_i64(v, 47);
    const int row = lane >> 4;
    return v + (row == 0 ? 0 : (row == 1 ? r0 : (row == 2 ? r0 + r1 : r0 + r1 + r2)));
}

// element_adder where x / u is no tie (all but one row in some 2^24 of a real table): the adder
// is the plain integer r whatever the parity of the sum, and a run of such adders composes to
// the sum of their r.  Reports the tie, and x that does not qualify.
__device__ __forceinline__ void element_round(double x, int k, int64_t &r, bool &tie, bool &bad)
{
    constexpr int64_t MANT = ((int64_t)1 << 52) - 1;
    const int64_t b = __double_as_longlong(x);
    const int e = (int)(b >> 52) & 0x7ff;
    const int64_t M = e ? ((b & MANT) | (MANT + 1)) : (b & MANT);
    const int sh = k + 1023 - (e ? e : 1);
    const int shc = sh > 63 ? 63 : (sh < 0 ? 0 : sh);
    const int64_t X0 = M >> shc;
    const uint64_t rem2 = (uint64_t)(M - (X0 << shc)) << 1, unit = (uint64_t)1 << shc;
    r = X0 + (rem2 > unit ? 1 : 0);
    tie = rem2 == unit;
    bad = !(x >= 0.0) || sh < 0;
}
// a thread's eight rows of series c as one adder; wavefronts without a tie take the integer way
__device__ __forceinline__ Adder ck_thread_adder(const Quad (&q)[CS_RPT], int c, int k, bool &plain, bool &bad,
                                                 bool &nonzero)
{
    int64_t sum = 0;
    bool tie = false;
    bad = nonzero = false;
#pragma unroll
    for (int j = 0; j < CS_RPT; ++j) {
        const double x = quad_get(q[j], c);
        int64_t r;
        bool t, b;
        element_round(x, k, r, t, b);
        sum += r;
        tie = tie || t;
        bad = bad || b;
        nonzero = nonzero || x != 0.0;
    }
    plain = !__any(tie);
    if (plain) return Adder{sum, sum};
    // the rare way: its arithmetic hangs on a value the compiler cannot see through, or it would
    // be done ahead of the branch, for every wavefront
    int k_tie = k;
    asm volatile("; a tie in the wavefront" : "+v"(k_tie));
    Adder f{0, 0};
#pragma unroll
    for (int j = 0; j < CS_RPT; ++j) {
        Adder g{0, 0};
        element_adder(quad_get(q[j], c), k_tie, g);
        f = adder_then(f, g);
    }
    return f;
}
__global__ __launch_bounds__(CS_THREADS) void ck_adders_kernel(const CkSource *src, int64_t ntab)
{
    __shared__ Adder sm[4][CS_THREADS / WAVE];
    __shared__ int sm_flag[4][2];
    __shared__ double stage[CS_THREADS / WAVE][CK_STAGE];
    const CkSource *s;
    int64_t chunk;
    if (!ck_which(src, ntab, s, chunk)) return;
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid / WAVE;
    Quad q[CS_RPT];
    {
        const ExposureLoad ld = s->ld;
        ck_rows(ld, chunk * CS_CHUNK + (int64_t)wave * WAVE * CS_RPT, s->n, lane, stage[wave], q);
    }
    const Quad e = s->est[chunk];
    if (tid < 8) sm_flag[tid >> 1][tid & 1] = 1;
    __syncthreads();
    int kk[4];
    bool hk[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const double ec = quad_get(e, c);
        hk[c] = ec > 0.0 && ec < INFINITY;
        kk[c] = hk[c] ? ilogb(ec) : 0;
        bool plain, bad, nonzero;
        Adder f = ck_thread_adder(q, c, kk[c], plain, bad, nonzero);
        if (plain) {
            // no tie in the wavefront: its adders compose to their sum
            f.even = f.odd = team_sum_i64<WAVE>(f.even);
        } else {
            // in order: lane i takes lane i + d after itself
            asm volatile("; a tie in the wavefront" ::: "memory");
#pragma unroll 1
            for (int d = 1; d < WAVE; d <<= 1) f = adder_then(f, adder_shfl_down(f, d));
        }
        if (__any(bad) && lane == 0) sm_flag[c][0] = 0;
        if (__any(nonzero) && lane == 0) sm_flag[c][1] = 0;
        if (lane == 0) sm[c][wave] = f;
    }
    __syncthreads();
    if (tid < 4) {
        const int c = tid;
        const int k = c == 0 ? kk[0] : (c == 1 ? kk[1] : (c == 2 ? kk[2] : kk[3]));
        const bool have_k = c == 0 ? hk[0] : (c == 1 ? hk[1] : (c == 2 ? hk[2] : hk[3]));
        Adder f = sm[c][0];
        for (int w = 1; w < CS_THREADS / WAVE; ++w) f = adder_then(f, sm[c][w]);
        ChunkInfo ci;
        if (sm_flag[c][1]) {
            ci = ChunkInfo{0, 0, CS_K_ANY, 1};
        } else {
            const bool reg = have_k && sm_flag[c][0] && f.even < CS_2P53 && f.odd < CS_2P53 && f.even >= 0 &&
                             f.odd >= 0;
            ci = ChunkInfo{f.even, f.odd, k, reg ? 1 : 0};
        }
        s->info[chunk * 4 + c] = ci;
    }
}
__global__ __launch_bounds__(CS_THREADS) void ck_walk_kernel(const CkSource *src)
{
    const CkSource &s = src[blockIdx.y];
    if (s.nchunk == 0) return;
    cs_walk_body<true>(s.ld, s.n, s.nchunk, s.info, s.entry, s.by_walk, s.ck, blockIdx.x);
}
// the sums after every sixteenth row of a chunk from its exact start: odd threads end such a
// row, and so does the thread that holds the table's last row
__global__ __launch_bounds__(CS_THREADS) void ck_fill_kernel(const CkSource *src, int64_t ntab)
{
    __shared__ Adder sm[4][CS_THREADS / WAVE];
    __shared__ double stage[CS_THREADS / WAVE][CK_STAGE];
    const CkSource *s;
    int64_t chunk;
    if (!ck_which(src, ntab, s, chunk)) return;
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid / WAVE;
    const int64_t first = chunk * CS_CHUNK + (int64_t)tid * CS_RPT;
    const int64_t n = s->n;
    Quad *ck = s->ck;
    // what the walk left for the chunk, asked for before the rows so that it arrives with them
    const int4 walked = *(const int4 *)(s->by_walk + chunk * 4);
    const double4 start = *(const double4 *)(s->entry + chunk * 4);
    const ChunkInfo *info = s->info + chunk * 4;
    const int k0 = info[0].k, k1 = info[1].k, k2 = info[2].k, k3 = info[3].k;
    Quad q[CS_RPT];
    {
        const ExposureLoad ld = s->ld;
        ck_rows(ld, chunk * CS_CHUNK + (int64_t)wave * WAVE * CS_RPT, n, lane, stage[wave], q);
    }
    const bool last_row = first < n && n - 1 < first + CS_RPT;
    const bool writes = ((tid & 1) || last_row) && first < n;
    // series c of the chunk: the sum after this thread's rows, if the walk left the chunk to us
    auto series = [&](int c, int by_walk, int k, double s0, double &out) -> bool {
        if (by_walk) return false;                    // uniform over the workgroup
        if (k == CS_K_ANY) {
            out = s0;
            return true;
        }
        bool plain, bad, nonzero;
        Adder f = ck_thread_adder(q, c, k, plain, bad, nonzero);
        // inclusive scan over the wavefront, order kept
        if (plain) {
            f.even = f.odd = wave_running_sum_i64(f.even, lane);
        } else {
            asm volatile("; a tie in the wavefront" ::: "memory");
#pragma unroll 1
            for (int d = 1; d < WAVE; d <<= 1) {
                const Adder up = adder_shfl_up(f, d);
                if (lane >= d) f = adder_then(up, f);
            }
        }
        if (lane == WAVE - 1) sm[c][wave] = f;
        __syncthreads();
        if (writes) {
            int64_t S = (int64_t)ldexp(s0, 52 - k);
            for (int w = 0; w < wave; ++w) S = adder_apply(S, sm[c][w]);
            S = adder_apply(S, f);
            out = ldexp((double)S, k - 52);
        }
        return true;
    };
    double o0 = 0, o1 = 0, o2 = 0, o3 = 0;
    const bool d0 = series(0, walked.x, k0, start.x, o0);
    const bool d1 = series(1, walked.y, k1, start.y, o1);
    const bool d2 = series(2, walked.z, k2, start.z, o2);
    const bool d3 = series(3, walked.w, k3, start.w, o3);
    if (!writes) return;
    const int64_t row = first / CK_ROWS;
    if (d0 && d1 && d2 && d3) {
        ck[row] = Quad{o0, o1, o2, o3};
    } else {
        double *ckd = (double *)ck + row * 4;
        if (d0) ckd[0] = o0;
        if (d1) ckd[1] = o1;
        if (d2) ckd[2] = o2;
        if (d3) ckd[3] = o3;
    }
}
// Tables whose running sums are formed serially, bit for bit as numpy adds them: one wavefront a
// table.  All 64 lanes fetch a tile of rows (the next tile's loads are in flight while the
// present one is walked) and lay its four series down in LDS; lanes 0..3 then walk one series
// each, sixteen rows to a checkpoint, two register batches so that the LDS reads of one are under
// way while the other goes through the chain of additions.  A wavefront takes a dozen cycles
// a row (measured: 7 to 11 ns) whatever else the chip does, and a thousand tables are a thousand
// wavefronts that walk at the same time.
// The sums start from -0.0 and the rows past the end of a table count as -0.0: s + (-0.0) = s
// for every s, and -0.0 + x = x for every x, the first row included.
constexpr int CKW_TILE = 512;
constexpr int CKW_RPL = CKW_TILE / 64;   // rows of a tile a lane fetches
__global__ __launch_bounds__(64) void ck_wave_kernel(const CkSource *src)
{
    __shared__ __align__(16) double s_x[2][4][CKW_TILE];
    const CkSource &s = src[blockIdx.x];
    if (s.nchunk > 0) return;   // the parallel steps build this one
    const int lane = threadIdx.x;
    const int64_t n = s.n, ntile = (n + CKW_TILE - 1) / CKW_TILE;
    const ExposureLoad ld = s.ld;
    Quad q[CKW_RPL];
    auto fetch = [&](int64_t tile) {
        const int64_t base = tile * CKW_TILE;
#pragma unroll
        for (int j = 0; j < CKW_RPL; ++j) {
            const int64_t r = base + lane + 64 * j;
            q[j] = Quad{-0.0, -0.0, -0.0, -0.0};
            if (r < n) q[j] = ld(r);
        }
    };
    auto put = [&](int buf) {
#pragma unroll
        for (int j = 0; j < CKW_RPL; ++j) {
            const int k = lane + 64 * j;
            s_x[buf][0][k] = q[j].a;
            s_x[buf][1][k] = q[j].b;
            s_x[buf][2][k] = q[j].c;
            s_x[buf][3][k] = q[j].d;
        }
    };
    // LDS serves a wavefront's accesses in order: only the compiler has to be kept from reordering
    auto fence = [] { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); };
    fetch(0);
    put(0);
    double sum = -0.0;
    double *out = (double *)s.ck;
    for (int64_t t = 0; t < ntile; ++t) {
        const int buf = (int)(t & 1);
        if (t + 1 < ntile) fetch(t + 1);
        fence();
        if (lane < 4) {
            const double *row = s_x[buf][lane];
            const int64_t base = t * CKW_TILE;
            const int m = (int)((n - base) < CKW_TILE ? (n - base) : CKW_TILE);
            double *o = out + (base / CK_ROWS) * 4 + lane;
            double a[16], b[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) a[u] = row[u];
            for (int k = 0; k < m; k += 32) {
#pragma unroll
                for (int u = 0; u < 16; ++u) b[u] = row[k + 16 + u];
#pragma unroll
                for (int u = 0; u < 16; ++u) sum = __dadd_rn(sum, a[u]);
                o[(k / CK_ROWS) * 4] = sum;
                if (k + 32 < CKW_TILE) {
#pragma unroll
                    for (int u = 0; u < 16; ++u) a[u] = row[k + 32 + u];
                }
                if (k + 16 < m) {
#pragma unroll
                    for (int u = 0; u < 16; ++u) sum = __dadd_rn(sum, b[u]);
                    o[(k / CK_ROWS + 1) * 4] = sum;
                }
            }
        }
        fence();
        if (t + 1 < ntile) put(buf ^ 1);
    }
}

// Pass two: the photon range of every cell, the exposures between its edges (differences of
// pass one's sums, as exposure_out_kernel forms them), its edges and its source.
__global__ __launch_bounds__(256) void sources_cells_kernel(const godot_source_cells *src, int64_t first,
                                                            const Quad *at, double *starts, double *stops,
                                                            double *o_exp, double *o_sexp, double *o_bexp,
                                                            double *o_dead, int64_t *lo, int64_t *hi,
                                                            int64_t *source)
{
    const int64_t k = first + blockIdx.y;
    const godot_source_cells s = src[k];
    if ((int64_t)blockIdx.x * blockDim.x >= s.ncell) return;   // the whole workgroup
    const int64_t o = s.cell_offset;
    const UniformEdge edge{s.t0, s.tcell, 0};
    bin_photons_body(s.ti, s.nph, edge(0), UniformEdge{s.t0, s.tcell, 1}, s.ncell, lo + o, hi + o,
                     (int64_t *)nullptr, s.photon_offset);
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= s.ncell) return;
    const Quad *a = at + o + k;
    const Quad v = a[c], u = a[c + 1];
    o_exp[o + c] = u.a - v.a;
    o_sexp[o + c] = u.b - v.b;
    o_bexp[o + c] = u.c - v.c;
    if (o_dead) o_dead[o + c] = u.d - v.d;
    if (starts) starts[o + c] = edge(c);
    if (stops) stops[o + c] = edge(c + 1);
    if (source) source[o + c] = k;
}

int exposure_table_impl(const ExposureLoad &ld, int64_t n_ft2, void *table, size_t table_bytes,
                        godot_stream_t stream, const char *fn)
{
    GODOT_REQUIRE(n_ft2 > 0, GODOT_E_INVALID, "%s: n_ft2 must be positive", fn);
    GODOT_REQUIRE(ld.tstart && ld.tstop && ld.exposure && ld.livetime, GODOT_E_INVALID,
                  "%s: null table pointer", fn);
    GODOT_REQUIRE(table && table_bytes >= godot_exposure_table_bytes(n_ft2), GODOT_E_WORKSPACE,
                  "%s: table smaller than godot_exposure_table_bytes(n_ft2)", fn);
    hipStream_t st = (hipStream_t)stream;
    Quad *cum = (Quad *)table;
    const size_t tsz = table_sums_bytes(n_ft2);
    hipLaunchKernelGGL(exposure_rows_kernel, dim3((unsigned)ceil_div(n_ft2, 256)), dim3(256), 0, st, ld, n_ft2,
                       (ExpRow *)((char *)table + tsz));
    if (n_ft2 < 4 * CS_CHUNK) {
        // short tables: one workgroup, serially
        hipLaunchKernelGGL(exposure_cumsum_kernel, dim3(1), dim3(SEQ_THREADS), 0, st, ld, n_ft2, cum);
        GODOT_LAUNCH_CHECK(fn);
        return GODOT_OK;
    }
    const int64_t nc = cs_chunks(n_ft2);
    char *w = (char *)table + tsz + table_rows_bytes(n_ft2);
    Quad *totals = (Quad *)w;
    w += align256(sizeof(Quad) * (size_t)nc);
    Quad *est = (Quad *)w;
    w += align256(sizeof(Quad) * (size_t)nc);
    ChunkInfo *info = (ChunkInfo *)w;
    w += align256(sizeof(ChunkInfo) * 4 * (size_t)nc);
    double *entry = (double *)w;
    w += align256(sizeof(double) * 4 * (size_t)nc);
    int32_t *by_walk = (int32_t *)w;
    hipLaunchKernelGGL(cs_totals_kernel, dim3((unsigned)nc), dim3(CS_THREADS), 0, st, ld, n_ft2, totals);
    hipLaunchKernelGGL(cs_estimate_kernel, dim3(1), dim3(1024), 0, st, (const Quad *)totals, nc, est);
    hipLaunchKernelGGL(cs_adders_kernel, dim3((unsigned)nc), dim3(CS_THREADS), 0, st, ld, n_ft2, (const Quad *)est,
                       info);
    hipLaunchKernelGGL(cs_walk_kernel, dim3(4), dim3(CS_THREADS), 0, st, ld, n_ft2, nc, (const ChunkInfo *)info,
                       entry, by_walk, cum);
    hipLaunchKernelGGL(cs_fill_kernel, dim3((unsigned)nc), dim3(CS_THREADS), 0, st, ld, n_ft2,
                       (const ChunkInfo *)info, (const double *)entry, (const int32_t *)by_walk, cum);
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
    // nothing to write: no query times, or a single one with differences asked for
    const int64_t n_out = cumulative ? n_t : n_t - 1;
    if (n_out <= 0) return GODOT_OK;
    GODOT_REQUIRE(times && out_exp && out_sexp && out_bexp, GODOT_E_INVALID, "%s: null query/output pointer", fn);
    GODOT_REQUIRE(workspace && workspace_bytes >= align256(sizeof(Quad) * (size_t)n_t), GODOT_E_WORKSPACE,
                  "%s: workspace smaller than godot_exposure_workspace_bytes(n_t)", fn);
    hipStream_t st = (hipStream_t)stream;
    Quad *at = (Quad *)workspace;
    hipLaunchKernelGGL(exposure_query_kernel, dim3((unsigned)ceil_div(n_t, 256)), dim3(256), 0, st, ld,
                       (const Quad *)table, (const ExpRow *)((const char *)table + table_sums_bytes(n_ft2)),
                       n_ft2, times, n_t, at);
    hipLaunchKernelGGL(exposure_out_kernel, dim3((unsigned)ceil_div(n_out, 256)), dim3(256), 0, st,
                       (const Quad *)at, n_out, cumulative, out_exp, out_sexp, out_bexp, out_dead);
    GODOT_LAUNCH_CHECK(fn);
    return GODOT_OK;
}

}   // namespace

extern "C" size_t godot_exposure_table_bytes(int64_t n_ft2)
{
    // the running sums, the rows as the queries read them, the per-chunk records of the parallel scan
    return n_ft2 < 0 ? 0 : table_sums_bytes(n_ft2) + table_rows_bytes(n_ft2) + cs_scratch_bytes(n_ft2);
}

extern "C" int godot_exposure_table(const double *tstart, const double *tstop, const double *exposure,
                                    const double *livetime, int64_t n_ft2, double s_over_e,
                                    double b_over_e, void *table, size_t table_bytes,
                                    godot_stream_t stream)
{
    GODOT_ENTRY();
    ExposureLoad ld{tstart, tstop, exposure, livetime, s_over_e, b_over_e};
    return exposure_table_impl(ld, n_ft2, table, table_bytes, stream, "godot_exposure_table");
}

extern "C" int godot_exposure_table_rows(const double *tstart, const double *tstop, const double *exposure,
                                         const double *sexposure, const double *bexposure,
                                         const double *livetime, int64_t n_ft2, void *table,
                                         size_t table_bytes, godot_stream_t stream)
{
    GODOT_ENTRY();
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
    GODOT_ENTRY();
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
    GODOT_ENTRY();
    GODOT_REQUIRE(sexposure && bexposure, GODOT_E_INVALID, "godot_exposure_at_rows: null scaled exposure");
    ExposureLoad ld{tstart, tstop, exposure, livetime, 0.0, 0.0, sexposure, bexposure};
    return exposure_at_impl(ld, n_ft2, table, times, n_t, cumulative, out_exp, out_sexp, out_bexp, out_dead,
                            workspace, workspace_bytes, stream, "godot_exposure_at_rows");
}

extern "C" int godot_search_sorted(const double *sorted, int64_t n, const double *values, int64_t nv,
                                   int right, int64_t *out, godot_stream_t stream)
{
    GODOT_ENTRY();
    GODOT_REQUIRE(n >= 0 && nv >= 0, GODOT_E_INVALID, "godot_search_sorted: negative size");
    if (nv == 0) return GODOT_OK;
    GODOT_REQUIRE((sorted || n == 0) && values && out, GODOT_E_INVALID, "godot_search_sorted: null argument");
    hipLaunchKernelGGL(search_sorted_kernel, dim3((unsigned)ceil_div(nv, 256)), dim3(256), 0,
                       (hipStream_t)stream, sorted, n, values, nv, right, out);
    GODOT_LAUNCH_CHECK("godot_search_sorted");
    return GODOT_OK;
}

extern "C" int godot_piecewise_average(const double *starts, const double *stops, const double *scales,
                                       const double *prod, const double *cprod, int64_t nseg, const double *t0,
                                       const double *t1, int64_t n, double *out, godot_stream_t stream)
{
    GODOT_ENTRY();
    GODOT_REQUIRE(nseg > 0 && n >= 0, GODOT_E_INVALID, "godot_piecewise_average: bad sizes");
    if (n == 0) return GODOT_OK;
    GODOT_REQUIRE(starts && stops && scales && prod && cprod && t0 && t1 && out, GODOT_E_INVALID,
                  "godot_piecewise_average: null argument");
    hipLaunchKernelGGL(piecewise_average_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0,
                       (hipStream_t)stream, starts, stops, scales, prod, cprod, nseg, t0, t1, n, out);
    GODOT_LAUNCH_CHECK("godot_piecewise_average");
    return GODOT_OK;
}

extern "C" int godot_piecewise_lookup(const double *stops, const double *scales, int64_t nseg,
                                      const double *t, int64_t n, double *out, godot_stream_t stream)
{
    GODOT_ENTRY();
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
    GODOT_ENTRY();
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
    GODOT_ENTRY();
    GODOT_REQUIRE(nph >= 0 && ncell >= 0, GODOT_E_INVALID, "godot_bin_photons: negative size");
    if (ncell == 0) return GODOT_OK;
    GODOT_REQUIRE((ti || nph == 0) && stops && lo && hi, GODOT_E_INVALID,
                  "godot_bin_photons: null argument");
    hipLaunchKernelGGL(bin_photons_kernel, dim3((unsigned)ceil_div(ncell, 256)), dim3(256), 0,
                       (hipStream_t)stream, ti, nph, start0, stops, ncell, lo, hi, counts);
    GODOT_LAUNCH_CHECK("godot_bin_photons");
    return GODOT_OK;
}

extern "C" size_t godot_bin_sources_workspace_bytes(int64_t nsrc, int64_t total_cells)
{
    if (nsrc < 0 || total_cells < 0) return 0;
    return align256(sizeof(godot_source_cells) * (size_t)nsrc) + align256(sizeof(Quad) * (size_t)(total_cells + nsrc)) +
           align256(sizeof(int64_t) * bracket_entries(nsrc, total_cells));
}

extern "C" int godot_bin_sources(const godot_source_cells *sources, int64_t nsrc, int64_t total_cells,
                                 double *starts, double *stops, double *out_exp, double *out_sexp,
                                 double *out_bexp, double *out_dead, int64_t *lo, int64_t *hi, int64_t *source,
                                 void *workspace, size_t workspace_bytes, godot_stream_t stream)
{
    GODOT_ENTRY();
    GODOT_REQUIRE(nsrc >= 0 && total_cells >= 0, GODOT_E_INVALID, "godot_bin_sources: negative size");
    if (nsrc == 0) return GODOT_OK;
    GODOT_REQUIRE(sources, GODOT_E_INVALID, "godot_bin_sources: null descriptor table");
    // the descriptors are host memory: everything the kernels will index by is checked here
    int64_t end = 0, max_ncell = 0;
    for (int64_t k = 0; k < nsrc; ++k) {
        const godot_source_cells &s = sources[k];
        GODOT_REQUIRE(s.n_ft2 > 0 && s.nph >= 0 && s.ncell >= 0, GODOT_E_INVALID,
                      "godot_bin_sources: source %lld: bad sizes", (long long)k);
        GODOT_REQUIRE(s.tstart && s.tstop && s.exposure && s.livetime && s.table && (s.ti || s.nph == 0),
                      GODOT_E_INVALID, "godot_bin_sources: source %lld: null pointer", (long long)k);
        GODOT_REQUIRE((s.sexposure == nullptr) == (s.bexposure == nullptr), GODOT_E_INVALID,
                      "godot_bin_sources: source %lld: one rescaled exposure without the other", (long long)k);
        GODOT_REQUIRE(s.tcell > 0 && isfinite(s.tcell) && isfinite(s.t0), GODOT_E_INVALID,
                      "godot_bin_sources: source %lld: cell length must be positive and finite, the first edge finite",
                      (long long)k);
        GODOT_REQUIRE(s.cell_offset >= end && s.cell_offset + s.ncell <= total_cells, GODOT_E_INVALID,
                      "godot_bin_sources: source %lld: its cells [%lld, %lld) overlap the source before or pass "
                      "total_cells", (long long)k, (long long)s.cell_offset, (long long)(s.cell_offset + s.ncell));
        end = s.cell_offset + s.ncell;
        if (s.ncell > max_ncell) max_ncell = s.ncell;
    }
    if (max_ncell == 0) return GODOT_OK;
    GODOT_REQUIRE(out_exp && out_sexp && out_bexp && lo && hi, GODOT_E_INVALID, "godot_bin_sources: null output");
    GODOT_REQUIRE(workspace && workspace_bytes >= godot_bin_sources_workspace_bytes(nsrc, total_cells),
                  GODOT_E_WORKSPACE, "godot_bin_sources: workspace smaller than godot_bin_sources_workspace_bytes");
    hipStream_t st = (hipStream_t)stream;
    godot_source_cells *d_src = (godot_source_cells *)workspace;
    Quad *at = (Quad *)((char *)workspace + align256(sizeof(godot_source_cells) * (size_t)nsrc));
    int64_t *brackets = (int64_t *)((char *)at + align256(sizeof(Quad) * (size_t)(total_cells + nsrc)));
    // pageable host memory: the call returns once the table has been staged, the caller may reuse
    // it.  The runtime stages such a copy only after what was queued on the stream before it has
    // finished, so this call waits for the stream's earlier work (a caller that wants the two
    // launches queued behind running kernels passes a descriptor table in page-locked memory)
    GODOT_REQUIRE(hipMemcpyAsync(d_src, sources, sizeof(godot_source_cells) * (size_t)nsrc, hipMemcpyHostToDevice,
                                 st) == hipSuccess,
                  GODOT_E_LAUNCH, "godot_bin_sources: copy of the descriptor table failed");
    constexpr int64_t MAX_Y = 65535;
    for (int64_t first = 0; first < nsrc; first += MAX_Y) {
        const unsigned ny = (unsigned)(nsrc - first < MAX_Y ? nsrc - first : MAX_Y);
        const int64_t nwg = ceil_div(max_ncell + 1, SQ_THREADS);
        hipLaunchKernelGGL(sources_brackets_kernel, dim3((unsigned)ceil_div(nwg + 1, 256), ny), dim3(256), 0, st,
                           (const godot_source_cells *)d_src, first, brackets);
        GODOT_REQUIRE((nwg + 8) * (int64_t)ny < ((int64_t)1 << 31), GODOT_E_UNSUPPORTED,
                      "godot_bin_sources: %lld sources of up to %lld cells in one call", (long long)ny,
                      (long long)max_ncell);
        hipLaunchKernelGGL(sources_query_kernel, dim3((unsigned)(ceil_div(nwg, 8) * 8 * ny)), dim3(SQ_THREADS), 0, st,
                           (const godot_source_cells *)d_src, first, (int64_t)ny, at, (const int64_t *)brackets);
        hipLaunchKernelGGL(sources_cells_kernel, dim3((unsigned)ceil_div(max_ncell, 256), ny), dim3(256), 0, st,
                           (const godot_source_cells *)d_src, first, (const Quad *)at, starts, stops, out_exp,
                           out_sexp, out_bexp, out_dead, lo, hi, source);
    }
    GODOT_LAUNCH_CHECK("godot_bin_sources");
    return GODOT_OK;
}

extern "C" size_t godot_exposure_checkpoints_bytes(int64_t n_ft2)
{
    return n_ft2 <= 0 ? 0 : align256(sizeof(Quad) * (size_t)ck_records(n_ft2));
}

// Which way a table's running sums are formed: by the parallel steps from four chunks on, serially
// by a wavefront of its own below that.  (Measured for tables of 32 440 rows: a thousand of them
// take 4.3 ms serially and 6.5 ms in parallel, 128 of them 1.09 and 0.89 ms, two 0.42 and 0.25 ms;
// from 131 072 rows on the parallel steps win whatever the number of tables.)
static bool ck_parallel(int64_t n) { return n >= 4 * CS_CHUNK; }

extern "C" size_t godot_exposure_checkpoints_workspace_bytes(const godot_exposure_rows *tables, int64_t ntab)
{
    if (!tables || ntab <= 0) return 0;
    size_t b = align256(sizeof(CkSource) * (size_t)ntab);
    for (int64_t k = 0; k < ntab; ++k)
        if (ck_parallel(tables[k].n_ft2)) b += cs_scratch_bytes(tables[k].n_ft2);
    return b;
}

extern "C" int godot_exposure_checkpoints(const godot_exposure_rows *tables, int64_t ntab, void *workspace,
                                          size_t workspace_bytes, godot_stream_t stream)
{
    GODOT_ENTRY();
    GODOT_REQUIRE(ntab >= 0, GODOT_E_INVALID, "godot_exposure_checkpoints: negative number of tables");
    if (ntab == 0) return GODOT_OK;
    GODOT_REQUIRE(tables, GODOT_E_INVALID, "godot_exposure_checkpoints: null descriptor table");
    GODOT_REQUIRE(workspace && workspace_bytes >= godot_exposure_checkpoints_workspace_bytes(tables, ntab),
                  GODOT_E_WORKSPACE,
                  "godot_exposure_checkpoints: workspace smaller than godot_exposure_checkpoints_workspace_bytes");
    std::vector<CkSource> src((size_t)ntab);
    char *w = (char *)workspace + align256(sizeof(CkSource) * (size_t)ntab);
    int64_t max_chunks = 0;
    bool any_short = false;
    for (int64_t k = 0; k < ntab; ++k) {
        const godot_exposure_rows &t = tables[k];
        GODOT_REQUIRE(t.n_ft2 > 0, GODOT_E_INVALID, "godot_exposure_checkpoints: table %lld: n_ft2 must be positive",
                      (long long)k);
        GODOT_REQUIRE(t.tstart && t.tstop && t.exposure && t.livetime && t.checkpoints, GODOT_E_INVALID,
                      "godot_exposure_checkpoints: table %lld: null pointer", (long long)k);
        GODOT_REQUIRE((t.sexposure == nullptr) == (t.bexposure == nullptr), GODOT_E_INVALID,
                      "godot_exposure_checkpoints: table %lld: one rescaled exposure without the other", (long long)k);
        CkSource &c = src[(size_t)k];
        c.ld = ExposureLoad{t.tstart, t.tstop, t.exposure, t.livetime, t.s_over_e, t.b_over_e, t.sexposure,
                            t.bexposure};
        c.n = t.n_ft2;
        c.ck = (Quad *)t.checkpoints;
        c.nchunk = 0;
        c.totals = c.est = nullptr;
        c.info = nullptr;
        c.entry = nullptr;
        c.by_walk = nullptr;
        if (!ck_parallel(t.n_ft2)) {
            any_short = true;
            continue;
        }
        const int64_t nc = cs_chunks(t.n_ft2);
        c.nchunk = nc;
        c.totals = (Quad *)w;
        w += align256(sizeof(Quad) * (size_t)nc);
        c.est = (Quad *)w;
        w += align256(sizeof(Quad) * (size_t)nc);
        c.info = (ChunkInfo *)w;
        w += align256(sizeof(ChunkInfo) * 4 * (size_t)nc);
        c.entry = (double *)w;
        w += align256(sizeof(double) * 4 * (size_t)nc);
        c.by_walk = (int32_t *)w;
        w += align256(sizeof(int32_t) * 4 * (size_t)nc);
        if (nc > max_chunks) max_chunks = nc;
    }
    hipStream_t st = (hipStream_t)stream;
    CkSource *d_src = (CkSource *)workspace;
    // pageable host memory: staged before the call returns, after the stream's earlier work
    GODOT_REQUIRE(hipMemcpyAsync(d_src, src.data(), sizeof(CkSource) * (size_t)ntab, hipMemcpyHostToDevice, st) ==
                      hipSuccess,
                  GODOT_E_LAUNCH, "godot_exposure_checkpoints: copy of the descriptor table failed");
    constexpr int64_t MAX_Y = 65535;
    for (int64_t first = 0; first < ntab; first += MAX_Y) {
        const int64_t nt = ntab - first < MAX_Y ? ntab - first : MAX_Y;
        const unsigned ny = (unsigned)nt;
        const CkSource *d = d_src + first;
        if (any_short) hipLaunchKernelGGL(ck_wave_kernel, dim3(ny), dim3(64), 0, st, d);
        if (max_chunks == 0) continue;
        GODOT_REQUIRE((max_chunks + 8) * nt < ((int64_t)1 << 31), GODOT_E_UNSUPPORTED,
                      "godot_exposure_checkpoints: %lld tables of up to %lld rows in one call", (long long)nt,
                      (long long)(max_chunks * CS_CHUNK));
        const unsigned nw = (unsigned)(ceil_div(max_chunks, 8) * 8 * nt);
        hipLaunchKernelGGL(ck_totals_kernel, dim3(nw), dim3(CS_THREADS), 0, st, d, nt);
        hipLaunchKernelGGL(ck_estimate_kernel, dim3(ny), dim3(1024), 0, st, d);
        hipLaunchKernelGGL(ck_adders_kernel, dim3(nw), dim3(CS_THREADS), 0, st, d, nt);
        hipLaunchKernelGGL(ck_walk_kernel, dim3(4, ny), dim3(CS_THREADS), 0, st, d);
        hipLaunchKernelGGL(ck_fill_kernel, dim3(nw), dim3(CS_THREADS), 0, st, d, nt);
    }
    GODOT_LAUNCH_CHECK("godot_exposure_checkpoints");
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
    GODOT_ENTRY();
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

// ---------------------------------------------------------------------------
// godot_select_cells for callers that cannot allocate: the tile counts of the
// scan live in a file-scope device array instead of a caller's workspace.
// ---------------------------------------------------------------------------
namespace {

constexpr int64_t SELECT_STATIC_MAX_TILES = 8192;     // tiles of SCAN_TILE = 2048 cells: 16 777 216 cells
constexpr int SELECT_STATIC_MAX_DEVICES = 64;

// 64 KB on every GPU the library runs on, until the process ends.  Each GPU holds its own
// copy at its own address; `scratch` itself is only the host-side handle of the symbol.
static __device__ __align__(256) int64_t scratch[SELECT_STATIC_MAX_TILES + 1];

// Device address of `scratch` on the current GPU, looked up once per device.
int select_static_scratch(int64_t **out)
{
    static std::mutex mu;
    static int64_t *addr[SELECT_STATIC_MAX_DEVICES] = {};
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) {
        godot_set_error("godot_select_cells_static: hipGetDevice: %s", hipGetErrorString(e));
        return GODOT_E_LAUNCH;
    }
    GODOT_REQUIRE(dev >= 0 && dev < SELECT_STATIC_MAX_DEVICES, GODOT_E_UNSUPPORTED,
                  "godot_select_cells_static: device index %d beyond %d", dev, SELECT_STATIC_MAX_DEVICES - 1);
    std::lock_guard<std::mutex> lock(mu);
    if (!addr[dev]) {
        void *p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p) {
            godot_set_error("godot_select_cells_static: hipGetSymbolAddress: %s", hipGetErrorString(e));
            return GODOT_E_LAUNCH;
        }
        addr[dev] = (int64_t *)p;
    }
    *out = addr[dev];
    return GODOT_OK;
}

}   // namespace

extern "C" int64_t godot_select_cells_static_max(void) { return SELECT_STATIC_MAX_TILES * SCAN_TILE; }

extern "C" int godot_select_cells_static(const double *exp, int64_t ncell, double threshold, int64_t *keep_idx,
                                         int64_t *n_keep, godot_stream_t stream)
{
    GODOT_ENTRY();
    GODOT_REQUIRE(ncell >= 0, GODOT_E_INVALID, "godot_select_cells_static: negative ncell");
    GODOT_REQUIRE(ncell <= SELECT_STATIC_MAX_TILES * SCAN_TILE, GODOT_E_UNSUPPORTED,
                  "godot_select_cells_static: %lld cells, the built-in scratch holds the counts of at most %lld "
                  "(use godot_select_cells with a workspace)", (long long)ncell,
                  (long long)(SELECT_STATIC_MAX_TILES * SCAN_TILE));
    GODOT_REQUIRE(n_keep, GODOT_E_INVALID, "godot_select_cells_static: null n_keep");
    hipStream_t st = (hipStream_t)stream;
    if (ncell == 0) {
        (void)hipMemsetAsync(n_keep, 0, sizeof(int64_t), st);
        return GODOT_OK;
    }
    GODOT_REQUIRE(exp && keep_idx, GODOT_E_INVALID, "godot_select_cells_static: null argument");
    int64_t *part = nullptr;
    int rc = select_static_scratch(&part);
    if (rc) return rc;
    // every entry the scan reads is written by the reduction before it: nothing has to
    // start at zero, whatever an earlier (larger) call left behind
    int64_t ntiles = ceil_div(ncell, SCAN_TILE);
    MaskLoad ld{exp, threshold};
    hipLaunchKernelGGL((tile_reduce_kernel<int64_t, MaskLoad>), dim3((unsigned)ntiles), dim3(SCAN_THREADS), 0,
                       st, ld, ncell, part);
    hipLaunchKernelGGL((scan_partials_kernel<int64_t>), dim3(1), dim3(SCAN_THREADS), 0, st, part, ntiles,
                       n_keep);
    hipLaunchKernelGGL((tile_scan_kernel<int64_t, MaskLoad, SelectStore>), dim3((unsigned)ntiles),
                       dim3(SCAN_THREADS), 0, st, ld, ncell, (const int64_t *)part,
                       SelectStore{exp, threshold, keep_idx});
    GODOT_LAUNCH_CHECK("godot_select_cells_static");
    return GODOT_OK;
}

extern "C" int godot_gather_f64(const double *src, const int64_t *idx, int64_t n, double *dst,
                                godot_stream_t stream)
{
    GODOT_ENTRY();
    GODOT_REQUIRE(n >= 0, GODOT_E_INVALID, "godot_gather_f64: negative n");
    if (n == 0) return GODOT_OK;
    GODOT_REQUIRE(src && idx && dst, GODOT_E_INVALID, "godot_gather_f64: null argument");
    hipLaunchKernelGGL(gather_kernel<double>, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0,
                       (hipStream_t)stream, src, idx, n, dst);
    GODOT_LAUNCH_CHECK("godot_gather_f64");
    return GODOT_OK;
}

extern "C" int godot_gather_columns(const void *const *src, void *const *dst, int ncol, const int64_t *idx,
                                    int64_t n, godot_stream_t stream)
{
    GODOT_ENTRY();
    GODOT_REQUIRE(n >= 0, GODOT_E_INVALID, "godot_gather_columns: negative n");
    GODOT_REQUIRE(ncol >= 0 && ncol <= GATHER_MAX_COLUMNS, GODOT_E_UNSUPPORTED,
                  "godot_gather_columns: at most 16 columns a call");
    if (n == 0 || ncol == 0) return GODOT_OK;
    GODOT_REQUIRE(src && dst && idx, GODOT_E_INVALID, "godot_gather_columns: null argument");
    GatherColumns c{};
    for (int a = 0; a < ncol; ++a) {
        GODOT_REQUIRE(src[a] && dst[a], GODOT_E_INVALID, "godot_gather_columns: null column");
        c.src[a] = (const uint64_t *)src[a];
        c.dst[a] = (uint64_t *)dst[a];
    }
    hipLaunchKernelGGL(gather_columns_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0,
                       (hipStream_t)stream, c, ncol, idx, n);
    GODOT_LAUNCH_CHECK("godot_gather_columns");
    return GODOT_OK;
}

extern "C" int godot_gather_i64(const int64_t *src, const int64_t *idx, int64_t n, int64_t *dst,
                                godot_stream_t stream)
{
    GODOT_ENTRY();
    GODOT_REQUIRE(n >= 0, GODOT_E_INVALID, "godot_gather_i64: negative n");
    if (n == 0) return GODOT_OK;
    GODOT_REQUIRE(src && idx && dst, GODOT_E_INVALID, "godot_gather_i64: null argument");
    hipLaunchKernelGGL(gather_kernel<int64_t>, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0,
                       (hipStream_t)stream, src, idx, n, dst);
    GODOT_LAUNCH_CHECK("godot_gather_i64");
    return GODOT_OK;
}
