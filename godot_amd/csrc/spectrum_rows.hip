// power_spectrum_fft (core.py:2683-2741) for long zero-padded series in TWO trips through
// memory (gfx950).
//
// A real series x of n samples, zero-padded to length L = R * B with R = 16 * R1 rows, is
// read as a matrix x[B r + b] of R rows and B columns.  With k = m + R p:
//
//   X[m + R p] = sum_b W_B^(b p) * [ W_L^(b m) * C_b[m] ],     C_b[m] = sum_r x[B r + b] W_R^(r m)
//
// Pass A (columns_kernel) transforms every column: the padding leaves only the first rows
// non-zero, r = r_lo + R1 r_hi with r_hi < nhi (nhi = 2 .. 7 for the default padding), so
// the sixteen values m mod 16 = c come from nhi samples each,
//   E_c[r_lo] = W_R^(r_lo c) sum_(r_hi < nhi) x[B (r_lo + R1 r_hi) + b] W_16^(r_hi c),
// and C_b[c + 16 m_hi] is the R1-point transform of E_c.  The column is real, so
// C_b[R - m] = conj C_b[m]: one complex transform per class c = 1 .. 7 yields the rows
// m = c + 16 m_hi (m_hi < R1/2) and, conjugated, the rows m = (16 - c) + 16 (R1 - 1 - m_hi)
// of the upper half of its output; the classes 0 and 8 are their own partners.  Only the
// rows m <= R/2 are kept: V[m][b] = W_L^(b m) C_b[m], (R/2 + 1) * B complex numbers, the same
// volume as the packed half-length transform.  A workgroup takes NC adjacent columns and
// one class, a wavefront per column (no workgroup barrier inside a transform); the samples
// were regrouped column by column beforehand (gather_kernel), and the nine workgroups that
// read the same columns follow one another on one XCD, whose L2 serves eight of the reads.
//
// Pass B (rows2048_kernel, rows_kernel) transforms rows: row m of V is contiguous, its
// B-point transform is X[m + R p], and because x is real X[(R - m) + R p'] = conj of the same
// row's output B - 1 - p'.  The estimators need the bins k <= l/4 only: outputs p < B/4 and
// p >= 3B/4 of a row.  The three second-moment series enter through the real parts of the
// even bins of their length-l transforms, i.e. through length-l/2 transforms with the same
// R rows of half the length, whose row m holds the same bins k.  A workgroup takes NR
// adjacent rows of all five series one series after the other, keeps what the estimators
// need of each in registers and writes the three spectra in runs of NR adjacent bins; no
// transform is ever written to memory.
#include "spectrum_common.hpp"

using namespace godot_spectrum;

#ifdef GODOT_FFT_STAMPS
__device__ unsigned long long g_stamps[2][4096][16];
#define STAMP(K, W, I) do { if ((W) < 4096 && threadIdx.x == 0) g_stamps[K][W][I] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP(K, W, I) do { } while (0)
#endif

namespace {

constexpr int R1 = 1024;        // points of a column transform after the pruned first stage
constexpr int ROWS = 16 * R1;   // R
constexpr int HALF = ROWS / 2;  // rows 0 .. HALF are stored

// exp(-2 pi i k / 16)
__device__ __forceinline__ cplx w16_any(int k)
{
    k &= 15;
    cplx w = w16(k & 7);
    return (k & 8) ? cplx{-w.x, -w.y} : w;
}

// v[k] *= w^k, k = 1 .. N-1; the powers one after the other (two registers instead of N:
// the rounding error grows to N ulp, far below what the spectra are held to)
template <int N>
__device__ __forceinline__ void twiddle_powers(cplx (&v)[N], cplx w)
{
    cplx p = w;
#pragma unroll
    for (int k = 1; k < N; ++k) {
        v[k] = cmul(v[k], p);
        if (k + 1 < N) p = cmul(p, w);
    }
}

// ---- transform of 16 T points by a team of T threads -----------------------------------
// in: thread t holds v[i] = x[t + T i].  Three stages (16, 16, T/16) with two exchanges
// through the team's LDS region; the result is left in the region in natural order and
// every thread of the workgroup (WAVE: of the wavefront) has passed the last barrier when the call returns (a team
// without a row of its own transforms zeros: a branch around the arithmetic would cost
// more registers than the arithmetic costs time).
template <int T>
struct Team {
    static constexpr int T3 = T / 16;
    static constexpr int RS = T3 == 1 ? 18 : ((T3 * 17 + 3) / 4) * 4;   // row stride of the exchanges
    static constexpr int REGION = 16 * RS;               // elements, >= 16 T
};

// What orders the LDS traffic of a team: a workgroup barrier, or, for a team that is one
// wavefront (whose LDS instructions execute in program order), nothing the hardware has to
// do: the fence only keeps the compiler from reordering across it.
template <bool WAVE>
__device__ __forceinline__ void team_sync()
{
    if constexpr (WAVE) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    } else {
        __syncthreads();
    }
}

template <int T, bool WAVE = false>
__device__ __forceinline__ void team_fft16(cplx (&v)[16], cplx *region, int t)
{
    static_assert(!WAVE || T == 64, "a wavefront team has 64 threads");
    constexpr int T3 = Team<T>::T3, RS = Team<T>::RS;
    // Two calls with the same t have the same twiddles; kept from one call to the next they
    // would hold over a hundred registers, so they are formed anew from an index the
    // compiler cannot tell from the last one.
    asm volatile("" : "+v"(t));
    dft<16>(v);
    twiddle_powers<16>(v, expi_cycles(-(double)t / (double)(16 * T)));
#pragma unroll
    for (int u1 = 0; u1 < 16; ++u1) region[u1 * RS + t] = v[u1];
    team_sync<WAVE>();
    const int t0 = t % T3, u1 = t / T3;
#pragma unroll
    for (int t1 = 0; t1 < 16; ++t1) v[t1] = region[u1 * RS + t0 + T3 * t1];
    dft<16>(v);
    if (T3 > 1) twiddle_powers<16>(v, expi_cycles(-(double)t0 / (double)T));
    team_sync<WAVE>();
#pragma unroll
    for (int u2 = 0; u2 < 16; ++u2) region[u2 * RS + t0 * 17 + u1] = v[u2];
    team_sync<WAVE>();
    const int u1b = t % 16, g = t / 16;
#pragma unroll
    for (int h = 0; h < 16 / T3; ++h)
#pragma unroll
        for (int q = 0; q < T3; ++q) v[h * T3 + q] = region[(g + T3 * h) * RS + q * 17 + u1b];
    if constexpr (T3 > 1) {
#pragma unroll
        for (int h = 0; h < 16 / T3; ++h) {
            cplx z[T3];
#pragma unroll
            for (int q = 0; q < T3; ++q) z[q] = v[h * T3 + q];
            dft<T3>(z);
#pragma unroll
            for (int q = 0; q < T3; ++q) v[h * T3 + q] = z[q];
        }
    }
    team_sync<WAVE>();
    // natural order: mu = u1 + 16 u2 + 256 u3, u2 = g + T3 h
#pragma unroll
    for (int h = 0; h < 16 / T3; ++h)
#pragma unroll
        for (int q = 0; q < T3; ++q) region[u1b + 16 * (g + T3 * h) + 256 * q] = v[h * T3 + q];
    team_sync<WAVE>();
}

// The same transform for a team that is one wavefront, with half the LDS: the real parts and
// then the imaginary parts go through the same 16 * RS doubles.  The sixteen results stay in
// registers: v[4 h + q] = X[u1 + 16 (g + 4 h) + 256 q], u1 = t % 16, g = t / 16.
__device__ __forceinline__ void wave_fft1024(cplx (&v)[16], double *region, int t)
{
    constexpr int T3 = 4, RS = Team<64>::RS;
    auto sync = [] {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    asm volatile("" : "+v"(t));
    dft<16>(v);
    twiddle_powers<16>(v, expi_cycles(-(double)t / 1024.0));
    const int t0 = t % T3, u1 = t / T3;
#pragma unroll
    for (int u = 0; u < 16; ++u) region[u * RS + t] = v[u].x;
    sync();
#pragma unroll
    for (int t1 = 0; t1 < 16; ++t1) v[t1].x = region[u1 * RS + t0 + T3 * t1];
    sync();
#pragma unroll
    for (int u = 0; u < 16; ++u) region[u * RS + t] = v[u].y;
    sync();
#pragma unroll
    for (int t1 = 0; t1 < 16; ++t1) v[t1].y = region[u1 * RS + t0 + T3 * t1];
    sync();
    dft<16>(v);
    twiddle_powers<16>(v, expi_cycles(-(double)t0 / 64.0));
    const int u1b = t % 16, g = t / 16;
#pragma unroll
    for (int u2 = 0; u2 < 16; ++u2) region[u2 * RS + t0 * 17 + u1] = v[u2].x;
    sync();
    double re[16];
#pragma unroll
    for (int h = 0; h < 4; ++h)
#pragma unroll
        for (int q = 0; q < T3; ++q) re[h * T3 + q] = region[(g + T3 * h) * RS + q * 17 + u1b];
    sync();
#pragma unroll
    for (int u2 = 0; u2 < 16; ++u2) region[u2 * RS + t0 * 17 + u1] = v[u2].y;
    sync();
#pragma unroll
    for (int h = 0; h < 4; ++h)
#pragma unroll
        for (int q = 0; q < T3; ++q) v[h * T3 + q] = cplx{re[h * T3 + q], region[(g + T3 * h) * RS + q * 17 + u1b]};
    sync();
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        cplx z[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) z[q] = v[h * 4 + q];
        dft<4>(z);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[h * 4 + q] = z[q];
    }
}

// ---- the samples, regrouped -----------------------------------------------------------------
// Pass A reads NC adjacent columns of every row: 32 bytes of each 128-byte line, nine times
// over (once per class).  gather_kernel first copies the series (differences formed on the
// way) column by column, XB[b][r], r < nhi * R1 (zeros past the end of the series), so that a
// wavefront of pass A, which owns one column, reads 512 contiguous bytes per instruction.
struct GatherSeries {
    const double *a;   // x = a - b (b may be null)
    const double *b;
    int64_t n_in;      // samples
    double *XB;        // [B][nhi * R1]
    int B;             // columns taken
    int rr;            // rows kept = nhi * R1
    int wg_end;        // workgroups of this and the earlier series
    int stride;        // columns of the whole series: sample (r, b) is element stride * r + b
    int bfirst;        // first column taken
};

struct GatherArgs {
    GatherSeries s[5];
};

constexpr int GC = 32, GR = 128;   // columns and rows of a workgroup's patch

// XB[b][r_hi][i / 2][t][i % 2] for row r = R1 r_hi + TA i + t: the two samples a lane of pass A
// takes for i and i + 1 are one 16-byte load.  A patch is the rows of two values of i.
__global__ __launch_bounds__(256) void gather_kernel(GatherArgs a)
{
    __shared__ double patch[GR][GC + 2];
    const int w = blockIdx.x;
    int si = 0;
#pragma unroll
    for (int i = 1; i < 5; ++i)
        if (w >= a.s[i - 1].wg_end) si = i;
    const GatherSeries &S = a.s[si];
    const int wl = w - (si ? a.s[si - 1].wg_end : 0);
    const int ncb = S.B / GC;
    const int b0 = (wl % ncb) * GC, r0 = (wl / ncb) * GR;
    const int tid = threadIdx.x;
    {
        const int col = tid & 31;
        for (int r = tid >> 5; r < GR; r += 8) {
            const int64_t n = (int64_t)S.stride * (r0 + r) + S.bfirst + b0 + col;
            double x = 0.0;
            if (n < S.n_in) {
                x = S.a[n];
                if (S.b) x -= S.b[n];
            }
            patch[r][col] = x;
        }
    }
    __syncthreads();
    // rows r0 .. r0 + 127 = (i, i + 1) x 64 lanes: element 2 t + (i & 1) of the pair's 128 doubles
    const int e = tid & 127;
    const int r = (e & 1) * 64 + (e >> 1);
#pragma unroll 4
    for (int col = tid >> 7; col < GC; col += 2) S.XB[(int64_t)(b0 + col) * S.rr + r0 + e] = patch[r][col];
}

// ---- pass A --------------------------------------------------------------------------------
struct ColumnSeries {
    const double *XB;  // [B][nhi * R1]
    int64_t L;         // padded length = ROWS * B
    cplx *V;           // [HALF + 1][B]
    int B;             // columns taken (the row length of V)
    int nhi;           // rows r_hi < nhi of sixteen hold samples
    int wg_end;        // workgroups of this and the earlier series
    int bfirst;        // first column taken: column b of V is column bfirst + b of the series
};

struct ColumnArgs {
    ColumnSeries s[5];
    int nseries;
    int nwg;           // workgroups with work (the grid is rounded up to a multiple of 8)
};

constexpr int NC = 8;                          // columns per workgroup: 128-byte pieces of a row
constexpr int TA = R1 / 16;                    // threads per column: one wavefront
static_assert(TA == 64, "columns_kernel gives every column to one wavefront");
constexpr int COL_REGION = 16 * Team<TA>::RS;  // doubles per wavefront: an exchange of real or imaginary parts,
                                               // or half of the column's results (512 complex numbers)
constexpr int COL_STAGE = R1 / 2 + 2;          // complex numbers per column while a half is written out:
                                               // the columns of one row on different banks
static_assert(2 * COL_STAGE <= COL_REGION, "half of a column's results fit its wavefront's region");

constexpr int NCL = 1;                         // classes per workgroup

__global__ __launch_bounds__(NCL *NC *TA, 2) void columns_kernel(ColumnArgs a)
{
    __shared__ double lds_all[NCL * NC * COL_REGION];
    // workgroups are dealt to the eight XCDs in turn: give each XCD a contiguous range
    const int per = gridDim.x >> 3;
    const int w = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (w >= a.nwg) return;        // uniform over the workgroup
    int si = 0;
#pragma unroll
    for (int i = 1; i < 5; ++i)
        if (i < a.nseries && w >= a.s[i - 1].wg_end) si = i;
    const ColumnSeries &S = a.s[si];
    const int wl = w - (si ? a.s[si - 1].wg_end : 0);
    // Two adjacent tiles follow each other class by class: their pieces of a row are written
    // together, and the samples are read again while the XCD's L2 still holds them.
    const int sub = 0, tid = threadIdx.x;
    const int c = (wl % 18) >> 1;                     // class: rows m = c and 16 - c (mod 16)
    const int tile = (wl / 18) * 2 + (wl & 1);
    const int b0 = tile * NC;
    double *lds = lds_all + sub * (NC * COL_REGION);
    const int jj = tid / TA, t = tid % TA;   // a wavefront per column: no workgroup barrier inside the transform
    STAMP(0, w, 0);
    const int B = S.B;
    cplx v[16];
    {
        // E_c[r_lo], r_lo = t + TA i
        const int nhi = S.nhi;
        const double *xb = S.XB + (int64_t)(b0 + jj) * nhi * R1 + 2 * t;
        double xt = -(double)((t * c) & (ROWS - 1)) / (double)ROWS;
        xt -= rint(xt);
        cplx tw = expi_cycles(xt);                                   // W_R^(t c)
        double xs = -(double)((TA * c) & (ROWS - 1)) / (double)ROWS;
        xs -= rint(xs);
        const cplx step = expi_cycles(xs);                           // W_R^(TA c)
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = cplx{0.0, 0.0};
        for (int rh = 0; rh < nhi; ++rh) {
            const cplx wr = w16_any(rh * c);     // W_16^(r_hi c)
            double x[16];
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                const double2 xx = *(const double2 *)(xb + TA * i + R1 * rh);   // samples i and i + 1
                x[i] = xx.x;
                x[i + 1] = xx.y;
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                v[i].x = fma(x[i], wr.x, v[i].x);
                v[i].y = fma(x[i], wr.y, v[i].y);
            }
        }
        STAMP(0, w, 1);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            v[i] = cmul(v[i], tw);
            tw = cmul(tw, step);
        }
    }
    STAMP(0, w, 2);
    wave_fft1024(v, lds + jj * COL_REGION, t);
    STAMP(0, w, 3);
    // v[4 h + q] = C[mu], mu = u1 + 16 (g + 4 h) + 256 q.  The halves mu < R1/2 (q < 2) and
    // mu >= R1/2 go through LDS one after the other: every wavefront lays its half down in
    // natural order, then adjacent lanes write the NC adjacent columns of a row.
    const int u1 = t % 16, g = t / 16;
    cplx *stage = (cplx *)lds;
    const int s = tid / NC, jw = tid % NC;
    const int64_t b = b0 + jw;             // column of V
    const int64_t bs = S.bfirst + b;       // column of the series: what the twiddles turn by
    const int64_t L = S.L;
    const double invL = 1.0 / (double)L;       // L is a power of two: exact
    cplx *V = S.V;
    double xs = -(double)((bs * (16 * TA)) & (L - 1)) * invL;      // rows step by 16 TA
    xs -= rint(xs);
    const cplx step = expi_cycles(xs);
    __syncthreads();   // the staged columns lie across the wavefronts' exchange regions: all transforms are done
    STAMP(0, w, 4);
#pragma unroll
    for (int h = 0; h < 4; ++h)
#pragma unroll
        for (int q = 0; q < 2; ++q) stage[jj * COL_STAGE + u1 + 16 * (g + 4 * h) + 256 * q] = v[h * 4 + q];
    __syncthreads();
    STAMP(0, w, 5);
    {
        // mu = s + TA h < R1/2: row c + 16 mu
        double x0 = -(double)((bs * (c + 16 * s)) & (L - 1)) * invL;
        x0 -= rint(x0);
        cplx wv = expi_cycles(x0);
#pragma unroll
        for (int h = 0; h < 8; ++h) {
            const int mu = s + TA * h;
            const cplx z = stage[jw * COL_STAGE + mu];
            V[(int64_t)(c + 16 * mu) * B + b] = cmul(z, wv);
            wv = cmul(wv, step);
        }
    }
    __syncthreads();
    STAMP(0, w, 6);
    // the upper half: classes 0 and 8 are their own partners and have nothing to write from it
    // but row R/2 (class 0), yet they go through the barrier with the others
#pragma unroll
    for (int h = 0; h < 4; ++h)
#pragma unroll
        for (int q = 2; q < 4; ++q) stage[jj * COL_STAGE + u1 + 16 * (g + 4 * h) + 256 * (q - 2)] = v[h * 4 + q];
    __syncthreads();
    STAMP(0, w, 7);
    if (c != 0 && c != 8) {
        // mu = R1/2 + s + TA h: conj -> row (16 - c) + 16 (R1 - 1 - mu); rows step down
        const int row8 = (16 - c) + 16 * (R1 / 2 - 1 - s);
        double x0 = -(double)((bs * row8) & (L - 1)) * invL;
        x0 -= rint(x0);
        cplx wv = expi_cycles(x0);
        const cplx back{step.x, -step.y};
#pragma unroll
        for (int h = 0; h < 8; ++h) {
            const cplx z = stage[jw * COL_STAGE + s + TA * h];
            V[(int64_t)(row8 - 16 * TA * h) * B + b] = cmul(cplx{z.x, -z.y}, wv);
            wv = cmul(wv, back);
        }
    } else if (c == 0 && s == 0) {
        // row R/2 = 16 * (R1/2): C[R1/2] is the first of the staged upper half
        const cplx z = stage[jw * COL_STAGE];
        double xh = -(double)((bs * HALF) & (L - 1)) * invL;
        xh -= rint(xh);
        V[(int64_t)HALF * B + b] = cmul(z, expi_cycles(xh));
    }
    STAMP(0, w, 8);
}

// ---- zero-frequency values of the second-moment series ----------------------------------
// X[0] = sum over row 0 of V (real: the twiddles of row 0 are 1)
__global__ __launch_bounds__(256) void rows_dc_kernel(const cplx *V3, const cplx *V4, const cplx *V5, int B2,
                                                      double *dc)
{
    __shared__ double red[3][256];
    const int r = threadIdx.x;
    double s3 = 0., s4 = 0., s5 = 0.;
    for (int b = r; b < B2; b += 256) {
        s3 += V3[b].x;
        s4 += V4[b].x;
        s5 += V5[b].x;
    }
    red[0][r] = s3;
    red[1][r] = s4;
    red[2][r] = s5;
    __syncthreads();
    for (int half = 128; half >= 1; half >>= 1) {
        if (r < half)
#pragma unroll
            for (int i = 0; i < 3; ++i) red[i][r] += red[i][r + half];
        __syncthreads();
    }
    if (r < 3) dc[r] = red[r][0];
}

// ---- pass B --------------------------------------------------------------------------------
struct RowArgs {
    const cplx *V1[2];   // (W - S), (Wb - B): [HALF + 1][B]
    const cplx *V2[3];   // WW, WbWb, W - WW:  [HALF + 1][B / 2]
    const double *dc;    // S2, B2, SB
    double *p0, *p1, *pb, *amps;
    int64_t nk;
    int ngroup;          // workgroups with work
    uint32_t stride2;    // elements from V2[0] to V2[1]
    int m_first, m_end;  // rows m_first <= m < m_end are transformed; the V arrays begin at row m_first
};

// `base` is the same for every lane (it stays in scalar registers), `off` the lane's element
template <int T>
__device__ __forceinline__ void row_load(cplx (&v)[16], const cplx *__restrict__ base, uint32_t off, bool valid)
{
    if (valid) {
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = base[off + (uint32_t)(T * i)];
    } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = cplx{0.0, 0.0};
    }
    __builtin_amdgcn_sched_barrier(0);   // what follows in the program stays behind these loads
}

// T = threads per row of a first-moment series (row length B = 16 T); NR = THREADS / T rows a workgroup
template <int T, int THREADS>
__global__ __launch_bounds__(THREADS) void rows_kernel(RowArgs a)
{
    constexpr int NR = THREADS / T, T2 = T / 2, B = 16 * T, B2 = 8 * T;
    constexpr int PAD = 16 / NR;   // rows of one bin on different banks when the estimators read
    constexpr int REG1 = Team<T>::REGION + PAD, REG2 = Team<T2>::REGION + PAD;
    constexpr int LDS_ELEMS = (NR * REG1 > 2 * NR * REG2) ? NR * REG1 : 2 * NR * REG2;
    __shared__ cplx lds[LDS_ELEMS];
    const int tid = threadIdx.x;
    // adjacent groups write adjacent bins: keep them on one XCD (its L2 merges the runs)
    const int per = gridDim.x >> 3;
    const int g = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (g >= a.ngroup) return;   // uniform
    const int l0 = g * NR, m0 = a.m_first + l0;     // first row of the group: in the arrays, of the transform
    // roles: a team of T2 threads per row of a second-moment series (two series at once),
    // a team of T threads per row of a first-moment series, and for the estimators row
    // rho = tid % NR (adjacent lanes = adjacent bins) with slot s = tid / NR
    const int vr = tid / T2, t2 = tid % T2, rho2 = vr % NR, ser2 = vr / NR;
    const int rho1 = tid / T, t1 = tid % T;
    const int rho = tid % NR, s = tid / NR;
    const bool valid2 = m0 + rho2 < a.m_end, valid1 = m0 + rho1 < a.m_end;
    const int m = m0 + rho;
    double g3[8], g4[8], g5[8];
    cplx f1[8];
    cplx v[16], nx[16];

    // direct bins p = s + T h (h < 4) read output p, mirrored bins output len - 1 - p
    // (the slot is handed over as a value the compiler cannot trace, or it forms every LDS
    // address of the kernel at the top and carries them, spilled, to where they are used)
    auto real_parts = [&](const cplx *reg, double (&gg)[8]) {
        int sl = s;
        asm volatile("" : "+v"(sl));
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const int p = sl + T * h;
            gg[h] = reg[p].x;
            gg[4 + h] = reg[B2 - 1 - p].x;
        }
    };

    STAMP(1, g, 0);
    row_load<T2>(v, a.V2[0], (uint32_t)ser2 * a.stride2 + (uint32_t)((l0 + rho2) * B2 + t2), valid2);
    row_load<T2>(nx, a.V2[2], (uint32_t)((l0 + rho2) * B2 + t2), valid2 && ser2 == 0);
    STAMP(1, g, 1);
    team_fft16<T2>(v, lds + vr * REG2, t2);
    STAMP(1, g, 2);
    real_parts(lds + rho * REG2, g3);
    real_parts(lds + (NR + rho) * REG2, g4);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = nx[i];
    row_load<T>(nx, a.V1[0], (uint32_t)((l0 + rho1) * B + t1), valid1);
    STAMP(1, g, 3);
    team_fft16<T2>(v, lds + vr * REG2, t2);
    STAMP(1, g, 4);
    real_parts(lds + rho * REG2, g5);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = nx[i];
    row_load<T>(nx, a.V1[1], (uint32_t)((l0 + rho1) * B + t1), valid1);
    STAMP(1, g, 5);
    team_fft16<T>(v, lds + rho1 * REG1, t1);
    STAMP(1, g, 6);
    {
        const cplx *reg = lds + rho * REG1;
        int sl = s;
        asm volatile("" : "+v"(sl));
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const int p = sl + T * h;
            f1[h] = reg[p];
            const cplx z = reg[B - 1 - p];
            f1[4 + h] = cplx{z.x, -z.y};
        }
    }
    __syncthreads();
    STAMP(1, g, 7);
    team_fft16<T>(nx, lds + rho1 * REG1, t1);
    STAMP(1, g, 8);
    if (m >= a.m_end) return;
    const double S2 = a.dc[0], B2v = a.dc[1], SB = a.dc[2];
    const cplx *reg = lds + rho * REG1;
    int sl = s;
    asm volatile("" : "+v"(sl));
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        const int p = sl + T * h;
        {
            const cplx f2 = reg[p];
            const int64_t k = m + (int64_t)ROWS * p;
            ps_estimators(k, a.nk, f1[h].x, -f1[h].y, f2.x, -f2.y, S2, B2v, SB, g3[h], g4[h], g5[h], a.p0, a.p1,
                          a.pb, a.amps);
        }
        // X[(R - m) + R p] = conj of output B - 1 - p; row 0 reaches only the bin l/4 this way,
        // row R/2 is its own mirror
        const bool mir = (m > 0 && m < HALF) || (m == 0 && p == B / 4 - 1);
        if (mir) {
            const cplx z = reg[B - 1 - p];
            const int64_t k = (ROWS - m) + (int64_t)ROWS * p;
            ps_estimators(k, a.nk, f1[4 + h].x, -f1[4 + h].y, z.x, z.y, S2, B2v, SB, g3[4 + h], g4[4 + h],
                          g5[4 + h], a.p0, a.p1, a.pb, a.amps);
        }
        __builtin_amdgcn_sched_barrier(0);   // one pair of bins at a time: interleaving them costs registers
    }
    STAMP(1, g, 9);
}

// Rows of 2 048 points (l = 2^25) with every transform done by ONE wavefront, so that no
// barrier of the workgroup stands inside a transform and the eight wavefronts drift apart:
// a row of a second-moment series (1 024 points) is one transform; a row of a first-moment
// series is split into its even and its odd samples, E and O, transformed by two wavefronts,
// and X[p] = E[p] + W_2048^p O[p], X[p + 1024] = E[p] - W_2048^p O[p] is formed where the
// estimators pick their bins up.  Barriers remain only between a transform and the reading
// of its results by the threads of the estimators (adjacent lanes = adjacent rows).
__global__ __launch_bounds__(512) void rows2048_kernel(RowArgs a)
{
    constexpr int NR = 4, T = 128, B = 2048, B2 = 1024;
    constexpr int REG = Team<64>::REGION + 4;   // rows of one bin on different banks when the estimators read
    __shared__ cplx lds[8 * REG];
    const int tid = threadIdx.x;
    // The last group holds row R/2 alone and runs through its phases faster than the others (one
    // wavefront of four has a row): it is the first workgroup, so that it is not the one the
    // launch ends on.  The others: adjacent groups, which write adjacent bins, on one XCD.
    const int per = (gridDim.x - 8) >> 3;
    int g;
    if (blockIdx.x < 8) {
        if (blockIdx.x) return;
        g = a.ngroup - 1;
    } else {
        const int w = blockIdx.x - 8;
        g = (w & 7) * per + (w >> 3);
        if (g >= a.ngroup - 1) return;   // uniform
    }
    const int l0 = g * NR, m0 = a.m_first + l0;     // first row of the group: in the arrays, of the transform
    // a wavefront: row m0 + wr of series (or parity) wh; a thread of the estimators: row
    // rho = tid % 4 with slot s = tid / 4
    const int wave = tid >> 6, lane = tid & 63, wr = wave & 3, wh = wave >> 2;
    const int rho = tid % NR, s = tid / NR;
    const bool valid = m0 + wr < a.m_end;
    const int m = m0 + rho;
    cplx *mine = lds + wave * REG;
    double g3[8], g4[8], g5[8];
    cplx f1[8];
    cplx v[16], nx[16];

    auto real_parts = [&](const cplx *reg, double (&gg)[8]) {
        int sl = s;
        asm volatile("" : "+v"(sl));
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const int p = sl + T * h;
            gg[h] = reg[p].x;
            gg[4 + h] = reg[B2 - 1 - p].x;
        }
    };
    // X[p] and conj X[B - 1 - p], p = s + T h, of a first-moment row from its halves
    auto combine = [&](cplx (&f)[8]) {
        int sl = s;
        asm volatile("" : "+v"(sl));
        const cplx *E = lds + rho * REG, *O = lds + (NR + rho) * REG;
        const cplx wd = expi_cycles(-(double)sl / 2048.0);            // W_2048^s
        const cplx wm = expi_cycles(-(double)(1023 - sl) / 2048.0);   // W_2048^(1023 - s)
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const int p = sl + T * h, q = 1023 - p;
            const cplx r = w16(h);                                   // W_2048^(128 h)
            const cplx od = cmul(O[p], h ? cmul(wd, r) : wd);
            f[h] = E[p] + od;
            const cplx om = cmul(O[q], h ? cmul(wm, cplx{r.x, -r.y}) : wm);
            const cplx xm = E[q] - om;
            f[4 + h] = cplx{xm.x, -xm.y};
        }
    };

    STAMP(1, g, 0);
    row_load<64>(v, a.V2[0], (uint32_t)wh * a.stride2 + (uint32_t)((l0 + wr) * B2 + lane), valid);
    row_load<64>(nx, a.V2[2], (uint32_t)((l0 + wr) * B2 + lane), valid && wh == 0);
    STAMP(1, g, 1);
    if (valid) team_fft16<64, true>(v, mine, lane);
    __syncthreads();
    STAMP(1, g, 2);
    real_parts(lds + rho * REG, g3);
    real_parts(lds + (NR + rho) * REG, g4);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = nx[i];
    // even (wh = 0) or odd (wh = 1) samples of the row
    row_load<128>(nx, a.V1[0], (uint32_t)((l0 + wr) * B + 2 * lane + wh), valid);
    STAMP(1, g, 3);
    if (wh == 0 && valid) team_fft16<64, true>(v, mine, lane);
    __syncthreads();
    STAMP(1, g, 4);
    real_parts(lds + rho * REG, g5);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = nx[i];
    row_load<128>(nx, a.V1[1], (uint32_t)((l0 + wr) * B + 2 * lane + wh), valid);
    STAMP(1, g, 5);
    if (valid) team_fft16<64, true>(v, mine, lane);
    __syncthreads();
    STAMP(1, g, 6);
    combine(f1);
    __syncthreads();
    STAMP(1, g, 7);
    if (valid) team_fft16<64, true>(nx, mine, lane);
    __syncthreads();
    STAMP(1, g, 8);
    if (m >= a.m_end) return;
    cplx f2[8];
    combine(f2);
    const double S2 = a.dc[0], B2v = a.dc[1], SB = a.dc[2];
    int sl = s;
    asm volatile("" : "+v"(sl));
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        const int p = sl + T * h;
        const int64_t k = m + (int64_t)ROWS * p;
        ps_estimators(k, a.nk, f1[h].x, -f1[h].y, f2[h].x, -f2[h].y, S2, B2v, SB, g3[h], g4[h], g5[h], a.p0, a.p1,
                      a.pb, a.amps);
        // X[(R - m) + R p] = conj of output B - 1 - p; row 0 reaches only the bin l/4 this way,
        // row R/2 is its own mirror
        const bool mir = (m > 0 && m < HALF) || (m == 0 && p == B / 4 - 1);
        if (mir) {
            const int64_t km = (ROWS - m) + (int64_t)ROWS * p;
            ps_estimators(km, a.nk, f1[4 + h].x, -f1[4 + h].y, f2[4 + h].x, -f2[4 + h].y, S2, B2v, SB, g3[4 + h],
                          g4[4 + h], g5[4 + h], a.p0, a.p1, a.pb, a.amps);
        }
        __builtin_amdgcn_sched_barrier(0);   // one pair of bins at a time: interleaving them costs registers
    }
    STAMP(1, g, 9);
}

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

}   // namespace

namespace godot_spectrum {

#ifndef ROWS_THREADS_1024
#define ROWS_THREADS_1024 256
#endif
#ifndef ROWS_THREADS_512
#define ROWS_THREADS_512 256
#endif

// The path takes transforms whose rows have 512, 1 024 or 2 048 points (l = 2^23, 2^24, 2^25) and
// series that need no wrapping (n <= l/2).
bool rows_path_applies(int64_t n, int64_t l)
{
    return (l == ((int64_t)1 << 23) || l == ((int64_t)1 << 24) || l == ((int64_t)1 << 25)) && n >= 1 && n <= l / 2;
}

size_t rows_path_workspace_bytes(int64_t l)
{
    // the stored rows of the five series, the three zero-frequency values, the regrouped samples
    // (at most one double per point of the padded lengths: 2 l + 3 l / 2)
    const size_t B = (size_t)(l / ROWS);
    return 2 * align256(sizeof(cplx) * (HALF + 1) * B) + 3 * align256(sizeof(cplx) * (HALF + 1) * (B / 2)) + 256 +
           sizeof(double) * (size_t)(2 * l + 3 * (l / 2));
}

// Pass A for the columns of part `part` of `nparts` equal parts (all of them: 0 of 1): the regrouped
// samples go to xb, the stored rows of series i to V[i] ([HALF + 1][columns of the part]); the order of
// the series: WW, WbWb, W - WW (second moments, l/2 points) then W - S, Wb - B.
void columns_part(const double *W, const double *WW, const double *S, const double *Wb, const double *WbWb,
                  const double *Bk, int64_t n, int64_t l, int part, int nparts, cplx *const (&V)[5], double *xb,
                  hipStream_t st)
{
    const int B = (int)(l / ROWS);
    GatherArgs ga{};
    ColumnArgs ca{};
    const double *sa[5] = {WW, WbWb, W, W, Wb};
    const double *sb[5] = {nullptr, nullptr, WW, S, Bk};
    int end = 0, gend = 0;
    for (int i = 0; i < 5; ++i) {
        const int64_t L = i < 3 ? l / 2 : l;
        const int Bs = i < 3 ? B / 2 : B;         // columns of the series
        const int Bp = Bs / nparts;               // columns of the part
        int nhi = (int)ceil_div(n, L / 16);
        if (nhi > 16) nhi = 16;
        end += (Bp / NC) * (9 / NCL);
        gend += (Bp / GC) * (nhi * R1 / GR);
        ga.s[i] = GatherSeries{sa[i], sb[i], n, xb, Bp, nhi * R1, gend, Bs, part * Bp};
        ca.s[i] = ColumnSeries{xb, L, V[i], Bp, nhi, end, part * Bp};
        xb += (size_t)Bp * nhi * R1;
    }
    hipLaunchKernelGGL(gather_kernel, dim3((unsigned)gend), dim3(256), 0, st, ga);
    ca.nseries = 5;
    ca.nwg = end;
    const unsigned gridA = (unsigned)(ceil_div(end, 8) * 8);
    hipLaunchKernelGGL(columns_kernel, dim3(gridA), dim3(NCL * NC * TA), 0, st, ca);
}

// Pass B for rows m_first <= m < m_end: V[i] holds those rows of series i (whole rows), dc the
// zero-frequency values of the three second-moment series.
void rows_part(const cplx *const (&V)[5], const double *dc, int64_t l, int m_first, int m_end, double *p0,
               double *p1, double *pb, double *amps, hipStream_t st)
{
    const int B = (int)(l / ROWS);
    const int nrow = m_end - m_first;
    RowArgs ra{{V[3], V[4]}, {V[0], V[1], V[2]}, dc, p0, p1, pb, amps, l / 4 + 1, 0, (uint32_t)(V[1] - V[0]),
               m_first, m_end};
    if (B == 2048) {
        ra.ngroup = (int)ceil_div(nrow, 4);
        hipLaunchKernelGGL(rows2048_kernel, dim3((unsigned)(8 + ceil_div(ra.ngroup - 1, 8) * 8)), dim3(512), 0, st, ra);
    } else if (B == 1024) {
        constexpr int TH = ROWS_THREADS_1024;
        ra.ngroup = (int)ceil_div(nrow, TH / 64);
        hipLaunchKernelGGL((rows_kernel<64, TH>), dim3((unsigned)(ceil_div(ra.ngroup, 8) * 8)), dim3(TH), 0, st, ra);
    } else {
        constexpr int TH = ROWS_THREADS_512;
        ra.ngroup = (int)ceil_div(nrow, TH / 32);
        hipLaunchKernelGGL((rows_kernel<32, TH>), dim3((unsigned)(ceil_div(ra.ngroup, 8) * 8)), dim3(TH), 0, st, ra);
    }
}

void power_spectrum_rows(const double *W, const double *WW, const double *S, const double *Wb, const double *WbWb,
                         const double *Bk, int64_t n, int64_t l, double *p0, double *p1, double *pb, double *amps,
                         void *workspace, hipStream_t st)
{
    const int B = (int)(l / ROWS);
    char *ws = (char *)workspace;
    const size_t sz1 = align256(sizeof(cplx) * (HALF + 1) * (size_t)B);
    const size_t sz2 = align256(sizeof(cplx) * (HALF + 1) * (size_t)(B / 2));
    cplx *V1a = (cplx *)ws, *V1b = (cplx *)(ws + sz1);
    cplx *V3 = (cplx *)(ws + 2 * sz1), *V4 = (cplx *)(ws + 2 * sz1 + sz2), *V5 = (cplx *)(ws + 2 * sz1 + 2 * sz2);
    double *dc = (double *)(ws + 2 * sz1 + 3 * sz2);
    double *xb = (double *)(ws + 2 * sz1 + 3 * sz2 + 256);
    cplx *const V[5] = {V3, V4, V5, V1a, V1b};
    columns_part(W, WW, S, Wb, WbWb, Bk, n, l, 0, 1, V, xb, st);
    hipLaunchKernelGGL(rows_dc_kernel, dim3(1), dim3(256), 0, st, (const cplx *)V3, (const cplx *)V4,
                       (const cplx *)V5, B / 2, dc);
    const cplx *const Vc[5] = {V3, V4, V5, V1a, V1b};
    rows_part(Vc, dc, l, 0, HALF + 1, p0, p1, pb, amps, st);
}

// ---- the same in parts, for a transform dealt over several GPUs ---------------------------
// A part of the columns is nparts-th of every series' columns; it must hold whole patches of the
// regrouping (32 columns) and pairs of pass A's tiles (16 columns).
bool parts_supported(int64_t n, int64_t l, int nparts)
{
    if (!rows_path_applies(n, l) || nparts < 1) return false;
    const int64_t B2 = l / ROWS / 2;
    return B2 % nparts == 0 && (B2 / nparts) % GC == 0;
}

int64_t part_rows() { return HALF + 1; }

// elements (complex numbers) of the five arrays of one part of the columns: 3 second-moment
// series of B/2 columns and 2 first-moment series of B
size_t part_elements(int64_t l, int nparts)
{
    const size_t B = (size_t)(l / ROWS);
    return (size_t)(HALF + 1) * (3 * (B / 2) + 2 * B) / (size_t)nparts;
}

size_t part_workspace_bytes(int64_t l, int nparts)
{
    return sizeof(double) * (size_t)(2 * l + 3 * (l / 2)) / (size_t)nparts + 256;
}

void power_spectrum_columns_part(const double *W, const double *WW, const double *S, const double *Wb,
                                 const double *WbWb, const double *Bk, int64_t n, int64_t l, int part, int nparts,
                                 void *v_part, void *workspace, hipStream_t st)
{
    const size_t B = (size_t)(l / ROWS);
    const size_t e2 = (size_t)(HALF + 1) * (B / 2) / (size_t)nparts, e1 = 2 * e2;
    cplx *base = (cplx *)v_part;
    cplx *const V[5] = {base, base + e2, base + 2 * e2, base + 3 * e2, base + 3 * e2 + e1};
    columns_part(W, WW, S, Wb, WbWb, Bk, n, l, part, nparts, V, (double *)workspace, st);
}

void power_spectrum_rows_part(const void *v_rows, const double *dc, int64_t l, int64_t m_first, int64_t m_end,
                              double *p0, double *p1, double *pb, double *amps, hipStream_t st)
{
    const size_t B = (size_t)(l / ROWS);
    const size_t e2 = (size_t)(m_end - m_first) * (B / 2), e1 = 2 * e2;
    const cplx *base = (const cplx *)v_rows;
    const cplx *const V[5] = {base, base + e2, base + 2 * e2, base + 3 * e2, base + 3 * e2 + e1};
    rows_part(V, dc, l, (int)m_first, (int)m_end, p0, p1, pb, amps, st);
}

}   // namespace godot_spectrum

#ifdef GODOT_FFT_STAMPS
extern "C" int godot_debug_fft_stamps(unsigned long long *out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(g_stamps));
}
#endif
