// Direct trig-sum periodograms (gfx950): the weighted Z^2_m / H-test search
// over a uniform frequency grid, and the per-cell trig sums of the
// reference's power_spectrum_dft / fit_harmonics.
//
// Z^2_m is FP64-ALU bound.  Each thread owns J consecutive trial frequencies
// and keeps their 2m running sums in registers; the photon table is read with
// wave-uniform (scalar) loads, so there is no cross-lane reduction and no LDS
// traffic in the inner loop.  Per photon the thread forms the exact phase of
// its first frequency once (double-double product, polynomial sin/cos) and
// then steps through its J frequencies with the three-term recurrence
// z[k+1] = 2 cos(d) z[k] - z[k-1], d = 2 pi t df: one fused multiply-add per
// component and step.  A rounding error made at step j reaches step k scaled
// by sin((k-j) d) / sin(d), at most k - j, so the error after J steps is below
// J^2 / 2 ulp whatever d is.  The harmonics of one frequency come from the
// same recurrence in the harmonic index.
#include "common.hpp"

namespace {

constexpr double TWO_PI_HI = 6.283185307179586232e+00;
constexpr double TWO_PI_LO = 2.449293598294706414e-16;

// sin(2 pi x), cos(2 pi x) for x in cycles, |x| up to a few turns (the quarter
// turns are folded exactly).  Quadrant folding to |r| <= 1/8, then the classic
// minimax kernels on |theta| <= pi/4.
__device__ __forceinline__ void sincos_cycles(double x, double &s, double &c)
{
    double q = rint(4.0 * x);            // a small integer
    double r = fma(q, -0.25, x);         // exact
    double th = fma(r, TWO_PI_HI, r * TWO_PI_LO);
    double z = th * th;
    double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = fma(z, ps, 2.75573137070700676789e-06);
    ps = fma(z, ps, -1.98412698298579493134e-04);
    ps = fma(z, ps, 8.33333333332248946124e-03);
    ps = fma(z, ps, -1.66666666666666324348e-01);
    double sn = fma(th * z, ps, th);
    double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = fma(z, pc, -2.75573143513906633035e-07);
    pc = fma(z, pc, 2.48015872894767294178e-05);
    pc = fma(z, pc, -1.38888888888741095749e-03);
    pc = fma(z, pc, 4.16666666666666019037e-02);
    double cs = fma(z * z, pc, fma(z, -0.5, 1.0));
    int qi = (int)q & 3;                 // two's complement: -1 -> 3, -2 -> 2
    double s1 = (qi & 1) ? cs : sn;
    double c1 = (qi & 1) ? sn : cs;
    s = (qi & 2) ? -s1 : s1;
    c = ((qi + 1) & 2) ? -c1 : c1;
}

// One row per photon, 64 bytes so that a wave-uniform load fetches it whole.
struct PhotonRow {
    double a;    // frac(t f0), cycles in [-0.5, 0.5]
    double uh;   // frac(t df): high part, cycles in [-0.5, 0.5]
    double ul;   //             low part
    double w;    // weight
    double tcd;  // 2 cos(2 pi t df)
    double cd;   // cos(2 pi t df)
    double sd;   // sin(2 pi t df)
    double sqw;  // sqrt(w), NaN for a negative weight
};

__global__ void z2m_prepare_kernel(const double *t, const double *w, int64_t nph, double f0, double df,
                                   PhotonRow *rows)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nph) return;
    double ti = t[i];
    double p = ti * f0;
    double e = fma(ti, f0, -p);
    double a = (p - rint(p)) + e;
    a -= rint(a);
    double u = ti * df;
    double ue = fma(ti, df, -u);
    double uh = u - rint(u);
    // renormalise (uh, ue) so that uh carries the leading bits
    double sum = uh + ue;
    double ul = ue - (sum - uh);
    uh = sum;
    double sd, cd;
    double xr = uh - rint(uh);
    sincos_cycles(xr, sd, cd);
    // first-order correction for the low part of the phase step
    double dth = ul * TWO_PI_HI;
    double cd2 = fma(-sd, dth, cd), sd2 = fma(cd, dth, sd);
    PhotonRow r;
    r.a = a;
    r.uh = uh;
    r.ul = ul;
    r.w = w[i];
    r.tcd = cd2 + cd2;
    r.cd = cd2;
    r.sd = sd2;
    r.sqw = sqrt(r.w);
    rows[i] = r;
}

// deterministic sum of w^2 by one workgroup; out[1] = 1 when some weight is
// negative or NaN (the sqrt(w) phasor of the M = 2 kernel needs w >= 0)
__global__ __launch_bounds__(1024) void sumsq_kernel(const double *w, int64_t n, double *out)
{
    __shared__ double part[1024 / WAVE];
    double s = 0.0;
    int bad = 0;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
        s = fma(w[i], w[i], s);
        bad |= !(w[i] >= 0.0);
    }
    s = team_sum<WAVE>(s);
    bad = __syncthreads_or(bad);
    if ((threadIdx.x % WAVE) == 0) part[threadIdx.x / WAVE] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0.0;
        for (int k = 0; k < 1024 / WAVE; ++k) tot += part[k];
        out[0] = tot;
        out[1] = bad ? 1.0 : 0.0;
    }
}

constexpr int Z_THREADS = 256;
#ifndef Z_AHEAD
#define Z_AHEAD 4
#endif
constexpr int Z_FOLD = 4096;      // photons between two corrections of the sum of c^2 (a multiple of Z_AHEAD)

// Frequencies per thread.  The sin/cos of a thread's first frequency costs
// about 50 instructions per photon, a further frequency 4 (m = 1), 6 (m = 2)
// or 4m + 1, so J is as large as the registers allow: about 96 running sums
// (192 VGPRs) at two waves per SIMD; m = 1 keeps four waves (measured best).
#ifndef Z_JCAP
#define Z_JCAP 48
#endif
constexpr int z2m_J(int m) { return m == 1 ? 24 : m == 2 ? 20 : Z_JCAP / m; }
// Waves per SIMD the kernels are compiled for (their launch bounds); with 256
// CUs of 4 SIMDs this fixes how many workgroups are resident at a time.
constexpr int z2m_waves(int m) { return m == 1 ? 4 : 2; }

// Accumulate the 2M trig sums of J frequencies over photons [i0, i1).
// acc layout: [k][h][0 = cos, 1 = sin].
template <int M, int J, bool SQ>
__device__ __forceinline__ void z2m_accumulate(const PhotonRow *__restrict__ rows, int64_t i0, int64_t i1,
                                               double jb, double (&acc)[J][M][2])
{
    // one photon: all J frequencies of this thread
    auto one = [&](const PhotonRow &r) {
        // phase of this thread's first frequency: a + frac(jb * u), u = uh + ul
        double p = jb * r.uh;
        double e = fma(jb, r.uh, -p);
        // |x| <= 1.5: sincos_cycles folds whole quarter turns itself
        const double x = (p - rint(p)) + fma(jb, r.ul, e) + r.a;
        double s, c;
        sincos_cycles(x, s, c);
        if constexpr (SQ) {
            // Non-negative weights, M == 2.  The phasor carries the amplitude sqrt(w), so that its square is
            // w exp(2 i phi): harmonic 2 is two multiply-adds on (c, s).  The sine sum is left at half
            // its value until the loop is over; the cosine sum, w cos 2 phi = 2 c^2 - w, is gathered as
            // sum c^2, and the caller takes half the weights off it block by block (see below).
            c *= r.sqw;
            s *= r.sqw;
            double cp = fma(c, r.cd, s * r.sd), sp = fma(s, r.cd, -(c * r.sd));
#pragma unroll
            for (int k = 0; k < J; ++k) {
                acc[k][0][0] = fma(r.sqw, c, acc[k][0][0]);
                acc[k][0][1] = fma(r.sqw, s, acc[k][0][1]);
                acc[k][1][0] = fma(c, c, acc[k][1][0]);
                acc[k][1][1] = fma(c, s, acc[k][1][1]);
                if (k + 1 < J) {
                    double cn = fma(r.tcd, c, -cp), sn = fma(r.tcd, s, -sp);
                    cp = c;
                    sp = s;
                    c = cn;
                    s = sn;
                }
            }
            return;
        }
        double cp = fma(c, r.cd, s * r.sd), sp = fma(s, r.cd, -(c * r.sd));
#pragma unroll
        for (int k = 0; k < J; ++k) {
            const double t2 = c + c;
            double chm = 1.0, shm = 0.0, ch = c, sh = s;
#pragma unroll
            for (int h = 0; h < M; ++h) {
                acc[k][h][0] = fma(r.w, ch, acc[k][h][0]);
                acc[k][h][1] = fma(r.w, sh, acc[k][h][1]);
                if (h + 1 < M) {
                    double cn = fma(t2, ch, -chm), sn = fma(t2, sh, -shm);
                    chm = ch;
                    shm = sh;
                    ch = cn;
                    sh = sn;
                }
            }
            if (k + 1 < J) {
                double cn = fma(r.tcd, c, -cp), sn = fma(r.tcd, s, -sp);
                cp = c;
                sp = s;
                c = cn;
                s = sn;
            }
        }
    };
    // The rows come by wave-uniform (scalar) loads.  Z_AHEAD of them are requested
    // together, so the wavefront waits out the load latency once per group and not
    // once per photon (at two wavefronts per SIMD nobody else hides it).
    if constexpr (!SQ) {
        int64_t i = i0;
        for (; i + Z_AHEAD <= i1; i += Z_AHEAD) {
            PhotonRow r[Z_AHEAD];
#pragma unroll
            for (int u = 0; u < Z_AHEAD; ++u) r[u] = rows[i + u];
#pragma unroll
            for (int u = 0; u < Z_AHEAD; ++u) one(r[u]);
        }
        for (; i < i1; ++i) one(rows[i]);
    } else {
        // sum c^2 runs ahead of the sum it stands for by half the weights gone by.  They are
        // taken off after every block of Z_FOLD photons, so that the running sums stay as small
        // as the trig sums themselves plus one block's weights and keep their low bits (left to
        // the end, a sum of 10^6 photons would carry 10^5 and lose four digits of the 10^2 that
        // matter).
        for (int64_t b0 = i0; b0 < i1; b0 += Z_FOLD) {
            const int64_t b1 = b0 + Z_FOLD < i1 ? b0 + Z_FOLD : i1;
            double wblk = 0.0;
            int64_t i = b0;
            for (; i + Z_AHEAD <= b1; i += Z_AHEAD) {
                PhotonRow r[Z_AHEAD];
#pragma unroll
                for (int u = 0; u < Z_AHEAD; ++u) r[u] = rows[i + u];
#pragma unroll
                for (int u = 0; u < Z_AHEAD; ++u) {
                    wblk += r[u].w;
                    one(r[u]);
                }
            }
            for (; i < b1; ++i) {
                wblk += rows[i].w;
                one(rows[i]);
            }
#pragma unroll
            for (int k = 0; k < J; ++k) acc[k][1][0] = fma(-0.5, wblk, acc[k][1][0]);
        }
#pragma unroll
        for (int k = 0; k < J; ++k) {
            acc[k][1][0] += acc[k][1][0];
            acc[k][1][1] += acc[k][1][1];
        }
    }
}

template <int M, int J>
__device__ __forceinline__ void z2m_finish(const double (&acc)[J][M][2], double norm, int64_t jrel, int64_t nf,
                                           double *z2, double *h)
{
#pragma unroll
    for (int k = 0; k < J; ++k) {
        double z = 0.0, hbest = -INFINITY;
#pragma unroll
        for (int hh = 0; hh < M; ++hh) {
            z += norm * (acc[k][hh][0] * acc[k][hh][0] + acc[k][hh][1] * acc[k][hh][1]);
            double cand = z - 4.0 * (hh + 1) + 4.0;
            hbest = cand > hbest ? cand : hbest;
        }
        if (jrel + k < nf) {
            if (z2) z2[jrel + k] = z;
            if (h) h[jrel + k] = hbest;
        }
    }
}

// One launch over all photons: thread -> J frequencies, finish in-kernel.
// For M = 2 both variants are launched and the weight-sign flag sumw2[1]
// decides which of them runs.
template <int M, int J, bool SQ>
__global__ __launch_bounds__(Z_THREADS, z2m_waves(M)) void z2m_kernel(const PhotonRow *__restrict__ rows,
                                                                      int64_t nph, const double *sumw2,
                                                                      int64_t j0, int64_t nf, double *z2,
                                                                      double *h)
{
    const int64_t jrel = ((int64_t)blockIdx.x * Z_THREADS + threadIdx.x) * J;
    if (jrel >= nf) return;
    if (M == 2 && (sumw2[1] == 0.0) != SQ) return;
    double acc[J][M][2];
#pragma unroll
    for (int k = 0; k < J; ++k)
#pragma unroll
        for (int hh = 0; hh < M; ++hh) acc[k][hh][0] = acc[k][hh][1] = 0.0;
    z2m_accumulate<M, J, SQ>(rows, 0, nph, (double)(j0 + jrel), acc);
    z2m_finish<M, J>(acc, 2.0 / *sumw2, jrel, nf, z2, h);
}

// Few frequencies: photons are split over gridDim.y chunks, partial sums go to
// `part` [chunk][freq][M][2] and a second launch adds them in chunk order.
template <int M, int J, bool SQ>
__global__ __launch_bounds__(Z_THREADS, z2m_waves(M)) void z2m_partial_kernel(const PhotonRow *__restrict__ rows,
                                                                              int64_t nph, const double *sumw2,
                                                                              int64_t j0, int64_t nf,
                                                                              double *part)
{
    const int64_t jrel = ((int64_t)blockIdx.x * Z_THREADS + threadIdx.x) * J;
    if (jrel >= nf) return;
    if (M == 2 && (sumw2[1] == 0.0) != SQ) return;
    const int64_t per = (nph + gridDim.y - 1) / gridDim.y;
    const int64_t i0 = (int64_t)blockIdx.y * per;
    const int64_t i1 = i0 + per < nph ? i0 + per : nph;
    double acc[J][M][2];
#pragma unroll
    for (int k = 0; k < J; ++k)
#pragma unroll
        for (int hh = 0; hh < M; ++hh) acc[k][hh][0] = acc[k][hh][1] = 0.0;
    z2m_accumulate<M, J, SQ>(rows, i0, i1, (double)(j0 + jrel), acc);
    double *o = part + ((int64_t)blockIdx.y * nf + jrel) * (M * 2);
#pragma unroll
    for (int k = 0; k < J; ++k) {
        if (jrel + k < nf) {
#pragma unroll
            for (int hh = 0; hh < M; ++hh) {
                o[(k * M + hh) * 2 + 0] = acc[k][hh][0];
                o[(k * M + hh) * 2 + 1] = acc[k][hh][1];
            }
        }
    }
}

__global__ void z2m_combine_kernel(const double *part, int nchunk, int m, const double *sumw2, int64_t nf,
                                   double *z2, double *h)
{
    int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nf) return;
    const double norm = 2.0 / *sumw2;
    double z = 0.0, hbest = -INFINITY;
    for (int hh = 0; hh < m; ++hh) {
        double c = 0.0, s = 0.0;
        for (int q = 0; q < nchunk; ++q) {
            const double *o = part + (((int64_t)q * nf + j) * m + hh) * 2;
            c += o[0];
            s += o[1];
        }
        z += norm * (c * c + s * s);
        double cand = z - 4.0 * (hh + 1) + 4.0;
        hbest = cand > hbest ? cand : hbest;
    }
    if (z2) z2[j] = z;
    if (h) h[j] = hbest;
}

// part [chunk][freq][m][2] -> sums [freq][m][2], chunks added in order
__global__ void z2m_add_chunks_kernel(const double *part, int nchunk, int64_t n, double *sums)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double t = 0.0;
    for (int q = 0; q < nchunk; ++q) t += part[(int64_t)q * n + i];
    sums[i] = t;
}

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }


constexpr int Z_SIMDS = 256 * 4;
constexpr int Z_MAX_CHUNKS = 64;

// Number of photon chunks (gridDim.y).  A launch runs in rounds of `resident`
// workgroups; a last round that is only partly full runs as long as a full one,
// so the chunk count is chosen to make blocks * chunks fill whole rounds.
inline int z2m_chunks(int64_t nph, int64_t nf, int m)
{
    const int64_t blocks = ceil_div(nf, (int64_t)Z_THREADS * z2m_J(m));
    const int64_t resident = (int64_t)Z_SIMDS * z2m_waves(m) / (Z_THREADS / WAVE);
    int64_t cmax = nph / 2048 > 1 ? nph / 2048 : 1;
    if (cmax > Z_MAX_CHUNKS) cmax = Z_MAX_CHUNKS;
    int best = 1;
    double best_cost = (double)ceil_div(blocks, resident);
    for (int64_t c = 2; c <= cmax; ++c) {
        double cost = (double)ceil_div(blocks * c, resident) / (double)c;
        if (cost < 0.97 * best_cost) {
            best = (int)c;
            best_cost = cost;
        }
        if (blocks * c >= 8 * resident) break;
    }
    return best;
}

template <int M, int J>
int z2m_launch(const PhotonRow *rows, int64_t nph, const double *sumw2, int64_t j0, int64_t nf, int chunks,
               double *part, double *z2, double *h, hipStream_t st)
{
    unsigned gx = (unsigned)ceil_div(nf, (int64_t)Z_THREADS * J);
    if (chunks <= 1) {
        hipLaunchKernelGGL((z2m_kernel<M, J, false>), dim3(gx), dim3(Z_THREADS), 0, st, rows, nph, sumw2, j0,
                           nf, z2, h);
        if constexpr (M == 2)
            hipLaunchKernelGGL((z2m_kernel<M, J, true>), dim3(gx), dim3(Z_THREADS), 0, st, rows, nph, sumw2,
                               j0, nf, z2, h);
    } else {
        hipLaunchKernelGGL((z2m_partial_kernel<M, J, false>), dim3(gx, (unsigned)chunks), dim3(Z_THREADS), 0,
                           st, rows, nph, sumw2, j0, nf, part);
        if constexpr (M == 2)
            hipLaunchKernelGGL((z2m_partial_kernel<M, J, true>), dim3(gx, (unsigned)chunks), dim3(Z_THREADS),
                               0, st, rows, nph, sumw2, j0, nf, part);
        hipLaunchKernelGGL(z2m_combine_kernel, dim3((unsigned)ceil_div(nf, 256)), dim3(256), 0, st,
                           (const double *)part, chunks, M, sumw2, nf, z2, h);
    }
    return 0;
}

// Trig sums only: partial kernel for every chunk count; one chunk writes `sums` directly.
template <int M, int J>
int z2m_sums_launch(const PhotonRow *rows, int64_t nph, const double *sumw2, int64_t j0, int64_t nf, int chunks,
                    double *part, double *sums, hipStream_t st)
{
    unsigned gx = (unsigned)ceil_div(nf, (int64_t)Z_THREADS * J);
    double *dst = chunks <= 1 ? sums : part;
    hipLaunchKernelGGL((z2m_partial_kernel<M, J, false>), dim3(gx, (unsigned)(chunks < 1 ? 1 : chunks)),
                       dim3(Z_THREADS), 0, st, rows, nph, sumw2, j0, nf, dst);
    if constexpr (M == 2)
        hipLaunchKernelGGL((z2m_partial_kernel<M, J, true>), dim3(gx, (unsigned)(chunks < 1 ? 1 : chunks)),
                           dim3(Z_THREADS), 0, st, rows, nph, sumw2, j0, nf, dst);
    if (chunks > 1) {
        int64_t n = nf * M * 2;
        hipLaunchKernelGGL(z2m_add_chunks_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st,
                           (const double *)part, chunks, n, sums);
    }
    return 0;
}

// ---------------------------------------------------------------------------
// Per-cell trig sums of the reference's power_spectrum_dft (core.py:2540-2596)
// and fit_harmonics (core.py:2426-2488).  Every photon of a cell shares the
// cell's phase, so sum_i w_i c_cell(i) = sum_cells c W1 and sum_i (w_i c)^2 =
// sum_cells c^2 W2 with the per-cell moments W1 = sum w, W2 = sum w^2.
// One workgroup per frequency, threads over cells.
// ---------------------------------------------------------------------------
template <int NS>
__device__ void block_sums(double (&v)[NS], double *smem)
{
#pragma unroll
    for (int k = 0; k < NS; ++k) v[k] = team_sum<WAVE>(v[k]);
    const int lane = threadIdx.x % WAVE, wave = threadIdx.x / WAVE, nw = blockDim.x / WAVE;
    if (lane == 0)
        for (int k = 0; k < NS; ++k) smem[wave * NS + k] = v[k];
    __syncthreads();
    for (int k = 0; k < NS; ++k) {
        double t = 0.0;
        for (int q = 0; q < nw; ++q) t += smem[q * NS + k];
        v[k] = t;
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void trigsum_cells_kernel(const double *w1, const double *w2,
                                                            const double *exp, const double *tmid,
                                                            int64_t ncell, const double *freqs,
                                                            double exp_scale, double *power)
{
#pragma clang fp contract(off)   // the table index must round as numpy's does
    __shared__ double sm[4 * 6];
    const double f = freqs[blockIdx.x];
    const double t0 = tmid[0];
    const double dt = tmid[1] - tmid[0];
    const double two_pi = 6.283185307179586;
    double q = dt * 2 * 3.141592653589793 * f;
    const double corr = 1 - (q * q) / 24;
    const double step = two_pi / 1000;   // np.linspace(0, 2 pi, 1001) spacing
    double v[6] = {0, 0, 0, 0, 0, 0};
    for (int64_t c = threadIdx.x; c < ncell; c += blockDim.x) {
        double ph = (tmid[c] - t0) * f;
        ph = ph - trunc(ph);
        int k = (int)(ph * 1000);
        // a negative phase (f < 0) indexes the table from its end, as numpy does
        if (k < 0) k += 1000;
        k = k < 0 ? 0 : (k > 999 ? 999 : k);
        double ang = k * step;
        double cph = cos(ang), sph = sin(ang);
        double a = w1[c], b = w2[c], ex = exp[c] * exp_scale;
        v[0] += cph * a;
        v[1] += (cph * cph) * b;
        v[2] += (cph * ex) * corr;
        v[3] += sph * a;
        v[4] += (sph * sph) * b;
        v[5] += (sph * ex) * corr;
    }
    block_sums<6>(v, sm);
    if (threadIdx.x == 0) {
        double dc = v[0] - v[2], ds = v[3] - v[5];
        power[blockIdx.x] = dc * dc / v[1] + ds * ds / v[4];
    }
}

__global__ __launch_bounds__(256) void fit_harmonics_kernel(const double *w1, const double *w2,
                                                            const double *exp, const double *tmid,
                                                            int64_t ncell, const double *freqs, double tmean,
                                                            double *out)
{
    __shared__ double sm[4 * 6];
    const double f = freqs[blockIdx.x];
    const double om = (2 * 3.141592653589793) * f;
    double v[6] = {0, 0, 0, 0, 0, 0};
    for (int64_t c = threadIdx.x; c < ncell; c += blockDim.x) {
        double ph = om * (tmid[c] - tmean);
        double sph, cph;
        sincos(ph, &sph, &cph);
        double a = w1[c], b = w2[c], ex = exp[c];
        v[0] += cph * a;
        v[1] += (cph * cph) * b;
        v[2] += ex * cph;
        v[3] += sph * a;
        v[4] += (sph * sph) * b;
        v[5] += ex * sph;
    }
    block_sums<6>(v, sm);
    if (threadIdx.x == 0) {
        double dc = v[0] - v[2], ds = v[3] - v[5];
        double *o = out + (int64_t)blockIdx.x * 3;
        o[0] = dc / v[1];
        o[1] = ds / v[4];
        o[2] = dc * dc / v[1] + ds * ds / v[4];
    }
}

// compute_ls (core.py:2490-2531): Lomb-Scargle-like sums of a normalised
// per-cell flux, phases from the 1000-entry table as in power_spectrum_dft.
__global__ __launch_bounds__(256) void ls_cells_kernel(const double *flux, const double *tmid, int64_t ncell,
                                                       const double *freqs, double *power)
{
#pragma clang fp contract(off)
    __shared__ double sm[4 * 2];
    const double f = freqs[blockIdx.x];
    const double t0 = tmid[0];
    const double step = 6.283185307179586 / 1000;
    double v[2] = {0, 0};
    for (int64_t c = threadIdx.x; c < ncell; c += blockDim.x) {
        double ph = (tmid[c] - t0) * f;
        ph = ph - trunc(ph);
        int k = (int)(ph * 1000);
        if (k < 0) k += 1000;
        k = k < 0 ? 0 : (k > 999 ? 999 : k);
        double ang = k * step;
        v[0] += cos(ang) * flux[c];
        v[1] += sin(ang) * flux[c];
    }
    block_sums<2>(v, sm);
    if (threadIdx.x == 0) power[blockIdx.x] = v[0] * v[0] + v[1] * v[1];
}

// get_orbital_modulation (core.py:2845-2859), one frequency: partial sums of
// cos*y, sin*y, cos^2*z, sin^2*z over the bins, combined in a fixed order.
constexpr int OM_BLOCKS = 256;

__global__ __launch_bounds__(256) void orbital_moments_kernel(const double *starts, const double *stops,
                                                              const double *y, const double *z, int64_t n,
                                                              double freq, double *partials)
{
    __shared__ double sm[4 * 4];
    const double t0 = (starts[0] + stops[0]) * 0.5;
    double v[4] = {0, 0, 0, 0};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        double x = ((starts[i] + stops[i]) * 0.5 - t0) * freq;
        double ph = (x - floor(x)) * 6.283185307179586;
        double sp, cp;
        sincos(ph, &sp, &cp);
        v[0] += cp * y[i];
        v[1] += sp * y[i];
        v[2] += cp * cp * z[i];
        v[3] += sp * sp * z[i];
    }
    block_sums<4>(v, sm);
    if (threadIdx.x == 0)
        for (int k = 0; k < 4; ++k) partials[blockIdx.x * 4 + k] = v[k];
}

__global__ void orbital_finish_kernel(const double *partials, int nblocks, int64_t n, double *out)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    for (int k = 0; k < 4; ++k) {
        double s = 0.0;
        for (int b = 0; b < nblocks; ++b) s += partials[b * 4 + k];
        out[k] = s / (double)n;   // np.mean
    }
}

__global__ void orbital_update_kernel(const double *starts, const double *stops, int64_t n, double freq,
                                      const double *means, double *rvals)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double t0 = (starts[0] + stops[0]) * 0.5;
    double x = ((starts[i] + stops[i]) * 0.5 - t0) * freq;
    double ph = (x - floor(x)) * 6.283185307179586;
    double sp, cp;
    sincos(ph, &sp, &cp);
    rvals[i] += 2 * (means[0] * cp + means[1] * sp);
}

}   // namespace

extern "C" int godot_ls_cells(const double *flux, const double *tmid, int64_t ncell, const double *freqs,
                              int64_t nfreq, double *power, godot_stream_t stream)
{
    GODOT_ENTRY();
    GODOT_REQUIRE(ncell >= 1 && nfreq >= 0, GODOT_E_INVALID, "godot_ls_cells: bad sizes");
    GODOT_REQUIRE(nfreq < ((int64_t)1 << 31), GODOT_E_UNSUPPORTED, "godot_ls_cells: too many frequencies");
    if (nfreq == 0) return GODOT_OK;
    GODOT_REQUIRE(flux && tmid && freqs && power, GODOT_E_INVALID, "godot_ls_cells: null argument");
    hipLaunchKernelGGL(ls_cells_kernel, dim3((unsigned)nfreq), dim3(256), 0, (hipStream_t)stream, flux, tmid,
                       ncell, freqs, power);
    GODOT_LAUNCH_CHECK("godot_ls_cells");
    return GODOT_OK;
}

extern "C" size_t godot_orbital_workspace_bytes(void) { return sizeof(double) * 4 * OM_BLOCKS; }

extern "C" int godot_orbital_modulation_step(const double *starts, const double *stops, const double *y,
                                             const double *z, int64_t n, double freq, double *means,
                                             double *rvals, void *workspace, size_t workspace_bytes,
                                             godot_stream_t stream)
{
    GODOT_ENTRY();
    GODOT_REQUIRE(n >= 1, GODOT_E_INVALID, "godot_orbital_modulation_step: n must be positive");
    GODOT_REQUIRE(starts && stops && y && z && means && rvals, GODOT_E_INVALID,
                  "godot_orbital_modulation_step: null argument");
    GODOT_REQUIRE(workspace && workspace_bytes >= godot_orbital_workspace_bytes(), GODOT_E_WORKSPACE,
                  "godot_orbital_modulation_step: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    int nblocks = (int)(ceil_div(n, 256) < OM_BLOCKS ? ceil_div(n, 256) : OM_BLOCKS);
    double *partials = (double *)workspace;
    hipLaunchKernelGGL(orbital_moments_kernel, dim3(nblocks), dim3(256), 0, st, starts, stops, y, z, n, freq,
                       partials);
    hipLaunchKernelGGL(orbital_finish_kernel, dim3(1), dim3(64), 0, st, (const double *)partials, nblocks, n,
                       means);
    hipLaunchKernelGGL(orbital_update_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st, starts, stops,
                       n, freq, (const double *)means, rvals);
    GODOT_LAUNCH_CHECK("godot_orbital_modulation_step");
    return GODOT_OK;
}

extern "C" size_t godot_z2m_workspace_bytes(int64_t nph, int64_t nf, int m)
{
    if (nph < 0 || nf < 0 || m < 1 || m > 20) return 0;
    int chunks = z2m_chunks(nph, nf, m);
    size_t part = chunks > 1 ? sizeof(double) * (size_t)chunks * (size_t)nf * (size_t)m * 2 : 0;
    return align256(sizeof(PhotonRow) * (size_t)nph) + 256 + align256(part);
}

extern "C" int godot_z2m(const double *t, const double *w, int64_t nph, double f0, double df, int64_t j0,
                         int64_t nf, int m, double *z2, double *h, void *workspace, size_t workspace_bytes,
                         godot_stream_t stream)
{
    GODOT_ENTRY();
    GODOT_REQUIRE(nph >= 0 && nf >= 0 && j0 >= 0, GODOT_E_INVALID, "godot_z2m: negative size");
    GODOT_REQUIRE(m >= 1 && m <= 20, GODOT_E_UNSUPPORTED, "godot_z2m: m must be in [1, 20]");
    GODOT_REQUIRE(j0 + nf < ((int64_t)1 << 52), GODOT_E_UNSUPPORTED, "godot_z2m: frequency index too large");
    if (nf == 0) return GODOT_OK;
    GODOT_REQUIRE(z2 || h, GODOT_E_INVALID, "godot_z2m: both outputs are null");
    GODOT_REQUIRE((t && w) || nph == 0, GODOT_E_INVALID, "godot_z2m: null photon pointer");
    GODOT_REQUIRE(workspace && workspace_bytes >= godot_z2m_workspace_bytes(nph, nf, m), GODOT_E_WORKSPACE,
                  "godot_z2m: workspace smaller than godot_z2m_workspace_bytes(nph, nf, m)");
    hipStream_t st = (hipStream_t)stream;
    PhotonRow *rows = (PhotonRow *)workspace;
    size_t off = align256(sizeof(PhotonRow) * (size_t)nph);
    double *sumw2 = (double *)((char *)workspace + off);
    double *part = (double *)((char *)workspace + off + 256);
    if (nph > 0)
        hipLaunchKernelGGL(z2m_prepare_kernel, dim3((unsigned)ceil_div(nph, 256)), dim3(256), 0, st, t, w, nph,
                           f0, df, rows);
    hipLaunchKernelGGL(sumsq_kernel, dim3(1), dim3(1024), 0, st, w, nph, sumw2);
    int chunks = z2m_chunks(nph, nf, m);
#define Z2M_CASE(M_) \
    case M_: z2m_launch<M_, z2m_J(M_)>(rows, nph, sumw2, j0, nf, chunks, part, z2, h, st); break;
    switch (m) {
        Z2M_CASE(1) Z2M_CASE(2) Z2M_CASE(3) Z2M_CASE(4) Z2M_CASE(5) Z2M_CASE(6) Z2M_CASE(7)
        Z2M_CASE(8) Z2M_CASE(9) Z2M_CASE(10) Z2M_CASE(11) Z2M_CASE(12) Z2M_CASE(13) Z2M_CASE(14)
        Z2M_CASE(15) Z2M_CASE(16) Z2M_CASE(17) Z2M_CASE(18) Z2M_CASE(19) Z2M_CASE(20)
    }
#undef Z2M_CASE
    GODOT_LAUNCH_CHECK("godot_z2m");
    return GODOT_OK;
}

extern "C" int godot_z2m_sums(const double *t, const double *w, int64_t nph, double f0, double df, int64_t j0,
                              int64_t nf, int m, double *sums, double *sumw2_out, void *workspace,
                              size_t workspace_bytes, godot_stream_t stream)
{
    GODOT_ENTRY();
    GODOT_REQUIRE(nph >= 0 && nf >= 0 && j0 >= 0, GODOT_E_INVALID, "godot_z2m_sums: negative size");
    GODOT_REQUIRE(m >= 1 && m <= 20, GODOT_E_UNSUPPORTED, "godot_z2m_sums: m must be in [1, 20]");
    GODOT_REQUIRE(j0 + nf < ((int64_t)1 << 52), GODOT_E_UNSUPPORTED,
                  "godot_z2m_sums: frequency index too large");
    GODOT_REQUIRE(sumw2_out, GODOT_E_INVALID, "godot_z2m_sums: null sumw2 output");
    GODOT_REQUIRE(sums || nf == 0, GODOT_E_INVALID, "godot_z2m_sums: null sums output");
    GODOT_REQUIRE((t && w) || nph == 0, GODOT_E_INVALID, "godot_z2m_sums: null photon pointer");
    GODOT_REQUIRE(workspace && workspace_bytes >= godot_z2m_workspace_bytes(nph, nf, m), GODOT_E_WORKSPACE,
                  "godot_z2m_sums: workspace smaller than godot_z2m_workspace_bytes(nph, nf, m)");
    hipStream_t st = (hipStream_t)stream;
    PhotonRow *rows = (PhotonRow *)workspace;
    size_t off = align256(sizeof(PhotonRow) * (size_t)nph);
    double *sumw2 = (double *)((char *)workspace + off);
    double *part = (double *)((char *)workspace + off + 256);
    hipLaunchKernelGGL(sumsq_kernel, dim3(1), dim3(1024), 0, st, w, nph, sumw2);
    GODOT_REQUIRE(hipMemcpyAsync(sumw2_out, sumw2, sizeof(double), hipMemcpyDeviceToDevice, st) == hipSuccess,
                  GODOT_E_LAUNCH, "godot_z2m_sums: copy of sum(w^2) failed");
    if (nf == 0) {
        GODOT_LAUNCH_CHECK("godot_z2m_sums");
        return GODOT_OK;
    }
    if (nph > 0)
        hipLaunchKernelGGL(z2m_prepare_kernel, dim3((unsigned)ceil_div(nph, 256)), dim3(256), 0, st, t, w, nph,
                           f0, df, rows);
    int chunks = z2m_chunks(nph, nf, m);
#define Z2M_CASE(M_) \
    case M_: z2m_sums_launch<M_, z2m_J(M_)>(rows, nph, sumw2, j0, nf, chunks, part, sums, st); break;
    switch (m) {
        Z2M_CASE(1) Z2M_CASE(2) Z2M_CASE(3) Z2M_CASE(4) Z2M_CASE(5) Z2M_CASE(6) Z2M_CASE(7)
        Z2M_CASE(8) Z2M_CASE(9) Z2M_CASE(10) Z2M_CASE(11) Z2M_CASE(12) Z2M_CASE(13) Z2M_CASE(14)
        Z2M_CASE(15) Z2M_CASE(16) Z2M_CASE(17) Z2M_CASE(18) Z2M_CASE(19) Z2M_CASE(20)
    }
#undef Z2M_CASE
    GODOT_LAUNCH_CHECK("godot_z2m_sums");
    return GODOT_OK;
}

extern "C" int godot_z2m_finish(const double *sums, const double *sumw2, int64_t nf, int m, double *z2,
                                double *h, godot_stream_t stream)
{
    GODOT_ENTRY();
    GODOT_REQUIRE(nf >= 0, GODOT_E_INVALID, "godot_z2m_finish: negative size");
    GODOT_REQUIRE(m >= 1 && m <= 20, GODOT_E_UNSUPPORTED, "godot_z2m_finish: m must be in [1, 20]");
    if (nf == 0) return GODOT_OK;
    GODOT_REQUIRE(sums && sumw2, GODOT_E_INVALID, "godot_z2m_finish: null input");
    GODOT_REQUIRE(z2 || h, GODOT_E_INVALID, "godot_z2m_finish: both outputs are null");
    hipLaunchKernelGGL(z2m_combine_kernel, dim3((unsigned)ceil_div(nf, 256)), dim3(256), 0, (hipStream_t)stream,
                       sums, 1, m, sumw2, nf, z2, h);
    GODOT_LAUNCH_CHECK("godot_z2m_finish");
    return GODOT_OK;
}

extern "C" int godot_trigsum_cells(const double *w1, const double *w2, const double *exp, const double *tmid,
                                   int64_t ncell, const double *freqs, int64_t nfreq, double exp_scale,
                                   double *power, godot_stream_t stream)
{
    GODOT_ENTRY();
    GODOT_REQUIRE(ncell >= 2 && nfreq >= 0, GODOT_E_INVALID,
                  "godot_trigsum_cells: needs at least two cells (the cell spacing enters the statistic)");
    GODOT_REQUIRE(nfreq < ((int64_t)1 << 31), GODOT_E_UNSUPPORTED, "godot_trigsum_cells: too many frequencies");
    if (nfreq == 0) return GODOT_OK;
    GODOT_REQUIRE(w1 && w2 && exp && tmid && freqs && power, GODOT_E_INVALID,
                  "godot_trigsum_cells: null argument");
    hipLaunchKernelGGL(trigsum_cells_kernel, dim3((unsigned)nfreq), dim3(256), 0, (hipStream_t)stream, w1, w2,
                       exp, tmid, ncell, freqs, exp_scale, power);
    GODOT_LAUNCH_CHECK("godot_trigsum_cells");
    return GODOT_OK;
}

extern "C" int godot_fit_harmonics(const double *w1, const double *w2, const double *exp, const double *tmid,
                                   int64_t ncell, double tmean, const double *freqs, int64_t nfreq,
                                   double *out, godot_stream_t stream)
{
    GODOT_ENTRY();
    GODOT_REQUIRE(ncell >= 1 && nfreq >= 0, GODOT_E_INVALID, "godot_fit_harmonics: bad sizes");
    GODOT_REQUIRE(nfreq < ((int64_t)1 << 31), GODOT_E_UNSUPPORTED, "godot_fit_harmonics: too many frequencies");
    if (nfreq == 0) return GODOT_OK;
    GODOT_REQUIRE(w1 && w2 && exp && tmid && freqs && out, GODOT_E_INVALID,
                  "godot_fit_harmonics: null argument");
    hipLaunchKernelGGL(fit_harmonics_kernel, dim3((unsigned)nfreq), dim3(256), 0, (hipStream_t)stream, w1, w2,
                       exp, tmid, ncell, freqs, tmean, out);
    GODOT_LAUNCH_CHECK("godot_fit_harmonics");
    return GODOT_OK;
}
