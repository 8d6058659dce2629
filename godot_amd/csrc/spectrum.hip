// Real FFTs of the binned light curve and the likelihood power-spectrum
// epilogue (gfx950).
//
// A real series of (zero-padded) length l is transformed as a complex series
// of length N = l/2 (even samples -> real parts, odd samples -> imaginary
// parts) followed by the split into the real transform's bins.  The complex
// transform is a Stockham autosort FFT ping-ponging between two HBM buffers.
// Long transforms take radix-256 passes: two radix-16 butterflies per kernel
// with a transpose through LDS in between, so a 2^24-point transform is three
// HBM round trips; what is left of log2 N goes in one more pass (radix 2 .. 128).
// Their first pass reads the real series themselves (difference of two moment
// arrays, even/odd packing, zero padding all formed while loading), so the
// padded complex input never exists in memory.
// Short transforms (N < 4096) take plain radix-16 passes (one thread = one
// 16-point butterfly in registers).  Reads are coalesced, writes come in runs
// of >= 256 bytes.  Twiddles are computed in registers from an exactly reduced
// angle, so no table is read.  HBM-bound: each pass moves 32 bytes per complex
// point.
#include "spectrum_common.hpp"
#include <stdlib.h>

using namespace godot_spectrum;

namespace {


// One Stockham pass of radix R over `batch` transforms of length N.
// Ns = product of the radices of the earlier passes.
template <int R>
__global__ __launch_bounds__(256) void stockham_pass_kernel(const cplx *__restrict__ in,
                                                            cplx *__restrict__ out, int64_t N, int64_t Ns)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = N / R;
    if (j >= stride) return;
    const cplx *src = in + (int64_t)blockIdx.y * N;
    cplx *dst = out + (int64_t)blockIdx.y * N;
    const int64_t k = j & (Ns - 1);   // Ns is a power of two
    cplx v[R];
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] = src[j + r * stride];
    if (Ns > 1) {
        // w = exp(-2 pi i k / (Ns R)); k / (Ns R) is exact (power-of-two denominator)
        double x = -(double)k / (double)(Ns * R);
        x -= rint(x);
        cplx w[R];
        w[1] = expi_cycles(x);
#pragma unroll
        for (int r = 2; r < R; ++r) w[r] = cmul(w[r / 2], w[r - r / 2]);
#pragma unroll
        for (int r = 1; r < R; ++r) v[r] = cmul(v[r], w[r]);
    }
    dft<R>(v);
    const int64_t j0 = (j - k) * R + k;
#pragma unroll
    for (int r = 0; r < R; ++r) dst[j0 + r * Ns] = v[r];
}


// Two Stockham passes in one kernel: radix R = A * 16 (A = 2 .. 16).  A workgroup
// takes 16 consecutive butterflies of R points; 16 threads share one of them.
// With r = t + 16 m and q = q_lo + A q_hi:
//   X[q] = sum_t W_16^(t q_hi) [ W_R^(t q_lo) sum_m W_A^(m q_lo) x_(t + 16 m) ],
// thread t forms the inner A-point transforms in registers, the bracket goes
// through LDS (transposed), thread q_lo forms the outer 16-point transforms.
// One HBM round trip instead of two.  FIRST: the pass with Ns = 1 (no outer
// twiddles; results of one butterfly are contiguous, so lanes run over q_lo).
// PACKED (with FIRST): the input is not a complex buffer but the real series
// themselves, z[n] = x[2n] + i x[2n+1] formed while loading; samples past the
// end of a zero-padded series cost no memory traffic.
// STREAM: the results are written with non-temporal stores (for buffers that the
// last-level cache cannot hold until the next pass reads them; a template parameter
// because the compiler merges the two stores of a run-time choice into plain ones).
template <int A, bool FIRST, bool PACKED, bool STREAM>
__global__ __launch_bounds__(256) void stockham_fused_kernel(const cplx *__restrict__ in, PackBatch pb,
                                                             cplx *__restrict__ out, int64_t N, int64_t Ns)
{
    constexpr int R = A * 16;
    // (q_lo, t, jj) with jj fastest; one element of padding per q_lo keeps the
    // lanes of the FIRST read-back (which run over q_lo) on different banks
    __shared__ cplx lds[A * 256 + A];
    const int tid = threadIdx.x;
    const int t = tid >> 4, jj = tid & 15;
    const int64_t S = N / R;
    const int64_t jb = (int64_t)blockIdx.x * 16;
    const int64_t j = jb + jj;
    const cplx *src = in + (int64_t)blockIdx.y * N;
    cplx *dst = out + (int64_t)blockIdx.y * N;
    cplx v[A];
    if constexpr (PACKED) {
        const PackArgs &p = pb.s[blockIdx.y];
#pragma unroll
        for (int m = 0; m < A; ++m) {
            const int64_t n = j + (int64_t)(t + 16 * m) * S;
            v[m] = cplx{series_folded(p, 2 * n), series_folded(p, 2 * n + 1)};
        }
    } else {
#pragma unroll
        for (int m = 0; m < A; ++m) v[m] = src[j + (int64_t)(t + 16 * m) * S];
    }
    if (!FIRST) {
        // x_r *= exp(-2 pi i k r / (Ns R)), r = t + 16 m: exp(-2 pi i k t / (Ns R)) times
        // powers of exp(-2 pi i 16 k / (Ns R)); both angles are exact binary fractions
        const int64_t k = j & (Ns - 1);
        const double th = -(double)k / (double)(Ns * R);
        double xt = th * (double)t, xb = th * 16.0;
        xt -= rint(xt);
        xb -= rint(xb);
        const cplx wt = expi_cycles(xt);
        cplx pw[A];
        pw[0] = cplx{1.0, 0.0};
        pw[1] = expi_cycles(xb);
#pragma unroll
        for (int m = 2; m < A; ++m) pw[m] = cmul(pw[m / 2], pw[m - m / 2]);
        v[0] = cmul(v[0], wt);
#pragma unroll
        for (int m = 1; m < A; ++m) v[m] = cmul(v[m], cmul(wt, pw[m]));
    }
    dft<A>(v);
    {
        // W_R^(t q_lo)
        cplx pw[A];
        pw[0] = cplx{1.0, 0.0};
        pw[1] = expi_cycles(-(double)t / (double)R);
#pragma unroll
        for (int q = 2; q < A; ++q) pw[q] = cmul(pw[q / 2], pw[q - q / 2]);
#pragma unroll
        for (int q = 1; q < A; ++q) v[q] = cmul(v[q], pw[q]);
    }
#pragma unroll
    for (int q = 0; q < A; ++q) lds[(q * 16 + t) * 16 + jj + q] = v[q];
    __syncthreads();
    const int u = FIRST ? (tid & 15) : (tid >> 4);
    const int j2 = FIRST ? (tid >> 4) : (tid & 15);
    if (u < A) {
        cplx z[16];
#pragma unroll
        for (int tt = 0; tt < 16; ++tt) z[tt] = lds[(u * 16 + tt) * 16 + j2 + u];
        dft<16>(z);
        const int64_t jo = jb + j2;
        const int64_t ko = jo & (Ns - 1);
        const int64_t obase = (jo - ko) * R + ko;
        if constexpr (STREAM) {
            // a buffer larger than the last-level cache: written past it
#pragma unroll
            for (int h = 0; h < 16; ++h) {
                cplx *o = dst + obase + (int64_t)(u + A * h) * Ns;
                __builtin_nontemporal_store(z[h].x, &o->x);
                __builtin_nontemporal_store(z[h].y, &o->y);
            }
        } else {
#pragma unroll
            for (int h = 0; h < 16; ++h) dst[obase + (int64_t)(u + A * h) * Ns] = z[h];
        }
    }
}

__global__ void pack_kernel(PackBatch pb, int64_t N, cplx *z);

// Complex forward FFT of `batch` series of power-of-two length N >= 2.
// Returns the buffer (a or b) that holds the result.  The input is the
// content of `a`, or, with `pack`, the real series it describes (batch <= 3):
// long transforms form the complex input inside their first pass, short ones
// pack it into `a` first.
// With last_radix (N >= 256 * last_radix, N / (256 * last_radix) <= 256) the passes are
// radix 256, what is left, last_radix, and the last one is NOT run: the caller does it.
cplx *fft_c2c(cplx *a, cplx *b, int64_t N, int batch, hipStream_t st, const PackBatch *pack = nullptr,
              int last_radix = 0)
{
    int lg = 0;
    while (((int64_t)1 << lg) < N) ++lg;
    int64_t Ns = 1;
    cplx *src = a, *dst = b;
    const bool packed_first = pack && lg >= 12;
    if (packed_first) {
        src = b;   // not read by the first pass; it writes a
        dst = a;
    } else if (pack) {
        hipLaunchKernelGGL(pack_kernel, dim3((unsigned)ceil_div(N, 256), (unsigned)batch), dim3(256), 0, st, *pack,
                           N, a);
    }
    const PackBatch nopack{};
    auto launch = [&](int R) {
        dim3 grid((unsigned)ceil_div(N / R, 256), (unsigned)batch);
        switch (R) {
        case 16:
            hipLaunchKernelGGL(stockham_pass_kernel<16>, grid, dim3(256), 0, st, (const cplx *)src, dst, N, Ns);
            break;
        case 8:
            hipLaunchKernelGGL(stockham_pass_kernel<8>, grid, dim3(256), 0, st, (const cplx *)src, dst, N, Ns);
            break;
        case 4:
            hipLaunchKernelGGL(stockham_pass_kernel<4>, grid, dim3(256), 0, st, (const cplx *)src, dst, N, Ns);
            break;
        default:
            hipLaunchKernelGGL(stockham_pass_kernel<2>, grid, dim3(256), 0, st, (const cplx *)src, dst, N, Ns);
            break;
        }
        Ns *= R;
        cplx *t = src;
        src = dst;
        dst = t;
    };
    // 256 MiB of last-level cache: what is larger than half of it is written past it
    const bool stream_out = (size_t)batch * (size_t)N * sizeof(cplx) > ((size_t)128 << 20);
    auto launch_fused = [&](int A, bool first) {
        const int R = A * 16;
        dim3 grid((unsigned)(N / R / 16), (unsigned)batch);
#define FUSED(A_)                                                                                     \
    if (A == A_) {                                                                                    \
        if (first && stream_out)                                                                      \
            hipLaunchKernelGGL((stockham_fused_kernel<A_, true, false, true>), grid, dim3(256), 0, st, \
                               (const cplx *)src, nopack, dst, N, Ns);                                \
        else if (first)                                                                               \
            hipLaunchKernelGGL((stockham_fused_kernel<A_, true, false, false>), grid, dim3(256), 0, st, \
                               (const cplx *)src, nopack, dst, N, Ns);                                \
        else if (stream_out)                                                                          \
            hipLaunchKernelGGL((stockham_fused_kernel<A_, false, false, true>), grid, dim3(256), 0, st, \
                               (const cplx *)src, nopack, dst, N, Ns);                                \
        else                                                                                          \
            hipLaunchKernelGGL((stockham_fused_kernel<A_, false, false, false>), grid, dim3(256), 0, st, \
                               (const cplx *)src, nopack, dst, N, Ns);                                \
    }
        if (first && packed_first) {
            if (stream_out)
                hipLaunchKernelGGL((stockham_fused_kernel<16, true, true, true>), grid, dim3(256), 0, st,
                                   (const cplx *)nullptr, *pack, dst, N, Ns);
            else
                hipLaunchKernelGGL((stockham_fused_kernel<16, true, true, false>), grid, dim3(256), 0, st,
                                   (const cplx *)nullptr, *pack, dst, N, Ns);
        } else {
            FUSED(2) FUSED(4) FUSED(8) FUSED(16)
        }
#undef FUSED
        Ns *= R;
        cplx *t = src;
        src = dst;
        dst = t;
    };
    if (last_radix) {
        launch_fused(16, true);
        const int64_t mid = N / 256 / last_radix;
        if (mid >= 32) launch_fused((int)(mid / 16), false);
        else if (mid > 1) launch((int)mid);
        return src;
    }
    if (lg >= 12) {
        // radix-256 passes fused through LDS (one HBM round trip each), then what
        // is left of log2 N: up to four bits as a plain pass, five to seven as a
        // fused pass of radix 32 .. 128
        const int nfused = lg / 8, rem = lg % 8;
        for (int p = 0; p < nfused; ++p) launch_fused(16, p == 0);
        if (rem >= 5) launch_fused(1 << (rem - 4), false);
        else if (rem) launch(1 << rem);
        return src;
    }
    // short transforms: the odd-sized pass goes first (Ns = 1: no twiddles)
    int rem = lg % 4;
    if (rem) launch(1 << rem);
    for (int p = 0; p < lg / 4; ++p) launch(16);
    return src;
}

__global__ void pack_kernel(PackBatch pb, int64_t N, cplx *z)
{
    int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const PackArgs &p = pb.s[blockIdx.y];
    z[(int64_t)blockIdx.y * N + n] = cplx{series_folded(p, 2 * n), series_folded(p, 2 * n + 1)};
}


// X[k] of the real transform of length 2N from the packed transform Z (length N)
__device__ __forceinline__ cplx unpack_bin(const cplx *Z, int64_t N, int64_t k)
{
    cplx zk = Z[k == N ? 0 : k];
    cplx zn = Z[k == 0 ? 0 : N - k];
    double x = -(double)k / (double)(2 * N);
    x -= rint(x);
    return unpack_pair(zk, zn, expi_cycles(x));
}

__global__ void unpack_kernel(const cplx *Z, int64_t N, int64_t nout, cplx *X)
{
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nout) return;
    X[k] = unpack_bin(Z, N, k);
}


// ---- power_spectrum_fft epilogue (core.py:2683-2741) -----------------------
// Z1: packed transforms of (W-S), (Wb-B), length N1 = l/2 each.
// Z2: packed transforms of WW, WbWb, W-WW folded to length l/2, length N2 = l/4 each.
__global__ void ps_epilogue_kernel(const cplx *Z1, int64_t N1, const cplx *Z2, int64_t N2, int64_t nk,
                                   double *p0, double *p1, double *pb, double *amps)
{
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nk) return;
    cplx f1 = unpack_bin(Z1, N1, k);
    cplx f2 = unpack_bin(Z1 + N1, N1, k);
    double WmS_cos = f1.x, WmS_sin = -f1.y;
    double WbmB_cos = f2.x, WbmB_sin = -f2.y;
    // zero-frequency bins of the three second-moment series
    double S2 = unpack_bin(Z2, N2, 0).x;
    double B2 = unpack_bin(Z2 + N2, N2, 0).x;
    double SB = unpack_bin(Z2 + 2 * N2, N2, 0).x;
    double g3 = unpack_bin(Z2, N2, k).x;            // Re rfft(WW)[2k]
    double g4 = unpack_bin(Z2 + N2, N2, k).x;       // Re rfft(WbWb)[2k]
    double g5 = unpack_bin(Z2 + 2 * N2, N2, k).x;   // Re rfft(W-WW)[2k]
    ps_estimators(k, nk, WmS_cos, WmS_sin, WbmB_cos, WbmB_sin, S2, B2, SB, g3, g4, g5, p0, p1, pb, amps);
}

// The same estimators from the five transforms handed over one by one, each cut to
// what the estimators read: F1 = rfft(W - S)[:nk], F2 = rfft(Wb - B)[:nk] and the even
// bins G3 = rfft(WW)[0 : 2 nk : 2], G4 = rfft(WbWb)[...], G5 = rfft(W - WW)[...]
// (core.py:2683-2741) -- the form in which transforms computed on different GPUs arrive.
__global__ void ps_epilogue_parts_kernel(const cplx *F1, const cplx *F2, const cplx *G3, const cplx *G4,
                                         const cplx *G5, int64_t nk, double *p0, double *p1, double *pb,
                                         double *amps)
{
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nk) return;
    const cplx f1 = F1[k], f2 = F2[k];
    double WmS_cos = f1.x, WmS_sin = -f1.y;
    double WbmB_cos = f2.x, WbmB_sin = -f2.y;
    const double S2 = G3[0].x, B2 = G4[0].x, SB = G5[0].x;
    const double g3 = G3[k].x, g4 = G4[k].x, g5 = G5[k].x;
    ps_estimators(k, nk, WmS_cos, WmS_sin, WbmB_cos, WbmB_sin, S2, B2, SB, g3, g4, g5, p0, p1, pb, amps);
}

// ---- last Stockham pass of the five transforms + split + estimators in one kernel ------
// For N1 = 256 S (first-moment transforms) and N2 = 128 S (second-moment transforms) the last
// pass of radix R = 256 or 128 forms Z[j + q S], q < R, from column j of the data before it.
// Bin k = j + q S of the real transform needs Z[k] and Z[N - k], and
// N - k = (S - j) + (R - 1 - q) S: the partner lives in column S - j.  A workgroup therefore
// takes eight columns j = 8 w + c and their eight partners S - j, runs the last pass of each of
// the five series on those sixteen columns (the two radix-16 stages of stockham_fused_kernel),
// leaves the sixteen transformed columns in LDS as [q][column], and every thread picks the pairs
// of its eight bins from there.  Only the three real parts of the second-moment bins and the
// first of the two first-moment bins wait in registers for the estimators, which are evaluated
// as the last series arrives; the transforms themselves are never written to memory.
// Columns 0 and S/2 are their own partners ((R - q) mod R and R - 1 - q in the same column):
// the first workgroup takes both, where the others have column 8 w and its partner.
// The loads of a series are issued before the pairs of the series before it are worked on.
// The bin k = l/4 (column 0, q = 128) and the zero-frequency values that every bin needs
// come from ps_dc_kernel.

template <int A>
__device__ __forceinline__ void final_pass_load(const cplx *__restrict__ src, int64_t S, int64_t jc, int tid,
                                                cplx (&v)[A])
{
    const int t = tid >> 4;
#pragma unroll
    for (int m = 0; m < A; ++m) v[m] = src[jc + (int64_t)(t + 16 * m) * S];
    // what follows in the program stays behind these loads
    __builtin_amdgcn_sched_barrier(0);
}

// One series: the radix-(16 A) butterflies of sixteen columns, result in lds[q * 16 + column].
template <int A>
__device__ __forceinline__ void final_pass_columns(cplx (&v)[A], int64_t S, int64_t jc, cplx *lds, int tid)
{
    constexpr int R = A * 16;
    const int t = tid >> 4, col = tid & 15;
    // The twiddles are the same for every series of one radix; kept across the series they
    // would hold over a hundred registers, so they are formed anew from an angle the compiler
    // cannot tell from the last one.
    double th = -(double)jc / (double)(S * R);
    double tr = -(double)t / (double)R;
    asm volatile("" : "+v"(th), "+v"(tr));
    {
        // x_r *= exp(-2 pi i jc r / (S R)), r = t + 16 m
        double xt = th * (double)t, xb = th * 16.0;
        xt -= rint(xt);
        xb -= rint(xb);
        const cplx step = expi_cycles(xb);
        cplx tw = expi_cycles(xt);
#pragma unroll
        for (int m = 0; m < A; ++m) {
            v[m] = cmul(v[m], tw);
            if (m + 1 < A) tw = cmul(tw, step);
        }
    }
    dft<A>(v);
    {
        const cplx step = expi_cycles(tr);   // W_R^t
        cplx tw = step;
#pragma unroll
        for (int q = 1; q < A; ++q) {
            v[q] = cmul(v[q], tw);
            if (q + 1 < A) tw = cmul(tw, step);
        }
    }
    __syncthreads();   // the pairs of the series before this one have been read
#pragma unroll
    for (int q = 0; q < A; ++q) lds[(q * 16 + t) * 16 + col + q] = v[q];
    __syncthreads();
    const int u = tid >> 4;
    cplx z[16];
    if (u < A) {
#pragma unroll
        for (int tt = 0; tt < 16; ++tt) z[tt] = lds[(u * 16 + tt) * 16 + col + u];
        dft<16>(z);
    }
    __syncthreads();
    if (u < A) {
#pragma unroll
        for (int h = 0; h < 16; ++h) lds[(u + A * h) * 16 + col] = z[h];
    }
    __syncthreads();
}

struct FinalArgs {
    const cplx *y1;   // [2][N1] before the last pass
    const cplx *y2;   // [3][N2] before the last pass
    int64_t S;        // N1 / 256 = N2 / 128
    const double *dc; // S2, B2, SB
    double *p0, *p1, *pb, *amps;
    int64_t nk;
};

__global__ __launch_bounds__(256, 2) void ps_final_kernel(FinalArgs a)
{
    __shared__ cplx lds[16 * 256 + 16];
    const int tid = threadIdx.x;
    const int col = tid & 15, qi = tid >> 4;
    const int64_t S = a.S, N1 = 256 * S, N2 = 128 * S;
    const int64_t ntile = S / 16;
    // consecutive tiles share cache lines of the partner columns: keep them on one XCD
    // (workgroups are dealt to the eight XCDs in turn)
    int64_t w = blockIdx.x;
    if ((ntile & 7) == 0) w = (w & 7) * (ntile >> 3) + (w >> 3);
    const int64_t ja = 8 * w + (col & 7);
    const bool selfp = ja == 0;   // tile 0: column 0 and, in place of its partner, column S/2
    const int64_t jc = col < 8 ? ja : (selfp ? S / 2 : S - ja);   // the column of this lane
    const int mcol = selfp ? col : (col ^ 8);
    const bool wrap = selfp && col < 8;   // column 0: partner of q is (R - q) mod R
    // exp(-2 pi i k / l) for k = jc + (qi + 16 h) S steps by exp(-2 pi i / 32) in h
    const cplx w0 = expi_cycles(-(double)(jc + (int64_t)qi * S) / (double)(2 * N1));
    const cplx rho{0.98078528040323044913, -0.19509032201612826785};
    double g3[8], g4[8], g5[8];
    cplx f1[8];

    auto second_moment = [&](double (&g)[8]) {
        cplx w1 = w0;
#pragma unroll
        for (int h = 0; h < 8; ++h) {
            const int q = qi + 16 * h;
            const int mq = wrap ? ((128 - q) & 127) : 127 - q;
            const cplx zk = lds[q * 16 + col], zn = lds[mq * 16 + mcol];
            const cplx w2 = cmul(w1, w1);   // exp(-2 pi i k / (2 N2))
            const double ox = 0.5 * (zk.y + zn.y), oy = -0.5 * (zk.x - zn.x);
            g[h] = 0.5 * (zk.x + zn.x) + fma(w2.x, ox, -(w2.y * oy));
            asm volatile("" : "+v"(g[h]));   // formed here, not where it is used: the pair would wait
            w1 = cmul(w1, rho);
        }
    };
    cplx va[8], vb[8], vc[16], vd[16];
    final_pass_load<8>(a.y2, S, jc, tid, va);
    final_pass_columns<8>(va, S, jc, lds, tid);
    final_pass_load<8>(a.y2 + N2, S, jc, tid, vb);
    second_moment(g3);
    final_pass_columns<8>(vb, S, jc, lds, tid);
    final_pass_load<8>(a.y2 + 2 * N2, S, jc, tid, va);
    second_moment(g4);
    final_pass_columns<8>(va, S, jc, lds, tid);
    final_pass_load<16>(a.y1, S, jc, tid, vc);
    second_moment(g5);
    final_pass_columns<16>(vc, S, jc, lds, tid);
    final_pass_load<16>(a.y1 + N1, S, jc, tid, vd);
    {
        cplx w1 = w0;
#pragma unroll
        for (int h = 0; h < 8; ++h) {
            const int q = qi + 16 * h;
            const int mq = wrap ? ((256 - q) & 255) : 255 - q;
            f1[h] = unpack_pair(lds[q * 16 + col], lds[mq * 16 + mcol], w1);
            asm volatile("" : "+v"(f1[h].x), "+v"(f1[h].y));
            w1 = cmul(w1, rho);
        }
    }
    final_pass_columns<16>(vd, S, jc, lds, tid);
    const double S2 = a.dc[0], B2 = a.dc[1], SB = a.dc[2];
    cplx w1 = w0;
#pragma unroll
    for (int h = 0; h < 8; ++h) {
        const int q = qi + 16 * h;
        const int mq = wrap ? ((256 - q) & 255) : 255 - q;
        const cplx f2 = unpack_pair(lds[q * 16 + col], lds[mq * 16 + mcol], w1);
        w1 = cmul(w1, rho);
        const int64_t k = jc + (int64_t)q * S;
        ps_estimators(k, a.nk, f1[h].x, -f1[h].y, f2.x, -f2.y, S2, B2, SB, g3[h], g4[h], g5[h], a.p0, a.p1, a.pb,
                      a.amps);
        __builtin_amdgcn_sched_barrier(0);   // one bin at a time: interleaving them costs registers
    }
}

// What ps_final_kernel does not reach: the zero-frequency values S2, B2, SB of the
// second-moment series (sum of column 0 before the last pass: real + imaginary part of
// Z[0]) and the bin k = l/4, made of Z1[N1/2] (alternating sum of column 0) and Z2[0].
__global__ __launch_bounds__(256) void ps_dc_kernel(FinalArgs a, double *dc)
{
    __shared__ double red[10][256];
    const int r = threadIdx.x;
    const int64_t S = a.S, N1 = 256 * S, N2 = 128 * S;
    const double sg = (r & 1) ? -1.0 : 1.0;
    const cplx u1 = a.y1[(int64_t)r * S], u2 = a.y1[N1 + (int64_t)r * S];
    red[0][r] = sg * u1.x;
    red[1][r] = sg * u1.y;
    red[2][r] = sg * u2.x;
    red[3][r] = sg * u2.y;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        cplx v{0.0, 0.0};
        if (r < 128) v = a.y2[s * N2 + (int64_t)r * S];
        red[4 + 2 * s][r] = v.x;
        red[5 + 2 * s][r] = v.y;
    }
    __syncthreads();
    for (int half = 128; half >= 1; half >>= 1) {
        if (r < half)
#pragma unroll
            for (int i = 0; i < 10; ++i) red[i][r] += red[i][r + half];
        __syncthreads();
    }
    if (r == 0) {
        const double S2 = red[4][0] + red[5][0], B2 = red[6][0] + red[7][0], SB = red[8][0] + red[9][0];
        dc[0] = S2;
        dc[1] = B2;
        dc[2] = SB;
        // k = l/4: F = conj Z1[N1/2]; even bins of the second moments at k = N2: Re Z2[0] - Im Z2[0]
        ps_estimators(a.nk - 1, a.nk, red[0][0], red[1][0], red[2][0], red[3][0], S2, B2, SB,
                      red[4][0] - red[5][0], red[6][0] - red[7][0], red[8][0] - red[9][0], a.p0, a.p1, a.pb,
                      a.amps);
    }
}

// power_spectrum_simple epilogue (core.py:2766-2782)
__global__ void ps_simple_epilogue_kernel(const cplx *Z1, int64_t N1, const cplx *Z2, int64_t N2, int64_t nk,
                                          double *dlogl, double *moments)
{
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nk) return;
    cplx f1 = unpack_bin(Z1, N1, k);
    double c = f1.x, s = -f1.y;
    double S2 = unpack_bin(Z2, N2, 0).x;
    double g = unpack_bin(Z2, N2, k).x;
    double cc = 0.5 * (S2 + g), ss = 0.5 * (S2 - g);
    if (k == 0) ss = 1.;
    if (dlogl) dlogl[k] = c * c / cc + s * s / ss;
    if (moments) {
        moments[0 * nk + k] = c;
        moments[1 * nk + k] = s;
        moments[2 * nk + k] = cc;
        moments[3 * nk + k] = ss;
    }
}

inline bool is_pow2(int64_t v) { return v > 0 && (v & (v - 1)) == 0; }
inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// ---- arbitrary transform lengths: Bluestein's chirp-z form -----------------
// X[k] = c[k] * sum_n (x[n] c[n]) d[k-n],  c[n] = exp(-i pi n^2 / L), d = conj(c):
// a circular convolution of length M >= 2L-1 (a power of two) done with the
// Stockham transform above.  n^2 is reduced modulo 2L in integers, so the
// chirp phases are exact however long the series is.
__device__ __forceinline__ cplx chirp(int64_t n, int64_t L, double sign)
{
    unsigned __int128 sq = (unsigned __int128)(uint64_t)n * (uint64_t)n;
    int64_t r = (int64_t)(sq % (unsigned __int128)(2 * L));   // n^2 mod 2L
    double x = (double)r / (double)(2 * L);                   // cycles in [0, 1)
    x -= rint(x);
    return expi_cycles(sign * x);
}

__global__ void bluestein_kernel_b(int64_t L, int64_t M, cplx *b)
{
    int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    cplx v{0.0, 0.0};
    if (m < L) v = chirp(m, L, +1.0);
    else if (M - m < L) v = chirp(M - m, L, +1.0);
    b[m] = v;
}

__global__ void bluestein_pre_kernel(PackArgs p, int64_t L, int64_t M, cplx *a)
{
    int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= M) return;
    cplx v{0.0, 0.0};
    if (n < L) {
        cplx c = chirp(n, L, -1.0);
        double x = series_at(p, n);
        v = cplx{x * c.x, x * c.y};
    }
    a[n] = v;
}

// y = conj(A * Bf): its forward transform is M * conj(inverse transform of A * Bf)
__global__ void bluestein_mul_kernel(const cplx *A, const cplx *Bf, int64_t M, cplx *y)
{
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= M) return;
    cplx t = cmul(A[k], Bf[k]);
    y[k] = cplx{t.x, -t.y};
}

__global__ void bluestein_post_kernel(const cplx *Y, int64_t L, int64_t M, int64_t nout, cplx *X)
{
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nout) return;
    cplx y = Y[k];
    double inv = 1.0 / (double)M;
    cplx conv{y.x * inv, -y.y * inv};
    cplx x = cmul(chirp(k, L, -1.0), conv);
    // the series is real: the zero-frequency and Nyquist bins are real, exactly (as numpy's
    // rfft returns them), not merely to rounding
    if (k == 0 || 2 * k == L) x.y = 0.0;
    X[k] = x;
}

inline int64_t bluestein_M(int64_t L)
{
    int64_t M = 1;
    while (M < 2 * L - 1) M <<= 1;
    return M;
}

// workspace: Bf, a, tmp (M complex each)
inline size_t bluestein_ws(int64_t L) { return 3 * align256(sizeof(cplx) * (size_t)bluestein_M(L)); }

// Transform of `batch` real series of length L into X[s][0..nout-1].
void bluestein_rfft(const PackArgs *series, int batch, int64_t L, int64_t nout, cplx *X, int64_t xstride,
                    void *workspace, hipStream_t st)
{
    const int64_t M = bluestein_M(L);
    size_t sz = align256(sizeof(cplx) * (size_t)M);
    cplx *Bf = (cplx *)workspace;
    cplx *a = (cplx *)((char *)workspace + sz);
    cplx *t = (cplx *)((char *)workspace + 2 * sz);
    unsigned gm = (unsigned)ceil_div(M, 256);
    hipLaunchKernelGGL(bluestein_kernel_b, dim3(gm), dim3(256), 0, st, L, M, a);
    cplx *bf = fft_c2c(a, t, M, 1, st);
    if (bf != Bf) (void)hipMemcpyAsync(Bf, bf, sizeof(cplx) * (size_t)M, hipMemcpyDeviceToDevice, st);
    for (int sidx = 0; sidx < batch; ++sidx) {
        hipLaunchKernelGGL(bluestein_pre_kernel, dim3(gm), dim3(256), 0, st, series[sidx], L, M, a);
        cplx *A = fft_c2c(a, t, M, 1, st);
        cplx *other = (A == a) ? t : a;
        hipLaunchKernelGGL(bluestein_mul_kernel, dim3(gm), dim3(256), 0, st, (const cplx *)A, (const cplx *)Bf, M,
                           other);
        cplx *Y = fft_c2c(other, A, M, 1, st);
        hipLaunchKernelGGL(bluestein_post_kernel, dim3((unsigned)ceil_div(nout, 256)), dim3(256), 0, st,
                           (const cplx *)Y, L, M, nout, X + (int64_t)sidx * xstride);
    }
}

// epilogues reading full spectra F[s][0..l/2] (any transform length)
__global__ void ps_epilogue_full_kernel(const cplx *F, int64_t fstride, int64_t nk, double *p0, double *p1,
                                        double *pb, double *amps)
{
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nk) return;
    cplx f1 = F[k], f2 = F[fstride + k];
    double WmS_cos = f1.x, WmS_sin = -f1.y;
    double WbmB_cos = f2.x, WbmB_sin = -f2.y;
    double S2 = F[2 * fstride].x, B2 = F[3 * fstride].x, SB = F[4 * fstride].x;
    double g3 = F[2 * fstride + 2 * k].x, g4 = F[3 * fstride + 2 * k].x, g5 = F[4 * fstride + 2 * k].x;
    ps_estimators(k, nk, WmS_cos, WmS_sin, WbmB_cos, WbmB_sin, S2, B2, SB, g3, g4, g5, p0, p1, pb, amps);
}

__global__ void ps_simple_epilogue_full_kernel(const cplx *F, int64_t fstride, int64_t nk, double *dlogl,
                                               double *moments)
{
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nk) return;
    cplx f1 = F[k];
    double c = f1.x, s = -f1.y;
    double S2 = F[fstride].x, g = F[fstride + 2 * k].x;
    double cc = 0.5 * (S2 + g), ss = 0.5 * (S2 - g);
    if (k == 0) ss = 1.;
    if (dlogl) dlogl[k] = c * c / cc + s * s / ss;
    if (moments) {
        moments[0 * nk + k] = c;
        moments[1 * nk + k] = s;
        moments[2 * nk + k] = cc;
        moments[3 * nk + k] = ss;
    }
}

}   // namespace

extern "C" size_t godot_rfft_workspace_bytes(int64_t l)
{
    if (l < 2) return 0;
    if (is_pow2(l) && l >= 4) return 2 * align256(sizeof(cplx) * (size_t)(l / 2));
    return bluestein_ws(l);
}

extern "C" int godot_rfft_f64(const double *x, int64_t n, int64_t l, double *out, void *workspace,
                              size_t workspace_bytes, godot_stream_t stream)
{
    GODOT_ENTRY();
    GODOT_REQUIRE(n >= 0 && l >= 2, GODOT_E_INVALID, "godot_rfft_f64: need n >= 0 and l >= 2");
    GODOT_REQUIRE(l < ((int64_t)1 << 40), GODOT_E_UNSUPPORTED, "godot_rfft_f64: l too large");
    GODOT_REQUIRE((x || n == 0) && out, GODOT_E_INVALID, "godot_rfft_f64: null argument");
    GODOT_REQUIRE(workspace && workspace_bytes >= godot_rfft_workspace_bytes(l), GODOT_E_WORKSPACE,
                  "godot_rfft_f64: workspace smaller than godot_rfft_workspace_bytes(l)");
    hipStream_t st = (hipStream_t)stream;
    if (!(is_pow2(l) && l >= 4)) {
        PackArgs pa{x, nullptr, n < l ? n : l, 0};
        bluestein_rfft(&pa, 1, l, l / 2 + 1, (cplx *)out, 0, workspace, st);
        GODOT_LAUNCH_CHECK("godot_rfft_f64");
        return GODOT_OK;
    }
    const int64_t N = l / 2;
    cplx *a = (cplx *)workspace;
    cplx *b = (cplx *)((char *)workspace + align256(sizeof(cplx) * (size_t)N));
    PackBatch pb{};
    pb.s[0] = PackArgs{x, nullptr, n < l ? n : l, 0};
    cplx *Z = fft_c2c(a, b, N, 1, st, &pb);
    hipLaunchKernelGGL(unpack_kernel, dim3((unsigned)ceil_div(N + 1, 256)), dim3(256), 0, st, (const cplx *)Z, N,
                       N + 1, (cplx *)out);
    GODOT_LAUNCH_CHECK("godot_rfft_f64");
    return GODOT_OK;
}

extern "C" size_t godot_power_spectrum_workspace_bytes(int64_t l)
{
    if (l < 8) return 0;
    if (!is_pow2(l))   // five full spectra + the chirp-z buffers
        return 5 * align256(sizeof(cplx) * (size_t)(l / 2 + 1)) + bluestein_ws(l);
    size_t N1 = (size_t)l / 2, N2 = (size_t)l / 4;
    size_t need = 2 * align256(sizeof(cplx) * 2 * N1) + 2 * align256(sizeof(cplx) * 3 * N2);
    if (rows_path_applies(1, l) && rows_path_workspace_bytes(l) > need) need = rows_path_workspace_bytes(l);
    return need;
}

extern "C" int godot_power_spectrum_parts_supported(int64_t n, int64_t l, int nparts)
{
    return godot_spectrum::parts_supported(n, l, nparts) ? 1 : 0;
}

extern "C" int64_t godot_power_spectrum_part_rows(void) { return godot_spectrum::part_rows(); }

extern "C" size_t godot_power_spectrum_part_bytes(int64_t l, int nparts)
{
    if (l <= 0 || nparts < 1) return 0;
    return 16 * godot_spectrum::part_elements(l, nparts);
}

extern "C" size_t godot_power_spectrum_part_workspace_bytes(int64_t l, int nparts)
{
    if (l <= 0 || nparts < 1) return 0;
    return godot_spectrum::part_workspace_bytes(l, nparts);
}

extern "C" int godot_power_spectrum_columns(const double *W, const double *WW, const double *S, const double *Wb,
                                            const double *WbWb, const double *B, int64_t n, int64_t l, int part,
                                            int nparts, void *v_part, void *workspace, size_t workspace_bytes,
                                            godot_stream_t stream)
{
    GODOT_ENTRY();
    GODOT_REQUIRE(godot_spectrum::parts_supported(n, l, nparts), GODOT_E_UNSUPPORTED,
                  "godot_power_spectrum_columns: n = %lld, l = %lld in %d parts is not served "
                  "(godot_power_spectrum_parts_supported)", (long long)n, (long long)l, nparts);
    GODOT_REQUIRE(part >= 0 && part < nparts, GODOT_E_INVALID, "godot_power_spectrum_columns: part %d of %d", part,
                  nparts);
    GODOT_REQUIRE(W && WW && S && Wb && WbWb && B && v_part, GODOT_E_INVALID,
                  "godot_power_spectrum_columns: null argument");
    GODOT_REQUIRE(workspace && workspace_bytes >= godot_spectrum::part_workspace_bytes(l, nparts), GODOT_E_WORKSPACE,
                  "godot_power_spectrum_columns: workspace smaller than godot_power_spectrum_part_workspace_bytes");
    godot_spectrum::power_spectrum_columns_part(W, WW, S, Wb, WbWb, B, n, l, part, nparts, v_part, workspace,
                                                (hipStream_t)stream);
    GODOT_LAUNCH_CHECK("godot_power_spectrum_columns");
    return GODOT_OK;
}

extern "C" int godot_power_spectrum_rows(const void *v_rows, const double *dc, int64_t l, int64_t m_first,
                                         int64_t m_end, double *p0, double *p1, double *pb, double *amps,
                                         godot_stream_t stream)
{
    GODOT_ENTRY();
    GODOT_REQUIRE(godot_spectrum::parts_supported(1, l, 1), GODOT_E_UNSUPPORTED,
                  "godot_power_spectrum_rows: l = %lld is not served", (long long)l);
    GODOT_REQUIRE(m_first >= 0 && m_first < m_end && m_end <= godot_spectrum::part_rows() && m_first % 16 == 0,
                  GODOT_E_INVALID, "godot_power_spectrum_rows: rows %lld .. %lld (the first a multiple of 16, the "
                  "last at most %lld)", (long long)m_first, (long long)m_end, (long long)godot_spectrum::part_rows());
    GODOT_REQUIRE(v_rows && dc && p0 && p1 && pb, GODOT_E_INVALID, "godot_power_spectrum_rows: null argument");
    godot_spectrum::power_spectrum_rows_part(v_rows, dc, l, m_first, m_end, p0, p1, pb, amps, (hipStream_t)stream);
    GODOT_LAUNCH_CHECK("godot_power_spectrum_rows");
    return GODOT_OK;
}

extern "C" int godot_power_spectrum_fft(const double *W, const double *WW, const double *S, const double *Wb,
                                        const double *WbWb, const double *B, int64_t n, int64_t l, double *p0,
                                        double *p1, double *pb, double *amps, void *workspace,
                                        size_t workspace_bytes, godot_stream_t stream)
{
    GODOT_ENTRY();
    GODOT_REQUIRE(n >= 1 && l >= 8, GODOT_E_INVALID, "godot_power_spectrum_fft: need n >= 1 and l >= 8");
    GODOT_REQUIRE(l < ((int64_t)1 << 40), GODOT_E_UNSUPPORTED, "godot_power_spectrum_fft: l too large");
    GODOT_REQUIRE(n <= l, GODOT_E_INVALID, "godot_power_spectrum_fft: n exceeds the padded length");
    GODOT_REQUIRE(W && WW && S && Wb && WbWb && B, GODOT_E_INVALID, "godot_power_spectrum_fft: null series");
    GODOT_REQUIRE(workspace && workspace_bytes >= godot_power_spectrum_workspace_bytes(l), GODOT_E_WORKSPACE,
                  "godot_power_spectrum_fft: workspace smaller than godot_power_spectrum_workspace_bytes(l)");
    hipStream_t st = (hipStream_t)stream;
    if (!is_pow2(l)) {
        // any length: five chirp-z transforms, then the epilogue on the full spectra
        GODOT_REQUIRE(((l / 2) / 2) * 2 <= l / 2 && 2 * (l / 4) <= l / 2, GODOT_E_INVALID,
                      "godot_power_spectrum_fft: bad length");
        const int64_t nf = l / 2 + 1, nkk = l / 4 + 1;
        size_t fsz = align256(sizeof(cplx) * (size_t)nf);
        cplx *F = (cplx *)workspace;
        PackArgs ser[5] = {PackArgs{W, S, n, 0}, PackArgs{Wb, B, n, 0}, PackArgs{WW, nullptr, n, 0},
                           PackArgs{WbWb, nullptr, n, 0}, PackArgs{W, WW, n, 0}};
        bluestein_rfft(ser, 5, l, nf, F, (int64_t)(fsz / sizeof(cplx)), (char *)workspace + 5 * fsz, st);
        hipLaunchKernelGGL(ps_epilogue_full_kernel, dim3((unsigned)ceil_div(nkk, 256)), dim3(256), 0, st,
                           (const cplx *)F, (int64_t)(fsz / sizeof(cplx)), nkk, p0, p1, pb, amps);
        GODOT_LAUNCH_CHECK("godot_power_spectrum_fft");
        return GODOT_OK;
    }
    if (rows_path_applies(n, l) && !getenv("GODOT_FFT_THREE_TRIPS")) {
        // long padded series: column transforms, then row transforms with the estimators
        power_spectrum_rows(W, WW, S, Wb, WbWb, B, n, l, p0, p1, pb, amps, workspace, st);
        GODOT_LAUNCH_CHECK("godot_power_spectrum_fft");
        return GODOT_OK;
    }
    const int64_t N1 = l / 2, N2 = l / 4, nk = l / 4 + 1;
    char *ws = (char *)workspace;
    size_t sz1 = align256(sizeof(cplx) * 2 * (size_t)N1), sz2 = align256(sizeof(cplx) * 3 * (size_t)N2);
    cplx *a1 = (cplx *)ws, *b1 = (cplx *)(ws + sz1);
    cplx *a2 = (cplx *)(ws + 2 * sz1), *b2 = (cplx *)(ws + 2 * sz1 + sz2);
    // second-moment series: only the even bins of their length-l transforms are
    // used, which are the bins of the length-l/2 transform of the series wrapped at l/2
    PackBatch pb2{};
    pb2.s[0] = PackArgs{WW, nullptr, n, l / 2};
    pb2.s[1] = PackArgs{WbWb, nullptr, n, l / 2};
    pb2.s[2] = PackArgs{W, WW, n, l / 2};
    PackBatch pb1{};
    pb1.s[0] = PackArgs{W, S, n, 0};
    pb1.s[1] = PackArgs{Wb, B, n, 0};
    if (N1 >= ((int64_t)1 << 16) && N1 <= ((int64_t)1 << 24)) {
        // the last pass of all five transforms, the split and the estimators in one kernel
        cplx *Y2 = fft_c2c(a2, b2, N2, 3, st, &pb2, 128);
        cplx *Y1 = fft_c2c(a1, b1, N1, 2, st, &pb1, 256);
        double *dc = (double *)(Y2 == a2 ? b2 : a2);   // the buffer the last pass would have written
        FinalArgs fa{Y1, Y2, N1 / 256, dc, p0, p1, pb, amps, nk};
        hipLaunchKernelGGL(ps_dc_kernel, dim3(1), dim3(256), 0, st, fa, dc);
        hipLaunchKernelGGL(ps_final_kernel, dim3((unsigned)(fa.S / 16)), dim3(256), 0, st, fa);
        GODOT_LAUNCH_CHECK("godot_power_spectrum_fft");
        return GODOT_OK;
    }
    cplx *Z2 = fft_c2c(a2, b2, N2, 3, st, &pb2);
    cplx *Z1 = fft_c2c(a1, b1, N1, 2, st, &pb1);
    hipLaunchKernelGGL(ps_epilogue_kernel, dim3((unsigned)ceil_div(nk, 256)), dim3(256), 0, st, (const cplx *)Z1,
                       N1, (const cplx *)Z2, N2, nk, p0, p1, pb, amps);
    GODOT_LAUNCH_CHECK("godot_power_spectrum_fft");
    return GODOT_OK;
}

extern "C" int godot_power_spectrum_from_parts(const double *f1, const double *f2, const double *g3,
                                               const double *g4, const double *g5, int64_t nk, double *p0,
                                               double *p1, double *pb, double *amps, godot_stream_t stream)
{
    GODOT_ENTRY();
    GODOT_REQUIRE(nk >= 1, GODOT_E_INVALID, "godot_power_spectrum_from_parts: need nk >= 1");
    GODOT_REQUIRE(f1 && f2 && g3 && g4 && g5, GODOT_E_INVALID, "godot_power_spectrum_from_parts: null transform");
    hipLaunchKernelGGL(ps_epilogue_parts_kernel, dim3((unsigned)ceil_div(nk, 256)), dim3(256), 0,
                       (hipStream_t)stream, (const cplx *)f1, (const cplx *)f2, (const cplx *)g3,
                       (const cplx *)g4, (const cplx *)g5, nk, p0, p1, pb, amps);
    GODOT_LAUNCH_CHECK("godot_power_spectrum_from_parts");
    return GODOT_OK;
}

namespace {
// freqs[i] = (i * step) / scale or (i * step) * scale: one rounding per operation, as numpy makes them
__global__ __launch_bounds__(256) void ps_frequencies_kernel(int64_t nk, double step, double scale, int divide,
                                                             double *__restrict__ freqs)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nk) return;
    const double x = (double)i * step;
    freqs[i] = divide ? x / scale : x * scale;
}
}   // namespace

extern "C" int godot_spectrum_frequencies(int64_t nk, double step, double scale, int divide, double *freqs,
                                          godot_stream_t stream)
{
    GODOT_ENTRY();
    GODOT_REQUIRE(nk >= 0, GODOT_E_INVALID, "godot_spectrum_frequencies: negative nk");
    if (nk == 0) return GODOT_OK;
    GODOT_REQUIRE(freqs, GODOT_E_INVALID, "godot_spectrum_frequencies: null pointer");
    hipLaunchKernelGGL(ps_frequencies_kernel, dim3((unsigned)ceil_div(nk, (int64_t)256)), dim3(256), 0,
                       (hipStream_t)stream, nk, step, scale, divide, freqs);
    GODOT_LAUNCH_CHECK("godot_spectrum_frequencies");
    return GODOT_OK;
}

namespace {
// (value, index) pairs in the order of numpy's argmax: a NaN before any number, the larger
// number, the smaller index among equals
struct Peak {
    double v;
    int64_t i;
};

__device__ __forceinline__ bool peak_before(double v, int64_t i, double bv, int64_t bi)
{
    const bool nv = v != v, nb = bv != bv;
    if (nv != nb) return nv;
    if (!nv && v != bv) return v > bv;
    return i < bi;
}

constexpr int PEAK_SPLIT = 64;      // workgroups a row at most
constexpr int PEAK_CHUNK = 1 << 15; // entries a workgroup at least

inline int peak_splits(int64_t ncol)
{
    const int64_t s = ceil_div(ncol, (int64_t)PEAK_CHUNK);
    return (int)(s < 1 ? 1 : s > PEAK_SPLIT ? PEAK_SPLIT : s);
}

__device__ __forceinline__ Peak block_peak(Peak p, Peak *sh)
{
    const int t = threadIdx.x;
    sh[t] = p;
    __syncthreads();
    for (int half = 128; half >= 1; half >>= 1) {
        if (t < half) {
            const Peak o = sh[t + half];
            if (peak_before(o.v, o.i, sh[t].v, sh[t].i)) sh[t] = o;
        }
        __syncthreads();
    }
    return sh[0];
}

__global__ __launch_bounds__(256) void row_peaks_part_kernel(const double *__restrict__ a, int64_t ncol, int64_t first,
                                                             int nsplit, Peak *__restrict__ part)
{
    __shared__ Peak sh[256];
    const int64_t row = blockIdx.y, n = ncol - first;
    const int64_t per = (n + nsplit - 1) / nsplit;
    const int64_t c0 = (int64_t)blockIdx.x * per, c1 = c0 + per < n ? c0 + per : n;
    const double *r = a + row * ncol + first;
    Peak best{-INFINITY, INT64_MAX};
    for (int64_t j = c0 + threadIdx.x; j < c1; j += 256) {
        const double v = r[j];
        if (peak_before(v, j, best.v, best.i)) best = Peak{v, j};
    }
    best = block_peak(best, sh);
    if (threadIdx.x == 0) part[row * nsplit + blockIdx.x] = best;
}

__global__ __launch_bounds__(256) void row_peaks_join_kernel(const Peak *__restrict__ part, int nsplit,
                                                             double *__restrict__ values, int64_t *__restrict__ indices)
{
    __shared__ Peak sh[256];
    const int64_t row = blockIdx.x;
    Peak best{-INFINITY, INT64_MAX};
    if ((int)threadIdx.x < nsplit) best = part[row * nsplit + threadIdx.x];
    best = block_peak(best, sh);
    if (threadIdx.x == 0) {
        values[row] = best.v;
        indices[row] = best.i;
    }
}
}   // namespace

extern "C" size_t godot_row_peaks_workspace_bytes(int64_t nrow, int64_t ncol)
{
    if (nrow <= 0 || ncol <= 0) return 0;
    return sizeof(Peak) * (size_t)nrow * (size_t)peak_splits(ncol);
}

extern "C" int godot_row_peaks(const double *a, int64_t nrow, int64_t ncol, int64_t first, double *values,
                               int64_t *indices, void *workspace, size_t workspace_bytes, godot_stream_t stream)
{
    GODOT_ENTRY();
    GODOT_REQUIRE(nrow >= 0 && ncol >= 0, GODOT_E_INVALID, "godot_row_peaks: negative size");
    GODOT_REQUIRE(first >= 0 && (nrow == 0 || first < ncol), GODOT_E_INVALID,
                  "godot_row_peaks: the rows hold nothing from the first column on");
    if (nrow == 0) return GODOT_OK;
    GODOT_REQUIRE(a && values && indices, GODOT_E_INVALID, "godot_row_peaks: null pointer");
    GODOT_REQUIRE(nrow <= 65535, GODOT_E_INVALID, "godot_row_peaks: more than 65 535 rows");
    GODOT_REQUIRE(workspace && workspace_bytes >= godot_row_peaks_workspace_bytes(nrow, ncol), GODOT_E_WORKSPACE,
                  "godot_row_peaks: workspace too small");
    const int nsplit = peak_splits(ncol);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(row_peaks_part_kernel, dim3((unsigned)nsplit, (unsigned)nrow), dim3(256), 0, st, a, ncol, first,
                       nsplit, (Peak *)workspace);
    hipLaunchKernelGGL(row_peaks_join_kernel, dim3((unsigned)nrow), dim3(256), 0, st, (const Peak *)workspace, nsplit,
                       values, indices);
    GODOT_LAUNCH_CHECK("godot_row_peaks");
    return GODOT_OK;
}

extern "C" int godot_power_spectrum_simple(const double *W, const double *WW, const double *S, int64_t n,
                                           double *dlogl, double *moments, void *workspace,
                                           size_t workspace_bytes, godot_stream_t stream)
{
    GODOT_ENTRY();
    GODOT_REQUIRE(n >= 8, GODOT_E_INVALID, "godot_power_spectrum_simple: need n >= 8");
    GODOT_REQUIRE(W && WW && S, GODOT_E_INVALID, "godot_power_spectrum_simple: null series");
    GODOT_REQUIRE(workspace && workspace_bytes >= godot_power_spectrum_workspace_bytes(n), GODOT_E_WORKSPACE,
                  "godot_power_spectrum_simple: workspace smaller than godot_power_spectrum_workspace_bytes(n)");
    hipStream_t st = (hipStream_t)stream;
    if (!is_pow2(n)) {
        const int64_t nf = n / 2 + 1, nkk = n / 4 + 1;
        size_t fsz = align256(sizeof(cplx) * (size_t)nf);
        cplx *F = (cplx *)workspace;
        PackArgs ser[2] = {PackArgs{W, S, n, 0}, PackArgs{WW, nullptr, n, 0}};
        bluestein_rfft(ser, 2, n, nf, F, (int64_t)(fsz / sizeof(cplx)), (char *)workspace + 5 * fsz, st);
        hipLaunchKernelGGL(ps_simple_epilogue_full_kernel, dim3((unsigned)ceil_div(nkk, 256)), dim3(256), 0, st,
                           (const cplx *)F, (int64_t)(fsz / sizeof(cplx)), nkk, dlogl, moments);
        GODOT_LAUNCH_CHECK("godot_power_spectrum_simple");
        return GODOT_OK;
    }
    const int64_t l = n, N1 = l / 2, N2 = l / 4, nk = l / 4 + 1;
    char *ws = (char *)workspace;
    size_t sz1 = align256(sizeof(cplx) * 2 * (size_t)N1), sz2 = align256(sizeof(cplx) * 3 * (size_t)N2);
    cplx *a1 = (cplx *)ws, *b1 = (cplx *)(ws + sz1);
    cplx *a2 = (cplx *)(ws + 2 * sz1), *b2 = (cplx *)(ws + 2 * sz1 + sz2);
    PackBatch pb2{};
    pb2.s[0] = PackArgs{WW, nullptr, n, l / 2};
    cplx *Z2 = fft_c2c(a2, b2, N2, 1, st, &pb2);
    PackBatch pb1{};
    pb1.s[0] = PackArgs{W, S, n, 0};
    cplx *Z1 = fft_c2c(a1, b1, N1, 1, st, &pb1);
    hipLaunchKernelGGL(ps_simple_epilogue_kernel, dim3((unsigned)ceil_div(nk, 256)), dim3(256), 0, st,
                       (const cplx *)Z1, N1, (const cplx *)Z2, N2, nk, dlogl, moments);
    GODOT_LAUNCH_CHECK("godot_power_spectrum_simple");
    return GODOT_OK;
}
