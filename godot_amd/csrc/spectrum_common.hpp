// Helpers shared by the transform kernels of spectrum.hip and spectrum_rows.hip:
// complex arithmetic, exact-angle twiddles, in-register butterflies, the description of a
// real series built from the per-bin moments, and the estimators of one frequency bin.
#pragma once
#include "common.hpp"

namespace godot_spectrum {

struct cplx {
    double x, y;
};
__device__ __forceinline__ cplx operator+(cplx a, cplx b) { return cplx{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ cplx operator-(cplx a, cplx b) { return cplx{a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ cplx cmul(cplx a, cplx b)
{
    return cplx{fma(a.x, b.x, -(a.y * b.y)), fma(a.x, b.y, a.y * b.x)};
}

constexpr double TWO_PI_HI = 6.283185307179586232e+00;
constexpr double TWO_PI_LO = 2.449293598294706414e-16;

// exp(2 pi i x) for |x| <= 0.5
__device__ __forceinline__ cplx expi_cycles(double x)
{
    double q = rint(4.0 * x);
    double r = fma(q, -0.25, x);
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
    int qi = (int)q & 3;
    double s1 = (qi & 1) ? cs : sn;
    double c1 = (qi & 1) ? sn : cs;
    return cplx{((qi + 1) & 2) ? -c1 : c1, (qi & 2) ? -s1 : s1};
}

// exp(-2 pi i k / 16), k = 0..7
__device__ __forceinline__ cplx w16(int k)
{
    constexpr double C1 = 0.92387953251128673848, S1 = 0.38268343236508978178, H = 0.70710678118654752440;
    switch (k) {
    case 0: return cplx{1.0, 0.0};
    case 1: return cplx{C1, -S1};
    case 2: return cplx{H, -H};
    case 3: return cplx{S1, -C1};
    case 4: return cplx{0.0, -1.0};
    case 5: return cplx{-S1, -C1};
    case 6: return cplx{-H, -H};
    default: return cplx{-C1, -S1};
    }
}

// In-register forward DFT of R points (R = 2, 4, 8, 16), natural order in and
// out: recursive decimation in time, fully unrolled.
template <int R>
__device__ __forceinline__ void dft(cplx (&v)[R])
{
    if constexpr (R == 2) {
        cplx a = v[0], b = v[1];
        v[0] = a + b;
        v[1] = a - b;
    } else {
        cplx e[R / 2], o[R / 2];
#pragma unroll
        for (int k = 0; k < R / 2; ++k) {
            e[k] = v[2 * k];
            o[k] = v[2 * k + 1];
        }
        dft<R / 2>(e);
        dft<R / 2>(o);
#pragma unroll
        for (int k = 0; k < R / 2; ++k) {
            cplx t = (k == 0) ? o[k] : cmul(o[k], w16(k * (16 / R)));
            v[k] = e[k] + t;
            v[k + R / 2] = e[k] - t;
        }
    }
}

// ---- packing real series into the complex input ---------------------------

// z[n] = x[2n] + i x[2n+1] for a series x of n_in samples zero-padded to 2N.
// kind selects the series built on the fly from the per-bin moments.
enum Series { SER_PLAIN = 0, SER_DIFF = 1 };

struct PackArgs {
    const double *a;   // plain: x = a ; diff: x = a - b
    const double *b;
    int64_t n_in;      // samples available
    int64_t fold;      // 0, or the length at which the series is wrapped (x[n] + x[n + fold])
};

__device__ __forceinline__ double series_at(const PackArgs &p, int64_t i)
{
    if (i >= p.n_in) return 0.0;
    double v = p.a[i];
    if (p.b) v -= p.b[i];
    return v;
}

__device__ __forceinline__ double series_folded(const PackArgs &p, int64_t i)
{
    double v = series_at(p, i);
    if (p.fold > 0)
        for (int64_t q = i + p.fold; q < p.n_in; q += p.fold) v += series_at(p, q);
    return v;
}

struct PackBatch {
    PackArgs s[3];
};

// X[k] of the real transform of length 2N from zk = Z[k], zn = Z[N - k] of the packed
// transform and w = exp(-2 pi i k / (2N))
__device__ __forceinline__ cplx unpack_pair(cplx zk, cplx zn, cplx w)
{
    cplx e = cplx{0.5 * (zk.x + zn.x), 0.5 * (zk.y - zn.y)};     // (Z[k] + conj Z[N-k]) / 2
    cplx o = cplx{0.5 * (zk.y + zn.y), -0.5 * (zk.x - zn.x)};    // (Z[k] - conj Z[N-k]) / (2i)
    return e + cmul(w, o);
}

// The estimators of one frequency bin (core.py:2697-2741) from the cosine and sine
// sums of the two first-moment series, the zero-frequency values S2, B2, SB of the three
// second-moment series and the real parts g3, g4, g5 of their bins 2k.
__device__ __forceinline__ void ps_estimators(int64_t k, int64_t nk, double WmS_cos, double WmS_sin,
                                              double WbmB_cos, double WbmB_sin, double S2, double B2, double SB,
                                              double g3, double g4, double g5, double *p0, double *p1,
                                              double *pb, double *amps)
{
    double WW_cos = 0.5 * (S2 + g3), WW_sin = 0.5 * (S2 - g3);
    double WbWb_cos = 0.5 * (B2 + g4), WbWb_sin = 0.5 * (B2 - g4);
    double WWb_cos = 0.5 * (SB + g5), WWb_sin = 0.5 * (SB - g5);
    if (k == 0) {
        WW_sin = 1.;
        WbWb_sin = 1.;
    }
    double a_cos0 = WmS_cos / WW_cos;
    double a_sin0 = WmS_sin / WW_sin;
    double b_cos0 = WbmB_cos / WbWb_cos;
    double b_sin0 = WbmB_sin / WbWb_sin;
    double den_cos = 1. / (WW_cos * WbWb_cos - WWb_cos * WWb_cos);
    double den_sin = 1. / (WW_sin * WbWb_sin - WWb_sin * WWb_sin);
    if (k == 0) den_sin = 0.;
    double a_cos = (WbWb_cos * WmS_cos - WWb_cos * WbmB_cos) * den_cos;
    double a_sin = (WbWb_sin * WmS_sin - WWb_sin * WbmB_sin) * den_sin;
    double b_cos = (WW_cos * WbmB_cos - WWb_cos * WmS_cos) * den_cos;
    double b_sin = (WW_sin * WbmB_sin - WWb_sin * WmS_sin) * den_sin;
    if (amps) {
        amps[0 * nk + k] = a_cos;
        amps[1 * nk + k] = a_sin;
        amps[2 * nk + k] = b_cos;
        amps[3 * nk + k] = b_sin;
        amps[4 * nk + k] = a_cos0;
        amps[5 * nk + k] = a_sin0;
        amps[6 * nk + k] = b_cos0;
        amps[7 * nk + k] = b_sin0;
    }
    double dl_null = b_cos0 * WbmB_cos + b_sin0 * WbmB_sin;
    double dl_nobg = a_cos0 * WmS_cos + a_sin0 * WmS_sin;
    double dl = a_cos * WmS_cos + a_sin * WmS_sin + b_cos * WbmB_cos + b_sin * WbmB_sin;
    if (p0) p0[k] = dl_nobg;
    if (p1) p1[k] = dl - dl_null;
    if (pb) pb[k] = dl_null;
}

// spectrum_rows.hip: power_spectrum_fft in two trips through memory for long zero-padded series
bool rows_path_applies(int64_t n, int64_t l);
size_t rows_path_workspace_bytes(int64_t l);
void power_spectrum_rows(const double *W, const double *WW, const double *S, const double *Wb, const double *WbWb,
                         const double *B, int64_t n, int64_t l, double *p0, double *p1, double *pb, double *amps,
                         void *workspace, hipStream_t stream);
// the same transform in parts (columns of one part, then rows of a range), for several GPUs
bool parts_supported(int64_t n, int64_t l, int nparts);
int64_t part_rows();
size_t part_elements(int64_t l, int nparts);
size_t part_workspace_bytes(int64_t l, int nparts);
void power_spectrum_columns_part(const double *W, const double *WW, const double *S, const double *Wb,
                                 const double *WbWb, const double *B, int64_t n, int64_t l, int part, int nparts,
                                 void *v_part, void *workspace, hipStream_t stream);
void power_spectrum_rows_part(const void *v_rows, const double *dc, int64_t l, int64_t m_first, int64_t m_end,
                              double *p0, double *p1, double *pb, double *amps, hipStream_t stream);

}   // namespace godot_spectrum
