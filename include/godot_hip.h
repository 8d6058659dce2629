/*
 * godot_hip.h -- C ABI of libgodot_hip.so: MI355X (gfx950) kernels for the
 * weighted-photon likelihood / periodogram hot path of kerrm/godot.
 *
 * The reference has no FFI of its own: the boundary it exposes for this path
 * is the Python API of its core.py.  Each entry point below replaces the
 * arithmetic of the reference function cited next to it (file:line relative to
 * the reference checkout); the Python classes in godot_amd/core.py keep the
 * reference's names and signatures and forward to these functions via ctypes.
 *
 * Conventions
 *   - Every pointer is a DEVICE pointer unless its name ends in _host.
 *   - The caller allocates every input, output and workspace; the library
 *     never allocates, frees or keeps a pointer past the call.  (Its only
 *     device memory of its own is a 34 KB file-scope array per GPU, through
 *     which godot_cell_fit / godot_cell_logpdf let the workgroups of one very
 *     large cell exchange partial sums; launches that use it are ordered by
 *     the library across streams and cannot be captured into a HIP graph.)
 *   - All work is enqueued on `stream` (a hipStream_t passed as void*; NULL =
 *     the default stream) and is asynchronous with respect to the host.
 *   - Return value: 0 on success, a negative GODOT_E_* code on failure;
 *     godot_last_error() returns a thread-local message for the last failure.
 *   - Photons of cell c are we[lo[c] .. hi[c]-1].  For a CSR layout pass
 *     lo = offsets and hi = offsets + 1.  `nph` is the number of elements of
 *     `we`; a cell whose range is not inside [0, nph] is not read: its outputs
 *     are NaN and it gets GODOT_F_FAILED.
 *   - Floating point is IEEE binary64 throughout; indices are int64_t.
 */
#ifndef GODOT_HIP_H
#define GODOT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GODOT_OK            0
#define GODOT_E_INVALID    -1   /* bad argument (null pointer, negative size ...) */
#define GODOT_E_WORKSPACE  -2   /* workspace too small */
#define GODOT_E_LAUNCH     -3   /* HIP reported a launch/runtime error */
#define GODOT_E_UNSUPPORTED -4  /* size outside what the kernels accept */

/* per-cell status bits written to `flags` outputs */
#define GODOT_F_AT_ZERO     0x01  /* alpha-hat is exactly 0 (boundary maximum)  */
#define GODOT_F_DEPTH_MASK  0x06  /* restart depth (0..3) << 1                 */
#define GODOT_F_BRENT       0x08  /* bracketing root finder was used           */
#define GODOT_F_FAILED      0x10  /* no bracket / no convergence: output NaN   */
#define GODOT_F_RECENTRED   0x20  /* profile grid was re-centred at least once */
#define GODOT_F_RETRIED     0x40  /* a very large cell spread over several workgroups was fitted
                                     again by one: the workgroups did not all become resident
                                     together (shared or busy GPU); the result is as good */
#define GODOT_F_BAD_RANGE   0x80  /* set with GODOT_F_FAILED: the cell's photon range [lo, hi)
                                     is not inside the photon arrays (a caller's error, not
                                     a property of the data)                               */

typedef void *godot_stream_t;

const char *godot_last_error(void);
int godot_version(void);

/* ---- cell likelihood: CellLogLikelihood, core.py:329-1044 ----------------- */

/* Maximum-likelihood flux multiplier and TS of every cell with the background
 * level fixed.  Replaces CellLogLikelihood.get_max(guess, beta) (core.py:632-691,
 * incl. its restart rule and the get_max_numerical fallback, core.py:693-739)
 * followed by get_ts(aopt) (core.py:982-994).
 *   S[c], B[c]   expected source / background counts of the cell (Cell.exp and
 *                Cell.exp / Cell.SonB, core.py:344-345)
 *   guess, beta  per-cell start values, or NULL to use guess0 / beta0 for all
 *   swap         non-zero: exchange source and background (core.py:346-348)
 *   alpha, ts    outputs [ncell]; ts may be NULL
 *   flags        output [ncell] of GODOT_F_* bits; may be NULL               */
int godot_cell_fit(const double *we, int64_t nph,
        const int64_t *lo, const int64_t *hi,
                   const double *S, const double *B, int64_t ncell,
                   const double *guess, double guess0,
                   const double *beta, double beta0, int swap,
                   double *alpha, double *ts, int32_t *flags,
                   godot_stream_t stream);

/* godot_cell_fit told how many photons lie inside the cells (the sum of hi - lo).  The
 * kernels choose how many lanes and registers a cell gets from the mean number of photons
 * per cell, which godot_cell_fit takes to be nph / ncell; a subset of the cells handed over
 * with the whole weight array (a slice of a cell list, one rank's share) would get a shape
 * for cells many times their size.  Same results: the choice changes speed only.  The
 * reference fits cell by cell (core.py:1087-1095) and has no counterpart of the choice.   */
int godot_cell_fit_shaped(const double *we, int64_t nph,
        const int64_t *lo, const int64_t *hi,
                          const double *S, const double *B, int64_t ncell,
                          int64_t photons_in_cells,
                          const double *guess, double guess0,
                          const double *beta, double beta0, int swap,
                          double *alpha, double *ts, int32_t *flags,
                          godot_stream_t stream);

/* logL(alpha_c) of every cell: CellLogLikelihood.log_likelihood
 * (core.py:353-364).  alpha is per cell.                                      */
int godot_cell_loglike(const double *we, int64_t nph,
        const int64_t *lo, const int64_t *hi,
                       const double *S, const double *B, int64_t ncell,
                       const double *alpha, int swap, double *logl,
                       godot_stream_t stream);

/* Adaptive-support log-likelihood profile of every cell: get_logpdf
 * (core.py:851-932) as driven by CellsLogLikelihood.__init__
 * (core.py:1087-1095: npt = 200, dlogl = 30).
 *   aopt    per-cell maximum to centre on, or NULL to run get_max() first
 *   dom,cod outputs [ncell][npt], row-major                                   */
int godot_cell_logpdf(const double *we, int64_t nph,
        const int64_t *lo, const int64_t *hi,
                      const double *S, const double *B, int64_t ncell,
                      const double *aopt, double dlogl, int npt,
                      int include_zero, int swap,
                      double *dom, double *cod, int32_t *flags,
                      godot_stream_t stream);
/* godot_cell_logpdf told how many photons lie inside the cells; see
 * godot_cell_fit_shaped.                                                        */
int godot_cell_logpdf_shaped(const double *we, int64_t nph,
        const int64_t *lo, const int64_t *hi,
                             const double *S, const double *B, int64_t ncell,
                             int64_t photons_in_cells,
                             const double *aopt, double dlogl, int npt,
                             int include_zero, int swap, double *dom,
                             double *cod, int32_t *flags,
                             godot_stream_t stream);


/* Post-processing of the cached profiles: CellsLogLikelihood.get_flux
 * (core.py:1229, 1247-1256).  aguess = dom[argmax cod]; xlo/xhi = quantiles
 * conf_lo/conf_hi of the normalised pdf (trapezoid cdf + linear interpolation). */
int godot_profile_quantiles(const double *dom, const double *cod,
                            int64_t ncell, int npt,
                            double conf_lo, double conf_hi,
                            double *aguess, double *xlo, double *xhi,
                            godot_stream_t stream);

/* godot_profile_quantiles for rows whose abscissae are numpy.linspace(dom[c][0],
 * dom[c][npt-1], npt) to the bit -- what godot_cell_logpdf and
 * godot_cell_logpdf_profile write (get_logpdf, core.py:851-932, takes its grid
 * from numpy.linspace): only the two ends of a row of dom are read and the
 * abscissae are formed as those kernels form them, so the call moves half the
 * bytes (the light curve of 4 10^5 cells: 0.71 GB instead of 1.42 GB).  Same
 * results as godot_profile_quantiles on such rows, bit for bit; on other rows
 * they are the quantiles of the linspace row, not of dom.                     */
int godot_profile_quantiles_linspace(const double *dom, const double *cod, int64_t ncell, int npt,
                                     double conf_lo, double conf_hi, double *aguess,
                                     double *xlo, double *xhi, godot_stream_t stream);

/* Assemble light-curve rows 1..5 of get_raw_lightcurve (core.py:1329-1334):
 * out[c] = {terr_days, alpha|UL, err_lo|0, err_hi|-1, TS}; column 0 (MJD of the
 * cell mid-time) is formed on the host in extended precision as the reference
 * does (core.py:24-26).  NaN quantiles (a profile the reference filled with NaN
 * without stopping) take their places in the row as they do there; a NaN alpha
 * (a cell the reference raises on) gives a row of NaN.                        */
int godot_lightcurve_rows(const double *tstart, const double *tstop,
                          const double *alpha, const double *ts,
                          const double *xlo, const double *xhi,
                          const double *S, int64_t ncell, double tsmin,
                          double *rows /* [ncell][5] */,
                          godot_stream_t stream);

/* Mission elapsed time (s) to MJD: met2mjd (core.py:24-26), bit for bit what the
 * reference stores: the product with 1/86400 and the sum with mjdref each rounded
 * to the 64-bit significand of numpy's longdouble on x86, the result then to
 * double.  Integer arithmetic on the device.  Defined for positive normal met
 * and mjdref; other inputs (zero, negative, subnormal, non-finite, or operands
 * more than 60 binary orders apart) give NaN for the caller to treat.           */
int godot_met2mjd(const double *met, int64_t n, double mjdref, double *mjd,
                  godot_stream_t stream);

/* Background level maximising logL at fixed alpha: get_beta_max
 * (core.py:741-823) and the profile likelihood built on it
 * (log_profile_likelihood, core.py:391-406).  alpha per cell or alpha0.
 * logl_profile may be NULL.  A cell where the reference would raise
 * ValueError (core.py:818) gets NaN and GODOT_F_FAILED.                       */
int godot_cell_beta_max(const double *we, int64_t nph,
        const int64_t *lo, const int64_t *hi,
                        const double *S, const double *B, int64_t ncell,
                        const double *alpha, double alpha0,
                        const double *guess, double guess0, int swap,
                        double *beta, double *logl_profile, int32_t *flags,
                        godot_stream_t stream);

/* Joint maximum of logL(alpha, beta) over alpha >= 0, beta >= 0: the result of
 * get_max(profile_background=True) (core.py:552-630).  The reference uses
 * scipy's truncated-Newton search with ftol = 1e-3; this is a projected Newton
 * iteration to convergence, so (alpha, beta) agree with the reference to that
 * optimiser's tolerance and logL is never lower than the reference's by more
 * than its own acceptance margin (2e-3, core.py:617-620).
 *   guess_a/guess_b  per-cell starts or NULL for the scalars
 *   logl             log_full_likelihood at the maximum (core.py:427-438), may be NULL */
int godot_cell_fit_joint(const double *we, int64_t nph,
                         const int64_t *lo, const int64_t *hi,
                         const double *S, const double *B, int64_t ncell,
                         const double *guess_a, double guess_a0,
                         const double *guess_b, double guess_b0, int swap,
                         double *alpha, double *beta, double *logl,
                         int32_t *flags, godot_stream_t stream);

/* Profile-likelihood grid with the background level free: _get_logpdf_profile
 * (core.py:934-974): support limits by two Brent root finds on
 * log_profile_likelihood - llmax + dlogl, then npt evaluations of
 * log_profile_likelihood (each a get_beta_max).  aopt: per-cell flux to centre
 * on (beta follows from get_beta_max) or NULL to start from the joint maximum. */
int godot_cell_logpdf_profile(const double *we, int64_t nph,
                              const int64_t *lo, const int64_t *hi,
                              const double *S, const double *B, int64_t ncell,
                              const double *aopt, double dlogl, int npt, int swap,
                              double *dom, double *cod, int32_t *flags,
                              godot_stream_t stream);

/* ---- binning: Data.get_exposure / Data.get_cells, core.py:1792-1843, 2063-2206 */

/* Running sums of the exposure table (np.cumsum at core.py:1826-1832) for the
 * four series exposure, (S/E) exposure, (B/E) exposure and dead time
 * (TSTOP-TSTART)-LIVETIME, added left to right exactly as numpy does so that
 * differences of the sums (and therefore the zero-exposure trimming mask) are
 * bit-identical to the reference's (long tables are summed in parallel with
 * the sequential result's own roundings).  `table` (device,
 * godot_exposure_table_bytes bytes: the sums, then the six numbers a query reads
 * of its row side by side -- 80 bytes a row in all -- and scratch of the build)
 * depends only on the exposure table and the two scale factors: build it
 * once, reuse it for every godot_exposure_at call.                            */
size_t godot_exposure_table_bytes(int64_t n_ft2);
int godot_exposure_table(const double *tstart, const double *tstop,
                         const double *exposure, const double *livetime,
                         int64_t n_ft2, double s_over_e, double b_over_e,
                         void *table, size_t table_bytes, godot_stream_t stream);

/* Exposure, source-scaled exposure, background-scaled exposure and dead time
 * at the query times (cumulative != 0) or between consecutive query times
 * (cumulative == 0: outputs have n_t - 1 entries).  Replaces
 * Data.get_exposure (core.py:1806-1841).  out_dead may be NULL.  With nothing
 * to write (n_t == 0, or n_t == 1 and cumulative == 0) the call succeeds and
 * looks at no query, output or workspace pointer.
 * workspace: godot_exposure_workspace_bytes(n_t).                             */
size_t godot_exposure_workspace_bytes(int64_t n_t);
int godot_exposure_at(const double *tstart, const double *tstop,
                      const double *exposure, const double *livetime,
                      int64_t n_ft2, double s_over_e, double b_over_e,
                      const void *table,
                      const double *times, int64_t n_t, int cumulative,
                      double *out_exp, double *out_sexp, double *out_bexp,
                      double *out_dead, void *workspace, size_t workspace_bytes,
                      godot_stream_t stream);

/* The same two calls when the source- and background-scaled exposures of the
 * table rows have been rescaled by the caller (Data.apply_scalers,
 * core.py:2282-2306: sexposure *= src_scaler(TSTART, TSTOP) etc.): the per-row
 * arrays replace the (S/E), (B/E) multiples of `exposure`.                     */
int godot_exposure_table_rows(const double *tstart, const double *tstop,
                              const double *exposure, const double *sexposure,
                              const double *bexposure, const double *livetime,
                              int64_t n_ft2, void *table, size_t table_bytes,
                              godot_stream_t stream);
int godot_exposure_at_rows(const double *tstart, const double *tstop,
                           const double *exposure, const double *sexposure,
                           const double *bexposure, const double *livetime,
                           int64_t n_ft2, const void *table,
                           const double *times, int64_t n_t, int cumulative,
                           double *out_exp, double *out_sexp, double *out_bexp,
                           double *out_dead, void *workspace, size_t workspace_bytes,
                           godot_stream_t stream);

/* numpy.searchsorted(sorted, values, side='left' | 'right') on the device: the
 * photon index ranges of Data.get_cells_from_time_intervals (core.py:2217-2218)
 * and the insertion points of the randomised variants (core.py:2148-2150, 2240). */
int godot_search_sorted(const double *sorted, int64_t n, const double *values,
                        int64_t nv, int right, int64_t *out, godot_stream_t stream);

/* out[k] = scales[searchsorted(stops, t[k])]: PiecewiseScaler called with event
 * times (scalers.py:46-51; stops/scales are the scaler's padded arrays).        */
int godot_piecewise_lookup(const double *stops, const double *scales, int64_t nseg,
                           const double *t, int64_t n, double *out,
                           godot_stream_t stream);

/* out[k] = mean over t0[k] .. t1[k] of the piecewise-constant function:
 * PiecewiseScaler called with the starts and stops of the spacecraft rows
 * (scalers.py:52-85; apply_scalers core.py:2297-2300 calls it on every row of
 * the exposure table).  starts/stops/scales/prod/cprod [nseg] are the scaler's
 * padded arrays (scalers.py:21-29).                                            */
int godot_piecewise_average(const double *starts, const double *stops, const double *scales,
                            const double *prod, const double *cprod, int64_t nseg,
                            const double *t0, const double *t1, int64_t n, double *out,
                            godot_stream_t stream);

/* out = a w / (a w + b (1 - w)) (apply_scalers, core.py:2304); a, b per photon
 * (device arrays) or NULL for the scalars a0, b0.                              */
int godot_rescale_weights(const double *w, const double *a, double a0,
                          const double *b, double b0, int64_t n, double *out,
                          godot_stream_t stream);

/* Photon range of every cell for time-sorted photons: the integer result of
 * searchsorted(ti,[starts[0],stops[-1]]) + searchsorted(stops,times) +
 * bincount (core.py:2097-2099, 2159-2161): photon i belongs to cell c iff
 * stops[c-1] < ti[i] <= stops[c], restricted to start0 <= ti[i] < stops[-1].
 * lo/hi index the ORIGINAL ti/we arrays; counts[c] = hi[c] - lo[c].           */
int godot_bin_photons(const double *ti, int64_t nph, double start0,
                      const double *stops, int64_t ncell,
                      int64_t *lo, int64_t *hi, int64_t *counts,
                      godot_stream_t stream);

/* Uniform cells of MANY sources in two launches: for every source what
 * Data.get_cells(tstart, tstop, tcell, use_barycenter=False,
 * trim_zero_exposure=False) computes for it (core.py:2064-2108) -- the edges
 * tstart + arange(ncell + 1) * tcell, the exposures between them
 * (Data.get_exposure(edges, cumulative=False), core.py:1806-1841) and the
 * photon range of every cell (core.py:2097-2099) -- written side by side into
 * arrays that hold the cells of all sources.  A loop over a thousand sky
 * positions (the reference calls get_cells once per source) is bound by launch
 * overhead when every source costs a handful of launches; here the launch count
 * does not depend on the number of sources.
 * `sources` is a HOST array of nsrc descriptors (copied to the workspace by the
 * call; reusable as soon as it returns); the pointers inside are device
 * pointers.  Outputs (device, total_cells entries; starts, stops, out_dead and
 * source may be NULL): cell c of source k is entry cell_offset + c; lo/hi are
 * indices into the source's own ti plus photon_offset (so that they index the
 * concatenation of all photon lists when photon_offset is the number of photons
 * before the source); source[...] = k.  The cells of consecutive descriptors
 * must not overlap (cell_offset ascending).  Same bits as the per-source calls
 * godot_exposure_at + godot_bin_photons on the same edges.
 * The descriptor table is copied to the device by the call: from pageable memory that copy
 * waits for the stream's earlier work, from page-locked memory (hipHostMalloc) it is queued
 * like the launches.  The grid covers the longest source, so one source with many more cells
 * than the others launches mostly empty workgroups for them.
 * workspace: godot_bin_sources_workspace_bytes(nsrc, total_cells).              */
typedef struct godot_source_cells {
    const double *tstart, *tstop, *exposure, *livetime; /* spacecraft rows [n_ft2]          */
    const double *sexposure, *bexposure; /* rescaled per-row exposures (both) or NULL (both) */
    const void *table;       /* godot_exposure_table (or _rows) of those rows                */
    const double *ti;        /* time-sorted photon times [nph]                               */
    int64_t n_ft2, nph;
    double s_over_e, b_over_e; /* as for godot_exposure_at; unused with sexposure/bexposure  */
    double t0, tcell;        /* edge i = i * tcell + t0, i = 0 .. ncell                      */
    int64_t ncell;
    int64_t cell_offset, photon_offset;
    int64_t checkpoints;     /* nonzero: `table` is what godot_exposure_checkpoints wrote     */
} godot_source_cells;
size_t godot_bin_sources_workspace_bytes(int64_t nsrc, int64_t total_cells);
int godot_bin_sources(const godot_source_cells *sources, int64_t nsrc, int64_t total_cells,
                      double *starts, double *stops, double *out_exp, double *out_sexp,
                      double *out_bexp, double *out_dead, int64_t *lo, int64_t *hi,
                      int64_t *source, void *workspace, size_t workspace_bytes,
                      godot_stream_t stream);

/* Running sums of MANY exposure tables in one call, kept for every sixteenth row
 * only: record j of a table holds the four sums godot_exposure_table keeps for
 * every row (exposure, its S/E and B/E multiples, dead time: core.py:1821-1831)
 * through row 16 j + 15 (the last record through the last row), with numpy.cumsum's
 * roundings.  A thousand sky positions over one spacecraft file need a thousand
 * tables; at 80 bytes a row the full tables do not fit beside the photons, at
 * 2 bytes a row these do.  godot_bin_sources takes them (`checkpoints` nonzero in
 * the descriptor) and adds the at most fifteen rows between a record and the row
 * a cell edge falls in, from the left, as the full table was summed: same bits.
 * `tables` is a HOST array of ntab descriptors with device pointers inside;
 * `checkpoints` of each: godot_exposure_checkpoints_bytes(n_ft2) bytes, written by
 * the call.  sexposure/bexposure as in godot_source_cells.  The launch count does
 * not depend on ntab.  The descriptors are copied to the device by the call (from
 * pageable memory that copy waits for the stream's earlier work).             */
typedef struct godot_exposure_rows {
    const double *tstart, *tstop, *exposure, *livetime; /* spacecraft rows [n_ft2]          */
    const double *sexposure, *bexposure; /* rescaled per-row exposures (both) or NULL (both) */
    int64_t n_ft2;
    double s_over_e, b_over_e;
    void *checkpoints;
} godot_exposure_rows;
size_t godot_exposure_checkpoints_bytes(int64_t n_ft2);
size_t godot_exposure_checkpoints_workspace_bytes(const godot_exposure_rows *tables, int64_t ntab);
int godot_exposure_checkpoints(const godot_exposure_rows *tables, int64_t ntab, void *workspace,
                               size_t workspace_bytes, godot_stream_t stream);

/* The per-photon sums behind the reference's small per-cell functions:
 * log_full_likelihood (core.py:427-438), fmin_tnc_func (440-460), f1/f2/f3
 * (479-507), nr (509-524), halley (526-542), f1_profile (483-497) and
 * log_profile_likelihood_approx (366-389).  With flux factors alpha, beta per
 * cell (arrays, or NULL and the scalar) and t_i = 1 + (beta-1)(1-w_i) +
 * (alpha-1) w_i, sums[cell][0..7] =
 *   sum log t, sum w/t, sum (1-w)/t, sum (w/t)^2, sum (w/t)^3,
 *   sum ((1-w)/t)^2, sum (w/t)((1-w)/t), sum (w/t)((1-w)/t)^2.
 * swap != 0: on the complemented weights.                                      */
int godot_cell_sums(const double *we, int64_t nph,
                    const int64_t *lo, const int64_t *hi, int64_t ncell,
                    const double *alpha, double alpha0,
                    const double *beta, double beta0, int swap,
                    double *sums, godot_stream_t stream);

/* Per-cell moments sum(w), sum(w^2) (core.py:2178-2183).                      */
int godot_cell_moments(const double *we, int64_t nph,
        const int64_t *lo, const int64_t *hi,
                       int64_t ncell, double *wsum, double *w2sum,
                       godot_stream_t stream);

/* The six series power_spectrum_fft transforms (core.py:2654-2661), from the
 * bins of a time series in one launch: W[c], WW[c] = the sums of the weights and
 * of their squares over photons lo[c] .. hi[c]-1 (godot_cell_moments); bins
 * with exp[c] > 0 false are emptied as CellTimeSeries empties them
 * (minimum_exposure = 0, core.py:241-244, 279-285): W, WW and the count become
 * 0 and sexp[c], bexp[c] are set to 0 IN PLACE; Wb = count - W and
 * WbWb = count - 2 W + WW, rounded operation by operation as numpy rounds them.
 * What get_cells(time_series_only=True) + the first lines of power_spectrum_fft
 * do in a dozen passes over the bins.                                          */
int godot_cell_series(const double *we, int64_t nph, const int64_t *lo, const int64_t *hi,
                      const double *exp, int64_t ncell, double *sexp, double *bexp,
                      double *W, double *WW, double *Wb, double *WbWb, godot_stream_t stream);

/* Indices of the cells with exp > threshold in ascending order (the
 * trim_zero_exposure mask, core.py:2110-2122).  n_keep is a device int64.
 * workspace: godot_select_workspace_bytes(ncell).                             */
size_t godot_select_workspace_bytes(int64_t ncell);
int godot_select_cells(const double *exp, int64_t ncell, double threshold,
                       int64_t *keep_idx, int64_t *n_keep,
                       void *workspace, size_t workspace_bytes,
                       godot_stream_t stream);
/* godot_select_cells without a workspace.  No counterpart in the reference: it
 * is for callers that cannot allocate device memory (code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc).  The tile
 * counts of the scan live in a 64 KB file-scope device array that every GPU the
 * library runs on holds until the process ends; ncell is limited to
 * godot_select_cells_static_max() = 16 777 216 (GODOT_E_UNSUPPORTED beyond).
 * Same outputs as godot_select_cells, bit for bit.  The call allocates and
 * frees nothing.  All calls on one GPU share that array: calls that overlap in
 * time on one GPU (two streams, two threads) race; calls on one stream do not.  */
int64_t godot_select_cells_static_max(void);
int godot_select_cells_static(const double *exp, int64_t ncell, double threshold,
                              int64_t *keep_idx, int64_t *n_keep,
                              godot_stream_t stream);
int godot_gather_f64(const double *src, const int64_t *idx, int64_t n,
                     double *dst, godot_stream_t stream);
int godot_gather_i64(const int64_t *src, const int64_t *idx, int64_t n,
                     int64_t *dst, godot_stream_t stream);
/* The trimming of every per-cell array by the one mask (core.py:2113-2122:
 * starts[mask], stops[mask], exp[mask], ...) in one launch: dst[a][k] =
 * src[a][idx[k]] for ncol <= 16 arrays of 8-byte elements (double or int64).
 * src and dst are HOST arrays of ncol device pointers.                        */
int godot_gather_columns(const void *const *src, void *const *dst, int ncol,
                         const int64_t *idx, int64_t n, godot_stream_t stream);

/* ---- power spectra: core.py:2598-2782 -------------------------------------- */

/* Real-to-complex forward DFT of x[0..n-1] zero-padded (or cut) to length
 * l >= 2: out[k] = sum_j x[j] exp(-2 pi i j k / l), k = 0..l/2, stored as
 * interleaved (re, im).  Powers of two run the Stockham transform directly,
 * other lengths go through a chirp-z (Bluestein) convolution on top of it.
 * Stands in for numpy.fft.rfft at core.py:2683-2701.
 * workspace: godot_rfft_workspace_bytes(l).                                   */
size_t godot_rfft_workspace_bytes(int64_t l);
int godot_rfft_f64(const double *x, int64_t n, int64_t l, double *out,
                   void *workspace, size_t workspace_bytes,
                   godot_stream_t stream);

/* power_spectrum_fft (core.py:2635-2741) from the six per-bin series; the
 * scalar options (tweak_exp, exposure_correction, exp_only, maxN) are applied
 * by the caller to the series before the call.  Outputs have l/4+1 entries.
 * l may be any length >= 8 (the default padded length is a power of two; the
 * reference's no_zero_pad=True gives l = n).
 * If amps != NULL it receives the 8 amplitude arrays of get_amps=True
 * (alpha_cos, alpha_sin, beta_cos, beta_sin, alpha_cos0, alpha_sin0,
 * beta_cos0, beta_sin0), [8][l/4+1].
 * workspace: godot_power_spectrum_workspace_bytes(l).                         */
size_t godot_power_spectrum_workspace_bytes(int64_t l);
int godot_power_spectrum_fft(const double *W, const double *WW, const double *S,
                             const double *Wb, const double *WbWb,
                             const double *B, int64_t n, int64_t l,
                             double *p0, double *p1, double *pb, double *amps,
                             void *workspace, size_t workspace_bytes,
                             godot_stream_t stream);

/* power_spectrum_fft (core.py:2598-2741) dealt over several GPUs: the transform of
 * godot_power_spectrum_fft for l = 2^23, 2^24, 2^25 is "columns, then rows" of the
 * series laid out as 16 384 rows (l = 16 384 B points: first the 16 384-point
 * transforms of the B columns, then, after a twiddle, the B-point transforms of the
 * rows, of which rows 0 .. 8 192 are kept: the series are real).  Part `part` of
 * `nparts` GPUs takes its nparts-th of the columns of the five series
 * (godot_power_spectrum_columns: every GPU holds the whole series), the GPUs
 * exchange what they made so that each holds whole rows m_first <= m < m_end of all
 * five (an all-to-all of 28 l bytes in total over all GPUs, the caller's job: RCCL),
 * and godot_power_spectrum_rows transforms those rows and writes the estimators of
 * the bins they hold (bins m + 16 384 p and 16 384 - m + 16 384 p; the other entries
 * of p0, p1, pb are not touched).
 * v_part: godot_power_spectrum_part_bytes(l, nparts) bytes: five arrays of complex
 * numbers back to back, [8 193][B / (2 nparts)] for WW, WbWb and W - WW, then
 * [8 193][B / nparts] for W - S and Wb - B.  v_rows: the same five arrays with whole
 * rows: [m_end - m_first][B / 2] three times, [m_end - m_first][B] twice.  dc: the
 * sums of WW, WbWb and W - WW over the series (device, three doubles).  m_first a
 * multiple of 16.  godot_power_spectrum_parts_supported says for which (n, l,
 * nparts) this is served: the lengths above, n <= l / 2, B / (2 nparts) a multiple
 * of 32.  No counterpart in the reference, which transforms on one CPU.          */
int godot_power_spectrum_parts_supported(int64_t n, int64_t l, int nparts);
int64_t godot_power_spectrum_part_rows(void);
size_t godot_power_spectrum_part_bytes(int64_t l, int nparts);
size_t godot_power_spectrum_part_workspace_bytes(int64_t l, int nparts);
int godot_power_spectrum_columns(const double *W, const double *WW, const double *S, const double *Wb,
                                 const double *WbWb, const double *B, int64_t n, int64_t l, int part,
                                 int nparts, void *v_part, void *workspace, size_t workspace_bytes,
                                 godot_stream_t stream);
int godot_power_spectrum_rows(const void *v_rows, const double *dc, int64_t l, int64_t m_first,
                              int64_t m_end, double *p0, double *p1, double *pb, double *amps,
                              godot_stream_t stream);

/* The estimators of power_spectrum_fft (core.py:2707-2741) from the five
 * transforms handed over one by one, each an array of nk (re, im) pairs cut to
 * what the estimators read: f1 = rfft(W - S)[:nk], f2 = rfft(Wb - B)[:nk],
 * g3 = rfft(WW)[0 : 2 nk : 2], g4 = rfft(WbWb)[0 : 2 nk : 2],
 * g5 = rfft(W - WW)[0 : 2 nk : 2] (core.py:2683-2704), nk = l/4 + 1.  This is
 * the form in which transforms computed on different GPUs arrive after their
 * all-gather; outputs as godot_power_spectrum_fft.                            */
int godot_power_spectrum_from_parts(const double *f1, const double *f2, const double *g3,
                                    const double *g4, const double *g5, int64_t nk,
                                    double *p0, double *p1, double *pb, double *amps,
                                    godot_stream_t stream);

/* power_spectrum_simple (core.py:2758-2782): l = n, any length >= 8.
 * moments (may be NULL) receives (cos, sin, cc, ss), [4][l/4+1].              */
int godot_power_spectrum_simple(const double *W, const double *WW,
                                const double *S, int64_t n,
                                double *dlogl, double *moments,
                                void *workspace, size_t workspace_bytes,
                                godot_stream_t stream);

/* The frequencies of the spectra above: np.fft.rfftfreq(l)[:nk] / tsamp
 * (power_spectrum_fft, core.py:2677-2679: divide = 1) or * (1 / tsamp)
 * (power_spectrum_simple, core.py:2765-2766: divide = 0, scale = 1 / tsamp),
 * element for element numpy's roundings: freqs[i] = (i * step) / scale or
 * (i * step) * scale with step = 1 / l.  Formed on the device because the host
 * takes longer over these 8.4e6 products and quotients (l = 2^25) than the
 * device over the five transforms.                                            */
int godot_spectrum_frequencies(int64_t nk, double step, double scale, int divide,
                               double *freqs, godot_stream_t stream);

/* The largest entry of every row of a [nrow][ncol] array from column `first`
 * on, and where it stands (counted from `first`): the peak of a spectrum and
 * its bin, what a periodicity search keeps of power_spectrum_fft's arrays
 * (the reference's callers take numpy's max / argmax of them, e.g.
 * paper_plots.py:396-420).  The first of equal entries, and a NaN before any
 * number, as numpy's argmax has it.  workspace: godot_row_peaks_workspace_bytes. */
size_t godot_row_peaks_workspace_bytes(int64_t nrow, int64_t ncol);
int godot_row_peaks(const double *a, int64_t nrow, int64_t ncol, int64_t first,
                    double *values, int64_t *indices,
                    void *workspace, size_t workspace_bytes, godot_stream_t stream);

/* ---- direct trig sums: core.py:2426-2596 ----------------------------------- */

/* power_spectrum_dft (core.py:2540-2596).  Every photon of a cell shares the
 * cell's phase, so the photon sums collapse to the per-cell moments
 * w1 = sum(w), w2 = sum(w^2) (godot_cell_moments); for unweighted=True pass the
 * counts as doubles for both and exp_scale = nph / sum(w) (core.py:2548-2551).
 * The per-cell phase uses the reference's 1000-entry cos/sin table with the
 * same truncation.  power has nfreq entries.                                  */
int godot_trigsum_cells(const double *w1, const double *w2, const double *exp,
                        const double *tmid, int64_t ncell,
                        const double *freqs, int64_t nfreq, double exp_scale,
                        double *power, godot_stream_t stream);

/* fit_harmonics (core.py:2426-2488) at nfreq frequencies with exact cos/sin:
 * out[f] = {camp, samp, ts}.  tmean = mean of tmid (core.py:2438).           */
int godot_fit_harmonics(const double *w1, const double *w2, const double *exp,
                        const double *tmid, int64_t ncell, double tmean,
                        const double *freqs, int64_t nfreq,
                        double *out /* [nfreq][3] */, godot_stream_t stream);

/* compute_ls (core.py:2490-2531): power[f] = (sum cos*flux)^2 + (sum sin*flux)^2
 * with the table phases of power_spectrum_dft; flux is the caller's normalised
 * per-cell series (core.py:2493-2501), the 0.5*sum(flux^2) scale is the caller's. */
int godot_ls_cells(const double *flux, const double *tmid, int64_t ncell,
                   const double *freqs, int64_t nfreq, double *power,
                   godot_stream_t stream);

/* One frequency of get_orbital_modulation (core.py:2845-2859): with
 * t = (starts+stops)/2, means = {mean(cos*y), mean(sin*y), mean(cos^2*z),
 * mean(sin^2*z)} (device, 4 doubles) and rvals += 2 (a cos + b sin).
 * y = (weights - sexp) / mean(sexp), z = weights2 (core.py:2847, 2856).       */
size_t godot_orbital_workspace_bytes(void);
int godot_orbital_modulation_step(const double *starts, const double *stops,
                                  const double *y, const double *z, int64_t n,
                                  double freq, double *means, double *rvals,
                                  void *workspace, size_t workspace_bytes,
                                  godot_stream_t stream);

/* Weighted Z^2_m on the uniform grid f_j = f0 + (j0 + j) df, j = 0..nf-1:
 *   Z2m = 2 / sum(w^2) * sum_{k=1..m} [(sum_i w_i cos 2 pi k f t_i)^2 +
 *                                      (sum_i w_i sin 2 pi k f t_i)^2].
 * No counterpart in the reference (it holds no Z^2_m / H-test routine; the
 * statistic is the photon-phase limit of fit_harmonics, core.py:2481).
 * f_j is the real number f0 + (j0 + j) df (f0, df the given doubles): phases
 * are formed from exactly reduced products, so accuracy does not degrade with
 * t f.  t is relative to any epoch the caller chooses.  1 <= m <= 20.
 *   z2   output [nf] (Z^2_m), may be NULL
 *   h    output [nf] (H = max_{k<=m} (Z^2_k - 4k + 4)), may be NULL
 * workspace: godot_z2m_workspace_bytes(nph, nf, m).                           */
size_t godot_z2m_workspace_bytes(int64_t nph, int64_t nf, int m);
int godot_z2m(const double *t, const double *w, int64_t nph,
              double f0, double df, int64_t j0, int64_t nf, int m,
              double *z2, double *h, void *workspace, size_t workspace_bytes,
              godot_stream_t stream);

/* The two halves of godot_z2m, for a photon list that is split over devices:
 * every device forms the trig sums of its own photons, the caller adds them
 * (and the sum(w^2) scalars) across devices, and godot_z2m_finish turns the
 * totals into the statistics.  No counterpart in the reference.
 *   sums       output [nf][m][2]: sum_i w_i cos(2 pi k f_j t_i), then the sine
 *   sumw2_out  output [1] (device): sum_i w_i^2 of this photon list
 * Workspace as for godot_z2m.                                                 */
int godot_z2m_sums(const double *t, const double *w, int64_t nph,
                   double f0, double df, int64_t j0, int64_t nf, int m,
                   double *sums, double *sumw2_out,
                   void *workspace, size_t workspace_bytes, godot_stream_t stream);
int godot_z2m_finish(const double *sums, const double *sumw2, int64_t nf, int m,
                     double *z2, double *h, godot_stream_t stream);

/* ---- Bayesian Blocks on the cached profiles ---------------------------------
 * CellsLogLikelihood.fitness (core.py:1108-1136) for every row at once: with
 * dom, cod the [ncell][npt] profiles of godot_cell_logpdf, row i (cells 0..i)
 * holds, for every block start j <= i,
 *   fit_val[i(i+1)/2 + j] = 2 max_k sum_{c=j..i} interp(grid_i[k]; dom[c], cod[c])
 *   fit_arg[i(i+1)/2 + j] = 2 grid_i[argmax]
 * where grid_i = linspace(dom[i][0], dom[i][npt-1], ngrid) and interp is
 * numpy.interp with -inf outside a cell's support (the reference uses ngrid =
 * 1000; ngrid <= 1024).  fitness(i0, i1) of the reference is entries i0..i1-1
 * of row i1-1.  Rows row_begin .. row_end-1 are computed (0, ncell for all:
 * the rows are independent, which is how the work is split over GPUs); the
 * outputs hold those rows packed one after the other, i.e. the entry above
 * sits at i(i+1)/2 - row_begin(row_begin+1)/2 + j.
 * workspace: godot_bb_workspace_bytes(ncell, npt).                            */
size_t godot_bb_workspace_bytes(int64_t ncell, int npt);
int godot_bb_fitness(const double *dom, const double *cod, int64_t ncell, int npt,
                     int ngrid, int64_t row_begin, int64_t row_end, void *workspace,
                     double *fit_val, double *fit_arg, godot_stream_t stream);

/* The dynamic programme of do_bb / _do_bb_cache (core.py:1146-1161, 1184-1194):
 *   best[0] = 0, best[i+1] = max_{r<=i} (fit_val[row i][r] - prior) + best[r],
 *   last[i] = the first r attaining it.
 * best: ncell + 1 doubles, last: ncell int64.  The walk back through `last`
 * (core.py:1164-1169) is left to the host.                                    */
int godot_bb_partition(const double *fit_val, int64_t ncell, double prior,
                       double *best, int64_t *last, godot_stream_t stream);

/* Test hook (no counterpart in the reference): polls that a workgroup of a cell
 * spread over several workgroups waits for the others before the cell goes to
 * the one-workgroup retry; 0 restores the default (2^22, seconds).  Synchronous. */
int godot_debug_spread_spin_limit(long polls);

#ifdef __cplusplus
}
#endif
#endif /* GODOT_HIP_H */
