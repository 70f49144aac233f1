// adh_log_f32.h - table-driven float64 log of a float32 argument (candidate selection: one log per smoothed cell)
#pragma once
#include "adh_log_table.h"

// log of a float32 x >= 1 in float64, error <= 1 ulp (the class of the library log; checked against an
// 80-bit log, tools/gen_log_table.py): x = 2^e * m, the top 7 mantissa bits pick c = 1 + i / 128, and
// log(m) = -log(1 / c) + log1p(m / c - 1) with |m / c - 1| < 2^-7 and a degree-9 series.  All terms are
// >= 0 (no cancellation) and entry 0 is exact, so values next to 1 keep their relative accuracy.  The
// smoothing needs one log per non-zero output cell, 127 000 per precursor: this is about half the
// instructions of the general routine.
// (`tab`: the table as 256 doubles - adh_log_tab itself, or a copy of it in LDS: a table look-up per log is a global load
// in the middle of a dependent chain, and the smoothing kernel of the ion-mobility selection takes one log per cell)
__device__ __forceinline__ double adh_log_f32(float x, const double *tab = &adh_log_tab[0][0]) {
    const uint32_t bits = __float_as_uint(x);
    const int e = (int)(bits >> 23) - 127;
    const uint32_t mant = bits & 0x007FFFFFu;
    const int i = (int)(mant >> 16);
    const double m = (double)__uint_as_float(mant | 0x3F800000u);
    const double t_inv = tab[2 * i], t_log = tab[2 * i + 1];
    const double r = fma(m, t_inv, -1.0);
    double q = 1.0 / 9.0;
    q = fma(q, r, -1.0 / 8.0);
    q = fma(q, r, 1.0 / 7.0);
    q = fma(q, r, -1.0 / 6.0);
    q = fma(q, r, 1.0 / 5.0);
    q = fma(q, r, -1.0 / 4.0);
    q = fma(q, r, 1.0 / 3.0);
    q = fma(q, r, -1.0 / 2.0);
    const double p = fma(r * r, q, r);
    const double ed = (double)e;
    return ed * 6.93147180369123816490e-01 + ((t_log + p) + ed * 1.90821492927058770002e-10);
}

// float32-rounded log of an argument outside [1, inf) (a smoothed value that is negative, NaN or overflowed: only a
// caller's kernel with negative factors gets here).  Out of line on purpose: the library routine is ~200
// instructions and a dozen live float64 constants, which inlined into every log site of the smoothing kernel
// cost its hot paths their registers.
__device__ __attribute__((noinline)) float adh_log_f32_rare(float x) { return (float)log((double)x); }
