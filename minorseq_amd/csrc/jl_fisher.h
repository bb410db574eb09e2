// jl_fisher.h — FP64 one-sided Fisher's exact test for juliet's 2x2 table (doc/JULIET.md:38-42; SPEC §5).
//
// Both rows of the table sum to the coverage n: [[a, n-a], [c, n-c]] with a = observed codon count and
// c = expected count under the error model.  Then X ~ Hypergeometric(2n, a+c, n) and
//     P(X = x) = Binom(x; K, 1/2) * Binom(n-x; 2n-K, 1/2) / Binom(n; 2n, 1/2),  K = a + c,
// so the point mass is three Binomial(., 1/2) log-masses evaluated in saddle-point (Loader) form:
// ~1e-14 relative accuracy at 1e7 coverage, where a plain lgamma difference loses 7 digits.  The tail is
// at most c+1 terms of a ratio recurrence.  Written once for device and host (the host build exists only
// so tests can check the algorithm against the mpmath golden vectors without a GPU).
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define JL_FHD __host__ __device__ inline
#else
#define JL_FHD inline
#endif

JL_FHD double stirlerr(double n)
{
    // ln n! - ln( sqrt(2 pi n) (n/e)^n ), integer n >= 0
    constexpr double tab[16] = {0.0,
                            0.08106146679532726,
                            0.04134069595540929,
                            0.02767792568499834,
                            0.02079067210376509,
                            0.01664469118982119,
                            0.01387612882307075,
                            0.01189670994589177,
                            0.01041126526197209,
                            0.009255462182712733,
                            0.008330563433362871,
                            0.007573675487951841,
                            0.006942840107209530,
                            0.006408994188004207,
                            0.005951370112758848,
                            0.005554733551962801};
    if (n < 16.0) return tab[(int)n];
    const double S0 = 1.0 / 12.0, S1 = 1.0 / 360.0, S2 = 1.0 / 1260.0, S3 = 1.0 / 1680.0, S4 = 1.0 / 1188.0;
    const double r = 1.0 / n;  // one division; the series is in powers of 1/n^2
    const double rr = r * r;
    if (n > 500.0) return (S0 - S1 * rr) * r;
    if (n > 80.0) return (S0 - (S1 - S2 * rr) * rr) * r;
    if (n > 35.0) return (S0 - (S1 - (S2 - S3 * rr) * rr) * rr) * r;
    return (S0 - (S1 - (S2 - (S3 - S4 * rr) * rr) * rr) * rr) * r;
}

JL_FHD double bd0(double x, double np)
{
    // x ln(x/np) + np - x, stable for x near np
    if (fabs(x - np) < 0.1 * (x + np)) {
        double v = (x - np) / (x + np);
        double s = (x - np) * v;
        double ej = 2.0 * x * v;
        v = v * v;
        // |v| < 0.1: the series gains two digits per term, so it ends within ~10 terms
        for (int j = 1; j < 1000; ++j) {
            ej *= v;
            const double s1 = s + ej / (double)(2 * j + 1);
            if (s1 == s) return s1;
            s = s1;
        }
        return s;
    }
    return x * log(x / np) + np - x;
}

// ln [ C(n, x) / 2^n ]
JL_FHD double log_binom_half(double x, double n)
{
    const double LN2 = 0.6931471805599453094;
    const double TWO_PI = 6.283185307179586477;
    if (n == 0.0) return 0.0;
    if (x == 0.0 || x == n) return -n * LN2;
    const double h = 0.5 * n;
    const double lc = stirlerr(n) - stirlerr(x) - stirlerr(n - x) - bd0(x, h) - bd0(n - x, h);
    return lc + 0.5 * log(n / (TWO_PI * x * (n - x)));
}

// ln P(X = x) for X ~ Hypergeometric(2n, K, n): the three Binomial(., 1/2) masses folded into one expression
// (one log for the sqrt factors, the Binom(n; 2n) deviance terms are exactly zero)
JL_FHD double jl_log_pmf_equal_rows(double x, double K, double n)
{
    const double TWO_PI = 6.283185307179586477;
    const double M = 2.0 * n, L = M - K;
    const double c = K - x, nx = n - x, nc = n - c;
    if (x == 0.0 || c == 0.0 || nx == 0.0 || nc == 0.0)
        return log_binom_half(x, K) + log_binom_half(nx, L) - log_binom_half(n, M);
    const double hK = 0.5 * K, hL = 0.5 * L;
    double s = stirlerr(K) + stirlerr(L) + 2.0 * stirlerr(n) - stirlerr(M);
    s -= stirlerr(x) + stirlerr(c) + stirlerr(nx) + stirlerr(nc);
    s -= bd0(x, hK) + bd0(c, hK) + bd0(nx, hL) + bd0(nc, hL);
    return s + 0.5 * log((K * L * n * n) / (TWO_PI * x * c * nx * nc * M));
}

// P(X = x) for K = (row sums' first column total) <= 64 by direct products — the common case (a noise codon
// seen a few dozen times against an expected count of a few): ~2K multiplications and K/16 + 1 divisions
// instead of the logs, divisions and series of the saddle-point form.  Relative error <= ~K * 1e-16.
//   P = C(K, x) * prod_{i<x} (n - i) * prod_{i<K-x} (n - i) / prod_{i<K} (2n - i)
JL_FHD double jl_pmf_equal_rows_small(double x, double K, double n)
{
    const double c = K - x;
    const double M = 2.0 * n;
    double num = 1.0, den = 1.0, p = 1.0;
    int pending = 0;
    for (double i = 0.0; i < K; i += 1.0) {
        num *= i < x ? n - i : n - (i - x);
        den *= M - i;
        if (++pending == 16) { p *= num / den; num = 1.0; den = 1.0; pending = 0; }  // keep both below 1e160
    }
    // binomial coefficient C(K, min(x, c)): at most 32 factors, value < 2^64
    const double m = x < c ? x : c;
    for (double i = 0.0; i < m; i += 1.0) { num *= K - i; den *= i + 1.0; }
    return p * (num / den);
}

// P(X >= a) for the table [[a, n-a], [c, n-c]] (both rows sum to n); returns p, *logp = ln p
JL_FHD double jl_fisher_greater_equal_rows(uint32_t a_, uint32_t c_, uint32_t n_, double *logp)
{
    const double a = a_, c = c_, n = n_;
    const double K = a + c;
    const double hi = K < n ? K : n;
    const double lo = K > n ? K - n : 0.0;
    if (a <= lo) { *logp = 0.0; return 1.0; }
    if (K <= 64.0) {  // direct products (no log/exp until the very end)
        if (a > c) {
            double term = 1.0, sum = 1.0;
            for (double x = a; x < hi; x += 1.0) {
                term *= ((K - x) * (n - x)) / ((x + 1.0) * (n - K + x + 1.0));
                sum += term;
            }
            double pv = jl_pmf_equal_rows_small(a, K, n) * sum;
            if (pv > 1.0) pv = 1.0;
            *logp = log(pv);
            return pv;
        }
        const double x0 = a - 1.0;
        double term = 1.0, sum = 1.0;
        for (double x = x0; x > lo; x -= 1.0) {
            term *= (x * (n - K + x)) / ((K - x + 1.0) * (n - x + 1.0));
            sum += term;
        }
        double lower = jl_pmf_equal_rows_small(x0, K, n) * sum;
        if (lower > 1.0) lower = 1.0;
        *logp = lower < 1.0 ? log1p(-lower) : -HUGE_VAL;
        return 1.0 - lower;
    }
    if (a > c) {  // above the mean K/2: sum the decreasing upper tail
        const double l0 = jl_log_pmf_equal_rows(a, K, n);
        double term = 1.0, sum = 1.0;
        for (double x = a; x < hi; x += 1.0) {
            term *= ((K - x) * (n - x)) / ((x + 1.0) * (n - K + x + 1.0));
            sum += term;
            if (term < sum * 1e-17) break;
        }
        const double lp = l0 + log(sum);
        *logp = lp < 0.0 ? lp : 0.0;
        return lp < -745.0 ? 0.0 : (lp < 0.0 ? exp(lp) : 1.0);
    }
    // at or below the mean: 1 - P(X <= a-1), lower tail summed downwards
    const double x0 = a - 1.0;
    const double l0 = jl_log_pmf_equal_rows(x0, K, n);
    double term = 1.0, sum = 1.0;
    for (double x = x0; x > lo; x -= 1.0) {
        term *= (x * (n - K + x)) / ((K - x + 1.0) * (n - x + 1.0));
        sum += term;
        if (term < sum * 1e-17) break;
    }
    double lower = exp(l0 + log(sum));
    if (lower > 1.0) lower = 1.0;
    const double p = 1.0 - lower;
    *logp = lower < 1.0 ? log1p(-lower) : -HUGE_VAL;
    return p;
}

// Two-sided p (sum of the probabilities of all tables no more likely than the observed one) for the same table.
// X ~ Hypergeometric(2n, K, n) is symmetric about K/2 when both rows sum to n: P(X = x) = P(X = K - x).  The tables
// at least as extreme as a are therefore x >= max(a, c) and their mirror images x <= min(a, c): p = 2 P(X >= max(a, c)),
// and p = 1 when a = c (SPEC §5, `tail` = 1).
JL_FHD double jl_fisher_two_sided_equal_rows(uint32_t a_, uint32_t c_, uint32_t n_, double *logp)
{
    if (a_ == c_) { *logp = 0.0; return 1.0; }
    const uint32_t hi = a_ > c_ ? a_ : c_, lo = a_ > c_ ? c_ : a_;
    double l1;
    const double p1 = jl_fisher_greater_equal_rows(hi, lo, n_, &l1);
    const double lp = l1 + 0.6931471805599453094;
    *logp = lp < 0.0 ? lp : 0.0;
    const double p = 2.0 * p1;
    return p < 1.0 ? p : 1.0;
}

// The same p-value, unless it provably cannot lead to a call: P(X >= a) >= P(X = a), so when already the point mass,
// Bonferroni-adjusted, reaches alpha the codon is not called and (uncalled codons are never reported) its p-value is
// not needed — the tail sum, a division per term, is skipped and *skipped set.  Every value that IS returned comes from
// the very expressions of jl_fisher_greater_equal_rows (the product pmf * sum, sum >= 1, is monotone in floating point
// as well; the log/exp branch keeps a 1e-6 margin against a non-monotone last bit of exp).
JL_FHD double jl_fisher_greater_equal_rows_or_skip(uint32_t a_, uint32_t c_, uint32_t n_, double n_tests, double alpha,
                                                   double *logp, bool *skipped)
{
    *skipped = false;
    if (a_ <= c_) return jl_fisher_greater_equal_rows(a_, c_, n_, logp);
    const double a = a_, c = c_, n = n_;
    const double K = a + c;
    const double hi = K < n ? K : n;
    const double lo = K > n ? K - n : 0.0;
    if (a <= lo) { *logp = 0.0; return 1.0; }
    const double gate = alpha * (1.0 + 1e-6);
    if (K <= 64.0) {
        const double pm = jl_pmf_equal_rows_small(a, K, n);
        double adj = pm * n_tests;
        if (adj > 1.0) adj = 1.0;
        if (adj >= gate) { *skipped = true; *logp = 0.0; return 1.0; }
        double term = 1.0, sum = 1.0;
        for (double x = a; x < hi; x += 1.0) {
            term *= ((K - x) * (n - x)) / ((x + 1.0) * (n - K + x + 1.0));
            sum += term;
        }
        double pv = pm * sum;
        if (pv > 1.0) pv = 1.0;
        *logp = log(pv);
        return pv;
    }
    const double l0 = jl_log_pmf_equal_rows(a, K, n);
    if (l0 > -700.0) {
        double adj = exp(l0) * n_tests;
        if (adj > 1.0) adj = 1.0;
        if (adj >= gate) { *skipped = true; *logp = 0.0; return 1.0; }
    }
    double term = 1.0, sum = 1.0;
    for (double x = a; x < hi; x += 1.0) {
        term *= ((K - x) * (n - x)) / ((x + 1.0) * (n - K + x + 1.0));
        sum += term;
        if (term < sum * 1e-17) break;
    }
    const double lp = l0 + log(sum);
    *logp = lp < 0.0 ? lp : 0.0;
    return lp < -745.0 ? 0.0 : (lp < 0.0 ? exp(lp) : 1.0);
}

