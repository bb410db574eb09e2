// format.hpp — how numbers are DISPLAYED (HTML), pinned by the reference's screenshots (tests/golden/appendix_a.json).
#pragma once
#include <cmath>
#include <cstdio>
#include <string>

namespace jlhost {

// Variant percentage: two significant digits, truncated, no trailing zeros ("0.91", "1.1", "1", "99", "100").
// 12 of the 66 printed rows cannot come from rounding to nearest (0.91 % of 2946 reads: 27 reads are 0.9165 %),
// every row is reproduced by truncation (docs/SPEC.md §6).
inline std::string format_percent(double x)
{
    if (!(x > 0.0)) return "0";
    const double e = std::floor(std::log10(x));
    const double f = std::pow(10.0, e - 1.0);
    const double v = std::floor(x / f + 1e-9) * f;
    char buf[48];
    snprintf(buf, sizeof buf, "%.10g", v);
    return buf;
}

// Haplotype percentage: one decimal, rounded, trailing zero dropped ("92.5", "1.2", "1"); the printed columns
// sum to 100.0 (juliet_hiv-phasing.png, juliet_major-after.png).
inline std::string format_hap_percent(double x)
{
    char buf[48];
    snprintf(buf, sizeof buf, "%g", std::round(x * 10.0 + 1e-9) / 10.0);
    return buf;
}

}  // namespace jlhost
