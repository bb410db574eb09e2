// Host build of minorseq_amd/host/format.hpp for tests/test_oracle_golden_rows.py.  Not part of the product.
#include <cstring>

#include "../../minorseq_amd/host/format.hpp"
extern "C" void shim_format_percent(double x, char *out) { std::strcpy(out, jlhost::format_percent(x).c_str()); }
extern "C" void shim_format_hap_percent(double x, char *out) { std::strcpy(out, jlhost::format_hap_percent(x).c_str()); }
