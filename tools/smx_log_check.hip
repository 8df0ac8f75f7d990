// Host evaluation of the scorers' table-driven logarithm (theta_amd/csrc/smx_log.hpp) for tests/test_smx_log_cpu.py.
//   hipcc -O2 -std=c++17 -fPIC -shared --offload-arch=gfx950 -ffp-contract=off tools/smx_log_check.hip -o build_ab/libsmx_log_check.so
#include "../theta_amd/csrc/smx_log.hpp"

static const unsigned long long host_table[256] = {
#include "../theta_amd/csrc/smx_log_table.inc"
};

extern "C" void smx_log_eval(const double *x, double *y, int n) {
    for (int i = 0; i < n; i++) y[i] = smx_log(x[i], (const double2 *)host_table);
}
