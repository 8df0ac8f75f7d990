// CPU execution of the n=2 "render" generator (theta_amd/csrc/n2_render.hpp): the kernel body of n2_enumerate_render_kernel,
// run lane by lane, wave by wave, with the per-wave LDS tile as an ordinary array -- the per-lane code is the very code the
// kernel compiles (N2_HD), the two wave_lds_sync() points become "all 64 lanes finish the phase before the next one starts".
// Test infrastructure (tests/test_n2_render_cpu.py); links the product library only for n2_build_host, its table builder.
//   hipcc -O2 -std=c++17 -fPIC -shared --offload-arch=gfx950 tools/n2_render_emul.hip -Ltheta_amd -ltheta_hip -o build_ab/libn2_emul.so
#include <vector>
#include <cstring>

#include "../theta_amd/csrc/n2_render.hpp"

template <int KV>
static void emulate(const N2Dev &P, unsigned long long begin, unsigned long long count, int T, unsigned char *out) {
    const int m = P.m;
    const unsigned long long threads = (count + T - 1) / T;
    const unsigned long long waves = (threads + 63) / 64;
    const int lines = (int)(((unsigned long long)T * m) >> 7);
    std::vector<unsigned> tile((size_t)N2R_TILE_DWORDS);
    std::vector<N2Run<KV>> R(64);
    std::vector<N2RStore> S(64);
    short ubp[KV + 1];
    for (int w = 0; w <= KV; w++) ubp[w] = n2r_ubpos(P.ub, m, w);
    for (unsigned long long w = 0; w < waves; w++) {
        const unsigned long long wave_first = w * 64;
        for (int lane = 0; lane < 64; lane++) {
            const unsigned long long tid = wave_first + lane, k0 = tid * (unsigned long long)T;
            const unsigned long long mine = k0 < count ? (count - k0 < (unsigned long long)T ? count - k0 : (unsigned long long)T) : 0;
            n2r_begin<KV>(R[lane], mine);
            if (mine) n2_unrank<KV>(P, P.P, begin + k0, R[lane].c);
            else
                for (int v = 0; v <= KV; v++) R[lane].c.s[v] = m;
            n2r_store_prepare(lane, wave_first, T, m, count, out, S[lane]);
        }
        for (int line = 0; line < lines; line++) {
            for (int lane = 0; lane < 64; lane++) {
                n2r_scatter_line<KV>(m, P.lbpos, ubp, R[lane], tile.data(), lane);
                n2r_prefix_line(tile.data(), lane);
            }
            for (int lane = 0; lane < 64; lane++) n2r_store_line(lane, line, S[lane], tile.data());
        }
    }
}

// out: count * m bytes (+ nothing beyond is written).  Returns 0, or a THETA_ERR code from the table builder; *total = size of the space.
extern "C" int n2_emul_enumerate(int m, const int32_t *lb, const int32_t *ub, unsigned long long begin, unsigned long long count,
                                 int T, unsigned char *out, unsigned long long *total) {
    N2Host h;
    int rc = n2_build_host(m, lb, ub, h);
    if (rc) return rc;
    if (total) *total = h.total;
    std::vector<unsigned char> lbb(m), ubb(m);
    std::vector<short> lbpos(N2_KVS + 1);
    for (int i = 0; i < m; i++) {
        lbb[i] = (unsigned char)h.lb[i];
        ubb[i] = (unsigned char)h.ub[i];
    }
    for (int v = 0; v <= N2_KVS; v++) {
        int pos = m;
        for (int i = 0; i < m; i++)
            if (h.lb[i] >= v) {
                pos = i;
                break;
            }
        lbpos[v] = (short)pos;
    }
    N2Dev P;
    memset(&P, 0, sizeof(P));
    P.m = m;
    P.kv = h.kv;
    P.tau = 2;
    P.P = h.P.data();
    P.lb = lbb.data();
    P.ub = ubb.data();
    P.lbpos = lbpos.data();
    P.total = h.total;
    if (count == 0) return 0;
    if (begin + count > h.total) return -1;
    if (T <= 0) {                                  // the launcher's choice (n2_launch_enumerate)
        int g = 128, a = m;
        while (a) { const int t = g % a; g = a; a = t; }
        T = 128 / g;
        while (T < 32) T *= 2;
    }
    if (((unsigned long long)T * m) % 128) return -2;
    if (h.kv <= 8) emulate<8>(P, begin, count, T, out);
    else emulate<16>(P, begin, count, T, out);
    return 0;
}
