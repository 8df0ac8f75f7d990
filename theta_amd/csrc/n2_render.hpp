// n = 2 materialised generator, "render" form: a lane's run of records is produced 128 bytes (one output line) at a time by
// SCATTER + PREFIX SUM instead of summing all break-points into every word.
//
// A record is a staircase: byte i = number of break-points s[v] (v = 1..KV-1, non-decreasing in v) at or before i.  The bytes
// of consecutive records of a run, laid end to end, are therefore the running sum of a sparse increment string: +1 at every
// break-point, and minus the record's last value where the next record begins.  Per line a lane
//   1. zeroes its 128-byte row of the per-wave LDS tile,
//   2. writes the increments of the records that intersect the line -- one byte store per DISTINCT position (equal positions
//      are merged in registers: break-points come sorted; increments that fall before the line collapse onto its byte 0,
//      which is what carries a record over a line boundary),
//   3. turns the row into running sums, word by word (two shift-adds inside the word, the previous word's top byte carried in).
//      The sums are taken MOD 16 in every byte (mask 0x0f0f0f0f after each add): copy numbers are at most 15, so the low nibble
//      is the value itself, a "minus last" increment is stored as (16 - last) & 15, and no add can carry from one byte into
//      its neighbour (plain 32-bit adds of 0xff-style negative bytes would).
// Cost: ~7 vector instructions per break-point and ~7 per word, against one pass over ALL break-points per word (35
// instructions per word at KV = 8) in n2_enumerate_lines_kernel.  The record boundaries inside a line are the same for all
// lanes of a wave (equal run lengths, runs start on line boundaries), so the control flow is wave-uniform; only positions and
// store predicates differ between lanes.
//
// Everything here is host + device (N2_HD): tools/n2_render_emul.hip executes the same code lane by lane on the CPU.
#pragma once
#include "n2_cand.hpp"

#define N2R_LINE 128       // bytes per output line
#define N2R_WORDS 32       // dwords of payload per LDS row
// The per-wave LDS tile is kept WORD-MAJOR: word w of lane l's row lives at dword w * N2R_PITCH + l.  Every access of steps
// 1-3 is "all lanes, (nearly) the same w": with the rows laid lane-major (stride 36 dwords, round 2) those hit 8 banks 8
// deep -- PMC: 2.9 bank-conflict cycles per active LDS cycle, the LDS busy for more than half of the kernel -- here they
// fall on consecutive banks.  The odd pitch makes the store stage's transposed read (lane (r8, ch) reads words 4ch..4ch+3
// of row r8 + 8 sidx: bank = 4 ch + j + r mod 32) conflict-free per half-wave as well.
#define N2R_PITCH 65
#define N2R_TILE_DWORDS (N2R_WORDS * N2R_PITCH)
N2_HD inline unsigned &n2r_word(unsigned *tile, int lane, int w) { return tile[w * N2R_PITCH + lane]; }
N2_HD inline const unsigned &n2r_word(const unsigned *tile, int lane, int w) { return tile[w * N2R_PITCH + lane]; }
N2_HD inline unsigned char &n2r_byte(unsigned *tile, int lane, int pos) {
    return ((unsigned char *)(tile + (pos >> 2) * N2R_PITCH + lane))[pos & 3];
}

template <int KV>
struct N2Run {
    N2Cand<KV> c;                 // current record
    unsigned long long left;      // records still to produce, the current one included (0: the run is exhausted)
    int recstart;                 // byte offset of the current record's first byte relative to the current line (<= 0 .. 127)
    int reset;                    // increment pending at the current record's first byte: minus the previous record's last value
};

template <int KV>
N2_HD inline void n2r_begin(N2Run<KV> &R, unsigned long long records) {
    R.left = records;
    R.recstart = 0;
    R.reset = 0;
}

// The reference's successor (Enumerator.py:134-152; n2_next of n2_cand.hpp) with both bound tests on wave-uniform tables:
//   lbp[w] = first position whose lower bound is >= w (m if none),  ubp[w] = first position whose upper bound is >= w (m if none).
// Run v = [s[v], s[v+1]) can be raised at its end iff that end lies at or behind ubp[v+1] (the bounds are non-decreasing after
// _check_bound_order), i.e. iff s[v+1] > max(s[v], ubp[v+1]) -- no per-lane bound reads; the first such run wins, and if there
// is none the enumeration is exhausted (a blocked run ending at m-1 is the last one).  Then every break-point up to the raised
// value moves to min(lbp[w], e).
template <int KV>
N2_HD inline bool n2r_next(const short *lbp, const short *ubp, N2Cand<KV> &c) {
    int e = -1, nv = 0;
    bool found = false;
#pragma unroll
    for (int v = 0; v < KV; v++) {
        const int hi = c.s[v + 1], b = (int)ubp[v + 1];
        const int lo = c.s[v] > b ? c.s[v] : b;
        const bool take = hi > lo && !found;
        e = take ? hi - 1 : e;
        nv = take ? v + 1 : nv;
        found = found || hi > lo;
    }
    if (!found) return false;
#pragma unroll
    for (int w = 1; w < KV; w++) {
        const int lp = (int)lbp[w];
        c.s[w] = w <= nv ? (lp < e ? lp : e) : c.s[w];
    }
    return true;
}

// ubp[0..KV] from the (non-decreasing) upper bounds: ubp[w] = number of positions whose bound is below w
N2_HD inline short n2r_ubpos(const unsigned char *ub, int m, int w) {
    int lo = 0, hi = m;                                           // first i with ub[i] >= w
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if ((int)ub[mid] >= w) hi = mid;
        else lo = mid + 1;
    }
    return (short)lo;
}

// the record after the current one (a run that ends early -- the last run of the range, or the end of the enumeration --
// continues with "null" records: no break-points, so its bytes come out as zeros; they are never stored)
template <int KV>
N2_HD inline void n2r_advance(const short *lbp, const short *ubp, N2Run<KV> &R) {
    if (R.left > 0) {
        R.left--;
        if (R.left > 0 && !n2r_next<KV>(lbp, ubp, R.c)) R.left = 0;
    }
}

// Steps 1 and 2 for one line.  `tile` is the wave's tile, `lane` selects the row.
template <int KV>
N2_HD inline void n2r_scatter_line(int m, const short *lbp, const short *ubp, N2Run<KV> &R, unsigned *tile, int lane) {
#pragma unroll
    for (int w = 0; w < N2R_WORDS; w++) n2r_word(tile, lane, w) = 0u;
    while (R.recstart < N2R_LINE) {                              // (uniform: recstart and m are the same in every lane)
        const bool live = R.left > 0;
        int curpos = R.recstart > 0 ? R.recstart : 0;
        int cnt = R.recstart >= 0 ? R.reset : 0;                  // (a record that began before this line: its reset is behind us)
        int last = 0;                                             // the record's last value: break-points inside the record
#pragma unroll
        for (int v = 1; v < KV; v++) {
            const int sv = R.c.s[v];
            const bool inrec = live && sv < m;
            last += inrec ? 1 : 0;
            int p = R.recstart + sv;
            p = p < 0 ? 0 : p;
            const bool act = inrec && p < N2R_LINE;
            const bool same = !act || p == curpos;
            if (!same) n2r_byte(tile, lane, curpos) = (unsigned char)(cnt & 15); // (predicated byte store: positions are visited in order)
            cnt = same ? cnt + (act ? 1 : 0) : 1;
            curpos = same ? curpos : p;
        }
        n2r_byte(tile, lane, curpos) = (unsigned char)(cnt & 15);
        const int e = R.recstart + m;
        if (e > N2R_LINE) break;                                  // the record continues in the next line
        n2r_advance<KV>(lbp, ubp, R);
        R.recstart = e;                                           // (e == 128: the next record opens the next line)
        R.reset = -last;
    }
    // next line: positions are relative to it; a record that opens it exactly needs no reset (nothing is carried over a line)
    R.recstart -= N2R_LINE;
    if (R.recstart == 0) R.reset = 0;
}

// Step 3: running sums (mod 16 per byte) over the row.
N2_HD inline void n2r_prefix_line(unsigned *tile, int lane) {
    unsigned carry = 0u;
#pragma unroll
    for (int w = 0; w < N2R_WORDS; w++) {
        unsigned x = n2r_word(tile, lane, w);                                      // every byte <= 15
        x = (x + (x << 8)) & 0x0f0f0f0fu;
        x = (x + (x << 16)) & 0x0f0f0f0fu;
        x = (x + carry * 0x01010101u) & 0x0f0f0f0fu;
        n2r_word(tile, lane, w) = x;
        carry = x >> 24;
    }
}

// Store stage of one line (after every lane of the wave has finished steps 1-3 and the tile is visible): lane (r8, ch) of
// pass `sidx` moves the 16-byte chunk `ch` of row r = r8 + 8 sidx -- one store instruction of the wave = 8 complete 128-byte
// lines.  `tile` is the wave's tile (word-major, see N2R_PITCH), wave_first the global index of the wave's first thread.
struct n2r_u4 {
    unsigned x, y, z, w;
};
// what a lane needs for its eight stores of every line, computed once per run: where row r = (lane >> 3) + 8 sidx of the tile
// goes (chunk ch = lane & 7 included) and how many bytes of that row's run exist (0: the row's run lies beyond the range)
struct N2RStore {
    unsigned char *dst[8];
    unsigned nv[8];
};
N2_HD inline void n2r_store_prepare(int lane, unsigned long long wave_first, int T, int m, unsigned long long count, unsigned char *out,
                                    N2RStore &S) {
    const unsigned long long RB = (unsigned long long)T * (unsigned long long)m;       // bytes per run, a multiple of 128
#pragma unroll
    for (int sidx = 0; sidx < 8; sidx++) {
        const int r = (lane >> 3) + 8 * sidx, ch = lane & 7;
        const unsigned long long tt = wave_first + (unsigned long long)r;
        const unsigned long long kk = tt * (unsigned long long)T;
        S.nv[sidx] = kk < count ? (unsigned)((count - kk < (unsigned long long)T ? count - kk : (unsigned long long)T) * (unsigned long long)m) : 0u;
        S.dst[sidx] = out + tt * RB + (unsigned long long)ch * 16;
    }
}
N2_HD inline void n2r_store_line(int lane, int line, const N2RStore &S, const unsigned *tile) {
    const unsigned off = ((unsigned)line << 7) + (unsigned)(lane & 7) * 16u;             // byte offset of this lane's chunk in its row's run
#pragma unroll
    for (int sidx = 0; sidx < 8; sidx++) {
        const unsigned nv = S.nv[sidx];
        if (off < nv) {
            const int r = (lane >> 3) + 8 * sidx, w0 = 4 * (lane & 7);
            n2r_u4 v;
            v.x = n2r_word(tile, r, w0);
            v.y = n2r_word(tile, r, w0 + 1);
            v.z = n2r_word(tile, r, w0 + 2);
            v.w = n2r_word(tile, r, w0 + 3);
            unsigned char *dst = S.dst[sidx] + ((size_t)line << 7);
            if (off + 16 <= nv) {
                *(n2r_u4 *)dst = v;
            } else {                                             // the tail of the very last record
                const unsigned w4[4] = {v.x, v.y, v.z, v.w};
                for (int bidx = 0; bidx < (int)(nv - off); bidx++) dst[bidx] = (unsigned char)(w4[bidx >> 2] >> (8 * (bidx & 3)));
            }
        }
    }
}
