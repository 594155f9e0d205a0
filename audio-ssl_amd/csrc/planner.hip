// Host-side (CPU) augmentation planner: draws every random parameter of a batch from the SAME generators, in the
// SAME order, as the reference's per-clip Python chain, and fills the tables the device kernels consume.
//
// The reference draws from two Mersenne Twisters (SURVEY a8'):
//   numpy legacy global RandomState  - MixupBYOLA: np.random.random(), np.random.randint(len(bank))
//                                      (`src/augmentations/augmentations.py:99-102`)
//                                    - RandomResizeCrop.get_params: np.random.uniform x2 (`:32-37`)
//   python `random`                  - RandomResizeCrop.get_params: random.randint x2 (`:36-37`)
//                                    - SpecAugment: random.randrange x3 per mask (`extras/delores-s/specaugment.py:80-88`)
// Both are MT19937 with the standard tempering; this file continues their 624-word states exactly:
//   numpy  random()/uniform : (a>>5, b>>6) 53-bit double, lower + range * u
//   numpy  randint(n)       : masked rejection on 32-bit draws (legacy use_masked=True), no draw when n == 1
//   python randint/randrange: _randbelow(n) = getrandbits(n.bit_length()) rejection, getrandbits(k<=32) = word >> (32-k)
// The Python planner costs ~19 ms per 512-clip batch; this one ~0.1 ms, which keeps the GPU step fed.
// No HIP calls here; pointers are HOST pointers.
#include <stdint.h>
#include "common.h"

namespace {

struct MT {
    uint32_t* key;   // 624 words, caller-owned
    int pos;         // 0..624 (624 = regenerate before the next draw)
    void regen() {
        const uint32_t UP = 0x80000000u, LO = 0x7fffffffu, MA = 0x9908b0dfu;
        int i;
        uint32_t y;
        for (i = 0; i < 624 - 397; ++i) {
            y = (key[i] & UP) | (key[i + 1] & LO);
            key[i] = key[i + 397] ^ (y >> 1) ^ ((y & 1u) ? MA : 0u);
        }
        for (; i < 623; ++i) {
            y = (key[i] & UP) | (key[i + 1] & LO);
            key[i] = key[i + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? MA : 0u);
        }
        y = (key[623] & UP) | (key[0] & LO);
        key[623] = key[396] ^ (y >> 1) ^ ((y & 1u) ? MA : 0u);
        pos = 0;
    }
    uint32_t next() {
        if (pos >= 624) regen();
        uint32_t y = key[pos++];
        y ^= (y >> 11);
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= (y >> 18);
        return y;
    }
};

inline double np_double(MT& g) {
    const int32_t a = (int32_t)(g.next() >> 5), b = (int32_t)(g.next() >> 6);
    return (a * 67108864.0 + b) / 9007199254740992.0;
}
inline uint32_t np_randint(MT& g, uint32_t n) {          // np.random.randint(n), n >= 1
    const uint32_t rng = n - 1;
    if (rng == 0) return 0;
    uint32_t mask = rng;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
    uint32_t v;
    do { v = g.next() & mask; } while (v > rng);
    return v;
}
inline uint32_t py_randbelow(MT& g, uint32_t n) {         // random._randbelow(n), 1 <= n < 2^32
    int k = 0;
    for (uint32_t t = n; t; t >>= 1) ++k;                 // n.bit_length()
    uint32_t r;
    do { r = g.next() >> (32 - k); } while (r >= n);
    return r;
}
inline int clipi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

}  // namespace

// ip [B][2][8] = {self_slot, partner_slot|-1, i, j, h, w, do_rrc, 0}; fp [B][2][2] = {coef_self, coef_partner};
// masks [B][2][max_masks][4] = {axis, start, end, 0} (axis -1 = unused), max_masks = spec_nf + spec_nt.
// lens/unit/starts (optional): clip lengths in samples, window length, and the drawn crop starts (per clip, BEFORE
// that clip's view draws - the reference's per-clip interleaving of the python stream).
extern "C" int audiossl_aug_plan_host(uint32_t* np_key, int* np_pos, uint32_t* py_key, int* py_pos, int B, int F, int T,
                                      long long clips_seen, long long* n_entries, int R, int n_memory, int use_mix,
                                      double ratio, int use_rrc, double fs_lo, double fs_hi, double ts_lo, double ts_hi,
                                      int canvas_h, int canvas_w, int use_spec, int spec_F, int spec_T, int spec_nf,
                                      int spec_nt, const int* lens, int unit, int* starts, int* ip, float* fp, int* masks) {
    ASSL_REQUIRE(np_key && np_pos && py_key && py_pos && n_entries && ip && fp && B > 0 && R > 0);
    ASSL_REQUIRE(*np_pos >= 0 && *np_pos <= 624 && *py_pos >= 0 && *py_pos <= 624);
    ASSL_REQUIRE(!use_spec || masks);
    ASSL_REQUIRE(!lens || (starts && unit > 0));
    MT npg{np_key, *np_pos}, pyg{py_key, *py_pos};
    long long ne = *n_entries;
    const int max_masks = spec_nf + spec_nt;
    for (int b = 0; b < B; ++b) {
        const long long c = clips_seen + b;
        if (lens) {       // extract_window (`src/utils/utils.py:166-182`): one python draw, only when the clip is longer
            const int over = lens[b] - unit;
            starts[b] = over > 0 ? (int)py_randbelow(pyg, (uint32_t)(over + 1)) : 0;
        }
        for (int v = 0; v < 2; ++v) {
            int* e = ip + ((long)b * 2 + v) * 8;
            float* f = fp + ((long)b * 2 + v) * 2;
            for (int k = 0; k < 8; ++k) e[k] = 0;
            f[0] = f[1] = 0.f;
            e[0] = (int)(c % R);
            e[1] = -1;
            if (use_mix) {
                const double alpha = ratio * np_double(npg);
                const long long n_bank = ne < n_memory ? ne : n_memory;
                if (n_bank > 0) {
                    const long long g = ne - n_bank + np_randint(npg, (uint32_t)n_bank);
                    e[1] = (int)((g / 2) % R);
                    const double a1 = 1.0 - alpha;
                    f[0] = (float)a1;
                    f[1] = (float)(1.0 - a1);
                }
                ++ne;
            }
            if (use_rrc) {
                const int h = clipi((int)((fs_lo + (fs_hi - fs_lo) * np_double(npg)) * F), 1, canvas_h);
                const int w = clipi((int)((ts_lo + (ts_hi - ts_lo) * np_double(npg)) * T), 1, canvas_w);
                const int i = canvas_h > h ? (int)py_randbelow(pyg, (uint32_t)(canvas_h - h + 1)) : 0;
                const int j = canvas_w > w ? (int)py_randbelow(pyg, (uint32_t)(canvas_w - w + 1)) : 0;
                e[2] = i; e[3] = j; e[4] = h; e[5] = w; e[6] = 1;
            }
            if (use_spec) {
                int* m = masks + ((long)b * 2 + v) * max_masks * 4;
                for (int k = 0; k < max_masks * 4; ++k) m[k] = -1;
                int out = 0;
                for (int fam = 0; fam < 2; ++fam) {
                    const int axis = fam == 0 ? 1 : 0, width = fam == 0 ? spec_F : spec_T, size = fam == 0 ? F : T;
                    const int count = fam == 0 ? spec_nf : spec_nt;
                    for (int k = 0; k < count; ++k) {
                        const int fw = (int)py_randbelow(pyg, (uint32_t)width);
                        ASSL_REQUIRE(size - fw > 0);
                        const int f0 = (int)py_randbelow(pyg, (uint32_t)(size - fw));
                        if (fw == 0) break;
                        const int end = f0 + (int)py_randbelow(pyg, (uint32_t)fw);
                        m[out * 4 + 0] = axis; m[out * 4 + 1] = f0; m[out * 4 + 2] = end; m[out * 4 + 3] = 0;
                        ++out;
                    }
                }
            }
        }
    }
    *np_pos = npg.pos;
    *py_pos = pyg.pos;
    *n_entries = ne;
    return ASSL_OK;
}
