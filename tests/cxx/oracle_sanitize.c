/* oracle_sanitize.c -- runs the CPU oracle under AddressSanitizer + UBSan on ragged shapes (CPU build only; GPU
 * sanitizers are not available on this pool).  Catches out-of-bounds reads in the REFLECT_101 / tile / stride logic
 * that a parity test would only see as garbage.  Built and run by tests/test_oracle.py::test_oracle_under_sanitizers. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int orc_equalize_hist_u8(const uint8_t*, size_t, uint8_t*, size_t, int, int);
int orc_clahe_u8(const uint8_t*, size_t, uint8_t*, size_t, int, int, double, int, int);
int orc_clahe_u16(const uint16_t*, size_t, uint16_t*, size_t, int, int, double, int, int);
int orc_nv12_frame(const uint8_t*, uint8_t*, int, int, int, int, double, int, int);
int orc_bgr_luma_op(const uint8_t*, uint8_t*, int, int, int, double, int, int);

static uint64_t s = 0x9E3779B97F4A7C15ull;
static uint32_t rnd(void) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 32); }

int main(void)
{
    int fails = 0;
    for (int it = 0; it < 120; ++it) {
        const int w = 1 + rnd() % 70, h = 1 + rnd() % 50, pad = rnd() % 9;
        const size_t step = (size_t)w + pad;
        /* exact-size heap blocks so that any over-read trips ASan */
        uint8_t* src = malloc(step * (h - 1) + w);
        uint8_t* dst = malloc(step * (h - 1) + w);
        for (size_t i = 0; i < step * (h - 1) + w; ++i) src[i] = (uint8_t)rnd();
        fails += orc_equalize_hist_u8(src, step, dst, step, w, h) != 0;
        const int tx = 1 + rnd() % 12, ty = 1 + rnd() % 12;
        fails += orc_clahe_u8(src, step, dst, step, w, h, (rnd() % 5) * 0.9, tx, ty) != 0;
        fails += orc_clahe_u8(src, step, src, step, w, h, 2.0, tx, ty) != 0;      /* in place */
        free(src); free(dst);
        if (it % 6 == 0) {
            uint16_t* s16 = malloc((size_t)w * h * 2);
            uint16_t* d16 = malloc((size_t)w * h * 2);
            for (int i = 0; i < w * h; ++i) s16[i] = (uint16_t)rnd();
            fails += orc_clahe_u16(s16, (size_t)w * 2, d16, (size_t)w * 2, w, h, 2.0, 1 + rnd() % 5, 1 + rnd() % 5) != 0;
            free(s16); free(d16);
            const size_t fb = (size_t)w * h + (size_t)w * h / 2;
            uint8_t* f = malloc(fb);
            uint8_t* o = malloc(fb);
            for (size_t i = 0; i < fb; ++i) f[i] = (uint8_t)rnd();
            fails += orc_nv12_frame(f, o, w, h, it & 1, (it >> 1) & 1, 2.0, tx, ty) != 0;
            free(f); free(o);
            uint8_t* bgr = malloc((size_t)w * h * 3);
            uint8_t* ob = malloc((size_t)w * h * 3);
            for (size_t i = 0; i < (size_t)w * h * 3; ++i) bgr[i] = (uint8_t)rnd();
            fails += orc_bgr_luma_op(bgr, ob, w, h, it & 1, 3.0, 4, 4) != 0;
            free(bgr); free(ob);
        }
    }
    printf(fails ? "oracle_sanitize: %d failing calls\n" : "oracle_sanitize: clean\n", fails);
    return fails != 0;
}
