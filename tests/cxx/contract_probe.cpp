// contract_probe.cpp -- the CLAHE interpolation expressions in their natural C++ form (the shape they have in OpenCV 4.4's
// clahe.cpp, restated), to be compiled TWICE by tests/test_oracle.py: with -ffp-contract=off and with -mfma -ffp-contract=fast.
// The second build shows what GCC's FMA contraction does to them (the same target-independent pass runs for aarch64, the
// reference's platform); the test checks that the oracle's explicit fmaf() pattern reproduces it bit for bit.
// Reads "x inv l11 l12 l21 l22 xa ya" lines (xa, ya in [0,1)), prints the raw bits of txf and res.
#include <cstdint>
#include <cstdio>
#include <cstring>

__attribute__((noinline)) float f_txf(int x, float inv_tw) { return x * inv_tw - 0.5f; }
__attribute__((noinline)) float f_res(unsigned char l11, unsigned char l12, unsigned char l21, unsigned char l22, float xa, float xa1, float ya, float ya1)
{
    float res = (l11 * xa1 + l12 * xa) * ya1 + (l21 * xa1 + l22 * xa) * ya;
    return res;
}
static uint32_t bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

int main()
{
    int x, l11, l12, l21, l22;
    float inv, xa, ya;
    while (scanf("%d %a %d %d %d %d %a %a", &x, &inv, &l11, &l12, &l21, &l22, &xa, &ya) == 8) {
        const float xa1 = 1.0f - xa, ya1 = 1.0f - ya;
        printf("%08x %08x\n", bits(f_txf(x, inv)), bits(f_res((unsigned char)l11, (unsigned char)l12, (unsigned char)l21, (unsigned char)l22, xa, xa1, ya, ya1)));
    }
    return 0;
}
